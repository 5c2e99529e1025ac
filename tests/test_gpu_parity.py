"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle in counter mode.
Bit-exact: integer / byte work.  Run with `-m gpu` on the MI355X box."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, seams_env

import scssim_amd

pytestmark = pytest.mark.gpu

MODELS = ["Illumina_HiSeq2500", "Illumina_HiSeqXTen", "Illumina_HiSeq2000", "Illumina_GenomeAnalyzerIIx"]


def _oracle_run(oracle_bin, fasta, profile, prefix, args, seed, dump=None, threads=8):
    cmd = [oracle_bin, "genreads", "-i", fasta, "-m", profile, "-o", prefix, "--rng", "counter", "--seed", str(seed),
           "-t", str(threads), "-q"] + args
    if dump:
        cmd += ["--dump", dump]
    subprocess.check_call(cmd)


def test_philox_device_matches_oracle(oracle_lib):
    g = scssim_amd.GenReads()
    rng = np.random.default_rng(1)
    ctr = rng.integers(0, 1 << 32, size=(4096, 4), dtype=np.uint64).astype(np.uint32)
    ctr[0] = 0
    ctr[1] = 0xFFFFFFFF
    key = np.array([0xa4093822, 0x299f31d0], np.uint32)
    got = g.philox(ctr, key)
    want = np.zeros_like(ctr)
    k = (ctypes.c_uint32 * 2)(*key.tolist())
    for i in range(ctr.shape[0]):
        c = (ctypes.c_uint32 * 4)(*ctr[i].tolist())
        o = (ctypes.c_uint32 * 4)()
        oracle_lib.scso_philox4x32_10(c, k, o)
        want[i] = list(o)
    assert np.array_equal(got, want)


def test_det_log_device_bitwise(oracle_lib):
    oracle_lib.scso_det_log.restype = ctypes.c_double
    oracle_lib.scso_det_log.argtypes = [ctypes.c_double]
    g = scssim_amd.GenReads()
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.random(20000), 10 ** rng.uniform(-300, 300, 5000), [1.0, 2.0, 0.5, 1e-310, 5e-324, 1.4142135623730951]])
    got = g.det_log(x)
    want = np.array([oracle_lib.scso_det_log(float(v)) for v in x])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # det_exp (arguments <= 0 of the same entry point): the exp(-lambda) of the product-form Poisson draw
    oracle_lib.scso_det_exp.restype = ctypes.c_double
    oracle_lib.scso_det_exp.argtypes = [ctypes.c_double]
    y = -np.concatenate([rng.random(20000) * 256, rng.random(5000) * 12, [0.0, 1e-12, 0.6931471805599453, 255.999, 256.0]])
    got = g.det_log(y)
    want = np.array([oracle_lib.scso_det_exp(float(v)) for v in y])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert np.max(np.abs(got - np.exp(y)) / np.exp(y)) < 4e-16


@pytest.mark.parametrize("model", MODELS)
def test_predict_batch_matches_oracle(model, models, oracle_lib):
    """Profile::predict (Profile.cpp:1582-1697) per read: indels, substitutions, qualities, N handling."""
    oracle_lib.scso_profile_load.restype = ctypes.c_void_p
    oracle_lib.scso_profile_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    oracle_lib.scso_predict_counter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                                ctypes.c_uint64, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p]
    seed = 0x1234567890ABCDEF
    g = scssim_amd.GenReads(profile=models[model], seed=seed)
    L = g.read_length
    h = oracle_lib.scso_profile_load(models[model].encode(), 1, 260)
    rng = np.random.default_rng(7)
    n = 6000
    win = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    win[rng.random((n, L)) < 0.004] = 4                 # sprinkle N
    win[5, :] = 4                                       # all-N read
    win[6, :3] = 4                                      # N in the leading context
    uids = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
    atts = rng.integers(0, 1 << 20, size=n, dtype=np.uint64).astype(np.uint32)
    rd1 = (rng.random(n) < 0.5).astype(np.uint8)
    gb, gq = g.predict_batch(win, uids, atts, rd1)
    ob = ctypes.create_string_buffer(4 * L + 256)
    oq = ctypes.create_string_buffer(4 * L + 256)
    lens = set()
    for i in range(n):
        m = oracle_lib.scso_predict_counter(h, win[i].ctypes.data_as(ctypes.c_void_p), L, int(rd1[i]), seed, int(uids[i]), int(atts[i]), ob, oq)
        assert m > 0
        lens.add(m)
        assert gb[i] == ob.raw[:m], "bases differ for read %d" % i
        assert gq[i] == oq.raw[:m], "qualities differ for read %d" % i
    assert len(lens) > 3, "indels did not occur; the test would not exercise the prefix-sum path"


def _fastq_diff(got, want, limit=12):
    """Where two FASTQ texts differ: record number, line of the record, column (for assertion messages)."""
    if got == want:
        return ""
    out = ["sizes %d / %d" % (len(got), len(want))]
    gl, wl = got.split(b"\n"), want.split(b"\n")
    for i, (a, b) in enumerate(zip(gl, wl)):
        if a != b:
            cols = [k for k in range(min(len(a), len(b))) if a[k] != b[k]]
            out.append("record %d (%s) line %d: cols %s got %r want %r" % (i // 4, wl[4 * (i // 4)].decode(), i % 4, cols[:8], bytes(a[c] for c in cols[:8]), bytes(b[c] for c in cols[:8])))
            if len(out) > limit:
                break
    return "\n".join(out)


CASES = [("g1_hiseq2500_pe", "Illumina_HiSeq2500", ["-c", "3"], "PE", 3.0, 260, {}),
         ("g2_xten_pe_nblock", "Illumina_HiSeqXTen", ["-c", "4", "-s", "300"], "PE", 4.0, 300, {}),
         ("g3_hiseq2000_se", "Illumina_HiSeq2000", ["-c", "2", "-l", "SE"], "SE", 2.0, 260, {}),
         ("g4_gaiix_pe_lowprimers", "Illumina_GenomeAnalyzerIIx", ["-c", "2", "-p", "1000", "-r", "1e-8"], "PE", 2.0, 260,
          dict(primers=1000, gamma=1e-8))]


def _load_dump(path):
    rows = []
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        errs = [tuple(int(v) for v in e.split(":")) for e in f[7].split(",")] if len(f) > 7 and f[7] else []
        rows.append((int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6]), errs))
    return rows


@pytest.mark.parametrize("case,model,oargs,layout,cov,isize,extra", CASES)
def test_full_pipeline_fastq_bit_exact(case, model, oargs, layout, cov, isize, extra, oracle_bin, models, golden_inputs, tmp_path):
    """End to end: fragments -> MALBAC cycles -> read allocation -> FASTQ, byte-identical to the oracle;
    intermediate amplicon tables are compared too so a mismatch points at its stage."""
    seed = 20240 + len(case)
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, golden_inputs[case], models[model], prefix, oargs, seed, dump=prefix)
    g = scssim_amd.GenReads(profile=models[model], input_fasta=golden_inputs[case], coverage=cov, isize=isize, layout=layout, seed=seed, **extra)
    g.create_frags()
    g.amplify()
    for kind, name in ((0, "semis"), (1, "fulls")):
        want = _load_dump(prefix + "." + name + ".tsv")
        a = g.download_amplicons(kind)
        assert len(want) == a["parent"].size, "%s count" % name
        w = np.array([r[:6] for r in want], dtype=np.uint64).reshape(-1, 6)
        for col, key in enumerate(("parent", "spos", "len", "gc")):
            assert np.array_equal(a[key].astype(np.uint64), w[:, col]), "%s.%s" % (name, key)
        assert np.array_equal(a["uid"], w[:, 5]), name + ".uid"
        if kind == 0:
            assert np.array_equal(a["primers"].astype(np.uint64), w[:, 4]), "semis.primers (last cycle budgets)"
        for i, r in enumerate(want):
            assert a["nerr"][i] == len(r[6]), "%s[%d] error count" % (name, i)
            for k, (pos, alt) in enumerate(r[6][:4]):
                assert a["errs"][i, k] == (pos << 3 | alt), "%s[%d] error %d" % (name, i, k)
    g.allocate_reads(0)
    rn = g.download_read_numbers()
    want_rn = np.zeros_like(rn)
    for line in open(prefix + ".readnum.tsv"):
        i, v = line.split()
        want_rn[int(i)] = int(v)
    assert np.array_equal(rn, want_rn), "read allocation differs"
    fq1, fq2 = g.yield_reads()
    if layout == "PE":
        assert fq1 == open(prefix + "_1.fq", "rb").read(), _fastq_diff(fq1, open(prefix + "_1.fq", "rb").read())
        assert fq2 == open(prefix + "_2.fq", "rb").read()
    else:
        assert fq1 == open(prefix + ".fq", "rb").read()
    st = g.stats()
    assert st["pairs_written"] == fq1.count(b"\n") // 4
    assert st["fastq_bytes"][0] == len(fq1) and st["fastq_bytes"][1] == len(fq2)
    # re-running the same ctx with the same seed must reproduce the bytes (buffers are reused)
    again1, again2 = g.run()
    assert (again1, again2) == (fq1, fq2)


@pytest.mark.parametrize("model", MODELS)
def test_seed_sweep_fastq_bit_exact(model, oracle_bin, models, golden_inputs, tmp_path):
    """The same small genome under every shipped model, both layouts and several seeds and insert sizes: every run moves
    the records to other byte offsets (sector phases of the FASTQ writer), other indel patterns and other deferred
    qualities.  FASTQ must equal the oracle's byte for byte each time."""
    fa = golden_inputs["g1_hiseq2500_pe"]
    for k, (layout, isize, seed) in enumerate((("PE", 260, 11), ("PE", 301, 12), ("SE", 260, 13), ("PE", 277, 14))):
        prefix = str(tmp_path / ("o%d" % k))
        _oracle_run(oracle_bin, fa, models[model], prefix, ["-c", "2", "-s", str(isize)] + (["-l", "SE"] if layout == "SE" else []), seed)
        g = scssim_amd.GenReads(profile=models[model], input_fasta=fa, coverage=2.0, isize=isize, layout=layout, seed=seed)
        fq1, fq2 = g.run()
        if layout == "PE":
            assert fq1 == open(prefix + "_1.fq", "rb").read(), (model, layout, isize, seed)
            assert fq2 == open(prefix + "_2.fq", "rb").read(), (model, layout, isize, seed)
        else:
            assert fq1 == open(prefix + ".fq", "rb").read(), (model, layout, isize, seed)


def test_cli_drop_in(oracle_bin, models, golden_inputs, tmp_path):
    """`scssim genreads` (the reference's CLI surface) writes the same files as the oracle CLI."""
    import shutil
    exe = os.path.join(ROOT, "scssim_amd", "bin", "scssim")
    out = str(tmp_path / "cli")
    fa = str(tmp_path / "input.fa")
    shutil.copyfile(golden_inputs["g1_hiseq2500_pe"], fa)
    r = subprocess.run([exe, "genreads", "-i", fa, "-m", models["Illumina_HiSeq2500"], "-c", "2",
                        "-o", out, "--seed", "99"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "MALBAC amplification..." in r.stderr and "Reads generation done!" in r.stderr
    # the reference leaves a fastahack index beside its input (golden: written by the compiled reference)
    assert open(fa + ".fai").read() == open(os.path.join(ROOT, "tests", "golden", "g1_hiseq2500_pe.simu.fa.fai")).read()
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"], prefix, ["-c", "2"], 99)
    assert open(out + "_1.fq", "rb").read() == open(prefix + "_1.fq", "rb").read()
    assert open(out + "_2.fq", "rb").read() == open(prefix + "_2.fq", "rb").read()
    bad = subprocess.run([exe, "genreads", "-i", "x.fa", "-m", "y", "-o", "z", "-p", "10"], capture_output=True, text=True)
    assert bad.returncode == 1 and "should be at least 1000" in bad.stderr
    ver = subprocess.run([exe, "-v"], capture_output=True, text=True)
    assert "SCSsim version 1.0" in ver.stderr


def test_full_size_properties(models, tmp_path):
    """BASELINE config sized run (1 Mb, 30x): size-independent invariants instead of an oracle diff."""
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "1000000", "--seed", "1", "--simu-out", fa])
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=fa, coverage=30, seed=5)
    fq1, fq2 = g.run()
    st = g.stats()
    L = g.read_length
    assert st["reads_requested"] == 1000000 * 30 // L
    assert abs(st["pairs_written"] * 2 - st["reads_requested"]) <= 2
    assert 350000 < st["full_amplicons"] < 470000 and 35000 < st["semi_amplicons"] < 55000     # ~4.1e5 / ~4.5e4 per Mb
    l1, l2 = fq1.split(b"\n"), fq2.split(b"\n")
    assert len(l1) == len(l2) == 4 * st["pairs_written"] + 1
    names1, names2 = l1[0::4][:-1], l2[0::4][:-1]
    assert all(a[:-1] == b[:-1] and a.endswith(b"/1") and b.endswith(b"/2") for a, b in zip(names1, names2))
    idx = np.array([int(n[1:n.index(b"#")]) for n in names1])
    assert (np.diff(idx) >= 0).all(), "records must come in amplicon order"
    lens = np.array([len(s) for s in l1[1::4]])
    assert lens.min() >= 50 and abs(lens.mean() - L) < 1.0 and (lens != L).mean() < 0.3
    assert all(len(s) == len(q) for s, q in zip(l1[1::4], l1[3::4]))
    q = np.frombuffer(b"".join(l1[3::4]), np.uint8)
    assert q.min() >= 33 and q.max() <= 126
    # same seed -> same bytes; different seed -> different bytes
    g2 = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=fa, coverage=30, seed=5)
    assert g2.run() == (fq1, fq2)
    g2.set_seed(6)
    assert g2.run()[0] != fq1


@pytest.mark.parametrize("world,case,model,cov,layout,hooks", [(2, "g1_hiseq2500_pe", "Illumina_HiSeq2500", "3", "PE", "host"),
                                                               (2, "g1_hiseq2500_pe", "Illumina_HiSeq2500", "3", "PE", "device"),
                                                               (3, "g2_xten_pe_nblock", "Illumina_HiSeqXTen", "2", "PE", "device")])
def test_sharded_gpu_job_equals_whole_job(world, case, model, cov, layout, hooks, oracle_bin, models, golden_inputs, tmp_path):
    """ONE job sharded by fragment lineage over `world` processes (sharing this box's single GPU, collectives over gloo):
    every rank writes its FASTQ shard + index through the library's file sink, the native range merge
    (scs_merge_fastq_shards) rebuilds the single-job files, and they must equal the unsharded job's -- the oracle's."""
    import socket
    import sys
    seed = "777"
    whole = str(tmp_path / "whole")
    _oracle_run(oracle_bin, golden_inputs[case], models[model], whole, ["-c", cov, "-l", layout], seed)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), golden_inputs[case], models[model],
                                       str(tmp_path / "shard"), cov, layout, seed, hooks], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for r in range(world):
        assert os.path.getsize(str(tmp_path / "shard") + ".r%d_1.fq" % r) > 0 and os.path.exists(str(tmp_path / "shard") + ".r%d.idx" % r)
    scssim_amd.merge_fastq_shards(str(tmp_path / "shard"), world, paired=True)
    for suffix in ("_1.fq", "_2.fq"):
        assert open(str(tmp_path / "shard") + suffix, "rb").read() == open(whole + suffix, "rb").read(), "sharded GPU job differs from the whole job (%s)" % suffix
    assert not os.path.exists(str(tmp_path / "shard") + ".r0_1.fq"), "shards are removed after the merge"


# ---- the exhaustion regime (Malbac::updatePrimerCount, Malbac.cpp:91-103): a primer type is used exactly `stock`
# times, by the first `stock` attachments in list order that ask for it.  The repeat-rich genome of tests/conftest.py at
# -p 10000 -r 1e-8 drives the 8-mers of its low-complexity runs dry; oracle: sequential live decrement (amplify_pass), HIP: the
# cut table and its fixed point (scs_k_amplify.hip "the primer stock, exactly", exact_stock in scs_amplify.cpp).
def test_primer_exhaustion_bit_exact(oracle_bin, models, repeat_genome, tmp_path):
    seed, stock = 31, 10000
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, repeat_genome, models["Illumina_HiSeq2500"], prefix, ["-c", "0.5", "-p", str(stock), "-r", "1e-8"], seed, dump=prefix)
    prim = np.loadtxt(prefix + ".primers.tsv", dtype=np.int64)                       # primer type, attachments, stock left
    want_stock = np.full(65536, stock, np.int64); want_stock[prim[:, 0]] = prim[:, 2]
    assert (want_stock == 0).sum() >= 4 and (prim[:, 1] + prim[:, 2] == stock).all(), "the case must drive primer types dry, exactly"
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=repeat_genome, coverage=0.5, seed=seed, primers=stock, gamma=1e-8)
    g.create_frags(); g.amplify()
    st = g.stats()
    assert st["stock_exhausted_passes"] >= 1 and st["stock_rounds"] >= 1, st
    got_stock = g.download_primer_stock()
    bad = np.nonzero(got_stock != want_stock)[0]
    assert len(bad) == 0, "primer stock differs for %d types, e.g. %s: got %s want %s" % (len(bad), bad[:6], got_stock[bad[:6]], want_stock[bad[:6]])
    for kind, name in ((0, "semis"), (1, "fulls")):
        want = _load_dump(prefix + "." + name + ".tsv")
        a = g.download_amplicons(kind)
        assert len(a["parent"]) == len(want), name
        w = np.array([r[:5] for r in want], np.int64)
        for k, col in enumerate(("parent", "spos", "len", "gc")):
            d = np.nonzero(a[col].astype(np.int64) != w[:, k])[0]
            assert len(d) == 0, "%s.%s differs at %d rows, first %d" % (name, col, len(d), d[0])
    g.allocate_reads(0)
    fq1, fq2 = g.yield_reads()
    assert fq1 == open(prefix + "_1.fq", "rb").read(), _fastq_diff(fq1, open(prefix + "_1.fq", "rb").read())
    assert fq2 == open(prefix + "_2.fq", "rb").read()
    print("exhaustion: %d types dry, %d passes with a dry type, %d rounds, %d checks" % ((got_stock == 0).sum(), st["stock_exhausted_passes"], st["stock_rounds"], st["stock_checks"]))


@pytest.mark.parametrize("gseed,seed,stock,gamma,n", [(1, 5, 10000, 1e-8, 1500000), (2, 6, 10000, 1e-8, 1800000), (3, 7, 12000, 1e-8, 2400000)])
def test_primer_exhaustion_seed_sweep(gseed, seed, stock, gamma, n, oracle_bin, models, tmp_path):
    """More genomes, seeds and stocks in the regime where primer types run dry (pool x gamma as the defaults': the growth per cycle that
    lets a stock of 10^4 run out within 2 Mb): the primer stock after the job and the FASTQ against the oracle."""
    from conftest import write_repeat_genome
    fa = write_repeat_genome(str(tmp_path / "rep.fa"), n=n, seed=gseed)
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, models["Illumina_HiSeq2000"], prefix, ["-c", "0.5", "-p", str(stock), "-r", repr(gamma)], seed, dump=prefix)
    prim = np.loadtxt(prefix + ".primers.tsv", dtype=np.int64)
    want_stock = np.full(65536, stock, np.int64); want_stock[prim[:, 0]] = prim[:, 2]
    assert (want_stock == 0).sum() >= 2 and (prim[:, 1] + prim[:, 2] == stock).all()
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2000"], input_fasta=fa, coverage=0.5, seed=seed, primers=stock, gamma=gamma)
    fq1, fq2 = g.run()
    assert np.array_equal(g.download_primer_stock(), want_stock)
    assert fq1 == open(prefix + "_1.fq", "rb").read() and fq2 == open(prefix + "_2.fq", "rb").read()
    assert g.stats()["stock_exhausted_passes"] >= 1


def test_primer_exhaustion_of_every_usable_type_bit_exact(oracle_bin, models, tmp_path):
    """A genome of A and T alone: 256 primer types carry every attachment, and at -p 10000 -r 1e-8 all of them run dry in the course of
    an 8 Mb job, dozens in the same pass -- the cuts of one round move each other's (exact_stock iterates), later passes start with
    most types at 0.  Amplicon tables, stock and FASTQ against the oracle."""
    rng = np.random.default_rng(23)
    n = 8000000
    seq = np.frombuffer(b"AT", np.uint8)[rng.integers(0, 2, size=n)]
    fa = str(tmp_path / "at.fa")
    with open(fa, "wb") as f:
        for hap in (1, 2):
            f.write(b">7_%d_%d\n" % (hap, n))
            f.write(np.concatenate([seq.reshape(-1, 100), np.full((n // 100, 1), 10, np.uint8)], axis=1).tobytes())
    seed, stock, gamma = 41, 10000, 1e-8
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, models["Illumina_HiSeqXTen"], prefix, ["-c", "0.2", "-p", str(stock), "-r", repr(gamma)], seed, dump=prefix)
    prim = np.loadtxt(prefix + ".primers.tsv", dtype=np.int64)
    want_stock = np.full(65536, stock, np.int64); want_stock[prim[:, 0]] = prim[:, 2]
    assert (want_stock == 0).sum() >= 200 and (prim[:, 1] + prim[:, 2] == stock).all()
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeqXTen"], input_fasta=fa, coverage=0.2, seed=seed, primers=stock, gamma=gamma)
    g.create_frags(); g.amplify()
    st = g.stats()
    got_stock = g.download_primer_stock()
    bad = np.nonzero(got_stock != want_stock)[0]
    assert len(bad) == 0, "primer stock differs for %d types, e.g. %s: got %s want %s" % (len(bad), bad[:6], got_stock[bad[:6]], want_stock[bad[:6]])
    for kind, name in ((0, "semis"), (1, "fulls")):
        a = g.download_amplicons(kind)
        w = np.loadtxt(prefix + "." + name + ".tsv", dtype=np.int64, usecols=(1, 2, 3, 4), delimiter="\t")
        assert len(a["parent"]) == len(w), name
        for k, col in enumerate(("parent", "spos", "len", "gc")):
            assert np.array_equal(a[col].astype(np.int64), w[:, k]), "%s.%s" % (name, col)
    g.allocate_reads(0)
    fq1, fq2 = g.yield_reads()
    assert fq1 == open(prefix + "_1.fq", "rb").read() and fq2 == open(prefix + "_2.fq", "rb").read()
    print("AT genome: %d types dry, %d passes with a dry type, %d rounds" % ((got_stock == 0).sum(), st["stock_exhausted_passes"], st["stock_rounds"]))
    assert st["stock_exhausted_passes"] >= 1 and st["stock_rounds"] >= 2, "hundreds of cuts in one pass move each other: more than one round"


@pytest.mark.parametrize("world,hooks", [(2, "device"), (3, "host")])
def test_primer_exhaustion_sharded_equals_whole_job(world, hooks, oracle_bin, models, repeat_genome, tmp_path):
    """The same regime as 2 and 3 shards: the shards' demand is summed, the pass in which a type runs dry is run again segment by
    segment in the whole job's list order (attach_pass), and the merged shards equal the unsharded job's -- the oracle's -- files."""
    import socket
    import sys
    seed, stock = "32", "10000"
    whole = str(tmp_path / "whole")
    _oracle_run(oracle_bin, repeat_genome, models["Illumina_HiSeq2500"], whole, ["-c", "0.5", "-p", stock, "-r", "1e-8"], seed, dump=whole)
    prim = np.loadtxt(whole + ".primers.tsv", dtype=np.int64)
    n_dry = int((prim[:, 2] == 0).sum())
    assert n_dry >= 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), repeat_genome, models["Illumina_HiSeq2500"],
                                       str(tmp_path / "shard"), "0.5", "PE", seed, hooks, "0", stock, "1e-8"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for o in outs:
        f = [l for l in o.splitlines() if l.startswith("STOCK ")][0].split()
        assert int(f[3]) >= 1 and int(f[5]) == n_dry, "every shard must see the dry types of the whole job: " + " ".join(f)
    scssim_amd.merge_fastq_shards(str(tmp_path / "shard"), world, paired=True)
    for suffix in ("_1.fq", "_2.fq"):
        assert open(str(tmp_path / "shard") + suffix, "rb").read() == open(whole + suffix, "rb").read(), "sharded GPU job differs from the whole job (%s)" % suffix


def test_gz_input_matches_oracle(oracle_bin, models, golden_inputs, tmp_path):
    """`-i simu.fa.gz` (Genome::loadRefSeq, lib/genome/Genome.cpp:183-187: inflated with gzip -cd beside itself), from a directory
    whose name needs quoting, through the library and through the CLI."""
    import gzip
    d = tmp_path / "in put's"
    d.mkdir()
    gz = str(d / "simu.fa.gz")
    with open(golden_inputs["g1_hiseq2500_pe"], "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"], prefix, ["-c", "2"], 19)
    w1, w2 = open(prefix + "_1.fq", "rb").read(), open(prefix + "_2.fq", "rb").read()
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=gz, coverage=2.0, seed=19)
    assert g.run() == (w1, w2)
    os.remove(gz[:-3]); os.remove(gz[:-3] + ".fai")
    out = str(tmp_path / "cli")
    r = subprocess.run([os.path.join(ROOT, "scssim_amd", "bin", "scssim"), "genreads", "-i", gz, "-m", models["Illumina_HiSeq2500"], "-c", "2", "-o", out, "--seed", "19"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out + "_1.fq", "rb").read() == w1 and open(out + "_2.fq", "rb").read() == w2


def test_sharded_medium_job_equals_oracle(oracle_bin, models, tmp_path):
    """The sharded path at a size where its bookkeeping has something to do: 12 Mb in two records, 3 shards (device hooks) -- 4.5 M
    amplicons = 4 500 allocation chunks, most list segments' ends inside a chunk (the boundary rows), tens of thousands of pairs per
    shard and segment; every shard writes two part files per mate; the merged files against the oracle's."""
    import socket
    import sys
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "7000000,5000000", "--seed", "33", "--n-block", "20000", "--simu-out", fa])
    whole = str(tmp_path / "whole")
    _oracle_run(oracle_bin, fa, models["Illumina_HiSeq2500"], whole, ["-c", "4"], 88, threads=min(32, os.cpu_count() or 1))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), fa, models["Illumina_HiSeq2500"],
                                       str(tmp_path / "shard"), "4", "PE", "88", "device", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    scssim_amd.merge_fastq_shards(str(tmp_path / "shard"), 3, paired=True)
    for suffix in ("_1.fq", "_2.fq"):
        assert _md5_file(str(tmp_path / "shard") + suffix) == _md5_file(whole + suffix), "sharded GPU job differs from the whole job (%s)" % suffix


def test_device_hooks_over_rccl_single_rank(models, golden_inputs, oracle_bin, tmp_path):
    """The RCCL code path of the device hooks (torch tensors aliasing the library's HBM buffers, collectives on the
    shared stream) with a 1-rank NCCL group: a 1-shard "sharded" job must equal the plain job."""
    import sys
    code = '''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import scssim_amd
from scssim_amd.dist import Collectives
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
stream = torch.cuda.Stream()
coll = Collectives(stream=stream)
t = torch.arange(8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
assert coll._allreduce_dev(None, t.data_ptr(), 8, 8) == 0
torch.cuda.synchronize()
assert t.tolist() == list(range(8))
g = scssim_amd.GenReads(profile=%r, input_fasta=%r, coverage=2.0, seed=5, stream=stream.cuda_stream, shard_rank=0, shard_count=1)
g.set_collectives(coll, device_hooks=True)
a = g.run()
dist.destroy_process_group()
open(%r, "wb").write(a[0])
''' % (ROOT, models["Illumina_HiSeq2500"], golden_inputs["g1_hiseq2500_pe"], str(tmp_path / "nccl_1.fq"))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29777", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"], prefix, ["-c", "2"], 5)
    assert open(str(tmp_path / "nccl_1.fq"), "rb").read() == open(prefix + "_1.fq", "rb").read()


def test_rccl_inside_the_library_single_rank_and_cli(models, golden_inputs, oracle_bin, tmp_path):
    """RCCL bound inside the library (scs_comm_unique_id / scs_comm_init, no Python hooks): a 1-rank communicator runs
    every exchange of the sharded path through ncclAllReduce / ncclAllGather on the ctx stream; the job must equal the
    plain one.  And the CLI's file sink (`scssim genreads`, reference file names) writes the same bytes."""
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"], prefix, ["-c", "2"], 5)
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=golden_inputs["g1_hiseq2500_pe"], coverage=2.0, seed=5)
    assert g.comm_count() == 0
    g.comm_init(scssim_amd.comm_unique_id(), 0, 1)
    assert g.comm_count() == 1                                       # what RCCL itself reports (ncclCommCount)
    g.create_frags(); g.amplify(); g.allocate_reads(0)
    g.yield_reads_files(str(tmp_path / "lib"))
    for suffix in ("_1.fq", "_2.fq"):
        assert open(str(tmp_path / "lib") + suffix, "rb").read() == open(prefix + suffix, "rb").read()
    # the same inside a process that already carries RCCL (torch's own copy: what bench.py --gpus N runs in): the library
    # binds to the loaded copy and works on a torch stream
    import sys
    code = '''
import sys, torch
sys.path.insert(0, %r)
import scssim_amd
torch.cuda.set_device(0)
st = torch.cuda.Stream()
g = scssim_amd.GenReads(profile=%r, input_fasta=%r, coverage=2.0, seed=5, stream=st.cuda_stream)
g.comm_init(scssim_amd.comm_unique_id(), 0, 1)
g.create_frags(); g.amplify(); g.allocate_reads(0)
g.yield_reads_files(%r)
''' % (ROOT, models["Illumina_HiSeq2500"], golden_inputs["g1_hiseq2500_pe"], str(tmp_path / "tor"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    for suffix in ("_1.fq", "_2.fq"):
        assert open(str(tmp_path / "tor") + suffix, "rb").read() == open(prefix + suffix, "rb").read()
    exe = os.path.join(ROOT, "scssim_amd", "bin", "scssim")
    r = subprocess.run([exe, "genreads", "-i", golden_inputs["g1_hiseq2500_pe"], "-m", models["Illumina_HiSeq2500"], "-c", "2", "-o", str(tmp_path / "cli"),
                        "--seed", "5", "--gpus", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for suffix in ("_1.fq", "_2.fq"):
        assert open(str(tmp_path / "cli") + suffix, "rb").read() == open(prefix + suffix, "rb").read()


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_two_rank_control_flow(ranks, tmp_path):
    """bench.py's N > 1 path (ONE job sharded N ways by fragment lineage -- strong scaling --, a FASTQ shard per rank,
    max-over-ranks timing) rehearsed as 2 and as 4 ranks on this box's one GPU on a scaled-down genome (4 + the launcher + this
    process: the most the pool lets share a card): gloo collectives staged through the CPU instead of RCCL.  Checks the control flow and the
    JSON contract -- the per-rank stage times and sink rates the scaling curve is read with --, not a rate."""
    import json
    import sys
    env = dict(os.environ, SCS_BENCH_BACKEND="gloo", SCS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # 2 ranks: the plain command, no launcher on the command line -- bench.py starts its own ranks before its first GPU call
    # (launch_ranks); 4 ranks: the driver's own command
    launcher = [] if ranks == 2 else ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                                      "--master-port", str(29791 + ranks)]
    r = subprocess.run([sys.executable] + launcher + [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--genome-mb", "4"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 2 and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert [p["rank"] for p in d["per_rank"]] == list(range(ranks)) and all(p["sink_GBps"] > 0 and p["amplify_s"] > 0 for p in d["per_rank"])
    assert sum(p["pairs"] for p in d["per_rank"]) == d["config"]["pairs_per_step"] and len(d["generation_hbm"]["per_rank"]) == ranks
    assert d["value"] > 0 and abs(d["value"] - d["config"]["pairs_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.05
    assert 350000 < d["config"]["pairs_per_step"] < 450000                         # the whole 4 Mb job at 30x, PE150: ~400 k pairs
    assert 0 < d["roofline"]["frac"] <= 1 and "cpu_baseline" not in d
    assert "part files per mate and rank" in d["config"]["output"] and d["config"]["sink_GBps"] > 0      # every rank wrote its shard's parts


def test_bench_single_gpu_contract(tmp_path):
    """bench.py at N = 1 on a scaled-down genome: every leg runs (timed steps through the file sink, generation only, D2H only, the
    small configurations, CLI wall, CPU baseline at -t 1 and -t cores) and the JSON line carries the contract's fields."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--genome-mb", "6", "--cpu-sample-mb", "0.2"],
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None and "workload" in d["config"] and d["scaling"] == "strong"
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["draws_per_s"] > 0
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["cpu_baseline"].get("one_thread", {}).get("value", 0) > 0 or d["cpu_baseline"]["kind"] == "port"
    assert "part files per mate" in d["config"]["output"] and "/" in d["config"]["output"] and d["config"]["sink_GBps"] > 0
    assert d["generation_hbm"]["value"] > d["value"] and d["d2h_only"]["value"] > 0
    assert rf["kernel"] == "k_reads" and rf["timed_launches"] > 0 and rf["generation_hbm_leg"]["timed_launches"] > 0 and d["roofline_amplification"]["frac"] > 0
    assert "timed region" in rf["timing"] and d["cpu_baseline"].get("unpinned", {}).get("value", 1) > 0
    assert isinstance(d["sweep"], list) and len(d["sweep"]) == 2 and all(x["generation_hbm_pairs_per_s"] > 0 for x in d["sweep"])
    assert d["cli_wall"].get("value", 0) > 0, d["cli_wall"]
    assert abs(d["value"] - d["config"]["pairs_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.05


def test_medium_genome_bit_exact(oracle_bin, models, tmp_path):
    """12 Mb, two records, 4x: thousands of allocation chunks, hundreds of fragments, the >4-errors overflow pool,
    multi-digit record names -- FASTQ still byte-identical to the oracle."""
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "7000000,5000000", "--seed", "31", "--n-block", "20000", "--simu-out", fa])
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, models["Illumina_HiSeqXTen"], prefix, ["-c", "4"], 8, threads=min(32, os.cpu_count() or 1))
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeqXTen"], input_fasta=fa, coverage=4.0, seed=8)
    fq1, fq2 = g.run()
    st = g.stats()
    assert st["full_amplicons"] > 4000000
    a = g.download_amplicons(1)
    assert (a["nerr"] > 4).sum() > 100, "overflow pool not exercised"
    assert fq1 == open(prefix + "_1.fq", "rb").read()
    assert fq2 == open(prefix + "_2.fq", "rb").read()


def _cnv_haplotype(base, rng, copies):
    """A copy-number-edited haplotype in the shape `simuvars` writes it (Genome::generateSegment, Genome.cpp:386-691: the
    segment of a `c` line is emitted CN times in tandem, or dropped for CN 0): 5 kb segments at random places, one per
    requested copy number."""
    n = len(base)
    cuts = sorted(int(x) for x in rng.choice(np.arange(1, n // 5000 - 1), size=len(copies), replace=False) * 5000)
    out, pos = [], 0
    for start, cn in zip(cuts, copies):
        out.append(base[pos:start]); out.append(base[start:start + 5000] * cn); pos = start + 5000
    out.append(base[pos:])
    return "".join(out)


def test_pe250_and_unequal_haplotypes(oracle_bin, models, tmp_path):
    """BASELINE config 5 shape in miniature: a synthesised 250 bp profile (4 lane chunks per read, -s 500) on a
    CNV-heavy input: the two haplotype records carry tandem copy numbers 0..8 (copy number "up to 8") and so
    differ in length from each other and from the reference length in their names (simuvars keeps the reference length
    there, Genome.cpp:368; the read budget uses that name field, Malbac.cpp:413-420)."""
    rng = np.random.default_rng(9)
    ref = "".join(rng.choice(list("ACGT"), size=300000, p=[0.3, 0.2, 0.2, 0.3]))
    hap1 = _cnv_haplotype(ref, rng, [0, 3, 8, 1, 5])
    hap2 = _cnv_haplotype(ref, rng, [2, 0, 4, 7, 6, 8])
    assert len(hap1) != len(hap2) and len(hap1) > len(ref) < len(hap2)
    fa = str(tmp_path / "cnv.fa")
    with open(fa, "w") as f:
        for name, s in (("7_1_300000", hap1), ("7_2_300000", hap2)):
            f.write(">%s\n" % name)
            f.write("\n".join(s[i:i + 100] for i in range(0, len(s), 100)) + "\n")
    prof = str(tmp_path / "pe250.profile")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeqXTen"], prof, "--read-length", "250"])
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "6", "-s", "500"], 21)
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=6.0, isize=500, seed=21)
    assert g.read_length == 250
    fq1, fq2 = g.run()
    assert g.stats()["reads_requested"] == 300000 * 6 // 250
    assert fq1 == open(prefix + "_1.fq", "rb").read()
    assert fq2 == open(prefix + "_2.fq", "rb").read()


def test_bench_profile_pe150_bit_exact(oracle_bin, models, tmp_path):
    """The bench's own model (HiSeq2500 resampled to 150 bins: 8-bit qualities, the big-row variant of k_reads) against
    the oracle on a 1 Mb genome at low coverage, plus a low-coverage 30x-shaped slice of the bench options (-s 260)."""
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "1000000", "--seed", "1", "--simu-out", fa])
    prof = str(tmp_path / "pe150.profile")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeq2500"], prof, "--read-length", "150"])
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "3", "-s", "260"], 1003, threads=min(32, os.cpu_count() or 1))
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=3.0, isize=260, seed=1003)
    assert g.read_length == 150
    fq1, fq2 = g.run()
    assert g.stats()["pairs_written"] == 10000
    assert fq1 == open(prefix + "_1.fq", "rb").read()
    assert fq2 == open(prefix + "_2.fq", "rb").read()


def test_config1_full_size_bit_exact(oracle_bin, models, tmp_path):
    """BASELINE configs[1] AT ITS STATED SIZE, bit for bit: 1 Mb synthetic reference (x 2 haplotypes), PE150 (the HiSeq2500 model
    resampled to 150 bins), 30x, -s 260 -- 100 000 pairs; the FASTQ of the HIP path == the oracle's, through the API and through the
    CLI's two files."""
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "1000000", "--seed", "1", "--simu-out", fa])
    prof = str(tmp_path / "pe150.profile")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeq2500"], prof, "--read-length", "150"])
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "30", "-s", "260"], 11, threads=min(32, os.cpu_count() or 1))
    want = [open(prefix + s, "rb").read() for s in ("_1.fq", "_2.fq")]
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=30.0, isize=260, seed=11)
    fq1, fq2 = g.run()
    st = g.stats()
    assert st["reads_requested"] == 200000 and st["pairs_written"] == 100000 == want[0].count(b"\n") // 4
    assert fq1 == want[0] and fq2 == want[1]
    g.close()
    exe = os.path.join(ROOT, "scssim_amd", "bin", "scssim")
    out = str(tmp_path / "cli")
    r = subprocess.run([exe, "genreads", "-i", fa, "-m", prof, "-c", "30", "-s", "260", "-o", out, "--seed", "11"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out + "_1.fq", "rb").read() == want[0] and open(out + "_2.fq", "rb").read() == want[1]


def test_read_classes_take_the_share_of_the_reads_they_are_built_for(models, tmp_path):
    """The straight-line walks only pay while the reads they are built for reach them: with the PE150 model 86 % of the reads have no indel
    event, 7 % one deleted base, 2 % one inserted base (the one-event walk takes both) and what is left goes through the general variant.
    The seams build prints the class lists' sizes per batch (SCS_DEBUG_CLASSES): general below 5.5 % and one-event above 8 % of the reads;
    with SCS_TEST_NO_I1 the inserted-base reads are back in the general class (above 6 %)."""
    import re
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "2000000", "--seed", "3", "--simu-out", fa])
    prof = str(tmp_path / "pe150.profile")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeq2500"], prof, "--read-length", "150"])
    exe = os.path.join(ROOT, "scssim_amd", "bin", "scssim_seams")                    # the seams build of the CLI (scs_seams.h)

    def shares(**knobs):
        r = subprocess.run([exe, "genreads", "-i", fa, "-m", prof, "-c", "30", "-s", "260", "-o", str(tmp_path / "o"), "--seed", "5"],
                           capture_output=True, text=True, env=dict(os.environ, SCS_DEBUG_CLASSES="1", **knobs))
        assert r.returncode == 0, r.stderr
        m = re.findall(r"\[classes\] batch of (\d+) pairs: general (\d+) / (\d+), one-event (\d+) / (\d+)", r.stderr)
        assert m, r.stderr[-2000:]
        pairs = sum(int(x[0]) for x in m)
        return sum(int(x[1]) + int(x[2]) for x in m) / (2.0 * pairs), sum(int(x[3]) + int(x[4]) for x in m) / (2.0 * pairs)

    gen, one = shares()
    assert 0.03 < gen < 0.055 and 0.08 < one < 0.11, (gen, one)
    gen0, one0 = shares(SCS_TEST_NO_I1="1")
    assert 0.06 < gen0 < 0.08 and 0.06 < one0 < 0.08, (gen0, one0)


def test_paired_end_job_on_a_model_without_insert_size_spread_fails_like_the_reference(oracle_bin, models, golden_inputs, tmp_path):
    """[Insert Size Standard Deviation] 0 + PE: the reference has no insert-size alphabet (Profile.cpp:908), its first yieldInsertSize
    asks Config for a parameter that does not exist and exit(1)s with `Error: unrecognized parameter name "insertSize"`
    (Profile.cpp:1482-1485, Config.cpp:85-93) after the amplification, its two files opened and empty.  The oracle fails the same way;
    the library returns SCS_EIO with that message (the ctx stays usable), the CLI prints it and leaves with status 1 -- and the same model
    runs a single-end job, equal to the oracle's."""
    import re
    fa = golden_inputs["g1_hiseq2500_pe"]
    prof = str(tmp_path / "sigma0.profile")
    text = open(models["Illumina_HiSeq2500"]).read()
    text2 = re.sub(r"(\[Insert Size Standard Deviation\]\s*\n)\S+", r"\g<1>0", text)
    assert text2 != text
    open(prof, "w").write(text2)
    msg = 'Error: unrecognized parameter name "insertSize"'
    r = subprocess.run([oracle_bin, "genreads", "-i", fa, "-m", prof, "-c", "2", "-o", str(tmp_path / "orc_pe"), "--rng", "counter", "--seed", "4", "-t", "4", "-q"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and msg in r.stderr
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=2.0, seed=4)
    g.create_frags(); g.amplify(); g.allocate_reads(0)
    with pytest.raises(scssim_amd.ScsError) as e:
        g.yield_reads_files(str(tmp_path / "lib_pe"))
    assert msg in str(e.value) and e.value.code == scssim_amd.SCS_EIO
    assert os.path.getsize(str(tmp_path / "lib_pe") + "_1.fq") == 0 and os.path.getsize(str(tmp_path / "lib_pe") + "_2.fq") == 0   # SeqWriter opened them (Malbac.cpp:426-435)
    with pytest.raises(scssim_amd.ScsError) as e:
        g.yield_reads_sink(None)
    assert msg in str(e.value)
    g.close()
    exe = os.path.join(ROOT, "scssim_amd", "bin", "scssim")
    r = subprocess.run([exe, "genreads", "-i", fa, "-m", prof, "-c", "2", "-o", str(tmp_path / "cli_pe"), "--seed", "4"], capture_output=True, text=True)
    assert r.returncode == 1 and msg in r.stderr and "MALBAC amplification..." in r.stderr, r.stderr
    # single end: no insert size is drawn, the model is fine
    _oracle_run(oracle_bin, fa, prof, str(tmp_path / "orc_se"), ["-c", "2", "-l", "SE"], 4)
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=2.0, layout="SE", seed=4)
    fq, _ = g.run()
    assert fq == open(str(tmp_path / "orc_se") + ".fq", "rb").read() and len(fq) > 100000
    g.close()


@pytest.mark.parametrize("rl,layout", [(17, "SE"), (36, "PE"), (51, "PE"), (75, "SE"), (257, "PE"), (130, "PE"), (145, "SE"), (148, "PE"), (149, "SE"), (160, "PE")])
def test_odd_read_lengths_bit_exact(rl, layout, oracle_bin, models, tmp_path):
    """Read lengths around the walk's block sizes (16-position blocks, 16-base window dwords, the 50-base floor under which indels
    are dropped): a model resampled to the length, 1 Mb genome, byte for byte against the oracle.  130 .. 160: where the one-event walk's
    last step falls in its block -- a one-deletion read idles through it beside the one-insertion reads (its character cleared: byte 1, 3,
    0 of the word, the block's last position), and at L = 16 k + 1 (145, 257) the insertion reads stay with the general variant."""
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "1000000", "--seed", "9", "--simu-out", fa])
    prof = str(tmp_path / "m.profile")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeq2500"], prof, "--read-length", str(rl)])
    prefix = str(tmp_path / "orc")
    isz = max(260, 2 * rl)
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "1", "-l", layout, "-s", str(isz)], 300 + rl, threads=min(32, os.cpu_count() or 1))
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=1.0, layout=layout, isize=isz, seed=300 + rl)
    assert g.read_length == rl
    fq1, fq2 = g.run()
    if layout == "PE":
        w1, w2 = open(prefix + "_1.fq", "rb").read(), open(prefix + "_2.fq", "rb").read()
        assert fq1 == w1, _fastq_diff(fq1, w1)
        assert fq2 == w2, _fastq_diff(fq2, w2)
    else:
        w1 = open(prefix + ".fq", "rb").read()
        assert fq1 == w1, _fastq_diff(fq1, w1)
    assert len(w1) > 50000


def test_very_long_reads_take_the_general_variant(oracle_bin, models, tmp_path):
    """The uniform walk gathers a read's window with at most 64 lanes (16 bases each): reads longer than 1008 bases all go through the
    general variant.  A model resampled to 1040 bins, single-end, on a 1 Mb genome, byte for byte against the oracle."""
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "1000000", "--seed", "7", "--simu-out", fa])
    prof = str(tmp_path / "se1040.profile")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeq2500"], prof, "--read-length", "1040"])
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "4", "-l", "SE"], 77, threads=min(32, os.cpu_count() or 1))
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=4.0, layout="SE", seed=77)
    assert g.read_length == 1040
    fq1, _ = g.run()
    want = open(prefix + ".fq", "rb").read()
    assert len(want) > 100000
    assert fq1 == want, _fastq_diff(fq1, want)
    # most amplicons are shorter than these reads: their planned reads are holes, counted once each (also when a batch boundary
    # falls inside an amplicon: the small-batch run repeats this test)
    assert g.stats()["pairs_written"] == want.count(b"\n") // 4


@pytest.mark.parametrize("noise", [0.02, 0.12])
def test_noisy_profile_many_substitutions_bit_exact(noise, oracle_bin, models, tmp_path):
    """The uniform walk keeps the window's base with one compare and sets a position whose draw does not keep it aside (three
    entries in front of the read's window in LDS, later ones over the window's consumed start, at most six; resolved after the
    pass); a read that runs out of room is made again by redo_read.  The shipped models substitute a few bases per thousand:
    a read has 0-2 entries.  Here every substitution row is mixed with the uniform row -- 1.5 % (2-3 entries per read, the
    overlaid slots in use) and 9 % (a dozen per read: nearly every read overflows and is made again) -- PE125, byte for byte."""
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "1200000", "--seed", "11", "--simu-out", fa])
    prof = str(tmp_path / "noisy.profile")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeq2500"], prof, "--read-length", "125", "--noise", "%g" % noise])
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "3"], 91, threads=min(32, os.cpu_count() or 1))
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=3.0, seed=91)
    fq1, fq2 = g.run()
    w1, w2 = open(prefix + "_1.fq", "rb").read(), open(prefix + "_2.fq", "rb").read()
    assert len(w1) > 1000000
    assert fq1 == w1, _fastq_diff(fq1, w1)
    assert fq2 == w2, _fastq_diff(fq2, w2)


@pytest.mark.parametrize("case", ["g1", "medium"])
def test_insert_size_give_up_path_bit_exact(case, oracle_bin, models, golden_inputs, tmp_path):
    """-s 1500: the mean insert is as long as the amplicons (1000-2000 bases), so most insert-size draws are rejected
    (`isize > ampLen`, Amplicon.cpp:484-489) and many amplicons give up after 1000 fails in a row: their planned pairs become
    holes in the record numbering.  FASTQ byte for byte against the oracle (which reproduces the reference in this regime), the
    holes counted once each -- also with a batch boundary inside an amplicon (the small-batch run repeats this test)."""
    if case == "g1":
        fa, prof, cov, seed = golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"], 6.0, 15
    else:
        fa = str(tmp_path / "simu.fa")
        subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "7000000,5000000", "--seed", "31", "--n-block", "20000", "--simu-out", fa])
        prof, cov, seed = models["Illumina_HiSeqXTen"], 3.0, 16
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "%g" % cov, "-s", "1500"], seed, threads=min(32, os.cpu_count() or 1))
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=cov, isize=1500, seed=seed)
    fq1, fq2 = g.run()
    w1, w2 = open(prefix + "_1.fq", "rb").read(), open(prefix + "_2.fq", "rb").read()
    assert fq1 == w1, _fastq_diff(fq1, w1)
    assert fq2 == w2
    st = g.stats()
    made = w1.count(b"\n") // 4
    assert st["pairs_written"] == made and st["reads_written"] == 2 * made
    assert 0 < made < 0.9 * (st["reads_requested"] // 2), "the give-up path was not reached"
    # the per-batch checksums of the text in HBM (scs_set_batch_checksums) against the oracle's checksum mode -- what pins
    # BASELINE configs[3] at full size (tests/golden/whole_genome_config3.json): a batch = the records of a range of PLANNED pairs,
    # holes included, so this regime (and the small-batch run's 4096-pair batches) is where the two could disagree
    shift = int(os.environ.get("SCS_TEST_BATCH_SHIFT", "23"))
    subprocess.check_call([oracle_bin, "genreads", "-i", fa, "-m", prof, "--rng", "counter", "--seed", str(seed), "-t", "8", "-q", "-c", "%g" % cov, "-s", "1500",
                           "--checksums", prefix + ".cks", "--batch-pairs", str(1 << shift)])
    want = [(int(f[1], 16), int(f[2], 16)) for f in (l.split() for l in open(prefix + ".cks") if not l.startswith("#"))]
    g.set_batch_checksums(True)
    g.yield_reads_sink(None)
    got = g.batch_checksums()
    assert len(want) == (st["reads_requested"] // 2 + (1 << shift) - 1) >> shift and (shift == 23 or case == "g1" or len(want) > 20)
    assert got == want, "batch checksums differ from the oracle's: first at batch %d of %d" % (next(i for i, (a, b) in enumerate(zip(got, want)) if a != b), len(want))
    assert sum(int(l.split()[5]) for l in open(prefix + ".cks") if not l.startswith("#")) == made


def _md5_file(path):
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


class _HashSink:
    """scs_sink_fn that hashes the two FASTQ streams batch by batch (a chr20-size job at 30x writes 4 GB)."""

    def __init__(self):
        import hashlib
        self.h = [hashlib.md5(), hashlib.md5()]
        self.bytes = [0, 0]

    def __call__(self, _u, p1, n1, p2, n2):
        for k, (p, n) in enumerate(((p1, n1), (p2, n2))):
            if n:
                self.h[k].update(memoryview((ctypes.c_char * n).from_address(p)))
                self.bytes[k] += n
        return 0


@pytest.mark.parametrize("cfg", ["config0_pe100_1x", "config2_novaseq_pe150_30x"])
def test_chr20_size_bit_exact(cfg, oracle_bin, models, tmp_path):
    """BASELINE configs[0] and configs[2] at full size: a 63 025 520-base chr20 stand-in (the reference's testData ref.fa.gz
    is absent from its mount; synthetic i.i.d. bases with a leading N block), two haplotype records.
      config 0: HiSeq2500 model resampled to PE100, 1x  (the reference's own plumbing case)
      config 2: "NovaSeq" = the HiSeq X Ten model (binned qualities, as NovaSeq reports them) resampled to 150 bins, 30x
    Both are compared with the oracle bit for bit: 25 M amplicons, every allocation chunk, 6.3 M pairs / 4 GB of FASTQ at
    30x (hashed stream against hashed files)."""
    fa = str(tmp_path / "chr20.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "63025520", "--seed", "20", "--n-block", "60000", "--simu-out", fa])
    prof = str(tmp_path / "m.profile")
    if cfg == "config0_pe100_1x":
        subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeq2500"], prof, "--read-length", "100"])
        cov, seed = 1.0, 100
    else:
        subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeqXTen"], prof, "--read-length", "150"])
        cov, seed = 30.0, 220
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "%g" % cov], seed, threads=min(64, os.cpu_count() or 1))
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=cov, seed=seed)
    g.create_frags(); g.amplify(); g.allocate_reads(0)
    if cfg == "config0_pe100_1x":                                                   # through a caller's sink ...
        sink = _HashSink()
        g.yield_reads_sink(sink)
        got_bytes, got_md5 = sink.bytes, [h.hexdigest() for h in sink.h]
    else:                                                                           # ... and through the library's file sink (a dozen batches, unaligned offsets)
        out = str(tmp_path / "gpu")
        g.yield_reads_files(out)
        got_bytes = [os.path.getsize(out + "_1.fq"), os.path.getsize(out + "_2.fq")]
        got_md5 = [_md5_file(out + "_1.fq"), _md5_file(out + "_2.fq")]
    st = g.stats()
    assert 24000000 < st["full_amplicons"] < 27000000                               # 25 359 057 measured on the reference
    assert st["reads_requested"] == int(63025520 * cov / g.read_length) and abs(2 * st["pairs_written"] - st["reads_requested"]) <= 2
    assert got_bytes == [os.path.getsize(prefix + "_1.fq"), os.path.getsize(prefix + "_2.fq")]
    assert got_md5 == [_md5_file(prefix + "_1.fq"), _md5_file(prefix + "_2.fq")]


def _write_fa(path, recs):
    with open(path, "w") as f:
        for name, s in recs:
            f.write(">%s\n" % name)
            for i in range(0, len(s), 100):
                f.write(s[i:i + 100] + "\n")


@pytest.mark.parametrize("name", ["short_record", "too_short_to_amplify", "all_N", "zero_reads", "empty_record_mixed"])
def test_degenerate_inputs_match_oracle(name, oracle_bin, models, tmp_path):
    """Ragged / empty inputs: records shorter than the fragment or amplicon minimum, all-N sequence, zero requested reads,
    an empty record next to a normal one.  No faults, and the (possibly empty) FASTQ equals the oracle's."""
    rng = np.random.default_rng(3)
    rnd = lambda n: "".join(rng.choice(list("ACGT"), size=n))
    recs, cov = {
        "short_record": ([("1_1_6000", rnd(6000)), ("1_2_6000", rnd(6000))], "40"),
        "too_short_to_amplify": ([("1_1_900", rnd(900)), ("1_2_900", rnd(900))], "30"),
        "all_N": ([("1_1_30000", "N" * 30000), ("1_2_30000", "N" * 30000)], "5"),
        "zero_reads": ([("1_1_20000", rnd(20000)), ("1_2_20000", rnd(20000))], "0.001"),
        "empty_record_mixed": ([("1_1_0", ""), ("2_1_15000", rnd(15000)), ("2_2_15000", rnd(15000))], "20"),
    }[name]
    fa = str(tmp_path / "in.fa")
    _write_fa(fa, recs)
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, models["Illumina_HiSeq2500"], prefix, ["-c", cov], 77, threads=2)
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=fa, coverage=float(cov), seed=77)
    fq1, fq2 = g.run()
    assert fq1 == open(prefix + "_1.fq", "rb").read()
    assert fq2 == open(prefix + "_2.fq", "rb").read()
    st = g.stats()
    assert st["pairs_written"] == fq1.count(b"\n") // 4
    if name in ("too_short_to_amplify", "all_N"):
        assert st["full_amplicons"] == 0 and fq1 == b""


def test_error_reporting_through_the_abi(models, tmp_path):
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"])
    with pytest.raises(scssim_amd.ScsError) as e:
        g.create_frags()
    assert e.value.code == 1 and "no genome" in str(e.value)
    with pytest.raises(scssim_amd.ScsError):
        g.load_genome(str(tmp_path / "nope.fa"))
    with pytest.raises(scssim_amd.ScsError) as e:
        scssim_amd.GenReads(primers=10)
    assert "at least 1000" in str(e.value)
    with pytest.raises(scssim_amd.ScsError) as e:
        scssim_amd.GenReads(gamma=1.0)
    assert "0~1e-8" in str(e.value)


def test_replayed_indel_reads_match_oracle(models, tmp_path):
    """A read whose indel events do not fit the 16-bit LDS slots is 'replayed' (phase 2 re-draws stream A).  The fallback is
    rare in production, so the same parity checks run once more in a child process with SCS_EV_REPLAY=1, which sends every
    read with an indel through it."""
    env = seams_env(SCS_EV_REPLAY="1")
    sel = ["tests/test_gpu_parity.py::test_predict_batch_matches_oracle", "tests/test_gpu_parity.py::test_full_pipeline_fastq_bit_exact"]
    if os.environ.get("SCS_EV_REPLAY"):
        pytest.skip("already inside the replay run")
    r = subprocess.run(["python", "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.parametrize("qk", ["64", "128"])
def test_wider_alias_rows_match_oracle(qk, models, tmp_path):
    """k_reads is instantiated for quality alias rows of 16, 64 and 128 columns; a model gets the smallest that holds its most
    varied row, and no shipped model needs 128.  SCS_TEST_QK makes the table builder (and the oracle's) use at least that many
    columns, so the wider instantiations run the same parity checks (child processes: the tables are built once per process)."""
    if os.environ.get("SCS_TEST_QK"):
        pytest.skip("already inside the wide-row run")
    env = seams_env(SCS_TEST_QK=qk)
    sel = ["tests/test_gpu_parity.py::test_full_pipeline_fastq_bit_exact", "tests/test_gpu_parity.py::test_predict_batch_matches_oracle"]
    r = subprocess.run(["python", "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


def test_one_call_entry_and_timer_switches(models, golden_inputs):
    """scs_run_genreads (the whole job in one call of the C ABI) writes what the four stage calls write; scs_set_kernel_timing switches
    the HIP-event timers of the hot kernels off and on again."""
    case, model, oargs, layout, cov, isize, extra = CASES[0]
    kw = dict(profile=models[model], input_fasta=golden_inputs[case], coverage=cov, isize=isize, layout=layout, seed=99, **extra)
    a = scssim_amd.GenReads(**kw)
    f1, f2 = a.run()
    b = scssim_amd.GenReads(**kw)
    b.set_kernel_timing(names=[], every=1)                              # no timers
    g1, g2 = b.run_genreads()
    assert (g1, g2) == (f1, f2) and len(f1) > 1000
    assert all(v["launches"] == 0 for v in b.kernel_times().values())
    b.set_kernel_timing(names=None, every=1)                            # all of them
    b.set_seed(99)
    h1, h2 = b.run_genreads()
    assert (h1, h2) == (f1, f2)
    kt = b.kernel_times()
    assert kt["k_reads"]["launches"] >= 1 and kt["k_reads"]["ms"] > 0 and kt["k_attach<semi>"]["launches"] >= 1


def test_device_resident_output_matches_oracle(oracle_bin, models, golden_inputs, tmp_path):
    """scs_yield_reads_device leaves the FASTQ text in caller-owned device memory (the boundary a GPU-side consumer binds): the same
    bytes as the files of the oracle; too small a buffer is reported, not overrun.  (The small-batch run repeats this test: many
    batches into one contiguous buffer.)"""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]

    def dmalloc(n):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), n) == 0
        return p

    def dtoh(p, n):
        buf = ctypes.create_string_buffer(n)
        assert hip.hipMemcpy(buf, p, n, 2) == 0                     # hipMemcpyDeviceToHost (synchronises)
        return buf.raw
    case, model, oargs, layout, cov, isize, extra = CASES[0]
    seed = 4242
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, golden_inputs[case], models[model], prefix, oargs, seed)
    w1, w2 = open(prefix + "_1.fq", "rb").read(), open(prefix + "_2.fq", "rb").read()
    g = scssim_amd.GenReads(profile=models[model], input_fasta=golden_inputs[case], coverage=cov, isize=isize, layout=layout, seed=seed, **extra)
    g.create_frags(); g.amplify(); g.allocate_reads(0)
    c1, c2 = len(w1) + 4096, len(w2) + 4096
    b1, b2 = dmalloc(c1), dmalloc(c2)
    try:
        n1, n2, pairs = g.yield_reads_device(b1, c1, b2, c2)
        assert (n1, n2) == (len(w1), len(w2)) and pairs == w1.count(b"\n") // 4
        assert dtoh(b1, n1) == w1 and dtoh(b2, n2) == w2
        with pytest.raises(scssim_amd.ScsError):
            g.yield_reads_device(b1, len(w1) // 2, b2, c2)
    finally:
        hip.hipFree(b1); hip.hipFree(b2)


def test_many_small_batches_match_oracle(models, tmp_path):
    """The read stage works batch by batch (8 M pairs with the text staying in HBM, 512 k towards a sink), the next batch's
    pre-pass queued ahead of the current base pass into a second set of buffers.  The parity cases fit one batch, so they run once
    more in a child process with 4096-pair batches (SCS_TEST_BATCH_SHIFT=12): dozens of batches, ragged last one."""
    if os.environ.get("SCS_TEST_BATCH_SHIFT"):
        pytest.skip("already inside the small-batch run")
    env = seams_env(SCS_TEST_BATCH_SHIFT="12")
    sel = ["tests/test_gpu_parity.py::test_full_pipeline_fastq_bit_exact", "tests/test_gpu_parity.py::test_medium_genome_bit_exact",
           "tests/test_gpu_parity.py::test_very_long_reads_take_the_general_variant", "tests/test_gpu_parity.py::test_device_resident_output_matches_oracle",
           "tests/test_gpu_parity.py::test_insert_size_give_up_path_bit_exact"]
    r = subprocess.run(["python", "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.parametrize("knob", ["SCS_TEST_REDO", "SCS_TEST_GENERAL", "SCS_TEST_NO_D1", "SCS_TEST_NO_I1", "SCS_READS_SERIAL", "SCS_READS_SPLIT", "SCS_ERRS_INLINE"])
def test_read_class_fallbacks_match_oracle(knob, models, tmp_path):
    """The base pass has a straight-line variant for event-free, ACGT-only reads and one for reads whose only event is the
    deletion or the insertion of one base.  A read of them that runs out of room for a substituted base's quality (or draws 0xFFFFFFFF) is made
    again by a scalar fallback (redo_read): SCS_TEST_REDO=1 sends every such read with a substitution through it.
    SCS_TEST_GENERAL=1 sends every read through the general variant instead, SCS_TEST_NO_D1=1 the one-event reads,
    SCS_TEST_NO_I1=1 only those with an inserted base;
    SCS_READS_SERIAL=1 launches the class kernels one after the other and SCS_READS_SPLIT=1 on three streams instead of as one merged
    launch; SCS_ERRS_INLINE=1 keeps the
    amplification's k_errs<semi->full> on the main stream (it has its own by default).  Same parity checks, in child
    processes (the knobs are read once per process)."""
    if os.environ.get(knob):
        pytest.skip("already inside the %s run" % knob)
    env = seams_env(**{knob: "1"})
    sel = ["tests/test_gpu_parity.py::test_full_pipeline_fastq_bit_exact", "tests/test_gpu_parity.py::test_medium_genome_bit_exact"]
    r = subprocess.run(["python", "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


def test_mapped_buffer_grows_in_place_by_what_is_asked():
    """A device buffer above 64 MB lives in a reserved address range and grows by mapping more memory behind it: it must keep
    its address and take the request + 3 % (rounded to the 128 MB mapping unit), not half its size again -- a whole-genome job
    whose slot count came out 0.01 % above the previous job's used to map ~25 GB in the middle of a timed step.  A small
    buffer is a plain block: it moves, and grows by half so that it is not copied at every step."""
    GB, MB = 1 << 30, 1 << 20
    c0, c1, same = scssim_amd.devbuf_probe(2 * GB, 2 * GB + MB)
    assert 2 * GB <= c0 <= 2 * GB + 128 * MB
    assert same and 2 * GB + MB <= c1 <= c0 + (2 * GB) // 25 + 128 * MB, (c0, c1, same)
    a0, a1, _ = scssim_amd.devbuf_probe(MB, MB + 1)
    assert a0 >= MB and a1 >= MB + MB // 2, (a0, a1)


def test_mapped_buffers_parity(models, tmp_path):
    """Device buffers above 64 MB live in a reserved address range and grow by mapping more memory behind them (HIP virtual
    memory management).  The parity cases are far smaller, so they run once more in a child process with the threshold at
    0 MB: every growable buffer is a mapped one."""
    if os.environ.get("SCS_VMM_FROM_MB"):
        pytest.skip("already inside the mapped-buffer run")
    env = seams_env(SCS_VMM_FROM_MB="0")
    sel = ["tests/test_gpu_parity.py::test_full_pipeline_fastq_bit_exact", "tests/test_gpu_parity.py::test_medium_genome_bit_exact",
           "tests/test_gpu_parity.py::test_degenerate_inputs_match_oracle"]
    r = subprocess.run(["python", "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sel, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


# ---- simuvars on the data plane -------------------------------------------------------------------------
def _sv_inputs(tmp_path):
    import gzip
    sv = os.path.join(ROOT, "tests", "golden", "simuvars")
    ref = str(tmp_path / "ref.fa")
    open(ref, "wb").write(gzip.open(os.path.join(sv, "ref.fa.gz")).read())
    return sv, ref


def test_simuvars_cli_matches_reference_output(tmp_path):
    """`scssim simuvars` (haplotypes assembled on the GPU from the host's piece plan) writes the file the compiled
    reference wrote for the same inputs (tests/golden/simuvars: copy numbers 0..8, SNPs on both strands, SNVs, insertions,
    deletions, het / homo), byte for byte; and without variation files the plain diploid copy."""
    import gzip
    import hashlib
    import json
    sv, ref = _sv_inputs(tmp_path)
    man = json.load(open(os.path.join(sv, "manifest.json")))
    exe = os.path.join(ROOT, "scssim_amd", "bin", "scssim")
    out = str(tmp_path / "full.fa")
    r = subprocess.run([exe, "simuvars", "-r", ref, "-s", os.path.join(sv, "snp.txt"), "-v", os.path.join(sv, "vars.txt"), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "CNV: 13" in r.stderr and "340 SNPs to simulate were loaded" in r.stderr
    assert open(out, "rb").read() == gzip.open(os.path.join(sv, "expected_full.fa.gz")).read()
    for case, extra in (("plain", []), ("snp_only", ["-s", os.path.join(sv, "snp.txt")])):
        o2 = str(tmp_path / (case + ".fa"))
        r = subprocess.run([exe, "simuvars", "-r", ref, "-o", o2] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert hashlib.sha256(open(o2, "rb").read()).hexdigest() == man[case]["sha256"]
    bad = subprocess.run([exe, "simuvars", "-o", out], capture_output=True, text=True)
    assert bad.returncode == 1 and "Use --ref to specify the reference file" in bad.stderr


def test_simuvars_then_genreads_without_a_fasta(oracle_bin, models, tmp_path):
    """BASELINE config 5's input path: the CNV-edited diploid genome goes from scs_simuvars straight into genreads, resident
    in HBM -- and the reads equal the oracle's reads from the simuvars FASTA file (PE250, -s 500, copy numbers up to 8)."""
    sv, ref = _sv_inputs(tmp_path)
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call([oracle_bin, "simuvars", "-r", ref, "-s", os.path.join(sv, "snp.txt"), "-v", os.path.join(sv, "vars.txt"), "-o", fa])
    prof = str(tmp_path / "pe250.profile")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_profile.py"), models["Illumina_HiSeqXTen"], prof, "--read-length", "250"])
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, prof, prefix, ["-c", "4", "-s", "500"], 61)
    g = scssim_amd.GenReads(profile=prof, coverage=4.0, isize=500, seed=61)
    g.simuvars(ref, os.path.join(sv, "snp.txt"), os.path.join(sv, "vars.txt"))
    fq1, fq2 = g.run()
    assert g.stats()["records"] == 6 and g.stats()["reads_requested"] == 610000 * 4 // 250
    assert fq1 == open(prefix + "_1.fq", "rb").read()
    assert fq2 == open(prefix + "_2.fq", "rb").read()


def test_simuvars_random_variations_match_oracle(oracle_bin, tmp_path):
    """A CNV-heavy random variation file (copy numbers 0..8, every major-copy split, hundreds of SNVs / insertions /
    deletions, thousands of SNPs) on a 3 Mb two-record reference: GPU-built haplotypes == the oracle's std::string edits."""
    rng = np.random.default_rng(77)
    ref = str(tmp_path / "r.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "2000000,1000000", "--seed", "9", "--first-chr", "7", "--lower-frac", "0.05", "--ref-out", ref])
    var, snp = str(tmp_path / "v.txt"), str(tmp_path / "s.txt")
    with open(var, "w") as f:
        for chrom, n in (("chr7", 2000000), ("chr8", 1000000)):
            pos = 1000
            while pos + 60000 < n:                                                  # sorted, non-overlapping CNV intervals
                ln = int(rng.integers(5000, 40000)); cn = int(rng.integers(0, 9)); mcn = int(rng.integers((cn + 1) // 2, cn + 1))
                f.write("c\t%s\t%d\t%d\t%d\t%d\n" % (chrom, pos, pos + ln, cn, mcn))
                pos += ln + int(rng.integers(1, 30000))
            for p in sorted(rng.choice(np.arange(500, n - 500), size=300, replace=False)):
                kind = rng.integers(3)
                ht = "het" if rng.random() < 0.5 else "homo"
                if kind == 0:
                    f.write("s\t%s\t%d\tN\t%s\t%s\n" % (chrom, p, "ACGT"[rng.integers(4)], ht))
                elif kind == 1:
                    f.write("i\t%s\t%d\t%s\t%s\n" % (chrom, p, "".join(rng.choice(list("acgt"), size=int(rng.integers(1, 30)))), ht))
                else:
                    f.write("d\t%s\t%d\t%d\t%s\n" % (chrom, p, int(rng.integers(1, 40)), ht))
    with open(snp, "w") as f:
        for chrom, n in (("chr7", 2000000), ("chr8", 1000000)):
            for i, p in enumerate(sorted(rng.choice(np.arange(1, n), size=4000, replace=False))):
                a, b = rng.choice(list("ACGT"), size=2, replace=False)
                f.write("rs%d\t%s\t%d\t%s/%s\t%s\t%s\n" % (i, chrom, p, a, b, "+-"[rng.integers(2)], a))
    want = str(tmp_path / "orc.fa")
    subprocess.check_call([oracle_bin, "simuvars", "-r", ref, "-s", snp, "-v", var, "-o", want])
    got = str(tmp_path / "gpu.fa")
    g = scssim_amd.GenReads()
    g.simuvars(ref, snp, var, got)
    assert open(got, "rb").read() == open(want, "rb").read()
    assert g.stats()["records"] == 4


def test_record_bound_guard_reports_instead_of_faulting(models, golden_inputs):
    """k_reads checks every FASTQ record against the size of the batch's text before it writes (round 1 saw two GPU memory
    faults from stores that trusted a length computed elsewhere: DESIGN.md section 9).  With the batch size mis-stated on
    purpose (SCS_TEST_SHRINK_OUT halves it) the job must end with SCS_EOVERFLOW 'internal', not with a fault."""
    import sys
    code = '''
import os, sys
sys.path.insert(0, %r)
import scssim_amd
g = scssim_amd.GenReads(profile=%r, input_fasta=%r, coverage=2.0, seed=5)
try:
    g.run()
except scssim_amd.ScsError as e:
    assert e.code == 4 and "internal" in str(e), str(e)
    print("GUARDED")
''' % (ROOT, models["Illumina_HiSeq2500"], golden_inputs["g1_hiseq2500_pe"])
    r = subprocess.run([sys.executable, "-c", code], env=seams_env(SCS_TEST_SHRINK_OUT="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GUARDED" in r.stdout, r.stdout + r.stderr


def test_fasta_parsed_on_the_device_handles_ragged_text(oracle_bin, models, tmp_path):
    """The device-side FASTA parser (raw file bytes -> bases, k_fa_*): CRLF line ends, ';' comment lines, blank lines, lower
    case, ragged line widths, one record on a single 40 kb line, a header with a description, no newline at the end of the
    file.  Same records as the host parser sees (names, lengths, checksum via the .fai and the job), same reads as the oracle."""
    rng = np.random.default_rng(12)
    rnd = lambda n: "".join(rng.choice(list("ACGTacgtN"), size=n, p=[0.24, 0.2, 0.2, 0.24, 0.03, 0.03, 0.02, 0.02, 0.02]))
    a, b, c = rnd(30011), rnd(40000), rnd(25000)
    fa = str(tmp_path / "ragged.fa")
    with open(fa, "wb") as f:
        f.write(b"\r\n>chr3_1_30011 first haplotype\r\n")
        for i in range(0, len(a), 70):
            f.write(a[i:i + 70].encode() + b"\r\n")
            if i == 700:
                f.write(b";a comment line in the middle of a record\r\n\r\n")
        f.write(b">3_2_40000\n" + b.encode() + b"\n")                             # one line
        f.write(b";comment before a header\n>chrom4_1_25000\n")
        w = 0
        for i, step in enumerate([13, 100, 1, 57] * 1000):
            if w >= len(c):
                break
            f.write(c[w:w + step].encode() + b"\n"); w += step
        f.write(b">4_2_9")                                                          # an empty record, and no final newline
    names, total, _ = scssim_amd.fasta_probe(fa)                                    # the host parser's view
    assert names == ["3_1_30011", "3_2_40000", "4_1_25000", "4_2_9"] and total == 95011
    prefix = str(tmp_path / "orc")
    _oracle_run(oracle_bin, fa, models["Illumina_HiSeq2500"], prefix, ["-c", "6"], 44, threads=2)
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=fa, coverage=6.0, seed=44)
    st = g.stats()
    assert st["records"] == 4 and st["genome_bases"] == 95011
    fq1, fq2 = g.run()
    assert fq1 == open(prefix + "_1.fq", "rb").read() and fq2 == open(prefix + "_2.fq", "rb").read()
    fai = [l.split("\t") for l in open(fa + ".fai").read().splitlines()]
    assert [(l[0], int(l[1])) for l in fai] == [("chr3_1_30011", 30011), ("3_2_40000", 40000), ("chrom4_1_25000", 25000), ("4_2_9", 0)]
    bad = str(tmp_path / "bad.fa")
    open(bad, "w").write("ACGT\n>x_1_4\nACGT\n")
    with pytest.raises(scssim_amd.ScsError) as e:
        scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=bad)
    assert "sequence before header" in str(e.value)
