"""Full-size GPU tests of BASELINE.json's configs[3] and configs[4] on one MI355X, and of the regimes the small parity cases do
not reach: byte offsets above 2^31 inside a batch, the insert-size give-up path.  Where the oracle cannot finish in seconds the
checks are size-independent properties (the read budget rule, amplicon density, checksums equal across runs of one seed,
well-formed text in amplicon order, a checksum computed on the device equal to the host's over the bytes a sink received)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import scssim_amd

pytestmark = pytest.mark.gpu


def _check_fastq_batch(b1, b2, L, paired=True):
    """One batch of the two mates' text: 4-line records, names @<amp>#<cnt>/1|/2 equal between the mates, amplicon indices ascending,
    bases ACGTN, qualities 33..126, sequence and quality of one length in 50..L+64.  Returns (records, first index, last index)."""
    l1 = b1.split(b"\n")
    assert l1[-1] == b"" and (len(l1) - 1) % 4 == 0
    n = (len(l1) - 1) // 4
    names1 = l1[0:4 * n:4]
    idx = np.array([int(x[1:x.index(b"#")]) for x in names1], dtype=np.int64)
    assert (np.diff(idx) >= 0).all(), "records must come in amplicon order"
    assert all(x == b"+" for x in l1[2:4 * n:4])
    if paired:
        l2 = b2.split(b"\n")
        assert len(l2) == len(l1)
        assert all(a.endswith(b"/1") and b.endswith(b"/2") and a[:-1] == b[:-1] for a, b in zip(names1, l2[0:4 * n:4]))
    for lines in ((l1, l2) if paired else (l1,)):
        ls = np.array([len(x) for x in lines[1:4 * n:4]])
        lq = np.array([len(x) for x in lines[3:4 * n:4]])
        assert (ls == lq).all() and ls.min() >= 50 and ls.max() <= L + 64 and abs(ls.mean() - L) < 1.0
        seq = np.frombuffer(b"".join(lines[1:4 * n:4]), np.uint8)
        assert np.isin(seq, np.frombuffer(b"ACGTN", np.uint8)).all()
        q = np.frombuffer(b"".join(lines[3:4 * n:4]), np.uint8)
        assert q.min() >= 33 and q.max() <= 126
    return n, int(idx[0]), int(idx[-1])


class _SpotSink:
    """scs_sink_fn: counts every batch's bytes, checksums every `check_every`-th batch's bytes on the host (numpy), keeps a copy of
    the first batch and of batch number `last_index`."""

    def __init__(self, check_every, last_index):
        self.bytes = [0, 0]; self.batches = 0; self.cks = []; self.first = None; self.last = None
        self.check_every = check_every; self.last_index = last_index

    def __call__(self, _u, p1, n1, p2, n2):
        v1 = (ctypes.c_char * n1).from_address(p1) if n1 else b""
        v2 = (ctypes.c_char * n2).from_address(p2) if n2 else b""
        self.bytes[0] += n1; self.bytes[1] += n2
        if self.batches % self.check_every == 0 or self.batches == self.last_index:
            self.cks.append((self.batches, scssim_amd.text_checksum(v1), scssim_amd.text_checksum(v2)))
        if self.batches == 0:
            self.first = (bytes(v1), bytes(v2))
        if self.batches == self.last_index:
            self.last = (bytes(v1), bytes(v2))
        self.batches += 1
        return 0


def _whole_genome(torch, scale=1.0):
    sys.path.insert(0, ROOT)
    import bench
    lens = bench.record_lengths(0.0 if scale == 1.0 else sum(bench.HG19) * scale / 1e6)
    return bench, lens


@pytest.mark.parametrize("cfg", ["config3_pe150_30x", "config4_pe250_60x_cnv"])
def test_whole_genome_properties(cfg, models, tmp_path):
    """BASELINE configs[3] (3.1 Gb diploid genome, PE150 30x, -s 260) and configs[4] (the same reference through scs_simuvars with a
    CNV-heavy variation file -- copy numbers 0..8 --, PE250 60x, -s 500), whole jobs on one GPU."""
    import torch
    bench, lens = _whole_genome(torch)
    G = sum(lens)
    dev = torch.device("cuda", 0)
    td = str(tmp_path)
    stream = torch.cuda.Stream()
    if cfg.startswith("config3"):
        prof = bench.make_profile(td)
        cov, isz, L = 30.0, 260, 150
        names, rl, bases = bench.synth_genome(torch, dev, lens, 3000)
        genome_fp = bench.genome_fingerprint(torch, bases)
        torch.cuda.synchronize()
        g = scssim_amd.GenReads(profile=prof, coverage=cov, isize=isz, seed=11, stream=stream.cuda_stream)
        g.upload_genome_device(names, rl, bases.data_ptr())
        del bases
        torch.cuda.empty_cache()
        total_bases = 2 * G
    else:
        src = os.path.join(td, "xten.profile")
        open(src, "wb").write(__import__("gzip").open(os.path.join(ROOT, "tests", "golden", "models", "Illumina_HiSeqXTen.profile.gz")).read())
        prof = os.path.join(td, "pe250.profile")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profile.py"), src, prof, "--read-length", "250"])
        cov, isz, L = 60.0, 500, 250
        # the reference as a FASTA file on tmpfs (60 columns, records chr1..chr24), a variation file with copy numbers 0..8
        shm = "/dev/shm" if os.path.isdir("/dev/shm") else td
        ref = os.path.join(shm, "scs_cfg4_ref_%d.fa" % os.getpid())
        var = os.path.join(td, "vars.txt")
        try:
            expect, genome_fp = bench.make_config4_inputs(torch, dev, lens, ref, var)
            g = scssim_amd.GenReads(profile=prof, coverage=cov, isize=isz, seed=11, stream=stream.cuda_stream)
            g.simuvars(ref, None, var)
        finally:
            for p in (ref, ref + ".fai"):
                if os.path.exists(p):
                    os.remove(p)
        torch.cuda.empty_cache()
        total_bases = g.stats()["genome_bases"]
        assert g.stats()["records"] == 48
        assert abs(total_bases - expect) <= 48 * 16, (total_bases, expect)          # the haplotypes carry the copies the file asked for (interval ends inclusive)
        assert total_bases > 2 * G * 1.15                                          # mean copy number 4 over a fifth of the genome
    assert g.read_length == L
    g.set_batch_checksums(True)
    golden = None
    if True:
        # the oracle's run of this very job (tools/whole_genome_golden.py, 7 / 13 minutes of 16 host cores): counts and the checksum of every
        # 8 M-pair batch's text per mate.  The genome comes from torch's device generator: its fingerprint says whether this box made
        # the genome the golden belongs to (a different generator is not a parity failure -- but it must be said, not skipped over)
        import json
        gpath = os.path.join(ROOT, "tests", "golden", "whole_genome_%s.json" % cfg.split("_")[0])
        golden = json.load(open(gpath))
        assert golden["genome"]["fingerprint"] == "%016x" % genome_fp, "this box's synthetic genome is not the one %s was made from: make it again (tools/whole_genome_golden.py)" % gpath

    def job(seed, sink=None):
        g.set_seed(seed)
        g.create_frags(); g.amplify(); g.allocate_reads(0)
        g.yield_reads_sink(sink)
        return g.stats(), g.batch_checksums()

    st, cks = job(11)                                                                 # a NULL sink: the text stays in HBM batch buffers (8 M pairs each)
    want_reads = int(G * cov / L)                                                     # Malbac.cpp:413-420: the names' reference lengths / 2, not the haplotypes' bases
    assert st["reads_requested"] == want_reads
    assert 0 <= want_reads // 2 + 1 - st["pairs_written"] <= 2 + want_reads // 200000, "pairs = planned - holes; holes are rare at this insert size"
    per_mb = st["full_amplicons"] / (total_bases / 2e6)
    assert 3.5e5 < per_mb < 4.7e5 and 3.5e4 < st["semi_amplicons"] / (total_bases / 2e6) < 5.5e4, per_mb      # ~4.1e5 / ~4.5e4 per haploid Mb
    assert len(cks) == (st["pairs_written"] + (1 << 23) - 1) >> 23 or len(cks) == ((want_reads + 1) // 2 + (1 << 23) - 1) >> 23
    assert all(a and b for a, b in cks)
    if golden:
        # BASELINE configs[3] / configs[4] bit for bit against the oracle at full size: 1.1e9 and more amplicons = 1.1e6 allocation chunks
        # (the third level of the chunk CDF, MyDefine.cpp:203-253), amplicon indices above 1e9 in the names (Amplicon.cpp:460,498,519),
        # 197 GB of text and more; configs[4] with its input made by simuvars on both sides
        c = golden["counts"]
        assert (st["fragments"], st["semi_amplicons"], st["full_amplicons"]) == (c["frags"], c["semis"], c["fulls"]), (st, c)
        assert st["full_amplicons"] > 1000000000 and len(cks) >= 37 and len(golden["batches"]) >= 5
        if len(golden["batches"]) == len(cks):                                        # (configs[4]'s golden holds a sample of the 45 batches)
            assert st["pairs_written"] == sum(r[5] for r in golden["batches"]) and st["fastq_bytes"] == [sum(r[3] for r in golden["batches"]), sum(r[4] for r in golden["batches"])]
        assert c["planned"] == (want_reads + 1) // 2 or c["planned"] == want_reads // 2
        print("%s: %d batches of %d compared with the oracle's; primer stock: %d checks, %d passes with a dry type, %d rounds" % (cfg, len(golden["batches"]), len(cks), st["stock_checks"], st["stock_exhausted_passes"], st["stock_rounds"]))
        bad = [r[0] for r in golden["batches"] if cks[r[0]] != (int(r[1], 16), int(r[2], 16))]
        assert not bad, "the text of batches %s differs from the oracle's (of %d)" % (bad[:8], len(cks))
    mean_rec = sum(st["fastq_bytes"]) / (2.0 * st["pairs_written"])
    assert 2 * L + 14 < mean_rec < 2 * L + 30                                         # "@<amp>#<cnt>/1\n" + bases + "\n+\n" + qualities + "\n"
    bytes_null = list(st["fastq_bytes"])
    st2, cks2 = job(11)
    assert cks2 == cks and st2["fastq_bytes"] == bytes_null and st2["full_amplicons"] == st["full_amplicons"], "one seed, one text"
    st3, cks3 = job(12)
    assert all(x != y for x, y in zip(cks3, cks)) and st3["full_amplicons"] != st["full_amplicons"]
    # the same job once more through a sink: 512 k-pair batches over PCIe.  Every byte is counted, a sample of the batches is
    # checksummed on the host and must equal the device's checksum of that batch; the first and the last batch are parsed.
    n_sink_batches = (st["pairs_written"] + (1 << 19) - 1) >> 19
    spot = _SpotSink(check_every=64, last_index=n_sink_batches - 1)
    st4, cks4 = job(11, spot)
    assert spot.bytes == bytes_null and st4["pairs_written"] == st["pairs_written"], "sum of the record sizes = bytes the sink received"
    assert spot.batches == len(cks4) and spot.batches > 500
    for b, c1, c2 in spot.cks:
        assert (c1, c2) == cks4[b], "batch %d: device checksum != checksum of the bytes that crossed PCIe" % b
    n_first, i0, _ = _check_fastq_batch(spot.first[0], spot.first[1], L)
    assert n_first == 1 << 19 and i0 < 64
    assert spot.last is not None
    n_last, _, i1 = _check_fastq_batch(spot.last[0], spot.last[1], L)
    assert n_first * (spot.batches - 1) + n_last == st["pairs_written"]
    assert st["full_amplicons"] - 64 < i1 < st["full_amplicons"], "the last record belongs to one of the last amplicons"


def test_offsets_above_2_31_inside_a_batch_bit_exact(oracle_bin, tmp_path):
    """One 180 Mb record x 2 haplotypes at 30x PE150: 18 M pairs left in HBM (scs_yield_reads_device) = three batches, the first two
    8 M pairs and 2.7 GB of text per mate each -- record offsets above 2^31 inside a batch -- compared with the oracle's files
    chunk by chunk."""
    import torch
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    work = os.path.join(shm, "scs_big_%d" % os.getpid())
    os.makedirs(work)
    try:
        fa = os.path.join(work, "simu.fa")
        subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "180000000", "--seed", "249", "--simu-out", fa])
        src = os.path.join(work, "h.profile")
        open(src, "wb").write(__import__("gzip").open(os.path.join(ROOT, "tests", "golden", "models", "Illumina_HiSeq2500.profile.gz")).read())
        prof = os.path.join(work, "pe150.profile")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profile.py"), src, prof, "--read-length", "150"])
        prefix = os.path.join(work, "orc")
        subprocess.check_call([oracle_bin, "genreads", "-i", fa, "-m", prof, "-o", prefix, "--rng", "counter", "--seed", "77", "-t", str(min(64, os.cpu_count() or 1)), "-q", "-c", "30"])
        sizes = [os.path.getsize(prefix + s) for s in ("_1.fq", "_2.fq")]
        assert min(sizes) > 5 * (1 << 30)
        g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=30.0, seed=77)
        g.create_frags(); g.amplify(); g.allocate_reads(0)
        d1 = torch.empty(sizes[0] + 4096, dtype=torch.uint8, device="cuda")
        d2 = torch.empty(sizes[1] + 4096, dtype=torch.uint8, device="cuda")
        n1, n2, pairs = g.yield_reads_device(d1.data_ptr(), d1.numel(), d2.data_ptr(), d2.numel())
        assert (n1, n2) == tuple(sizes) and abs(2 * pairs - 36000000) <= 2
        step = 1 << 28
        for d, n, suffix in ((d1, n1, "_1.fq"), (d2, n2, "_2.fq")):
            with open(prefix + suffix, "rb") as f:
                for o in range(0, n, step):
                    want = np.frombuffer(f.read(step), np.uint8)
                    got = d[o:o + len(want)].cpu().numpy()
                    assert np.array_equal(got, want), "%s differs in bytes [%d, %d)" % (suffix, o, o + len(want))
    finally:
        import shutil
        shutil.rmtree(work, ignore_errors=True)
