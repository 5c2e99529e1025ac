"""GPU tests of the FASTQ sink (SeqWriter's replacement): part files written by several writer threads, the
shards of a sharded job written in parts, and what a job leaves behind when its sink fails.  Run with `-m gpu`."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, SEAMS_CLI, seams_env

import scssim_amd

pytestmark = pytest.mark.gpu


def _oracle(oracle_bin, fasta, profile, prefix, args, seed, threads=8):
    subprocess.check_call([oracle_bin, "genreads", "-i", fasta, "-m", profile, "-o", prefix, "--rng", "counter", "--seed", str(seed), "-t", str(threads), "-q"] + args)


def _cat(files):
    return b"".join(open(f, "rb").read() for f in files)


_CHILD = '''
import os, sys
sys.path.insert(0, %(root)r)
import scssim_amd
g = scssim_amd.GenReads(profile=%(prof)r, input_fasta=%(fa)r, coverage=%(cov)r, layout=%(layout)r, seed=%(seed)d)
g.create_frags(); g.amplify(); g.allocate_reads(0)
for K in %(ks)r:
    g.yield_reads_files(%(out)r + "_k%%d" %% K, K)
print("pairs", g.stats()["pairs_written"])
'''


@pytest.mark.parametrize("case,layout,shift,ks", [("medium", "PE", "12", (2, 3, 7)), ("g1", "PE", "9", (4, 64)), ("g3", "SE", "9", (3,))])
def test_part_files_concatenate_to_the_oracle_files(case, layout, shift, ks, oracle_bin, models, golden_inputs, tmp_path):
    """scs_yield_reads_files with K writers: the job's records are cut into K contiguous ranges whose batches are made round-robin
    and written by K threads into K part files per mate.  `cat` of the parts in order must be the oracle's file, for K below,
    at and far above the number of batches (empty parts), PE and SE.  Child processes: small batches (SCS_TEST_BATCH_SHIFT) so
    that every region has many."""
    if case == "medium":
        fa = str(tmp_path / "simu.fa")
        subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "7000000,5000000", "--seed", "31", "--n-block", "20000", "--simu-out", fa])
        prof, cov, seed = models["Illumina_HiSeqXTen"], 4.0, 8
    elif case == "g1":
        fa, prof, cov, seed = golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"], 3.0, 21
    else:
        fa, prof, cov, seed = golden_inputs["g3_hiseq2000_se"], models["Illumina_HiSeq2000"], 2.0, 22
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, fa, prof, prefix, ["-c", "%g" % cov, "-l", layout], seed, threads=min(32, os.cpu_count() or 1))
    out = str(tmp_path / "gpu")
    code = _CHILD % dict(root=ROOT, prof=prof, fa=fa, cov=cov, layout=layout, seed=seed, ks=tuple(ks), out=out)
    r = subprocess.run([sys.executable, "-c", code], env=seams_env(SCS_TEST_BATCH_SHIFT=shift), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    paired = layout == "PE"
    want = [open(prefix + s, "rb").read() for s in (("_1.fq", "_2.fq") if paired else (".fq",))]
    assert len(want[0]) > 100000
    for K in ks:
        base = out + "_k%d" % K
        files = scssim_amd.part_paths(base, K, paired)
        assert all(os.path.exists(f) for m in files for f in m)
        for m in range(len(want)):
            assert _cat(files[m]) == want[m], "K = %d, mate %d" % (K, m + 1)
        sizes = [l.split("\t") for l in open(base + ".parts").read().splitlines() if not l.startswith("#")]
        assert [int(x[1]) for x in sizes] == [os.path.getsize(f) for f in files[0]]
        if case == "medium":
            assert min(os.path.getsize(f) for f in files[0]) > 0, "every writer had work"
    # ... and the host-side merge rebuilds the reference's file names from them
    base = out + "_k%d" % ks[0]
    scssim_amd.merge_fastq_parts(base, paired=paired)
    for m, f in enumerate(scssim_amd.part_paths(base, 1, paired)):
        assert open(f[0], "rb").read() == want[m]


def test_in_place_sink_leaves_the_same_files_as_truncating(oracle_bin, models, golden_inputs, tmp_path):
    """SCS_SINK_IN_PLACE: output files that exist are overwritten where they lie and cut to their new length when they are finished.
    A big job, then a smaller one, then the big one again into the SAME prefix (part files of 3 writers x 2 generations, BGZF and the
    reference's two files): after every job the files are exactly what a job into fresh files leaves -- no tail of the older, longer
    file, the parts index of the new job; the CLI's --in-place the same."""
    import gzip
    import threading
    fa, prof = golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"]
    want = {}
    for cov, seed in ((3.0, 21), (1.0, 22)):
        prefix = str(tmp_path / ("orc%d" % seed))
        _oracle(oracle_bin, fa, prof, prefix, ["-c", "%g" % cov], seed)
        want[seed] = [open(prefix + s, "rb").read() for s in ("_1.fq", "_2.fq")]
    assert len(want[21][0]) > 2 * len(want[22][0]) > 100000
    out, single, gz = str(tmp_path / "parts"), str(tmp_path / "two"), str(tmp_path / "gz")
    for cov, seed in ((3.0, 21), (1.0, 22), (3.0, 21)):
        g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=cov, seed=seed)
        g.create_frags(); g.amplify(); g.allocate_reads(0)
        # a consumer that follows the protocol beside the job: whenever it sees part p + writers it reads part p and requires the FINAL
        # bytes (the files of the job before, still lying under the later generation's names, must not give that signal: they are set aside)
        files_now = scssim_amd.part_paths(out, 6, True)
        stop, seen_bad = threading.Event(), []

        def follow():
            while not stop.is_set():
                for p_ in range(3):
                    if os.path.exists(files_now[0][p_ + 3]):
                        got = open(files_now[0][p_], "rb").read()
                        if got not in want[seed][0]:
                            seen_bad.append((seed, p_, len(got)))
        th = threading.Thread(target=follow); th.start()
        try:
            g.yield_reads_files(out, 3, 2, in_place=True)
        finally:
            stop.set(); th.join()
        assert not seen_bad, seen_bad
        assert not [f for f in os.listdir(str(tmp_path)) if f.endswith(".prev")]
        g.yield_reads_files(single, 1, in_place=True)
        g.yield_reads_files(gz, 2, 1, bgzf=True, in_place=True)
        files = scssim_amd.part_paths(out, 6, True)
        for m in range(2):
            assert _cat(files[m]) == want[seed][m], (seed, m)
            assert open(single + ("_1.fq", "_2.fq")[m], "rb").read() == want[seed][m], (seed, m)
            assert b"".join(gzip.decompress(open(f, "rb").read()) for f in scssim_amd.part_paths(gz, 2, True, ".fq.gz")[m]) == want[seed][m], (seed, m)
        sizes = [l.split("\t") for l in open(out + ".parts").read().splitlines() if not l.startswith("#")]
        assert [int(x[1]) for x in sizes] == [os.path.getsize(f) for f in files[0]]
        g.close()
    # the command line: a short job over the long job's files
    rc, err, left = _cli(["-i", fa, "-m", prof, "-c", "1", "-o", single, "--seed", "22", "--in-place"])
    assert rc == 0 and not left, err
    for m in range(2):
        assert open(single + ("_1.fq", "_2.fq")[m], "rb").read() == want[22][m]


def test_two_jobs_on_two_host_threads_at_once(oracle_bin, models, golden_inputs, tmp_path):
    """One ctx per host thread, no state shared between them ("no global state, thread-compatible"): two different jobs --
    another model, layout, seed, sink -- run at the same time from two threads of one process, three times over, and each leaves
    the oracle's files."""
    import threading
    jobs = [dict(fa=golden_inputs["g1_hiseq2500_pe"], prof=models["Illumina_HiSeq2500"], cov=3.0, layout="PE", seed=61, writers=3),
            dict(fa=golden_inputs["g3_hiseq2000_se"], prof=models["Illumina_HiSeq2000"], cov=2.0, layout="SE", seed=62, writers=1)]
    for k, j in enumerate(jobs):
        j["want_prefix"] = str(tmp_path / ("orc%d" % k))
        _oracle(oracle_bin, j["fa"], j["prof"], j["want_prefix"], ["-c", "%g" % j["cov"], "-l", j["layout"]], j["seed"])
    errors = []

    def run(k, j):
        try:
            for rep in range(3):
                g = scssim_amd.GenReads(profile=j["prof"], input_fasta=j["fa"], coverage=j["cov"], layout=j["layout"], seed=j["seed"])
                g.create_frags(); g.amplify(); g.allocate_reads(0)
                out = str(tmp_path / ("t%d_%d" % (k, rep)))
                g.yield_reads_files(out, j["writers"])
                g.close()
                paired = j["layout"] == "PE"
                files = scssim_amd.part_paths(out, j["writers"], paired)
                for m, suf in enumerate(("_1.fq", "_2.fq") if paired else (".fq",)):
                    if _cat(files[m]) != open(j["want_prefix"] + suf, "rb").read():
                        errors.append("job %d, repetition %d, file %s differs from the oracle's" % (k, rep, suf))
        except Exception as e:                                           # (an exception in a thread would otherwise pass silently)
            errors.append("job %d: %r" % (k, e))
    th = [threading.Thread(target=run, args=(k, j)) for k, j in enumerate(jobs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_sharded_job_written_in_parts_merges_to_the_whole_job(oracle_bin, models, golden_inputs, tmp_path):
    """Two shards (two processes on this box's GPU, gloo hooks), each writing its shard as 3 part files per mate with small
    batches: the native range merge reads the parts as one logical file per shard and rebuilds the unsharded job's files."""
    import socket
    case, model, cov, seed, world = "g2_xten_pe_nblock", "Illumina_HiSeqXTen", "3", "778", 2
    whole = str(tmp_path / "whole")
    _oracle(oracle_bin, golden_inputs[case], models[model], whole, ["-c", cov], seed)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    procs = []
    for r in range(world):
        env = seams_env(RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LOCAL_RANK="0", SCS_TEST_BATCH_SHIFT="9")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), golden_inputs[case], models[model],
                                       str(tmp_path / "shard"), cov, "PE", seed, "device", "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert os.path.exists(str(tmp_path / "shard") + ".r1.p02_2.fq") and os.path.exists(str(tmp_path / "shard") + ".r0.parts")
    scssim_amd.merge_fastq_shards(str(tmp_path / "shard"), world, paired=True)
    for suffix in ("_1.fq", "_2.fq"):
        assert open(str(tmp_path / "shard") + suffix, "rb").read() == open(whole + suffix, "rb").read(), suffix
    assert not os.path.exists(str(tmp_path / "shard") + ".r0.p00_1.fq")


def test_failing_sink_aborts_the_job_and_the_ctx_stays_usable(models, golden_inputs, tmp_path):
    """A sink that reports an error (a full disk, a closed pipe) ends the call with SCS_EIO -- no hang with batches in flight --
    and an output file that cannot be opened is reported before anything runs; the ctx then produces the right reads again."""
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=golden_inputs["g1_hiseq2500_pe"], coverage=2.0, seed=5)
    good = g.run()
    g.create_frags(); g.amplify(); g.allocate_reads(0)
    calls = [0]

    def bad_sink(_u, _p1, n1, _p2, n2):
        calls[0] += 1
        return 1
    with pytest.raises(scssim_amd.ScsError) as e:
        g.yield_reads_sink(bad_sink)
    assert e.value.code == 2 and calls[0] >= 1
    with pytest.raises(scssim_amd.ScsError) as e:
        g.yield_reads_files(str(tmp_path / "no_such_dir" / "x"), 3)
    assert e.value.code == 2 and "can not open fastq file" in str(e.value)
    assert g.run() == good


def _cli(args, env=None, timeout=300):
    import signal
    exe = SEAMS_CLI if env else os.path.join(ROOT, "scssim_amd", "bin", "scssim")   # a test seam in the environment: the seams build of the CLI (scs_seams.h)
    p = subprocess.Popen([exe, "genreads"] + args, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        _, err = p.communicate(timeout=timeout)
    finally:
        left = subprocess.run(["pgrep", "-g", str(p.pid)], capture_output=True, text=True).stdout.split()
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
    return p.returncode, err, left


def test_cli_forks_two_ranks_on_one_gpu_and_merges_their_shards(oracle_bin, models, golden_inputs, tmp_path):
    """`scssim genreads --gpus 2 --one-device --host-collectives`: the CLI's multi-rank path end to end on this box's one GPU --
    fork before the first HIP call, the communicator id down the pipes, a shard + index per rank (exchanges through host memory
    shared by the ranks: RCCL refuses two ranks on one device), rank 0's watcher reaping rank 1, the native merge -- and the
    files equal the oracle's.  Then with part files per rank and --keep-shards."""
    fa, prof = golden_inputs["g2_xten_pe_nblock"], models["Illumina_HiSeqXTen"]
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, fa, prof, prefix, ["-c", "3"], 31)
    want = [open(prefix + s, "rb").read() for s in ("_1.fq", "_2.fq")]
    out = str(tmp_path / "cli")
    rc, err, left = _cli(["-i", fa, "-m", prof, "-c", "3", "-o", out, "--seed", "31", "--gpus", "2", "--one-device", "--host-collectives"])
    assert rc == 0, err
    assert "2 ranks, collectives: host memory" in err and not left
    assert [open(out + s, "rb").read() for s in ("_1.fq", "_2.fq")] == want
    assert not os.path.exists(out + ".r0_1.fq") and not os.path.exists(out + ".r1.idx")
    out = str(tmp_path / "cli3")
    rc, err, left = _cli(["-i", fa, "-m", prof, "-c", "3", "-o", out, "--seed", "31", "--gpus", "3", "--one-device", "--host-collectives", "--writers", "2", "--keep-shards"],
                         env={"SCS_TEST_BATCH_SHIFT": "9"})
    assert rc == 0, err
    assert os.path.exists(out + ".r2.p01_2.fq") and os.path.exists(out + ".r2.idx") and not os.path.exists(out + "_1.fq")
    scssim_amd.merge_fastq_shards(out, 3, paired=True)
    assert [open(out + s, "rb").read() for s in ("_1.fq", "_2.fq")] == want


def test_cli_five_ranks_on_one_gpu(oracle_bin, models, golden_inputs, tmp_path):
    """`scssim genreads --gpus 5` through the same seam (five ranks + this process: the most the pool lets share one card): more ranks
    than some list segments have amplicons, shards with empty segments, the merge over five indexes -- and the oracle's bytes."""
    fa, prof = golden_inputs["g2_xten_pe_nblock"], models["Illumina_HiSeqXTen"]
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, fa, prof, prefix, ["-c", "3"], 37)
    out = str(tmp_path / "cli5")
    rc, err, left = _cli(["-i", fa, "-m", prof, "-c", "3", "-o", out, "--seed", "37", "--gpus", "5", "--one-device", "--host-collectives"])
    assert rc == 0 and not left, err
    for s in ("_1.fq", "_2.fq"):
        assert open(out + s, "rb").read() == open(prefix + s, "rb").read(), s


def test_cli_sharded_job_with_primer_types_running_dry(oracle_bin, models, repeat_genome, tmp_path):
    """`scssim genreads --gpus 3 -p 10000 -r 1e-8` on the repeat-rich genome: the pass in which primer types run dry is run again segment
    by segment across the forked ranks (the exchanges through the CLI's host-memory seam), and the merged files are the oracle's."""
    prof = models["Illumina_HiSeq2500"]
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, repeat_genome, prof, prefix, ["-c", "0.5", "-p", "10000", "-r", "1e-8"], 53)
    out = str(tmp_path / "cli")
    rc, err, left = _cli(["-i", repeat_genome, "-m", prof, "-c", "0.5", "-p", "10000", "-r", "1e-8", "-o", out, "--seed", "53", "--gpus", "3", "--one-device", "--host-collectives"])
    assert rc == 0 and not left, err
    for s in ("_1.fq", "_2.fq"):
        assert open(out + s, "rb").read() == open(prefix + s, "rb").read(), s


@pytest.mark.parametrize("where,who", [("amplify", 1), ("reads", 1), ("comm", 2), ("amplify", 0)])
def test_cli_rank_failure_ends_the_job(where, who, models, golden_inputs, tmp_path):
    """A rank that dies mid-job (injected: SCS_TEST_FAIL_AT / SCS_TEST_FAIL_RANK) leaves its siblings inside an exchange that
    never completes.  Rank 0's watcher must notice, end the other ranks and exit non-zero within seconds -- no hang, no orphan
    holding the GPU; when rank 0 itself fails it ends its children first."""
    import time
    fa, prof = golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"]
    t0 = time.time()
    rc, err, left = _cli(["-i", fa, "-m", prof, "-c", "2", "-o", str(tmp_path / "f"), "--seed", "3", "--gpus", "3", "--one-device", "--host-collectives"],
                         env={"SCS_TEST_FAIL_AT": where, "SCS_TEST_FAIL_RANK": str(who)}, timeout=120)
    assert rc in (2, 3) and time.time() - t0 < 90, (rc, err)
    assert not left, "ranks left behind: %s" % left
    assert ("rank %d of the sharded job failed" % who) in err or "test failure injected" in err
    assert not os.path.exists(str(tmp_path / "f") + "_1.fq")


def test_cli_writers_and_unopenable_output(oracle_bin, models, golden_inputs, tmp_path):
    """--writers K on one GPU: part files whose concatenation is the reference's file; an output that cannot be opened ends with the
    reference's exit(-1) (SeqWriter.cpp:17-33)."""
    fa, prof = golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"]
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, fa, prof, prefix, ["-c", "2"], 99)
    out = str(tmp_path / "w")
    rc, err, _ = _cli(["-i", fa, "-m", prof, "-c", "2", "-o", out, "--seed", "99", "--writers", "3"], env={"SCS_TEST_BATCH_SHIFT": "8"})
    assert rc == 0, err
    files = scssim_amd.part_paths(out, 3, True)
    for m, s in enumerate(("_1.fq", "_2.fq")):
        assert _cat(files[m]) == open(prefix + s, "rb").read()
    rc, err, _ = _cli(["-i", fa, "-m", prof, "-c", "2", "-o", str(tmp_path / "nodir" / "x"), "--seed", "99"])
    assert rc == 255 and "can not open fastq file" in err


def test_cli_option_errors_exit_like_the_reference(models, golden_inputs, tmp_path):
    """src/scssim.cpp:349-393: every option check prints the reference's line and leaves with status 1 (before any GPU call)."""
    fa, prof, out = golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"], str(tmp_path / "o")
    for args, msg in ((["-m", prof, "-o", out], "reference file (.fasta) not specified"), (["-i", fa, "-o", out], "sequencing profile must be specified"),
                      (["-i", fa, "-m", prof], "prefix of output file not specified"), (["-i", fa, "-m", prof, "-o", out, "-p", "999"], "should be at least 1000"),
                      (["-i", fa, "-m", prof, "-o", out, "-r", "1e-7"], "should be in 0~1e-8"), (["-i", fa, "-m", prof, "-o", out, "-l", "XX"], "sequence layout incorrectly specified"),
                      (["-i", fa, "-m", prof, "-o", out, "-c", "0"], "coverage not properly specified"), (["-i", fa, "-m", prof, "-o", out, "-t", "0"], "threads should be a positive integer")):
        rc, err, _ = _cli(args)
        assert rc == 1 and msg in err, (args, rc, err)
    assert not os.path.exists(out + "_1.fq")


def test_cli_input_errors_exit_like_the_reference(models, golden_inputs, tmp_path):
    """The reference's exit sites behind the option checks ("Error behaviour"): an input FASTA that cannot be opened or holds
    no sequence -> status 1 (Fasta.cpp:236-237, Genome.cpp:190-193); a model that cannot be opened -> 255 (Profile.cpp:933-936: exit(-1));
    a model with a section missing or a malformed line -> 1 (Profile.cpp:950-1231)."""
    fa, prof, out = golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"], str(tmp_path / "o")
    rc, err, _ = _cli(["-i", str(tmp_path / "missing.fa"), "-m", prof, "-o", out])
    assert rc == 1 and "could not open" in err, err
    empty = tmp_path / "empty.fa"; empty.write_text("")
    rc, err, _ = _cli(["-i", str(empty), "-m", prof, "-o", out])
    assert rc == 1 and "reference sequence cannot be empty" in err, err
    rc, err, _ = _cli(["-i", fa, "-m", str(tmp_path / "missing.profile"), "-o", out])
    assert rc == 255 and "can not open file" in err, err
    text = open(prof).read()
    cut = tmp_path / "cut.profile"; cut.write_text(text[:text.index("[Base Quality Distribution]")])
    rc, err, _ = _cli(["-i", fa, "-m", str(cut), "-o", out])
    assert rc == 1 and "corrupted model file" in err, err
    bad = tmp_path / "bad.profile"; bad.write_text(text.replace("kmer: XXA", "kmer: XXZ", 1))          # a k-mer that is not one
    assert bad.read_text() != text
    rc, err, _ = _cli(["-i", fa, "-m", str(bad), "-o", out])
    assert rc == 1 and "model file" in err, err
    assert not os.path.exists(out + "_1.fq")


def test_bgzf_made_on_the_gpu_inflates_to_the_oracle_text(oracle_bin, models, golden_inputs, tmp_path):
    """scs_yield_reads_files_ex(..., bgzf): the batches' text becomes BGZF blocks on the GPU (histogram, Huffman lengths, bit packing,
    CRC-32: scs_bgzf.hip), crosses PCIe compressed and lands in <prefix>_1.fq.gz / part files.  zlib is the checker: every part is
    a run of well-framed BGZF blocks + the end-of-file block; gunzip of the parts in order is the oracle's text, byte for byte --
    one writer and several, generations, small batches (blocks cut at batch ends) and one big batch (63 KB blocks), PE and SE."""
    import gzip
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "7000000,5000000", "--seed", "31", "--n-block", "20000", "--simu-out", fa])
    prof = models["Illumina_HiSeq2500"]
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, fa, prof, prefix, ["-c", "4"], 41, threads=min(32, os.cpu_count() or 1))
    want = [open(prefix + s, "rb").read() for s in ("_1.fq", "_2.fq")]
    eof = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    code = '''
import sys
sys.path.insert(0, %r)
import scssim_amd
g = scssim_amd.GenReads(profile=%r, input_fasta=%r, coverage=4.0, seed=41)
g.create_frags(); g.amplify(); g.allocate_reads(0)
for K, G in %%s:
    g.yield_reads_files(%r + "_%%%%d_%%%%d" %%%% (K, G), K, G, True)
    st = g.stats()
    print("RATIO", K, G, sum(st["fastq_bytes"]) / sum(st["sink_bytes"]))
''' % (ROOT, prof, fa, str(tmp_path / "z"))
    for shift, combos in ((None, [(1, 1), (3, 1)]), ("13", [(1, 1), (2, 3), (5, 1)])):
        env = dict(os.environ) if shift is None else seams_env(SCS_TEST_BATCH_SHIFT=shift)
        r = subprocess.run([sys.executable, "-c", code % repr(combos)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        ratios = [float(l.split()[3]) for l in r.stdout.splitlines() if l.startswith("RATIO")]
        assert len(ratios) == len(combos) and min(ratios) > 1.8, r.stdout
        for K, G in combos:
            base = str(tmp_path / "z") + "_%d_%d" % (K, G)
            files = scssim_amd.part_paths(base, K * G, True, ".fq.gz")
            for m in range(2):
                text = b""
                for f in files[m]:
                    z = open(f, "rb").read()
                    assert z.endswith(eof), f
                    blocks = scssim_amd.bgzf_blocks(z)
                    assert all(i <= 64512 for _, i in blocks) and blocks[-1][1] == 0
                    text += gzip.decompress(z)
                assert text == want[m], "bgzf K=%d G=%d mate %d (batch shift %s)" % (K, G, m + 1, shift)
    # single end, through the CLI
    fa3, prof3 = golden_inputs["g3_hiseq2000_se"], models["Illumina_HiSeq2000"]
    prefix = str(tmp_path / "orc_se")
    _oracle(oracle_bin, fa3, prof3, prefix, ["-c", "2", "-l", "SE"], 42)
    out = str(tmp_path / "cli_se")
    rc, err, _ = _cli(["-i", fa3, "-m", prof3, "-c", "2", "-l", "SE", "-o", out, "--seed", "42", "--bgzf"])
    assert rc == 0, err
    assert gzip.decompress(open(out + ".fq.gz", "rb").read()) == open(prefix + ".fq", "rb").read()


def test_generations_finish_their_parts_while_the_job_runs(oracle_bin, models, tmp_path):
    """writers = 2, generations = 4: eight parts per mate made generation by generation.  A watcher thread sees part p complete (its
    size final) by the time part p + 2 exists, while the job is still running; the parts concatenate to the oracle's file."""
    import threading
    import time
    fa = str(tmp_path / "simu.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "7000000,5000000", "--seed", "31", "--n-block", "20000", "--simu-out", fa])
    prof = models["Illumina_HiSeqXTen"]
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, fa, prof, prefix, ["-c", "4"], 8, threads=min(32, os.cpu_count() or 1))
    code = '''
import sys
sys.path.insert(0, %r)
import scssim_amd
g = scssim_amd.GenReads(profile=%r, input_fasta=%r, coverage=4.0, seed=8)
g.create_frags(); g.amplify(); g.allocate_reads(0)
g.yield_reads_files(%r, 2, 4)
''' % (ROOT, prof, fa, str(tmp_path / "gen"))
    files = scssim_amd.part_paths(str(tmp_path / "gen"), 8, True)
    seen = {}
    stop = [False]

    def watch():
        while not stop[0]:
            for p in range(6):
                if p not in seen and os.path.exists(files[0][p + 2]):
                    seen[p] = os.path.getsize(files[0][p])          # final by the rule
            time.sleep(0.002)
    th = threading.Thread(target=watch); th.start()
    try:
        r = subprocess.run([sys.executable, "-c", code], env=seams_env(SCS_TEST_BATCH_SHIFT="10"), capture_output=True, text=True, timeout=600)
    finally:
        stop[0] = True; th.join()
    assert r.returncode == 0, r.stdout + r.stderr
    for m, s in enumerate(("_1.fq", "_2.fq")):
        assert _cat(files[m]) == open(prefix + s, "rb").read()
    assert len(seen) >= 4, "the watcher saw too few parts appear while the job ran"
    for p, size in seen.items():
        assert size == os.path.getsize(files[0][p]) and size > 0, "part %d changed after part %d existed" % (p, p + 2)


@pytest.mark.parametrize("layout", ["regular", "crlf_and_short_last_record"])
def test_sharded_ranks_stage_only_their_own_stretch_of_the_genome(layout, oracle_bin, models, tmp_path):
    """A shard of a sharded job reads, uploads and indexes only the stretch of the genome its own fragments cover when the FASTA
    has a usable .fai ("genome slices needed per GPU = its own fragments only"): three ranks on a 3-record genome
    stage about a third each (scs_stats.staged_bases) and the merged output still equals the whole job's; without the index (or
    with SCS_STAGE_WHOLE) every rank stages everything, same output."""
    import shutil
    import socket
    fa0 = str(tmp_path / "simu0.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "900000,500000,313131", "--seed", "5", "--n-block", "7000", "--lower-frac", "0.03", "--simu-out", fa0])
    fa = str(tmp_path / "simu.fa")
    if layout == "regular":
        shutil.copyfile(fa0, fa)
    else:                                                               # CRLF line ends, 70 columns, no newline at the end of the file
        recs = open(fa0).read().split(">")[1:]
        with open(fa, "wb") as f:
            for i, r in enumerate(recs):
                name, seq = r.split("\n", 1)
                seq = seq.replace("\n", "")
                body = "\r\n".join(seq[k:k + 70] for k in range(0, len(seq), 70))
                f.write((">%s\r\n%s" % (name, body)).encode() + (b"" if i == len(recs) - 1 else b"\r\n"))
    prof = models["Illumina_HiSeq2500"]
    whole = str(tmp_path / "whole")
    _oracle(oracle_bin, fa, prof, whole, ["-c", "3"], 91)
    total = 2 * (900000 + 500000 + 313131)

    def sharded(tag, env_extra):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
        procs = []
        for r in range(3):
            env = (seams_env if env_extra else dict)(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LOCAL_RANK="0", **env_extra)
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), fa, prof, str(tmp_path / tag), "3", "PE", "91", "device"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=600)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        staged = {}
        for o in outs:
            for l in o.splitlines():
                if l.startswith("STAGED"):
                    _, r, a, b = l.split(); staged[int(r)] = (int(a), int(b))
        scssim_amd.merge_fastq_shards(str(tmp_path / tag), 3, paired=True)
        for suffix in ("_1.fq", "_2.fq"):
            assert open(str(tmp_path / tag) + suffix, "rb").read() == open(whole + suffix, "rb").read(), (tag, suffix)
        return staged
    if os.path.exists(fa + ".fai"):
        os.remove(fa + ".fai")
    st = sharded("noidx", {})                                           # no index yet: whole-file staging (which writes it; a rank that comes late may find it there)
    assert all(b == total for _, b in st.values()) and max(a for a, _ in st.values()) == total and os.path.exists(fa + ".fai")
    st = sharded("sliced", {})
    assert all(b == total for _, b in st.values())
    assert sum(a for a, _ in st.values()) <= total + 3 * 100001 and max(a for a, _ in st.values()) < 0.5 * total, st
    st = sharded("forced_whole", {"SCS_STAGE_WHOLE": "1"})
    assert all(v == (total, total) for v in st.values())


def test_a_job_without_reads_leaves_well_formed_empty_outputs(models, tmp_path):
    """Zero requested reads (coverage too low for one read) and a genome too short to amplify: the file sink still leaves the files a
    consumer expects -- empty text files, every part of a parts job with its index, BGZF files that are just the end-of-file block
    (gunzip gives the empty string) -- and the merges accept them."""
    import gzip
    rng = __import__("numpy").random.default_rng(3)
    fa = str(tmp_path / "tiny.fa")
    with open(fa, "w") as f:
        for name in ("1_1_20000", "1_2_20000"):
            f.write(">%s\n%s\n" % (name, "".join(rng.choice(list("ACGT"), size=20000))))
    g = scssim_amd.GenReads(profile=models["Illumina_HiSeq2500"], input_fasta=fa, coverage=0.001, seed=7)
    g.create_frags(); g.amplify(); g.allocate_reads(0)
    assert g.stats()["reads_requested"] == 0
    g.yield_reads_files(str(tmp_path / "a"))
    assert os.path.getsize(str(tmp_path / "a") + "_1.fq") == 0 and os.path.getsize(str(tmp_path / "a") + "_2.fq") == 0
    g.yield_reads_files(str(tmp_path / "b"), 3, 2)
    files = scssim_amd.part_paths(str(tmp_path / "b"), 6, True)
    assert all(os.path.getsize(f) == 0 for m in files for f in m) and os.path.exists(str(tmp_path / "b") + ".parts")
    scssim_amd.merge_fastq_parts(str(tmp_path / "b"))
    assert os.path.getsize(str(tmp_path / "b") + "_1.fq") == 0 and not os.path.exists(files[0][0])
    g.yield_reads_files(str(tmp_path / "c"), 2, 1, True)
    for f in [x for m in scssim_amd.part_paths(str(tmp_path / "c"), 2, True, ".fq.gz") for x in m]:
        z = open(f, "rb").read()
        assert len(z) == 28 and gzip.decompress(z) == b""
    assert g.stats()["pairs_written"] == 0 and g.stats()["sink_bytes"] == [0, 0]
