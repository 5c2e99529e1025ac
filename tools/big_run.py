#!/usr/bin/env python3
"""Whole-genome-scale run through the C ABI with the FASTQ left on the device (sink = None: generate + count).
Usage: big_run.py <total_Mb> <n_records> [coverage]   -- synthetic diploid genome, HiSeqXTen model (151 bp)."""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scssim_amd  # noqa: E402


def main():
    mb, nrec = int(sys.argv[1]), int(sys.argv[2])
    cov = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    td = tempfile.mkdtemp(prefix="scsbig_", dir=os.environ.get("TMPDIR", "/tmp"))
    fa = os.path.join(td, "simu.fa")
    per = mb * 1000000 // nrec
    t0 = time.time()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", ",".join([str(per)] * nrec), "--seed", "3000",
                           "--first-chr", "1", "--simu-out", fa])
    prof = os.path.join(td, "xten.profile")
    open(prof, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "models", "Illumina_HiSeqXTen.profile.gz")).read())
    t1 = time.time()
    print("genome written in %.1f s (%.1f GB)" % (t1 - t0, os.path.getsize(fa) / 1e9), flush=True)
    g = scssim_amd.GenReads(profile=prof, coverage=cov, seed=1)
    g.load_genome(fa)
    t2 = time.time()
    print("FASTA parsed + uploaded + bit index in %.1f s" % (t2 - t1), flush=True)
    os.remove(fa)
    # two passes over the hot path: the first maps the device buffers (cost depends on the state of the box's memory: a
    # box handed over with freshly freed memory has been seen to spend 2 s there at 3 Gb), the second runs in them
    for label, seed in (("first pass (maps device memory)", 1), ("second pass (buffers mapped)", 2)):
        g.set_seed(seed)
        t3 = time.time()
        g.create_frags(); g.amplify(); t4 = time.time()
        print("%s: amplify %.2f s" % (label, t4 - t3), g.stats(), flush=True)
        g.allocate_reads(0); t5 = time.time()
        print("allocate %.2f s" % (t5 - t4), flush=True)
        g.yield_reads(collect=False); t6 = time.time()
        st = g.stats()
        kt = g.kernel_times()                                   # of this pass (the library restarts them per amplify / yield call)
        print("yield %.2f s" % (t6 - t5))
        print("TOTAL hot path %.2f s for %d pairs -> %.2f M pairs/s ; fastq bytes %s ; kernels %s" %
              (t6 - t3, st["pairs_written"], st["pairs_written"] / (t6 - t3) / 1e6, st["fastq_bytes"], kt), flush=True)


if __name__ == "__main__":
    main()
