#!/usr/bin/env python3
"""configs[1] (1 Mb x 2 haplotypes, PE150 30x) with the text left in HBM: wall per job over N jobs (the bench's `sweep` line), for a
launch-latency view of the small configurations.  Under `rocprofv3 --kernel-trace --stats` the kernel statistics give launches per job
and the GPU's busy time:  python tools/small_job.py [--mb 1] [--jobs 50]"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=1.0)
    ap.add_argument("--jobs", type=int, default=50)
    a = ap.parse_args()
    import torch
    import bench
    import scssim_amd
    dev = torch.device("cuda", 0)
    td = tempfile.mkdtemp(prefix="scs_small_")
    prof = bench.make_profile(td)
    stream = torch.cuda.Stream()
    names, rl, bases = bench.synth_genome(torch, dev, [int(a.mb * 1e6)], 7000 + int(a.mb))
    g = scssim_amd.GenReads(profile=prof, coverage=30.0, isize=260, layout="PE", seed=1, device=0, stream=stream.cuda_stream)
    g.upload_genome_device(names, rl, bases.data_ptr())

    def one(i):
        g.set_seed(500 + i)
        g.create_frags(); g.amplify(); g.allocate_reads(0)
        g.yield_reads_sink(None)
        return g.stats()
    one(0); one(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); pairs = 0; t_st = [0.0] * 4
    for i in range(a.jobs):
        g.set_seed(600 + i)
        ta = time.perf_counter(); g.create_frags(); tb = time.perf_counter(); g.amplify(); tc = time.perf_counter(); g.allocate_reads(0); td_ = time.perf_counter()
        g.yield_reads_sink(None); te = time.perf_counter()
        pairs += g.stats()["pairs_written"]
        for k, v in enumerate((tb - ta, tc - tb, td_ - tc, te - td_)):
            t_st[k] += v
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = g.stats()
    print("%.1f Mb: %.3f ms per job (frags %.3f, amplify %.3f, allocate %.3f, reads %.3f), %.1f M pairs/s; %d pairs, %d amplicons per job; stock checks %d" %
          (a.mb, 1e3 * dt / a.jobs, *(1e3 * v / a.jobs for v in t_st), pairs / dt / 1e6, pairs // a.jobs, st["semi_amplicons"] + st["full_amplicons"], st["stock_checks"]))


if __name__ == "__main__":
    main()
