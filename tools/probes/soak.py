#!/usr/bin/env python3
"""Soak: N whole jobs on a 600 Mb genome in one process (text left in HBM, then through a counting sink), fresh seed each: free
device memory and the host's resident set before and after (a leak shows as a trend), every job's counts sane.  python tools/probes/soak.py [--jobs 60]"""
import argparse, os, sys, tempfile, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--jobs", type=int, default=60); ap.add_argument("--mb", type=float, default=600.0); a = ap.parse_args()
import torch, bench, scssim_amd
dev = torch.device("cuda", 0)
td = tempfile.mkdtemp(prefix="scs_soak_")
prof = bench.make_profile(td)
lens = bench.record_lengths(a.mb)
names, rl, bases = bench.synth_genome(torch, dev, lens, 3000)
stream = torch.cuda.Stream()
g = scssim_amd.GenReads(profile=prof, coverage=30.0, isize=260, layout="PE", seed=1, device=0, stream=stream.cuda_stream)
g.upload_genome_device(names, rl, bases.data_ptr()); del bases; torch.cuda.empty_cache()
seen = [0]
def sink(_u, p1, n1, p2, n2):
    seen[0] += n1 + n2
    return 0
def rss():                                  # current resident set (not the high-water mark)
    return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 1e9
marks = []
t0 = time.time()
for i in range(a.jobs):
    g.set_seed(9000 + i)
    g.create_frags(); g.amplify(); g.allocate_reads(0)
    g.yield_reads_sink(sink if i % 4 == 3 else None)
    st = g.stats()
    assert abs(st["pairs_written"] - int(sum(lens) * 30 / 150) // 2) <= 2 + st["reads_requested"] // 200000, st
    if i in (4, a.jobs // 4, a.jobs // 2, 3 * a.jobs // 4, a.jobs - 1):
        free, total = torch.cuda.mem_get_info()
        marks.append((i, free / 1e9, rss()))
        print("job %d: free HBM %.2f GB, host RSS %.2f GB, %.1f s" % (i, free / 1e9, rss(), time.time() - t0), flush=True)
assert abs(marks[-1][1] - marks[0][1]) < 2.0, "device memory drifts: %s" % marks
print("soak ok: %d jobs, %.1f GB through the counting sink" % (a.jobs, seen[0] / 1e9))
