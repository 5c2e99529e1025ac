#!/usr/bin/env python3
"""How many 1 Mb jobs (configs[1]) at once fill the chip?  bench.py's small_config leg with 4 / 8 / 16 / 32 ctxs (streams, host threads).
    python3 tools/probes/small_concurrency.py"""
import os, sys, tempfile, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, scssim_amd, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
td = tempfile.mkdtemp(prefix="scs_small_"); prof = bench.make_profile(td); stream = torch.cuda.Stream()
for c in (4, 8, 16, 32):
    r = bench.small_config(torch, scssim_amd, dev, stream, prof, 1.0, 30.0, "1 Mb", "/dev/shm", steps=5, concurrent=c)
    cj = r["concurrent_jobs"]
    print("%2d ctxs: %.1f M pairs/s aggregate, %.3f ms per job amortised (one job alone: %.2f ms)" % (c, cj["aggregate_pairs_per_s"] / 1e6, cj["ms_per_job_amortised"], r["ms_per_step_hbm"]), flush=True)
