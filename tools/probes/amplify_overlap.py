#!/usr/bin/env python3
"""Would amplification kernels gain from running beside each other?  Two ctxs (streams, host threads) amplify a 600 Mb genome at once,
against the same two jobs one after the other: the aggregate says what an overlap of k_errs<semi->full> with the next cycle's
k_attach_dense could be worth at best.    python3 tools/probes/amplify_overlap.py [genome Mb]"""
import os, sys, tempfile, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, scssim_amd, bench
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
td = tempfile.mkdtemp(prefix="scs_ov_"); prof = bench.make_profile(td)
lens = bench.record_lengths(mb)
names, rl, bases = bench.synth_genome(torch, dev, lens, 3000)
gs = []
for k in range(2):
    st = torch.cuda.Stream()
    g = scssim_amd.GenReads(profile=prof, coverage=30.0, isize=260, layout="PE", seed=1, device=0, stream=st.cuda_stream)
    g.upload_genome_device(names, rl, bases.data_ptr())
    gs.append((g, st))
def job(k, seed, what):
    g = gs[k][0]; g.set_seed(seed); g.create_frags(); g.amplify()
    if what == "all": g.allocate_reads(0); g.yield_reads_sink(None)
for what in ("amplify", "all"):
    for k in range(2): job(k, 10 + k, what)                       # warm-up: buffers mapped
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter(); job(0, 100 + rep, what); job(1, 200 + rep, what); torch.cuda.synchronize(); seq = time.perf_counter() - t0
        th = [threading.Thread(target=job, args=(k, 300 + 10 * rep + k, what)) for k in range(2)]
        t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize(); con = time.perf_counter() - t0
        print("%s, %.0f Mb: two jobs one after the other %.1f ms, at once %.1f ms (%.1f %%)" % (what, mb, 1e3 * seq, 1e3 * con, 100.0 * (con / seq - 1.0)), flush=True)
