import os, sys, time, tempfile, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, scssim_amd
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
td = tempfile.mkdtemp()
prof = bench.make_profile(td)
lens = bench.record_lengths(600)
names, rl, bases = bench.synth_genome(torch, dev, lens, 3000)
g = scssim_amd.GenReads(profile=prof, coverage=30.0, seed=1)
g.upload_genome_device(names, rl, bases.data_ptr()); del bases
g.create_frags(); g.amplify(); g.allocate_reads(0)
t = time.perf_counter(); g.yield_reads_sink(None); t0 = time.perf_counter() - t
nb = [0]
def nop(_u, p1, n1, p2, n2):
    nb[0] += n1 + n2
    return 0
t = time.perf_counter(); g.yield_reads_sink(nop); t1 = time.perf_counter() - t
print("NULL sink %.2f s ; no-op sink (D2H only) %.2f s for %.1f GB -> %.1f GB/s" % (t0, t1, nb[0] / 1e9, nb[0] / 1e9 / t1), flush=True)
import glob
for thr, gen in ((1, 1), (4, 1), (8, 1), (12, 1), (12, 4), (16, 1)):
    t = time.perf_counter(); g.yield_reads_files("/dev/shm/sinkdiag", thr, gen); t2 = time.perf_counter() - t
    print("file sink /dev/shm, %d writers x %d generations: %.2f s -> %.1f GB/s" % (thr, gen, t2, nb[0] / 1e9 / t2), flush=True)
    for f in glob.glob("/dev/shm/sinkdiag*"):
        os.remove(f)
t = time.perf_counter(); g.yield_reads_files("/dev/shm/sinkdiag", 12, 1, True); t2 = time.perf_counter() - t
print("file sink /dev/shm, BGZF, 12 writers: %.2f s -> %.1f GB/s of text, %.2fx" % (t2, nb[0] / 1e9 / t2, sum(g.stats()["fastq_bytes"]) / sum(g.stats()["sink_bytes"])), flush=True)
for f in glob.glob("/dev/shm/sinkdiag*"):
    os.remove(f)
# raw tmpfs write bandwidth
import numpy as np
buf = np.random.randint(0, 255, size=1 << 28, dtype=np.uint8).tobytes()
def wr(path, n):
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    for _ in range(n):
        os.write(fd, buf)
    os.close(fd)
for nt in (1, 4, 8):
    ths = [threading.Thread(target=wr, args=("/dev/shm/raw%d" % i, 8)) for i in range(nt)]
    t = time.perf_counter(); [x.start() for x in ths]; [x.join() for x in ths]; dt = time.perf_counter() - t
    print("raw tmpfs write, %d threads x 2 GB: %.1f GB/s" % (nt, nt * 8 * len(buf) / 1e9 / dt), flush=True)
    for i in range(nt):
        os.remove("/dev/shm/raw%d" % i)
# pinned D2H bandwidth
x = torch.empty(1 << 30, dtype=torch.uint8, device=dev); h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(8):
    h.copy_(x, non_blocking=True)
torch.cuda.synchronize(); print("torch pinned D2H: %.1f GB/s" % (8 * (1 << 30) / 1e9 / (time.perf_counter() - t)), flush=True)
