#!/usr/bin/env python3
"""What a part file's pages cost: T threads pwrite() G GiB each into tmpfs files that are (a) new, (b) there already (no O_TRUNC:
the pages are overwritten where they lie), (c) new while T/3 other threads unlink the previous round's files (the bench loop).
    python tools/probes/overwrite_probe.py [threads=12] [GiB per thread=6]"""
import os, sys, threading, time
T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
G = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
d = "/dev/shm/ovw_probe"; os.makedirs(d, exist_ok=True)
chunk = bytes(os.urandom(1 << 20)) * 8                       # 8 MiB
n = int(G * (1 << 30)) // len(chunk)
def write(path, flags):
    fd = os.open(path, flags, 0o644)
    for i in range(n): os.pwrite(fd, chunk, i * len(chunk))
    os.ftruncate(fd, n * len(chunk)); os.close(fd)
def run(tag, names, flags, also=None):
    th = [threading.Thread(target=write, args=(p, flags)) for p in names]
    ex = [threading.Thread(target=f) for f in (also or [])]
    t0 = time.time()
    for t in th + ex: t.start()
    for t in th: t.join()
    t1 = time.time()
    for t in ex: t.join()
    print(f"{tag}: {T * n * len(chunk) / (t1 - t0) / 1e9:.1f} GB/s ({t1 - t0:.2f} s; helpers done +{time.time() - t1:.2f} s)", flush=True)
a = [f"{d}/a{i}" for i in range(T)]; b = [f"{d}/b{i}" for i in range(T)]
run("new files            ", a, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
run("overwrite in place   ", a, os.O_WRONLY | os.O_CREAT)
run("overwrite in place #2", a, os.O_WRONLY | os.O_CREAT)
def unlinker(part):
    def f():
        for p in part: os.unlink(p)
    return f
k = max(1, T // 3)
run("new + unlink previous", b, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, [unlinker(a[i::k]) for i in range(k)])
run("O_TRUNC over existing", b, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
for p in b: os.unlink(p)
os.rmdir(d)
