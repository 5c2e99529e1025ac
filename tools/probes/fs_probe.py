#!/usr/bin/env python3
"""Buffered-write, overwrite and unlink rates of a file system from T threads on T files (what bounds a FASTQ sink on the GPU
box): python tools/probes/fs_probe.py DIR [threads] [GB per file].  os.pwrite / os.unlink release the GIL."""
import os, sys, threading, time

d = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
gb = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
buf = os.urandom(1 << 20) * 64                       # 64 MB
n = int(gb * (1 << 30)) // len(buf)
paths = [os.path.join(d, "fsprobe_%d_%d" % (os.getpid(), k)) for k in range(T)]


def run(fn):
    th = [threading.Thread(target=fn, args=(k,)) for k in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0


def write(k, flags=os.O_WRONLY | os.O_CREAT | os.O_TRUNC):
    fd = os.open(paths[k], flags, 0o644)
    for i in range(n):
        os.pwrite(fd, buf, i * len(buf))
    os.close(fd)


tot = T * n * len(buf) / 1e9
t = run(write); print("%s: %d threads x %d files fresh   : %.1f GB in %.2f s = %.1f GB/s" % (d, T, T, tot, t, tot / t), flush=True)
t = run(lambda k: write(k, os.O_WRONLY)); print("%s: overwrite in place (no truncate)  : %.1f GB in %.2f s = %.1f GB/s" % (d, tot, t, tot / t), flush=True)
t = run(lambda k: os.unlink(paths[k])); print("%s: unlink, %d threads               : %.1f GB in %.2f s = %.1f GB/s" % (d, T, tot, t, tot / t), flush=True)
t = run(write)
t0 = time.perf_counter()
for p in paths: os.unlink(p)
t = time.perf_counter() - t0; print("%s: unlink, one thread               : %.1f GB in %.2f s = %.1f GB/s" % (d, tot, t, tot / t), flush=True)
