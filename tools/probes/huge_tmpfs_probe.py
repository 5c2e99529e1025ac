#!/usr/bin/env python3
"""Can an ordinary user get a tmpfs with huge pages on the GPU box (a mount of its own in a user + mount namespace), and what do 12 writer
threads reach there against /dev/shm?  (The sink's host side is page allocation and freeing in 4 KB pages: EXPERIMENTS.md.)
    python3 tools/probes/huge_tmpfs_probe.py"""
import os, sys, time, threading, ctypes, subprocess
def thp(f):
    try: return open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip()
    except Exception as e: return repr(e)
print("enabled:", thp("enabled"), "| shmem_enabled:", thp("shmem_enabled"), "| hpage_pmd_size:", thp("hpage_pmd_size"), flush=True)
libc = ctypes.CDLL("libc.so.6", use_errno=True)
CLONE_NEWNS, CLONE_NEWUSER = 0x00020000, 0x10000000
uid, gid = os.getuid(), os.getgid()
if libc.unshare(CLONE_NEWUSER | CLONE_NEWNS) != 0:
    print("unshare(CLONE_NEWUSER | CLONE_NEWNS) failed: errno", ctypes.get_errno(), os.strerror(ctypes.get_errno())); sys.exit(0)
try:
    open("/proc/self/setgroups", "w").write("deny")
    open("/proc/self/uid_map", "w").write("0 %d 1" % uid); open("/proc/self/gid_map", "w").write("0 %d 1" % gid)
except Exception as e:
    print("uid/gid map failed:", e); sys.exit(0)
mp = "/tmp/huge_tmpfs_probe"; os.makedirs(mp, exist_ok=True)
def mount(opts):
    r = libc.mount(b"none", mp.encode(), b"tmpfs", 0, opts.encode())
    if r != 0: print("mount -t tmpfs -o %s failed: errno %d %s" % (opts, ctypes.get_errno(), os.strerror(ctypes.get_errno())))
    return r == 0
N = 4 << 30; src = bytes(os.urandom(1 << 20)) * 64
def w(d, k):
    fd = os.open("%s/p%d" % (d, k), os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    for o in range(0, N, len(src)): os.pwrite(fd, src, o)
    os.close(fd)
def run(d, T):
    th = [threading.Thread(target=w, args=(d, k)) for k in range(T)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    hp = [l.strip() for l in open("/proc/meminfo") if l.startswith("ShmemHugePages")]
    t1 = time.perf_counter()
    for k in range(T): os.unlink("%s/p%d" % (d, k))
    du = time.perf_counter() - t1
    return T * N / 1e9 / dt, T * N / 1e9 / du, hp
for opts in ("huge=always,size=80g", "huge=within_size,size=80g", "size=80g"):
    if not mount(opts): continue
    for T in (1, 12):
        wr, ul, hp = run(mp, T)
        print("own tmpfs (%s), %2d threads: written %.1f GB/s, unlinked %.1f GB/s  %s" % (opts, T, wr, ul, hp), flush=True)
    libc.umount(mp.encode())
for T in (1, 12):
    wr, ul, hp = run("/dev/shm", T)
    print("/dev/shm, %2d threads: written %.1f GB/s, unlinked %.1f GB/s  %s" % (T, wr, ul, hp), flush=True)
# does the GPU still open from inside the namespace?
print(subprocess.run([sys.executable, "-c", "import torch; print('cuda in the namespace:', torch.cuda.is_available(), torch.cuda.device_count())"], capture_output=True, text=True).stdout.strip())
