import os, mmap, time, threading, ctypes
for f in ("enabled","shmem_enabled","defrag","hpage_pmd_size"):
    try: print(f, open("/sys/kernel/mm/transparent_hugepage/"+f).read().strip())
    except Exception as e: print(f, e)
print(os.popen("mount | grep -E 'shm|tmpfs' | head -5").read())
print(os.popen("grep -E 'Huge|Shmem' /proc/meminfo").read())
libc = ctypes.CDLL("libc.so.6", use_errno=True)
MADV_HUGEPAGE = 14
N = 2 << 30
src = bytearray(os.urandom(1 << 20)) * 256   # 256 MB
def w_pwrite(k):
    fd = os.open("/dev/shm/thp_probe_%d" % k, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    for o in range(0, N, len(src)): os.pwrite(fd, src, o)
    os.close(fd)
def w_mmap(k, huge):
    fd = os.open("/dev/shm/thp_probe_%d" % k, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    os.ftruncate(fd, N)
    m = mmap.mmap(fd, N)
    if huge:
        addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
        r = libc.madvise(ctypes.c_void_p(addr), ctypes.c_size_t(N), MADV_HUGEPAGE)
        if r != 0 and k == 0: print("madvise failed", ctypes.get_errno())
    for o in range(0, N, len(src)): m[o:o+len(src)] = src
    m.close(); os.close(fd)
def run(fn, T, *a):
    th = [threading.Thread(target=fn, args=(k,)+a) for k in range(T)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    for k in range(T):
        try: os.unlink("/dev/shm/thp_probe_%d" % k)
        except OSError: pass
    return T * N / 1e9 / dt
for T in (1, 12):
    print("T=%d pwrite %.1f GB/s  mmap %.1f GB/s  mmap+MADV_HUGEPAGE %.1f GB/s" % (T, run(w_pwrite, T), run(w_mmap, T, False), run(w_mmap, T, True)))
print(os.popen("grep -E 'ShmemHugePages|ShmemPmdMapped' /proc/meminfo").read())
