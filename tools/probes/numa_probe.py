#!/usr/bin/env python3
"""Where the GPU sits (NUMA node of its PCIe function), which CPUs are local to it, and what tmpfs writes cost from local and
from remote CPUs: python tools/probes/numa_probe.py"""
import glob, os, subprocess, sys, time, threading
import torch
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
node = open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip()
print("GPU 0 at", bdf, "numa node", node)
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node*")):
    nodes[os.path.basename(d)[4:]] = open(d + "/cpulist").read().strip()
print("nodes:", nodes)
print("affinity now:", len(os.sched_getaffinity(0)), "cpus")


def cpus(lst):
    out = set()
    for part in lst.split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


buf = os.urandom(1 << 20) * 64
def write(k, d, n):
    fd = os.open(os.path.join(d, "numaprobe_%d_%d" % (os.getpid(), k)), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    for i in range(n):
        os.pwrite(fd, buf, i * len(buf))
    os.close(fd)
def run(T, n=48):
    th = [threading.Thread(target=write, args=(k, "/dev/shm", n)) for k in range(T)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    for f in glob.glob("/dev/shm/numaprobe_%d_*" % os.getpid()):
        os.unlink(f)
    return T * n * len(buf) / 1e9 / dt
# pinned D2H into a buffer allocated under each affinity, then tmpfs writes under each affinity
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for name, lst in [("all", None)] + [("node" + k, v) for k, v in nodes.items()]:
    os.sched_setaffinity(0, cpus(lst) if lst else set(range(os.cpu_count())))
    h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        h.copy_(x, non_blocking=True)
    torch.cuda.synchronize(); d2h = 8 * (1 << 30) / 1e9 / (time.perf_counter() - t0)
    print("%-6s: pinned D2H %.1f GB/s; tmpfs fresh writes 12 threads %.1f GB/s, 16 threads %.1f GB/s" % (name, d2h, run(12), run(16)), flush=True)
    del h
