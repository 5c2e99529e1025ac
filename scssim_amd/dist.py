"""Multi-process glue (one process per GPU / shard): collective hooks over torch.distributed and the
merge of per-shard FASTQ pools into the single-job order.

A job is sharded by fragment lineage.  What crosses shards:
  * each pass: ONE all-reduce of the 65536 primer-stock decrements + a 16-word tail that carries what setPrimers needs
    from the other shards (new semi amplicons: count and length; the budgets handed out)          (256 KB, 11 per job)
  * read allocation: all-reduce of the 5x6 segment sizes + all-gather of the GC weights
  * output: every record name carries the amplicon's index in the whole job's list, and each shard's pool is
    already sorted by it, so the writer k-way merges the pools (`merge_fastq`).
The hooks work on host buffers (numpy views of the C pointers); with the NCCL(=RCCL) backend they are staged through
a device tensor, with gloo they stay on the CPU.
"""
import ctypes as C
import heapq

import numpy as np

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64)
ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64))
ALLREDUCE_DEV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int)
ALLGATHER_DEV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)


class _DevArray:
    """Zero-copy view of raw device memory for torch (CUDA array interface; works on ROCm builds too)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 3}


class Collectives:
    """ctypes callbacks implementing the scso_* / scs_* collective hooks with torch.distributed."""

    def __init__(self, device=None, stream=None):
        """stream: the torch.cuda.Stream the GenReads ctx runs on (GenReads(stream=stream.cuda_stream)); the device
        hooks issue their collectives on it so they are ordered with the library's kernels."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.stream = stream
        self._nccl = dist.get_backend() == "nccl"
        # the hooks are called from the thread that drives the job: make the ctx stream that thread's current stream once,
        # instead of entering a stream context per collective (a Python context manager per call costs ~10 us)
        self._stream_is_current = False
        if stream is not None and self._nccl:
            torch.cuda.set_stream(stream)
            self._stream_is_current = True
        self.device = device if device is not None else ("cuda" if self._nccl else "cpu")
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.allreduce_cb = ALLREDUCE_FN(self._allreduce)
        self.allgatherv_cb = ALLGATHERV_FN(self._allgatherv)
        self.allreduce_dev_cb = ALLREDUCE_DEV_FN(self._allreduce_dev)
        self.allgather_dev_cb = ALLGATHER_DEV_FN(self._allgather_dev)
        self.calls = dict(allreduce=0, allgatherv=0, allreduce_dev=0, allgather_dev=0, bytes=0)
        self._views = {}

    # ---- device-memory hooks: the tensors alias the library's HBM buffers; with the NCCL (= RCCL) backend the
    # collective runs on torch's current stream (give GenReads that stream), with gloo it is staged through the CPU.
    def _dev_tensor(self, ptr, n, elem_bytes):
        key = (int(ptr), int(n), int(elem_bytes))
        t = self._views.get(key)
        if t is None:                                            # the library reuses its buffers: cache the aliasing views
            t = self.torch.as_tensor(_DevArray(ptr, n, "<i8" if elem_bytes == 8 else ("<i4" if elem_bytes == 4 else "|u1")), device="cuda")
            if len(self._views) > 64:
                self._views.clear()
            self._views[key] = t
        return t

    def _on_stream(self):
        import contextlib
        if self.stream is None or (self._stream_is_current and self.torch.cuda.current_stream() == self.stream):
            return contextlib.nullcontext()
        return self.torch.cuda.stream(self.stream)

    def _allreduce_dev(self, _user, d_vals, n, elem_bytes):
        try:
            with self._on_stream():
                t = self._dev_tensor(d_vals, n, elem_bytes)      # uint sums as two's-complement ints: same bits
                if self._nccl:
                    self.dist.all_reduce(t)
                else:
                    h = t.cpu()
                    self.dist.all_reduce(h)
                    t.copy_(h)
                    self.torch.cuda.current_stream().synchronize()
            self.calls["allreduce_dev"] += 1
            self.calls["bytes"] += int(n) * int(elem_bytes)
            return 0
        except Exception as e:
            print("device allreduce hook failed:", repr(e))
            return 1

    def _allgather_dev(self, _user, d_send, d_recv, nbytes):
        try:
            with self._on_stream():
                src = self._dev_tensor(d_send, nbytes, 1)
                dst = self._dev_tensor(d_recv, int(nbytes) * self.world, 1)
                if self._nccl:
                    self.dist.all_gather_into_tensor(dst, src)
                else:
                    h = src.cpu()
                    outs = [self.torch.empty_like(h) for _ in range(self.world)]
                    self.dist.all_gather(outs, h)
                    dst.copy_(self.torch.cat(outs))
                    self.torch.cuda.current_stream().synchronize()
            self.calls["allgather_dev"] += 1
            self.calls["bytes"] += int(nbytes) * self.world
            return 0
        except Exception as e:
            print("device allgather hook failed:", repr(e))
            return 1

    def _allreduce(self, _user, vals, n):
        try:
            a = np.ctypeslib.as_array(vals, (int(n),)).view(np.int64)
            t = self.torch.from_numpy(a.copy()).to(self.device)
            self.dist.all_reduce(t)
            a[:] = t.cpu().numpy()
            self.calls["allreduce"] += 1
            self.calls["bytes"] += 8 * int(n)
            return 0
        except Exception as e:                                   # never let an exception cross the C boundary
            print("allreduce hook failed:", e)
            return 1

    def _allgatherv(self, _user, send, send_bytes, recv, stride, sizes):
        try:
            torch, dist = self.torch, self.dist
            n = int(send_bytes)
            sz = torch.tensor([n], dtype=torch.int64, device=self.device)
            allsz = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
            dist.all_gather(allsz, sz)
            mx = max(int(s[0]) for s in allsz)
            buf = torch.zeros(max(mx, 1), dtype=torch.uint8)
            if n:
                buf[:n] = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), (n,)).copy())
            buf = buf.to(self.device)
            outs = [torch.zeros_like(buf) for _ in range(self.world)]
            dist.all_gather(outs, buf)
            dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), (int(stride) * self.world,))
            for r in range(self.world):
                k = int(allsz[r][0])
                sizes[r] = k
                if k:
                    dst[r * int(stride): r * int(stride) + k] = outs[r][:k].cpu().numpy()
            self.calls["allgatherv"] += 1
            self.calls["bytes"] += mx * self.world
            return 0
        except Exception as e:
            print("allgatherv hook failed:", e)
            return 1


def _records(buf):
    """Yield (amplicon index, record bytes) for a FASTQ pool; records are 4 lines."""
    pos, n = 0, len(buf)
    while pos < n:
        e = pos
        for _ in range(4):
            e = buf.index(b"\n", e) + 1
        name_end = buf.index(b"#", pos)
        yield int(buf[pos + 1:name_end]), buf[pos:e]
        pos = e


def merge_fastq(pools):
    """k-way merge of per-shard FASTQ pools (each sorted by the amplicon's whole-job list index, which is the number
    after '@' in the record name) into the order the unsharded job writes."""
    its = [_records(p) for p in pools]
    return b"".join(rec for _, rec in heapq.merge(*its, key=lambda kv: kv[0]))
