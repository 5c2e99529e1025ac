"""Worker for tests/test_distributed_cpu.py: one shard of one oracle job, collectives over gloo."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Params(C.Structure):
    _fields_ = [("input_fasta", C.c_char_p), ("profile", C.c_char_p), ("output_prefix", C.c_char_p), ("dump_prefix", C.c_char_p),
                ("primers", C.c_long), ("gamma", C.c_double), ("coverage", C.c_double), ("isize", C.c_int), ("paired", C.c_int),
                ("threads", C.c_int), ("rng_mode", C.c_int), ("seed", C.c_uint64), ("fixed_time", C.c_longlong), ("verbose", C.c_int),
                ("shard_rank", C.c_int), ("shard_count", C.c_int), ("allreduce", C.c_void_p), ("allgatherv", C.c_void_p), ("coll_user", C.c_void_p),
                ("checksum_file", C.c_char_p), ("batch_pairs", C.c_uint64), ("checksum_batches", C.c_char_p)]


def main():
    fasta, profile, prefix, coverage, layout, seed = sys.argv[1:7]
    primers, gamma = (int(sys.argv[7]), float(sys.argv[8])) if len(sys.argv) > 8 else (None, None)
    import torch.distributed as dist
    from scssim_amd.dist import Collectives
    dist.init_process_group("gloo")
    coll = Collectives()
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libscs_oracle.so"))
    lib.scso_last_error.restype = C.c_char_p
    p = Params()
    lib.scso_default_params(C.byref(p))
    p.input_fasta, p.profile = fasta.encode(), profile.encode()
    p.output_prefix = ("%s.r%d" % (prefix, dist.get_rank())).encode()
    p.coverage, p.paired, p.seed, p.rng_mode, p.threads, p.verbose = float(coverage), int(layout == "PE"), int(seed), 1, 2, 0
    if primers is not None:
        p.primers, p.gamma = primers, gamma
    p.shard_rank, p.shard_count = dist.get_rank(), dist.get_world_size()
    p.allreduce = C.cast(coll.allreduce_cb, C.c_void_p)
    p.allgatherv = C.cast(coll.allgatherv_cb, C.c_void_p)
    rc = lib.scso_genreads(C.byref(p))
    if rc:
        print("oracle failed:", lib.scso_last_error().decode())
    print("rank %d collectives: %s" % (dist.get_rank(), coll.calls))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
