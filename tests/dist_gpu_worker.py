"""Worker for the sharded GPU test: one shard of one genreads job on the (shared) GPU, collectives over gloo."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    fasta, profile, prefix, coverage, layout, seed, hooks = sys.argv[1:8]
    writers = int(sys.argv[8]) if len(sys.argv) > 8 else 0   # > 1: the shard is written as that many part files per mate
    extra = dict(primers=int(sys.argv[9]), gamma=float(sys.argv[10])) if len(sys.argv) > 10 else {}
    import torch.distributed as dist
    import scssim_amd
    from scssim_amd.dist import Collectives
    dist.init_process_group("gloo")
    import torch
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    coll = Collectives(device="cpu", stream=stream)
    g = scssim_amd.GenReads(profile=profile, input_fasta=fasta, coverage=float(coverage), layout=layout, seed=int(seed), device=0,
                            stream=stream.cuda_stream, shard_rank=dist.get_rank(), shard_count=dist.get_world_size(), **extra)
    g.set_collectives(coll, device_hooks=(hooks == "device"))
    g.create_frags(); g.amplify(); g.allocate_reads(0)
    g.yield_reads_files(prefix, writers)            # this rank's shard + index: <prefix>.r<rank>_1.fq / _2.fq / .idx (or its parts)
    st = g.stats()
    print("rank %d: %d fragments, %d fulls, %d pairs, collectives %s" % (dist.get_rank(), st["fragments"], st["full_amplicons"], st["pairs_written"], coll.calls))
    print("STAGED %d %d %d" % (dist.get_rank(), st["staged_bases"], st["genome_bases"]))
    print("STOCK %d %d %d %d %d" % (dist.get_rank(), st["stock_checks"], st["stock_exhausted_passes"], st["stock_rounds"], int((g.download_primer_stock() == 0).sum())))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
