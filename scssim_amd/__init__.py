"""scssim_amd -- host-side mirror of the SCSsim `genreads` interface over the MI355X C ABI.

The product is `libscssim_hip.so` (hand-written gfx950 kernels + C ABI, include/scssim_hip.h) and the
`scssim` CLI.  This package is ctypes plumbing for tests, bench.py and torch.distributed drivers:
method names follow the reference's call sequence (src/scssim.cpp:46-67: loadData, train(profile),
createFrags, amplify, yieldReads).  There is no CPU fallback: importing works anywhere, creating a
`GenReads` needs the built library and a GPU, and fails loudly otherwise.
"""
from .api import (GenReads, ScsError, Profile, fasta_probe, fasta_write_index, lib_path, load_library, build,  # noqa: F401
                  merge_fastq_shards, merge_fastq_parts, part_paths, text_checksum, gpu_local_cpus, bgzf_probe, bgzf_blocks, comm_unique_id, simuvars_probe, devbuf_probe,
                  SCS_OK, SCS_EINVAL, SCS_EIO, SCS_EDEVICE, SCS_EOVERFLOW)
