import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # SCSSIM_HIP_LIB: another build of the same library (A/B measurements of a kernel change on one box: tools/ab_kernels.sh)
    return os.environ.get("SCSSIM_HIP_LIB") or os.path.join(HERE, "libscssim_hip.so")


def build(verbose=False):
    """Compile the HIP library and the CLI for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), "-j4"] + ([] if verbose else ["-s"]))


SCS_OK, SCS_EINVAL, SCS_EIO, SCS_EDEVICE, SCS_EOVERFLOW = 0, 1, 2, 3, 4       # include/scssim_hip.h


class ScsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("scssim_hip error %d: %s" % (code, msg))
        self.code = code


class _Config(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("seed", C.c_uint64), ("primers", C.c_long),
                ("gamma", C.c_double), ("coverage", C.c_double), ("isize", C.c_int), ("paired", C.c_int),
                ("ber", C.c_double), ("amplicon_min_len", C.c_int), ("amplicon_max_len", C.c_int),
                ("frag_size", C.c_int), ("frag_min", C.c_int), ("frag_max", C.c_int),
                ("shard_rank", C.c_int), ("shard_count", C.c_int), ("verbose", C.c_int)]


class _Stats(C.Structure):
    _fields_ = [("records", C.c_uint64), ("genome_bases", C.c_uint64), ("fragments", C.c_uint64),
                ("semi_amplicons", C.c_uint64), ("full_amplicons", C.c_uint64), ("primers_left", C.c_uint64),
                ("reads_requested", C.c_uint64), ("pairs_written", C.c_uint64), ("reads_written", C.c_uint64),
                ("fastq_bytes", C.c_uint64 * 2), ("algorithmic_bytes", C.c_uint64), ("t_stage", C.c_double * 8), ("sink_bytes", C.c_uint64 * 2), ("staged_bases", C.c_uint64),
                ("stock_checks", C.c_uint64), ("stock_exhausted_passes", C.c_uint64), ("stock_rounds", C.c_uint64)]


_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
_lib = None


def load_library():
    """dlopen the in-tree library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError("%s is missing: run scssim_amd.build() / make -C scssim_amd/csrc (no CPU fallback exists)" % p)
    L = C.CDLL(p)
    L.scs_last_error.restype = C.c_char_p
    L.scs_last_error.argtypes = [C.c_void_p]
    L.scs_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
    L.scs_destroy.argtypes = [C.c_void_p]
    L.scs_set_seed.argtypes = [C.c_void_p, C.c_uint64]
    L.scs_load_profile.argtypes = [C.c_void_p, C.c_char_p]
    L.scs_read_length.argtypes = [C.c_void_p]
    L.scs_load_genome_fasta.argtypes = [C.c_void_p, C.c_char_p]
    L.scs_upload_genome.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]
    L.scs_upload_genome_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_void_p]
    for f in ("scs_create_frags", "scs_amplify"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.scs_allocate_reads.argtypes = [C.c_void_p, C.c_uint64]
    L.scs_yield_reads.argtypes = [C.c_void_p, _SINK, C.c_void_p]
    L.scs_run_genreads.argtypes = [C.c_void_p, _SINK, C.c_void_p]
    L.scs_yield_reads_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.scs_yield_reads_files.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.scs_merge_fastq_shards.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.scs_comm_unique_id.argtypes = [C.c_void_p]
    L.scs_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.scs_get_stats.argtypes = [C.c_void_p, C.POINTER(_Stats)]
    L.scs_set_collectives.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scs_set_collectives_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.scs_kernel_time.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.scs_set_kernel_timing.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    L.scs_predict_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.scs_philox_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.scs_detlog_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.scs_download_amplicons.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
    L.scs_download_read_numbers.argtypes = [C.c_void_p, C.c_void_p]
    L.scs_download_primer_stock.argtypes = [C.c_void_p, C.c_void_p]
    L.scs_profile_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.scs_profile_table.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_size_t)]
    L.scs_profile_scalars.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.scs_profile_close.argtypes = [C.c_void_p]
    L.scs_fasta_probe.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.scs_fasta_write_index.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.scs_simuvars.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
    L.scs_simuvars_probe.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    _lib = L
    return L


def merge_fastq_shards(prefix, nranks, paired=True, keep_shards=False):
    """Host-only: rebuild the single-job FASTQ files from the per-rank shards + indexes that a sharded job wrote with
    GenReads.yield_reads_files (byte-range copies in list order; no record is parsed)."""
    L = load_library()
    err = C.create_string_buffer(512)
    rc = L.scs_merge_fastq_shards(os.fsencode(prefix), int(nranks), int(paired), int(keep_shards), err, 512)
    if rc:
        raise ScsError(rc, err.value.decode())


def merge_fastq_parts(prefix, paired=True, keep_parts=False):
    """Host-only: <prefix>.p*_1.fq ... (written by yield_reads_files(prefix, writers=K)) -> <prefix>_1.fq ... by byte-range
    copies; the parts and <prefix>.parts are removed unless keep_parts."""
    L = load_library()
    L.scs_merge_fastq_parts.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    err = C.create_string_buffer(512)
    rc = L.scs_merge_fastq_parts(os.fsencode(prefix), int(paired), int(keep_parts), err, 512)
    if rc:
        raise ScsError(rc, err.value.decode())


def part_paths(base, parts, paired=True, suffix=".fq"):
    """The files yield_reads_files(base, writers=parts) writes, per mate, in record order."""
    mates = ("_1", "_2") if paired else ("",)
    if parts <= 1:
        return [[base + m + suffix] for m in mates]
    return [[base + ".p%02d" % k + m + suffix for k in range(parts)] for m in mates]


def gpu_local_cpus(device=0):
    """CPUs of the NUMA node the GPU hangs on, within this process's affinity mask ([]: unknown or no choice)."""
    L = load_library()
    buf = (C.c_int * 4096)()
    n = L.scs_gpu_local_cpus(int(device), buf, 4096)
    return [buf[i] for i in range(min(n, 4096))]


def text_checksum(data):
    """The library's batch checksum (scs_set_batch_checksums) of a bytes-like object, in numpy: the text as little-endian 64-bit
    words w_i (the last zero-padded), sum_i fmix64(w_i + (i + 1) * 0x9E3779B97F4A7C15) mod 2^64."""
    import numpy as np
    b = np.frombuffer(data, np.uint8)
    pad = (-len(b)) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    w = b.view("<u8").astype(np.uint64)
    with np.errstate(over="ignore"):
        x = w + (np.arange(1, len(w) + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        x ^= x >> np.uint64(33); x *= np.uint64(0xFF51AFD7ED558CCD); x ^= x >> np.uint64(33); x *= np.uint64(0xC4CEB9FE1A85EC53); x ^= x >> np.uint64(33)
        return int(np.add.reduce(x, dtype=np.uint64)) if len(x) else 0


def bgzf_probe(data, lds_out_cap=0):
    """Host-only: the BGZF blocks the device kernels would make of `data` (their arithmetic on the CPU; no end-of-file block)."""
    L = load_library()
    L.scs_bgzf_probe.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    n = C.c_uint64()
    data = bytes(data)
    rc = L.scs_bgzf_probe(data, len(data), lds_out_cap, None, 0, C.byref(n))
    if rc:
        raise ScsError(rc, "scs_bgzf_probe")
    out = C.create_string_buffer(max(1, n.value))
    rc = L.scs_bgzf_probe(data, len(data), lds_out_cap, out, n.value, C.byref(n))
    if rc:
        raise ScsError(rc, "scs_bgzf_probe")
    return out.raw[:n.value]


def bgzf_blocks(data):
    """Split BGZF bytes into (block bytes, ISIZE) after checking every block's frame: gzip magic, the BC subfield, BSIZE."""
    out, o = [], 0
    while o < len(data):
        h = data[o:o + 18]
        assert len(h) == 18 and h[:4] == b"\x1f\x8b\x08\x04" and h[10:16] == b"\x06\x00BC\x02\x00", "not a BGZF block at %d" % o
        size = int.from_bytes(h[16:18], "little") + 1
        assert o + size <= len(data) and size <= 65536
        out.append((data[o:o + size], int.from_bytes(data[o + size - 4:o + size], "little")))
        o += size
    return out


COMM_ID_BYTES = 128


def comm_unique_id():
    """RCCL communicator id (ncclGetUniqueId) as bytes: rank 0 makes it, every rank passes it to GenReads.comm_init."""
    L = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = L.scs_comm_unique_id(buf)
    if rc:
        raise ScsError(rc, L.scs_last_error(None).decode())
    return buf.raw


def simuvars_probe(ref_fasta, snp_file=None, var_file=None):
    """Host-only: (records, total haplotype bases, FNV-1a of the FASTA text) that GenReads.simuvars would produce."""
    L = load_library()
    n, tot, h = C.c_int(), C.c_uint64(), C.c_uint64()
    err = C.create_string_buffer(512)
    enc = lambda p: os.fsencode(p) if p else None
    rc = L.scs_simuvars_probe(enc(ref_fasta), enc(snp_file), enc(var_file), C.byref(n), C.byref(tot), C.byref(h), err, 512)
    if rc:
        raise ScsError(rc, err.value.decode())
    return n.value, tot.value, h.value


def fasta_write_index(path):
    """Host-only: leave <path>.fai beside the FASTA if there is none, as loading the genome does (fastahack index)."""
    L = load_library()
    err = C.create_string_buffer(512)
    rc = L.scs_fasta_write_index(os.fsencode(path), err, 512)
    if rc:
        raise ScsError(rc, err.value.decode())


def devbuf_probe(first_bytes, second_bytes, device=0):
    """Test seam (needs a GPU): capacities of a library device buffer after reserve(first) and reserve(second), and
    whether the second reserve kept the buffer's address."""
    L = load_library()
    L.scs_devbuf_probe.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    caps, same = (C.c_uint64 * 2)(), C.c_int()
    rc = L.scs_devbuf_probe(device, first_bytes, second_bytes, caps, C.byref(same))
    if rc:
        raise ScsError(rc, (L.scs_last_error(None) or b"").decode())
    return caps[0], caps[1], bool(same.value)


def fasta_probe(path):
    """Host-only: (names, total bases, FNV-1a checksum of the upper-cased sequence) as the library stages the file."""
    L = load_library()
    n, tot, h = C.c_int(), C.c_uint64(), C.c_uint64()
    names, err = C.create_string_buffer(1 << 16), C.create_string_buffer(512)
    rc = L.scs_fasta_probe(os.fsencode(path), C.byref(n), C.byref(tot), C.byref(h), names, 1 << 16, err, 512)
    if rc:
        raise ScsError(rc, err.value.decode())
    return names.value.decode().split("\n")[:n.value], tot.value, h.value


class Profile:
    """Host-only view of a .profile model: the exact uint32 thresholds and the double CDFs they come from
    (mirrors Profile::train(file), reference lib/profile/Profile.cpp:1432-1436).  Needs no GPU."""

    TABLES = {"subs1": 0, "subs2": 1, "qual": 2, "ins": 3, "del": 4, "isize": 5, "qual_alias": 6}

    def __init__(self, path, paired=True, isize=260):
        import numpy as np
        self._np = np
        L = load_library()
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.scs_profile_open(os.fsencode(path), int(paired), int(isize), C.byref(self._h), err, 512)
        if rc:
            raise ScsError(rc, err.value.decode())
        sc = (C.c_double * 10)()
        L.scs_profile_scalars(self._h, sc)
        (self.read_length, self.bins, self.t_insert, self.t_delete, self.isize_min, self.have_cdf2) = [int(v) for v in sc[:6]]
        self.insert_rate, self.del_rate = sc[6], sc[7]
        self.t_indel, self.qual_k = int(sc[8]), int(sc[9])

    def table(self, name):
        np = self._np
        thr = C.POINTER(C.c_uint32)()
        cdf = C.POINTER(C.c_double)()
        n = C.c_size_t()
        rc = load_library().scs_profile_table(self._h, self.TABLES[name], C.byref(thr), C.byref(cdf), C.byref(n))
        if rc:
            raise ScsError(rc, "bad table")
        if n.value == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.float64)
        if not cdf:                                   # tables without a double twin (compact quality rows)
            return np.ctypeslib.as_array(thr, (n.value,)).copy(), None
        return (np.ctypeslib.as_array(thr, (n.value,)).copy(), np.ctypeslib.as_array(cdf, (n.value,)).copy())

    def close(self):
        if self._h:
            load_library().scs_profile_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GenReads:
    """One `scssim genreads` job on one MI355X.  Mirrors the reference's driver (src/scssim.cpp:46-67)."""

    def __init__(self, profile=None, input_fasta=None, primers=100000, gamma=1e-9, coverage=5.0, isize=260,
                 layout="PE", seed=1, device=0, stream=None, shard_rank=0, shard_count=1, verbose=False):
        import numpy as np
        self._np = np
        self._L = load_library()
        cfg = _Config()
        self._L.scs_default_config.argtypes = [C.POINTER(_Config)]
        self._L.scs_default_config(C.byref(cfg))
        cfg.device, cfg.stream, cfg.seed = device, stream, seed
        cfg.primers, cfg.gamma, cfg.coverage, cfg.isize = primers, gamma, coverage, isize
        if layout not in ("PE", "SE"):
            raise ValueError("Error: sequence layout incorrectly specified!")
        cfg.paired = 1 if layout == "PE" else 0
        cfg.shard_rank, cfg.shard_count, cfg.verbose = shard_rank, shard_count, int(verbose)
        self.paired = bool(cfg.paired)
        self._ctx = C.c_void_p()
        rc = self._L.scs_create(C.byref(cfg), C.byref(self._ctx))
        if rc:
            raise ScsError(rc, self._L.scs_last_error(None).decode())
        if input_fasta is not None:
            self.load_genome(input_fasta)
        if profile is not None:
            self.load_profile(profile)

    # ---- plumbing
    def _ck(self, rc):
        if rc:
            raise ScsError(rc, self._L.scs_last_error(self._ctx).decode())

    def close(self):
        if self._ctx:
            self._L.scs_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference call sequence
    def load_genome(self, path):            # Genome::loadData
        self._ck(self._L.scs_load_genome_fasta(self._ctx, os.fsencode(path)))

    def upload_genome(self, names, seqs):
        n = len(names)
        bn = [s.encode() if isinstance(s, str) else s for s in names]
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        self._ck(self._L.scs_upload_genome(self._ctx, n, (C.c_char_p * n)(*bn), (C.c_char_p * n)(*bs),
                                           (C.c_uint64 * n)(*[len(s) for s in bs])))

    def upload_genome_device(self, names, lens, d_bases):
        """Genome already in HBM: d_bases = device pointer to the records' ASCII bases, concatenated (sum(lens) bytes)."""
        n = len(names)
        bn = [s.encode() if isinstance(s, str) else s for s in names]
        self._ck(self._L.scs_upload_genome_device(self._ctx, n, (C.c_char_p * n)(*bn), (C.c_uint64 * n)(*[int(x) for x in lens]), C.c_void_p(int(d_bases))))

    def simuvars(self, ref_fasta, snp_file=None, var_file=None, out_fasta=None):
        """`scssim simuvars` on the data plane: the two haplotypes of every chromosome are built in HBM and stay resident
        as the genreads input (no intermediate FASTA); out_fasta additionally writes the reference's simuvars file."""
        enc = lambda p: os.fsencode(p) if p else None
        self._ck(self._L.scs_simuvars(self._ctx, enc(ref_fasta), enc(snp_file), enc(var_file), enc(out_fasta)))

    def load_profile(self, path):           # Profile::train(file)
        self._ck(self._L.scs_load_profile(self._ctx, os.fsencode(path)))

    @property
    def read_length(self):
        return self._L.scs_read_length(self._ctx)

    def set_collectives(self, coll, device_hooks=False):
        """Sharded single job (shard_count > 1): `coll` = scssim_amd.dist.Collectives (torch.distributed hooks).
        device_hooks=True: collectives act on the library's HBM buffers directly (create the GenReads on torch's
        current stream: stream=torch.cuda.current_stream().cuda_stream)."""
        self._coll = coll                      # keep the ctypes callbacks alive
        self._ck(self._L.scs_set_collectives(self._ctx, C.cast(coll.allreduce_cb, C.c_void_p), C.cast(coll.allgatherv_cb, C.c_void_p), None))
        if device_hooks:
            self._ck(self._L.scs_set_collectives_device(self._ctx, C.cast(coll.allreduce_dev_cb, C.c_void_p), C.cast(coll.allgather_dev_cb, C.c_void_p), None))

    def set_seed(self, seed):
        self._ck(self._L.scs_set_seed(self._ctx, seed))

    def create_frags(self):                 # Malbac::createFrags
        self._ck(self._L.scs_create_frags(self._ctx))

    def amplify(self):                      # Malbac::amplify
        self._ck(self._L.scs_amplify(self._ctx))

    def allocate_reads(self, reads=0):      # Malbac::setReadCounts
        self._ck(self._L.scs_allocate_reads(self._ctx, reads))

    def yield_reads(self, collect=True):    # Malbac::yieldReads -> (fastq1, fastq2) bytes
        parts1, parts2 = [], []

        def sink(_u, p1, n1, p2, n2):
            if collect:
                parts1.append(C.string_at(p1, n1) if n1 else b"")
                parts2.append(C.string_at(p2, n2) if n2 else b"")
            return 0
        cb = _SINK(sink) if collect else _SINK()   # NULL sink: generate on the device and count only
        self._ck(self._L.scs_yield_reads(self._ctx, cb, None))
        return b"".join(parts1), b"".join(parts2)

    def yield_reads_sink(self, sink=None):
        """Malbac::yieldReads with a caller-supplied sink(user, p1, n1, p2, n2) -> int (host pointers valid during the call),
        or None: the FASTQ text is generated batch by batch into HBM buffers and counted only."""
        cb = _SINK(sink) if sink is not None else _SINK()
        self._ck(self._L.scs_yield_reads(self._ctx, cb, None))

    def yield_reads_files(self, prefix, writers=0, generations=1, bgzf=False, in_place=False):
        """Malbac::yieldReads + SeqWriter: <prefix>_1.fq/_2.fq (.fq), or this shard's <prefix>.r<rank>_*.fq + .idx.
        writers = K > 1: K part files per mate (<base>.p00_1.fq ...: contiguous record ranges, one writer thread each; their
        concatenation is the single file) + <base>.parts.  generations = G > 1: K x G parts made generation by generation (part p
        is final once part p + K exists).  bgzf: <...>.fq.gz, BGZF blocks made on the GPU.  in_place: files that exist are overwritten
        where they lie and cut to length at the end, not truncated first (SCS_SINK_IN_PLACE)."""
        self._L.scs_yield_reads_files_ex.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        self._ck(self._L.scs_yield_reads_files_ex(self._ctx, os.fsencode(prefix), int(writers), int(generations), (1 if bgzf else 0) | (2 if in_place else 0)))

    def comm_init(self, comm_id, rank, nranks):
        """RCCL inside the library: every rank of a sharded job calls this with rank 0's comm_unique_id()."""
        self._id = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        self._ck(self._L.scs_comm_init(self._ctx, self._id, int(rank), int(nranks)))

    def comm_count(self):
        """Ranks of the ctx's RCCL communicator as RCCL reports them (ncclCommCount); 0 without one."""
        self._L.scs_comm_count.argtypes = [C.c_void_p]
        return int(self._L.scs_comm_count(self._ctx))

    def yield_reads_device(self, d_fq1, cap1, d_fq2, cap2):
        """FASTQ pool stays in HBM: d_fq1/d_fq2 are device pointers (e.g. torch uint8 tensors' data_ptr())."""
        n1, n2, pairs = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ck(self._L.scs_yield_reads_device(self._ctx, d_fq1, cap1, d_fq2, cap2, C.byref(n1), C.byref(n2), C.byref(pairs)))
        return n1.value, n2.value, pairs.value

    def run_genreads(self, collect=True):
        """scs_run_genreads: createFrags + amplify + allocate + yield in ONE call of the C ABI (main()'s genreads branch,
        src/scssim.cpp:59-65)."""
        out1, out2 = bytearray(), bytearray()

        def on_batch(_u, p1, n1, p2, n2):
            if collect:
                if n1:
                    out1.extend(C.string_at(p1, n1))
                if n2:
                    out2.extend(C.string_at(p2, n2))
            return 0
        cb = _SINK(on_batch)
        self._ck(self._L.scs_run_genreads(self._ctx, cb, None))
        return bytes(out1), bytes(out2)

    def run(self, collect=True):
        self.create_frags()
        self.amplify()
        self.allocate_reads(0)
        return self.yield_reads(collect)

    def run_to_files(self, prefix):
        fq1, fq2 = self.run()
        if self.paired:
            open(prefix + "_1.fq", "wb").write(fq1)
            open(prefix + "_2.fq", "wb").write(fq2)
        else:
            open(prefix + ".fq", "wb").write(fq1)

    def set_batch_checksums(self, on=True):
        """Every batch of the following yield_reads* calls gets a 64-bit checksum per mate, computed on the device."""
        self._L.scs_set_batch_checksums.argtypes = [C.c_void_p, C.c_int]
        self._ck(self._L.scs_set_batch_checksums(self._ctx, int(on)))

    def batch_checksums(self):
        """[(mate 1, mate 2), ...] of the last yield call's batches, in record order (see text_checksum)."""
        self._L.scs_batch_checksums.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        n = C.c_size_t()
        self._ck(self._L.scs_batch_checksums(self._ctx, None, 0, C.byref(n)))
        out = (C.c_uint64 * (2 * n.value))()
        self._ck(self._L.scs_batch_checksums(self._ctx, out, 2 * n.value, C.byref(n)))
        return [(out[2 * i], out[2 * i + 1]) for i in range(n.value)]

    # ---- introspection
    def stats(self):
        st = _Stats()
        self._ck(self._L.scs_get_stats(self._ctx, C.byref(st)))
        d = {k: getattr(st, k) for k, _ in _Stats._fields_ if k not in ("fastq_bytes", "t_stage", "sink_bytes")}
        d["fastq_bytes"] = list(st.fastq_bytes)
        d["sink_bytes"] = list(st.sink_bytes)
        d["t_stage"] = list(st.t_stage)
        return d

    def kernel_times(self):
        out = {}
        for i in range(len(self.KERNELS)):
            name, n, ms, units = C.c_char_p(), C.c_uint64(), C.c_double(), C.c_uint64()
            self._ck(self._L.scs_kernel_time(self._ctx, i, C.byref(name), C.byref(n), C.byref(ms), C.byref(units)))
            out[name.value.decode()] = dict(launches=n.value, ms=ms.value, units=units.value)
        return out

    KERNELS = ("k_errs<semi->full>", "k_errs<frag->semi>", "k_reads", "k_attach<semi>", "k_indels", "k_attach<frag>")

    def set_kernel_timing(self, names=None, every=1):
        """Keep HIP event pairs only around the named kernels (None = all six), on every `every`-th amplify / yield call.
        Every event record is a packet on the stream (about 6 us each on the latency-bound 1 Mb job)."""
        mask = (1 << len(self.KERNELS)) - 1 if names is None else sum(1 << self.KERNELS.index(n) for n in names)
        self._ck(self._L.scs_set_kernel_timing(self._ctx, mask, every))

    def download_amplicons(self, kind):
        np = self._np
        st = self.stats()
        n = st["semi_amplicons"] if kind == 0 else st["full_amplicons"]
        a = {k: np.zeros(n, np.uint32) for k in ("parent", "spos", "len", "gc", "primers")}
        a["uid"] = np.zeros(n, np.uint64)
        a["errs"] = np.zeros((n, 4), np.uint32)
        a["nerr"] = np.zeros(n, np.uint32)
        ptr = lambda x: x.ctypes.data_as(C.c_void_p)
        self._ck(self._L.scs_download_amplicons(self._ctx, kind, ptr(a["parent"]), ptr(a["spos"]), ptr(a["len"]), ptr(a["gc"]),
                                                ptr(a["primers"]), ptr(a["uid"]), ptr(a["errs"]), ptr(a["nerr"])))
        return a

    def download_primer_stock(self):
        """Copies of every primer type left after amplify (PrimerIndex.count, lib/malbac/Malbac.h:18-24), index = 2-bit-packed 8-mer."""
        import numpy as np
        st = np.zeros(65536, np.int64)
        self._ck(self._L.scs_download_primer_stock(self._ctx, st.ctypes.data_as(C.c_void_p)))
        return st

    def download_read_numbers(self):
        np = self._np
        rn = np.zeros(self.stats()["full_amplicons"], np.uint32)
        self._ck(self._L.scs_download_read_numbers(self._ctx, rn.ctypes.data_as(C.c_void_p)))
        return rn

    # ---- kernel-level entry points
    def predict_batch(self, windows, uids, attempts, is_read1):
        """Profile::predict for a batch of windows (n x L uint8 codes).  Returns (list of bases, list of quals)."""
        np = self._np
        w = np.ascontiguousarray(windows, np.uint8)
        n, L = w.shape
        assert L == self.read_length
        stride = ((L + 64 + 63) // 64) * 64
        u = np.ascontiguousarray(uids, np.uint64)
        a = np.ascontiguousarray(attempts, np.uint32)
        r = np.ascontiguousarray(is_read1, np.uint8)
        ob = np.zeros((n, stride), np.uint8)
        oq = np.zeros((n, stride), np.uint8)
        ol = np.zeros(n, np.int32)
        ptr = lambda x: x.ctypes.data_as(C.c_void_p)
        self._ck(self._L.scs_predict_batch(self._ctx, ptr(w), n, ptr(u), ptr(a), ptr(r), ptr(ob), ptr(oq), ptr(ol), stride))
        return [bytes(ob[i, :ol[i]]) for i in range(n)], [bytes(oq[i, :ol[i]]) for i in range(n)]

    def philox(self, ctr, key):
        np = self._np
        c = np.ascontiguousarray(ctr, np.uint32).reshape(-1, 4)
        k = np.ascontiguousarray(key, np.uint32)
        out = np.zeros_like(c)
        self._ck(self._L.scs_philox_batch(self._ctx, c.ctypes.data_as(C.c_void_p), c.shape[0], k.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def det_log(self, x):
        np = self._np
        x = np.ascontiguousarray(x, np.float64)
        out = np.zeros_like(x)
        self._ck(self._L.scs_detlog_batch(self._ctx, x.ctypes.data_as(C.c_void_p), x.size, out.ctypes.data_as(C.c_void_p)))
        return out
