#!/usr/bin/env python3
"""bench.py -- the genreads hot path on N MI355X (one process per GPU), on the metric's configuration.

Metric (BASELINE.json): paired-end read pairs/s at PE150, 30x, quoted on the 3 Gb whole genome (configs[3]); plus the
HBM roofline fraction of the dominant kernel and the reference's CPU path timed on this box's host cores.

Workload (fits one MI355X): synthetic diploid whole genome, 24 records with the hg19 chromosome lengths
(3 095 677 412 bases per haplotype, two identical haplotype records per chromosome named <chr>_<hap>_<len> as
`simuvars` writes them), i.i.d. bases P(A,C,G,T) = (0.3, 0.2, 0.2, 0.3) drawn on the GPU (torch, seeded) and handed to
the library in HBM (scs_upload_genome_device) -- a 6 GB FASTA written and parsed per run would only time the disk.
PE150 = the shipped HiSeq2500 profile with its bin axis resampled 125 -> 150 (tools/make_profile.py; the reference
takes the read length from the profile only and ships no 150 bp model); 30x, -p 100000 -r 1e-9 -s 260.

One step = one complete job with a fresh seed: fragment split, 1+5 MALBAC cycles, GC-biased read allocation, fragment
sampling, indel / substitution / quality injection, FASTQ text -- AND THE TEXT IN FILES: every timed step writes its 197 GB of
plain FASTQ through the library's file sink (scs_yield_reads_files_ex: D2H on a copy stream, `--writers` threads each appending
to its own part file per mate, `--generations` generations) into fresh files on tmpfs.  `value` = pairs / that time: SURVEY 8(d)'s
"records in _1.fq / wall".  The memory cgroup cannot hold two steps' text, so background threads unlink every part once it is
final and a step's sink starts when <= 40 GB of older text are left -- inside the timed region.
Beside it, at N = 1 and outside the timed region:
  generation_hbm : 5 steps with a NULL sink (text generated batch by batch into HBM buffers and counted): what the kernels do
                   when nothing crosses PCIe; the dominant kernel's roofline is taken here (8 M pairs per launch);
  d2h_only       : one step through a sink that only counts: what PCIe allows;
  bgzf           : 2 steps with the text made into BGZF blocks ON THE GPU before it crosses PCIe (.fq.gz parts; an extension);
  sweep          : configs[1] (1 Mb) and configs[2]'s size (63 Mb) with the same model and options;
  cli_wall       : the `scssim genreads` binary end to end (FASTA parse, profile, job, files) at chr20 size (configs[2]);
  cpu_baseline   : the reference itself (oracle/_ref/scssim_ref, compiled from /root/reference by oracle/Makefile) with
                   -t <cores> and -t 1 on bounded samples of the same genome; the oracle port beside it.
--hbm-only times the steps with a NULL sink instead (round 2's headline; what tools/profile_bench.sh profiles).
N > 1 (driver: torch.distributed.run, one rank per GPU): the SAME job sharded N ways by fragment lineage (strong
scaling); per-pass primer-stock all-reduce and the allocation partials over RCCL; every rank writes its own FASTQ shard
(its own part files, its own PCIe link and writer threads).
"""
import argparse
import glob
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_MEASURED = 7.72e11   # wave-instructions/s: the best integer issue rate tools/probes/valu_peak.hip reached (3.18 cycles, profiles/r02_valu_peak.log)
VALU_PEAK_SPEC = 256 * 4 * 2.4e9 / 2   # the guide's 2-cycle wave64 issue per SIMD (v_fma_f32): 1.23e12/s
HG19 = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431, 135534747,
        135006516, 133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983, 63025520,
        48129895, 51304566, 155270560, 59373566]


def record_lengths(genome_mb):
    """24 hg19-like records; --genome-mb scales them (rehearsals and tests only: the default is the real lengths)."""
    if not genome_mb:
        return list(HG19)
    f = genome_mb * 1e6 / sum(HG19)
    return [max(50000, int(x * f)) for x in HG19]


def make_profile(td):
    src = os.path.join(td, "hiseq2500.profile")
    with open(src, "wb") as f:
        f.write(gzip.open(os.path.join(ROOT, "tests", "golden", "models", "Illumina_HiSeq2500.profile.gz")).read())
    prof = os.path.join(td, "pe150.profile")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profile.py"), src, prof, "--read-length", "150"])
    return prof


def synth_genome(torch, dev, lens, seed):
    """Diploid genome in HBM as ASCII: per chromosome two identical haplotype records (what `simuvars` emits without a
    variation file).  Returns (names, lens per record, uint8 tensor)."""
    bases = torch.empty(2 * sum(lens), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    names, rl, off = [], [], 0
    for i, n in enumerate(lens):
        u = torch.rand(n, device=dev, generator=gen)
        rec = (u >= 0.3).to(torch.uint8) * 2 + (u >= 0.5).to(torch.uint8) * 4 + (u >= 0.7).to(torch.uint8) * 13 + 65    # A C G T = 65 67 71 84
        del u
        for hap in (1, 2):
            bases[off:off + n] = rec
            off += n
            names.append("%d_%d_%d" % (i + 1, hap, n))
            rl.append(n)
        del rec
    return names, rl, bases


def genome_fingerprint(torch, bases):
    """Position-weighted 64-bit sum of the genome's bytes in HBM (as int64 words w_i: sum w_i (2 i + 1) mod 2^64): tells whether a box's
    torch generator made the genome the golden checksums of tests/golden/whole_genome_config3.json belong to."""
    n = bases.numel() // 8 * 8
    acc, step = 0, 1 << 27
    w = bases[:n].view(torch.int64)
    for o in range(0, w.numel(), step):
        part = w[o:o + step]
        idx = torch.arange(o, o + part.numel(), device=bases.device, dtype=torch.int64)
        acc = (acc + int((part * (2 * idx + 1)).sum().item())) & 0xFFFFFFFFFFFFFFFF
    for k in range(n, bases.numel()):
        acc = (acc + int(bases[k].item()) * (k + 12345)) & 0xFFFFFFFFFFFFFFFF
    return acc


def make_config4_inputs(torch, dev, lens, ref_path, var_path):
    """BASELINE configs[4]'s inputs: the bench genome's haploid records (the same generator stream as synth_genome, seed 3000) as a plain
    reference FASTA (60 columns, chr1..chr24) and a CNV-heavy variation file (sorted, non-overlapping intervals, copy numbers 0..8 in every
    major-copy split; numpy generator 5).  Returns (bases the two haplotypes of all chromosomes must have, fingerprint of the reference)."""
    import numpy as np
    rng = np.random.default_rng(5)
    gen = torch.Generator(device=dev); gen.manual_seed(3000)
    expect, fp = 0, 0
    with open(ref_path, "wb") as f, open(var_path, "w") as fv:
        for i, n in enumerate(lens):
            u = torch.rand(n, device=dev, generator=gen)
            rec_d = (u >= 0.3).to(torch.uint8) * 2 + (u >= 0.5).to(torch.uint8) * 4 + (u >= 0.7).to(torch.uint8) * 13 + 65
            del u
            fp = (fp * 1000003 + genome_fingerprint(torch, rec_d)) & 0xFFFFFFFFFFFFFFFF
            rec = rec_d.cpu().numpy()
            del rec_d
            f.write(b">chr%d\n" % (i + 1))
            full = (n // 60) * 60
            f.write(np.concatenate([rec[:full].reshape(-1, 60), np.full((full // 60, 1), 10, np.uint8)], axis=1).tobytes())
            if n > full:
                f.write(rec[full:].tobytes() + b"\n")
            pos, hap = 100000, [n, n]
            while pos + 3000000 < n:                                 # sorted, non-overlapping CNV intervals, every major-copy split
                ln = int(rng.integers(50000, 1500000)); cn = int(rng.integers(0, 9)); mcn = int(rng.integers((cn + 1) // 2, cn + 1))
                fv.write("c\tchr%d\t%d\t%d\t%d\t%d\n" % (i + 1, pos, pos + ln, cn, mcn))
                hap[0] += (mcn - 1) * (ln + 1); hap[1] += (cn - mcn - 1) * (ln + 1)
                pos += ln + int(rng.integers(500000, 4000000))
            expect += hap[0] + hap[1]
    return expect, fp


def write_simu_fasta(path, names, seqs):
    """simuvars-style FASTA (100 columns) from numpy uint8 arrays."""
    import numpy as np
    with open(path, "wb") as f:
        for name, s in zip(names, seqs):
            f.write(b">" + name.encode() + b"\n")
            n = len(s)
            full = (n // 100) * 100
            if full:
                body = np.concatenate([s[:full].reshape(-1, 100), np.full((full // 100, 1), 10, np.uint8)], axis=1)
                f.write(body.tobytes())
            if n > full:
                f.write(s[full:].tobytes() + b"\n")


def host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                # a cgroup CPU quota bounds what a thread pool really gets
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(td, prof, names, seqs, sample_desc, coverage):
    """The reference's own CPU path on this box's host cores, on a bounded sample of the bench genome (same profile,
    coverage and options).  oracle/_ref/scssim_ref is the reference compiled from its sources (oracle/Makefile, target
    ref); when it is absent the oracle port stands in.  Checker code timed as a baseline -- never the thing shipped."""
    fa = os.path.join(td, "cpu_sample.fa")
    write_simu_fasta(fa, names, seqs)
    cores = host_cores()
    out = {}
    ref = os.path.join(ROOT, "oracle", "_ref", "scssim_ref")
    oracle = os.path.join(ROOT, "oracle", "_build", "scs_oracle")
    if os.path.exists(ref):
        t0 = time.perf_counter()
        r = subprocess.run([ref, "genreads", "-i", fa, "-m", prof, "-c", "%g" % coverage, "-t", str(cores), "-o", fa + ".ref"], capture_output=True, text=True)
        secs = time.perf_counter() - t0
        if r.returncode == 0 and os.path.exists(fa + ".ref_1.fq"):
            pairs = sum(1 for _ in open(fa + ".ref_1.fq", "rb")) // 4
            aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            out = dict(value=pairs / secs, unit="pairs/s", cores=cores, kind="reference",
                       sample="%s: %d pairs in %.1f s wall of `scssim genreads -t %d` (whole process: load + amplify + reads + files)" % (sample_desc, pairs, secs, cores),
                       affinity="%d CPUs in the affinity mask, cgroup CPU quota %d cores; the reference pins worker i to CPU i (ThreadPool.cpp:26-39)" % (aff, cores))
        for suf in (".ref_1.fq", ".ref_2.fq"):
            if os.path.exists(fa + suf):
                os.remove(fa + suf)
        noaff = os.path.join(ROOT, "oracle", "_ref", "libnoaffinity.so")
        if out and os.path.exists(noaff):                            # the same run with the workers NOT pinned to CPUs 0..t-1 (oracle/noaffinity.cpp)
            t0 = time.perf_counter()
            r = subprocess.run([ref, "genreads", "-i", fa, "-m", prof, "-c", "%g" % coverage, "-t", str(cores), "-o", fa + ".ref"], capture_output=True, text=True,
                               env=dict(os.environ, LD_PRELOAD=noaff))
            secs = time.perf_counter() - t0
            if r.returncode == 0 and os.path.exists(fa + ".ref_1.fq"):
                pairs = sum(1 for _ in open(fa + ".ref_1.fq", "rb")) // 4
                out["unpinned"] = dict(value=pairs / secs, unit="pairs/s", cores=cores,
                                       sample="the same run with pthread_setaffinity_np made a no-op (LD_PRELOAD oracle/_ref/libnoaffinity.so): %d pairs in %.1f s" % (pairs, secs))
            for suf in (".ref_1.fq", ".ref_2.fq"):
                if os.path.exists(fa + suf):
                    os.remove(fa + suf)
        if out:
            # what the same binary did in the survey container (Xeon 2.1 GHz, 8 vCPU, 1 Mb, HiSeq2500 125 bp, 30x)
            out["baseline_md_container"] = {"t1_pairs_per_s": 5200, "t8_pairs_per_s": 27900, "what": "BASELINE.md section 2: 1 Mb diploid, HiSeq2500 (125 bp) PE 30x, -t 1 / -t 8 on 8 vCPU"}
        if out:                                                     # -t 1 beside -t cores, on the sample's first quarter (the rate per core)
            n1 = max(100000, len(seqs[0]) // 16)
            fa1 = os.path.join(td, "cpu_sample_t1.fa")
            write_simu_fasta(fa1, ["1_1_%d" % n1, "1_2_%d" % n1], [seqs[0][:n1], seqs[0][:n1]])
            t0 = time.perf_counter()
            r = subprocess.run([ref, "genreads", "-i", fa1, "-m", prof, "-c", "%g" % coverage, "-t", "1", "-o", fa1 + ".ref"], capture_output=True, text=True)
            secs = time.perf_counter() - t0
            if r.returncode == 0 and os.path.exists(fa1 + ".ref_1.fq"):
                pairs = sum(1 for _ in open(fa1 + ".ref_1.fq", "rb")) // 4
                out["one_thread"] = dict(value=pairs / secs, unit="pairs/s", cores=1, sample="first %.2f Mb of the same sample: %d pairs in %.1f s of `scssim genreads -t 1`" % (n1 / 1e6, pairs, secs))
            for suf in (".ref_1.fq", ".ref_2.fq", "", ".fai"):
                if os.path.exists(fa1 + suf):
                    os.remove(fa1 + suf)
    if os.path.exists(oracle):
        r = subprocess.run([oracle, "genreads", "-i", fa, "-m", prof, "-c", "%g" % coverage, "-t", str(cores), "-o", fa + ".cpu", "--rng", "counter", "--seed", "7"],
                           capture_output=True, text=True)
        m = re.search(r"pairs=(\d+) \| load ([\d.]+)s frag ([\d.]+)s amplify ([\d.]+)s alloc ([\d.]+)s readgen ([\d.]+)s", r.stderr)
        if r.returncode == 0 and m:
            pairs = int(m.group(1))
            secs = sum(float(m.group(i)) for i in (3, 4, 5, 6))
            port = dict(value=pairs / secs, unit="pairs/s", cores=cores, kind="port",
                        sample="%s: %d pairs in %.1f s (amplify %.1f, allocate %.1f, readgen + files %.1f)" % (sample_desc, pairs, secs, float(m.group(4)), float(m.group(5)), float(m.group(6))))
            if out:
                out["port"] = port
            else:
                out = port
        for suf in (".cpu_1.fq", ".cpu_2.fq"):
            if os.path.exists(fa + suf):
                os.remove(fa + suf)
    os.remove(fa)
    return out or None


def summary_rows(path):
    """rows (kernel, a, b, c) of a profiles/ summary written by tools/pmc_summary.py / sq_summary.py: csv with quoted kernel names (rounds
    1-4 wrote them unquoted: the name is then what precedes the last three fields)"""
    import csv
    for f in csv.reader(l for l in open(path) if not l.startswith("#")):
        if len(f) >= 4 and f[0] != "kernel":
            yield [",".join(f[:-3])] + f[-3:]


def committed_counters(kernel_prefix):
    """Counters bench.py cannot collect itself, read from the newest committed rocprofv3 summaries of this same command
    (tools/profile_bench.sh -> profiles/r*_bench_pmc_hbm.csv: FETCH_SIZE / WRITE_SIZE in separate --pmc passes, KB -> bytes, per
    launch; profiles/r*_bench_sq.csv: SQ_INSTS_VALU, active lanes, SQ_INSTS_SALU per launch).  A timed "launch" of k_reads is
    its two class kernels (event-free reads, reads with indel events) back to back: their rows are added.  None when absent."""
    res = {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_hbm.csv")))
    if files:
        tot = totf = 0.0
        for name, fetch, write, n in summary_rows(files[-1]):
            if name.startswith(kernel_prefix):
                tot += (float(fetch) + float(write)) * 1024.0
                totf += float(fetch) * 1024.0
        if tot:
            res["traffic"] = tot
            res["traffic_fetch"] = totf
            res["traffic_source"] = os.path.relpath(files[-1], ROOT)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_sq.csv")))
    if files:
        valu = salu = lanes_w = 0.0
        for f in summary_rows(files[-1]):
            if f[0].startswith(kernel_prefix):
                try:
                    v = float(f[1]); valu += v; lanes_w += v * float(f[2]); salu += float(f[3])
                except ValueError:
                    pass
        if valu:
            res["valu_insts_per_launch"] = valu
            res["lanes_per_valu_inst"] = lanes_w / valu
            res["salu_insts_per_launch"] = salu
            res["sq_source"] = os.path.relpath(files[-1], ROOT)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_meta.json")))
    if files:
        try:
            res["profile_pairs_per_launch"] = float(json.load(open(files[-1]))["pairs_per_launch"])
        except (OSError, ValueError, KeyError):
            pass
    return res


def committed_lane_table():
    """Active lanes per VALU instruction of every kernel in the newest committed SQ summary (profiles/r*_bench_sq.csv): the
    lane-utilisation side of the kernels that are bound by instruction issue rather than by HBM."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_sq.csv")))
    if not files:
        return None
    rows = {}
    for f in summary_rows(files[-1]):
        if not f[0].startswith("scs::"):
            continue
        try:
            rows[f[0][5:]] = {"lanes_per_valu_inst": float(f[2]), "valu_insts_per_launch": float(f[1]), "salu_per_valu": float(f[3]) / max(1.0, float(f[1]))}
        except ValueError:
            pass
    top = dict(sorted(rows.items(), key=lambda kv: -kv[1]["valu_insts_per_launch"])[:12])   # the twelve with the most instructions per launch
    return {"source": os.path.relpath(files[-1], ROOT), "kernels": top} if top else None


class Cleaner:
    """Unlinks the files of finished steps on background threads (os.unlink releases the GIL), so that tmpfs holds about one
    step's FASTQ at a time.  `threads` of them work while a step's writers need the cores; gate(limit) -- wait until at most
    `limit` bytes are still queued: the next step's sink starts behind it, INSIDE the timed region -- lets all `burst` of them
    work, the writers being idle then."""

    def __init__(self, threads=4, burst=16, cpus=None):
        import queue
        import threading
        self.cpus = set(cpus) if cpus else None                      # the GPU's NUMA node: freeing a page from the node it lies on
        self.q = queue.Queue()
        self.lock = threading.Lock()
        self.cv = threading.Condition(self.lock)
        self.outstanding = 0
        self.freed = 0
        self.base, self.burst = threads, max(threads, burst)
        self.limit = threads                                         # workers allowed to unlink at the same time
        self.active = 0
        self.th = [threading.Thread(target=self._run, daemon=True) for _ in range(self.burst)]
        for t in self.th:
            t.start()

    def _run(self):
        if self.cpus:
            try:
                os.sched_setaffinity(0, self.cpus)                   # (the calling thread)
            except OSError:
                pass
        while True:
            p = self.q.get()
            if p is None:
                return
            with self.cv:
                while self.active >= self.limit:
                    self.cv.wait(0.05)
                self.active += 1
            try:
                n = os.path.getsize(p)
                os.unlink(p)
            except OSError:
                n = 0
            with self.cv:
                self.active -= 1
                self.outstanding -= n if n <= self.outstanding else self.outstanding
                self.freed += n
                self.cv.notify_all()

    def add(self, paths):
        sized = []
        for p in paths:
            try:
                sized.append((os.path.getsize(p), p))
            except OSError:
                pass
        with self.cv:
            self.outstanding += sum(n for n, _ in sized)
        for _, p in sorted(sized, reverse=True):
            self.q.put(p)

    def boost(self):
        """all `burst` workers may unlink from now until the next gate() returns: a step's first phases (fragments, amplification,
        allocation: 0.15 s) run on the GPU alone, no writer needs a core yet"""
        with self.cv:
            self.limit = self.burst
            self.cv.notify_all()

    def gate(self, limit):
        t0 = time.perf_counter()
        with self.cv:
            self.limit = self.burst
            self.cv.notify_all()
            while self.outstanding > limit:
                self.cv.wait(0.02)
            self.limit = self.base
        return time.perf_counter() - t0

    def watch(self, files_by_part, k):
        """files_by_part[p] = the files of part p, made generation by generation with k writers: part p is final once part p + k
        exists (scs_yield_reads_files_ex) -- a polling thread hands the finished parts to the unlinkers while the step still runs.
        Returns a function that stops the watch and queues what is left."""
        import threading
        stop = threading.Event()
        done = set()

        def poll():
            while not stop.is_set():
                for p in range(len(files_by_part) - k):
                    if p not in done and os.path.exists(files_by_part[p + k][0]):
                        done.add(p)
                        self.add(files_by_part[p])
                stop.wait(0.02)
        th = threading.Thread(target=poll, daemon=True)
        th.start()

        def finish(extra):
            stop.set(); th.join()
            self.add([f for p in range(len(files_by_part)) if p not in done for f in files_by_part[p]] + list(extra))
        return finish

    def close(self):
        self.gate(0)
        for _ in self.th:
            self.q.put(None)


def roofline_of(ktimes, fq_bytes, L, dom=None):
    """roofline object of the dominant kernel of a set of timed launches (HIP events on the stream each kernel is launched on)."""
    dom = dom or max(ktimes, key=lambda k: ktimes[k]["ms"])
    kd = ktimes[dom]
    if dom in ("k_reads", "k_indels"):
        # per pair: insert-size template bytes (mean 261 at -s 260) + the FASTQ bytes of both records; the
        # indel pass alone: the pair record read + 20 B per read written
        alg = (261.0 * kd["units"] + fq_bytes) if dom == "k_reads" else (56.0 + 2 * 20.0 + 8.0) * kd["units"]
        note = ("(261 B template + FASTQ bytes of both records) x %d pairs over %d launches" if dom == "k_reads" else "(56 B pair record + 2 x 20 B events + 8 B sizes) x %d pairs over %d launches") % (kd["units"], kd["launches"])
        draws = 4.0 * L * kd["units"] if dom == "k_reads" else 2.0 * L * kd["units"]
        draws_note = "k_reads: 2 draws (substitution, quality) per output base x 2 mates" if dom == "k_reads" else "k_indels: 1 draw per input base x 2 mates"
        survey_alg = None
    else:
        # amplification: the implementation's compulsory bytes per created amplicon (28 B record written, 8 B slot written and
        # read back, 8 primer bases read, 4 B stock counter RMW, ~20 B of the parent's record read); SURVEY 8(d)'s 1526 B
        # (the 1-2 kb template window read once) is not what this design moves -- GC comes from the bit index and the
        # error count from one binomial draw -- so it is kept as a second figure only (it would put frac above 1)
        made = ktimes["k_errs<semi->full>"]["units"] + ktimes["k_errs<frag->semi>"]["units"]
        alg = 68.0 * made * kd["ms"] / max(1e-9, sum(ktimes[k]["ms"] for k in ktimes if k.startswith("k_attach") or k.startswith("k_errs")))
        note = "68 B compulsory per created amplicon x %d amplicons, share of %s in the amplification time" % (made, dom)
        draws = None; draws_note = None
        survey_alg = 1526.0 * made
    achieved = alg / (kd["ms"] * 1e-3) / 1e9 if kd["ms"] > 0 else 0.0
    prefix = {"k_reads": "scs::k_reads", "k_indels": "scs::k_indels", "k_attach<semi>": "scs::k_attach_dense", "k_attach<frag>": "scs::k_attach<true",
              "k_errs<semi->full>": "scs::k_errs<false>", "k_errs<frag->semi>": "scs::k_errs<true>"}[dom]
    cc = committed_counters(prefix)
    # the committed counters are per launch of the profiled command (one GPU, 8 M-pair batches); a sharded run launches smaller
    # batches: scaled by pairs per launch (the k_reads / k_indels counters are proportional to the pairs of a launch)
    if dom in ("k_reads", "k_indels") and cc.get("profile_pairs_per_launch") and kd["launches"]:
        scale = (kd["units"] / kd["launches"]) / cc["profile_pairs_per_launch"]
        for key in ("traffic", "traffic_fetch", "valu_insts_per_launch", "salu_insts_per_launch"):
            if key in cc:
                cc[key] *= scale
    roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": cc.get("traffic"), "traffic_source": cc.get("traffic_source"), "algorithmic_bytes": note,
            # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of WIDE coalesced reads (16 B per lane) and is uncalibrated for other
            # widths.  tools/probes/fetch_calib.hip (profiles/r03_fetch_size_calibration.txt): the counter is 64 B per memory-side request -- 1/2 for wide
            # streaming reads, 64 B per random dword gather and per 44-byte window gather.  This kernel's reads are mostly such gathers, so the raw
            # `traffic` is close to the bytes moved; this figure (every fetched byte doubled) is the bound if all of them were wide
            "traffic_if_every_read_is_doubled": (cc["traffic"] + cc["traffic_fetch"]) if "traffic_fetch" in cc else None,
            "algorithmic_bytes_per_launch": alg / max(1, kd["launches"]), "avg_launch_ms": kd["ms"] / max(1, kd["launches"]),
            "timed_launches": kd["launches"], "pairs_per_launch": kd["units"] / max(1, kd["launches"]) if dom in ("k_reads", "k_indels") else None}
    if draws:
        roof["draws_per_s"] = draws / (kd["ms"] * 1e-3)
        roof["draws"] = draws_note
    if "valu_insts_per_launch" in cc:
        # VALU issue: 256 CUs x 4 SIMDs at 2.4 GHz.  Priced against the BEST rate tools/probes/valu_peak.hip reached with independent integer
        # chains (3.18 cycles per wave-instruction: 7.72e11/s, profiles/r02_valu_peak.log); the microarchitecture guide's 2-cycle
        # issue (1.23e12/s, quoted for v_fma_f32) beside it
        per_s = cc["valu_insts_per_launch"] / (kd["ms"] * 1e-3 / max(1, kd["launches"]))
        roof["valu"] = {"insts_per_launch": cc["valu_insts_per_launch"], "lanes_per_inst": cc.get("lanes_per_valu_inst"),
                        "salu_per_valu": cc.get("salu_insts_per_launch", 0) / max(1.0, cc["valu_insts_per_launch"]),
                        "issue_frac": per_s / VALU_PEAK_MEASURED, "issue_frac_vs_2_cycle_spec": per_s / VALU_PEAK_SPEC,
                        "peak_wave_insts_per_s": VALU_PEAK_MEASURED, "spec_wave_insts_per_s": VALU_PEAK_SPEC, "source": cc.get("sq_source")}
    if survey_alg:
        roof["survey_model_bytes"] = survey_alg
    return roof


def small_config(torch, scssim_amd, dev, stream, prof, mb, cov, label, shm, steps=5, concurrent=0):
    """One of the smaller BASELINE configurations as an extra line of the same record: the whole job with the text left in HBM
    (`steps` steps) and once with its two FASTQ files on tmpfs."""
    lens = [int(mb * 1e6)]
    names, rl, bases = synth_genome(torch, dev, lens, 7000 + int(mb))
    g = scssim_amd.GenReads(profile=prof, coverage=cov, isize=260, layout="PE", seed=1, device=dev.index, stream=stream.cuda_stream)
    g.upload_genome_device(names, rl, bases.data_ptr())
    bases_keep = bases if concurrent > 1 else None
    del bases
    last_stats = {}

    def one(i, files=None):
        g.set_seed(500 + i)
        g.create_frags(); g.amplify(); g.allocate_reads(0)
        if files:
            g.yield_reads_files(files)
        else:
            g.yield_reads_sink(None)
        last_stats.update(g.stats())
        return g.stats()["pairs_written"]
    one(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); pairs = sum(one(1 + i) for i in range(steps)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sd = tempfile.mkdtemp(prefix="scsbench_small_", dir=shm)
    try:
        one(50, os.path.join(sd, "w"))                              # (pins the sink's buffers)
        t1 = time.perf_counter(); pf = one(51, os.path.join(sd, "r")); tf = time.perf_counter() - t1
    finally:
        shutil.rmtree(sd, ignore_errors=True)
    g.close()
    out = {"workload": label, "pairs_per_step": pairs // steps, "generation_hbm_pairs_per_s": pairs / dt, "ms_per_step_hbm": 1e3 * dt / steps,
           "with_two_fastq_files_on_tmpfs_pairs_per_s": pf / tf, "ms_per_step_files": 1e3 * tf,
           # fresh pages of ONE tmpfs file come at 5.3-5.7 GB/s whatever the number of threads (the inode lock: profiles/r04_onefile_probe.log), so the
           # reference's two-file layout cannot take this job's text faster than bytes / 2 / 5.5 GB/s, whatever the GPU does
           "two_file_estimate_ms_at_5p5_GBps_per_file": 1e3 * (g_fastq_bytes / 2) / 5.5e9 if (g_fastq_bytes := sum(last_stats["fastq_bytes"])) else None}
    if concurrent > 1:
        # many small jobs at once (a many-cell run): `concurrent` ctxs on as many streams and host threads, each with its own copy of the genome,
        # the text left in HBM -- the aggregate rate is what such a run sees of the GPU (one job alone leaves the chip mostly empty: 780 workgroups
        # in its largest launch)
        import threading
        gs = []
        for k in range(concurrent):
            st_k = torch.cuda.Stream()
            gk = scssim_amd.GenReads(profile=prof, coverage=cov, isize=260, layout="PE", seed=1, device=dev.index, stream=st_k.cuda_stream)
            gk.upload_genome_device(names, rl, bases_keep.data_ptr())
            gs.append((gk, st_k))
        done = [0] * concurrent

        def work(k, n):
            gk = gs[k][0]
            for i in range(n):
                gk.set_seed(9000 + 97 * k + i)
                gk.create_frags(); gk.amplify(); gk.allocate_reads(0); gk.yield_reads_sink(None)
                done[k] += gk.stats()["pairs_written"]
        for n, timed_run in ((2, False), (4 * steps, True)):
            for k in range(concurrent):
                done[k] = 0
            th = [threading.Thread(target=work, args=(k, n)) for k in range(concurrent)]
            torch.cuda.synchronize(); t2 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize(); dt2 = time.perf_counter() - t2
            if timed_run:
                out["concurrent_jobs"] = {"ctxs": concurrent, "jobs": concurrent * n, "aggregate_pairs_per_s": sum(done) / dt2, "ms_per_job_amortised": 1e3 * dt2 / (concurrent * n),
                                          "what": "%d ctxs on %d streams and host threads, each running %d jobs of this size back to back, text left in HBM" % (concurrent, concurrent, n)}
        for gk, _ in gs:
            gk.close()
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher on the command line: start the N ranks as fresh children -- the same
    `python -m torch.distributed.run` command the driver uses -- BEFORE this process has made any GPU call (it never makes one: torch
    is not even imported here), pass rank 0's one JSON line through (the children inherit stdout / stderr), and return the launcher's
    exit status: non-zero when any rank died (torch.distributed.run ends the other ranks then).  Replaces the reference's pool fan-out,
    lib/malbac/Malbac.cpp:318-368,438-454, at process granularity."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env)
    try:
        return p.wait()
    except KeyboardInterrupt:
        p.terminate()
        try:
            return p.wait(10) or 130
        except subprocess.TimeoutExpired:
            p.kill()
            return 130


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mb", type=float, default=0.0, help="scale the 24 hg19-like records to this many Mb (default: the real 3096 Mb)")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--writers", type=int, default=0, help="part files per mate and writer threads of the FASTQ sink (default: host cores - 4, at most 12)")
    ap.add_argument("--generations", type=int, default=6, help="the part files are made in this many generations (writers x generations parts per mate): a generation's files are final when the next starts")
    ap.add_argument("--cleaners", type=int, default=6, help="background threads that unlink finished part files")
    ap.add_argument("--out-dir", default="", help="where the timed steps write their FASTQ part files (default: a fresh directory on /dev/shm)")
    ap.add_argument("--hbm-only", action="store_true", help="time the steps with a NULL sink (text generated into HBM buffers, no files): the generation_hbm leg as the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the generation-only, D2H-only, small-configuration and CLI legs")
    ap.add_argument("--cpu-sample-mb", type=float, default=4.0)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))                               # (this process never touches the GPU)

    import torch
    import torch.distributed as dist
    import scssim_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d: one rank per GPU" % (a.gpus, world))
    # rehearsal of the N > 1 control flow on a box with ONE GPU: SCS_BENCH_BACKEND=gloo SCS_BENCH_ONE_DEVICE=1 (all ranks on
    # cuda:0, collectives staged through the CPU).  Never a measurement.
    backend = os.environ.get("SCS_BENCH_BACKEND", "nccl")
    cpu_coll = backend != "nccl"
    if os.environ.get("SCS_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cpu_coll:
            dist.init_process_group(backend)
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if cpu_coll else dev

    td = tempfile.mkdtemp(prefix="scsbench_")
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else td
    prof = make_profile(td)
    lens = record_lengths(a.genome_mb)
    t_gen = time.perf_counter()
    names, rl, bases = synth_genome(torch, dev, lens, 3000)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    stream = torch.cuda.Stream()
    g = scssim_amd.GenReads(profile=prof, coverage=a.coverage, isize=260, layout="PE", seed=1, device=local,
                            stream=stream.cuda_stream, shard_rank=rank, shard_count=world)
    t_up = time.perf_counter()
    g.upload_genome_device(names, rl, bases.data_ptr())
    t_up = time.perf_counter() - t_up
    # samples for the CPU / CLI legs come from the same genome: taken before it is released
    cpu_sample = cli_sample = None
    if rank == 0 and world == 1:
        n_cpu = min(rl[0], int(a.cpu_sample_mb * 1e6))
        cpu_sample = bases[:n_cpu].cpu().numpy()
        if not a.no_extra_legs:
            i20 = 2 * 19 if len(rl) >= 40 else 0                     # chromosome 20 (configs[2]: chr20-size)
            o20 = sum(rl[:i20])
            cli_sample = bases[o20:o20 + rl[i20]].cpu().numpy()
    del bases
    torch.cuda.empty_cache()
    coll_path = None
    if world > 1:
        if cpu_coll:                                                 # 1-GPU rehearsal: torch.distributed (gloo) hooks
            from scssim_amd.dist import Collectives
            coll = Collectives(stream=stream)
            g.set_collectives(coll, device_hooks=True)
            coll_path = "rehearsal: torch.distributed (%s) hooks staged through the CPU" % backend
        else:                                                        # RCCL inside the library: rank 0's id to every rank
            why = ""
            try:
                ids = [scssim_amd.comm_unique_id() if rank == 0 else None]
            except Exception as e:                                   # the broadcast below must still be entered by every rank
                ids, why = [None], repr(e)
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is not None:
                try:
                    g.comm_init(ids[0], rank, world)
                except Exception as e:
                    why = repr(e)
            else:
                why = why or "rank 0 could not create a communicator id"
            good = torch.tensor([0 if why else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            coll_path = "RCCL communicator inside the library (scs_comm_init): ncclCommCount = %d ranks" % g.comm_count()
            if int(good[0]) == 0:                                    # every rank takes the same path: the device hooks over torch's RCCL group
                if why:
                    print("rank %d: scs_comm_init failed (%s); collectives through torch.distributed's RCCL group" % (rank, why), file=sys.stderr)
                from scssim_amd.dist import Collectives
                coll = Collectives(stream=stream)
                g.set_collectives(coll, device_hooks=True)
                coll_path = "torch.distributed RCCL group on the library's HBM buffers (device hooks), %d ranks" % dist.get_world_size()

    cores = host_cores()
    # N ranks on one node share its cores: every rank's sink gets cores / N of them (12 writers + 6 cleaners on the 16 cores a one-GPU box
    # has is the measured best, profiles/r03_sink_tune_writers_cleaners.log), on the CPUs of its own GPU's NUMA node (the library binds
    # its writer threads and pinned slots there itself; the cleaners here, and this rank's main thread when the node has several GPUs)
    share = cores // max(1, world if world > 1 and not cpu_coll else 1)
    writers = a.writers or max(1, min(12, share - 4))
    n_clean = max(1, a.cleaners if world == 1 else min(a.cleaners, max(2, share // 4)))
    local_cpus = scssim_amd.gpu_local_cpus(local)
    if world > 1 and local_cpus and not cpu_coll:
        try:
            os.sched_setaffinity(0, set(local_cpus))
        except OSError:
            pass
    out_dir = a.out_dir or tempfile.mkdtemp(prefix="scsbench_out_", dir=shm)
    os.makedirs(out_dir, exist_ok=True)
    generations = a.generations
    cleaner = Cleaner(threads=n_clean, burst=max(share, n_clean), cpus=local_cpus)
    GATE_BYTES = (40 << 30) // max(1, world)                                            # a step's sink starts when at most this much of the older steps' text is still on tmpfs

    def acc(ktimes, kt, names_):
        for k in names_:
            d = ktimes.setdefault(k, dict(launches=0, ms=0.0, units=0))
            for f in ("launches", "ms", "units"):
                d[f] += kt[k][f]

    def step(i, ktimes, stage, mode):
        """one whole job with a fresh seed; mode: 'files' (K part files per mate), 'null' (text stays in HBM), or a sink callable"""
        g.set_seed(1000 + i)
        if mode in ("files", "bgzf"):
            cleaner.boost()                                          # (the step before's files: every core may free them until this step's sink starts)
        t0 = time.perf_counter(); g.create_frags()
        t1 = time.perf_counter(); g.amplify()
        acc(ktimes, g.kernel_times(), ("k_errs<semi->full>", "k_errs<frag->semi>", "k_attach<semi>", "k_attach<frag>"))
        t2 = time.perf_counter(); g.allocate_reads(0)
        t3 = time.perf_counter()
        waited = 0.0
        if mode in ("files", "bgzf", "files_alone"):
            waited = cleaner.gate(0 if mode == "files_alone" else GATE_BYTES)
            base = os.path.join(out_dir, "step%d" % i)
            lbase = base + (".r%d" % rank if world > 1 else "")
            pp = scssim_amd.part_paths(lbase, writers * generations, True, ".fq.gz" if mode == "bgzf" else ".fq")
            if mode == "files_alone":                                # one job on an EMPTY tmpfs, nothing unlinked while it runs: what a single run sees
                def finish(extra):
                    cleaner.add([f for p in range(writers * generations) for f in (pp[0][p], pp[1][p])] + list(extra))
            else:
                finish = cleaner.watch([[pp[0][p], pp[1][p]] for p in range(writers * generations)], writers)
            try:
                # the library's file sink (SeqWriter): D2H + `writers` threads, a part file per mate each, generation by generation
                g.yield_reads_files(base, writers, generations, mode == "bgzf")
            finally:
                finish([f for f in (lbase + ".parts", lbase + ".idx") if os.path.exists(f)])
        elif mode == "files_in_place":                               # every step writes over the files of the step before (SCS_SINK_IN_PLACE): nothing to unlink
            g.yield_reads_files(os.path.join(out_dir, "inplace"), writers, generations, False, True)
        elif mode == "null":
            g.yield_reads_sink(None)                                 # generate into HBM batch buffers and count
        else:
            g.yield_reads_sink(mode)
        t4 = time.perf_counter()
        acc(ktimes, g.kernel_times(), ("k_reads", "k_indels"))
        for k, v in (("frags", t1 - t0), ("amplify", t2 - t1), ("allocate", t3 - t2), ("reads", t4 - t3 - waited), ("wait_for_cleanup", waited if mode != "files_alone" else 0.0)):
            stage[k] = stage.get(k, 0.0) + v
        return g.stats()

    def timed(n_warm, n_steps, mode, first):
        """n_warm untimed + n_steps timed steps; returns (elapsed, pairs, fastq bytes, amplicons, kernel times, stage seconds)"""
        kt, stage = {}, {}
        for i in range(n_warm):                                      # the first step also maps the device buffers and pins the sink's
            step(first + i, kt, stage, mode)
        kt.clear(); stage.clear()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pairs = fqb = amps = 0; per_step = []
        for i in range(n_steps):
            ts = time.perf_counter()
            last = step(first + n_warm + i, kt, stage, mode)
            per_step.append(time.perf_counter() - ts)                   # (this rank's view; the timed region is the barrier-to-barrier total)
            pairs += last["pairs_written"]; fqb += sum(last["fastq_bytes"]); amps += last["semi_amplicons"] + last["full_amplicons"]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        per_rank = None
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
            pt = torch.tensor([pairs, fqb, amps], dtype=torch.int64, device=cdev)
            dist.all_reduce(pt)
            # every rank's own stage times and bytes: the curve over N separates what the GPUs do (amplify, allocate, reads into HBM) from
            # what the host does (the sink: bytes / reads-stage seconds per rank)
            mine = torch.tensor([stage.get(k, 0.0) / n_steps for k in ("amplify", "allocate", "reads", "wait_for_cleanup")] + [fqb / n_steps, pairs / n_steps], dtype=torch.float64, device=cdev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [dict(rank=r, amplify_s=float(v[0]), allocate_s=float(v[1]), reads_s=float(v[2]), wait_for_cleanup_s=float(v[3]), fastq_GB=float(v[4]) / 1e9,
                             pairs=int(v[5]), sink_GBps=(float(v[4]) / 1e9 / max(1e-9, float(v[2]))) if mode != "null" else None) for r, v in enumerate(allr)]
            pairs, fqb_all, amps = int(pt[0]), int(pt[1]), int(pt[2])
        else:
            fqb_all = fqb
        return dict(elapsed=elapsed, pairs=pairs, fq_bytes_local=fqb, fq_bytes=fqb_all, amps=amps, ktimes=kt, stage={k: v / n_steps for k, v in stage.items()}, per_rank=per_rank, per_step=per_step)

    main_mode = "null" if a.hbm_only else "files"
    R = timed(a.warmup, a.steps, main_mode, 0)
    cleaner.gate(0)                                                  # (outside the timed region: the last step's files)
    # ---- generation only (every rank takes part: the steps hold the job's collectives): the same job with a NULL sink -- the text is
    # written into HBM batch buffers (8 M pairs per launch) and counted.  What the kernels do when nothing has to cross PCIe; the
    # dominant kernel's roofline is taken here.
    H = timed(1, 5, "null", 100) if not a.no_extra_legs and not a.hbm_only else None

    if rank == 0:
        L = g.read_length
        elapsed, pairs_total = R["elapsed"], R["pairs"]
        out = {
            "metric": "paired-end read pairs/s (whole genreads job: MALBAC amplification + read allocation + read generation + FASTQ files, PE150 30x)",
            "value": pairs_total / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,   # the same job on 1, 2, 4, 8 GPUs
            "dtype": "u8/u32 (integer draws, byte sequences; fp64 only in the GC-weight draw and the allocation sums)", "data": "synthetic",
            "config": {"workload": ("configs[3] on %d GPU(s): %.0f Mb synthetic diploid genome (24 hg19-length records x 2 haplotypes, i.i.d. 30/20/20/30 %% ACGT, generated in HBM), "
                                    "PE150 %gx, HiSeq2500 model resampled to 150 bins, -p 100000 -r 1e-9 -s 260" % (world, sum(lens) / 1e6, a.coverage)),
                       "pairs_per_step": pairs_total // a.steps, "amplicons_per_step": R["amps"] // a.steps, "fastq_bytes_per_step": R["fq_bytes"] // a.steps,
                       "sharding": ("one job over %d GPUs by fragment lineage; per-pass primer-stock all-reduce + allocation partials over RCCL; a FASTQ shard per rank" % world) if world > 1 else "single GPU",
                       "collectives": coll_path,
                       "output": ("FASTQ text generated batch by batch into HBM buffers (NULL sink): --hbm-only" if a.hbm_only else
                                  "plain FASTQ files on tmpfs (%s): %d part files per mate%s, each a contiguous range of the records (scs_yield_reads_files_ex: %d writer threads x %d generations; "
                                  "`cat` of the parts in order is the reference's <prefix>_1.fq / _2.fq).  Every step writes fresh files; the memory cgroup cannot hold two steps' text, so %d background "
                                  "threads unlink every part once it is final (a generation's parts are final when the next generation starts) and a step's sink starts when <= 40 GB of older text (over all ranks) "
                                  "are left (`wait_for_cleanup` in stages_s_per_step): all of it inside the timed region"
                                  % (out_dir, writers * generations, " and rank" if world > 1 else "", writers, generations, max(1, a.cleaners))),
                       "sink_GBps": None if a.hbm_only else R["fq_bytes"] / max(1e-9, R["stage"]["reads"] * a.steps) / 1e9,
                       "host_cores": cores, "sink_threads_per_rank": {"writers": writers, "cleaners": n_clean, "bound_to_gpu_numa_node_cpus": len(local_cpus) or None}},
            "stages_s_per_step": R["stage"],
            # the steps one by one (rank 0's clock): the job with files is bound by the host, and its spread from step to step and box to box
            # (46-60 M pairs/s over rounds 3 and 4) is the host's
            "ms_per_step_each": [1e3 * t for t in R["per_step"]],
            "per_rank": R["per_rank"],
            "kernels_ms_per_step": {k: v["ms"] / a.steps for k, v in R["ktimes"].items()},
            "setup_s": {"genome_generated_in_hbm": t_gen, "genome_staged_(encode+bit_index)": t_up},
            "lane_utilisation": committed_lane_table(),
        }
        if H:
            out["generation_hbm"] = {"value": H["pairs"] / H["elapsed"], "unit": "pairs/s", "steps": 5, "ms_per_step": 1e3 * H["elapsed"] / 5, "stages_s_per_step": H["stage"],
                                     "kernels_ms_per_step": {k: v["ms"] / 5 for k, v in H["ktimes"].items()},
                                     "what": "the same job, FASTQ text generated batch by batch into HBM buffers and counted (NULL sink): no PCIe, no files",
                                     "per_rank": H["per_rank"],
                                     # whole job against HBM: the implementation's compulsory bytes (68 B per amplicon, 56 B pair record written + read,
                                     # 40 B events, 261 B template + FASTQ per pair); SURVEY 8(d)'s model (1526 B per amplicon) beside it
                                     "whole_job_GBps": {"compulsory": (68.0 * H["amps"] + (261.0 + 152.0) * H["pairs"] + H["fq_bytes"]) / H["elapsed"] / 1e9,
                                                        "survey_8d_model": (1526.0 * H["amps"] + 261.0 * H["pairs"] + H["fq_bytes"]) / H["elapsed"] / 1e9}}
        # the dominant kernel's roofline: its launches INSIDE the timed region (sink batches of 2 M pairs on the whole genome; 8 M with
        # --hbm-only), HIP events on the ctx stream around every one; the generation_hbm leg's launches (8 M pairs, nothing else on the
        # chip's memory system) beside it
        roof = roofline_of(R["ktimes"], R["fq_bytes_local"], L, "k_reads")
        roof["timing"] = "HIP events on the ctx stream around every launch of the timed region"
        if main_mode == "files":
            # (measured: profiles/r04_sdma_vs_blit.log, r04_bench_files_kernel_stats.csv)
            roof["under_a_profiler"] = ("rocprofv3 --kernel-trace makes the runtime copy D2H with shader kernels (__amd_rocclr_copyBuffer) instead of the SDMA engines; they share the CUs "
                                        "with k_reads, whose launches then take 22 ms instead of 1.6 (HIP events and the trace agree on that inside such a run; HSA_ENABLE_SDMA=0 without a "
                                        "profiler gives the same 21 ms). `value` does not move (the host binds it). The undisturbed cross-check of the kernel's duration is the generation_hbm "
                                        "leg against profiles/r04_bench_kernel_stats.csv (no copies in flight)")
        if H:
            hb = roofline_of(H["ktimes"], H["fq_bytes_local"], L, roof["kernel"])
            roof["generation_hbm_leg"] = {k: hb[k] for k in ("achieved", "frac", "avg_launch_ms", "timed_launches", "pairs_per_launch", "algorithmic_bytes_per_launch")}
            amp = {k: v for k, v in H["ktimes"].items() if k.startswith("k_attach") or k.startswith("k_errs")}
            made = H["ktimes"]["k_errs<semi->full>"]["units"] + H["ktimes"]["k_errs<frag->semi>"]["units"]
            amp_ms = sum(v["ms"] for v in amp.values())
            lanes = (committed_lane_table() or {}).get("kernels", {})
            # amplification has its own entry: this design's compulsory 68 B per created amplicon over the four kernels' time
            out["roofline_amplification"] = {"bound": "hbm (latency / issue bound in practice)", "achieved": 68.0 * made / max(1e-9, amp_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": 68.0 * made / max(1e-9, amp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "compulsory_bytes_per_amplicon": 68, "amplicons": made, "kernels_ms": {k: v["ms"] for k, v in amp.items()},
                                             "lanes_per_valu_inst": {k: v["lanes_per_valu_inst"] for k, v in lanes.items() if k.startswith("k_attach") or k.startswith("k_errs")},
                                             "salu_per_valu": {k: v["salu_per_valu"] for k, v in lanes.items() if k.startswith("k_attach") or k.startswith("k_errs")}}
        out["roofline"] = roof
        if world == 1 and not a.no_extra_legs and not a.hbm_only:
            # ---- D2H only: the same batches through a sink that only counts -- what PCIe allows
            seen = [0]

            def count_only(_u, _p1, n1, _p2, n2):
                seen[0] += n1 + n2
                return 0
            D = timed(0, 1, count_only, 200)
            out["d2h_only"] = {"value": D["pairs"] / D["elapsed"], "unit": "pairs/s", "seconds": D["elapsed"], "GBps": seen[0] / max(1e-9, D["stage"]["reads"]) / 1e9,
                               "what": "one step with a sink that only counts the bytes it is handed: D2H into pinned slots, nothing written"}
            # ---- ONE job alone: the tmpfs empty when it starts, nothing unlinked while it runs.  The timed steps above pay, inside the
            # timed region, for getting rid of the step before (the memory cgroup cannot hold two steps' text): 6 unlinking threads beside
            # the 12 writers on 16 cores.  A user's single job has no step before it.
            try:
                cleaner.gate(0)
                A1 = timed(0, 1, "files_alone", 250)
                out["single_job_on_empty_tmpfs"] = {"value": A1["pairs"] / A1["elapsed"], "unit": "pairs/s", "seconds": A1["elapsed"], "stages_s": A1["stage"],
                                                    "what": "one whole job, its 144 part files written to an empty tmpfs and kept until it is over (no older text to unlink beside it): what ONE run of the job sees; "
                                                            "`value` above is the steady state of steps back to back, each unlinking the one before inside the timed region"}
                cleaner.gate(0)
            except Exception as e:
                out["single_job_on_empty_tmpfs"] = {"error": repr(e)}
            # ---- steps that REPLACE the files of the step before (the same prefix, SCS_SINK_IN_PLACE: the files are overwritten where they
            # lie and cut to their new length): what a loop of jobs costs when it does not have to give 197 GB of pages back and take
            # them again.  Two untimed steps: the first makes the files, the second is the first to write over them -- and takes twice as
            # long as every later one (the kernel moves every page it is written to a second time to the active list); steady state from the third.
            try:
                P = timed(2, 3, "files_in_place", 260)
                out["files_overwritten_in_place"] = {"value": P["pairs"] / P["elapsed"], "unit": "pairs/s", "steps": 3, "ms_per_step": 1e3 * P["elapsed"] / 3, "stages_s_per_step": P["stage"],
                                                     "per_step_s": P["per_step"],
                                                     "what": "the same job, every step (fresh seed) writing over the part files of the step before: scs_yield_reads_files_ex(..., SCS_SINK_IN_PLACE) / "
                                                             "`scssim genreads --in-place`.  Same bytes generated, copied and left in the files as in `value`'s steps; no unlink, no page allocation (two untimed steps first: "
                                                             "making the files, and the first pass over them, which costs 8 s -- profiles/r04_bench_inplace.json.log)"}
                base = os.path.join(out_dir, "inplace")
                cleaner.add([f for m in scssim_amd.part_paths(base, writers * generations, True, ".fq") for f in m] + [base + ".parts"])
                cleaner.gate(0)
            except Exception as e:
                out["files_overwritten_in_place"] = {"error": repr(e)}
            # ---- BGZF: the same job with the text compressed on the GPU before it crosses PCIe (<...>.fq.gz parts); an extension (the
            # reference writes plain text), so a leg of its own
            try:
                Z = timed(1, 2, "bgzf", 300)
                cleaner.gate(0)
                zs = g.stats()
                out["bgzf"] = {"value": Z["pairs"] / Z["elapsed"], "unit": "pairs/s", "steps": 2, "ms_per_step": 1e3 * Z["elapsed"] / 2, "stages_s_per_step": Z["stage"],
                               "compression": sum(zs["fastq_bytes"]) / max(1, sum(zs["sink_bytes"])), "bytes_per_step_in_files": sum(zs["sink_bytes"]),
                               "what": "the same job with --bgzf: every batch's text made into BGZF blocks by two kernels where it lies in HBM (dynamic-Huffman deflate of literals, CRC-32), D2H and %d x %d .fq.gz part files per mate on tmpfs" % (writers, generations)}
            except Exception as e:
                out["bgzf"] = {"error": repr(e)}
            # ---- the sweep north_star asks for (1 Mb -> 3 Gb) in one record: configs[1] and configs[2]'s sizes, same model and options
            try:
                out["sweep"] = [small_config(torch, scssim_amd, dev, stream, prof, 1.0, a.coverage, "configs[1]: 1 Mb x 2 haplotypes, PE150 %gx" % a.coverage, shm, concurrent=8),
                                small_config(torch, scssim_amd, dev, stream, prof, 63.02552, a.coverage, "configs[2] size: 63 Mb (chr20) x 2 haplotypes, PE150 %gx" % a.coverage, shm)]
                # what the DROP-IN default (the reference's two files, `scssim genreads` without --writers) does: bound by the inode lock of a
                # tmpfs file (5.3-5.7 GB/s of fresh pages per file, whatever the size of the job), measured here at chr20 size
                big = out["sweep"][1]
                out["two_file_layout"] = {"value": big["with_two_fastq_files_on_tmpfs_pairs_per_s"], "unit": "pairs/s", "measured_on": big["workload"],
                                          "what": "the same library writing the reference's <prefix>_1.fq / _2.fq (writers = 1: the CLI's default).  `value` above uses an EXTENSION -- "
                                                  "%d part files per mate written by %d threads in %d generations -- because fresh pages of one tmpfs file come at 5.5 GB/s whatever the "
                                                  "thread count; the two-file rate does not depend on the job's size, so the 3 Gb job with two files takes about %d s"
                                                  % (writers * generations, writers, generations, round(R["fq_bytes"] / a.steps / 2 / 5.5e9))}
            except Exception as e:                                   # an extra leg must not cost the record
                out["sweep"] = {"error": repr(e)}
            # ---- CLI wall at chr20 size
            cli = os.path.join(ROOT, "scssim_amd", "bin", "scssim")
            if cli_sample is not None and os.path.exists(cli):
                cd = tempfile.mkdtemp(prefix="scsbench_cli_", dir=shm)
                try:
                    fa = os.path.join(cd, "chr20.fa")
                    n20 = len(cli_sample)
                    write_simu_fasta(fa, ["20_1_%d" % n20, "20_2_%d" % n20], [cli_sample, cli_sample])
                    t1 = time.perf_counter()
                    r = subprocess.run([cli, "genreads", "-i", fa, "-m", prof, "-c", "%g" % a.coverage, "-o", os.path.join(cd, "reads"), "--seed", "5", "--device", str(local)],
                                       capture_output=True, text=True)
                    dt = time.perf_counter() - t1
                    m = re.search(r"pairs (\d+);", r.stderr)
                    if r.returncode == 0 and m:
                        out["cli_wall"] = {"value": int(m.group(1)) / dt, "unit": "pairs/s", "seconds": dt, "pairs": int(m.group(1)),
                                           "what": "`scssim genreads` end to end on a chr20-size record (%.1f Mb x 2 haplotypes; configs[2]): process start, FASTA parse + .fai, profile, job, both FASTQ files on tmpfs" % (n20 / 1e6)}
                    else:
                        out["cli_wall"] = {"error": r.stderr[-300:]}
                finally:
                    shutil.rmtree(cd, ignore_errors=True)
        if world == 1 and not a.no_cpu_baseline and cpu_sample is not None:
            n = len(cpu_sample)
            out["cpu_baseline"] = cpu_baseline(td, prof, ["1_1_%d" % n, "1_2_%d" % n], [cpu_sample, cpu_sample],
                                               "first %.1f Mb of chromosome 1 of the bench genome (x 2 haplotypes), PE150 %gx, same profile and options" % (n / 1e6, a.coverage), a.coverage)
        print(json.dumps(out))
    cleaner.close()
    g.close()
    shutil.rmtree(td, ignore_errors=True)
    if not a.out_dir:
        shutil.rmtree(out_dir, ignore_errors=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
