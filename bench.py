#!/usr/bin/env python3
"""bench.py -- genreads hot path on N MI355X (one process per GPU).

Metric (BASELINE.json): paired-end read pairs/s, whole job, inputs resident in HBM, plus the HBM
roofline fraction of the dominant kernel and the CPU path timed on this box's host cores.

Workload at N=1 = BASELINE configs[1]: 1 Mb synthetic reference (i.i.d. 30/20/20/30 % A/C/G/T, seed 1,
diploid simuvars-style FASTA), PE150 at 30x, HiSeq2500 model.  The reference ships no 150 bp model and
takes the read length from the profile only, so "PE150" = the shipped HiSeq2500 profile with its bin
axis resampled 125 -> 150 (tools/make_profile.py).  One step = one complete job
(fragment split, 1+5 MALBAC cycles, read allocation, fragment sampling, error/quality injection,
FASTQ formatting into an HBM pool) with a fresh seed.  N>1: weak scaling, each rank runs the same
workload on its own 1 Mb record (fragment-lineage shard = one record per rank, no data-path
collective), then the FASTQ pools are gathered on the writer rank over RCCL.
"""
import argparse
import gzip
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_inputs(td, rank, n_records=1):
    """n_records x 1 Mb records (weak scaling: one record's worth of fragments per rank); every rank builds the same file."""
    fa = os.path.join(td, "simu_%d.fa" % rank)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", ",".join(["1000000"] * n_records),
                           "--seed", "1", "--first-chr", "20", "--simu-out", fa])
    src = os.path.join(td, "hiseq2500_%d.profile" % rank)
    open(src, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "models", "Illumina_HiSeq2500.profile.gz")).read())
    prof = os.path.join(td, "pe150_%d.profile" % rank)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profile.py"), src, prof, "--read-length", "150"])
    return fa, prof


def cpu_baseline(fa, prof, seed):
    """The oracle (CPU restatement, counter mode, thread pool) on this box's host cores: the same
    workload, one step.  Checker code timed as a baseline -- never the thing shipped."""
    oracle = os.path.join(ROOT, "oracle", "_build", "scs_oracle")
    if not os.path.exists(oracle):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    r = subprocess.run([oracle, "genreads", "-i", fa, "-m", prof, "-c", "30", "-t", str(cores), "-o", fa + ".cpu",
                        "--rng", "counter", "--seed", str(seed)], capture_output=True, text=True, check=True)
    m = re.search(r"pairs=(\d+) \| load ([\d.]+)s frag ([\d.]+)s amplify ([\d.]+)s alloc ([\d.]+)s readgen ([\d.]+)s", r.stderr)
    pairs = int(m.group(1))
    secs = sum(float(m.group(i)) for i in (3, 4, 5, 6))
    for suf in ("_1.fq", "_2.fq"):
        try:
            os.remove(fa + ".cpu" + suf)
        except OSError:
            pass
    return dict(value=pairs / secs, unit="pairs/s", cores=cores, kind="port",
                sample="the full N=1 workload, 1 step (%d pairs in %.2f s: amplify %.2f, allocate %.2f, readgen %.2f)" %
                       (pairs, secs, float(m.group(4)), float(m.group(5)), float(m.group(6))))


def pmc_traffic(members):
    """HBM bytes per launch of the dominant kernel (group) from the committed rocprofv3 PMC summary of this same command
    (FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes; KB -> bytes; tools/pmc_summary.py).  bench.py cannot
    collect PMC counters itself, so the number is read from profiles/; None if the summary is absent.  For the
    amplification group one launch = one pass: bytes of all member kernels / passes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1_pmc_hbm.csv")))
    if not files:
        return None, None
    prefix = {"k_attach<semi>": "scs::k_attach<false", "k_attach<frag>": "scs::k_attach<true", "k_errs<semi->full>": "scs::k_errs<false>", "k_errs<frag->semi>": "scs::k_errs<true>",
              "k_reads": "scs::k_reads", "k_indels": "scs::k_indels"}
    tot, launches0 = 0.0, 0
    for line in open(files[-1]):
        if line.startswith("#") or line.startswith("kernel"):
            continue
        name, fetch, write, n = line.rstrip("\n").rsplit(",", 3)
        for i, m in enumerate(members):
            if name.startswith(prefix[m]):
                # gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM); byte/dword gathers
                # as here are uncalibrated, so the raw value is reported and the x2 bound is in DESIGN.md
                tot += (float(fetch) + float(write)) * 1024.0 * int(n)
                if m.startswith("k_attach") or len(members) == 1:
                    launches0 += int(n)
    return (tot / launches0 if launches0 else None), os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-hooks", action="store_true", help="install the collective hooks even at N=1 (measures their cost)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import scssim_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..." % (a.gpus, a.gpus))
    # rehearsal of the N > 1 control flow on a box with ONE GPU: SCS_BENCH_BACKEND=gloo SCS_BENCH_ONE_DEVICE=1 (all ranks on
    # cuda:0, collectives and the pool transfer staged through the CPU).  Never a measurement.
    backend = os.environ.get("SCS_BENCH_BACKEND", "nccl")
    cpu_coll = backend != "nccl"
    if os.environ.get("SCS_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cpu_coll:
            dist.init_process_group(backend)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if cpu_coll else dev                     # where the bench's own collectives live

    td = tempfile.mkdtemp(prefix="scsbench_")
    fa, prof = make_inputs(td, rank, world)
    # N > 1: ONE job (world x 1 Mb genome) sharded by fragment lineage; the setPrimers totals, the primer stock and the
    # weight normalisation are exchanged through RCCL (scssim_amd/dist.py hooks on the ctx stream); FASTQ is identical to
    # the 1-GPU run of the same genome.
    stream = torch.cuda.Stream()
    g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=30.0, isize=260, layout="PE", seed=1, device=local,
                            stream=stream.cuda_stream, shard_rank=rank, shard_count=world)
    if world > 1 or a.force_hooks:
        from scssim_amd.dist import Collectives
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        coll = Collectives(stream=stream)
        g.set_collectives(coll, device_hooks=True)
    cap = 96 << 20
    # FASTQ pools are double-buffered so the gather of step i (RCCL point-to-point on its own stream) overlaps the
    # compute of step i+1; every transfer completes inside the timed region (final synchronize + barrier).
    pools = [[torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)] for _ in range(2)]
    gathered = [torch.empty(cap, dtype=torch.uint8, device=cdev) for _ in range(2 * (world - 1))] if (world > 1 and rank == 0) else []
    inflight = [[], []]
    ktimes = {}

    def step(i, record):
        b = i & 1
        with torch.cuda.stream(stream):
            for w in inflight[b]:                       # the ctx stream waits until this buffer's previous gather is done
                w.wait()
        inflight[b] = []
        g.set_seed(1000 + i)
        g.create_frags()
        g.amplify()
        if record:
            acc(g.kernel_times(), ("k_errs<semi->full>", "k_errs<frag->semi>", "k_attach<semi>", "k_attach<frag>"))
        g.allocate_reads(0)
        pool1, pool2 = pools[b]
        n1, n2, pairs = g.yield_reads_device(pool1.data_ptr(), cap, pool2.data_ptr(), cap)
        if record:
            acc(g.kernel_times(), ("k_reads", "k_indels"))
        if world > 1:                                   # read pool -> writer rank (RCCL point-to-point over xGMI)
            with torch.cuda.stream(stream):
                sizes = torch.tensor([n1, n2], dtype=torch.int64, device=cdev)
                allsz = torch.zeros(2 * world, dtype=torch.int64, device=cdev)
                dist.all_gather_into_tensor(allsz, sizes)
                ops = []
                if rank == 0:
                    hs = allsz.cpu().tolist()
                    for r in range(1, world):
                        ops.append(dist.P2POp(dist.irecv, gathered[2 * (r - 1)][:hs[2 * r]], r))
                        ops.append(dist.P2POp(dist.irecv, gathered[2 * (r - 1) + 1][:hs[2 * r + 1]], r))
                else:
                    ops.append(dist.P2POp(dist.isend, pool1[:n1].cpu() if cpu_coll else pool1[:n1], 0))
                    ops.append(dist.P2POp(dist.isend, pool2[:n2].cpu() if cpu_coll else pool2[:n2], 0))
                inflight[b] = dist.batch_isend_irecv(ops)
        return pairs, g.stats()

    def acc(kt, names):
        for k in names:
            d = ktimes.setdefault(k, dict(launches=0, ms=0.0, units=0))
            for f in ("launches", "ms", "units"):
                d[f] += kt[k][f]

    # warm-up: every kernel timed -> per-kernel breakdown and the dominant kernel.  Timed region: HIP events only around
    # that dominant kernel (each event record is a packet on the stream of this latency-bound job).
    for i in range(a.warmup):
        step(i, True)
    warm_ktimes = {k: dict(v) for k, v in ktimes.items()}
    # the amplification pass is one unit of SURVEY 8(d) (1526 B per created amplicon = attach + error scan together)
    GROUPS = {"k_attach+k_errs": ("k_attach<semi>", "k_attach<frag>", "k_errs<semi->full>", "k_errs<frag->semi>"), "k_reads": ("k_reads",), "k_indels": ("k_indels",)}
    # dominant = the kernel with the largest total time (rocprof's ranking); an amplification kernel stands for its pass
    top = max(warm_ktimes, key=lambda k: warm_ktimes[k]["ms"]) if warm_ktimes else "k_reads"
    dominant = top if top in GROUPS else "k_attach+k_errs"
    ktimes.clear()
    TIMING_EVERY = 4                                    # events around the dominant kernel on every 4th step of the timed region
    g.set_kernel_timing(list(GROUPS[dominant]), every=TIMING_EVERY)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pairs_total, fq_bytes, alg_bytes, last = 0, 0, 0, None
    for i in range(a.steps):
        p, last = step(a.warmup + i, True)
        pairs_total += p
        if i % TIMING_EVERY == 0:                      # the steps whose launches carry events
            fq_bytes += sum(last["fastq_bytes"])
        alg_bytes += last["algorithmic_bytes"]
    for q in inflight:
        for w in q:
            w.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        pt = torch.tensor([pairs_total], dtype=torch.int64, device=cdev)
        dist.all_reduce(pt)
        pairs_total = int(pt[0])

    if rank == 0:
        dom = dominant
        passes = [k for k in GROUPS[dom] if k.startswith("k_attach")] or [GROUPS[dom][0]]     # one launch of the group = one pass
        kd = dict(launches=sum(ktimes[k]["launches"] for k in passes), ms=sum(ktimes[k]["ms"] for k in GROUPS[dom]), units=0)
        if dom == "k_attach+k_errs":
            # 1526 algorithmic bytes per created amplicon (template window read once + descriptor
            # write + primer-counter RMW + error entries) x amplicons created; one launch = one pass (attach + error scan)
            kd["units"] = ktimes["k_errs<semi->full>"]["units"] + ktimes["k_errs<frag->semi>"]["units"]
            alg = 1526.0 * kd["units"]
            note = "1526 B x %d amplicons created over %d passes (k_attach + k_errs)" % (kd["units"], kd["launches"])
        else:
            # per pair: insert-size template bytes + FASTQ bytes of both records
            kd["units"] = ktimes[dom]["units"]
            alg = 261.0 * kd["units"] + fq_bytes
            note = "(261 B template + FASTQ bytes) x %d pairs over %d launches" % (kd["units"], kd["launches"])
        achieved = alg / (kd["ms"] * 1e-3) / 1e9 if kd["ms"] > 0 else 0.0
        traffic, traffic_src = pmc_traffic(GROUPS[dom])
        out = {
            "metric": "paired-end read pairs/s (whole genreads job: MALBAC amplification + read allocation + read generation)",
            "value": pairs_total / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u32 (integer draws, byte sequences; fp64 only in the GC-weight draw)", "data": "synthetic",
            "config": {"workload": "configs[1]: 1 Mb synthetic reference (diploid simuvars FASTA), PE150 30x, HiSeq2500 model resampled to 150 bins, -p 100000 -r 1e-9 -s 260",
                       "pairs_per_step_per_gpu": last["pairs_written"], "full_amplicons_per_step": last["full_amplicons"],
                       "semi_amplicons_per_step": last["semi_amplicons"], "sharding": ("one job over %d GPUs: %d x 1 Mb records sharded by fragment lineage, RCCL all-reduce of primer stock / setPrimers totals, "
                                    "all-gather of GC weights, read pool gathered on rank 0" % (world, world)) if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": note,
                         "algorithmic_bytes_per_launch": alg / max(1, kd["launches"]), "avg_launch_ms": kd["ms"] / max(1, kd["launches"]),
                         "timed_launches": kd["launches"], "timed_steps": "every %d-th step of the timed region (HIP event records cost ~6 us each)" % TIMING_EVERY},
            "kernels_ms_per_step_warmup": {k: v["ms"] / max(1, a.warmup) for k, v in warm_ktimes.items()},
            "whole_job_algorithmic_GBps": alg_bytes / elapsed / 1e9,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fa, prof, 1000 + a.warmup)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
