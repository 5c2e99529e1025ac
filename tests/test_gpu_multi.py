"""Tests that need TWO OR MORE MI355X in one box: the sharded job over a real RCCL communicator (ncclCommInitRank with 2+ ranks, the
per-pass 65 536-counter all-reduce on the ctx stream, the allocation partials, scs_comm_abort beside a blocked collective).  On a
one-GPU box they are collected and skipped; the same control flow runs there through the host-memory seam (test_gpu_sink.py) and gloo
(test_gpu_parity.py::test_bench_two_rank_control_flow).  Replaces the pool fan-out of lib/malbac/Malbac.cpp:318-368,438-454."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _gpus():
    import torch
    return torch.cuda.device_count()                                  # (counting devices does not initialise the GPU)


need2 = pytest.mark.skipif(_gpus() < 2, reason="needs at least two GPUs (RCCL refuses two ranks on one device)")


def _oracle(oracle_bin, fasta, profile, prefix, args, seed, threads=8):
    subprocess.check_call([oracle_bin, "genreads", "-i", fasta, "-m", profile, "-o", prefix, "--rng", "counter", "--seed", str(seed), "-t", str(threads), "-q"] + args)


def _cli(args, env=None, timeout=300):
    from test_gpu_sink import _cli as run
    return run(args, env=env, timeout=timeout)


@need2
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_cli_sharded_job_over_rccl_equals_the_oracle(ranks, oracle_bin, models, golden_inputs, tmp_path):
    """`scssim genreads --gpus N`: one process per GPU forked before the first HIP call, rank 0's RCCL id down the pipes,
    ncclCommInitRank inside the library, every exchange of the sharded path through RCCL on the ctx streams, a shard per rank, the
    native merge -- and the files are the oracle's (= the unsharded job's) byte for byte; RCCL itself reports N ranks."""
    if _gpus() < ranks:
        pytest.skip("needs %d GPUs" % ranks)
    fa, prof = golden_inputs["g2_xten_pe_nblock"], models["Illumina_HiSeqXTen"]
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, fa, prof, prefix, ["-c", "3"], 31)
    out = str(tmp_path / "cli")
    rc, err, left = _cli(["-i", fa, "-m", prof, "-c", "3", "-o", out, "--seed", "31", "--gpus", str(ranks)])
    assert rc == 0 and not left, err
    assert ("RCCL communicator of %d ranks" % ranks) in err             # scs_comm_count == N (ncclCommCount)
    for s in ("_1.fq", "_2.fq"):
        assert open(out + s, "rb").read() == open(prefix + s, "rb").read(), s


@need2
def test_cli_sharded_job_over_rccl_with_primer_types_running_dry(oracle_bin, models, repeat_genome, tmp_path):
    """The exhaustion regime over RCCL: the pass in which primer types run dry is run again segment by segment, every segment by its
    owner against the stock the segments before it left (an all-reduce to which only the owner contributes)."""
    prof = models["Illumina_HiSeq2500"]
    prefix = str(tmp_path / "orc")
    _oracle(oracle_bin, repeat_genome, prof, prefix, ["-c", "0.5", "-p", "10000", "-r", "1e-8"], 53)
    out = str(tmp_path / "cli")
    rc, err, left = _cli(["-i", repeat_genome, "-m", prof, "-c", "0.5", "-p", "10000", "-r", "1e-8", "-o", out, "--seed", "53", "--gpus", "2"])
    assert rc == 0 and not left, err
    for s in ("_1.fq", "_2.fq"):
        assert open(out + s, "rb").read() == open(prefix + s, "rb").read(), s


@need2
@pytest.mark.parametrize("where,who", [("amplify", 1), ("amplify", 0)])
def test_rank_killed_before_amplify_ends_the_rccl_job(where, who, models, golden_inputs, tmp_path):
    """A rank that dies before `amplify` leaves its sibling inside an RCCL all-reduce that never completes.  Rank 0's watcher ends the
    others and calls scs_comm_abort (ncclCommAbort beside the blocked collective); the job ends with status 2 (3 when rank 0 itself is
    the one that fails) within seconds of the failure, nothing left in the process group."""
    fa, prof = golden_inputs["g1_hiseq2500_pe"], models["Illumina_HiSeq2500"]
    base = ["-i", fa, "-m", prof, "-c", "2", "-o", str(tmp_path / "f"), "--seed", "3", "--gpus", "2"]
    t0 = time.time()
    rc, err, left = _cli(base, timeout=300)                           # the healthy job: how long start-up + job take on this box
    healthy = time.time() - t0
    assert rc == 0, err
    t0 = time.time()
    rc, err, left = _cli(base, env={"SCS_TEST_FAIL_AT": where, "SCS_TEST_FAIL_RANK": str(who)}, timeout=120)
    took = time.time() - t0
    assert rc in (2, 3), (rc, err)
    assert took < healthy + 5.0, "the failing job took %.1f s, the healthy one %.1f s" % (took, healthy)
    assert not left, "ranks left behind: %s" % left
    assert ("rank %d of the sharded job failed" % who) in err or "test failure injected" in err


@need2
def test_bench_two_gpus_over_rccl(tmp_path):
    """`python bench.py --gpus 2` as the driver runs it for N = 1 -- no launcher on the command line: bench.py starts its ranks itself,
    one per GPU, the communicator is RCCL's (config.collectives names ncclCommCount = 2) and rank 0 prints the one JSON line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SCS_BENCH_BACKEND", "SCS_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-mb", "60"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "ncclCommCount = 2" in d["config"]["collectives"]
    assert [p["rank"] for p in d["per_rank"]] == [0, 1] and sum(p["pairs"] for p in d["per_rank"]) == d["config"]["pairs_per_step"]
