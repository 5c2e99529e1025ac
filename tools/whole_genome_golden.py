#!/usr/bin/env python3
"""Makes tests/golden/whole_genome_config3.json: BASELINE configs[3] at FULL size through the CPU oracle (counter mode), as
per-batch checksums of the FASTQ text -- what tests/test_gpu_fullsize.py::test_whole_genome_properties compares the HIP path's
device-side batch checksums with (scs_set_batch_checksums), so that the 3.1 Gb job is pinned bit for bit against the oracle and
not only against a second run of itself: 1.25e9 amplicons = 1.25e6 allocation chunks (the third level of the chunk CDF,
MyDefine.cpp:203-253), amplicon indices above 1e9 in the record names (Amplicon.cpp:460,498,519).

Run on a GPU box (the genome is the bench's: torch's device generator, seed 3000; the oracle needs ~150 GB of host memory and
~15 minutes on 16 cores):

    gpurun --timeout 1200 -- 'python tools/whole_genome_golden.py --out gpurun_out/whole_genome_config3.json > gpurun_out/wg_golden.log 2>&1'

then copy the JSON to tests/golden/.  --batches: which 8 M-pair batches to sum (every batch's text depends on the whole job's
amplicon tables and read allocation; the default takes the first two, a middle one and the last two -- the oracle makes a batch
in ~20 s, all 37 would not fit one gpurun call).  --genome-mb scales the genome down for rehearsals."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", type=int, default=3, help="3: BASELINE configs[3] (PE150 30x, plain diploid genome); 4: configs[4] (PE250 60x -s 500, the haplotypes made by simuvars from a CNV-heavy variation file)")
    ap.add_argument("--genome-mb", type=float, default=0.0)
    ap.add_argument("--batches", default="0,1,18,-2,-1")
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    dev = torch.device("cuda", 0)
    lens = bench.record_lengths(a.genome_mb)
    t0 = time.time()
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    td = tempfile.mkdtemp(prefix="scs_wg_", dir=shm)
    fa = os.path.join(td, "simu.fa")
    oracle = os.path.join(ROOT, "oracle", "_build", "scs_oracle")
    if a.config == 4:
        return config4(a, torch, bench, dev, lens, td, fa, oracle, t0)
    names, rl, bases = bench.synth_genome(torch, dev, lens, 3000)
    fp = bench.genome_fingerprint(torch, bases)
    try:
        with open(fa, "wb") as f:
            off = 0
            for name, n in zip(names, rl):
                rec = bases[off:off + n].cpu().numpy(); off += n
                f.write(b">" + name.encode() + b"\n")
                full = (n // 100) * 100
                if full:
                    f.write(np.concatenate([rec[:full].reshape(-1, 100), np.full((full // 100, 1), 10, np.uint8)], axis=1).tobytes())
                if n > full:
                    f.write(rec[full:].tobytes() + b"\n")
                del rec
        del bases
        torch.cuda.empty_cache()
        print("genome in %s: %.1f s, fingerprint %016x" % (fa, time.time() - t0, fp), flush=True)
        prof = bench.make_profile(td)
        cks = os.path.join(td, "cks.tsv")
        threads = a.threads or bench.host_cores()
        cmd = [os.path.join(ROOT, "oracle", "_build", "scs_oracle"), "genreads", "-i", fa, "-m", prof, "--rng", "counter", "--seed", str(a.seed), "-t", str(threads),
               "-c", "30", "-s", "260", "--checksums", cks, "--batch-pairs", str(1 << 23)]
        if a.batches != "all":
            cmd += ["--checksum-batches", a.batches]
        t1 = time.time()
        subprocess.check_call(cmd)
        head, rows = None, []
        for line in open(cks):
            if line.startswith("#"):
                f = line.split(); head = {f[i]: int(f[i + 1]) for i in range(1, len(f) - 1, 2)}
            else:
                f = line.split(); rows.append([int(f[0]), f[1], f[2], int(f[3]), int(f[4]), int(f[5])])
        out = dict(what="BASELINE configs[3] through oracle/scs_oracle --rng counter: per-batch checksums of the FASTQ text (tools/whole_genome_golden.py)",
                   genome=dict(records="bench.record_lengths(%g)" % a.genome_mb, bases_per_haplotype=sum(lens), generator="bench.synth_genome(torch, cuda:0, lens, 3000)", fingerprint="%016x" % fp),
                   job=dict(profile="bench.make_profile (HiSeq2500 resampled to 150 bins)", coverage=30, isize=260, seed=a.seed, primers=100000, gamma=1e-9, batch_pairs=1 << 23),
                   counts=head, batches=rows, batch_columns=["batch", "checksum mate 1", "checksum mate 2", "bytes mate 1", "bytes mate 2", "pairs"],
                   made=dict(oracle_threads=threads, oracle_seconds=round(time.time() - t1, 1), torch=torch.__version__))
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote %s: %d batches, oracle %.0f s" % (a.out, len(rows), time.time() - t1), flush=True)
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)


def parse_cks(path):
    head, rows = None, []
    for line in open(path):
        f = line.split()
        if line.startswith("#"):
            head = {f[i]: int(f[i + 1]) for i in range(1, len(f) - 1, 2)}
        else:
            rows.append([int(f[0]), f[1], f[2], int(f[3]), int(f[4]), int(f[5])])
    return head, rows


def config4(a, torch, bench, dev, lens, td, fa, oracle, t0):
    """configs[4]: reference + variation file -> `scs_oracle simuvars` -> the haplotypes' FASTA -> `scs_oracle genreads` (PE250 = HiSeq X Ten
    resampled to 250 bins, 60x, -s 500), the test's inputs and options to the letter (tests/test_gpu_fullsize.py)."""
    import gzip
    import shutil
    try:
        ref, var = os.path.join(td, "ref.fa"), os.path.join(td, "vars.txt")
        expect, fp = bench.make_config4_inputs(torch, dev, lens, ref, var)
        torch.cuda.empty_cache()
        print("reference + variation file: %.1f s, fingerprint %016x, %d haplotype bases expected" % (time.time() - t0, fp, expect), flush=True)
        t1 = time.time()
        subprocess.check_call([oracle, "simuvars", "-r", ref, "-v", var, "-o", fa])
        os.remove(ref)
        print("oracle simuvars: %.1f s, %.2f GB" % (time.time() - t1, os.path.getsize(fa) / 1e9), flush=True)
        src = os.path.join(td, "xten.profile")
        open(src, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "models", "Illumina_HiSeqXTen.profile.gz")).read())
        prof = os.path.join(td, "pe250.profile")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profile.py"), src, prof, "--read-length", "250"])
        cks = a.out + ".cks"                                                             # (under gpurun_out/: what was made survives a time limit)
        threads = a.threads or bench.host_cores()
        cmd = [oracle, "genreads", "-i", fa, "-m", prof, "--rng", "counter", "--seed", str(a.seed), "-t", str(threads), "-c", "60", "-s", "500",
               "--checksums", cks, "--batch-pairs", str(1 << 23)]
        if a.batches != "all":
            cmd += ["--checksum-batches", a.batches]
        t2 = time.time()
        subprocess.check_call(cmd)
        head, rows = parse_cks(cks)
        out = dict(what="BASELINE configs[4] through oracle/scs_oracle (simuvars, then genreads --rng counter): per-batch checksums of the FASTQ text (tools/whole_genome_golden.py --config 4)",
                   genome=dict(records="bench.record_lengths(%g)" % a.genome_mb, bases_per_haplotype_of_the_reference=sum(lens), generator="bench.make_config4_inputs(torch, cuda:0, lens, ref, var)",
                               fingerprint="%016x" % fp, haplotype_bases=expect),
                   job=dict(profile="tools/make_profile.py Illumina_HiSeqXTen --read-length 250", coverage=60, isize=500, seed=a.seed, primers=100000, gamma=1e-9, batch_pairs=1 << 23),
                   counts=head, batches=rows, batch_columns=["batch", "checksum mate 1", "checksum mate 2", "bytes mate 1", "bytes mate 2", "pairs"],
                   made=dict(oracle_threads=threads, oracle_seconds=round(time.time() - t2, 1), torch=torch.__version__))
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote %s: %d batches, oracle genreads %.0f s" % (a.out, len(rows), time.time() - t2), flush=True)
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    main()
