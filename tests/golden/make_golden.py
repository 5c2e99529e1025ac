#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Runs only in the build container (needs /root/reference and `make -C oracle ref`).
For each case it writes
    <case>.simu.fa.gz       input FASTA (simuvars format)  -- data
    <case>.simu.fa.fai      the index the reference writes beside its input -- data
    <case>.ref_1.fq.gz ...  FASTQ written by oracle/_ref/scssim_ref at -t 1 under
                            oracle/_ref/libseedshim.so (SCS_FIXED_TIME pinned) -- data
and copies the profile *data files* the reference ships (testData/models) as
models/*.profile.gz.  Nothing here is reference source code.

The reference has no tests or golden vectors of its own, so
these reference-generated outputs are what pins the oracle: tests/test_oracle_golden.py
requires `scs_oracle --rng ref` to reproduce every FASTQ byte-for-byte.
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref", "scssim_ref")
SHIM = os.path.join(ROOT, "oracle", "_ref", "libseedshim.so")
MODELS = "/root/reference/testData/models"

CASES = [
    # name, genome args, profile, extra genreads args, fixed time
    dict(name="g1_hiseq2500_pe", genome=["--lengths", "120000", "--seed", "11"],
         profile="Illumina_HiSeq2500", args=["-c", "1"], time=1234567890),
    dict(name="g2_xten_pe_nblock", genome=["--lengths", "70000,50000", "--seed", "12", "--n-block", "3000", "--lower-frac", "0.06"],
         profile="Illumina_HiSeqXTen", args=["-c", "1.5", "-s", "300"], time=1500000000),
    dict(name="g3_hiseq2000_se", genome=["--lengths", "90000", "--seed", "13"],
         profile="Illumina_HiSeq2000", args=["-c", "0.6", "-l", "SE"], time=1600000001),
    dict(name="g4_gaiix_pe_lowprimers", genome=["--lengths", "80000", "--seed", "14"],
         profile="Illumina_GenomeAnalyzerIIx", args=["-c", "0.8", "-p", "1000", "-r", "1e-8"], time=1700000002),
]


def gz_write(path, data):
    with open(path, "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, compresslevel=9) as g:
            g.write(data)


def main():
    if not os.path.exists(REFBIN):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(os.path.join(HERE, "models"), exist_ok=True)
    for p in sorted(os.listdir(MODELS)):
        if p.endswith(".profile"):
            gz_write(os.path.join(HERE, "models", p + ".gz"), open(os.path.join(MODELS, p), "rb").read())
    manifest = {}
    for c in CASES:
        with tempfile.TemporaryDirectory() as td:
            fa = os.path.join(td, "simu.fa")
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_genome.py"),
                                   "--simu-out", fa] + c["genome"])
            # MALLOC_PERTURB_=255 makes glibc zero-fill malloc'd memory.  The reference reads an
            # UNINITIALISED `long count` for primer 8-mers that contain N (PrimerIndex(), lib/malbac/
            # Malbac.h:18-24, read at Malbac.cpp:96); zero-fill pins that read to 0 = "no stock",
            # which is the interpretation the oracle implements.  Cases without N are unaffected
            # (their hashes are identical with and without the variable).
            env = dict(os.environ, LD_PRELOAD=SHIM, SCS_FIXED_TIME=str(c["time"]), MALLOC_PERTURB_="255")
            cmd = [REFBIN, "genreads", "-i", fa, "-m", os.path.join(MODELS, c["profile"] + ".profile"),
                   "-t", "1", "-o", os.path.join(td, "ref")] + c["args"]
            subprocess.check_call(cmd, env=env, stderr=subprocess.DEVNULL)
            gz_write(os.path.join(HERE, c["name"] + ".simu.fa.gz"), open(fa, "rb").read())
            # the FASTA index the reference leaves beside its input (fastahack, lib/fastahack/Fasta.cpp:241-249) -- data
            shutil.copyfile(fa + ".fai", os.path.join(HERE, c["name"] + ".simu.fa.fai"))
            entry = dict(profile=c["profile"], args=c["args"], fixed_time=c["time"], files={})
            for suffix in ("_1.fq", "_2.fq", ".fq"):
                src = os.path.join(td, "ref" + suffix)
                if os.path.exists(src):
                    data = open(src, "rb").read()
                    gz_write(os.path.join(HERE, c["name"] + ".ref" + suffix + ".gz"), data)
                    entry["files"][suffix] = dict(sha256=hashlib.sha256(data).hexdigest(),
                                                  records=data.count(b"\n") // 4)
            manifest[c["name"]] = entry
            print(c["name"], entry["files"])
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
