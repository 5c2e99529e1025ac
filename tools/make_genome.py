#!/usr/bin/env python3
"""Synthetic genome generator.

Writes (a) a plain reference FASTA (60 columns, records chr<k>) and/or (b) the
`simuvars`-style diploid FASTA that `scssim genreads -i` consumes: two
identical haplotype records per chromosome named <chr>_<hap>_<reflen>, 100
columns (format: reference lib/genome/Genome.cpp:365-381).

Bases are i.i.d. with P(A,C,G,T) = (0.3,0.2,0.2,0.3); optional leading N block
and lower-case fraction exercise N handling and upper-casing.
"""
import argparse
import numpy as np


def synth_record(rng, length, n_block=0, lower_frac=0.0):
    codes = rng.choice(4, size=length, p=[0.3, 0.2, 0.2, 0.3]).astype(np.uint8)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
    if n_block > 0:
        seq[: min(n_block, length)] = ord("N")
    if lower_frac > 0:
        m = rng.random(length) < lower_frac
        seq[m] |= 0x20
    return seq


def write_fasta(path, records, width):
    with open(path, "wb") as f:
        for name, seq in records:
            f.write(b">" + name.encode() + b"\n")
            n = len(seq)
            full = (n // width) * width
            if full:
                body = seq[:full].reshape(-1, width)
                nl = np.full((body.shape[0], 1), 10, dtype=np.uint8)
                f.write(np.concatenate([body, nl], axis=1).tobytes())
            if n > full:
                f.write(seq[full:].tobytes() + b"\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lengths", required=True,
                    help="comma separated record lengths, e.g. 1000000 or 300000,200000")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--n-block", type=int, default=0)
    ap.add_argument("--lower-frac", type=float, default=0.0)
    ap.add_argument("--first-chr", type=int, default=20)
    ap.add_argument("--ref-out", default=None, help="plain reference FASTA (60 col)")
    ap.add_argument("--simu-out", default=None, help="simuvars-style diploid FASTA (100 col)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    lens = [int(x) for x in a.lengths.split(",")]
    recs = []
    for i, L in enumerate(lens):
        recs.append((str(a.first_chr + i), synth_record(rng, L, a.n_block, a.lower_frac)))
    if a.ref_out:
        write_fasta(a.ref_out, [("chr" + n, s) for n, s in recs], 60)
    if a.simu_out:
        out = []
        for n, s in recs:
            up = s & 0xDF  # simuvars upper-cases (Genome.cpp:326)
            for hap in (1, 2):
                out.append(("%s_%d_%d" % (n, hap, len(s)), up))
        write_fasta(a.simu_out, out, 100)


if __name__ == "__main__":
    main()
