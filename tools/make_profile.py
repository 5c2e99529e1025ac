#!/usr/bin/env python3
"""Synthesise a .profile with a different read length from a shipped one.

The reference takes the read length only from the profile (lib/profile/Profile.cpp:976-999) and ships
no PE100/PE150/PE250 model, so BASELINE configs that ask for them use a profile whose per-bin rows
(substitution and quality tables) are the source profile's rows resampled along the bin axis
(new bin j <- old bin floor(j * old / new)); rates, length frequencies, insert-size sigma and the GC
rows are kept.  Output is in the reference's own format (Profile::saveResults, Profile.cpp:1236-1361).
"""
import argparse
import gzip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--read-length", type=int, required=True)
    ap.add_argument("--noise", type=float, default=0.0, help="tests: mix every substitution row with the uniform row, p' = (1 - F) p + F / 4 (a substitution rate of about 3 F / 4)")
    a = ap.parse_args()
    op = gzip.open if a.src.endswith(".gz") else open
    lines = op(a.src, "rt").read().split("\n")
    old = None
    for ln in lines:
        if ln.startswith("readLength:"):
            old = int(ln.split(":")[1])
    new = a.read_length
    pick = [min(old - 1, j * old // new) for j in range(new)]
    out, i = [], 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("readLength:"):
            out.append("readLength: %d" % new)
        elif ln.startswith("binCount:"):
            out.append("binCount: %d" % new)
        elif ln.startswith("kmer:") and not ln.startswith("kmer: 3") or (ln.startswith("kmer: ") and len(ln.split(":")[1].strip()) == 3 and not ln.split(":")[1].strip().isdigit()):
            out.append(ln)                                  # "kmer: XXA" block: 2*old rows (read 1, read 2)
            rows = lines[i + 1:i + 1 + 2 * old]
            if a.noise > 0:
                rows = ["\t".join("%.9g" % ((1 - a.noise) * float(v) + a.noise / 4) for v in r.split("\t")) for r in rows]
            out += [rows[j] for j in pick] + [rows[old + j] for j in pick]
            i += 2 * old
        elif ln.startswith("basePairIndx:"):
            out.append(ln)
            rows = lines[i + 1:i + 1 + old]
            out += [rows[j] for j in pick]
            i += old
        else:
            out.append(ln)
        i += 1
    open(a.dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
