#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately as MI355X_MICROARCH.md prescribes)
into per-kernel HBM KB per launch:  pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> "<command>"."""
import csv
import collections
import re
import sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    disp = collections.defaultdict(float)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        disp[(row["Kernel_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
    for (k, _), v in disp.items():
        tot[k] += v
        n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def short(name):
    m = re.match(r"(?:void )?(scs::\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def main():
    fetch, write, out, cmd = sys.argv[1:5]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    rows = sorted(((short(k), f[k][0], w.get(k, (0.0, 0))[0], f[k][1]) for k in f), key=lambda r: -(r[1] + r[2]) * r[3])
    with open(out, "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), %s; values in KB per launch, raw (no gfx950 x2 read correction applied)\n" % cmd)
        wr = csv.writer(o, lineterminator="\n")                       # (kernel names hold commas: scs::k_attach<true, 64> -- quoted)
        wr.writerow(["kernel", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "launches"])
        for k, a, b, n in rows:
            wr.writerow([k, "%.1f" % a, "%.1f" % b, "%d" % n])


if __name__ == "__main__":
    main()
