#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass with SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU into per-kernel,
per-launch figures: VALU wave-instructions, active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU),
SALU instructions.  sq_summary.py <counter_collection.csv> <out.csv> "<command>".  bench.py reads the file (roofline.valu)."""
import collections
import csv
import re
import sys


def short(name):
    m = re.match(r"(?:void )?(scs::\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def main():
    path, out, cmd = sys.argv[1:4]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
    rows = sorted(((k, v, len(disp[k])) for k, v in tot.items()), key=lambda r: -r[1].get("SQ_INSTS_VALU", 0))
    with open(out, "w") as o:
        o.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU, %s; per launch\n" % cmd)
        wr = csv.writer(o, lineterminator="\n")                       # (kernel names hold commas: quoted)
        wr.writerow(["kernel", "valu_insts_per_launch", "lanes_per_valu_inst", "salu_insts_per_launch"])
        for k, v, n in rows:
            if not k.startswith("scs::"):
                continue
            act = v.get("SQ_ACTIVE_INST_VALU", 0.0)
            wr.writerow([k, "%.0f" % (v.get("SQ_INSTS_VALU", 0.0) / n), "%.1f" % (v.get("SQ_THREAD_CYCLES_VALU", 0.0) / act if act else 0.0), "%.0f" % (v.get("SQ_INSTS_SALU", 0.0) / n)])


if __name__ == "__main__":
    main()
