#!/usr/bin/env python3
"""Kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from the rocpd SQLite file that rocprofv3 of
ROCm 7.2 writes by default: name, calls, total / average / min / max duration in ns, share.  Usage: rocpd_stats.py <results.db> [out.csv]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for name, n, t, avg, mn, mx in rows:
        out.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (short(name), n, t, avg, 100.0 * t / tot, mn, mx))


if __name__ == "__main__":
    main()
