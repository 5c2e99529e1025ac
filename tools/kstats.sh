#!/bin/bash
# Kernel durations of one reduced bench job: tools/kstats.sh <tag> [bench.py arguments]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-extra-legs --no-cpu-baseline --genome-mb 300 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p0 -o t -- python3 $ROOT/bench.py $ARGS > $OUT/p0.log 2>&1
cp $(find $OUT/p0 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/p0
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"].split("(")[0][-60:]))
rows.sort(reverse=True)
for r in rows[:28]:
    print("%8.3f ms %5s calls %9.1f us  %s" % r)
PY
