#!/bin/bash
# Instruction-cache counters of the reads kernels on a reduced genome: tools/probes/ic_diag.sh <tag>
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-extra-legs --no-cpu-baseline --genome-mb 300 $*"
timeout -k 10 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o a -- python3 $ROOT/bench.py $ARGS > $OUT/p1.log 2>&1 || echo "p1 failed"
echo "p1 done"
python3 - $OUT <<'PY'
import csv, sys, glob, collections, re
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?(scs::\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1); tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
with open(out + "/summary.txt", "w") as o:
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:6]:
        o.write(k + "\n")
        for c, x in sorted(v.items()):
            o.write("   %-36s %.4g per launch (%d launches)\n" % (c, x / len(n[(k, c)]), len(n[(k, c)])))
PY
rm -rf $OUT/p1
cat $OUT/summary.txt
