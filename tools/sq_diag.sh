#!/bin/bash
# Diagnostic SQ / LDS counters of the reads kernels on a reduced genome: tools/sq_diag.sh <tag> [bench.py arguments]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/diag_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --hbm-only --no-extra-legs --no-cpu-baseline --genome-mb 300 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p0 -o t -- python3 $ROOT/bench.py $ARGS > $OUT/p0.log 2>&1
cp $(find $OUT/p0 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/p0
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o a -- python3 $ROOT/bench.py $ARGS > $OUT/p1.log 2>&1
echo "p1 done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU --output-format csv -d $OUT/p2 -o b -- python3 $ROOT/bench.py $ARGS > $OUT/p2.log 2>&1
echo "p2 done"
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
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
        o.write(k + "\n")
        for c, x in sorted(v.items()):
            o.write("   %-24s %.4g total  %.4g per launch (%d launches)\n" % (c, x, x / len(n[(k, c)]), len(n[(k, c)])))
PY
rm -rf $OUT/p1 $OUT/p2
head -8 $OUT/kernel_stats.csv | cut -c1-120; head -32 $OUT/summary.txt
