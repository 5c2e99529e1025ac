#!/bin/bash
# Diagnostic L2 / L1 write counters of the reads kernels on a reduced genome: tools/probes/mem_diag.sh <tag>
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --hbm-only --no-extra-legs --no-cpu-baseline --genome-mb 300 $*"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_REQ_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d $OUT/p1 -o a -- python3 $ROOT/bench.py $ARGS > $OUT/p1.log 2>&1 || echo "p1 failed"
echo "p1 done"
# (no TCP_* pass: five TCP_* counters in one --pmc pass are more than the hardware collects at once -- rocprofv3 fails in
# rocprofiler_create_counter_config with error 38, "Request exceeds the capabilities of the hardware to collect", aborts with
# signal 6 while torch is in its first kernel, and the process then lingers in the tool's signal handler until the watchdog
# kills it (gpurun_out/mem_a/p2.log of round 2).  If the TCP numbers are wanted: one or two TCP_* counters per pass.)
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum --output-format csv -d $OUT/p3 -o c -- python3 $ROOT/bench.py $ARGS > $OUT/p3.log 2>&1 || echo "p3 failed"
echo "p3 done"
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
    for k, v in sorted(tot.items(), key=lambda kv: -sum(kv[1].values())):
        o.write(k + "\n")
        for c, x in sorted(v.items()):
            o.write("   %-36s %.4g per launch (%d launches)\n" % (c, x / len(n[(k, c)]), len(n[(k, c)])))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
head -40 $OUT/summary.txt
