#!/bin/bash
# tools/probes/tcp_rate.sh: the probe's timings, then its modes one by one under the TCP / TA / TD counters (calibration of tools/reads_mem_diag.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LOG=$ROOT/gpurun_out/tcp_rate.log
$ROOT/tools/probes/tcp_rate.bin > $LOG 2>&1
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 3 4 5 6 10 12; do
  rm -rf /tmp/tr$m
  rocprofv3 --pmc TCP_TOTAL_ACCESSES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN2_sum --output-format csv -d /tmp/tr$m -o a -- $ROOT/tools/probes/tcp_rate.bin $m > /tmp/tr$m.log 2>&1
  python3 - /tmp/tr$m $m >> $LOG <<'PY'
import csv, glob, sys, collections
t = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_probe" in r["Kernel_Name"]: t[r["Counter_Name"]] += float(r["Counter_Value"])
print("mode", sys.argv[2], "per launch:", {k: "%.4g" % (v / 3) for k, v in sorted(t.items())})
PY
done
cat $LOG
