#!/bin/bash
# k_reads' three classes as three launches one after the other (SCS_READS_SERIAL, seams build): how long does each take?  tools/reads_by_class.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/reads_by_class
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SCSSIM_HIP_LIB=$ROOT/scssim_amd/libscssim_hip_seams.so SCS_READS_SERIAL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 $ROOT/bench.py --steps 1 --warmup 1 --hbm-only --no-extra-legs --no-cpu-baseline > $OUT/run.log 2>&1
python3 - $(find $OUT/t -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 70 and float(r["TotalDurationNs"]) > 2e6 or "k_reads" in r["Name"]:   # the reads stage's per-batch kernels (74 batches in two jobs): base pass classes, pre-pass, scans
        print("%-60s calls %4s  avg %9.1f us  total %8.1f ms" % (r["Name"].split("(")[0][-60:] if "rocprim" not in r["Name"] else "rocprim " + r["Name"][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 $OUT/run.log | cut -c1-200
rm -rf $OUT/t
