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
    if "k_reads" in r["Name"] or "k_indels" in r["Name"] or "k_plan" in r["Name"]:
        print("%-60s calls %4s  avg %9.1f us  total %8.1f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 $OUT/run.log | cut -c1-200
rm -rf $OUT/t
