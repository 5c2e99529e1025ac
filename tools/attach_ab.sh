#!/bin/bash
# k_attach<semi>: the dense form (one lane = one primer) against the lane groups (SCS_ATTACH_GROUPS, seams build), on a 600 Mb job:
# kernel durations and the SQ counters (VALU wave-instructions, lanes per instruction, SALU) of each.  tools/attach_ab.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/attach_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SCSSIM_HIP_LIB=$ROOT/scssim_amd/libscssim_hip_seams.so
ARGS="--steps 1 --warmup 1 --hbm-only --no-extra-legs --no-cpu-baseline --genome-mb 600"
for v in dense groups; do
  if [ $v = groups ]; then export SCS_ATTACH_GROUPS=1; else unset SCS_ATTACH_GROUPS; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$v -o t -- python3 $ROOT/bench.py $ARGS > $OUT/trace_$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/s_$v -o s -- python3 $ROOT/bench.py $ARGS > $OUT/sq_$v.log 2>&1
  python3 $ROOT/tools/sq_summary.py $(find $OUT/s_$v -name "*counter_collection.csv" | head -1) $OUT/sq_$v.csv "bench.py $ARGS ($v)"
  echo "== $v"
  python3 - $(find $OUT/t_$v -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_attach" in n or "k_errs" in n or "k_stock" in n or "k_poisson" in n:
        print("%9.3f ms total %5s calls %9.1f us avg  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, n.split("(")[0][-50:]))
PY
  grep -E "k_attach|k_errs" $OUT/sq_$v.csv
  rm -rf $OUT/t_$v $OUT/s_$v
done
