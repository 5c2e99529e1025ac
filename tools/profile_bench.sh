#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's roofline on the GPU box (run through gpurun from the repo root):
#   kernel trace (durations), HBM traffic (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md
#   prescribes) and the SQ instruction counters; summaries land in gpurun_out/prof_<tag>/ for copying into profiles/.
# usage: tools/profile_bench.sh <tag> [bench.py arguments]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --hbm-only: the generation_hbm leg as the timed region (8 M pairs per k_reads launch: where bench.py takes the roofline of the dominant kernel)
ARGS="--steps 1 --warmup 1 --hbm-only --no-extra-legs --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -o s -- python3 $ROOT/bench.py $ARGS > $OUT/bench_sq.log 2>&1
echo "sq done"
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1); S=$(find $OUT/sq -name "*counter_collection.csv" | head -1)
K=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 $ROOT/tools/pmc_summary.py $F $W $OUT/bench_pmc_hbm.csv "python3 bench.py $ARGS"
python3 $ROOT/tools/sq_summary.py $S $OUT/bench_sq.csv "python3 bench.py $ARGS"
cp $K $OUT/bench_kernel_stats.csv
# launch geometry of the profiled command (bench.py scales the per-launch counters by it)
python3 - $OUT/bench_trace.log $OUT/bench_meta.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ppl = d["config"]["pairs_per_step"] * d["steps"] / d["roofline"]["timed_launches"]
json.dump({"pairs_per_launch": ppl, "note": "k_reads launches of the profiled command: %d launches over %d timed job(s) of %d pairs" % (d["roofline"]["timed_launches"], d["steps"], d["config"]["pairs_per_step"])}, open(sys.argv[2], "w"), indent=1)
PY
# the raw per-dispatch tables are large: keep the summaries only
rm -rf $OUT/fetch $OUT/write $OUT/sq $OUT/trace
ls -la $OUT
