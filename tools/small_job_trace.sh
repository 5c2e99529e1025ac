#!/bin/bash
# launches per job and GPU busy time of the 1 Mb configuration: tools/small_job_trace.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/small_job
mkdir -p $OUT
python3 $ROOT/tools/small_job.py --jobs 200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 $ROOT/tools/small_job.py --jobs 50 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
python3 - $(find $OUT/t -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
rows = [(float(r["TotalDurationNs"]), int(r["Calls"]), r["Name"].split("(")[0][-48:]) for r in csv.DictReader(open(sys.argv[1]))]
jobs = 52.0
print("launches per job: %.0f; GPU busy per job: %.3f ms" % (sum(r[1] for r in rows) / jobs, sum(r[0] for r in rows) / jobs / 1e6))
for t, n, name in sorted(rows, reverse=True)[:22]:
    print("%7.1f us/job %5.1f launches/job  %s" % (t / jobs / 1e3, n / jobs, name))
PY
rm -rf $OUT/t
