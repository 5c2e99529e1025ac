#!/bin/bash
# like abl_run.sh with the reads stage's kernels one after the other (SCS_READS_SERIAL=1): a kernel's duration without its neighbours
for lib in "$@"; do
  SCS_READS_SERIAL=1 SCSSIM_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --hbm-only --no-extra-legs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial $lib', round(d['ms_per_step'],1), 'k_reads', round(d['roofline']['avg_launch_ms'],3), {k: round(v*1e3,1) for k,v in d['stages_s_per_step'].items() if v})"
done
