#!/bin/bash
# runs bench.py --hbm-only once per library build given (ablation / variant builds): tools/probes/abl_run.sh lib1.so lib2.so ...
for lib in "$@"; do
  SCSSIM_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --hbm-only --no-extra-legs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],1), 'k_reads', round(d['roofline']['avg_launch_ms'],3), {k: round(v*1e3,1) for k,v in d['stages_s_per_step'].items() if v}, {k: round(v,1) for k,v in d['kernels_ms_per_step'].items()})"
done
