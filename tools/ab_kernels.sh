#!/bin/bash
# A/B of two builds of the library on the same box, alternating, three rounds: tools/ab_kernels.sh OLD.so NEW.so [bench.py arguments]
A=$1; B=$2; shift 2
for r in 1 2 3; do
  for lib in "$A" "$B"; do
    SCSSIM_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --hbm-only --no-extra-legs --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],1), 'k_reads', round(d['roofline']['avg_launch_ms'],3), {k: round(v*1e3,1) for k,v in d['stages_s_per_step'].items() if v})"
  done
done
