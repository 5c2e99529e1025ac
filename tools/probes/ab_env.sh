#!/bin/bash
# A/B of one environment knob on the same box: tools/probes/ab_env.sh KNOB [bench.py arguments]; alternates default / KNOB=1 twice
K=$1; shift
for r in 1 2; do
  for v in "" 1; do
    if [ -n "$v" ]; then export $K=1; else unset $K; fi
    timeout -k 10 300 python3 bench.py --no-extra-legs --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$K=$v', round(d['ms_per_step'],1), {k: round(v*1e3,1) for k,v in d['stages_s_per_step'].items()})"
  done
done
