#!/bin/bash
cd $GRAFT_REPO_ROOT
for sh in 0 20 21; do
  if [ $sh = 0 ]; then env=""; else env="SCSSIM_HIP_LIB=$GRAFT_REPO_ROOT/scssim_amd/libscssim_hip_seams.so SCS_TEST_BATCH_SHIFT=$sh"; fi
  env $env python bench.py --steps 2 --warmup 1 --no-extra-legs --no-cpu-baseline > gpurun_out/r04_batch_$sh.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04_batch_$sh.log") if l.startswith("{")][-1])
print("shift $sh value %.1f M pairs/s ms/step %.0f stages %s sink %.1f GB/s" % (d["value"]/1e6, d["ms_per_step"], {k:round(v,2) for k,v in d["stages_s_per_step"].items()}, d["config"]["sink_GBps"]))
print("   k_reads in timed region: frac %.3f avg %.2f ms launches %d" % (d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["timed_launches"]))
PY
done
