set -e
cp scssim_amd/libscssim_hip.so /tmp/x.so
timeout -k 10 400 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
run() { cp $1 scssim_amd/libscssim_hip.so; python bench.py --steps 400 --warmup 20 --no-cpu-baseline > gpurun_out/b_$2.log 2>&1; }
run /tmp/x.so xa; run ab_base.so ba; run /tmp/x.so xb; run ab_base.so bb; run /tmp/x.so xc; run ab_base.so bc
cp /tmp/x.so scssim_amd/libscssim_hip.so
python - <<'P'
import json
for t in ("xa","ba","xb","bb","xc","bc"):
    d=[json.loads(l) for l in open("gpurun_out/b_%s.log"%t) if l.startswith("{")][0]
    print(t, round(d["ms_per_step"],4), int(d["value"]))
P
