#!/bin/bash
# builds and runs tools/probes/fetch_calib.hip under rocprofv3 (FETCH_SIZE and WRITE_SIZE in separate passes): gpurun -- 'bash tools/probes/fetch_calib.sh'
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/tools/probes/fetch_calib.hip -o $OUT/fetch_calib
cd /tmp && export TMPDIR=/tmp
$OUT/fetch_calib > $OUT/truth.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- $OUT/fetch_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- $OUT/fetch_calib > /dev/null 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for tag, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = r["Kernel_Name"].split("(")[0]
                res[k][name] = res[k].get(name, 0.0) + float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as o:
    o.write(open(out + "/truth.txt").read())
    o.write("rocprofv3 (KB reported x 1024):\n")
    for k, v in res.items():
        o.write("  %-12s FETCH_SIZE %.0f bytes   WRITE_SIZE %.0f bytes\n" % (k, v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024))
print(open(out + "/summary.txt").read())
PY
rm -rf $OUT/f $OUT/w $OUT/fetch_calib
