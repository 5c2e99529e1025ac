#!/bin/bash
# k_reads: how busy are the vector pipe and the LDS pipe?  SQ cycle counters on a 600 Mb job (8 M-pair launches, text left in HBM).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/reads_diag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --hbm-only --no-extra-legs --no-cpu-baseline --genome-mb 600"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o a -- python3 $ROOT/bench.py $ARGS > $OUT/p1.log 2>&1
echo p1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU --output-format csv -d $OUT/p2 -o b -- python3 $ROOT/bench.py $ARGS > $OUT/p2.log 2>&1
echo p2
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d $OUT/p3 -o c -- python3 $ROOT/bench.py $ARGS > $OUT/p3.log 2>&1
echo p3
python3 - $OUT <<'PY'
import csv, sys, glob, collections, re
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?(scs::\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        if "k_reads" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
for k, v in sorted(tot.items()):
    print(k)
    for c, x in sorted(v.items()):
        print("   %-30s %.4g per launch (%d launches)" % (c, x / len(n[(k, c)]), len(n[(k, c)])))
PY
tail -2 $OUT/p1.log | cut -c1-300
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
