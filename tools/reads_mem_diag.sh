#!/bin/bash
# k_reads: how busy is the CU's vector-memory path (TA / TD / TCP) and its address translation (UTCL1)?  600 Mb job, text left in HBM.
#   tools/reads_mem_diag.sh [library.so]      (a few counters per pass: more TCP_* counters than the hardware collects at once abort rocprofv3)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/reads_mem_diag
mkdir -p $OUT
[ -n "$1" ] && export SCSSIM_HIP_LIB=$ROOT/$1
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --hbm-only --no-extra-legs --no-cpu-baseline --genome-mb 600"
i=0
for C in "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum GRBM_GUI_ACTIVE" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
         "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TD_TC_STALL_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
         "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
         "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
         "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TAGRAM0_REQ_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o a -- python3 $ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed: $C"
  echo "pass $i"
done
python3 - $OUT <<'PY'
import csv, sys, glob, collections, re
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?(scs::\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        if not any(s in k for s in ("k_reads", "k_attach_dense", "k_errs<false>", "k_indels", "k_plan_pairs")): continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
with open(out + "/summary.txt", "w") as o:
    for k, v in sorted(tot.items()):
        o.write(k + "\n")
        for c, x in sorted(v.items()):
            o.write("   %-46s %.4g per launch (%d launches)\n" % (c, x / len(n[(k, c)]), len(n[(k, c)])))
print(open(out + "/summary.txt").read())
PY
rm -rf $OUT/p[0-9]*/
