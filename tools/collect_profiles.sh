#!/bin/bash
# Collect the measurements kept under profiles/ (run on the GPU box: gpurun -- 'bash tools/collect_profiles.sh r01').
# Output goes to gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards.
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
python3 bench.py > $O/${TAG}_bench_n1.json.log 2> $O/bench.err
# the bench line's kernel timings come from HIP events; the same command under rocprofv3 gives the per-kernel averages
rocprofv3 --kernel-trace --stats -d $O/kt -o b --output-format csv -- python3 bench.py --no-cpu-baseline > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv") $O/${TAG}_bench_n1_kernel_stats.csv
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o f --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o w --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/pw.log 2>&1
python3 tools/pmc_summary.py $(find $O/pf -name "*counter_collection.csv") $(find $O/pw -name "*counter_collection.csv") $O/${TAG}_bench_n1_pmc_hbm.csv "python3 bench.py --steps 4 --warmup 1"
# 300 Mb in 3 records (6 haplotype sequences), 30x
python3 tools/big_run.py 300 3 30 > $O/${TAG}_300mb_1gpu.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt3 -o b --output-format csv -- python3 tools/big_run.py 300 3 30 > $O/kt3.log 2>&1
cp $(find $O/kt3 -name "*kernel_stats.csv") $O/${TAG}_300mb_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf3 -o f --output-format csv -- python3 tools/big_run.py 300 3 30 > $O/pf3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw3 -o w --output-format csv -- python3 tools/big_run.py 300 3 30 > $O/pw3.log 2>&1
python3 tools/pmc_summary.py $(find $O/pf3 -name "*counter_collection.csv") $(find $O/pw3 -name "*counter_collection.csv") $O/${TAG}_300mb_pmc_hbm.csv "python3 tools/big_run.py 300 3 30"
rm -rf $O/kt $O/kt3 $O/pf $O/pw $O/pf3 $O/pw3
tail -n 1 $O/${TAG}_bench_n1.json.log | cut -c1-400
tail -n 1 $O/${TAG}_300mb_1gpu.log | cut -c1-300
