#!/bin/bash
# builds abl/lib_<tag>.so: the library with ONE kernel file compiled with extra flags (ablation / variant builds for tools/probes/abl_run.sh)
#   tools/probes/abl_build.sh <tag> <scs_k_file.hip> [extra hipcc flags]
set -e
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../../scssim_amd/csrc"
mkdir -p ../../abl build
STEM=${SRC%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-function "$@" -c $SRC -o build/${STEM}_$TAG.o
OBJS=""
for o in scs_k_reads scs_k_amplify scs_k_staging scs_k_allocate scs_k_misc scs_pipeline scs_stage scs_amplify scs_reads scs_tables scs_comm scs_simuvars scs_bgzf scs_seams_on; do
  if [ $o = $STEM ]; then OBJS="$OBJS build/${STEM}_$TAG.o"; else OBJS="$OBJS build/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -lpthread -o ../../abl/lib_$TAG.so
echo abl/lib_$TAG.so
