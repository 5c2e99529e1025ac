#!/bin/bash
# phase times of the uniform walk's workgroups and of k_attach<semi>'s waves: tools/phase_clock.sh [LIB.so]   (LIB: `make -C scssim_amd/csrc phase-clock` -> scssim_amd/libscssim_hip_phaseclock.so)
SCS_PHASE_CLOCK=1 SCSSIM_HIP_LIB=$PWD/${1:-scssim_amd/libscssim_hip_phaseclock.so} timeout -k 10 300 python3 bench.py --hbm-only --no-extra-legs --no-cpu-baseline --steps 1 --warmup 1 2>&1 | grep "phase clock" | tail -2
