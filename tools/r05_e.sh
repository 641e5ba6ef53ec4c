#!/bin/bash
T=${1:-r05e}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modes.py -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
A=$PWD/hvqm4_amd/abl
tools/r04_ab.sh $T "dense natural" 2 new x4:HVQM4_AMD_LIB=$A/libhvq_x4.so
tools/pmc_insts.sh ${T}_insts base a31 a32 a33 a36 a37 2>&1 | grep -v "^$" | tail -8
