#!/bin/bash
T=${1:-r05m}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modes.py -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
A=$PWD/hvqm4_amd/abl
tools/ab.sh $T "dense natural flat" 2 new prev:HVQM4_AMD_LIB=$A/libhvq_prev.so
tools/pmc_insts.sh ${T}_insts base 2>&1 | grep -v "^$" | tail -1
