#!/bin/bash
# GPU box: parity suite of the working tree, then same-box A/B of the working tree's library against round 4's (hvqm4_amd/abl/libhvq_r04.so)
T=${1:-r05a}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 560 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?
tail -5 $O/gpu_tests.txt
[ $rc -ne 0 ] && { echo "tests failed rc $rc"; exit $rc; }
tools/r04_ab.sh $T "dense natural flat" 2 new old:HVQM4_AMD_LIB=$PWD/hvqm4_amd/abl/libhvq_r04.so
