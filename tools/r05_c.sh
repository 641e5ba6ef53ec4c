#!/bin/bash
# GPU box: full parity suite, then A/B of the library variants named on the command line and a stamped run of the dense stream
T=${1:-r05c}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?
tail -6 $O/gpu_tests.txt
[ $rc -ne 0 ] && { echo "tests failed rc $rc"; grep -n "Error\|FAILED\|assert" $O/gpu_tests.txt | head -20; }
A=$PWD/hvqm4_amd/abl
tools/r04_ab.sh $T "dense natural flat" 2 new x4:HVQM4_AMD_LIB=$A/libhvq_x4.so old:HVQM4_AMD_LIB=$A/libhvq_r04.so
tools/r04_ab.sh ${T}_tpw "dense" 2 tpw2:HVQM4_AMD_TILES_PER_WG=2 tpw1:HVQM4_AMD_TILES_PER_WG=1
B="--cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache /tmp/hvq_clip_cache"
for p in dense natural; do
  HVQM4_AMD_LIB=$A/libhvq_stamps.so HVQM4_AMD_STAMPS=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 $B --preset $p > $O/stamps_$p.json 2> $O/stamps_$p.err
  grep -h "^stamps" $O/stamps_$p.err | tail -7 > $O/stamps_$p.txt; cat $O/stamps_$p.txt
done
