#!/bin/bash
# GPU box: timing-only ablations of hvq_recon_inline_kernel (wrong pictures: --no-verify), alternating with the shipped build
T=$1; shift
O=gpurun_out/$T; mkdir -p $O
C="--clip-cache /tmp/hvq_clip_cache --no-sdk --no-gpu-parse --cpu-seconds 0 --no-verify"
python bench.py $C --steps 1 --warmup 0 > /dev/null 2>&1
for p in dense flat; do
  for v in base "$@" base; do
    lib=hvqm4_amd/libhvqm4_amd.so; [ $v != base ] && lib=hvqm4_amd/abl/libhvq_$v.so
    HVQM4_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py $C --preset $p > $O/${p}_$v.json 2> $O/${p}_$v.err || { echo "$p $v failed"; continue; }
    python3 -c "
import json; d=json.load(open('$O/${p}_$v.json')); print('$p %-8s stage %.4f %.1f us' % ('$v', d['roofline']['frac'], d['roofline']['stage_us_per_step']))" | tee -a $O/abl.txt
  done
done
