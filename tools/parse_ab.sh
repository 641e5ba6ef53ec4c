#!/bin/bash
# GPU box: same-box A/B of the parse kernel inside the streaming leg of the default bench.
# usage: tools/parse_ab.sh <reps> name[=lib] ...    (name `new` = the working tree's library; others hvqm4_amd/abl/libhvq_<name>.so)
reps=${1:-2}; shift
for r in $(seq 1 $reps); do
 for v in "$@"; do
  L=$PWD/hvqm4_amd/abl/libhvq_$v.so; [ $v = new ] && L=$PWD/hvqm4_amd/libhvqm4_amd.so
  HVQM4_AMD_LIB=$L timeout -k 10 300 python bench.py --no-sdk --cpu-seconds 0 --clip-cache /tmp/hvq_clip_cache 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end_gpu_parse']
print('%-8s rep $r: streaming %.1f (%.2f ms) parse kernel %.3f ms (one batch %.3f) recon frac %.4f' % ('$v', e['streaming_value'], e['streaming_ms_per_batch'], e['streaming_parse_kernel_ms'], e['parse_kernel_ms'], d['roofline']['frac']))"
 done
done
