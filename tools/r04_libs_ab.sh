#!/bin/bash
# parse kernel time of library variants.  usage: tools/r04_libs_ab.sh <outdir> base|<variant>...
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
  L=$GRAFT_REPO_ROOT/hvqm4_amd/abl/libhvq_$v.so; [ $v = base ] && L=$GRAFT_REPO_ROOT/hvqm4_amd/libhvqm4_amd.so
  HVQM4_AMD_LIB=$L timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-sdk --cpu-seconds 0 > $O/$v.$rep.json 2> $O/$v.$rep.err || { tail -3 $O/$v.$rep.err; continue; }
  python3 -c "
import json
e=json.loads(open('$O/$v.$rep.json').read().strip().split('\n')[-1])['end_to_end_gpu_parse']
print('$v', e['parse_kernel_ms'], e['streaming_parse_kernel_ms'], e['streaming_ms_per_batch'], e['streaming_ms_per_batch_median'])"
done; done
