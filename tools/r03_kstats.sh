#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the default bench command (all legs).  usage: tools/r03_kstats.sh <tag> [bench args]
T=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
CACHE=/tmp/hvq_clip_cache
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE "$@" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE "$@" > $O/bench.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cat $O/kernel_stats.csv
python3 -c "
import json; d=json.load(open('$O/bench.json')); e=d['end_to_end_gpu_parse']
print('value', d['value'], 'frac', d['roofline']['frac'], 'streaming', e['streaming_value'], 'ms/batch', e['streaming_ms_per_batch'], 'parse', e['streaming_parse_kernel_ms'], 'readback', e['streaming_with_readback'])"
