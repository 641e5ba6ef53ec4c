#!/bin/bash
# instruction-fetch counters of the parse kernel.  usage: tools/r04_parse_ifetch.sh <outdir>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
CACHE=/tmp/hvq_clip_cache
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/a --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES -- $B > $OUT/a.json 2> $OUT/a.err || { tail -3 $OUT/a.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/b --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -- $B > $OUT/b.json 2> $OUT/b.err || { tail -3 $OUT/b.err; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | grep -A16 "hvq_parse_kernel"
