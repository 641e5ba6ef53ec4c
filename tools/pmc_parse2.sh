#!/bin/bash
# PMC passes on the entropy-parse kernel: instruction fetch / i-cache and LDS pressure.  usage: tools/pmc_parse2.sh <outdir> [bench args]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# clips are generated once, OUTSIDE the profiler: a profiled run must start no worker processes (the profiler's preload has
# initialised the GPU in the parent; see tools/pmc_passes.sh)
CACHE=/tmp/hvq_clip_cache
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE $BENCH_ARGS "$@" > /dev/null 2>&1 || true
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --gen-workers 1 --clip-cache $CACHE --no-verify --no-sdk $*"
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_BRANCH -- $B > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY -- $B > $OUT/p2.json 2> $OUT/p2.err
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
grep -A17 "hvq_parse_kernel" $OUT/summary.txt
