#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
CACHE=/tmp/hvq_clip_cache
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/c --pmc InstrFetchLatency -- $B > $OUT/c.json 2> $OUT/c.err || { tail -3 $OUT/c.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/d --pmc SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_REQ SQ_BUSY_CYCLES -- $B > $OUT/d.json 2> $OUT/d.err || { tail -3 $OUT/d.err; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | grep -B1 -A6 "hvq_parse_kernel\|hvq_recon_inline"
