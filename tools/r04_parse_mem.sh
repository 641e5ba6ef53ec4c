#!/bin/bash
# memory-side counters of the parse kernel for library variants.  usage: tools/r04_parse_mem.sh <outdir> base|<variant>...
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT
CACHE=/tmp/hvq_clip_cache
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE"
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/hvqm4_amd/abl/libhvq_$v.so; [ $v = base ] && L=$GRAFT_REPO_ROOT/hvqm4_amd/libhvqm4_amd.so
  export HVQM4_AMD_LIB=$L
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${v}_a --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum -- $B > $OUT/${v}_a.json 2> $OUT/${v}_a.err || { tail -3 $OUT/${v}_a.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${v}_b --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum -- $B > $OUT/${v}_b.json 2> $OUT/${v}_b.err || { tail -3 $OUT/${v}_b.err; exit 1; }
  echo "== $v"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/${v}_a | grep -A7 "hvq_parse_kernel"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/${v}_b | grep -A6 "hvq_parse_kernel"
done
