#!/bin/bash
# The parse kernel under the counters: per-phase timeline, HBM traffic, instruction mix.  usage: tools/r04_parse_probe.sh <outdir-under-gpurun_out> [lib]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
[ -n "$1" ] && export HVQM4_AMD_LIB=$GRAFT_REPO_ROOT/$1
CACHE=/tmp/hvq_clip_cache
cd $GRAFT_REPO_ROOT
HVQM4_AMD_PARSE_TIMING=1 timeout -k 10 400 python3 bench.py --steps 6 --warmup 2 --no-sdk --cpu-seconds 0 --no-verify --clip-cache $CACHE > $OUT/timing.json 2> $OUT/timing.err
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE"
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$n --pmc "$@" -- $B > $OUT/$n.json 2> $OUT/$n.err; }
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
pass p3 FETCH_SIZE
pass p3r TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass p4w TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
grep -A40 "hvq_parse_kernel" $OUT/summary.txt | head -60
grep "hvqm4_amd parse" $OUT/timing.err | tail -12
