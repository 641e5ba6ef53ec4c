#!/bin/bash
# PMC passes for the bench workload (separate rocprofv3 runs, --kernel-trace only; see MI355X_MICROARCH.md
# "rocprofv3 PMC slots"), each followed by the same pass over the calibration kernels (tools/ubench/pmc_calib).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [bench args...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
CACHE=/tmp/hvq_clip_cache
# clips are generated once, outside the profiler (the profiled runs start no worker processes)
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE "$@" > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --gen-workers 1 --clip-cache $CACHE $@"
CAL="$GRAFT_REPO_ROOT/tools/ubench/pmc_calib"
pass() {  # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$n --pmc "$@" -- $B > $OUT/$n.json 2> $OUT/$n.err
  rocprofv3 --kernel-trace --output-format csv -d $OUT/cal_$n --pmc "$@" -- $CAL > $OUT/cal_$n.txt 2> $OUT/cal_$n.err
}
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass p3 FETCH_SIZE
pass p3r TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass p4w TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
echo done
