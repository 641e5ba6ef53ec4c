#!/bin/bash
# PMC passes for the bench workload (separate rocprofv3 runs, --kernel-trace only; see MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [bench args...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk $@"
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $B > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $B > $OUT/p2.json 2> $OUT/p2.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc FETCH_SIZE -- $B > $OUT/p3.json 2> $OUT/p3.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p4 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $B > $OUT/p4.json 2> $OUT/p4.err
echo done
