#!/bin/bash
# PMC passes that characterise the entropy-parse kernel (instruction mix and busy/wait cycles); separate rocprofv3
# runs with --kernel-trace only.  usage: tools/pmc_parse.sh <outdir-under-gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify"
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -- $B > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU -- $B > $OUT/p2.json 2> $OUT/p2.err
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt
grep -A18 "hvq_parse_kernel" $OUT/summary.txt
