#!/bin/bash
# GPU box: per-dependency-level time, instruction mix and VALU busy of hvq_recon_kernel for one or more builds of the library
# (two rocprofv3 --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes).
# usage: tools/r03_levels.sh <tag> <preset> <name=lib>...
T=$1; P=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $OUT
CACHE=/tmp/hvq_clip_cache
python3 $GRAFT_REPO_ROOT/bench.py --preset $P --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
for nl in "$@"; do
  name=${nl%%=*}; lib=${nl#*=}
  export HVQM4_AMD_LIB=$GRAFT_REPO_ROOT/$lib
  B="python3 $GRAFT_REPO_ROOT/bench.py --preset $P --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --gen-workers 1 --clip-cache $CACHE"
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name/p1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $B > $OUT/$name.p1.json 2> $OUT/$name.p1.err
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name/p2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $B > $OUT/$name.p2.json 2> $OUT/$name.p2.err
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name/p3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -- $B > $OUT/$name.p3.json 2> $OUT/$name.p3.err
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/$name/t -- $B > $OUT/$name.t.json 2> $OUT/$name.t.err
  echo "== $name ($P)" | tee -a $OUT/levels.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_levels_lite.py $OUT/$name | tee -a $OUT/levels.txt
done
