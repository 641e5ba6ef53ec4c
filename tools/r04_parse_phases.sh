#!/bin/bash
# What the parse kernel has cost up to each of its phase stamps: the probe build (tools/variant.sh probe -DGP_PROBE) runs the flat kernel
# up to stamp E in front of every real parse launch; instruction counters of both launches.  usage: tools/r04_parse_phases.sh <outdir> [stamps...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export HVQM4_AMD_LIB=$GRAFT_REPO_ROOT/hvqm4_amd/abl/libhvq_probe.so
CACHE=/tmp/hvq_clip_cache
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE"
for E in ${@:-1 13 12 3 4 9 10 6}; do
  HVQM4_AMD_PARSE_EXIT=$E timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/e$E --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -- $B > $OUT/e$E.json 2> $OUT/e$E.err
  grep "parse probe" $OUT/e$E.err | tail -2
done
python3 $GRAFT_REPO_ROOT/tools/parse_phases.py $OUT | tee $OUT/phases.txt
