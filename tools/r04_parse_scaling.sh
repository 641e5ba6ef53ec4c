#!/bin/bash
# probe build: time of the flat parse kernel up to an exit word (stamp | skip << 8 | K << 16), no profiler.  usage: r04_parse_scaling.sh <outdir> words...
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export HVQM4_AMD_LIB=$GRAFT_REPO_ROOT/hvqm4_amd/abl/libhvq_${PROBE_LIB:-probe}.so
CACHE=/tmp/hvq_clip_cache
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE > $OUT/p0.json 2> $OUT/p0.err
for E in "$@"; do
  HVQM4_AMD_PARSE_EXIT=$E timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE > $OUT/e$E.json 2> $OUT/e$E.err
  echo "word $E stamp $((E & 255)) skip $(((E >> 8) & 255)) K $(((E >> 16) & 15)): $(grep 'parse probe' $OUT/e$E.err | awk '{print $(NF-3)}' | sort -n | head -6 | tr '\n' ' ')" | tee -a $OUT/summary.txt
done
