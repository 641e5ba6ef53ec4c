#!/bin/bash
# GPU box: A/B of environment-selected variants (and/or library builds) on the resident-descriptor bench.
# usage: tools/ab.sh <tag> "<presets>" <reps> name[:ENV=v[,ENV=v...]] ...        (ENV may include HVQM4_AMD_LIB=path)
T=$1; P=$2; R=$3; shift 3
O=gpurun_out/$T; mkdir -p $O
C="--clip-cache /tmp/hvq_clip_cache --no-sdk --no-gpu-parse --cpu-seconds 0"
for p in $P; do
  for rep in $(seq 1 $R); do
    for v in "$@"; do
      name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
      f=$O/${p}_${name}_$rep
      env $envs timeout -k 10 240 python bench.py $C $BENCH_EXTRA --preset $p > $f.json 2> $f.err || { echo "$p $name FAILED: $(tail -2 $f.err)" | tee -a $O/summary.txt; continue; }
      python3 - $f.json "$p $name rep $rep" <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("%-34s value %8.0f joined %.4f (%.1f us) | free-running %.4f (%.1f us)" % (sys.argv[2], d["value"], r["frac"], r["stage_us_per_step_joined"],
      r["free_running"]["frac"], r["free_running"]["us_per_step"]))
PY
    done
  done
done
