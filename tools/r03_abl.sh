#!/bin/bash
# GPU box: timing of the ablation builds (hvqm4_amd/abl/libhvq_<name>.so, tools/variant.sh) next to the shipped library, alternating.
# usage: tools/r03_abl.sh <tag> "<presets>" <name>...
T=$1; P=$2; shift 2
O=gpurun_out/$T; mkdir -p $O
for p in $P; do
  for rep in 1 2; do
    for v in base "$@"; do
      lib=hvqm4_amd/libhvqm4_amd.so; [ $v != base ] && lib=hvqm4_amd/abl/libhvq_$v.so
      HVQM4_AMD_LIB=$PWD/$lib timeout -k 10 200 python bench.py --preset $p --steps 20 --warmup 3 --no-sdk --no-gpu-parse --cpu-seconds 0 --no-verify > $O/${p}_${v}_$rep.json 2> $O/${p}_${v}_$rep.err || { tail -3 $O/${p}_${v}_$rep.err; continue; }
      python - <<PY | tee -a $O/abl.txt
import json
d=json.load(open("$O/${p}_${v}_$rep.json"))
print("$p %-10s rep $rep: value %8.0f frac %.4f ms/step %.3f" % ("$v", d["value"], d["roofline"]["frac"], d["ms_per_step"]))
PY
    done
  done
done
