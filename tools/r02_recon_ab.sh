#!/bin/bash
# GPU box: A/B of two builds of the library on the reconstruction bench (descriptors resident), alternating, one box.
# usage: tools/r02_recon_ab.sh <tag> <libA> <libB> [presets...]
set -o pipefail
tag=$1; A=$2; B=$3; shift 3
presets=${*:-dense natural flat}
O=gpurun_out/$tag; mkdir -p $O
for p in $presets; do
  for rep in 1 2; do
    for v in A B; do
      lib=$A; [ $v = B ] && lib=$B
      HVQM4_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --preset $p --steps 20 --warmup 3 --no-sdk --no-gpu-parse --cpu-seconds 0 > $O/${p}_${v}_$rep.json 2> $O/${p}_${v}_$rep.err || { tail -5 $O/${p}_${v}_$rep.err; exit 1; }
      python - <<PY
import json
d=json.load(open("$O/${p}_${v}_$rep.json"))
print("$p $v rep $rep: value %.0f frac %.4f ms/step %.3f" % (d["value"], d["roofline"]["frac"], d["ms_per_step"]))
PY
    done
  done
done
