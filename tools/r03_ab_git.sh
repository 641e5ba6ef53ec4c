#!/bin/bash
# GPU box: the working tree's library against hvqm4_amd/abl/libhvq_<name>.so (tools/variant_from_git.sh), alternating, per preset;
# the e2e streaming figures too when E2E=1.   usage: tools/r03_ab_git.sh <tag> "<presets>" <name>
T=$1; P=$2; N=$3
O=gpurun_out/$T; mkdir -p $O
for p in $P; do
  for rep in 1 2; do
    for v in $N tree; do
      lib=hvqm4_amd/libhvqm4_amd.so; [ $v != tree ] && lib=hvqm4_amd/abl/libhvq_$v.so
      extra="--no-gpu-parse"; [ "$E2E" = 1 ] && extra=""
      HVQM4_AMD_LIB=$PWD/$lib timeout -k 10 250 python bench.py --preset $p --steps 20 --warmup 3 --no-sdk $extra --cpu-seconds 0 > $O/${p}_${v}_$rep.json 2> $O/${p}_${v}_$rep.err || { tail -3 $O/${p}_${v}_$rep.err; continue; }
      python - <<PY | tee -a $O/ab.txt
import json
d=json.loads(open("$O/${p}_${v}_$rep.json").read().strip().splitlines()[-1])
g=d.get("end_to_end_gpu_parse") or {}
print("$p %-6s rep $rep: value %8.0f frac %.4f ms/step %.3f  e2e streaming %s ms/batch %s parse %s" % ("$v", d["value"], d["roofline"]["frac"], d["ms_per_step"], g.get("streaming_value"), g.get("streaming_ms_per_batch"), g.get("streaming_parse_kernel_ms")))
PY
    done
  done
done
