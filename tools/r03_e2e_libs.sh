#!/bin/bash
# GPU box: variant libraries against each other on the end-to-end legs and the headline, alternating.  usage: tools/r03_e2e_libs.sh <tag> "<presets>" <name>...
T=$1; P=$2; shift 2
O=gpurun_out/$T; mkdir -p $O
for p in $P; do
for rep in 1 2 3; do
  for v in "$@"; do
    HVQM4_AMD_LIB=$PWD/hvqm4_amd/abl/libhvq_$v.so timeout -k 10 250 python bench.py --preset $p --steps 20 --warmup 3 --no-sdk --cpu-seconds 0 > $O/${p}_${v}_$rep.json 2> $O/${p}_${v}_$rep.err || { tail -3 $O/${p}_${v}_$rep.err; continue; }
    python - <<PY | tee -a $O/e2e.txt
import json
d=json.loads(open("$O/${p}_${v}_$rep.json").read().strip().splitlines()[-1])
g=d["end_to_end_gpu_parse"]
print("$p %-8s rep $rep: frac %.4f | parse %.3f ms  one batch %.0f  streaming %.0f Mpx/s  %.2f ms/batch" % ("$v", d["roofline"]["frac"], g["streaming_parse_kernel_ms"], g["value"], g["streaming_value"], g["streaming_ms_per_batch"]))
PY
  done
done
done
