#!/bin/bash
# GPU box: one environment switch off / on, alternating, end-to-end legs included.  usage: tools/r03_env_ab.sh <tag> <VAR> "<presets>" [reps]
T=$1; V=$2; P=$3; R=${4:-3}
O=gpurun_out/$T; mkdir -p $O
for p in $P; do
for rep in $(seq 1 $R); do
  for v in 0 1; do
    env $V=$v timeout -k 10 250 python bench.py --preset $p --steps 10 --warmup 2 --no-sdk --cpu-seconds 0 > $O/${p}_${v}_$rep.json 2> $O/${p}_${v}_$rep.err || { tail -3 $O/${p}_${v}_$rep.err; continue; }
    python - <<PY | tee -a $O/ab.txt
import json
d=json.loads(open("$O/${p}_${v}_$rep.json").read().strip().splitlines()[-1])
g=d["end_to_end_gpu_parse"]
print("$p $V=$v rep $rep: frac %.4f | parse %.3f ms  one batch %.0f  streaming %.0f Mpx/s  %.2f ms/batch  readback %.0f" % (d["roofline"]["frac"], g["streaming_parse_kernel_ms"], g["value"], g["streaming_value"], g["streaming_ms_per_batch"], g["streaming_with_readback"].get("value", 0)))
PY
  done
done
done
