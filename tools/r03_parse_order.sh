#!/bin/bash
# GPU box: device entropy parse with the job list in submission order (HVQM4_AMD_PARSE_ORDER=0) and shuffled (default), alternating
O=gpurun_out/$1; mkdir -p $O
for p in ${2:-dense}; do
for rep in 1 2; do
  for v in 0 1; do
    HVQM4_AMD_PARSE_ORDER=$v timeout -k 10 250 python bench.py --preset $p --steps 10 --warmup 2 --no-sdk --cpu-seconds 0 > $O/${p}_order${v}_$rep.json 2> $O/${p}_order${v}_$rep.err || { tail -3 $O/${p}_order${v}_$rep.err; continue; }
    python - <<PY | tee -a $O/order.txt
import json
d=json.loads(open("$O/${p}_order${v}_$rep.json").read().strip().splitlines()[-1])
g=d["end_to_end_gpu_parse"]
print("$p order $v rep $rep: parse kernel %.3f ms (streaming %.3f)  one batch %.0f  streaming %.0f Mpx/s  %.2f ms/batch  readback %.0f" % (g["parse_kernel_ms"], g["streaming_parse_kernel_ms"], g["value"], g["streaming_value"], g["streaming_ms_per_batch"], g["streaming_with_readback"].get("value", 0)))
PY
  done
done
done
