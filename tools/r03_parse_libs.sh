#!/bin/bash
# GPU box: device entropy parse time with variant libraries (hvqm4_amd/abl/libhvq_<name>.so), alternating.  usage: tools/r03_parse_libs.sh <tag> "<presets>" <name>...
T=$1; P=$2; shift 2
O=gpurun_out/$T; mkdir -p $O
for p in $P; do
for rep in 1 2; do
  for v in "$@"; do
    HVQM4_AMD_LIB=$PWD/hvqm4_amd/abl/libhvq_$v.so timeout -k 10 250 python bench.py --preset $p --steps 5 --warmup 1 --no-sdk --cpu-seconds 0 > $O/${p}_${v}_$rep.json 2> $O/${p}_${v}_$rep.err || { tail -3 $O/${p}_${v}_$rep.err; continue; }
    python - <<PY | tee -a $O/parse.txt
import json
d=json.loads(open("$O/${p}_${v}_$rep.json").read().strip().splitlines()[-1])
g=d["end_to_end_gpu_parse"]
print("$p %-12s rep $rep: parse kernel %.3f ms (streaming %.3f)  streaming %.0f Mpx/s  %.2f ms/batch  checked %s" % ("$v", g["parse_kernel_ms"], g["streaming_parse_kernel_ms"], g["streaming_value"], g["streaming_ms_per_batch"], g["pictures_checked_against_host_parsed"]))
PY
  done
done
done
