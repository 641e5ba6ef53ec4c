#!/bin/bash
# GPU: parity tests of the device entropy parse, then the end-to-end bench leg with the flat path on and off
# usage: tools/r02_parse.sh <tag> [preset]
set -o pipefail
tag=${1:-r02p}; preset=${2:-dense}
out=gpurun_out/$tag; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_gparse.py -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  tail -3 $out/tests.log
fi
for flat in 1 0; do
  HVQM4_AMD_PARSE_FLAT=$flat HVQM4_AMD_PARSE_TIMING=1 timeout -k 10 500 python bench.py --steps 6 --warmup 2 --no-sdk --cpu-seconds 0 --preset $preset \
      > $out/bench_flat$flat.json 2> $out/bench_flat$flat.err || { tail -20 $out/bench_flat$flat.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$out/bench_flat$flat.json"))
e=d["end_to_end_gpu_parse"]
print("flat=$flat", "$preset", "parse_kernel_ms", e.get("parse_kernel_ms"), "streaming", e.get("streaming"), "value", d["value"])
PY
  grep "pictures (" $out/bench_flat$flat.err | tail -3
  grep "slowest picture" $out/bench_flat$flat.err | tail -1
done
grep "handed to the chains" $out/bench_flat1.err | tail -1
