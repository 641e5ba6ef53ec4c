#!/bin/bash
T=${1:-r04j}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_c_player.py tests/test_gpu_reject.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
HVQM4_AMD_SDK_TIMING=1 timeout -k 10 120 python tools/sdk_loop.py 2 > $O/sdk_timing.txt 2>&1; tail -4 $O/sdk_timing.txt
timeout -k 10 120 python tools/sdk_loop.py 5 2>&1 | tail -1
timeout -k 10 400 python bench.py --clip-cache /tmp/hvq_clip_cache --no-gpu-parse > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("value %.0f stage %.4f" % (d["value"], r["frac"]), "| sdk", json.dumps(d["sdk_path"])[:200], "| host e2e", d["end_to_end"]["value"], d["end_to_end"]["parse_only_mpix_s"])
print("cpu baseline", d["cpu_baseline"]["value"], "C4 all cores", d["cpu_baseline"]["configs"]["all_cores_C4"]["value"])
PY
