#!/bin/bash
# GPU box, round 4 session 1: stage-inclusive bench line, queue-build variants, flush timeline, SDK call breakdown, TA calibration
T=${1:-r04a}; O=gpurun_out/$T; mkdir -p $O
C="--clip-cache /tmp/hvq_clip_cache"
line() { python3 - "$1" "$2" <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%-22s value %8.0f stage frac %.4f (%.1f us) | recon only %.4f (%.1f us) | queue build %.1f us %s GB/s" % (sys.argv[2], d["value"], r["frac"], r["stage_us_per_step"],
          r["recon_only"]["frac"], r["recon_only"]["us_per_step"], r["queue_build"]["us_per_step"], r["queue_build"]["GB/s"]))
    for k in ("two_queues", "c5_staggered"):
        if k in d: print("   ", k, json.dumps(d[k])[:300])
    e = d.get("end_to_end_gpu_parse")
    if e: print("    streaming %.0f Mpx/s, %.2f ms per batch, parse kernel %.3f ms; calls %s" % (e["streaming_value"], e["streaming_ms_per_batch"], e["streaming_parse_kernel_ms"], e["streaming_submit_end_begin_ms"]))
    if "sdk_path" in d: print("    sdk", json.dumps(d["sdk_path"])[:200])
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
timeout -k 10 400 python bench.py $C > $O/bench.json 2> $O/bench.err && line $O/bench.json default && \
for sp in 1 2 4 0; do
  HVQM4_AMD_TILEQ_SPLITS=$sp timeout -k 10 200 python bench.py $C --no-sdk --no-gpu-parse --cpu-seconds 0 > $O/bench_sp$sp.json 2> $O/bench_sp$sp.err || exit 1
  line $O/bench_sp$sp.json "dense splits=$sp"
done && \
for sp in 0 1; do
  HVQM4_AMD_TILEQ_SPLITS=$sp timeout -k 10 200 python bench.py $C --preset natural --no-sdk --no-gpu-parse --cpu-seconds 0 > $O/bench_nat_sp$sp.json 2> $O/bench_nat_sp$sp.err || exit 1
  line $O/bench_nat_sp$sp.json "natural splits=$sp"
done && \
HVQM4_AMD_FLUSH_TIMING=1 timeout -k 10 300 python bench.py $C --no-sdk --cpu-seconds 0 > $O/bench_ft.json 2> $O/bench_ft.err && line $O/bench_ft.json "flush timing" && \
HVQM4_AMD_SDK_TIMING=1 timeout -k 10 120 python tools/sdk_loop.py 1 > $O/sdk_timing.txt 2>&1 && tail -20 $O/sdk_timing.txt && \
HVQM4_AMD_TILEQ_SPLITS=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu > $O/gpu_tests_loop.txt 2>&1; tail -3 $O/gpu_tests_loop.txt
timeout -k 10 300 tools/r04_ta_calib.sh ${T}_ta
