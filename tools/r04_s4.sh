#!/bin/bash
# GPU box: GPU test suite, full default bench line, streaming timeline, eight ranks on the one GPU
T=${1:-r04g}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
timeout -k 10 400 python bench.py --clip-cache /tmp/hvq_clip_cache > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - $O/bench.json <<'PY' | tee $O/summary.txt
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("value %.0f stage %.4f (%.1f us) recon-only %.4f" % (d["value"], r["frac"], r["stage_us_per_step"], r["recon_only"]["frac"]))
for k in ("two_pass_tile_queues", "two_queues", "c5_staggered", "c4_share", "sdk_path", "rgb_epilogue"):
    print(k, json.dumps(d.get(k))[:420])
e = d["end_to_end_gpu_parse"]
print("streaming %.0f Mpx/s %.2f ms/batch parse %.3f ms calls %s; one batch %.0f; readback %s" % (e["streaming_value"], e["streaming_ms_per_batch"], e["streaming_parse_kernel_ms"], e["streaming_submit_end_begin_ms"], e["value"], json.dumps(e["streaming_with_readback"])[:200]))
PY
tools/trace_streaming.sh ${T}_stream 2>&1 | tail -45 | tee $O/stream_trace.txt
HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 8 --streams 16 --clip-cache /tmp/hvq_clip_cache > $O/bench_8rank.json 2> $O/bench_8rank.err; tail -c 600 $O/bench_8rank.json; tail -3 $O/bench_8rank.err
HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 8 --workload c4 --clip-cache /tmp/hvq_clip_cache > $O/bench_8rank_c4.json 2> $O/bench_8rank_c4.err; tail -c 400 $O/bench_8rank_c4.json; tail -3 $O/bench_8rank_c4.err
