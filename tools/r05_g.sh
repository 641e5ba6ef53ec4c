#!/bin/bash
# GPU box: the default bench line end to end (every leg), to see the new legs work
T=${1:-r05g}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python3 - $O/bench.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print("value",j["value"],"frac",j["roofline"]["frac"],"ms/step",j["ms_per_step"])
e=j["end_to_end_gpu_parse"]
for k in ("value","streaming_value","streaming_ms_per_batch","streaming_ms_per_batch_median","streaming_submit_end_begin_ms","streaming_parse_kernel_ms","parse_kernel_ms","host_copy_GBs","host_copy_GBs_all_ranks","streaming_zero_copy","affinity"): print(k, e.get(k))
for k in ("c4_share","sdk_path","c5_staggered","two_pass_tile_queues","rgb_epilogue","cpu_baseline"): print(k, json.dumps(j.get(k))[:400])
PY
