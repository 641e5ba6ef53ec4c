#!/bin/bash
T=${1:-r05h}; O=gpurun_out/$T; mkdir -p $O
A=$PWD/hvqm4_amd/abl
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched_path or table_div" > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
tools/r04_ab.sh $T "dense natural flat" 2 nt nont:HVQM4_AMD_LIB=$A/libhvq_nont.so
# HIP graph replay of the small batch (one GPU's share of config 4): with / without
for g in 0 1; do
HVQM4_AMD_GRAPH=$g python3 - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench
cl = bench.gen_clips(bench.c4_share_configs(), 8, "/tmp/hvq_clip_cache")
r = bench.c4_share_leg(0, 200, 20, 8, cl)
print("c4 share HVQM4_AMD_GRAPH=%s:" % os.environ["HVQM4_AMD_GRAPH"], r.get("us_per_step"), "us per step,", r.get("value"), "Mpixel/s", r.get("error", ""))
PY
done
# ranks sharing the one GPU: 1 rank and 6 ranks, 64 streams each: per-rank streaming rate and host copy rate
for n in 1 6; do
  HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus $n --steps 5 --warmup 1 --streams 64 --cpu-seconds 0 --no-sdk > $O/ranks$n.json 2> $O/ranks$n.err || { tail -5 $O/ranks$n.err; }
  python3 - $O/ranks$n.json $n <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); e=j["end_to_end_gpu_parse"]
print("ranks", sys.argv[2], "value", j["value"], "streaming sum", e["streaming_value"], "min/max rank", e["streaming_value_min_rank"], e["streaming_value_max_rank"],
      "host_copy_GBs rank0", e["host_copy_GBs"], "all", e["host_copy_GBs_all_ranks"], "zero copy", e["streaming_zero_copy"]["value"], "calls", e["streaming_submit_end_begin_ms"], e["streaming_zero_copy"]["submit_end_begin_ms"], e["affinity"])
PY
done
