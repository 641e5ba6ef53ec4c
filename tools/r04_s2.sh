#!/bin/bash
# GPU box, round 4 session 2: inline-queue reconstruction kernel vs the tile-queue pair of kernels, then the GPU test suite
T=${1:-r04b}; O=gpurun_out/$T; mkdir -p $O
C="--clip-cache /tmp/hvq_clip_cache --no-sdk --no-gpu-parse --cpu-seconds 0"
line() { python3 - "$1" "$2" <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%-26s value %8.0f stage frac %.4f (%.1f us) | recon only %.4f (%.1f us) | queue build %.1f us" % (sys.argv[2], d["value"], r["frac"], r["stage_us_per_step"],
          r["recon_only"]["frac"], r["recon_only"]["us_per_step"], r["queue_build"]["us_per_step"]))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run() { # name, env..., -- bench args
  name=$1; shift
  env "$@" timeout -k 10 240 python bench.py $C $BARGS > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; return 1; }
  line $O/$name.json $name
}
for p in dense natural flat; do
  BARGS="--preset $p"
  run ${p}_inline_auto X=1 && run ${p}_inline_tpw1 HVQM4_AMD_TILES_PER_WG=1 && run ${p}_inline_tpw2 HVQM4_AMD_TILES_PER_WG=2 && run ${p}_tileq HVQM4_AMD_TILE_QUEUES=1 || exit 1
done
BARGS="--workload c4" run c4_inline X=1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
