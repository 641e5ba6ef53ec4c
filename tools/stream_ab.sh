#!/bin/bash
# GPU box: the streaming leg of the default bench (hvq_flush_next), its zero-copy variant and the plain hvq_flush_end /
# hvq_flush_begin loop of the same run, `reps` times.  usage: tools/stream_ab.sh <tag> [reps] [extra bench args]
set -o pipefail
tag=${1:-sab}; reps=${2:-2}; shift 2 || true
out=gpurun_out/$tag; mkdir -p $out
for r in $(seq 1 $reps); do
  for ps in 1; do
    timeout -k 10 400 python bench.py --no-sdk --cpu-seconds 0 --clip-cache /tmp/hvq_clip_cache "$@" > $out/ps${ps}_$r.json 2> $out/ps${ps}_$r.err || { tail -5 $out/ps${ps}_$r.err; exit 1; }
    python - $out/ps${ps}_$r.json "rep $r" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["end_to_end_gpu_parse"]
print("%-24s streaming %8.1f (%.2f ms, median %.2f) | zero copy %8.1f | plain pair %8.1f (%.2f ms) | two contexts %8.1f | parse kernel %.3f ms | recon frac %.4f" % (
    sys.argv[2], e["streaming_value"], e["streaming_ms_per_batch"], e["streaming_ms_per_batch_median"], e["streaming_zero_copy"]["value"],
    e["streaming_plain_pair"]["value"], e["streaming_plain_pair"]["ms_per_batch"], (e.get("streaming_two_contexts") or {}).get("value", 0.0), e["streaming_parse_kernel_ms"], d["roofline"]["frac"]))
PY
  done
done
