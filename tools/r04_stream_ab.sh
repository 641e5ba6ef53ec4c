#!/bin/bash
# streaming period over many batches under environment switches.  usage: tools/r04_stream_ab.sh <outdir> "ENV=.. ENV=.." ["..." ...]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O; cd $GRAFT_REPO_ROOT
k=0
for rep in 1 2; do for e in "$@"; do
  k=$((k+1))
  env $e HVQM4_BENCH_STREAM_BATCHES=${BATCHES:-60} HVQM4_AMD_FLUSH_TIMING=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-sdk --cpu-seconds 0 --no-verify > $O/r$k.json 2> $O/r$k.err
  python3 - "$e" $O/r$k.json $O/r$k.err <<'PY'
import json, re, sys, statistics
e=json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])["end_to_end_gpu_parse"]
ends=[float(m.group(1)) for m in re.finditer(r"launches queued ([\d.]+) ms", open(sys.argv[3]).read())]
per=[b-a for a,b in zip(ends,ends[1:]) if 3.5 < b-a < 12]
print(f"{sys.argv[1]:40s} period mean {e['streaming_ms_per_batch']:.2f} median {statistics.median(per):.2f} p90 {sorted(per)[int(len(per)*0.9)]:.2f} -> {e['streaming_value']:.0f} Mpx/s  calls {e['streaming_submit_end_begin_ms']}")
PY
done; done
