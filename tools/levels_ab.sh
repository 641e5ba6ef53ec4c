#!/bin/bash
# clips are generated once, OUTSIDE the profiler: a profiled run must start no worker processes (the profiler's preload has
# initialised the GPU in the parent; see tools/pmc_passes.sh)
CACHE=/tmp/hvq_clip_cache
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE $BENCH_ARGS > /dev/null 2>&1 || true
# GPU box: per-dependency-level launch times (rocprofv3 kernel trace) of the bench workload for settings of one environment
# variable.  usage: tools/levels_ab.sh <tag> <VAR> <value>...
T=$1; V=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  env_line="$V=$t"
  export $V=$t
  rocprofv3 --kernel-trace --output-format csv -d $O/t$t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --gen-workers 1 --clip-cache $CACHE --no-verify --no-gpu-parse --no-sdk $BENCH_ARGS > $O/t$t.json 2> $O/t$t.err || { tail -3 $O/t$t.err; exit 1; }
  python3 - <<PY
import csv,glob,json
f=glob.glob("$O/t$t/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "hvq_recon" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000 for r in rows]
n=7; lv=[sum(d[n+i::n])/len(d[n+i::n]) for i in range(n)]
print("$env_line  levels us:", " ".join("%.1f"%x for x in lv), " step %.1f"%sum(lv), " value", json.load(open("$O/t$t.json"))["value"])
PY
done
