#!/bin/bash
# kernel + memory-copy trace of the bench (streaming leg at the end): where the GPU idles between batches
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# clips are generated once, OUTSIDE the profiler: a profiled run must start no worker processes (the profiler's preload has
# initialised the GPU in the parent; see tools/pmc_passes.sh)
CACHE=/tmp/hvq_clip_cache
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE $BENCH_ARGS > /dev/null 2>&1 || true
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --gen-workers 1 --clip-cache $CACHE --no-verify --no-sdk > $O/bench.json 2> $O/err.txt
python3 - <<PY
import csv, glob
O="$O"
ev=[]
for f in glob.glob(O+"/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28]))
for f in glob.glob(O+"/t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy "+r.get("Direction","")[:24]+" "+r.get("Bytes", r.get("Size",""))))
ev.sort()
# the last 10 parse kernels and what lies between them
idx=[i for i,e in enumerate(ev) if e[2].startswith("hvq_parse_kernel")]
lo=idx[max(0, len(idx) - 8)]
t0=ev[lo][0]
for s,e,n in ev[lo:]:
    if (e-s) > 200000 or n.startswith("hvq_parse"):
        print("%9.3f -> %9.3f ms (%7.3f)  %s" % ((s-t0)/1e6, (e-t0)/1e6, (e-s)/1e6, n))
PY
