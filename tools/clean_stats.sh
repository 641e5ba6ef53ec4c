#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of the bench's TIMED leg alone (resident-descriptor replays, no SDK / streaming / two-queue
# legs on the same kernel), so that the profiler's average launch duration can be held against roofline.avg_launch_us.
# usage: tools/clean_stats.sh <tag>
T=$1
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
CACHE=/tmp/hvq_clip_cache
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/clean -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-verify --no-sdk --no-gpu-parse --gen-workers 1 --clip-cache $CACHE > $O/clean_bench.json 2> $O/clean.err || { tail -5 $O/clean.err; exit 1; }
find $O/clean -name "*kernel_stats.csv" -exec cp {} $O/clean_kernel_stats.csv \;
cat $O/clean_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/trace_levels.py $(find $O/clean -name "*kernel_trace.csv" | head -1) $O/clean_bench.json | tee $O/clean_levels.txt
