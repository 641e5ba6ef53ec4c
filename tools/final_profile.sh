#!/bin/bash
# GPU box: the evidence run of a round -- GPU test suite, smoke, bench lines (default dense, natural, flat, c4, 2 ranks on the
# one GPU), rocprofv3 kernel stats of the default bench command.  usage: tools/final_profile.sh <tag>   (writes gpurun_out/<tag>_*)
set -o pipefail
T=${1:-final}
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1 || { tail -20 $O/${T}_gpu_tests.log; exit 1; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1 || { tail -5 $O/${T}_smoke.log; exit 1; }
timeout -k 10 500 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -5 $O/${T}_bench.err; exit 1; }
for p in natural realistic flat; do
  timeout -k 10 400 python bench.py --preset $p --cpu-seconds 0 --no-sdk > $O/${T}_bench_$p.json 2> $O/${T}_bench_$p.err || exit 1
done
timeout -k 10 500 python bench.py --workload c4 --steps 5 --warmup 1 --cpu-seconds 0 --no-sdk > $O/${T}_bench_c4.json 2> $O/${T}_bench_c4.err || exit 1
HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 1 --streams 64 --cpu-seconds 0 > $O/${T}_bench_2rank.json 2> $O/${T}_bench_2rank.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-verify --gen-workers 1 --distinct 8 > $O/${T}_prof_bench.json 2> $O/${T}_prof.err || { tail -5 $O/${T}_prof.err; exit 1; }
find $O/${T}_prof -name "*kernel_stats.csv" -exec cp {} $O/${T}_kernel_stats.csv \;
tail -2 $O/${T}_gpu_tests.log
cat $O/${T}_kernel_stats.csv
python3 - <<PY
import json
for n in ("bench","bench_natural","bench_realistic","bench_flat","bench_c4","bench_2rank"):
    j=json.load(open("$O/${T}_%s.json"%n)); e=j.get("end_to_end_gpu_parse") or {}
    print("%-16s value %9.0f frac %.4f n_gpus %d  e2e streaming %s"%(n, j["value"], j["roofline"]["frac"], j["n_gpus"], e.get("streaming_value")))
PY
