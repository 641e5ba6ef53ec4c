#!/bin/bash
# GPU box: the evidence run of a round -- GPU test suite, bench lines (dense + natural), rocprofv3 kernel stats of the
# same bench command.  usage: tools/final_profile.sh <tag>   (writes gpurun_out/<tag>_*)
set -o pipefail
T=${1:-final}
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/${T}_gpu_tests.log 2>&1 || { tail -20 $O/${T}_gpu_tests.log; exit 1; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1 || { tail -5 $O/${T}_smoke.log; exit 1; }
timeout -k 10 400 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -5 $O/${T}_bench.err; exit 1; }
timeout -k 10 400 python bench.py --preset natural > $O/${T}_bench_natural.json 2> $O/${T}_bench_natural.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-verify > $O/${T}_prof_bench.json 2> $O/${T}_prof.err || { tail -5 $O/${T}_prof.err; exit 1; }
find $O/${T}_prof -name "*kernel_stats.csv" -exec cp {} $O/${T}_kernel_stats.csv \;
tail -2 $O/${T}_gpu_tests.log
cat $O/${T}_kernel_stats.csv
