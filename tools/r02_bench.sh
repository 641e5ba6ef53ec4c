#!/bin/bash
# GPU box: the bench lines of the round: default (c5), c4, 2 self-started ranks sharing the GPU (rehearsal of --gpus N)
set -o pipefail
T=${1:-r02g}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "default ok"; python -c "import json;j=json.load(open('$O/bench.json'));print(j['value'],j['roofline']['frac'],j['cpu_baseline']['value'],j['end_to_end_gpu_parse']['streaming_value'], j.get('sdk_path'))"
timeout -k 10 500 python bench.py --workload c4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err || { tail -20 $O/bench_c4.err; exit 1; }
echo "c4 ok"; python -c "import json;j=json.load(open('$O/bench_c4.json'));print(j['value'],j['roofline']['frac'],j['config']['pictures_per_step'], j['end_to_end_gpu_parse']['streaming_value'])"
HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 1 --streams 64 --cpu-seconds 0 > $O/bench_2rank.json 2> $O/bench_2rank.err || { tail -20 $O/bench_2rank.err; exit 1; }
echo "2-rank ok"; python -c "import json;j=json.load(open('$O/bench_2rank.json'));print(j['n_gpus'],j['value'],j['end_to_end_gpu_parse']['streaming_value'],j['end_to_end_gpu_parse']['streaming_value_min_rank'])"
