#!/bin/bash
T=${1:-r05i}; O=gpurun_out/$T; mkdir -p $O
for g in 0 1 0 1; do HVQM4_AMD_GRAPH=$g timeout -k 10 150 python tools/c4_share_ab.py 200 2>&1 | tail -1; done | tee $O/c4_graph_ab.txt
for n in 1 6; do
  HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus $n --steps 5 --warmup 1 --streams 64 --cpu-seconds 0 --no-sdk --no-verify > $O/ranks$n.json 2> $O/ranks$n.err || { echo "ranks $n failed"; tail -5 $O/ranks$n.err; continue; }
  python3 tools/rank_line.py $O/ranks$n.json $n | tee -a $O/ranks.txt
done
