#!/bin/bash
# GPU box: the host side of the streaming path with several ranks on the ONE GPU (HVQM4_BENCH_SHARE_GPU=1): 1 rank and 6 ranks (the pool's
# process guard admits no more), 64 streams each -- per-rank submit time, copy rate, zero-copy streaming.  usage: tools/rank_rehearsal.sh <tag>
T=${1:-ranks}; O=gpurun_out/$T; mkdir -p $O
for n in 1 6; do
  HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus $n --steps 5 --warmup 1 --streams 64 --cpu-seconds 0 --no-sdk --no-verify > $O/ranks$n.json 2> $O/ranks$n.err || { echo "ranks $n failed"; tail -5 $O/ranks$n.err; continue; }
  python3 tools/rank_line.py $O/ranks$n.json $n | tee -a $O/ranks.txt
done
