#!/bin/bash
# GPU box: six ranks on the one GPU (the pool's process guard allows 6 processes on a GPU; 8 were killed, profiles/r04_rank_rehearsal.txt)
T=${1:-r04l}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 120 python tools/sdk_loop.py 5 2>&1 | tail -1 | tee $O/sdk.txt
HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 6 --streams 16 --clip-cache /tmp/hvq_clip_cache > $O/bench_6rank.json 2> $O/bench_6rank.err; echo "rc $?"; tail -c 300 $O/bench_6rank.json; tail -3 $O/bench_6rank.err
HVQM4_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 6 --workload c4 --clip-cache /tmp/hvq_clip_cache > $O/bench_6rank_c4.json 2> $O/bench_6rank_c4.err; echo "rc $?"; tail -c 300 $O/bench_6rank_c4.json; tail -3 $O/bench_6rank_c4.err
