#!/bin/bash
# GPU box: randomized parity sweeps of the round-4 build in its modes
O=gpurun_out/${1:-r04s}; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 500 python tools/parity_sweep.py $N $SEED both > $O/$name.txt 2>&1; echo "$name: $(tail -1 $O/$name.txt)"; }
N=3000 SEED=4004; N=$N SEED=$SEED; run default X=1
N=1000; SEED=4005; run pair_cap20 HVQM4_AMD_PAIR_CAP=20
SEED=4006; run pool_cap24 HVQM4_AMD_POOL_CAP=24
SEED=4007; run tpw1 HVQM4_AMD_TILES_PER_WG=1
SEED=4008; run tpw2 HVQM4_AMD_TILES_PER_WG=2
SEED=4009; run two_pass HVQM4_AMD_TILE_QUEUES=1
