#!/bin/bash
# GPU box: randomized parity sweeps of the build in its modes (host-parsed AND GPU-parsed clips, tools/parity_sweep.py)
O=gpurun_out/${1:-sweep}; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 500 python tools/parity_sweep.py $N $SEED both > $O/$name.txt 2>&1; echo "$name: $(tail -1 $O/$name.txt)"; }
N=2000; SEED=5104; run default X=1
N=700; SEED=5105; run pair_cap20 HVQM4_AMD_PAIR_CAP=20
SEED=5106; run pool_cap24 HVQM4_AMD_POOL_CAP=24
SEED=5107; run tpw1 HVQM4_AMD_TILES_PER_WG=1
SEED=5108; run tpw2 HVQM4_AMD_TILES_PER_WG=2
SEED=5109; run two_pass HVQM4_AMD_TILE_QUEUES=1
