#!/bin/bash
# GPU box: randomized parity sweeps of the build in its modes (host-parsed AND GPU-parsed clips, tools/parity_sweep.py)
O=gpurun_out/${1:-sweep}; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 1000 python tools/parity_sweep.py $N $SEED both > $O/$name.txt 2>&1; echo "$name: $(tail -1 $O/$name.txt)"; }
# usage: tools/sweep.sh <tag> [seed base (default 5104)] [clips of the default mode (2000)] [clips of each other mode (700)]
B=${2:-5104}; N0=${3:-2000}; N1=${4:-700}
N=$N0; SEED=$B; run default X=1
N=$N1; SEED=$((B+1)); run pair_cap20 HVQM4_AMD_PAIR_CAP=20
SEED=$((B+2)); run pool_cap24 HVQM4_AMD_POOL_CAP=24
SEED=$((B+3)); run tpw1 HVQM4_AMD_TILES_PER_WG=1
SEED=$((B+4)); run tpw2 HVQM4_AMD_TILES_PER_WG=2
SEED=$((B+6)); run two_queues HVQM4_AMD_QUEUES=2
