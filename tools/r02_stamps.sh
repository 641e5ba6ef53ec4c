#!/bin/bash
# GPU box: phase timeline of hvq_recon_kernel (diagnostic -DHVQ_STAMPS build) next to the shipped build on the same box.
# usage: tools/r02_stamps.sh <tag> [presets...]
set -o pipefail
T=${1:-r02a}; shift
P=${@:-dense}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T; mkdir -p $O
B="--cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk"
for p in $P; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 $B --preset $p > $O/base_$p.json 2> $O/base_$p.err || { tail -5 $O/base_$p.err; exit 1; }
  HVQM4_AMD_LIB=$PWD/hvqm4_amd/abl/libhvq_stamps.so HVQM4_AMD_STAMPS=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 $B --preset $p > $O/stamps_$p.json 2> $O/stamps_$p.err || { tail -5 $O/stamps_$p.err; exit 1; }
  grep -h "^stamps" $O/stamps_$p.err | tail -7 > $O/stamps_$p.txt
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/base_*.json')):
    j=json.load(open(f)); print(f, j['value'], j['roofline']['frac'])
PY
cat $O/stamps_*.txt
