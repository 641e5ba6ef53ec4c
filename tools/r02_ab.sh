#!/bin/bash
# GPU box: GPU test suite on the shipped build, then A/B of library variants on the bench workload (same box, same
# process order), optionally the phase timeline of the -DHVQ_STAMPS build.
# usage: tools/r02_ab.sh <tag> "<presets>" <variant names...>     (variant "ship" = hvqm4_amd/libhvqm4_amd.so)
set -o pipefail
T=$1; P=$2; shift 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T; mkdir -p $O
B="--cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk"
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
  tail -2 $O/gpu_tests.log
fi
for p in $P; do
  for v in "$@"; do
    L=$PWD/hvqm4_amd/abl/libhvq_$v.so; [ $v = ship ] && L=$PWD/hvqm4_amd/libhvqm4_amd.so
    if [ $v = stamps ]; then
      HVQM4_AMD_LIB=$L HVQM4_AMD_STAMPS=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 $B --preset $p > $O/${v}_$p.json 2> $O/${v}_$p.err || { tail -5 $O/${v}_$p.err; exit 1; }
      grep -h "^stamps" $O/${v}_$p.err | tail -7 > $O/stamps_$p.txt
    else
      HVQM4_AMD_LIB=$L timeout -k 10 300 python bench.py --steps 10 --warmup 2 $B --preset $p > $O/${v}_$p.json 2> $O/${v}_$p.err || { tail -5 $O/${v}_$p.err; exit 1; }
    fi
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/*_*.json')):
    try: j=json.load(open(f)); print('%-40s %10.0f Mpx/s  frac %.4f  launch %.1f us' % (f.split('/')[-1], j['value'], j['roofline']['frac'], j['roofline']['avg_launch_us']))
    except Exception as e: print(f, 'unreadable', e)
PY
cat $O/stamps_*.txt 2>/dev/null
