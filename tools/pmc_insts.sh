#!/bin/bash
# GPU box: vector / scalar / LDS instructions per wave of the reconstruction kernel for library variants (one counter pass each).
# usage: tools/pmc_insts.sh <tag> base|<variant>...   (variants: hvqm4_amd/abl/libhvq_<variant>.so)
T=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $OUT
CACHE=/tmp/hvq_clip_cache
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE $BENCH_EXTRA > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/hvqm4_amd/abl/libhvq_$v.so; [ $v = base ] && L=$GRAFT_REPO_ROOT/hvqm4_amd/libhvqm4_amd.so
  export HVQM4_AMD_LIB=$L
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$v --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --gen-workers 1 --clip-cache $CACHE $BENCH_EXTRA > $OUT/$v.json 2> $OUT/$v.err
  python3 - $OUT/$v $v <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
by=collections.defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    if 'hvq_recon' in r['Kernel_Name']: by[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
rows=[by[i] for i in sorted(by)][-7:]
def pw(rows,k): 
    w=sum(r['SQ_WAVES'] for r in rows); return sum(r[k] for r in rows)/w
print("%-8s I level: VALU %.0f SALU %.0f LDS %.1f VMEM_RD %.1f | P/B levels: VALU %.0f SALU %.0f LDS %.1f VMEM_RD %.1f SMEM %.1f" % (sys.argv[2],
      pw(rows[:1],'SQ_INSTS_VALU'),pw(rows[:1],'SQ_INSTS_SALU'),pw(rows[:1],'SQ_INSTS_LDS'),pw(rows[:1],'SQ_INSTS_VMEM_RD'),
      pw(rows[1:],'SQ_INSTS_VALU'),pw(rows[1:],'SQ_INSTS_SALU'),pw(rows[1:],'SQ_INSTS_LDS'),pw(rows[1:],'SQ_INSTS_VMEM_RD'),pw(rows[1:],'SQ_INSTS_SMEM')))
PY
done
