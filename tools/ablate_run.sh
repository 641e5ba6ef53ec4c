#!/bin/bash
# GPU box: time each ablation variant per dependency level with rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export HVQM4_AMD_LIB=$GRAFT_REPO_ROOT/hvqm4_amd/abl/libhvq_abl$v.so
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --distinct 4 > $GRAFT_REPO_ROOT/gpurun_out/abl_$v.json 2> $GRAFT_REPO_ROOT/gpurun_out/abl_$v.err || exit 1
  python3 - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/abl_$v/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'hvq_recon' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in rows]
n=7
lv=[sum(d[n+i::n])/len(d[n+i::n]) for i in range(n)]
print('abl $v: '+' '.join('%6.1f'%x for x in lv)+'  | step %.1f us'%sum(lv))
PY
done
