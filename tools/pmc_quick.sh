#!/bin/bash
# GPU box: the two SQ counter passes + a kernel trace of the resident-descriptor bench (no calibration kernels, no traffic passes):
# instructions per wave, VALU busy, wait fractions, LDS activity of the reconstruction kernel.  usage: tools/pmc_quick.sh <tag> [bench args]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
CACHE=/tmp/hvq_clip_cache
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE "$@" > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --gen-workers 1 --clip-cache $CACHE $@"
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $B > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $B > $OUT/p2.json 2> $OUT/p2.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 -- $B > $OUT/p3.json 2> $OUT/p3.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p5 --pmc TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE -- $B > $OUT/p5.json 2> $OUT/p5.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p6 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_WAVE_DEP_WAIT -- $B > $OUT/p6.json 2> $OUT/p6.err
cd $GRAFT_REPO_ROOT
python3 tools/pmc_valu.py $OUT $OUT/pmc_valu.json 7 | head -c 1500

python3 - $OUT <<'PY'
import csv,glob,collections,sys
root=sys.argv[1]
for p in ['p1','p2','p5','p6']:
    f=glob.glob(f'{root}/{p}/**/*counter_collection.csv',recursive=True)
    if not f: print(p,'no data'); continue
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if 'hvq_recon' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    for k,v in sorted(agg.items()): print(p,k,round(v/n[k]),'avg per launch over',n[k])
PY
