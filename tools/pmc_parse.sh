#!/bin/bash
# GPU box: SQ counter passes over the device entropy parse kernel (hvq_parse_kernel_t) of the bench's streaming leg: instructions per launch by
# kind, busy and wait cycles.  usage: tools/pmc_parse.sh <tag> [bench args]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
CACHE=/tmp/hvq_clip_cache
export HVQM4_BENCH_STREAM_BATCHES=6 HVQM4_BENCH_STREAM_CONTEXTS=1
python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE "$@" > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE $@"
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- $B > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -- $B > $OUT/p2.json 2> $OUT/p2.err
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, glob, collections, sys
root = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(f'{root}/p1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hvq_parse_kernel' in r['Kernel_Name']:
            dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in dur.items():
    v = sorted(v)
    print(f"{k}: {len(v)} launches, median {v[len(v)//2]:.3f} ms under the counters")
for p in ('p1', 'p2'):
    f = glob.glob(f'{root}/{p}/**/*counter_collection.csv', recursive=True)
    if not f:
        print(p, 'no data'); continue
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if 'hvq_parse_kernel_t<true>' in r['Kernel_Name'] or 'hvq_parse_kernel_tILb1' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k, v in sorted(agg.items()):
        print(f"{p} {k:24s} {v / n[k]:16.0f} per launch (mean of {n[k]})")
PY
