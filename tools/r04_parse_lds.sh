#!/bin/bash
# LDS counters of the parse kernel (probe build: up to a stamp, and the full kernel).  usage: tools/r04_parse_lds.sh <outdir> [stamps...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
export HVQM4_AMD_LIB=$GRAFT_REPO_ROOT/hvqm4_amd/abl/libhvq_probe.so
CACHE=/tmp/hvq_clip_cache
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --clip-cache $CACHE > $OUT/p0.json 2> $OUT/p0.err
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-sdk --gen-workers 1 --clip-cache $CACHE"
for E in ${@:-1 13}; do
  HVQM4_AMD_PARSE_EXIT=$E timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/e$E --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $B > $OUT/e$E.json 2> $OUT/e$E.err || { tail -3 $OUT/e$E.err; exit 1; }
done
python3 - <<PY
import collections, csv, glob, re
for d in sorted(glob.glob("$OUT/e*/")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        if "hvq_parse_kernel_t<true>" in r["Kernel_Name"]:
            disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = disp[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    for name, which in (("probe", ids[0::2][2:]), ("full", ids[1::2][2:])):
        if not which: continue
        print(d.rstrip("/").split("/")[-1], name, {c: round(sum(disp[i][c] for i in which) / len(which) / 1e6, 1) for c in disp[ids[0]]})
PY
