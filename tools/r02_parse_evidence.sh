#!/bin/bash
# GPU box: evidence for the device entropy parse -- flat path against the chains on one box (dense, natural, realistic),
# phase times, instruction mix (PMC), the table-lookup microbenchmark.  usage: tools/r02_parse_evidence.sh <tag>
set -o pipefail
T=${1:-r02ah}
O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
cd $GRAFT_REPO_ROOT
for preset in dense natural realistic; do
  for flat in 1 0; do
    HVQM4_AMD_PARSE_FLAT=$flat HVQM4_AMD_PARSE_TIMING=1 timeout -k 10 500 python bench.py --steps 6 --warmup 2 --no-sdk --cpu-seconds 0 --preset $preset \
        > $O/bench_${preset}_flat$flat.json 2> $O/bench_${preset}_flat$flat.err || { tail -20 $O/bench_${preset}_flat$flat.err; exit 1; }
  done
done
HVQM4_AMD_PARSE_TIMING=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-sdk --cpu-seconds 0 --streams 1 --distinct 1 --no-verify > $O/lone.json 2> $O/lone.err || exit 1
hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/ubench/lut_chain.hip -o tools/ubench/lut_chain && timeout -k 5 120 tools/ubench/lut_chain > $O/lut_chain.txt 2>&1 || true
python3 - <<PY > $O/summary.txt
import json, re
O="$O"
print("Device entropy parse, one MI355X box, 2048 pictures 640x480 per launch (128 streams x GOP 16), HIP-event kernel time (min over 5 timed batches):")
for preset in ("dense","natural","realistic"):
    for flat in (1,0):
        j=json.load(open(f"{O}/bench_{preset}_flat{flat}.json")); e=j["end_to_end_gpu_parse"]
        err=open(f"{O}/bench_{preset}_flat{flat}.err").read()
        slow=[l for l in err.splitlines() if "slowest picture" in l][-1].split("kind:")[1].strip()
        print(f"  {preset:9s} {'flat path' if flat else 'chains   '}: parse kernel {e['parse_kernel_ms']:.3f} ms, end to end one batch {e['value']:.0f} Mpixel/s, streaming {e.get('streaming_value')} Mpixel/s (parse kernel streaming {e.get('streaming_parse_kernel_ms')} ms); slowest picture / last end  {slow}")
print()
for preset in ("dense","natural"):
    for flat in (1,0):
        err=open(f"{O}/bench_{preset}_flat{flat}.err").read().splitlines()
        ph=[l for l in err if "pictures (" in l][-3:]
        print(f"phase stamps, {preset}, {'flat path' if flat else 'chains'} (us since the picture's start, mean over the pictures of a kind):")
        for l in ph: print(" ", l.strip())
        print()
err=open(f"{O}/lone.err").read().splitlines()
print("a lone stream (16 pictures on an idle GPU), flat path:")
for l in [l for l in err if "pictures (" in l][-3:]: print(" ", l.strip())
print()
print("table-lookup chain microbenchmark (tools/ubench/lut_chain.hip): LDS read -> 64-bit shift -> LDS read, one wave per workgroup")
print(open(f"{O}/lut_chain.txt").read())
PY
cat $O/summary.txt | cut -c1-400 | head -20
tools/pmc_parse.sh $T/pmc --no-sdk > $O/pmc.txt 2>&1; head -20 $O/pmc.txt
