#!/bin/bash
# GPU box: what the flat parse kernel costs up to each of its stamps -- the probe build (tools/variant.sh probe -DGP_PROBE) runs the
# kernel in front of every real parse launch and lets every workgroup leave at HVQM4_AMD_PARSE_EXIT; the time of that launch is the
# cumulative cost of everything up to the stamp (2048 dense pictures, streaming loop of tools/overlap_probe.py).
# HVQM4_AMD_PARSE_EXIT = stamp | skip << 8 (1 decode waves, 2 type/proc chains, 4 vector chains)
# usage: tools/parse_exit_times.sh <tag> [exit words...]
tag=${1:-pet}; shift
out=gpurun_out/$tag; mkdir -p $out
words=${@:-1 13 269 1549 1805 12 3 4 9 10 6}
for e in $words; do
  HVQM4_AMD_PARSE_EXIT=$e HVQM4_AMD_LIB=$PWD/hvqm4_amd/abl/libhvq_${PROBE_LIB:-probe}.so timeout -k 10 120 python tools/overlap_probe.py 8 1 128 > $out/e$e.txt 2>&1 || { tail -3 $out/e$e.txt; exit 1; }
  python3 - $out/e$e.txt $e <<'PY'
import re, sys
t = sorted(float(m.group(1)) for m in re.finditer(r"exit at stamp \d+: ([0-9.]+) ms", open(sys.argv[1]).read()))
e = int(sys.argv[2])
print("exit %5d (stamp %2d skip %d): median %.3f ms over %d launches" % (e, e & 255, e >> 8, t[len(t) // 2] if t else float("nan"), len(t)))
PY
done
