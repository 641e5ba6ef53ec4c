#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels in a HIP source: tools/isa_count.py <file.hip> [name filter] [-D...]
(hipcc -S --cuda-device-only; counts per kernel: VALU, SALU, VMEM, LDS (ds_*), SMEM, barriers, v_cndmask, scratch)."""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
out = "/tmp/isa_count.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-fvisibility=hidden", "-S", "--cuda-device-only",
                       "-I" + os.path.dirname(os.path.abspath(src)), src, "-o", out] + extra, stderr=subprocess.DEVNULL)
cur = None
stats = {}
for line in open(out):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1); stats[cur] = dict(valu=0, salu=0, vmem=0, lds=0, smem=0, barrier=0, cndmask=0, scratch=0, dpp=0)
        continue
    if cur is None: continue
    if line.startswith("\t.") or line.startswith("."):
        if ".end_amdhsa_kernel" in line or line.startswith("\t.section"): pass
        continue
    t = line.strip().split()
    if not t or t[0].startswith(";"): continue
    op = t[0]
    st = stats[cur]
    if op == "s_endpgm": continue
    if op.startswith("v_"):
        st["valu"] += 1
        if "cndmask" in op: st["cndmask"] += 1
        if "dpp" in line: st["dpp"] += 1
    elif op.startswith("s_barrier"): st["barrier"] += 1
    elif op.startswith("s_load") or op.startswith("s_buffer") or op.startswith("s_memtime"): st["smem"] += 1
    elif op.startswith("s_"): st["salu"] += 1
    elif op.startswith("ds_"): st["lds"] += 1
    elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): st["vmem"] += 1
    elif op.startswith("scratch_"): st["scratch"] += 1
for k, v in stats.items():
    if flt in k and sum(v.values()):
        print(k[:60].ljust(60), " ".join(f"{a}={b}" for a, b in v.items()))
