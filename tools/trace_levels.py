#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of bench.py -> the reconstruction launches grouped by grid (= dependency level and launch queue), and the
mean over the launches of the timed passes, which is what bench.py's roofline.avg_launch_us states from its HIP events: the average
duration of ONE launch, the launches of the two queues running side by side (a step lasts launches / queues of them).  The default
bench command also runs other legs on the same kernel (one-picture SDK calls, the streaming passes), which a plain --stats average
mixes in.   usage: tools/trace_levels.py <kernel_trace.csv> [bench.json]"""
import csv, json, sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hvq_recon" in r["Kernel_Name"]]
by = defaultdict(list)
for r in rows:
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    by[g].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
print("%-22s %6s %10s %10s %10s  queues" % ("grid (x, y, z) blocks", "calls", "mean us", "min us", "max us"))
big = {}
for g, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    d = [(e - s) / 1e3 for s, e, _ in v]
    print("%-22s %6d %10.2f %10.2f %10.2f  %s" % (g, len(d), sum(d) / len(d), min(d), max(d), sorted({q for _, _, q in v})))
    if g[0] * g[2] >= 64 and len(d) >= 8:
        big[g] = d
# the timed passes: grids of at least 64 picture slots (a queue's launch of the 128-stream batch holds 64, 192 or 128; the SDK leg one)
n = sum(len(d) for d in big.values()); tot = sum(sum(d) for d in big.values())
print("launches with >= 64 picture slots: %d, mean %.2f us" % (n, tot / n if n else 0))
if len(sys.argv) > 2:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("bench.py roofline.avg_launch_us (HIP events over the timed region): %.2f" % j["roofline"]["avg_launch_us"])
