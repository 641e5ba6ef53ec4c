"""Summarise tools/r04_parse_phases.sh: per exit stamp, counters of the probe launches (odd dispatches of the flat parse kernel) and
of the full launches behind them; differences between consecutive stamps = what a phase issues."""
import collections, csv, glob, re, sys
root = sys.argv[1]
names = {1: "trees", 13: "chains + decode wave", 12: "tags + lists (P/B) / expansions (I)", 3: "DC placed", 4: "run sums", 9: "entries",
         10: "emit count", 6: "scans", 0: "merge (full kernel)"}
rows = []
for d in sorted(glob.glob(root + "/e*/")):
    E = int(re.search(r"/e(\d+)/$", d).group(1))
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: continue
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        if "hvq_parse_kernel_t<true>" in r["Kernel_Name"]:
            disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = disp[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    probe, full = ids[0::2], ids[1::2]
    def mean(which, c):
        v = [disp[i][c] for i in which[2:]] or [disp[i][c] for i in which]     # skip the cold batches
        return sum(v) / len(v)
    ms = None
    try:
        t = [float(m.group(1)) for m in re.finditer(r"exit at stamp \d+: ([0-9.]+) ms", open(f"{root}/e{E}.err").read())]
        ms = sorted(t)[len(t) // 2] if t else None
    except OSError: pass
    rows.append((E, {c: mean(probe, c) for c in disp[ids[0]]}, {c: mean(full, c) for c in disp[ids[0]]}, ms))
order = [1, 13, 12, 3, 4, 9, 10, 6]
rows.sort(key=lambda r: order.index(r[0]) if r[0] in order else 99)
cs = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
print("per launch of 2048 pictures, millions of wave instructions; cumulative up to the stamp, and the step from the previous stamp")
print(f"{'up to':38s} {'ms':>6s} " + " ".join(f"{c[9:]:>9s}" for c in cs) + "   | step: " + " ".join(f"{c[9:]:>8s}" for c in cs))
prev = {c: 0.0 for c in cs}; full = None
for E, p, fu, ms in rows:
    full = fu
    print(f"{names.get(E, str(E)):38s} {ms if ms is not None else float('nan'):6.3f} " + " ".join(f"{p[c] / 1e6:9.1f}" for c in cs) + "   |       " +
          " ".join(f"{(p[c] - prev[c]) / 1e6:8.1f}" for c in cs))
    prev = p
if full:
    print(f"{'full kernel':38s} {'':6s} " + " ".join(f"{full[c] / 1e6:9.1f}" for c in cs) + "   |       " + " ".join(f"{(full[c] - prev[c]) / 1e6:8.1f}" for c in cs))
