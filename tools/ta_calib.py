#!/usr/bin/env python3
"""Summary of tools/r04_ta_calib.sh: per kernel of the microbenchmarks, TA_TA_BUSY_sum under both normalisations."""
import csv, glob, collections, sys
root = sys.argv[1]
CLK_MHZ = 2400.0
for b in ("gather_rate", "copy_rate"):
    try:
        f = glob.glob(f"{root}/{b}/p/**/*counter_collection.csv", recursive=True)[0]
    except IndexError:
        print(f"{b}: no counter file"); continue
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for k in ("Start_Timestamp", "End_Timestamp"):
            if k in r and r[k]: d[k] = int(r[k])
    tr = {}
    for tf in glob.glob(f"{root}/{b}/p/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(tf)):
            tr[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    print(f"== {b}: dispatch, kernel, us (profiled run), GRBM_GUI_ACTIVE, GRBM / (us x 2400), TA_TA_BUSY_sum / (256 x us x 2400), / (256 x GRBM), [TA_BUSY_avr / GRBM]")
    for i, d in by.items():
        us = tr.get(i)
        if us is None and "Start_Timestamp" in d: us = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1000.0
        g = d.get("GRBM_GUI_ACTIVE", 0.0); ta = d.get("TA_TA_BUSY_sum", 0.0)
        if not us or us < 20: continue
        clk = us * CLK_MHZ
        extra = f"  avr/GRBM {d['TA_BUSY_avr'] / g:.3f} max/GRBM {d.get('TA_BUSY_max', 0) / g:.3f}" if d.get("TA_BUSY_avr") and g else ""
        print(f"{i:4d} {d['name'][:44]:44s} {us:9.1f} us  GRBM {g:12.0f}  GRBM/clk {g / clk:5.2f}  TA/(256 clk) {ta / (256 * clk):.3f}  TA/(256 GRBM) {ta / (256 * g) if g else 0:.3f}{extra}")
