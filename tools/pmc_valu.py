"""Vector-instruction bound of hvq_recon_kernel from tools/pmc_passes.sh output (passes p1, p2, p3 = trace): instructions per wave
and VALU busy fraction per dependency level and over the step.  usage: pmc_valu.py <passes dir> <out.json> [launches per step]"""
import collections, csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
nlev = int(sys.argv[3]) if len(sys.argv) > 3 else 7
def load(pdir):
    f = glob.glob(f'{root}/{pdir}/**/*counter_collection.csv', recursive=True)[0]
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'hvq_recon' in r['Kernel_Name']:
            by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    return [by[i] for i in sorted(by)][-nlev:]
def trace(pdir):
    f = glob.glob(f'{root}/{pdir}/**/*kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'hvq_recon' in r['Kernel_Name']]
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in rows][-nlev:]
d1, d2, t = load('p1'), load('p2'), trace('p3')
lv = []
for a, b, us in zip(d1, d2, t):
    w = a['SQ_WAVES']
    lv.append({"us": round(us, 1), "waves": int(w), "valu_per_wave": round(b['SQ_INSTS_VALU'] / w, 1), "salu_per_wave": round(b['SQ_INSTS_SALU'] / w, 1),
               "valu_busy": round(4 * a['SQ_ACTIVE_INST_VALU'] / (1024 * us * 2400), 3), "wait_frac": round(a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES'], 3)})
W = sum(x["waves"] for x in lv)
res = {"source": root, "levels": lv,
       "insts_per_wave": round(sum(x["valu_per_wave"] * x["waves"] for x in lv) / W, 1),
       "busy_frac": round(sum(x["valu_busy"] * x["us"] for x in lv) / sum(x["us"] for x in lv), 3),
       "note": "per wave = 64 blocks of each tile of its workgroup (the dense bench runs one tile per workgroup since round 4, round 3 ran two); busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x clocks at 2.4 GHz)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])
