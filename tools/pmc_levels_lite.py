"""Per-dependency-level summary of hvq_recon_kernel from tools/r03_levels.sh output (p1, p2 = PMC passes, t = plain kernel trace).
The levels of a step repeat every N launches; the LAST step of each run is used."""
import csv, glob, collections, sys
root = sys.argv[1]; nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 7
def load(pdir):
    f = glob.glob(f'{root}/{pdir}/**/*counter_collection.csv', recursive=True)[0]
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'hvq_recon' in r['Kernel_Name']:
            by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    return [by[i] for i in sorted(by)]
def trace(pdir):
    f = glob.glob(f'{root}/{pdir}/**/*kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'hvq_recon' in r['Kernel_Name']]
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in rows]
d1, d2, t = load('p1'), load('p2'), trace('t')
try:
    d3 = load('p3')
except Exception:
    d3 = None
tot = 0
for lvl in range(nlev):
    i = len(t) - nlev + lvl
    a, b = d1[len(d1) - nlev + lvl], d2[len(d2) - nlev + lvl]
    w = a['SQ_WAVES']
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs: busy = 4 * count / (1024 SIMDs * kernel clocks at 2.4 GHz)
    busy = 4 * a['SQ_ACTIVE_INST_VALU'] / (1024 * t[i] * 2400)
    tot += t[i]
    print(f"L{lvl}: waves {w:8.0f} us {t[i]:7.1f} VALU/w {b['SQ_INSTS_VALU']/w:5.0f} SALU/w {b['SQ_INSTS_SALU']/w:4.0f} LDS/w {b['SQ_INSTS_LDS']/w:4.0f} "
          f"VMRD/w {b['SQ_INSTS_VMEM_RD']/w:4.1f} VMWR/w {b['SQ_INSTS_VMEM_WR']/w:4.1f} SMEM/w {b['SQ_INSTS_SMEM']/w:4.1f} cyc/w {a['SQ_WAVE_CYCLES']*4/w:6.0f} "
          f"wait% {100*a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES']:3.0f} valu_busy {busy:.2f} ldsconf% {100*b['SQ_LDS_BANK_CONFLICT']/max(1,b['SQ_LDS_IDX_ACTIVE']):3.0f}", end="")
    if d3:
        c = d3[len(d3) - nlev + lvl]
        # GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs (profiles/r04_ta_calibration.txt: GRBM / (us x 2400) = 8.0 on every
        # kernel); round 3 divided by it as if it were one clock and printed TA busy 8x too small (0.03-0.10 for what is 0.3-0.8)
        clk = c['GRBM_GUI_ACTIVE'] / 8.0 if c.get('GRBM_GUI_ACTIVE') else t[i] * 2400
        # TA_*_sum: summed over the 256 texture addressers (one per CU)
        print(f" | TA busy {c['TA_TA_BUSY_sum']/(256*clk):.2f} rd wavefronts/w {c['TA_FLAT_READ_WAVEFRONTS_sum']/w:5.1f} L1 accesses/w {c['TCP_TOTAL_CACHE_ACCESSES_sum']/w:6.0f} "
              f"TCP pending-stall {c['TCP_PENDING_STALL_CYCLES_sum']/(256*clk):.2f}")
    else:
        print()
print("step total us %.1f" % tot)
