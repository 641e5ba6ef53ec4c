"""Per-dependency-level PMC breakdown from tools/pmc_passes.sh output (levels repeat every N launches)."""
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
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in rows], rows
d1, d2, d3, d4 = load('p1'), load('p2'), load('p3'), load('p4')
t, rows = trace('p3')
for lvl in range(nlev):
    i = nlev + lvl
    a, b, c, e = d1[i], d2[i], d3[i], d4[i]
    w = a['SQ_WAVES']
    print(f"L{lvl}: WGs {(int(rows[i]['Grid_Size_X'])*int(rows[i].get('Grid_Size_Y',1) or 1)*int(rows[i].get('Grid_Size_Z',1) or 1))//256:6d} us {t[i]:7.1f} VALU/w {b['SQ_INSTS_VALU']/w:5.0f} SALU/w {b['SQ_INSTS_SALU']/w:4.0f} "
          f"LDS/w {b['SQ_INSTS_LDS']/w:4.0f} VMRD/w {b['SQ_INSTS_VMEM_RD']/w:4.0f} VMWR/w {b['SQ_INSTS_VMEM_WR']/w:4.1f} SMEM/w {b['SQ_INSTS_SMEM']/w:4.1f} "
          f"cyc/w {a['SQ_WAVE_CYCLES']*4/w:6.0f} wait% {100*a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES']:3.0f} issuewait% {100*a['SQ_WAIT_INST_ANY']/a['SQ_WAVE_CYCLES']:3.0f} "
          f"busy_cyc {a['SQ_BUSY_CYCLES']:.0f} FETCH {c['FETCH_SIZE']/1024:5.0f}MB WRITE {e['WRITE_SIZE']/1024:5.0f}MB L2hit {100*e['TCC_HIT_sum']/(e['TCC_HIT_sum']+e['TCC_MISS_sum']):3.0f}%")
print("step total us %.1f" % sum(t[nlev:2 * nlev]))
