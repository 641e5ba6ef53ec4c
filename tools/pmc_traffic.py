"""HBM traffic per launch of hvq_recon_kernel from tools/pmc_passes.sh output, calibrated as MI355X_MICROARCH.md ("HBM")
prescribes: FETCH_SIZE tallies every read request at 64 B, so the read side is rebuilt from the size-split request counters
(TCC_EA0_RDREQ_32B/_64B/_128B), and both that rule and WRITE_SIZE are checked on kernels that move a known number of bytes
in this kernel's access shapes (tools/ubench/pmc_calib.hip, same passes).
usage: pmc_traffic.py <passes dir> <out.json> [launches per step] [algorithmic bytes per launch]"""
import collections, csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
nlev = int(sys.argv[3]) if len(sys.argv) > 3 else 7
algo = float(sys.argv[4]) if len(sys.argv) > 4 else None


def per_dispatch(pdir, kernel, names):
    f = glob.glob(f'{root}/{pdir}/**/*counter_collection.csv', recursive=True)[0]
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    order = {}
    for r in csv.DictReader(open(f)):
        if kernel in r['Kernel_Name'] and r['Counter_Name'] in names:
            k = int(r['Dispatch_Id'])
            d[k][r['Counter_Name']] += float(r['Counter_Value'])
            order[k] = r['Kernel_Name']
    return [(order[k], d[k]) for k in sorted(d)]


def read_bytes(c):        # size-split fabric read requests -> bytes
    n32, n64, n128, n = c['TCC_EA0_RDREQ_32B_sum'], c['TCC_EA0_RDREQ_64B_sum'], c['TCC_EA0_RDREQ_128B_sum'], c['TCC_EA0_RDREQ_sum']
    return 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(0.0, n - n32 - n64 - n128)


RD = ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_EA0_RDREQ_64B_sum', 'TCC_EA0_RDREQ_128B_sum')
# ---- calibration: known bytes vs counters
GiB = float(1 << 30)
known = {'stream_read16': GiB, 'gather8<37>': 4 * 2 ** 20 * 128.0, 'gather8<60>': 4 * 2 ** 20 * 128.0, 'rows8': None, 'stream_write16': GiB}
cal = {}
for name, c in per_dispatch('cal_p3', '', ('FETCH_SIZE',)):
    for k in known:
        if k in name: cal.setdefault(k, {})['FETCH_SIZE_bytes'] = c['FETCH_SIZE'] * 1024
for name, c in per_dispatch('cal_p3r', '', RD):
    for k in known:
        if k in name:
            cal.setdefault(k, {})['rdreq_bytes'] = read_bytes(c)
            cal[k]['rdreq'] = {n: c[n] for n in RD}
for name, c in per_dispatch('cal_p4', '', ('WRITE_SIZE',)):
    for k in known:
        if k in name: cal.setdefault(k, {})['WRITE_SIZE_bytes'] = c['WRITE_SIZE'] * 1024
for name, c in per_dispatch('cal_p4w', '', ('TCC_EA0_WRREQ_sum', 'TCC_EA0_WRREQ_64B_sum')):
    for k in known:
        if k in name: cal.setdefault(k, {})['wrreq_bytes'] = 64 * c['TCC_EA0_WRREQ_64B_sum'] + 32 * (c['TCC_EA0_WRREQ_sum'] - c['TCC_EA0_WRREQ_64B_sum'])
for k, v in cal.items():
    v['known_bytes'] = known[k]
# ---- the reconstruction kernel: last full step
fe = [c['FETCH_SIZE'] * 1024 for _n, c in per_dispatch('p3', 'hvq_recon', ('FETCH_SIZE',))][-nlev:]
rd = [read_bytes(c) for _n, c in per_dispatch('p3r', 'hvq_recon', RD)][-nlev:]
wr = [c['WRITE_SIZE'] * 1024 for _n, c in per_dispatch('p4', 'hvq_recon', ('WRITE_SIZE',))][-nlev:]
res = {"source": root, "launches": nlev,
       "fetch_size_bytes_per_launch_raw": sum(fe) / nlev, "read_bytes_per_launch_calibrated": sum(rd) / nlev,
       "write_bytes_per_launch": sum(wr) / nlev,
       "read_bytes_by_level_calibrated": rd, "fetch_size_by_level_raw": fe, "write_bytes_by_level": wr,
       "calibration": cal}
res["hbm_bytes_per_launch"] = res["fetch_size_bytes_per_launch_raw"] + res["write_bytes_per_launch"]
res["hbm_bytes_per_launch_calibrated"] = res["read_bytes_per_launch_calibrated"] + res["write_bytes_per_launch"]
if algo:
    res["algorithmic_bytes_per_launch"] = algo
    res["over_algorithmic"] = round(res["hbm_bytes_per_launch_calibrated"] / algo, 4)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.endswith('by_level') and k != 'calibration'}))
for k, v in cal.items():
    print(k, {a: (round(b / 2 ** 20, 1) if isinstance(b, float) else b) for a, b in v.items() if a != 'rdreq'}, 'MiB')
