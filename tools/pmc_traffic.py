"""HBM traffic per launch of hvq_recon_kernel from tools/pmc_passes.sh output (passes p3 = FETCH_SIZE, p4 = WRITE_SIZE).
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide
coalesced streaming reads by 2x; this kernel's reads are narrow scattered 8-byte accesses, for which the counter is
uncalibrated -- the raw value is reported and the ratio to the algorithmic bytes is what is judged (waste = re-reads)."""
import csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
nlev = int(sys.argv[3]) if len(sys.argv) > 3 else 7
def per_dispatch(pdir, name):
    f = glob.glob(f'{root}/{pdir}/**/*counter_collection.csv', recursive=True)[0]
    d = {}
    for r in csv.DictReader(open(f)):
        if 'hvq_recon' in r['Kernel_Name'] and r['Counter_Name'] == name:
            d[int(r['Dispatch_Id'])] = d.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
    return [d[k] for k in sorted(d)]
fe, wr = per_dispatch('p3', 'FETCH_SIZE'), per_dispatch('p4', 'WRITE_SIZE')
# skip the first pass (flush) and warm-up: take the last full step
fe, wr = fe[-nlev:], wr[-nlev:]
res = {"source": root, "launches": nlev,
       "fetch_bytes_per_launch": sum(fe) * 1024 / nlev, "write_bytes_per_launch": sum(wr) * 1024 / nlev,
       "fetch_bytes_by_level": [x * 1024 for x in fe], "write_bytes_by_level": [x * 1024 for x in wr]}
res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch"] + res["write_bytes_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
