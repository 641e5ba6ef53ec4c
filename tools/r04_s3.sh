#!/bin/bash
# GPU box: full default bench line with the inline-queue kernel, per-level kernel trace
T=${1:-r04d}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 400 python bench.py --clip-cache /tmp/hvq_clip_cache > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - $O/bench.json <<'PY' | tee $O/summary.txt
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("value %.0f stage %.4f (%.1f us) recon-only %.4f" % (d["value"], r["frac"], r["stage_us_per_step"], r["recon_only"]["frac"]))
for k in ("two_queues", "c5_staggered", "sdk_path", "rgb_epilogue"):
    print(k, json.dumps(d.get(k))[:400])
e = d["end_to_end_gpu_parse"]
print("streaming %.0f Mpx/s %.2f ms/batch parse %.3f ms calls %s; one batch %.0f; readback %s" % (e["streaming_value"], e["streaming_ms_per_batch"], e["streaming_parse_kernel_ms"], e["streaming_submit_end_begin_ms"], e["value"], json.dumps(e["streaming_with_readback"])[:200]))
print("end_to_end host", json.dumps(d["end_to_end"])[:300])
PY
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-verify --no-gpu-parse --no-sdk --gen-workers 1 --clip-cache /tmp/hvq_clip_cache"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- $B > $GRAFT_REPO_ROOT/$O/trace.json 2> $GRAFT_REPO_ROOT/$O/trace.err
cd $GRAFT_REPO_ROOT && python3 - $O <<'PY' | tee -a $O/summary.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "hvq_recon" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows]
print("launches", len(d), "last step levels (us):", ["%.1f" % x for x in d[-7:]], "sum %.1f" % sum(d[-7:]))
PY
