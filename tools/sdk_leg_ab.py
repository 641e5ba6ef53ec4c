"""GPU box: the SDK boundary (one synchronous HVQM4Decode*pic call per picture) with 1 / 2 / 3 / 4 / 6 threads parsing a picture's
sections (HVQM4_AMD_SDK_PARSE_THREADS, hvq_parser_set_threads); each setting in a child process (the variable is read once)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, os.environ["HVQ_ROOT"])
import bench
from hvqm4_amd.synth import SynthConfig, make_clip
from hvqm4_amd.container import video_pictures
clip = make_clip(SynthConfig(width=640, height=480, version="1.5", gop=bench.GOP16, seed=1000, preset=os.environ.get("PRESET", "dense")))
print(json.dumps(bench.sdk_leg(clip, list(video_pictures(clip.data)))))
'''
for preset in ("dense", "natural"):
    for t in (1, 2, 3, 4, 6):
        env = dict(os.environ, HVQ_ROOT=ROOT, HVQM4_AMD_SDK_PARSE_THREADS=str(t), PRESET=preset)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{preset:8s} parse threads {t}: {j['value']:8.1f} Mpixel/s  {j['ms_per_picture']:.3f} ms per picture", flush=True)
        except Exception:
            print(preset, t, "failed:", (r.stdout + r.stderr)[-600:], flush=True)
