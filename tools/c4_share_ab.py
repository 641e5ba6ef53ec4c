"""GPU box: the small-batch leg of bench.py (one GPU's share of BASELINE config 4: 128 pictures in 7 launches) alone, e.g. with
HVQM4_AMD_GRAPH=0 / 1 in the environment.  usage: python tools/c4_share_ab.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    clips = bench.gen_clips(bench.c4_share_configs(), 1, "/tmp/hvq_clip_cache")      # in-process: no pool
    r = bench.c4_share_leg(0, steps, 20, 8, clips)
    print("c4 share HVQM4_AMD_GRAPH=%s: %s us per step, %s Mpixel/s %s" % (os.environ.get("HVQM4_AMD_GRAPH", "(default)"), r.get("us_per_step"), r.get("value"), r.get("error", "")))
