#!/usr/bin/env python3
"""One 640x480 HVQM4 1.5 clip through the seven SDK entry points, a few passes: for profiling the boundary itself
(HVQM4_AMD_SDK_TIMING=1 prints the per-call breakdown; under rocprofv3 --hip-trace the calls and copies)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hvqm4_amd import sdk
from hvqm4_amd.container import video_pictures
from hvqm4_amd.synth import SynthConfig, make_clip

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
clip = make_clip(SynthConfig(width=640, height=480, version="1.5", gop="IPBBPBBPBBPBBPBB", seed=1000, preset="dense"))
seq = list(video_pictures(clip.data))
pl = sdk.Player(clip.width, clip.height, 2, 2, True)
for ft, _d, pic in seq:
    pl.decode(ft, bytes(pic))
t0 = time.perf_counter()
for _ in range(passes):
    for ft, _d, pic in seq:
        pl.decode(ft, bytes(pic))
dt = (time.perf_counter() - t0) / (passes * len(seq))
pl.close()
print(f"sdk path: {dt * 1e3:.3f} ms per picture, {clip.width * clip.height / dt / 1e6:.1f} Mpixels/s")
