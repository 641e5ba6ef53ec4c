"""Host: stage times of the host entropy parse of one dense 640x480 GOP with 1..6 threads sharing a picture's sections
(HVQM4_AMD_PARSE_TIMING=1 makes hvq_parser_destroy print them)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HVQM4_AMD_PARSE_TIMING"] = "1"
import numpy as np
from hvqm4_amd._lib import lib
from hvqm4_amd.container import video_pictures
from hvqm4_amd.synth import SynthConfig, make_clip

l = lib()
preset = sys.argv[1] if len(sys.argv) > 1 else "dense"
clip = make_clip(SynthConfig(width=640, height=480, version="1.5", gop="IPBBPBBPBBPBBPBB", seed=1000, preset=preset))
pics = [(ft, bytes(p) + b"\0" * 8, len(p)) for ft, _d, p in video_pictures(clip.data)]
for threads in (1, 2, 3, 4, 6):
    prs = l.hvq_parser_create(640, 480, 2, 2, 1)
    l.hvq_parser_set_threads(prs, threads)
    bound = l.hvq_parser_blob_bound(prs)
    blob = np.zeros(bound + 64, np.uint8)
    n = C.c_size_t(0)
    best = 1e9
    for rep in range(8):
        t0 = time.perf_counter()
        for ft, p, ln in pics:
            l.hvq_parse_picture(prs, ft, p, ln, blob.ctypes.data, bound, C.byref(n))
        best = min(best, time.perf_counter() - t0)
    print(f"{preset} threads {threads}: {best / len(pics) * 1e3:.3f} ms per picture", flush=True)
    sys.stdout.flush()
    l.hvq_parser_destroy(prs)
