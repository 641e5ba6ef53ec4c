"""Host: how the host entropy parse scales over threads that each own a parser (the batched calls' arrangement): pictures per second with
1 .. 32 threads, each parsing the same dense 640x480 GOP into a blob buffer of its own.  No GPU involved."""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from hvqm4_amd._lib import lib  # noqa: E402
from hvqm4_amd.container import video_pictures  # noqa: E402
from hvqm4_amd.synth import SynthConfig, make_clip  # noqa: E402

l = lib()
clip = make_clip(SynthConfig(width=640, height=480, version="1.5", gop="IPBBPBBPBBPBBPBB", seed=1000, preset="dense"))
pics = [(ft, bytes(p) + b"\0" * 8, len(p)) for ft, _d, p in video_pictures(clip.data)]
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def work(out, k):
    prs = l.hvq_parser_create(640, 480, 2, 2, 1)
    bound = l.hvq_parser_blob_bound(prs)
    blob = np.zeros(bound + 64, np.uint8)
    n = C.c_size_t(0)
    for ft, p, ln in pics:                                   # warm
        l.hvq_parse_picture(prs, ft, p, ln, blob.ctypes.data, bound, C.byref(n))
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(REPS):
        for ft, p, ln in pics:
            l.hvq_parse_picture(prs, ft, p, ln, blob.ctypes.data, bound, C.byref(n))
    out[k] = time.perf_counter() - t0
    l.hvq_parser_destroy(prs)


for nt in (1, 2, 4, 8, 12, 16, 24, 32):
    barrier = threading.Barrier(nt)
    out = [0.0] * nt
    th = [threading.Thread(target=work, args=(out, k)) for k in range(nt)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    per = max(out) / (REPS * len(pics))
    print(f"{nt:2d} threads: {per * 1e3:.3f} ms per picture and thread, {nt * 640 * 480 / per / 1e6:8.0f} Mpixel/s in total", flush=True)
