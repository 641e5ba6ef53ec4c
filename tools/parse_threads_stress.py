"""Host (run it where the cores are real: the GPU box): the task-structured host parser with 1 thread against 2 / 4 / 8 threads sharing a
picture's sections, on randomized clips (the parity sweep's generator) -- every blob byte for byte, every return code and flag word.
No GPU involved.  usage: python tools/parse_threads_stress.py [clips] [seed]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from hvqm4_amd._lib import lib  # noqa: E402
from hvqm4_amd.container import video_pictures  # noqa: E402
from hvqm4_amd.synth import make_clip  # noqa: E402
from tools.parity_sweep import draw  # noqa: E402


def blobs(l, clip, pics, threads):
    prs = l.hvq_parser_create(clip.width, clip.height, clip.samp_h, clip.samp_v, 1 if clip.version == "1.5" else 0)
    l.hvq_parser_set_threads(prs, threads)
    bound = l.hvq_parser_blob_bound(prs)
    blob = np.zeros(bound + 64, np.uint8)
    out = []
    n = C.c_size_t(0)
    for ft, p in pics:
        rc = l.hvq_parse_picture(prs, ft, p + b"\0" * 8, len(p), blob.ctypes.data, bound, C.byref(n))
        out.append((rc, l.hvq_parser_last_flags(prs), blob[:n.value].tobytes() if rc == 0 else b""))
    l.hvq_parser_destroy(prs)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 606
    l = lib()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    npic = bad = 0
    for i in range(n):
        clip = make_clip(draw(rng))
        pics = [(ft, bytes(p)) for ft, _d, p in video_pictures(clip.data)]
        one = blobs(l, clip, pics, 1)
        for threads in (2, 4, 8):
            if blobs(l, clip, pics, threads) != one:
                bad += 1
                print(f"MISMATCH clip {i} ({clip.width}x{clip.height}) with {threads} threads", flush=True)
        npic += len(pics)
        if (i + 1) % 200 == 0:
            print(f"{i + 1} clips, {npic} pictures, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"stress done: {n} clips, {npic} pictures x 3 thread counts, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
