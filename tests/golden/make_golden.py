#!/usr/bin/env python3
"""Generate the committed golden vectors from the UNMODIFIED reference decoder.

Runs only in the build container (needs oracle/_ref/libh4mref.so, i.e. /root/reference).  For every
clip of tests/clips.py it stores the expected output of the reference (Tilka/hvqm4
h4m_audio_decode.c, compiled by oracle/Makefile):
  - small clips: the .h4m bytes (synthetic, produced by hvqm4_amd/synth.py) + SHA-256 per picture
  - medium clips: SHA-256 of the clip bytes (guards against generator drift) + SHA-256 per picture
The fixtures are data (inputs and expected outputs), not reference source.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import bridge  # noqa: E402
from tests import clips  # noqa: E402


def main():
    if not bridge.have_ref():
        raise SystemExit("oracle/_ref/libh4mref.so missing: run `make -C oracle ref` where /root/reference exists")
    manifest = {"generator": "tests/golden/make_golden.py", "reference": "Tilka/hvqm4 h4m_audio_decode.c (gcc -O2, -DNATIVE=1)",
                "clips": {}}
    for name, cfg in clips.SMALL + clips.MEDIUM + clips.C4_SHARE:
        clip = clips.get((name, cfg))
        pics, probe = bridge.ref_decode(clip.data, clip.n_pictures, probe=True)
        entry = {
            "width": clip.width, "height": clip.height, "version": clip.version,
            "frame_types": clip.kinds,
            "clip_sha256": hashlib.sha256(clip.data).hexdigest(),
            "picture_sha256": [hashlib.sha256(p.tobytes()).hexdigest() for p in pics],
        }
        if clip.width * clip.height <= 128 * 96 and clip.samp == 2 and clip.samp_v == 2:
            # display epilogue: RGB of every picture through the reference's own dumpRGB (h4m:901-926)
            entry["rgb_sha256"] = [hashlib.sha256(bridge.ref_rgb(p, clip.width, clip.height).tobytes()).hexdigest() for p in pics]
        if (name, cfg) in clips.SMALL:
            with open(os.path.join(HERE, name + ".h4m"), "wb") as f:
                f.write(clip.data)
            entry["file"] = name + ".h4m"
        manifest["clips"][name] = entry
        print(name, len(clip.data), "bytes", clip.n_pictures, "pictures")
    # known-answer vectors of the pixel primitives, from the reference's own functions
    import ctypes as C
    import numpy as np
    lib = bridge.ref()
    lib.ref_weight_block.argtypes = [C.c_void_p] + [C.c_uint8] * 5
    rng = np.random.default_rng(7)
    kats = []
    cases = [(0, 0, 255, 0, 255), (255, 0, 0, 0, 0), (0, 255, 255, 255, 255), (128, 127, 129, 126, 130)]
    cases += [tuple(int(x) for x in rng.integers(0, 256, 5)) for _ in range(60)]
    for v, t, b, l, r in cases:
        out = np.zeros(16, dtype=np.uint8)
        lib.ref_weight_block(out.ctypes.data, v, t, b, l, r)
        kats.append({"in": [v, t, b, l, r], "out": out.tolist()})
    manifest["weight_block_kat"] = kats
    lib.ref_motion_comp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    src = rng.integers(0, 256, 64, dtype=np.uint8)
    mck = []
    for hx in (0, 1):
        for hy in (0, 1):
            out = np.zeros(16, dtype=np.uint8)
            lib.ref_motion_comp(out.ctypes.data, src.ctypes.data, 8, hx, hy)
            mck.append({"hx": hx, "hy": hy, "out": out.tolist()})
    manifest["motion_comp_kat"] = {"src8x8": src.tolist(), "cases": mck}
    d16 = np.zeros(16, dtype=np.int32); m512 = np.zeros(512, dtype=np.int32)
    lib.ref_tables.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_tables(d16.ctypes.data, m512.ctypes.data)
    manifest["divTable"] = d16.tolist()
    manifest["mcdivTable_sha256"] = hashlib.sha256(m512.tobytes()).hexdigest()
    lay = np.zeros(5, dtype=np.uint32)
    lib.ref_layout.argtypes = [C.c_void_p]
    lib.ref_layout(lay.ctypes.data)
    manifest["layout_x86_64"] = {"sizeof_VideoState": int(lay[0]), "offsetof_padding": int(lay[1]),
                                 "sizeof_SeqObj": int(lay[2]), "sizeof_VideoInfo": int(lay[3])}
    manifest["buffsize"] = {"320x240": int(lib.ref_buffsize(320, 240, 2, 2)), "640x480": int(lib.ref_buffsize(640, 480, 2, 2)),
                            "64x48": int(lib.ref_buffsize(64, 48, 2, 2))}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote manifest.json")


if __name__ == "__main__":
    main()
