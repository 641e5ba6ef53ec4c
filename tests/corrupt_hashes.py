"""Helper of tests/test_gpu_modes.py: decode corrupted pictures through the device entropy parse and print one line per
variant (error code or hash of the decoded picture).  Run in a child process per parse mode (HVQM4_AMD_PARSE_FLAT)."""
import hashlib
import sys

import numpy as np


def main():
    from hvqm4_amd import batch
    from hvqm4_amd._lib import HvqError
    from hvqm4_amd.container import parse_header, video_pictures
    from hvqm4_amd.synth import SynthConfig, make_clip
    ctx = batch.Context(0)
    rng = np.random.default_rng(17)
    for seed, (w, h) in enumerate([(96, 64), (160, 128), (64, 96)]):
        clip = make_clip(SynthConfig(width=w, height=h, gop="IPB", seed=seed + 3, preset=["dense", "natural", "realistic"][seed]))
        hdr = parse_header(clip.data)
        pics = [(ft, bytes(p)) for ft, _d, p in video_pictures(clip.data)]
        for k, (ft, p) in enumerate(pics):
            for v in range(12):
                q = bytearray(p)
                if v % 3 == 0:
                    for _ in range(int(rng.integers(1, 10))):
                        q[int(rng.integers(8, len(q)))] ^= 1 << int(rng.integers(0, 8))
                elif v % 3 == 1:
                    q = q[:int(rng.integers(0x60, len(q)))]
                else:
                    o = int(rng.integers(0x50, len(q) - 8))
                    q[o:o + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
                sid = ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, 4)
                line = f"{seed} {k} {v} "
                try:
                    # the pictures before k intact, then the corrupted one
                    ctx.submit_many_device([sid] * (k + 1), [pics[i][0] for i in range(k)] + [ft], [pics[i][1] for i in range(k)] + [bytes(q)])
                    ctx.flush()
                    line += hashlib.sha1(ctx.read_picture(sid, k).tobytes()).hexdigest()
                except HvqError as e:
                    line += f"error {e.code}"
                print(line, flush=True)
                ctx.close_stream(sid)
    ctx.close()


if __name__ == "__main__":
    sys.exit(main())
