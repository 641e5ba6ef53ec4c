"""Diagnostic (GPU box): decode a clip on the GPU, compare with the oracle block by block and
print which block kinds mismatch.  Test infrastructure; uses oracle/."""
import ctypes as C
import struct
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hvqm4_amd import batch
from hvqm4_amd._lib import lib
from hvqm4_amd.synth import SynthConfig, make_clip
from hvqm4_amd.container import parse_header, video_pictures
from oracle import bridge


def blob_maps(blob):
    (magic, total, w, h, kind, unk, dcs, ws, hs) = struct.unpack_from("<IIHHBBBBB", blob, 0)
    flags, = struct.unpack_from("<I", blob, 20)
    hb = struct.unpack_from("<3H", blob, 24); vb = struct.unpack_from("<3H", blob, 30)
    plane_off = struct.unpack_from("<3I", blob, 36)
    map_off = struct.unpack_from("<3I", blob, 52)
    maps = []
    for p in range(3):
        n = (hb[p] + 2) * (vb[p] + 2)
        m = np.frombuffer(blob, dtype=np.uint8, count=2 * n, offset=map_off[p]).reshape(vb[p] + 2, hb[p] + 2, 2)
        maps.append(m[1:-1, 1:-1])
    return dict(w=w, h=h, kind=kind, flags=flags, hb=hb, vb=vb, plane_off=plane_off, maps=maps, ws=ws, hs=hs)


def main():
    w, h = int(sys.argv[1]), int(sys.argv[2])
    gop = sys.argv[3] if len(sys.argv) > 3 else "IPBBP"
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1234
    ver = sys.argv[5] if len(sys.argv) > 5 else "1.5"
    clip = make_clip(SynthConfig(width=w, height=h, gop=gop, seed=seed, version=ver))
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    ctx = batch.Context(0)
    got = batch.decode_clip(ctx, clip.data)
    l = lib()
    prs = l.hvq_parser_create(w, h, 2, 2, 1 if ver == "1.5" else 0)
    bound = l.hvq_parser_blob_bound(prs)
    buf = np.zeros(bound, dtype=np.uint8)
    for i, (ft, _d, pic) in enumerate(video_pictures(clip.data)):
        n = C.c_size_t(0)
        rc = l.hvq_parse_picture(prs, ft, pic + b"\0" * 8, len(pic), buf.ctypes.data, bound, C.byref(n))
        assert rc == 0
        info = blob_maps(buf[:n.value].tobytes())
        bad_total = int((got[i] != want[i]).sum())
        print(f"picture {i} type {ft:#x} kind {info['kind']} flags {info['flags']:#x}: {bad_total} bytes differ")
        if not bad_total:
            continue
        for p in range(3):
            pw = w >> (info['ws'] if p else 0); ph = h >> (info['hs'] if p else 0)
            a = got[i][info['plane_off'][p]:info['plane_off'][p] + pw * ph].reshape(ph, pw)
            b = want[i][info['plane_off'][p]:info['plane_off'][p] + pw * ph].reshape(ph, pw)
            blk_bad = (a != b).reshape(ph // 4, 4, pw // 4, 4).any(axis=(1, 3))
            types = info['maps'][p][:, :, 1]
            hist = {}
            for t in np.unique(types):
                tot = int((types == t).sum()); bad = int(((types == t) & blk_bad).sum())
                if bad:
                    hist[f"{t:#04x}"] = f"{bad}/{tot}"
            print(f"   plane {p}: bad blocks {int(blk_bad.sum())}/{blk_bad.size}  by type: {hist}")
            if blk_bad.any():
                by, bx = np.argwhere(blk_bad)[0]
                print(f"   first bad block ({by},{bx}) type {types[by, bx]:#04x} dc {info['maps'][p][by, bx, 0]}")
                print("   got :", a[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4].tolist())
                print("   want:", b[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4].tolist())
    ctx.close()


if __name__ == "__main__":
    main()
