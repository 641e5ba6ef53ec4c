"""What the test clips exercise: a histogram over (picture kind x plane x block class x macroblock reference x proc) and
over the half-sample cases of motion compensation, read off the product's own descriptor blobs (host parser, CPU)."""
import ctypes as C
import struct
from collections import Counter

import numpy as np

I_FRAME, P_FRAME, B_FRAME = 0x10, 0x20, 0x30
PIC = {I_FRAME: "I", P_FRAME: "P", B_FRAME: "B"}


def _kind_class(kind: int) -> str:
    if kind == 0:
        return "k0"
    if kind == 6:
        return "k6"
    if kind == 8:
        return "k8"
    if 1 <= kind <= 5:
        return "k1-5"
    if kind == 7:
        return "k7"
    if 9 <= kind <= 15:
        return "k9-15"
    return "k>15"


def histogram(clip) -> Counter:
    """Counter over cells:
      ("blk", pic, plane, ctx, cls)   ctx = "intra" | "past" | "future" (+ "/proc" for proc = 1 macroblocks)
      ("mc", version, "luma"|"chroma", ref, hx, hy)   half-sample case of every motion-compensated block"""
    from hvqm4_amd._lib import lib
    l = lib()
    is15 = clip.version == "1.5"
    prs = l.hvq_parser_create(clip.width, clip.height, clip.samp_h, clip.samp_v, 1 if is15 else 0)
    bound = l.hvq_parser_blob_bound(prs)
    blob = np.zeros(bound, dtype=np.uint8)
    h = Counter()
    for ft, pic in zip(clip.kinds, clip.pictures):
        n = C.c_size_t(0)
        rc = l.hvq_parse_picture(prs, ft, pic + b"\0" * 8, len(pic), blob.ctypes.data, bound, C.byref(n))
        assert rc == 0, rc
        hdr = blob[:128].tobytes()
        (width, height, pic_kind, _unk, _dcs, wsh, hsh) = struct.unpack_from("<HHBBBBB", hdr, 8)
        hb = struct.unpack_from("<3H", hdr, 24); vb = struct.unpack_from("<3H", hdr, 30)
        map_off = struct.unpack_from("<3I", hdr, 52); mv_off = struct.unpack_from("<I", hdr, 64)[0]
        mcb_w = struct.unpack_from("<I", hdr, 100)[0]
        mv = None
        if ft != I_FRAME:
            nmb = mcb_w * struct.unpack_from("<I", hdr, 104)[0]
            mv = np.frombuffer(blob[mv_off:mv_off + 4 * nmb].tobytes(), dtype="<i2").reshape(-1, 2)
        for p in range(3):
            ms = hb[p] + 2
            m = blob[map_off[p]:map_off[p] + 2 * ms * (vb[p] + 2)].reshape(vb[p] + 2, ms, 2)[1:-1, 1:-1, 1].astype(int)
            ws, hs = (wsh, hsh) if p else (0, 0)
            for by in range(vb[p]):
                for bx in range(hb[p]):
                    t = int(m[by, bx])
                    if ft == I_FRAME:
                        kind = t if p == 0 else t & 15
                        h[("blk", "I", "YUV"[p], "intra", _kind_class(kind))] += 1
                        continue
                    kind, proc, ref = t & 15, (t >> 4) & 1, (t >> 5) & 3
                    ctx = "intra" if ref == 0 else ("past" if ref == 1 else "future")
                    h[("blk", PIC[ft], "YUV"[p], ctx + ("/proc" if ref and proc else ""), _kind_class(kind))] += 1
                    if ref and (proc or kind != 6):                      # motion compensated (plain or the MC part of a residual block)
                        rx, ry = mv[(by >> (1 - hs)) * mcb_w + (bx >> (1 - ws))]
                        pdx, pdy = int(rx) >> ws, int(ry) >> hs
                        hx, hy = ((pdx & 1), (pdy & 1)) if is15 else (int(rx) & 1, int(ry) & 1)
                        h[("mc", clip.version, "luma" if p == 0 else "chroma", ctx, hx, hy)] += 1
    l.hvq_parser_destroy(prs)
    return h


def required_cells():
    """every cell a legal stream can hit and the path treats differently"""
    cells = []
    for plane in "YUV":
        for cls in ("k0", "k8", "k6", "k1-5", "k7", "k9-15"):
            cells.append(("blk", "I", plane, "intra", cls))
    cells.append(("blk", "I", "Y", "intra", "k>15"))                     # I-luma type bytes above 15 (h4m:1093)
    for pic, refs in (("P", ("past",)), ("B", ("past", "future"))):
        for plane in "YUV":
            for cls in ("k0", "k8", "k6", "k1-5", "k7", "k9-15"):
                cells.append(("blk", pic, plane, "intra", cls))
            for ref in refs:
                for cls in ("k0", "k6", "k1-5", "k7", "k8", "k9-15"):      # proc = 0: kind selects none / literal / residual bases
                    cells.append(("blk", pic, plane, ref, cls))
                cells.append(("blk", pic, plane, ref + "/proc", "k0"))     # proc = 1: plain MC whatever the kind nibble says
    for version in ("1.3", "1.5"):
        for comp in ("luma", "chroma"):
            for ref in ("past", "future"):
                for hx in (0, 1):
                    for hy in (0, 1):
                        cells.append(("mc", version, comp, ref, hx, hy))
    return cells
