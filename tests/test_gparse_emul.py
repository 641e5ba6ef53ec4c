"""CPU: the GPU parse core (hvqm4_amd/csrc/hvq_gparse_core.h), run phase by phase on the CPU by
tests/native/gparse_emul.c, must produce the host parser's descriptor blobs byte for byte: same maps, motion
vectors, run bases, pool, header; the nest it writes aside must be the one the host parser embeds."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import clips

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "gparse_emul.c")
OUT = os.path.join(HERE, "native", "_build", "libgparse_emul.so")
NESTP = (70 * 38 // 2 + 14 + 15) & ~15

HDR = struct.Struct("<II HH BBBBB3x I 3H3H 3I I 3I I I I I I 4I II H2x I 3I")
# magic total | w h | kind unk dc ws hs | flags | hb vb | plane_off | pic_bytes | map_off | mv | wave | pool | pool_dw | nest | tile_first | mcb | max_items | max_pairs


class Result(C.Structure):
    _fields_ = [("status", C.c_uint32), ("flags", C.c_uint32), ("max_items", C.c_uint32), ("max_pairs", C.c_uint32),
                ("pool_dwords", C.c_uint32), ("total_bytes", C.c_uint32), ("pad", C.c_uint32 * 2)]


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [SRC] + [os.path.join(HERE, "..", "hvqm4_amd", "csrc", f) for f in ("hvq_gparse_core.h", "hvq_gparse_flat.h", "hvq_desc.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.run(["gcc", "-O2", "-Wall", "-shared", "-fPIC", SRC, "-o", OUT], check=True)
    lib = C.CDLL(OUT)
    lib.gparse_emul2.restype = C.c_int
    lib.gparse_emul2.argtypes = [C.c_char_p, C.c_uint32] + [C.c_int] * 6 + [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(Result), C.c_int]
    return lib


def header(blob):
    f = HDR.unpack_from(blob, 0)
    names = ["magic", "total", "w", "h", "kind", "unk", "dc", "ws", "hs", "flags"]
    d = dict(zip(names, f[:10]))
    d["hb"], d["vb"], d["plane_off"] = f[10:13], f[13:16], f[16:19]
    d["pic_bytes"] = f[19]
    d["map_off"] = f[20:23]
    d["mv_off"], d["wave_off"], d["pool_off"], d["pool_dwords"], d["nest_off"] = f[23:28]
    d["tile_first"] = f[28:32]
    d["mcb_w"], d["mcb_h"], d["max_items"], d["max_pairs"] = f[32:36]
    return d


CHAINS, FLAT, FLAT_ONLY = 0, 1, 2      # gparse_emul2 modes: round 1's chains | flat path with fallback | flat path or fail


def compare_clip(emul, clip, mode=FLAT_ONLY):
    from hvqm4_amd._lib import lib
    l = lib()
    is15 = 1 if clip.version == "1.5" else 0
    prs = l.hvq_parser_create(clip.width, clip.height, clip.samp_h, clip.samp_v, is15)
    assert prs
    bound = l.hvq_parser_blob_bound(prs)
    a = np.zeros(bound, dtype=np.uint8)
    b = np.zeros(bound, dtype=np.uint8)
    nest = np.zeros(NESTP, dtype=np.uint8)
    last_nest = np.zeros(NESTP, dtype=np.uint8)
    for idx, (ft, pic) in enumerate(zip(clip.kinds, clip.pictures)):
        n = C.c_size_t(0)
        assert l.hvq_parse_picture(prs, ft, pic + b"\0" * 8, len(pic), a.ctypes.data, bound, C.byref(n)) == 0
        b[:] = 0xEE
        res = Result()
        assert emul.gparse_emul2(pic, len(pic), ft, clip.width, clip.height, clip.samp_h, clip.samp_v, is15,
                                 b.ctypes.data, bound, nest.ctypes.data, C.byref(res), mode) == 0, (idx, "flat path gave up")
        assert res.status == 0, (idx, res.status)
        ha, hb = header(a.tobytes()), header(b.tobytes())
        where = f"picture {idx} type {ft:#x}"
        for k in ha:
            if k in ("total", "nest_off"):
                continue
            assert ha[k] == hb[k], (where, k, ha[k], hb[k])
        assert res.flags == ha["flags"] and res.max_items == ha["max_items"] and res.max_pairs == ha["max_pairs"], where
        assert res.pool_dwords == ha["pool_dwords"], where
        for i in range(3):
            nmap = 2 * (ha["hb"][i] + 2) * (ha["vb"][i] + 2)
            o = ha["map_off"][i]
            assert np.array_equal(a[o:o + nmap], b[o:o + nmap]), (where, "map", i)
        if ft != 0x10:
            o, nmv = ha["mv_off"], 4 * ha["mcb_w"] * ha["mcb_h"]
            assert np.array_equal(a[o:o + nmv], b[o:o + nmv]), (where, "mv")
        o, nw = ha["wave_off"], 4 * ha["tile_first"][3] * 4
        assert np.array_equal(a[o:o + nw], b[o:o + nw]), (where, "wave_base")
        o, npool = ha["pool_off"], 4 * ha["pool_dwords"]
        assert np.array_equal(a[o:o + npool], b[o:o + npool]), (where, "pool")
        if ft == 0x10:
            last_nest = nest.copy()
        if ha["nest_off"]:
            o = ha["nest_off"]
            assert np.array_equal(a[o:o + NESTP], last_nest), (where, "nest")
    l.hvq_parser_destroy(prs)


@pytest.mark.parametrize("mode", [CHAINS, FLAT_ONLY], ids=["chains", "flat"])
@pytest.mark.parametrize("case", clips.SMALL + clips.MEDIUM + clips.REGRESSION, ids=lambda c: c[0])
def test_gpu_parse_core_matches_host_parser(emul, case, mode):
    if case[0].startswith(("longescape", "bigscalars")) and mode == FLAT_ONLY:
        mode = FLAT        # overflow runs of hundreds of symbols are what the flat path hands to the chains by design
    compare_clip(emul, clips.get(case), mode)


def test_gpu_parse_core_random_geometries(emul):
    from hvqm4_amd.synth import SynthConfig, make_clip
    rng = np.random.default_rng(99)
    for _ in range(40):
        cfg = SynthConfig(width=int(rng.integers(1, 30)) * 8, height=int(rng.integers(1, 24)) * 8,
                          version=str(rng.choice(["1.3", "1.5"])), gop=str(rng.choice(["IPB", "IPBBPBB", "IPPP"])),
                          seed=int(rng.integers(0, 1 << 30)), preset=str(rng.choice(["dense", "realistic", "flat", "natural"])),
                          sampling=str(rng.choice(["420", "444", "422"])), weird_kinds=bool(rng.random() < 0.3),
                          runoff_prob=float(rng.choice([0.0, 0.3])))
        compare_clip(emul, make_clip(cfg))


@pytest.mark.parametrize("w,h,samp", [(1920, 1088, "420"), (2048, 8, "420"), (8, 2048, "420"), (1024, 16, "444"), (16, 1024, "422")])
def test_gpu_parse_core_large_and_extreme_geometries(emul, w, h, samp):
    from hvqm4_amd.synth import SynthConfig, make_clip
    compare_clip(emul, make_clip(SynthConfig(width=w, height=h, gop="IPB", seed=w + h, sampling=samp, runoff_prob=0.2)))


def test_flat_path_hands_unusual_pictures_to_the_chains(emul):
    """Pictures the flat path cannot serve -- a one-leaf DC tree whose only value lies outside the overflow window (every
    value runs to the chains' cap of 256 symbols), and sections truncated so that lanes run dry -- must come out of the
    fallback exactly as the host parser decodes them, and the result must say that the chains did it."""
    import copy
    from hvqm4_amd.container import video_pictures
    from hvqm4_amd.synth import SynthConfig, make_clip
    clip = make_clip(SynthConfig(width=96, height=64, gop="IPB", seed=5))
    pics = []
    for ft, _d, pic in video_pictures(clip.data):
        p = bytearray(pic)
        data = 8 + (0x40 if ft == 0x10 else 0x44)
        off = data + struct.unpack_from(">I", p, 8 + 4 * 4)[0] + 4          # section 4 = DC buffer of the luma plane
        p[0] = 0
        p[off:off + 2] = b"\x3f\x80"                                         # tree = single leaf 0x7F
        pics.append((ft, bytes(p)))
    from hvqm4_amd._lib import lib
    l = lib()
    prs = l.hvq_parser_create(clip.width, clip.height, clip.samp_h, clip.samp_v, 1 if clip.version == "1.5" else 0)
    bound = l.hvq_parser_blob_bound(prs)
    a = np.zeros(bound, dtype=np.uint8)
    b = np.zeros(bound, dtype=np.uint8)
    nest = np.zeros(NESTP, dtype=np.uint8)
    retried = capped = 0
    for ft, pic in pics:
        n = C.c_size_t(0)
        rc = l.hvq_parse_picture(prs, ft, pic + b"\0" * 8, len(pic), a.ctypes.data, bound, C.byref(n))
        res = Result()
        assert emul.gparse_emul2(pic, len(pic), ft, clip.width, clip.height, clip.samp_h, clip.samp_v, 1 if clip.version == "1.5" else 0,
                                 b.ctypes.data, bound, nest.ctypes.data, C.byref(res), FLAT) == 0
        assert (rc == 0) == (res.status == 0)
        if rc == 0:
            ha = header(a.tobytes())
            # a run that never ends: both parsers flag the picture (HVQ_F_CAPPED, 0x40) and it is refused -- since round 5 the host
            # parser follows a run beyond the fast cap to the end of the picture first and stops decoding values once the flag is
            # up, so the blobs of a REFUSED picture are no longer comparable (nothing ever reads them); an unflagged one must match
            assert bool(ha["flags"] & 0x40) == bool(res.flags & 0x40)
            capped += bool(ha["flags"] & 0x40)
            if not ha["flags"] & 0x40:
                o, npool = ha["pool_off"], 4 * ha["pool_dwords"]
                assert np.array_equal(a[o:o + npool], b[o:o + npool])
                for i in range(3):
                    nmap = 2 * (ha["hb"][i] + 2) * (ha["vb"][i] + 2)
                    o = ha["map_off"][i]
                    assert np.array_equal(a[o:o + nmap], b[o:o + nmap])
        retried += res.pad[0]
    assert retried >= 1 and capped >= 1
    l.hvq_parser_destroy(prs)


def test_flat_path_equals_the_chains_on_corrupted_pictures(emul):
    """"Same blob either way" must also hold where no golden exists: bit flips, truncations and garbage spliced into
    pictures of four geometries and presets -- the flat path (with its hand-over to the chains) and the chains alone must
    agree on status, result record, the whole blob and the nest, byte for byte."""
    from hvqm4_amd.container import video_pictures
    from hvqm4_amd.synth import SynthConfig, make_clip
    rng = np.random.default_rng(5)
    cap = 4 << 20
    a = np.zeros(cap, dtype=np.uint8); b = np.zeros(cap, dtype=np.uint8)
    na = np.zeros(2048, dtype=np.uint8); nb = np.zeros(2048, dtype=np.uint8)
    total = fell_back = 0
    for seed, (w, h) in enumerate([(96, 64), (64, 96), (160, 128), (48, 48)]):
        clip = make_clip(SynthConfig(width=w, height=h, gop="IPBB", seed=seed + 1, preset=["dense", "natural", "realistic", "flat"][seed % 4]))
        is15 = 1 if clip.version == "1.5" else 0
        for ft, _d, pic in video_pictures(clip.data):
            p = bytes(pic)
            for v in range(24):
                q = bytearray(p)
                if v % 3 == 0:
                    for _ in range(int(rng.integers(1, 12))):
                        q[int(rng.integers(8, len(q)))] ^= 1 << int(rng.integers(0, 8))
                elif v % 3 == 1:
                    q = q[:int(rng.integers(0x60, len(q)))]
                else:
                    o = int(rng.integers(0x50, len(q) - 8))
                    q[o:o + 8] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))
                q = bytes(q)
                a[:] = 0; b[:] = 0; na[:] = 0; nb[:] = 0
                ra, rb = Result(), Result()
                assert emul.gparse_emul2(q, len(q), ft, w, h, clip.samp_h, clip.samp_v, is15, a.ctypes.data, cap, na.ctypes.data, C.byref(ra), CHAINS) == 0
                assert emul.gparse_emul2(q, len(q), ft, w, h, clip.samp_h, clip.samp_v, is15, b.ctypes.data, cap, nb.ctypes.data, C.byref(rb), FLAT) == 0
                total += 1; fell_back += rb.pad[0]
                assert ra.status == rb.status, (seed, hex(ft), v)
                if ra.status == 0:
                    assert (ra.flags, ra.total_bytes, ra.max_items, ra.max_pairs, ra.pool_dwords) == \
                           (rb.flags, rb.total_bytes, rb.max_items, rb.max_pairs, rb.pool_dwords), (seed, hex(ft), v)
                    assert np.array_equal(a[:ra.total_bytes], b[:rb.total_bytes]), (seed, hex(ft), v, "blob")
                    assert np.array_equal(na, nb), (seed, hex(ft), v, "nest")
    assert total == 4 * 4 * 24 and fell_back > 0
