"""CPU: the product's host entropy parse (hvq_parse.c, through the C ABI) + the descriptor-blob
specification (oracle/hvq_desc_recon.c interprets blobs on the CPU) against the oracle.  This checks
everything the GPU kernels are fed with, without a GPU."""
import ctypes as C
import struct

import numpy as np
import pytest

from tests import clips

I_FRAME, P_FRAME, B_FRAME = 0x10, 0x20, 0x30


def decode_via_descriptors(clip, truncate=None):
    from hvqm4_amd._lib import lib
    from oracle import bridge
    l = lib()
    o = bridge.oracle()
    o.hvqd_recon.restype = C.c_int
    o.hvqd_recon.argtypes = [C.c_void_p] * 4 + [C.c_uint32]
    ps = clip.picsize
    slot = ps + 64
    prs = l.hvq_parser_create(clip.width, clip.height, clip.samp_h, clip.samp_v, 1 if clip.version == "1.5" else 0)
    assert prs
    bound = l.hvq_parser_blob_bound(prs)
    blob = np.zeros(bound, dtype=np.uint8)
    bufs = [np.zeros(slot, dtype=np.uint8) for _ in range(3)]        # past, present, future
    out, flags = [], 0
    for ft, pic in zip(clip.kinds, clip.pictures):
        if ft != B_FRAME:
            bufs[0], bufs[2] = bufs[2], bufs[0]
        n = C.c_size_t(0)
        data = pic if truncate is None else pic[:truncate]
        rc = l.hvq_parse_picture(prs, ft, data + b"\0" * 8, len(data), blob.ctypes.data, bound, C.byref(n))
        assert rc == 0, rc
        hdr = blob[:128].tobytes()
        assert struct.unpack_from("<I", hdr, 0)[0] == 0x34515648
        assert struct.unpack_from("<I", hdr, 4)[0] == n.value
        flags |= struct.unpack_from("<I", hdr, 20)[0]
        ref1 = bufs[1] if ft == P_FRAME else bufs[2]
        assert o.hvqd_recon(blob.ctypes.data, bufs[1].ctypes.data, bufs[0].ctypes.data, ref1.ctypes.data, slot) == 0
        out.append(bufs[1][:ps].copy())
        if ft != B_FRAME:
            bufs[1], bufs[2] = bufs[2], bufs[1]
    l.hvq_parser_destroy(prs)
    return np.stack(out), flags


@pytest.mark.parametrize("case", clips.SMALL + clips.MEDIUM[:2] + clips.REGRESSION, ids=lambda c: c[0])
def test_parse_plus_descriptor_spec_matches_oracle(case):
    from oracle import bridge
    clip = clips.get(case)
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    got, flags = decode_via_descriptors(clip)
    assert np.array_equal(got, want)
    assert not flags & 0x60, "legal streams must not be flagged CLAMPED / CAPPED"
    assert bool(flags & 0x08) == case[0].startswith("pselfref"), "SELF_REF: exactly the P pictures with future-referencing macroblocks"


def test_big_aot_flag_only_for_weird_streams():
    _, f_plain = decode_via_descriptors(clips.get(clips.SMALL[3]))
    _, f_weird = decode_via_descriptors(clips.get(clips.SMALL[9]))
    assert not f_plain & 0x10
    assert f_weird & 0x10          # I-luma type bytes > 15 present


def test_truncated_pictures_do_not_crash_the_parser():
    """malformed input: the reference has no validation at all; ours must stay in bounds"""
    clip = clips.get(clips.SMALL[3])
    for cut in (0x60, 0x80, 200, 400):
        decode_via_descriptors(clip, truncate=cut)


def test_random_garbage_pictures_do_not_crash_the_parser():
    from hvqm4_amd._lib import lib
    l = lib()
    rng = np.random.default_rng(5)
    prs = l.hvq_parser_create(64, 48, 2, 2, 1)
    bound = l.hvq_parser_blob_bound(prs)
    blob = np.zeros(bound, dtype=np.uint8)
    for i in range(200):
        junk = rng.integers(0, 256, int(rng.integers(100, 3000)), dtype=np.uint8).tobytes()
        n = C.c_size_t(0)
        rc = l.hvq_parse_picture(prs, (0x10, 0x20, 0x30)[i % 3], junk + b"\0" * 8, len(junk), blob.ctypes.data, bound, C.byref(n))
        assert rc in (0, -2)
    l.hvq_parser_destroy(prs)


def test_unsupported_geometry_is_rejected():
    from hvqm4_amd._lib import lib
    l = lib()
    assert not l.hvq_parser_create(60, 48, 2, 2, 1)       # width not a multiple of 8 (h4m:1749-1752)
    assert not l.hvq_parser_create(64, 48, 1, 2, 1)       # the sampling the reference's own tables cannot decode (h4m:863-870)
    assert not l.hvq_parser_create(64, 48, 4, 2, 1)
    prs = l.hvq_parser_create(64, 48, 2, 1, 1)           # 4:2:2 as the reference spells it
    assert prs and l.hvq_parser_pic_bytes(prs) == 64 * 48 * 2
    l.hvq_parser_destroy(prs)
    assert not l.hvq_parser_create(0, 0, 2, 2, 1)


def test_sdk_host_side_stays_inside_exactly_sized_frames_under_asan(tmp_path):
    """The SDK signatures carry no frame length: it is derived from the picture's own section table
    (hvq_picture_length) and bounds the host parse.  Compiled with AddressSanitizer, fed legal, truncated and
    bit-flipped pictures in heap buffers without a spare byte (tests/native/sdk_bounds_asan.c)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "sdk_bounds_asan"
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=shift", "-fno-omit-frame-pointer",
                           os.path.join(root, "tests", "native", "sdk_bounds_asan.c"), os.path.join(root, "hvqm4_amd", "csrc", "hvq_parse.c"),
                           "-lpthread", "-o", str(exe)])
    for case in (clips.SMALL[3], clips.SMALL[4], clips.SMALL[9]):
        clip = clips.get(case)
        rec = tmp_path / "pics.bin"
        with open(rec, "wb") as f:
            for ft, pic in zip(clip.kinds, clip.pictures):
                f.write(struct.pack("<II", ft, len(pic)) + pic)
        r = subprocess.run([str(exe), str(clip.width), str(clip.height), "1" if clip.version == "1.5" else "0", str(rec)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert "legal lengths exact" in r.stdout


def test_overflow_runs_beyond_the_fast_cap_are_followed_to_their_end():
    """The reference sums overflow symbols for as long as the stream says (h4m:654-677).  One value of 5 000 symbols -- beyond the
    4096 the fast loops of both parsers stop at -- must come out of the host parser exactly (rounds 3-4 flagged the picture
    HVQ_F_CAPPED and refused it), in an I picture's DC deltas and in a P picture's scalars; against the oracle and, when it is
    built here, against the compiled reference."""
    from hvqm4_amd.synth import SynthConfig, make_clip
    from oracle import bridge
    for cfg in (SynthConfig(width=64, height=48, gop="IPBBPB", seed=5, long_escape_pb=5000),
                SynthConfig(width=64, height=48, gop="IPB", seed=35, long_escape=5000),
                SynthConfig(width=96, height=64, gop="IPB", seed=8, long_escape=9000, long_escape_pb=7000)):
        clip = make_clip(cfg)
        want = bridge.oracle_decode(clip.data, clip.n_pictures)
        if bridge.have_ref():
            assert np.array_equal(want, bridge.ref_decode(clip.data, clip.n_pictures)[0])
        got, flags = decode_via_descriptors(clip)
        assert not flags & 0x40, "a run that ends inside its picture must not be flagged CAPPED"
        assert np.array_equal(got, want)


def test_a_run_that_never_ends_is_flagged_and_costs_no_time():
    """a one-leaf DC tree whose value lies outside the overflow window: the reference would sum for ever.  The host parser gives up
    at the end of the picture (HVQ_F_CAPPED -> the back end refuses the picture) and does not spend the rest of a full-size picture
    spinning through the same cap"""
    import time
    from hvqm4_amd._lib import lib
    from hvqm4_amd.synth import SynthConfig, make_clip
    l = lib()
    clip = make_clip(SynthConfig(width=640, height=480, gop="IP", seed=77))
    prs = l.hvq_parser_create(clip.width, clip.height, clip.samp_h, clip.samp_v, 1)
    bound = l.hvq_parser_blob_bound(prs)
    blob = np.zeros(bound, dtype=np.uint8)
    t0 = time.time()
    for ft, pic in zip(clip.kinds, clip.pictures):
        p = bytearray(pic)
        data = 8 + (0x40 if ft == I_FRAME else 0x44)
        off = data + struct.unpack_from(">I", p, 8 + 4 * 4)[0] + 4          # section 4 = DC buffer of the luma plane
        p[0] = 0
        p[off:off + 2] = b"\x3f\x80"                                         # tree = single leaf 0x7F, window (-128, 127)
        n = C.c_size_t(0)
        rc = l.hvq_parse_picture(prs, ft, bytes(p) + b"\0" * 8, len(p), blob.ctypes.data, bound, C.byref(n))
        assert rc == 0
        assert struct.unpack_from("<I", blob[:128].tobytes(), 20)[0] & 0x40
    assert time.time() - t0 < 2.0
    l.hvq_parser_destroy(prs)


def _blobs(clip, threads, mangle=None):
    """every picture's blob (bytes), return code and late flags with `threads` threads parsing one picture"""
    from hvqm4_amd._lib import lib
    l = lib()
    prs = l.hvq_parser_create(clip.width, clip.height, clip.samp_h, clip.samp_v, 1 if clip.version == "1.5" else 0)
    assert prs
    assert l.hvq_parser_set_threads(prs, threads) == threads
    bound = l.hvq_parser_blob_bound(prs)
    blob = np.zeros(bound, dtype=np.uint8)
    out = []
    for k, (ft, pic) in enumerate(zip(clip.kinds, clip.pictures)):
        data = mangle(k, pic) if mangle else pic
        n = C.c_size_t(0)
        rc = l.hvq_parse_picture(prs, ft, data + b"\0" * 8, len(data), blob.ctypes.data, bound, C.byref(n))
        out.append((rc, blob[:n.value].tobytes() if rc == 0 else b"", l.hvq_parser_last_flags(prs)))
    l.hvq_parser_destroy(prs)
    return out


@pytest.mark.parametrize("case", clips.SMALL + clips.MEDIUM[:2] + clips.REGRESSION, ids=lambda c: c[0])
def test_sections_parsed_side_by_side_give_the_same_blob(case):
    """hvq_parser_set_threads: block kinds, DC values, vectors and the planes' payloads are separate bit buffers (h4m:1981-1993,
    2030-2044) decoded by a small pool -- the blob must not depend on how many threads shared the work (the SDK boundary runs 4)"""
    clip = clips.get(case)
    one = _blobs(clip, 1)
    for threads in (2, 4, 8):
        assert _blobs(clip, threads) == one, threads


def test_threaded_parse_of_corrupted_pictures_stays_in_bounds_and_agrees_on_acceptance():
    """bit flips behind the section table: every thread count must take or refuse the same pictures (a cap met in one section stops
    the others at different points, so only the verdict -- return code and refusal flags -- is compared), and none may crash"""
    clip = clips.get(clips.SMALL[3])
    rng = np.random.default_rng(11)
    flips = {k: rng.integers(0x60, max(0x61, len(p)), 24) for k, p in enumerate(clip.pictures)}

    def mangle(k, pic):
        b = bytearray(pic)
        for at in flips[k]:
            b[int(at) % len(b)] ^= 1 << (int(at) % 8)
        return bytes(b)
    verdicts = []
    for threads in (1, 4):
        verdicts.append([(rc, fl & 0x60) for rc, _b, fl in _blobs(clip, threads, mangle)])
    assert verdicts[0] == verdicts[1]


def test_parser_pool_is_created_and_torn_down_repeatedly():
    from hvqm4_amd._lib import lib
    l = lib()
    for _ in range(20):
        prs = l.hvq_parser_create(64, 48, 2, 2, 1)
        assert l.hvq_parser_set_threads(prs, 4) == 4
        assert l.hvq_parser_set_threads(prs, 1) == 1
        assert l.hvq_parser_set_threads(prs, 3) == 3
        l.hvq_parser_destroy(prs)


def test_4k_pictures_through_parse_and_descriptor_spec():
    """4096x2176: 1024 x 544 luma blocks -- the geometry fields, plane offsets and pool offsets far beyond the catalogue's sizes"""
    from hvqm4_amd.synth import SynthConfig, make_clip
    from oracle import bridge
    clip = make_clip(SynthConfig(width=4096, height=2176, gop="IP", seed=4096 + 2176, runoff_prob=0.2))
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    got, flags = decode_via_descriptors(clip)
    assert np.array_equal(got, want)
    assert not flags & 0x60
    assert _blobs(clip, 4) == _blobs(clip, 1)
