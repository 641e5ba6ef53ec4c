"""GPU parity: HIP reconstruction path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from tests import clips

pytestmark = pytest.mark.gpu


def _first_diff(a, b):
    for i in range(a.shape[0]):
        d = np.nonzero(a[i] != b[i])[0]
        if len(d):
            return f"picture {i}: {len(d)} bytes differ, first at {d[:8].tolist()}"
    return "equal"


@pytest.mark.parametrize("case", clips.SMALL + clips.MEDIUM + clips.REGRESSION, ids=lambda c: c[0])
def test_batched_path_matches_oracle(case, gpu_ctx):
    from hvqm4_amd.batch import decode_clip
    from oracle import bridge
    clip = clips.get(case)
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    got = decode_clip(gpu_ctx, clip.data)
    assert got.shape == want.shape
    assert np.array_equal(got, want), _first_diff(got, want)      # integer pixel work: bit-exact
    # the chain is closed HERE, on the GPU box: when the compiled reference travelled with the snapshot (oracle/_ref), the same
    # pictures must equal what the unmodified reference decoder produces -- not only what this repo's restatement of it does
    if case in clips.SMALL and bridge.have_ref():
        ref = bridge.ref_decode(clip.data, clip.n_pictures)[0]
        assert np.array_equal(got, ref), "GPU vs compiled reference: " + _first_diff(got, ref)


@pytest.mark.parametrize("case", clips.SMALL[:8] + clips.SMALL[17:18] + clips.MEDIUM[1:2], ids=lambda c: c[0])
def test_sdk_entry_points_match_oracle(case):
    """The seven SDK symbols with the reference player's buffer rotation (h4m:2078-2138)."""
    from hvqm4_amd import sdk
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    clip = clips.get(case)
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    hdr = parse_header(clip.data)
    pl = sdk.Player(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15)
    got = np.stack([pl.decode(ft, pic) for ft, _d, pic in video_pictures(clip.data)])
    pl.close()
    assert np.array_equal(got, want), _first_diff(got, want)


def test_sdk_entry_points_on_four_threads_each_with_its_own_seqobj():
    """The SDK calls are reentrant per SeqObj: every SeqObj has its own device context and lock, so players on
    different threads decode concurrently (ctypes releases the GIL inside the calls).  Four clips of different geometry and
    version, three passes each, every picture against the oracle."""
    import threading
    from hvqm4_amd import sdk
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    cases = [clips.SMALL[3], clips.SMALL[4], clips.SMALL[5], clips.MEDIUM[1]]
    work = []
    for case in cases:
        clip = clips.get(case)
        work.append((clip, bridge.oracle_decode(clip.data, clip.n_pictures), parse_header(clip.data),
                     [(ft, bytes(pic)) for ft, _d, pic in video_pictures(clip.data)]))
    errors = []

    def run(k):
        try:
            clip, want, hdr, pics = work[k]
            pl = sdk.Player(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15)
            for _ in range(3):
                for i, (ft, pic) in enumerate(pics):
                    if not np.array_equal(pl.decode(ft, pic), want[i]):
                        errors.append((cases[k][0], i))
            pl.close()
        except Exception as e:                      # noqa: BLE001 -- reported below, on the main thread
            errors.append((cases[k][0], repr(e)))

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(work))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_ring_of_three_slots_matches_reference_rotation(gpu_ctx):
    """nslots=3 reproduces the reference's past/present/future rotation; last pictures stay readable."""
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    clip = clips.get(clips.SMALL[14])          # three GOPs
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    hdr = parse_header(clip.data)
    sid = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, nslots=3)
    for i, (ft, _d, pic) in enumerate(video_pictures(clip.data)):
        gpu_ctx.submit(sid, ft, pic)
        gpu_ctx.flush()
        got = gpu_ctx.read_picture(sid, i)
        assert np.array_equal(got, want[i]), f"picture {i}"
    gpu_ctx.close_stream(sid)


@pytest.mark.parametrize("case", [clips.SMALL[3], clips.SMALL[5], clips.SMALL[13], clips.MEDIUM[1]], ids=lambda c: c[0])
def test_rgb_epilogue_matches_oracle(case, gpu_ctx):
    """display epilogue (dumpRGB, h4m:897-926) on the GPU: float math with one rounding per operation -> bit-exact"""
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    clip = clips.get(case)
    hdr = parse_header(clip.data)
    pics = list(video_pictures(clip.data))
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    sid = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, len(pics) + 3)
    for ft, _d, pic in pics:
        gpu_ctx.submit(sid, ft, pic)
    gpu_ctx.flush()
    for i in range(len(pics)):
        got = gpu_ctx.read_picture_rgb(sid, i, hdr.width, hdr.height)
        exp = bridge.oracle_rgb(want[i], hdr.width, hdr.height).reshape(hdr.height, hdr.width, 3)
        assert np.array_equal(got, exp), i          # tolerance 0: IEEE single precision on both sides
    gpu_ctx.close_stream(sid)


@pytest.mark.parametrize("trust", [False, True], ids=["host_authoritative", "trusted_pictures"])
def test_sdk_calls_with_exactly_sized_frames(trust, monkeypatch):
    """The SDK entry points get the frame without any slack behind it (the length comes from the picture's section table)
    and, with HVQM4_AMD_TRUST_PICTURES=1, keep the reference pictures they wrote themselves on the device."""
    import ctypes as C
    from hvqm4_amd import sdk
    from hvqm4_amd._lib import lib
    from oracle import bridge
    if trust:
        monkeypatch.setenv("HVQM4_AMD_TRUST_PICTURES", "1")
    else:
        monkeypatch.delenv("HVQM4_AMD_TRUST_PICTURES", raising=False)
    for case in (clips.SMALL[3], clips.SMALL[4], clips.MEDIUM[2]):
        cl = clips.get(case)
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        pl = sdk.Player(cl.width, cl.height, 2, 2, cl.version == "1.5")
        lib().HVQM4SetMaxFrameSize(C.byref(pl.seqobj), max(len(p) for p in cl.pictures))
        for k, (ft, pic) in enumerate(zip(cl.kinds, cl.pictures)):
            # the rotation of Player.decode (h4m:2087-2137) with the frame handed over exactly as long as it is
            if ft != 0x30:
                pl.past, pl.future = pl.future, pl.past
            frame = (C.c_uint8 * len(pic)).from_buffer_copy(pic)
            if ft == 0x10:
                lib().HVQM4DecodeIpic(C.byref(pl.seqobj), C.cast(frame, C.c_char_p), pl.present.ctypes.data)
            elif ft == 0x20:
                lib().HVQM4DecodePpic(C.byref(pl.seqobj), C.cast(frame, C.c_char_p), pl.present.ctypes.data, pl.past.ctypes.data)
            else:
                lib().HVQM4DecodeBpic(C.byref(pl.seqobj), C.cast(frame, C.c_char_p), pl.present.ctypes.data, pl.past.ctypes.data, pl.future.ctypes.data)
            assert lib().HVQM4GetLastError() == 0
            assert np.array_equal(pl.present, want[k]), (case[0], k)
            if ft != 0x30:
                pl.present, pl.future = pl.future, pl.present
        pl.close()


def test_seeded_slice_of_the_randomised_parity_sweep(gpu_ctx):
    """200 clips of tools/parity_sweep.py (geometry, version, sampling, preset, GOP, shifts, ring size and flush cadence drawn from a
    fixed seed), host-parsed AND GPU-parsed, every picture against the oracle.  The full sweeps (thousands of clips per mode,
    tools/sweep.sh) found a real refusal bug in round 4 that no fixed clip had caught; this slice runs where the driver looks."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_sweep
    bad, pics, refused = parity_sweep.sweep(gpu_ctx, 200, 5005, "both", log=lambda *a, **k: None)
    assert bad == 0, f"{bad} of 200 random clips differ from the oracle"
    assert pics > 800 and refused == 0


def test_table_divisions_on_the_gpu(gpu_ctx):
    """the kernels compute the reference's divTable / mcdivTable entries (h4m:265-273: 256 / d for d < 16 -- the table stores 16 times
    that --, 0x1000 / d for d < 256, entry 0 = 0) with a reciprocal and no integer fix-up: every divisor, on the device"""
    import ctypes as C
    from hvqm4_amd._lib import check, lib
    out = (C.c_uint32 * 272)()
    check(lib().hvq_debug_table_divisions(gpu_ctx._h, out))
    assert list(out[:16]) == [0] + [256 // d for d in range(1, 16)]
    assert list(out[16:]) == [0] + [4096 // d for d in range(1, 256)]
