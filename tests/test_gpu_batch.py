"""GPU: the batched device-resident path beyond single clips -- mixed sizes and versions in one launch
sequence (BASELINE config 4 in miniature), replay, slot reuse, malformed input."""
import os
import numpy as np
import pytest

from tests import clips

pytestmark = pytest.mark.gpu


def _submit_all(ctx, cl, nslots=None):
    from hvqm4_amd.container import parse_header, video_pictures
    hdr = parse_header(cl.data)
    pics = list(video_pictures(cl.data))
    sid = ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, nslots or len(pics) + 3)
    return sid, pics


NQ = 2 if os.environ.get("HVQM4_AMD_QUEUES", "") == "2" else 1      # launch queues forced by the mode test (tests/test_gpu_modes.py)


def test_mixed_batch_interleaved_streams(gpu_ctx):
    """streams of different geometry and version decoded by shared launches, submitted round-robin"""
    from oracle import bridge
    cases = [clips.SMALL[3], clips.SMALL[4], clips.SMALL[5], clips.SMALL[8], clips.SMALL[11], clips.MEDIUM[2],
             clips.SMALL[3], clips.SMALL[14]]
    streams = []
    for c in cases:
        cl = clips.get(c)
        sid, pics = _submit_all(gpu_ctx, cl)
        streams.append((cl, sid, pics))
    k = 0
    while any(k < len(p) for _c, _s, p in streams):
        for cl, sid, pics in streams:
            if k < len(pics):
                gpu_ctx.submit(sid, pics[k][0], pics[k][2])
        k += 1
    gpu_ctx.flush()
    st = gpu_ctx.stats()
    assert st.pictures == sum(len(p) for _c, _s, p in streams)
    assert st.launches <= NQ * max(len(p) for _c, _s, p in streams)     # pictures of many streams share launches (per launch queue)
    for cl, sid, pics in streams:
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        for i in range(len(pics)):
            assert np.array_equal(gpu_ctx.read_picture(sid, i), want[i]), (cl.width, cl.height, i)
        gpu_ctx.close_stream(sid)


def test_replay_is_idempotent(gpu_ctx):
    from oracle import bridge
    cl = clips.get(clips.MEDIUM[1])              # 640x480 IPBBPBB
    sid, pics = _submit_all(gpu_ctx, cl)
    for ft, _d, pic in pics:
        gpu_ctx.submit(sid, ft, pic)
    gpu_ctx.flush()
    ms = gpu_ctx.replay(3)
    assert ms > 0
    # the stage form: everything a new batch costs behind its parse (the launch queues forked and joined per pass, as in a flush)
    assert gpu_ctx.replay_stage(2, 1) > 0
    assert gpu_ctx.replay_stage(2, 2) == 0          # (timed the two-pass variant's queue build until round 5: nothing to run since)
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    for i in range(len(pics)):
        assert np.array_equal(gpu_ctx.read_picture(sid, i), want[i])
    gpu_ctx.close_stream(sid)


def test_full_gop_in_few_launches_with_small_ring(gpu_ctx):
    """16-picture GOP, 6 slots: dependency levels (not one launch per picture); the newest pictures stay readable"""
    from hvqm4_amd.synth import SynthConfig, make_clip
    from oracle import bridge
    cl = make_clip(SynthConfig(width=128, height=96, gop="IPBBPBBPBBPBBPBB", seed=77))
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    sid, pics = _submit_all(gpu_ctx, cl, nslots=6)
    for ft, _d, pic in pics:
        gpu_ctx.submit(sid, ft, pic)
    gpu_ctx.flush()
    st = gpu_ctx.stats()
    assert st.launches < NQ * len(pics)
    checked = 0
    for i in range(len(pics)):
        try:
            got = gpu_ctx.read_picture(sid, i)
        except Exception:
            continue
        assert np.array_equal(got, want[i]), i
        checked += 1
    assert checked >= 4
    gpu_ctx.close_stream(sid)


def test_malformed_pictures_do_not_fault_the_gpu(gpu_ctx):
    """bitstream fuzz: the reference reads and writes out of bounds on such input; here every
    device address is clamped or derived from host-validated sizes, so the kernel must simply complete."""
    rng = np.random.default_rng(11)
    cl = clips.get(clips.SMALL[3])
    sid = gpu_ctx.open_stream(cl.width, cl.height, 2, 2, True, 4)
    good = cl.pictures
    n = 0
    for trial in range(40):
        src = bytearray(good[trial % len(good)])
        ft = cl.kinds[trial % len(good)]
        for _ in range(int(rng.integers(1, 30))):
            src[int(rng.integers(0, len(src)))] = int(rng.integers(0, 256))
        if trial % 5 == 0:
            src = src[: max(0x60, len(src) // 2)]
        try:
            gpu_ctx.submit(sid, ft, bytes(src))
            n += 1
        except Exception:
            pass                     # rejected on the host (overflow / bad argument) is fine
        gpu_ctx.flush()
    gpu_ctx.sync()
    assert n > 0
    # the context is still healthy: a clean clip decodes bit-exactly afterwards
    from hvqm4_amd.batch import decode_clip
    from oracle import bridge
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    assert np.array_equal(decode_clip(gpu_ctx, cl.data), want)
    gpu_ctx.close_stream(sid)


def test_threaded_submit_equals_sequential(gpu_ctx):
    """hvq_submit_many: parse on a thread pool, queue in array order -> same pictures as one-by-one submission"""
    from oracle import bridge
    cases = [clips.SMALL[3], clips.SMALL[4], clips.SMALL[11], clips.SMALL[12], clips.SMALL[5]]
    streams = []
    for c in cases:
        cl = clips.get(c)
        sid, pics = _submit_all(gpu_ctx, cl)
        streams.append((cl, sid, pics))
    a_sid, a_ft, a_pic = [], [], []
    k = 0
    while any(k < len(p) for _c, _s, p in streams):
        for cl, sid, pics in streams:
            if k < len(pics):
                a_sid.append(sid); a_ft.append(pics[k][0]); a_pic.append(pics[k][2])
        k += 1
    ords = gpu_ctx.submit_many(a_sid, a_ft, a_pic, threads=4)
    assert len(ords) == len(a_pic)
    gpu_ctx.flush()
    for cl, sid, pics in streams:
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        for i in range(len(pics)):
            assert np.array_equal(gpu_ctx.read_picture(sid, i), want[i]), (cl.width, i)
        gpu_ctx.close_stream(sid)


# ---------------------------------------------------------------------------------------------- round 5: the submit contracts
def _device_pics(cl):
    from hvqm4_amd.container import parse_header, video_pictures
    hdr = parse_header(cl.data)
    return hdr, [(ft, bytes(pic)) for ft, _d, pic in video_pictures(cl.data)]


def test_synchronous_device_submit_lets_go_of_the_callers_buffers():
    """hvq_submit_many_device copies before it returns (advisor finding of round 4: it silently deferred the copy while a batch was
    in flight): the caller's buffers are poisoned right after every call -- also in streaming, with a batch in flight -- and the
    pictures must still be the oracle's"""
    import ctypes as C
    from hvqm4_amd import batch
    from hvqm4_amd._lib import check, lib
    from oracle import bridge
    cl = clips.get(clips.SMALL[3])
    hdr, pics = _device_pics(cl)
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    ctx = batch.Context(0)
    sid = ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, 3 * len(pics) + 3)
    n = len(pics)

    def submit_and_poison():
        bufs = [C.create_string_buffer(p, len(p)) for _ft, p in pics]
        a_p = (C.c_char_p * n)(*[C.cast(b, C.c_char_p) for b in bufs])
        a_s = (C.c_int * n)(*([sid] * n)); a_t = (C.c_int * n)(*[ft for ft, _p in pics])
        a_l = (C.c_size_t * n)(*[len(p) for _ft, p in pics]); a_o = (C.c_int * n)()
        check(lib().hvq_submit_many_device(ctx._h, n, a_s, a_t, a_p, a_l, a_o))
        for b in bufs:                               # the call has returned: the bytes are the caller's again
            C.memset(b, 0xA5, len(b))
        return list(a_o)

    o1 = submit_and_poison()
    ctx.flush_begin()
    o2 = submit_and_poison()                         # a batch is in flight: round 4 deferred exactly this copy
    ctx.flush_end()
    ctx.flush()
    for ords in (o1, o2):
        for i, o in enumerate(ords):
            assert np.array_equal(ctx.read_picture(sid, o), want[i]), (o, i)
    ctx.close()


def test_deferred_and_zero_copy_submits_decode_the_same_pictures():
    """hvq_submit_many_device_async (copy on a worker, joined by flush_begin) and hvq_arena_reserve + hvq_submit_many_arena (the
    caller writes the pinned arena itself, nothing is copied) against the oracle; bad arena layouts are refused"""
    from hvqm4_amd import batch
    from hvqm4_amd._lib import HVQ_E_ARG, HVQ_E_STATE, HvqError
    from oracle import bridge
    cl = clips.get(clips.SMALL[4])
    hdr, pics = _device_pics(cl)
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    ctx = batch.Context(0)
    sid = ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, 3 * len(pics) + 3)
    n = len(pics)
    fts = [ft for ft, _p in pics]
    o1 = ctx.submit_many_device([sid] * n, fts, [p for _ft, p in pics], defer=True)
    ctx.flush_begin()
    # zero copy while that batch is in flight: lay the pictures out in a reservation of the OTHER arena
    offs, at = [], 0
    for _ft, p in pics:
        offs.append(at); at += ctx.arena_stride(len(p))
    view = ctx.arena_reserve(at)
    with pytest.raises(HvqError) as e:               # one reservation at a time
        ctx.arena_reserve(256)
    assert e.value.code == HVQ_E_STATE
    view[:] = 0x5A                                   # the library must write the padding itself
    for (ft, p), o in zip(pics, offs):
        view[o:o + len(p)] = np.frombuffer(p, np.uint8)
    with pytest.raises(HvqError) as e:               # unaligned offset
        ctx.submit_many_arena([sid] * n, fts, [o + 8 for o in offs], [len(p) for _ft, p in pics])
    assert e.value.code == HVQ_E_ARG
    with pytest.raises(HvqError) as e:               # beyond the reservation
        ctx.submit_many_arena([sid] * n, fts, [o + at for o in offs], [len(p) for _ft, p in pics])
    assert e.value.code == HVQ_E_ARG
    with pytest.raises(HvqError) as e:               # a length whose padded span would wrap size_t (advisor, round 5): refused before any arithmetic on it
        ctx.submit_many_arena([sid] * n, fts, offs, [2 ** 64 - 16] + [len(p) for _ft, p in pics[1:]])
    assert e.value.code == HVQ_E_ARG
    o2 = ctx.submit_many_arena([sid] * n, fts, offs, [len(p) for _ft, p in pics])
    with pytest.raises(HvqError) as e:               # the reservation is spent
        ctx.submit_many_arena([sid] * n, fts, offs, [len(p) for _ft, p in pics])
    assert e.value.code == HVQ_E_STATE
    ctx.flush_end()
    ctx.flush()
    for ords in (o1, o2):
        for i, o in enumerate(ords):
            assert np.array_equal(ctx.read_picture(sid, o), want[i]), (o, i)
    assert ctx.stats().gpu_parsed == n
    ctx.close()


@pytest.mark.parametrize("defer", [False, True], ids=["sync", "deferred"])
def test_failed_bitstream_upload_drops_the_batch_and_the_stream_recovers(defer, monkeypatch):
    """an upload that fails: synchronously the call fails and leaves the context as it was; on the worker the failure is reported by
    the next flush_begin, the queued batch is gone, its stream waits for an I picture -- and then decodes again (copy_join's rollback)"""
    from hvqm4_amd import batch
    from hvqm4_amd._lib import HVQ_E_HIP, HVQ_E_STATE, HvqError
    from oracle import bridge
    cl = clips.get(clips.SMALL[3])
    hdr, pics = _device_pics(cl)
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    ctx = batch.Context(0)
    sid = ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, 2 * len(pics) + 3)
    n = len(pics)
    fts = [ft for ft, _p in pics]
    raw = [p for _ft, p in pics]
    monkeypatch.setenv("HVQM4_AMD_TEST_FAIL_COPY", "1")
    if defer:
        ctx.submit_many_device([sid] * n, fts, raw, defer=True)      # returns at once; the worker fails
        with pytest.raises(HvqError) as e:
            ctx.flush_begin()
        assert e.value.code == HVQ_E_HIP and "dropped" in str(e.value)
    else:
        with pytest.raises(HvqError) as e:
            ctx.submit_many_device([sid] * n, fts, raw)
        assert e.value.code == HVQ_E_HIP
    monkeypatch.delenv("HVQM4_AMD_TEST_FAIL_COPY")
    ctx.flush()                                                       # nothing is queued
    if defer:
        # the dropped pictures consumed ordinals; none of them reads as resident
        with pytest.raises(HvqError) as e:
            ctx.read_picture(sid, 0)
        assert e.value.code == HVQ_E_STATE
    ords = ctx.submit_many_device([sid] * n, fts, raw)                # the clip starts with an I picture: the stream resumes
    ctx.flush()
    for i, o in enumerate(ords):
        assert np.array_equal(ctx.read_picture(sid, o), want[i]), (o, i)
    ctx.close()


def test_lone_stream_parsed_by_four_threads_per_picture(gpu_ctx):
    """hvq_stream_set_parse_threads: the sections of ONE picture parsed side by side (what the SDK entry points do by default), through
    the batched API: same pictures as the oracle, and switching the count between pictures of a stream changes nothing"""
    from oracle import bridge
    for case in (clips.MEDIUM[1], clips.SMALL[17], clips.SMALL[9]):           # 640x480 IPBBPBB, 4:4:4, odd kinds 1.3
        cl = clips.get(case)
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        sid, pics = _submit_all(gpu_ctx, cl)
        assert gpu_ctx.set_parse_threads(sid, 4) == 4
        for k, (ft, _d, pic) in enumerate(pics):
            if k == len(pics) // 2:
                assert gpu_ctx.set_parse_threads(sid, 2) == 2
            gpu_ctx.submit(sid, ft, pic)
        gpu_ctx.flush()
        for i in range(len(pics)):
            assert np.array_equal(gpu_ctx.read_picture(sid, i), want[i]), (case[0], i)
        gpu_ctx.close_stream(sid)
