"""GPU: entropy parse ON the device (hvq_submit_many_device -> hvq_gparse.hip) followed by the reconstruction
kernels must give the oracle's pictures bit for bit -- no host core parses a bit of these streams."""
import os

import numpy as np
import pytest

from tests import clips

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ctx():
    from hvqm4_amd import batch
    ctx = batch.Context(0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("case", clips.SMALL + clips.MEDIUM + clips.REGRESSION, ids=lambda c: c[0])
def test_gpu_parsed_clip_matches_oracle(gpu_ctx, case):
    from hvqm4_amd import batch
    from oracle import bridge
    clip = clips.get(case)
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    got = batch.decode_clip(gpu_ctx, clip.data, gpu_parse=True)
    assert np.array_equal(got, want)
    if case in clips.SMALL and bridge.have_ref():       # ... and the unmodified reference decoder itself, when it is on this box (oracle/_ref)
        assert np.array_equal(got, bridge.ref_decode(clip.data, clip.n_pictures)[0]), "GPU-parsed pictures differ from the compiled reference"
    st = gpu_ctx.stats()
    assert st.gpu_parsed == clip.n_pictures
    if not os.environ.get("HVQM4_AMD_PARSE_FLAT") == "0":
        if case[0].startswith("longescape"):      # an overflow run of 300 symbols goes to the chains by design: same pictures
            assert st.gpu_parse_retried == 1
        elif case[0].startswith("bigscalars"):    # so do the runs of several hundred symbols behind its 16-bit-plus scalars (P and B picture)
            assert 1 <= st.gpu_parse_retried <= 2
        else:
            assert st.gpu_parse_retried == 0, "the flat parse path handed a regular picture to the chains"


@pytest.mark.parametrize("every", [1, 2, 5])
def test_nest_of_the_last_I_picture_survives_flushes(gpu_ctx, every):
    """P/B pictures with intra AOT blocks use the nest of the most recent I picture (h4m:1823 -> 1367); with one
    flush per few pictures that I picture sits in an earlier batch"""
    from hvqm4_amd import batch
    from oracle import bridge
    clip = clips.get(clips.SMALL[3])
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    got = batch.decode_clip(gpu_ctx, clip.data, gpu_parse=True, flush_every=every)
    assert np.array_equal(got, want)


def test_many_streams_one_parse_launch(gpu_ctx):
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    cases = [clips.get(c) for c in clips.SMALL[:8]]
    sids, per = [], []
    for cl in cases:
        hdr = parse_header(cl.data)
        sids.append(gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, cl.n_pictures + 3))
        per.append(list(video_pictures(cl.data)))
    a_s, a_t, a_p = [], [], []
    for k in range(max(len(p) for p in per)):
        for sid, p in zip(sids, per):
            if k < len(p):
                a_s.append(sid); a_t.append(p[k][0]); a_p.append(bytes(p[k][2]))
    gpu_ctx.submit_many_device(a_s, a_t, a_p)
    gpu_ctx.flush()
    assert gpu_ctx.stats().gpu_parsed == len(a_p)
    for sid, cl in zip(sids, cases):
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        for k in range(cl.n_pictures):
            assert np.array_equal(gpu_ctx.read_picture(sid, k), want[k]), (cl, k)
        gpu_ctx.close_stream(sid)


def test_host_and_gpu_parsed_streams_share_a_batch(gpu_ctx):
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    cl = clips.get(clips.SMALL[4])
    hdr = parse_header(cl.data)
    pics = list(video_pictures(cl.data))
    s_host = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, len(pics) + 3)
    s_dev = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, len(pics) + 3)
    for ft, _d, pic in pics:
        gpu_ctx.submit(s_host, ft, pic)
    gpu_ctx.submit_many_device([s_dev] * len(pics), [p[0] for p in pics], [bytes(p[2]) for p in pics])
    gpu_ctx.flush()
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    for k in range(cl.n_pictures):
        assert np.array_equal(gpu_ctx.read_picture(s_host, k), want[k])
        assert np.array_equal(gpu_ctx.read_picture(s_dev, k), want[k])
    # a stream keeps its parser
    with pytest.raises(Exception):
        gpu_ctx.submit(s_dev, pics[0][0], pics[0][2])
    with pytest.raises(Exception):
        gpu_ctx.submit_many_device([s_host], [pics[0][0]], [bytes(pics[0][2])])
    gpu_ctx.close_stream(s_host)
    gpu_ctx.close_stream(s_dev)


def test_corrupted_streams_do_not_fault_the_gpu_parser(gpu_ctx):
    """bit flips, truncation and garbage: the device parser must stay in bounds and terminate; pictures are either
    rejected by flush or decoded to something"""
    from hvqm4_amd.container import parse_header, video_pictures
    cl = clips.get(clips.SMALL[3])
    hdr = parse_header(cl.data)
    pics = [(ft, bytes(p)) for ft, _d, p in video_pictures(cl.data)]
    rng = np.random.default_rng(11)
    variants = []
    for ft, p in pics[:4]:
        a = bytearray(p)
        for _ in range(20):
            a[int(rng.integers(8, len(a)))] ^= 1 << int(rng.integers(0, 8))
        variants.append((ft, bytes(a)))
        variants.append((ft, p[:max(0x60, len(p) // 3)]))
        variants.append((ft, p[:0x50] + bytes(rng.integers(0, 256, len(p) - 0x50, dtype=np.uint8))))
    for ft, data in variants:
        sid = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, 4)
        try:
            gpu_ctx.submit_many_device([sid], [ft], [data])
            gpu_ctx.flush()
            gpu_ctx.sync()
        except Exception:
            pass
        gpu_ctx.close_stream(sid)
    # the context is still healthy
    test_nest_of_the_last_I_picture_survives_flushes(gpu_ctx, 2)


@pytest.mark.parametrize("w,h,samp", [(1920, 1088, "420"), (4096, 2176, "420"), (8192, 8, "420"), (8, 8192, "420"), (8192, 16, "444"), (2048, 8, "420"), (8, 2048, "420"), (1024, 16, "444"), (720, 576, "444"), (720, 576, "422"), (8, 1024, "422")])
def test_large_and_extreme_geometries(gpu_ctx, w, h, samp):
    """full-HD, 4K (557 056 luma blocks: 2 176 tiles, plane offsets beyond 8 MB), one-macroblock-high strips (longest DC row buffer, ragged
    last tiles) and wide 4:4:4, through both parsers; the oracle is the checker"""
    from hvqm4_amd import batch
    from hvqm4_amd.synth import SynthConfig, make_clip
    from oracle import bridge
    clip = make_clip(SynthConfig(width=w, height=h, gop="IPB", seed=w + h, sampling=samp, runoff_prob=0.2))
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    for gpu_parse in (False, True):
        got = batch.decode_clip(gpu_ctx, clip.data, gpu_parse=gpu_parse)
        assert np.array_equal(got, want), ("gpu parse" if gpu_parse else "host parse")


@pytest.mark.parametrize("gpu_parse", [False, True], ids=["host_parse", "gpu_parse"])
def test_pipelined_batches_begin_submit_next_end(gpu_ctx, gpu_parse):
    """streaming use: the next batch is submitted (copied, uploaded) while the batch in flight is parsed --
    hvq_flush_begin / hvq_submit_* / hvq_flush_end; two arenas alternate"""
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    cases = [clips.get(c) for c in (clips.SMALL[14], clips.SMALL[3], clips.SMALL[15])]      # 3 GOPs / 1 GOP / 1 GOP
    hdrs = [parse_header(cl.data) for cl in cases]
    pics = [list(video_pictures(cl.data)) for cl in cases]
    sids = [gpu_ctx.open_stream(h.width, h.height, h.h_samp, h.v_samp, h.is15, len(p) + 3) for h, p in zip(hdrs, pics)]
    step = 3                                                     # pictures per stream per batch
    nb = max((len(p) + step - 1) // step for p in pics)

    def submit(b):
        a_s, a_t, a_p = [], [], []
        for sid, p in zip(sids, pics):
            for ft, _d, pic in p[b * step:(b + 1) * step]:
                a_s.append(sid); a_t.append(ft); a_p.append(bytes(pic))
        if not a_p:
            return
        if gpu_parse:
            gpu_ctx.submit_many_device(a_s, a_t, a_p)
        else:
            gpu_ctx.submit_many(a_s, a_t, a_p, 2)

    submit(0)
    gpu_ctx.flush_begin()
    for b in range(1, nb):
        submit(b)                                               # while batch b-1 is in flight
        gpu_ctx.flush_end()
        gpu_ctx.flush_begin()
    gpu_ctx.flush_end()
    for sid, cl in zip(sids, cases):
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        for k in range(cl.n_pictures):
            assert np.array_equal(gpu_ctx.read_picture(sid, k), want[k]), (cl.width, k)
        gpu_ctx.close_stream(sid)


@pytest.mark.parametrize("submit_form", ["copy", "deferred", "arena", "host_parse"])
def test_flush_next_streams_like_the_plain_pair(gpu_ctx, submit_form):
    """hvq_flush_next = hvq_flush_end + hvq_flush_begin with the queued batch's parse kernel launched first: many small batches, so
    that the two parse-buffer sets and the two arenas each come round several times; three streams of different length (one runs
    out early, its later batches carry the others only); every picture against the oracle.  Host-parsed batches take the plain
    order inside the call."""
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    cases = [clips.get(c) for c in (clips.SMALL[14], clips.SMALL[3], clips.SMALL[15])]      # 3 GOPs / 1 GOP / 1 GOP
    hdrs = [parse_header(cl.data) for cl in cases]
    pics = [list(video_pictures(cl.data)) for cl in cases]
    sids = [gpu_ctx.open_stream(h.width, h.height, h.h_samp, h.v_samp, h.is15, len(p) + 3) for h, p in zip(hdrs, pics)]
    step = 2
    nb = max((len(p) + step - 1) // step for p in pics)
    keep = []                                                    # deferred copies read the caller's buffers until the next flush

    def submit(b):
        a_s, a_t, a_p = [], [], []
        for sid, p in zip(sids, pics):
            for ft, _d, pic in p[b * step:(b + 1) * step]:
                a_s.append(sid); a_t.append(ft); a_p.append(bytes(pic))
        if not a_p:
            return
        keep.append(a_p)
        if submit_form == "host_parse":
            gpu_ctx.submit_many(a_s, a_t, a_p, 2)
        elif submit_form == "arena":
            offs, at = [], 0
            for q in a_p:
                offs.append(at); at += gpu_ctx.arena_stride(len(q))
            view = gpu_ctx.arena_reserve(at)
            for q, o in zip(a_p, offs):
                view[o:o + len(q)] = np.frombuffer(q, np.uint8)
            gpu_ctx.submit_many_arena(a_s, a_t, offs, [len(q) for q in a_p])
        else:
            gpu_ctx.submit_many_device(a_s, a_t, a_p, defer=submit_form == "deferred")

    assert nb >= 6
    submit(0)
    gpu_ctx.flush_next()                                         # nothing in flight yet: a plain begin
    for b in range(1, nb):
        submit(b)
        gpu_ctx.flush_next()
    gpu_ctx.flush_next()                                         # nothing queued: a plain end
    for sid, cl in zip(sids, cases):
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        for k in range(cl.n_pictures):
            assert np.array_equal(gpu_ctx.read_picture(sid, k), want[k]), (cl.width, k)
        gpu_ctx.close_stream(sid)


def test_paired_contexts_stream_like_one(gpu_ctx):
    """batch.PairedContexts: the streams dealt to two contexts of the GPU, one thread advancing them in turn with hvq_flush_next; every
    picture of every stream against the oracle, ordinals per stream as with one context"""
    from hvqm4_amd import batch
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    cases = [clips.get(c) for c in (clips.SMALL[14], clips.SMALL[3], clips.SMALL[15], clips.SMALL[4], clips.SMALL[3])]
    hdrs = [parse_header(cl.data) for cl in cases]
    pics = [list(video_pictures(cl.data)) for cl in cases]
    pair = batch.PairedContexts(0)
    sids = [pair.open_stream(h.width, h.height, h.h_samp, h.v_samp, h.is15, len(p) + 3) for h, p in zip(hdrs, pics)]
    step = 2
    nb = max((len(p) + step - 1) // step for p in pics)
    keep, ords = [], [[] for _ in sids]

    def submit(b):
        a_s, a_t, a_p, a_i = [], [], [], []
        for i, (sid, p) in enumerate(zip(sids, pics)):
            for ft, _d, pic in p[b * step:(b + 1) * step]:
                a_s.append(sid); a_t.append(ft); a_p.append(bytes(pic)); a_i.append(i)
        if a_p:
            keep.append(a_p)
            for i, o in zip(a_i, pair.submit_many_device(a_s, a_t, a_p, defer=True)):
                ords[i].append(o)

    submit(0)
    pair.flush_begin()
    for b in range(1, nb):
        submit(b)
        pair.flush_next()
    pair.flush_end()
    pair.sync()
    for i, (sid, cl) in enumerate(zip(sids, cases)):
        assert ords[i] == list(range(cl.n_pictures))
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        for k in range(cl.n_pictures):
            assert np.array_equal(pair.read_picture(sid, k), want[k]), (i, k)
    assert sum(st.gpu_parsed for st in pair.stats()) > 0
    pair.close()


def test_flush_next_and_the_plain_pair_mix(gpu_ctx):
    """a caller may change between the two forms batch by batch (the parse-buffer set in use follows the batch in flight), and a
    replay of the last ended batch still reads that batch's blobs"""
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    cl = clips.get(clips.SMALL[14])
    hdr = parse_header(cl.data)
    pics = list(video_pictures(cl.data))
    sid = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, len(pics) + 3)
    step = 2
    nb = (len(pics) + step - 1) // step

    def submit(b):
        part = pics[b * step:(b + 1) * step]
        gpu_ctx.submit_many_device([sid] * len(part), [ft for ft, _d, _p in part], [bytes(p) for _ft, _d, p in part])

    submit(0)
    gpu_ctx.flush_begin()
    for b in range(1, nb):
        submit(b)
        if b % 3 == 0:
            gpu_ctx.flush_end(); gpu_ctx.flush_begin()
        elif b % 3 == 1:
            gpu_ctx.flush_next()
        else:
            gpu_ctx.flush_next()
            gpu_ctx.replay(2)                                    # ends the batch begun just now, replays it twice
    gpu_ctx.flush_end()
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    for k in range(cl.n_pictures):
        assert np.array_equal(gpu_ctx.read_picture(sid, k), want[k]), k
    gpu_ctx.close_stream(sid)


def test_pathological_trees_cannot_stall_the_gpu_parser(gpu_ctx):
    """A one-leaf DC tree whose only value lies outside the overflow window makes every DC read spin until its cap
    (the reference would never return).  Full-size pictures of that kind, of zeros and of ones must come back
    quickly -- the caps bound a picture's decode time whatever the stream says."""
    import struct
    import time
    from hvqm4_amd.container import parse_header, video_pictures
    from hvqm4_amd.synth import SynthConfig, make_clip
    clip = make_clip(SynthConfig(width=640, height=480, gop="IPB", seed=77))
    hdr = parse_header(clip.data)
    variants = []
    for ft, _d, pic in video_pictures(clip.data):
        p = bytearray(pic)
        data = 8 + (0x40 if ft == 0x10 else 0x44)
        off = data + struct.unpack_from(">I", p, 8 + 4 * 4)[0] + 4          # section 4 = DC buffer of the luma plane
        p[0] = 0                                                             # dc_shift 0: window is (-128, 127)
        p[off:off + 2] = b"\x3f\x80"                                         # tree = single leaf 0x7F
        variants.append((ft, bytes(p)))
        variants.append((ft, bytes(p[:0x60]) + bytes(len(p) - 0x60)))        # all zero after the header
        variants.append((ft, bytes(p[:0x60]) + b"\xff" * (len(p) - 0x60)))   # all ones
    t0 = time.time()
    for ft, data in variants:
        sid = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, 4)
        try:
            gpu_ctx.submit_many_device([sid], [ft], [data])
            gpu_ctx.flush()
            gpu_ctx.sync()
        except Exception:
            pass
        gpu_ctx.close_stream(sid)
    assert time.time() - t0 < 20.0


def test_reading_a_picture_of_the_batch_in_flight_ends_it(gpu_ctx):
    from hvqm4_amd.container import parse_header, video_pictures
    from oracle import bridge
    cl = clips.get(clips.SMALL[3])
    hdr = parse_header(cl.data)
    pics = list(video_pictures(cl.data))
    sid = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, len(pics) + 3)
    gpu_ctx.submit_many_device([sid] * len(pics), [p[0] for p in pics], [bytes(p[2]) for p in pics])
    gpu_ctx.flush_begin()                                  # no flush_end: read_picture has to finish the batch itself
    want = bridge.oracle_decode(cl.data, cl.n_pictures)
    for k in range(cl.n_pictures):
        assert np.array_equal(gpu_ctx.read_picture(sid, k), want[k])
    gpu_ctx.close_stream(sid)


def test_a_rejected_picture_fails_the_flush_and_leaves_the_context_usable(gpu_ctx):
    """a prefix tree that nests deeper than 255 levels is rejected by the device parser (GP_ST_BADTREE): hvq_flush_end
    reports it, and the next batch decodes normally"""
    import struct
    from hvqm4_amd._lib import HvqError
    from hvqm4_amd.container import parse_header, video_pictures
    cl = clips.get(clips.SMALL[3])
    hdr = parse_header(cl.data)
    ft, _d, pic = next(iter(video_pictures(cl.data)))
    bad = bytearray(pic)
    off = 8 + 0x40 + struct.unpack_from(">I", bad, 8)[0] + 4          # section 0 carries the block-kind tree
    bad[off:off + 64] = b"\xff" * 64                                   # 512 inner nodes in a row
    sid = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, 4)
    gpu_ctx.submit_many_device([sid], [ft], [bytes(bad)])
    gpu_ctx.flush_begin()
    with pytest.raises(HvqError):
        gpu_ctx.flush_end()
    gpu_ctx.close_stream(sid)
    test_nest_of_the_last_I_picture_survives_flushes(gpu_ctx, 2)


def test_a_picture_whose_overflow_runs_never_end_is_refused_by_both_parsers(gpu_ctx):
    """A one-leaf DC tree whose value lies outside the overflow window: the reference would sum for ever (h4m:654-664).  Both
    parsers stop at their cap, flag the picture (HVQ_F_CAPPED) and the back end refuses it -- never different pixels.  (The
    flat parse path hands such a picture to the chains first; a long but finite run decodes exactly: clip longescape64x48.)"""
    import struct
    from hvqm4_amd._lib import HVQ_E_UNSUPPORTED, HvqError
    from hvqm4_amd.container import parse_header, video_pictures
    from hvqm4_amd.synth import SynthConfig, make_clip
    clip = make_clip(SynthConfig(width=96, height=64, gop="I", seed=5))
    hdr = parse_header(clip.data)
    ft, _d, pic = next(iter(video_pictures(clip.data)))
    p = bytearray(pic)
    off = 8 + 0x40 + struct.unpack_from(">I", p, 8 + 4 * 4)[0] + 4          # section 4 = DC buffer of the luma plane
    p[0] = 0
    p[off:off + 2] = b"\x3f\x80"                                          # tree = single leaf 0x7F
    for gpu_parse in (False, True):
        sid = gpu_ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15, 4)
        with pytest.raises(HvqError) as e:
            if gpu_parse:
                gpu_ctx.submit_many_device([sid], [ft], [bytes(p)])
                gpu_ctx.flush()
            else:
                gpu_ctx.submit(sid, ft, bytes(p))
        assert e.value.code == HVQ_E_UNSUPPORTED and "overflow-symbol" in str(e.value)
        gpu_ctx.close_stream(sid)
