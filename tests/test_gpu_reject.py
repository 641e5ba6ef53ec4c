"""GPU: pictures this back end refuses and what a refusal leaves behind.

* A picture with an overflow-symbol run that NEVER ends (a one-leaf tree whose only value lies outside the overflow window: the
  reference sums for as long as the stream says, h4m:654-677, and would not return): HVQ_F_CAPPED.  Every entry point reports
  HVQ_E_UNSUPPORTED instead of decoding something else, `present` stays untouched, the stream resumes at its next I picture.
  (A run that is merely LONG -- 5 000 symbols, beyond the fast parsers' cap of 4096 -- is decoded like the reference since round 5:
  the host parser follows it to its end, bounded by the bits the picture has left, and a GPU-parsed picture that comes back capped
  is parsed again on the host; rounds 3-4 refused it.  P pictures with future-referencing macroblocks, refused in round 2, are
  decoded like the reference: tests/clips.py pselfref*.)
* One bad picture must not poison the batch: the other streams of the same flush decode bit-exactly."""
import numpy as np
import pytest

from tests import clips

pytestmark = pytest.mark.gpu


def _long_run_clip(seed=5, w=64, h=48, gop="IPBBPB"):
    """a clip whose SECOND picture (the first P) carries an overflow-symbol run of 5000 symbols: over the fast parsers' cap"""
    from hvqm4_amd.synth import SynthConfig, make_clip
    return make_clip(SynthConfig(width=w, height=h, gop=gop, seed=seed, long_escape_pb=5000))


class _Patched:
    """a clip with one picture replaced (what the container iterator would hand out)"""
    def __init__(self, clip, pics):
        self.width, self.height, self.version, self.data, self.n_pictures, self.pics = clip.width, clip.height, clip.version, clip.data, clip.n_pictures, pics


def _self_ref_clip(seed=5, w=64, h=48, gop="IPBBPB"):
    """a clip whose SECOND picture (the first P) can never be decoded like the reference: its luma DC tree is a single leaf 0x7F
    with dc_shift 0, outside the overflow window (-128, 127) -- every overflow read of the section spins for ever"""
    import struct
    from hvqm4_amd.synth import SynthConfig, make_clip
    clip = make_clip(SynthConfig(width=w, height=h, gop=gop, seed=seed))
    pics = _pics(clip)
    b = bytearray(pics[1][1])
    off = 8 + 0x44 + struct.unpack_from(">I", b, 8 + 4 * 4)[0] + 4           # section 4 = DC buffer of the luma plane
    b[0] = 0
    b[off:off + 2] = b"\x3f\x80"
    pics[1] = (pics[1][0], bytes(b))
    return _Patched(clip, pics)


def _pics(cl):
    from hvqm4_amd.container import video_pictures
    if hasattr(cl, "pics"):
        return list(cl.pics)
    return [(ft, bytes(p)) for ft, _d, p in video_pictures(cl.data)]


def test_host_parsed_capped_picture_is_refused_and_the_stream_resumes_at_an_I_picture(gpu_ctx):
    from hvqm4_amd._lib import HVQ_E_STATE, HVQ_E_UNSUPPORTED, HvqError
    from oracle import bridge
    bad = _self_ref_clip()
    good = clips.get(clips.SMALL[3])
    bp, gp = _pics(bad), _pics(good)
    sb = gpu_ctx.open_stream(bad.width, bad.height, 2, 2, True, 8)
    sg = gpu_ctx.open_stream(good.width, good.height, 2, 2, True, len(gp) + 3)
    gpu_ctx.submit(sb, *bp[0])                                   # the I picture is fine
    with pytest.raises(HvqError) as e:
        gpu_ctx.submit(sb, *bp[1])                               # P with an overflow run that never ends
    assert e.value.code == HVQ_E_UNSUPPORTED and "overflow-symbol" in str(e.value)
    with pytest.raises(HvqError) as e:
        gpu_ctx.submit(sb, *bp[2])                               # the B picture would reference the refused P
    assert e.value.code == HVQ_E_STATE
    for ft, p in gp:
        gpu_ctx.submit(sg, ft, p)
    gpu_ctx.flush()
    want_g = bridge.oracle_decode(good.data, good.n_pictures)
    for k in range(len(gp)):
        assert np.array_equal(gpu_ctx.read_picture(sg, k), want_g[k])
    want_b = bridge.oracle_decode(bad.data, 1)
    assert np.array_equal(gpu_ctx.read_picture(sb, 0), want_b[0])
    # a legal clip on the same stream, starting with its I picture, decodes exactly
    ok = clips.get(clips.SMALL[3])
    assert (ok.width, ok.height) == (bad.width, bad.height)
    first = None
    for ft, p in _pics(ok):
        o = gpu_ctx.submit(sb, ft, p)
        first = o if first is None else first
    gpu_ctx.flush()
    want = bridge.oracle_decode(ok.data, ok.n_pictures)
    for k in range(ok.n_pictures):
        assert np.array_equal(gpu_ctx.read_picture(sb, first + k), want[k])
    gpu_ctx.close_stream(sb); gpu_ctx.close_stream(sg)


@pytest.mark.parametrize("what", ["capped", "bad_tree"])
def test_one_refused_picture_does_not_poison_the_other_streams_of_a_gpu_parsed_batch(gpu_ctx, what):
    import struct
    from hvqm4_amd._lib import HVQ_E_STATE, HVQ_E_UNSUPPORTED, HvqError
    from oracle import bridge
    good = [clips.get(clips.SMALL[3]), clips.get(clips.SMALL[4]), clips.get(clips.SMALL[11])]
    bad = _self_ref_clip(seed=9)
    bp = _pics(bad)
    if what == "bad_tree":                                       # a tree nested deeper than 255: GP_ST_BADTREE at picture 1
        legal = clips.get(clips.SMALL[3])
        bad, bp = legal, _pics(legal)
        b = bytearray(bp[1][1])
        off = 8 + 0x44 + struct.unpack_from(">I", b, 8)[0] + 4
        b[off:off + 64] = b"\xff" * 64
        bp[1] = (bp[1][0], bytes(b))
    streams = [(cl, _pics(cl), gpu_ctx.open_stream(cl.width, cl.height, 2, 2, cl.version == "1.5", 12)) for cl in good]
    sb = gpu_ctx.open_stream(bad.width, bad.height, 2, 2, True, 12)
    sids, fts, data = [], [], []
    for k in range(max(len(bp), max(len(p) for _c, p, _s in streams))):      # decode-order interleave over all four streams
        for cl, pics, sid in streams + [(bad, bp, sb)]:
            if k < len(pics):
                sids.append(sid); fts.append(pics[k][0]); data.append(pics[k][1])
    gpu_ctx.submit_many_device(sids, fts, data)
    with pytest.raises(HvqError) as e:
        gpu_ctx.flush()
    if what == "capped":
        assert e.value.code == HVQ_E_UNSUPPORTED
    assert f"stream {sb} picture 1" in str(e.value)
    gpu_ctx.sync()
    for cl, pics, sid in streams:                                # every other stream: all pictures, bit-exact
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        for k in range(len(pics)):
            assert np.array_equal(gpu_ctx.read_picture(sid, k), want[k]), (cl.width, k)
    want_b = bridge.oracle_decode(bad.data, 1)
    assert np.array_equal(gpu_ctx.read_picture(sb, 0), want_b[0])           # what came before the refused picture stands
    for k in range(1, len(bp)):                                             # the refused picture and what followed it: not resident
        with pytest.raises(HvqError) as e2:
            gpu_ctx.read_picture(sb, k)
        assert e2.value.code == HVQ_E_STATE
    with pytest.raises(HvqError) as e3:                                     # the stream waits for an I picture ...
        gpu_ctx.submit_many_device([sb], [bp[2][0]], [bp[2][1]])
    assert e3.value.code == HVQ_E_STATE
    ok = clips.get(clips.SMALL[3])                                          # ... and then decodes exactly again
    op = _pics(ok)
    ords = gpu_ctx.submit_many_device([sb] * len(op), [p[0] for p in op], [p[1] for p in op])
    gpu_ctx.flush()
    want = bridge.oracle_decode(ok.data, ok.n_pictures)
    for k, o in enumerate(ords):
        assert np.array_equal(gpu_ctx.read_picture(sb, o), want[k])
    # the resident batch replays (the dropped picture's tiles are padding entries)
    assert gpu_ctx.replay(2) > 0
    for _c, _p, sid in streams:
        gpu_ctx.close_stream(sid)
    gpu_ctx.close_stream(sb)


def test_sdk_call_refuses_the_picture_and_leaves_present_untouched():
    from hvqm4_amd import sdk
    from hvqm4_amd._lib import HVQ_E_UNSUPPORTED, HvqError
    from oracle import bridge
    bad = _self_ref_clip(seed=11, gop="IP")
    bp = _pics(bad)
    pl = sdk.Player(bad.width, bad.height, 2, 2, True)
    got_i = pl.decode(*bp[0])
    assert np.array_equal(got_i, bridge.oracle_decode(bad.data, 1)[0])
    pl.present[:] = 0xA5
    with pytest.raises(HvqError) as e:
        pl.decode(*bp[1])
    assert e.value.code == HVQ_E_UNSUPPORTED
    assert (pl.present == 0xA5).all(), "`present` must be left as it was"
    pl.close()


def test_clamped_nest_origin_is_refused_unless_opted_in(gpu_ctx, monkeypatch):
    """an I picture whose nest window leaves the bordered block map altogether (the reference would read foreign memory;
    a window that merely overlaps the border is reproduced exactly, tests/clips.py nest_border*): refused by default, decoded
    with the clamped origin (and without faulting) under HVQM4_AMD_ALLOW_CLAMPED=1"""
    import struct
    from hvqm4_amd._lib import HVQ_E_UNSUPPORTED, HvqError
    from hvqm4_amd.synth import SynthConfig, make_clip
    cl = make_clip(SynthConfig(width=320, height=240, gop="I", seed=3))
    ft, pic = _pics(cl)[0]
    b = bytearray(pic)
    struct.pack_into(">H", b, 4, 80 - 70 + 5)                     # nest_x: 5 columns too far to the right ...
    struct.pack_into(">H", b, 6, 60 - 38 + 3)                     # ... and nest_y 3 rows too low: the window leaves the bordered map
    sid = gpu_ctx.open_stream(320, 240, 2, 2, True, 4)
    monkeypatch.delenv("HVQM4_AMD_ALLOW_CLAMPED", raising=False)
    with pytest.raises(HvqError) as e:
        gpu_ctx.submit(sid, ft, bytes(b))
    assert e.value.code == HVQ_E_UNSUPPORTED
    monkeypatch.setenv("HVQM4_AMD_ALLOW_CLAMPED", "1")
    o = gpu_ctx.submit(sid, ft, bytes(b))
    gpu_ctx.flush()
    assert gpu_ctx.read_picture(sid, o).size == 320 * 240 * 3 // 2
    assert gpu_ctx.stats().flags_or & 0x20
    gpu_ctx.close_stream(sid)


@pytest.mark.parametrize("step", ["flush_end", "flush_next"])
def test_rejection_while_the_next_batch_of_the_stream_is_already_queued(gpu_ctx, step):
    """Streaming: batch N is judged in hvq_flush_end (or hvq_flush_next, which has queued the parse of N+1 by then) when batch N+1 is
    queued already.  The P/B pictures of the rejected stream
    in N+1 follow the rejected picture: they are dropped -- and must read as NOT resident, not as whatever their slot held --
    up to the stream's next I picture, which restarts it without any HVQ_E_STATE in between."""
    from hvqm4_amd._lib import HVQ_E_STATE, HVQ_E_UNSUPPORTED, HvqError
    from oracle import bridge
    bad = _self_ref_clip(seed=21, gop="IPBB")                     # picture 1 (P) is over the cap
    good = clips.get(clips.SMALL[3])                             # 64x48 1.5, IPBBPBB
    bp, gp = _pics(bad), _pics(good)
    sb = gpu_ctx.open_stream(bad.width, bad.height, 2, 2, True, 12)
    so = gpu_ctx.open_stream(good.width, good.height, 2, 2, True, 12)
    # batch N: the bad clip's I and P on sb, the good clip's first two pictures on so
    gpu_ctx.submit_many_device([sb, so, sb, so], [bp[0][0], gp[0][0], bp[1][0], gp[1][0]], [bp[0][1], gp[0][1], bp[1][1], gp[1][1]])
    gpu_ctx.flush_begin()
    # batch N+1, queued before N is judged: two more B pictures of the bad clip, then a whole good clip on the same stream
    o_b = gpu_ctx.submit_many_device([sb, sb], [bp[2][0], bp[3][0]], [bp[2][1], bp[3][1]])
    o_g = gpu_ctx.submit_many_device([sb] * len(gp), [p[0] for p in gp], [p[1] for p in gp])
    o_o = gpu_ctx.submit_many_device([so] * (len(gp) - 2), [p[0] for p in gp[2:]], [p[1] for p in gp[2:]])
    with pytest.raises(HvqError) as e:
        getattr(gpu_ctx, step)()
    assert e.value.code == HVQ_E_UNSUPPORTED and f"stream {sb} picture 1" in str(e.value)
    gpu_ctx.flush()                                              # batch N+1 (begun already by flush_next): no error of its own
    assert gpu_ctx.stats().dropped == 2
    for o in o_b:                                                # the B pictures behind the rejected P: not resident
        with pytest.raises(HvqError) as e2:
            gpu_ctx.read_picture(sb, o)
        assert e2.value.code == HVQ_E_STATE
    want = bridge.oracle_decode(good.data, good.n_pictures)
    for k, o in enumerate(o_g):                                  # the stream restarted at the queued I picture
        assert np.array_equal(gpu_ctx.read_picture(sb, o), want[k]), k
    for k in range(2):
        assert np.array_equal(gpu_ctx.read_picture(so, k), want[k])
    for k, o in enumerate(o_o):
        assert np.array_equal(gpu_ctx.read_picture(so, o), want[k + 2])
    gpu_ctx.submit_many_device([sb], [gp[1][0]], [gp[1][1]])     # and takes P pictures again without an I picture first
    gpu_ctx.flush()
    gpu_ctx.close_stream(sb); gpu_ctx.close_stream(so)


@pytest.mark.gpu
def test_ring_of_4_gib_is_refused_before_anything_is_allocated(gpu_ctx):
    """ring-relative 32-bit offsets (HvqJob::ref0_off, MC source offsets, window offsets) must not wrap: 22 slots of an
    8192x8192 4:4:4 stream are 4.4 GB -- refused with HVQ_E_OVERFLOW, and the context stays usable"""
    from hvqm4_amd._lib import HVQ_E_OVERFLOW, HvqError
    with pytest.raises(HvqError) as e:
        gpu_ctx.open_stream(8192, 8192, 1, 1, True, 21)
    assert e.value.code == HVQ_E_OVERFLOW
    sid = gpu_ctx.open_stream(64, 48, 2, 2, True, 4)
    gpu_ctx.close_stream(sid)


def test_an_overflow_run_beyond_the_fast_cap_is_decoded_like_the_reference_by_every_entry_point(gpu_ctx):
    """5 000 overflow symbols in one value (the fast parsers stop at 4096): the reference sums them all (h4m:654-677).  SDK symbols,
    batched path with the host parser and batched path with the GPU parser (whose capped picture is parsed again on the host) must
    give the oracle's pictures -- and the compiled reference's, when it is on this box."""
    from hvqm4_amd import sdk
    from hvqm4_amd.batch import decode_clip
    from oracle import bridge
    for seed, gop in ((5, "IPBBPB"), (12, "IPB")):
        clip = _long_run_clip(seed=seed, gop=gop)
        want = bridge.oracle_decode(clip.data, clip.n_pictures)
        if bridge.have_ref():
            assert np.array_equal(want, bridge.ref_decode(clip.data, clip.n_pictures)[0])
        got = decode_clip(gpu_ctx, clip.data)
        assert np.array_equal(got, want), "host parser"
        got = decode_clip(gpu_ctx, clip.data, gpu_parse=True)
        assert np.array_equal(got, want), "GPU parser + host re-parse of the capped picture"
        assert gpu_ctx.stats().dropped == 0
        pl = sdk.Player(clip.width, clip.height, 2, 2, True)
        got = np.stack([pl.decode(ft, p) for ft, p in _pics(clip)])
        pl.close()
        assert np.array_equal(got, want), "SDK symbols"


def test_host_reparses_of_capped_pictures_are_bounded_per_batch(gpu_ctx):
    """A GPU-parsed picture that comes back capped (an overflow run beyond 4096 symbols) is parsed again on the host inside the flush -- serially,
    on the caller's thread.  At most 32 such pictures per batch (round 6): a crafted batch whose pictures are ALL capped must not turn a flush
    of milliseconds into seconds for every stream in it.  40 streams, each with one capped P picture: 32 are decoded like the oracle, 8 are
    refused (HVQ_E_UNSUPPORTED, dropped), and nothing else of the batch is touched."""
    from hvqm4_amd._lib import HVQ_E_STATE, HvqError
    from oracle import bridge
    clip = _long_run_clip(seed=12, gop="IP")
    want = bridge.oracle_decode(clip.data, clip.n_pictures)
    pics = _pics(clip)
    n = 40
    sids = [gpu_ctx.open_stream(clip.width, clip.height, 2, 2, clip.version == "1.5", 4) for _ in range(n)]
    s_, f_, d_ = [], [], []
    for k in range(2):
        for sid in sids:
            s_.append(sid); f_.append(pics[k][0]); d_.append(pics[k][1])
    gpu_ctx.submit_many_device(s_, f_, d_)
    with pytest.raises(HvqError) as e:
        gpu_ctx.flush()
    assert e.value.code == -8                                   # HVQ_E_UNSUPPORTED: the first picture beyond the bound
    gpu_ctx.sync()
    assert gpu_ctx.stats().dropped == 8
    decoded = refused = 0
    for sid in sids:
        assert np.array_equal(gpu_ctx.read_picture(sid, 0), want[0])          # every I picture is there
        try:
            got = gpu_ctx.read_picture(sid, 1)
        except HvqError as err:
            assert err.code == HVQ_E_STATE
            refused += 1
            continue
        assert np.array_equal(got, want[1])
        decoded += 1
    assert (decoded, refused) == (32, 8)
    for sid in sids:
        gpu_ctx.close_stream(sid)
