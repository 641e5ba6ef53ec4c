"""GPU: BASELINE.json's configs C4 and C5 at the shape one GPU sees them.

C4  per-GPU share of the 64-clip batch at 8 GPUs: 8 clips (4 x 320x240 + 4 x 640x480, HVQM4 1.3 and 1.5 alternating, IPBB...
    GOPs) submitted round-robin so that pictures of all clips share launches; both parsers; every picture against the
    oracle AND against the reference's own SHA-256 (tests/golden/manifest.json).
C5  per-GPU share of the 1024-stream stress: 128 concurrent 640x480 1.5 streams x 16 pictures from 8 distinct clips, GPU
    entropy parse, streaming flushes (begin / submit next / end), six slots per stream; every still-resident picture of 16
    streams against the oracle, and the newest picture of all 128 streams equal across the replicas of a clip."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import clips

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _pics(cl):
    from hvqm4_amd.container import video_pictures
    return [(ft, bytes(p)) for ft, _d, p in video_pictures(cl.data)]


@pytest.fixture(scope="module")
def c4_share():
    import bench
    cls = bench.gen_clips([cfg for _n, cfg in clips.C4_SHARE], min(8, os.cpu_count() or 1))
    return [(name, cl) for (name, _cfg), cl in zip(clips.C4_SHARE, cls)]


@pytest.mark.parametrize("gpu_parse", [False, True], ids=["host_parse", "gpu_parse"])
def test_c4_share_mixed_sizes_and_versions_share_launches(gpu_ctx, c4_share, gpu_parse):
    from oracle import bridge
    manifest = json.load(open(os.path.join(HERE, "golden", "manifest.json")))["clips"]
    assert {(cl.width, cl.version) for _n, cl in c4_share} == {(320, "1.3"), (320, "1.5"), (640, "1.3"), (640, "1.5")}
    streams = []
    for name, cl in c4_share:
        assert hashlib.sha256(cl.data).hexdigest() == manifest[name]["clip_sha256"], "generator drift"
        streams.append((name, cl, _pics(cl), gpu_ctx.open_stream(cl.width, cl.height, 2, 2, cl.version == "1.5", 16 + 3)))
    sids, fts, data = [], [], []
    for k in range(16):                                        # picture k of every clip, then k + 1: decode-order interleave
        for _n, _cl, pics, sid in streams:
            sids.append(sid); fts.append(pics[k][0]); data.append(pics[k][1])
    if gpu_parse:
        gpu_ctx.submit_many_device(sids, fts, data)
    else:
        gpu_ctx.submit_many(sids, fts, data, 8)
    gpu_ctx.flush()
    st = gpu_ctx.stats()
    nq = 2 if os.environ.get("HVQM4_AMD_QUEUES", "") == "2" else 1
    assert st.pictures == 8 * 16 and st.launches <= 8 * nq      # dependency levels (per launch queue), shared by the eight clips
    assert not st.flags_or & 0x28
    for name, cl, pics, sid in streams:
        want = bridge.oracle_decode(cl.data, cl.n_pictures)
        for k in range(16):
            got = gpu_ctx.read_picture(sid, k)
            assert np.array_equal(got, want[k]), (name, k)
            assert hashlib.sha256(got.tobytes()).hexdigest() == manifest[name]["picture_sha256"][k], (name, k, "reference hash")
        gpu_ctx.close_stream(sid)


def test_c5_share_128_streams_streaming_flushes_six_slots(gpu_ctx):
    import bench
    from hvqm4_amd._lib import HVQ_E_STATE, HvqError
    from hvqm4_amd.synth import SynthConfig
    from oracle import bridge
    distinct, nstreams, nslots, gop = 8, 128, 6, bench.GOP16
    cls = bench.gen_clips([SynthConfig(width=640, height=480, version="1.5", gop=gop, seed=1000 + i) for i in range(distinct)],
                          min(8, os.cpu_count() or 1))
    pics = [_pics(cl) for cl in cls]
    sids = [gpu_ctx.open_stream(640, 480, 2, 2, True, nslots) for _ in range(nstreams)]

    def batch(k0, k1):
        s, f, d = [], [], []
        for k in range(k0, k1):
            for j, sid in enumerate(sids):
                ft, p = pics[j % distinct][k]
                s.append(sid); f.append(ft); d.append(p)
        return s, f, d

    # four batches of four pictures per stream, pipelined: batch n + 1 is copied and uploaded while batch n is parsed
    gpu_ctx.submit_many_device(*batch(0, 4))
    gpu_ctx.flush_begin()
    for b in range(1, 4):
        gpu_ctx.submit_many_device(*batch(4 * b, 4 * b + 4))
        gpu_ctx.flush_end()
        gpu_ctx.flush_begin()
    gpu_ctx.flush_end()
    gpu_ctx.sync()
    want = [bridge.oracle_decode(cl.data, cl.n_pictures) for cl in cls]
    checked = 0
    for j in range(16):                                         # two replicas of every distinct clip
        for k in range(len(gop)):
            try:
                got = gpu_ctx.read_picture(sids[j], k)
            except HvqError as e:
                assert e.code == HVQ_E_STATE                    # slot reused by a later picture
                continue
            assert np.array_equal(got, want[j % distinct][k]), (j, k)
            checked += 1
    assert checked == 16 * nslots
    last = len(gop) - 1
    for j, sid in enumerate(sids):                              # all 128 streams: newest picture equal across replicas
        assert np.array_equal(gpu_ctx.read_picture(sid, last), want[j % distinct][last]), j
    for sid in sids:
        gpu_ctx.close_stream(sid)
