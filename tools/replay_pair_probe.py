"""GPU box: does the reconstruction STAGE gain from two contexts?  The bench's dense batch (128 streams x 16 pictures) resident in ONE context
against the same streams resident in two contexts of 64 streams each, replayed side by side from two threads (launch streams on hardware queues
of their own).  usage: python tools/replay_pair_probe.py [steps]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def resident(ctx, nstreams, pics):
    sids = [ctx.open_stream(640, 480, 2, 2, True, 6) for _ in range(nstreams)]
    a_s, a_t, a_p = [], [], []
    for k in range(16):
        for s, sid in enumerate(sids):
            ft, _d, pic = pics[s % 8][k]
            a_s.append(sid); a_t.append(ft); a_p.append(bytes(pic))
    ctx.submit_many_device(a_s, a_t, a_p)
    ctx.flush()
    ctx.sync()
    return 640 * 480 * len(a_p)


if __name__ == "__main__":
    from hvqm4_amd import batch
    from hvqm4_amd.container import video_pictures
    from hvqm4_amd.synth import SynthConfig
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cfgs = [SynthConfig(width=640, height=480, version="1.5", gop=bench.GOP16, seed=1000 + i, preset=os.environ.get("PROBE_PRESET", "dense"), mv_res_bits=(0, 1, 2)) for i in range(8)]
    clips = bench.gen_clips(cfgs, 1, "/tmp/hvq_clip_cache")
    pics = [list(video_pictures(c.data)) for c in clips]
    one = batch.Context(0); px1 = resident(one, 128, pics)
    pair = [batch.Context(0), batch.Context(0)]; px2 = [resident(c, 64, pics) for c in pair]
    for c in [one] + pair:
        c.replay_stage(10, 1)
    for rep in range(3):
        ms = one.replay_stage(steps, 1)
        print("one context of 128 streams : %.1f us per step = %.0f Mpixel/s" % (ms * 1e3 / steps, px1 * steps / ms / 1e3))
        out = [0.0, 0.0]
        def run(i):
            out[i] = pair[i].replay_stage(steps, 1)
        th = [threading.Thread(target=run, args=(i,)) for i in (0, 1)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        wall = time.perf_counter() - t0
        print("two contexts of 64 streams : wall %.1f us per step of both = %.0f Mpixel/s (their own event times %.1f / %.1f us per step)" % (
            wall * 1e6 / steps, sum(px2) * steps / wall / 1e6, out[0] * 1e3 / steps, out[1] * 1e3 / steps))
