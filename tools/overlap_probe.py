"""GPU box: would the entropy-parse kernel and the reconstruction kernel gain from running side by side?  Two contexts of one process
stream the same 128-stream workload from two threads (each context has its own HIP streams); if the two together finish their batches
sooner than one context does twice as many, the kernels overlap usefully and a two-deep pipeline inside ONE context (parse k + 1
beside reconstruction k) is worth building.  usage: python tools/overlap_probe.py [batches]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


NEXT = os.environ.get("PROBE_FLUSH_NEXT", "0") == "1"      # the loop's step: hvq_flush_next instead of flush_end + flush_begin


def stream_loop(ctx, sids, fts, raw, n, out):
    ctx.submit_many_device(sids, fts, raw, defer=True)
    ctx.flush_begin()
    t = []
    for _ in range(n):
        ctx.submit_many_device(sids, fts, raw, defer=True)
        if NEXT:
            ctx.flush_next()
        else:
            ctx.flush_end()
            ctx.flush_begin()
        t.append(time.perf_counter())
    ctx.flush_end()
    ctx.sync()
    out.append(t)


if __name__ == "__main__":
    from hvqm4_amd import batch
    from hvqm4_amd.container import video_pictures
    from hvqm4_amd.synth import SynthConfig
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2                 # contexts running side by side
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 128                # streams per context
    cfgs = [SynthConfig(width=640, height=480, version="1.5", gop=bench.GOP16, seed=1000 + i, preset="dense", mv_res_bits=(0, 1, 2)) for i in range(8)]
    clips = bench.gen_clips(cfgs, 1, "/tmp/hvq_clip_cache")
    pics = [list(video_pictures(c.data)) for c in clips]
    ctxs = []
    for _ in range(nctx):
        ctx = batch.Context(0)
        sids = [ctx.open_stream(640, 480, 2, 2, True, 6) for _ in range(per)]
        a_s, a_t, a_p = [], [], []
        for k in range(16):
            for s, sid in enumerate(sids):
                ft, _d, pic = pics[s % 8][k]
                a_s.append(sid); a_t.append(ft); a_p.append(bytes(pic))
        ctxs.append((ctx, a_s, a_t, a_p))
    for ctx, a_s, a_t, a_p in ctxs:                        # warm-up: arenas, buffers
        o = []; stream_loop(ctx, a_s, a_t, a_p, 3, o)
    o = []
    t0 = time.perf_counter(); stream_loop(*ctxs[0], nb, o); t1 = time.perf_counter() - t0
    px = per * 16 * 640 * 480
    print("one context of %d streams: %d batches in %.1f ms = %.3f ms per batch = %.1f Gpixel/s" % (per, nb + 1, t1 * 1e3, t1 * 1e3 / (nb + 1), px * (nb + 1) / t1 / 1e9))
    outs = [[] for _ in range(nctx)]
    if os.environ.get("PROBE_ONE_THREAD", "0") == "1":      # ONE thread drives all contexts in turn (submit / flush_next of A, then of B, ...)
        t0 = time.perf_counter()
        for c, a_s, a_t, a_p in ctxs:
            c.submit_many_device(a_s, a_t, a_p, defer=True); c.flush_begin()
        for _ in range(nb):
            for c, a_s, a_t, a_p in ctxs:
                c.submit_many_device(a_s, a_t, a_p, defer=True); c.flush_next()
        for c, *_ in ctxs:
            c.flush_end()
        for c, *_ in ctxs:
            c.sync()
        t2 = time.perf_counter() - t0
    else:
        th = [threading.Thread(target=stream_loop, args=(*ctxs[i], nb, outs[i])) for i in range(nctx)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        t2 = time.perf_counter() - t0
    print("%d contexts of %d streams side by side: %d x %d batches in %.1f ms = %.1f Gpixel/s (one context alone: %.1f)" % (
        nctx, per, nctx, nb + 1, t2 * 1e3, nctx * px * (nb + 1) / t2 / 1e9, px * (nb + 1) / t1 / 1e9))
    for ctx, *_ in ctxs:
        ctx.close()
