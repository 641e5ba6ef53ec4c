"""GPU box: soak test of the streaming path -- many batches through hvq_flush_begin / submit next / hvq_flush_end with the
GPU parser, slots recycled (small ring), pictures checked against the CPU oracle every few batches, device memory
watched for growth.  Test infrastructure (uses oracle/).  usage: python tools/soak_streaming.py [batches] [streams] [next]
With `next`: the loop's step is hvq_flush_next (two batches in flight), rings hold two batches, and every 10th batch ALL 16 pictures of
five streams are read back right after the call -- while the next batch's parse kernel runs -- and compared."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from hvqm4_amd import batch  # noqa: E402
from hvqm4_amd.container import video_pictures  # noqa: E402
from hvqm4_amd.synth import SynthConfig, make_clip  # noqa: E402
from oracle import bridge  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    use_next = len(sys.argv) > 3 and sys.argv[3] == "next"
    import torch
    clips = [make_clip(SynthConfig(width=320, height=240, gop="IPBBPBBPBBPBBPBB", seed=500 + i, preset=p))
             for i, p in enumerate(["dense", "natural", "realistic", "flat"])]
    pics = [list(video_pictures(c.data)) for c in clips]
    want = [bridge.oracle_decode(c.data, c.n_pictures) for c in clips]
    ctx = batch.Context(0)
    sids = [ctx.open_stream(320, 240, 2, 2, True, 36 if use_next else 6) for _ in range(ns)]
    a_s, a_t, a_p = [], [], []
    for k in range(16):
        for s, sid in enumerate(sids):
            ft, _d, pic = pics[s % 4][k]
            a_s.append(sid); a_t.append(ft); a_p.append(bytes(pic))
    free0 = None
    t0 = time.time()
    ctx.submit_many_device(a_s, a_t, a_p, defer=True)
    ctx.flush_begin()
    bad = 0
    for b in range(1, nb if use_next else 0):
        ctx.submit_many_device(a_s, a_t, a_p, defer=True)
        ctx.flush_next()                        # batch b - 1 is launched, batch b in flight
        if b % 10 == 0:
            for s in (0, 1, 2, 3, ns - 1):
                for k in range(16):
                    got = ctx.read_picture(sids[s], 16 * (b - 1) + k)
                    bad += not np.array_equal(got, want[s % 4][k])
        if b % 50 == 0:
            free, _tot = torch.cuda.mem_get_info(0)
            free0 = free0 or free
            print(f"batch {b}: {bad} mismatches, device free {free >> 20} MiB (start {free0 >> 20}), {time.time() - t0:.1f} s", flush=True)
    for b in range(1, 0 if use_next else nb):
        if b % 50 == 0:                         # every 50th batch: finish the batch in flight and check it BEFORE the next
            ctx.flush_end()                     # submit re-assigns its slots
            ctx.sync()
            for s in (0, 1, 2, 3, ns - 1):
                got = ctx.read_picture(sids[s], 16 * b - 1)
                bad += not np.array_equal(got, want[s % 4][15])
            free, _tot = torch.cuda.mem_get_info(0)
            free0 = free0 or free
            print(f"batch {b}: {bad} mismatches, device free {free >> 20} MiB (start {free0 >> 20}), {time.time() - t0:.1f} s", flush=True)
            ctx.submit_many_device(a_s, a_t, a_p, defer=True)
        else:
            ctx.submit_many_device(a_s, a_t, a_p, defer=True)
            ctx.flush_end()
        ctx.flush_begin()
    ctx.flush_end()
    ctx.sync()
    free, _tot = torch.cuda.mem_get_info(0)
    ctx.close()
    grew = (free0 - free) >> 20 if free0 else 0
    print(f"soak done: {nb} batches x {ns} streams x 16 pictures, {bad} mismatches, device memory grew {grew} MiB")
    sys.exit(1 if bad or grew > 64 else 0)


if __name__ == "__main__":
    main()
