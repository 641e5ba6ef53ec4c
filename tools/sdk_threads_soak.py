"""GPU box: the seven SDK entry points from several host threads at once, each with its own SeqObj (own device context, own pool of parse
threads), many passes over clips of different shapes, every picture against the CPU oracle.  usage: python tools/sdk_threads_soak.py [threads] [passes]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from hvqm4_amd import sdk  # noqa: E402
from hvqm4_amd.container import video_pictures  # noqa: E402
from hvqm4_amd.synth import SynthConfig, make_clip  # noqa: E402
from oracle import bridge  # noqa: E402

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfgs = [SynthConfig(width=640, height=480, version="1.5", gop="IPBBPBBPBBPBBPBB", seed=1000),
        SynthConfig(width=320, height=240, version="1.3", gop="IPBBPBB", seed=22),
        SynthConfig(width=296, height=160, version="1.5", gop="IPBB", seed=9, runoff_prob=0.5, sampling="444"),
        SynthConfig(width=128, height=96, version="1.5", gop="IPBBPBB", seed=17, preset="natural"),
        SynthConfig(width=720, height=576, version="1.5", gop="IPB", seed=5, sampling="422")]
clips = [make_clip(c) for c in cfgs[:max(1, min(len(cfgs), nthreads))]]
wants = [bridge.oracle_decode(c.data, c.n_pictures) for c in clips]
bad = [0] * nthreads
done = [0] * nthreads


def run(k):
    clip, want = clips[k % len(clips)], wants[k % len(clips)]
    seq = [(ft, bytes(p)) for ft, _d, p in video_pictures(clip.data)]
    pl = sdk.Player(clip.width, clip.height, clip.samp_h, clip.samp_v, clip.version == "1.5")
    for _ in range(passes):
        for i, (ft, pic) in enumerate(seq):
            got = pl.decode(ft, pic)
            if not np.array_equal(got, want[i]):
                bad[k] += 1
            done[k] += 1
    pl.close()


t0 = time.time()
th = [threading.Thread(target=run, args=(k,)) for k in range(nthreads)]
for t in th: t.start()
for t in th: t.join()
print(f"sdk soak: {nthreads} threads x {passes} passes, {sum(done)} pictures, {sum(bad)} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if sum(bad) else 0)
