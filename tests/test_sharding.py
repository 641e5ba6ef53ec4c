"""CPU, 2 ranks over gloo: the one-clip-per-GPU sharding and the end-of-run reduction used by bench.py
(no collective on the data path; ranks only meet at the barrier and the final reduce)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, time, json
sys.path.insert(0, os.environ["HVQ_ROOT"])
import numpy as np
from hvqm4_amd.distrib import Group, shard, picture_crc, checksum_of_checksums
from hvqm4_amd.synth import SynthConfig, make_clip
from oracle import bridge
g = Group(backend="gloo")
n_clips = 5
mine = shard(n_clips, g.rank, g.world)
crcs, px = [], 0
g.barrier()
t0 = time.perf_counter()
for i in mine:
    clip = make_clip(SynthConfig(width=32, height=32, gop="IPB", seed=100 + i))
    pics = bridge.oracle_decode(clip.data, clip.n_pictures)      # CPU stand-in for the per-rank GPU decode
    crcs += [picture_crc(p) for p in pics]
    px += clip.n_pictures * 32 * 32
g.barrier()
wall = g.max(time.perf_counter() - t0)
total_px = g.sum(px)
digest = g.sum64(checksum_of_checksums(crcs))
if g.rank == 0:
    g.emit(json.dumps({"world": g.world, "mine": mine, "total_px": total_px, "digest": digest, "wall": wall}))
g.close()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HVQ_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    return json.loads(outs[0][0].strip().splitlines()[-1])


def test_two_ranks_cover_all_clips_once_and_agree_with_one_rank():
    from hvqm4_amd.distrib import shard
    assert sorted(shard(5, 0, 2) + shard(5, 1, 2)) == [0, 1, 2, 3, 4]
    one = _run(1)
    two = _run(2)
    assert two["world"] == 2 and two["mine"] == [0, 2, 4]
    assert two["total_px"] == one["total_px"] == 5 * 3 * 32 * 32
    assert two["digest"] == one["digest"], "the same pictures must come out whatever the sharding"
