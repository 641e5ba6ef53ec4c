"""The drop-in boundary from C: examples/h4m_player.c uses the seven SDK entry points with the reference player's
call sequence and buffer rotation.  CPU: it compiles and links against libhvqm4_amd.so and fails loudly without a GPU.
GPU: the pictures it writes for the golden clips have the sha256 of the pictures the REFERENCE decoded
(tests/golden/manifest.json) and equal the oracle's; the FNV-1a it prints per picture matches too."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "native", "_build", "h4m_player")


BIN_BATCH = os.path.join(ROOT, "tests", "native", "_build", "h4m_batch")


def build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    lib = os.path.join(ROOT, "hvqm4_amd")
    for src, out in (("h4m_player.c", BIN), ("h4m_batch.c", BIN_BATCH)):
        subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", src), "-L" + lib, "-lhvqm4_amd",
                        "-Wl,-rpath," + lib, "-o", out], check=True)


def fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for x in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_player_builds_and_refuses_to_run_without_gpu():
    import torch
    build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([BIN, os.path.join(ROOT, "tests", "golden", "i16.h4m")], capture_output=True, text=True)
    assert r.returncode != 0
    assert "GPU-only" in r.stderr


@pytest.mark.gpu
def test_c_player_decodes_the_golden_clips_like_the_reference(tmp_path):
    from oracle import bridge
    build()
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    done = 0
    for name, ent in man["clips"].items():
        if "file" not in ent:                        # larger clips are kept as hashes only (regenerated from seeds)
            continue
        path = os.path.join(ROOT, "tests", "golden", ent["file"])
        if not os.path.exists(path):
            continue
        out = tmp_path / (name + ".yuv")
        r = subprocess.run([BIN, path, str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = [l.split() for l in r.stdout.strip().splitlines()]
        assert [int(l[1], 16) for l in lines] == ent["frame_types"]
        data = open(path, "rb").read()
        want = bridge.oracle_decode(data, len(lines))
        got = np.fromfile(out, dtype=np.uint8).reshape(len(lines), -1)
        assert np.array_equal(got, want), name
        for k, l in enumerate(lines):
            assert int(l[3], 16) == fnv1a(want[k].tobytes()), (name, k)
            assert hashlib.sha256(got[k].tobytes()).hexdigest() == ent["picture_sha256"][k], (name, k)   # the REFERENCE's output
        done += 1
    assert done >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("parser", ["gpu", "host"])
@pytest.mark.parametrize("name", ["gop64x48_15", "twogops64x48", "pselfref64x48_15"])
def test_c_batch_example_drains_every_picture_in_display_order(tmp_path, parser, name):
    """examples/h4m_batch.c: N streams of one clip through the batched C API (GPU or host entropy parse), streamed with
    hvq_flush_begin / hvq_flush_end, every picture read back with hvq_read_pictures (bulk, pinned, beside the next batch's
    parse) and written in DISPLAY order: all of them must be the oracle's pictures, reordered by the container's disp_id"""
    from hvqm4_amd.container import display_order
    from oracle import bridge
    build()
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    ent = man["clips"][name]
    path = os.path.join(ROOT, "tests", "golden", ent["file"])
    out = tmp_path / "all.yuv"
    r = subprocess.run([BIN_BATCH, path, "5", parser, str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    data = open(path, "rb").read()
    npic = len(ent["frame_types"])
    want = bridge.oracle_decode(data, npic)                               # decode order
    order = list(display_order(data))                                       # decode ordinal of every display index (gop_start + disp_id)
    assert sorted(order) == list(range(npic))
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(lines) == npic
    got = np.fromfile(out, dtype=np.uint8).reshape(npic, -1)
    for k, l in enumerate(lines):
        o = int(l[3])
        assert int(l[1]) == k and order[k] == o
        assert np.array_equal(got[k], want[o]), (k, o)
        assert int(l[-1], 16) == fnv1a(want[o].tobytes())
