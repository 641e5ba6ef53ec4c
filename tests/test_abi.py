"""CPU: the drop-in boundary.  The C-ABI library loads, exports every symbol the public headers declare,
struct layouts match the reference's x86-64 build, and without a GPU the pixel path fails loudly."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANIFEST = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b((?:HVQM4|hvq_)\w+)\s*\(", text))


def test_every_declared_symbol_is_exported():
    from hvqm4_amd._lib import SYMBOLS, lib
    l = lib()
    names = declared_functions("hvqm4.h") | declared_functions("hvqm4_amd.h")
    assert {"HVQM4InitDecoder", "HVQM4InitSeqObj", "HVQM4BuffSize", "HVQM4SetBuffer", "HVQM4DecodeIpic",
            "HVQM4DecodePpic", "HVQM4DecodeBpic"} <= names            # symbols.inc:2-8
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/ but not exported"
        assert n in SYMBOLS, f"{n} has no ctypes prototype"


def test_struct_layout_matches_reference_x86_64():
    from hvqm4_amd._lib import SeqObj, VideoInfo, VideoState
    lay = MANIFEST["layout_x86_64"]
    assert C.sizeof(VideoState) == lay["sizeof_VideoState"] == 28120
    assert VideoState.padding.offset == lay["offsetof_padding"] == 28097
    assert C.sizeof(SeqObj) == lay["sizeof_SeqObj"]
    assert C.sizeof(VideoInfo) == lay["sizeof_VideoInfo"]


def test_buffsize_equals_reference():
    from hvqm4_amd import sdk
    from hvqm4_amd._lib import SeqObj, VideoInfo
    for key, want in MANIFEST["buffsize"].items():
        w, h = map(int, key.split("x"))
        s = SeqObj()
        sdk.HVQM4InitSeqObj(s, VideoInfo(w, h, 2, 2, 0))
        assert (s.width, s.height, s.h_samp, s.v_samp) == (w, h, 2, 2)
        assert sdk.HVQM4BuffSize(s) == want


def test_no_gpu_is_a_loud_error_not_a_fallback():
    """on a box without a HIP device every pixel-producing entry point must fail (no CPU path exists)"""
    from hvqm4_amd import batch, sdk
    from hvqm4_amd._lib import HVQ_E_NOGPU, HvqError, lib
    n = C.c_int(0)
    try:
        hip = C.CDLL("libamdhip64.so")
        have_gpu = hip.hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0
    except OSError:
        have_gpu = False
    if have_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(HvqError) as e:
        batch.Context(0)
    assert e.value.code == HVQ_E_NOGPU
    with pytest.raises(HvqError):
        sdk.HVQM4InitDecoder()
    pl_present = np.full(16 * 16 * 3 // 2, 0xAB, dtype=np.uint8)
    from hvqm4_amd._lib import SeqObj, VideoInfo
    s = SeqObj()
    sdk.HVQM4InitSeqObj(s, VideoInfo(16, 16, 2, 2, 0))
    work = np.zeros(sdk.HVQM4BuffSize(s), dtype=np.uint8)
    sdk.HVQM4SetBuffer(s, work)
    from tests import clips
    clip = clips.get(clips.SMALL[0])
    with pytest.raises(HvqError):
        sdk.HVQM4DecodeIpic(s, clip.pictures[0] + b"\0" * 8, pl_present)
    assert (pl_present == 0xAB).all(), "present must be left untouched on failure"
    sdk.HVQM4ReleaseBuffer(s)


def test_product_does_not_reference_the_oracle():
    """hvqm4_amd/ must not import, link or call anything under oracle/"""
    pkg = os.path.join(ROOT, "hvqm4_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "hvqo_" not in text and "hvqd_recon" not in text and "libhvqoracle" not in text, f
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_ring_size_is_host_computable_and_bounded():
    """every reference read of the kernels is ring base + 32-bit offset: hvq_stream_open refuses rings of 4 GiB and more
    (advisor finding of round 3); the size is computable without a GPU"""
    from hvqm4_amd._lib import lib
    l = lib()
    slot = (640 * 480 * 3 // 2 + 64 + 255) // 256 * 256
    assert l.hvq_stream_ring_bytes(640, 480, 2, 2, 6) == 7 * slot
    big = (8192 * 8192 * 3 + 64 + 255) // 256 * 256
    assert l.hvq_stream_ring_bytes(8192, 8192, 1, 1, 20) == 21 * big < 1 << 32
    assert l.hvq_stream_ring_bytes(8192, 8192, 1, 1, 21) == 22 * big >= 1 << 32          # hvq_stream_open: HVQ_E_OVERFLOW
    assert l.hvq_stream_ring_bytes(8192, 8192, 2, 2, 42) >= 1 << 32
    assert l.hvq_stream_ring_bytes(100, 100, 2, 2, 3) == 0                                # refused geometry
