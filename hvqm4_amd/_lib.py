"""Loader for the product's C-ABI library (hvqm4_amd/libhvqm4_amd.so).

The library is the product: HIP kernels + host parse + runtime.  It is built in-tree by
`make -C hvqm4_amd/csrc` (or `__graft_entry__.build()`); importing this module with the
library missing is an error -- there is no Python or CPU fallback for the pixel path.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HVQM4_AMD_LIB") or os.path.join(HERE, "libhvqm4_amd.so")   # override: ablation builds only

HVQ_OK, HVQ_E_ARG, HVQ_E_OVERFLOW, HVQ_E_GEOMETRY, HVQ_E_NOGPU, HVQ_E_HIP, HVQ_E_STATE, HVQ_E_CONTAINER = 0, -1, -2, -3, -4, -5, -6, -7
HVQ_E_UNSUPPORTED = -8
ERROR_NAMES = {HVQ_E_ARG: "HVQ_E_ARG", HVQ_E_OVERFLOW: "HVQ_E_OVERFLOW", HVQ_E_GEOMETRY: "HVQ_E_GEOMETRY",
               HVQ_E_NOGPU: "HVQ_E_NOGPU", HVQ_E_HIP: "HVQ_E_HIP", HVQ_E_STATE: "HVQ_E_STATE",
               HVQ_E_CONTAINER: "HVQ_E_CONTAINER", HVQ_E_UNSUPPORTED: "HVQ_E_UNSUPPORTED"}

HVQM4_VIDEOSTATE_SIZE = 28120
HVQM4_VIDEOSTATE_PADDING = 28097


class HvqError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {msg}")
        self.code = code


class VideoState(C.Structure):
    _fields_ = [("opaque0", C.c_uint8 * HVQM4_VIDEOSTATE_PADDING),
                ("padding", C.c_uint8 * 3),
                ("opaque1", C.c_uint8 * (HVQM4_VIDEOSTATE_SIZE - HVQM4_VIDEOSTATE_PADDING - 3))]


class SeqObj(C.Structure):                       # h4m_audio_decode.c:516-523
    _fields_ = [("state", C.POINTER(VideoState)), ("width", C.c_uint16), ("height", C.c_uint16),
                ("h_samp", C.c_uint8), ("v_samp", C.c_uint8)]


class VideoInfo(C.Structure):                    # h4m_audio_decode.c:533-540
    _fields_ = [("hres", C.c_uint16), ("vres", C.c_uint16), ("h_samp", C.c_uint8), ("v_samp", C.c_uint8),
                ("video_mode", C.c_uint8)]


class HvqH4mInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("header_size", "body_size", "blocks", "video_frames", "audio_frames",
                                          "usec_per_frame", "max_frame_size", "pic_bytes")] + \
               [("width", C.c_uint16), ("height", C.c_uint16), ("h_samp", C.c_uint8), ("v_samp", C.c_uint8),
                ("video_mode", C.c_uint8), ("is_1_5", C.c_uint8)]


class HvqH4mIter(C.Structure):
    _fields_ = [("pos", C.c_size_t), ("block_end", C.c_size_t)] + \
               [(n, C.c_uint32) for n in ("block", "v_left", "a_left", "video_seen", "gop_start", "in_block")]


class HvqStats(C.Structure):
    _fields_ = [("pictures", C.c_uint64), ("luma_pixels", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("descriptor_bytes", C.c_uint64), ("launches", C.c_uint32), ("workgroups", C.c_uint32),
                ("parse_seconds", C.c_double), ("flags_or", C.c_uint32), ("gpu_parsed", C.c_uint32),
                ("gpu_parse_ms", C.c_double), ("gpu_parse_retried", C.c_uint32), ("dropped", C.c_uint32),
                ("launch_queues", C.c_uint32), ("queue_bytes", C.c_uint64), ("copy_bytes", C.c_uint64), ("copy_seconds", C.c_double)]


# every symbol include/hvqm4.h and include/hvqm4_amd.h declare: (restype, argtypes)
SYMBOLS = {
    "HVQM4InitDecoder": (None, []),
    "HVQM4InitSeqObj": (None, [C.POINTER(SeqObj), C.POINTER(VideoInfo)]),
    "HVQM4BuffSize": (C.c_uint32, [C.POINTER(SeqObj)]),
    "HVQM4SetBuffer": (None, [C.POINTER(SeqObj), C.c_void_p]),
    "HVQM4DecodeIpic": (None, [C.POINTER(SeqObj), C.c_char_p, C.c_void_p]),
    "HVQM4DecodePpic": (None, [C.POINTER(SeqObj), C.c_char_p, C.c_void_p, C.c_void_p]),
    "HVQM4DecodeBpic": (None, [C.POINTER(SeqObj), C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "HVQM4GetLastError": (C.c_int, []),
    "HVQM4GetLastErrorString": (C.c_char_p, []),
    "HVQM4SetVersion15": (None, [C.POINTER(SeqObj), C.c_int]),
    "HVQM4ReleaseBuffer": (None, [C.POINTER(SeqObj)]),
    "HVQM4SetMaxFrameSize": (None, [C.POINTER(SeqObj), C.c_uint32]),
    "hvq_context_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "hvq_context_destroy": (None, [C.c_void_p]),
    "hvq_context_set_launch_queues": (C.c_int, [C.c_void_p, C.c_int]),
    "hvq_stream_open": (C.c_int, [C.c_void_p] + [C.c_int] * 6),
    "hvq_stream_close": (C.c_int, [C.c_void_p, C.c_int]),
    "hvq_stream_ring_bytes": (C.c_uint64, [C.c_int] * 5),
    "hvq_stream_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "hvq_submit_many": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p),
                                  C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_int)]),
    "hvq_submit_many_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p),
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "hvq_submit_many_device_async": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p),
                                               C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "hvq_arena_reserve": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "hvq_arena_stride": (C.c_size_t, [C.c_size_t]),
    "hvq_submit_many_arena": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "hvq_flush": (C.c_int, [C.c_void_p]),
    "hvq_flush_begin": (C.c_int, [C.c_void_p]),
    "hvq_flush_end": (C.c_int, [C.c_void_p]),
    "hvq_flush_next": (C.c_int, [C.c_void_p]),
    "hvq_sync": (C.c_int, [C.c_void_p]),
    "hvq_replay": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "hvq_replay_stage": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "hvq_read_picture": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "hvq_stream_pic_bytes": (C.c_uint32, [C.c_void_p, C.c_int]),
    "hvq_read_picture_rgb": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "hvq_rgb_bench": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "hvq_convert_yuv420_rgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hvq_read_pictures": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "hvq_stream_set_parse_threads": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "hvq_h2d_probe": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "hvq_pinned_alloc": (C.c_void_p, [C.c_size_t]),
    "hvq_pinned_free": (None, [C.c_void_p]),
    "hvq_picture_device_ptr": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "hvq_get_stats": (C.c_int, [C.c_void_p, C.POINTER(HvqStats)]),
    "hvq_debug_table_divisions": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "hvq_last_error_string": (C.c_char_p, []),
    "hvq_h4m_header": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(HvqH4mInfo)]),
    "hvq_h4m_begin": (None, [C.POINTER(HvqH4mIter)]),
    "hvq_h4m_next": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(HvqH4mIter), C.POINTER(C.c_int), C.POINTER(C.c_uint32),
                               C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "hvq_parser_create": (C.c_void_p, [C.c_int] * 5),
    "hvq_parser_destroy": (None, [C.c_void_p]),
    "hvq_parser_blob_bound": (C.c_size_t, [C.c_void_p]),
    "hvq_parser_pic_bytes": (C.c_uint32, [C.c_void_p]),
    "hvq_picture_length": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.POINTER(C.c_size_t)]),
    "hvq_parser_set_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "hvq_parser_last_flags": (C.c_uint32, [C.c_void_p]),
    "hvq_parse_picture": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_size_t)]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C hvqm4_amd/csrc` "
                "(hipcc --offload-arch=gfx950).  The HVQM4 reconstruction path is GPU-only; "
                "there is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if not os.environ.get("HVQM4_AMD_LIB"):
                    raise                      # the product library must export every symbol the headers declare (tests/test_abi.py)
                continue                       # an older build named for an A/B experiment: it simply lacks the newer entry points
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise HvqError(rc, lib().hvq_last_error_string().decode(errors="replace"))
    return rc
