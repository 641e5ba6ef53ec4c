"""Python mirror of the HVQM4 1.5 SDK C API served by the MI355X back end.

Same names, argument meaning and ownership rules as the reference's entry points
(h4m_audio_decode.c:275, 819, 828, 957, 1970, 2018, 2058): the caller owns the work buffer and
the picture buffers, calls are synchronous, `frame` starts after the disp_id word.  Unlike the
reference's void functions, failures raise HvqError (the C ABI reports them through
HVQM4GetLastError).  This is a thin ctypes layer; all work happens in libhvqm4_amd.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from ._lib import HvqError, SeqObj, VideoInfo, VideoState, lib

__all__ = ["SeqObj", "VideoInfo", "HVQM4InitDecoder", "HVQM4InitSeqObj", "HVQM4BuffSize", "HVQM4SetBuffer",
           "HVQM4DecodeIpic", "HVQM4DecodePpic", "HVQM4DecodeBpic", "HVQM4ReleaseBuffer", "Player"]


def _raise_if_error():
    l = lib()
    code = l.HVQM4GetLastError()
    if code:
        raise HvqError(code, l.HVQM4GetLastErrorString().decode(errors="replace"))


def HVQM4InitDecoder() -> None:
    lib().HVQM4InitDecoder()
    _raise_if_error()


def HVQM4InitSeqObj(seqobj: SeqObj, videoinfo: VideoInfo) -> None:
    lib().HVQM4InitSeqObj(C.byref(seqobj), C.byref(videoinfo))


def HVQM4BuffSize(seqobj: SeqObj) -> int:
    return int(lib().HVQM4BuffSize(C.byref(seqobj)))


def HVQM4SetBuffer(seqobj: SeqObj, workbuff) -> None:
    """`workbuff`: a ctypes buffer / numpy uint8 array of HVQM4BuffSize bytes, owned by the caller."""
    lib().HVQM4SetBuffer(C.byref(seqobj), _addr(workbuff))
    _raise_if_error()


def HVQM4ReleaseBuffer(seqobj: SeqObj) -> None:
    lib().HVQM4ReleaseBuffer(C.byref(seqobj))


def _addr(buf) -> int:
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data
    return C.addressof(buf)


def HVQM4DecodeIpic(seqobj: SeqObj, frame: bytes, present) -> None:
    lib().HVQM4DecodeIpic(C.byref(seqobj), frame, _addr(present))
    _raise_if_error()


def HVQM4DecodePpic(seqobj: SeqObj, frame: bytes, present, past) -> None:
    lib().HVQM4DecodePpic(C.byref(seqobj), frame, _addr(present), _addr(past))
    _raise_if_error()


def HVQM4DecodeBpic(seqobj: SeqObj, frame: bytes, present, past, future) -> None:
    lib().HVQM4DecodeBpic(C.byref(seqobj), frame, _addr(present), _addr(past), _addr(future))
    _raise_if_error()


class Player:
    """The reference's `main`/`decode_video` sequence (h4m:2409-2419, 2078-2138) over the SDK calls:
    init, work buffer, three picture buffers, past/present/future rotation."""

    def __init__(self, width: int, height: int, h_samp: int = 2, v_samp: int = 2, is15: bool = True):
        HVQM4InitDecoder()
        self.seqobj = SeqObj()
        info = VideoInfo(width, height, h_samp, v_samp, 0)
        HVQM4InitSeqObj(self.seqobj, info)
        self.work = np.zeros(HVQM4BuffSize(self.seqobj), dtype=np.uint8)
        self.work[28097] = 1 if is15 else 0              # state->padding[0], h4m:2414-2417
        HVQM4SetBuffer(self.seqobj, self.work)
        ss = h_samp * v_samp
        self.pic_bytes = width * height * (ss + 2) // ss
        self.past = np.zeros(self.pic_bytes, dtype=np.uint8)
        self.present = np.zeros(self.pic_bytes, dtype=np.uint8)
        self.future = np.zeros(self.pic_bytes, dtype=np.uint8)

    def decode(self, frame_type: int, picture: bytes) -> np.ndarray:
        frame = picture + b"\0" * 8
        if frame_type != 0x30:
            self.past, self.future = self.future, self.past
        if frame_type == 0x10:
            HVQM4DecodeIpic(self.seqobj, frame, self.present)
        elif frame_type == 0x20:
            HVQM4DecodePpic(self.seqobj, frame, self.present, self.past)
        else:
            HVQM4DecodeBpic(self.seqobj, frame, self.present, self.past, self.future)
        out = self.present.copy()
        if frame_type != 0x30:
            self.present, self.future = self.future, self.present
        return out

    def close(self):
        if self.seqobj is not None:
            HVQM4ReleaseBuffer(self.seqobj)
            self.seqobj = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
