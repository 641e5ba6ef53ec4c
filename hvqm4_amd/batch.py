"""Batched, device-resident decode path (include/hvqm4_amd.h) -- thin ctypes layer."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional

import numpy as np

from ._lib import HvqStats, check, lib


class Context:
    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib().hvq_context_create(device, C.byref(self._h)))

    def set_launch_queues(self, n: int) -> None:
        """1 or 2 launch queues for this context's batches (hvq_context_set_launch_queues); 1 when two contexts share the GPU"""
        check(lib().hvq_context_set_launch_queues(self._h, n))

    def open_stream(self, width: int, height: int, h_samp: int = 2, v_samp: int = 2, is15: bool = True,
                    nslots: int = 4) -> int:
        return check(lib().hvq_stream_open(self._h, width, height, h_samp, v_samp, int(is15), nslots))

    def close_stream(self, sid: int) -> None:
        check(lib().hvq_stream_close(self._h, sid))

    def set_parse_threads(self, sid: int, threads: int) -> int:
        """host threads that share the parse of ONE picture of this stream (hvq_stream_set_parse_threads); returns the count in effect"""
        return check(lib().hvq_stream_set_parse_threads(self._h, sid, threads))

    def submit(self, sid: int, frame_type: int, picture: bytes) -> int:
        return check(lib().hvq_stream_submit(self._h, sid, frame_type, picture, len(picture)))

    def submit_many(self, sids, frame_types, pictures, threads: int = 1):
        """parse `pictures` on a host thread pool (per-stream order kept) and queue them in list order"""
        n = len(pictures)
        a_s = (C.c_int * n)(*sids)
        a_t = (C.c_int * n)(*frame_types)
        a_p = (C.c_char_p * n)(*pictures)
        a_l = (C.c_size_t * n)(*[len(p) for p in pictures])
        a_o = (C.c_int * n)()
        check(lib().hvq_submit_many(self._h, n, a_s, a_t, a_p, a_l, threads, a_o))
        return list(a_o)

    def submit_many_device(self, sids, frame_types, pictures, defer: bool = False):
        """queue raw bitstreams; flush() parses them on the GPU (no host entropy parse at all).  defer=True
        (hvq_submit_many_device_async): the copy into the pinned arena runs on a worker thread of the library and the call returns at
        once; the pictures are kept alive here until the next flush_begin / sync joins the worker"""
        n = len(pictures)
        a_s = (C.c_int * n)(*sids)
        a_t = (C.c_int * n)(*frame_types)
        a_p = (C.c_char_p * n)(*pictures)
        a_l = (C.c_size_t * n)(*[len(p) for p in pictures])
        a_o = (C.c_int * n)()
        fn = lib().hvq_submit_many_device_async if defer else lib().hvq_submit_many_device
        check(fn(self._h, n, a_s, a_t, a_p, a_l, a_o))
        if defer:
            self._submit_keep = (a_p, pictures)
        return list(a_o)

    def arena_reserve(self, nbytes: int):
        """hvq_arena_reserve: a writable uint8 view of `nbytes` of the pinned arena the library uploads from (zero-copy submit);
        valid until submit_many_arena / the next flush_begin"""
        import numpy as np
        ptr = C.c_void_p()
        check(lib().hvq_arena_reserve(self._h, nbytes, C.byref(ptr)))
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr.value))

    @staticmethod
    def arena_stride(length: int) -> int:
        return int(lib().hvq_arena_stride(length))

    def submit_many_arena(self, sids, frame_types, offsets, lengths):
        """queue pictures the caller wrote into the reservation itself (picture i at offsets[i], lengths[i] bytes): no copy"""
        n = len(offsets)
        a_s = (C.c_int * n)(*sids)
        a_t = (C.c_int * n)(*frame_types)
        a_f = (C.c_size_t * n)(*offsets)
        a_l = (C.c_size_t * n)(*lengths)
        a_o = (C.c_int * n)()
        check(lib().hvq_submit_many_arena(self._h, n, a_s, a_t, a_f, a_l, a_o))
        return list(a_o)

    def flush(self) -> None:
        check(lib().hvq_flush(self._h))

    def flush_begin(self) -> None:
        """first half of a flush (uploads + entropy-parse kernel queued); the next batch may be submitted before flush_end"""
        check(lib().hvq_flush_begin(self._h))

    def flush_end(self) -> None:
        check(lib().hvq_flush_end(self._h))

    def flush_next(self) -> None:
        """streaming step: end the batch in flight and begin the queued one, the queued batch's parse kernel launched first
        (hvq_flush_next); equals flush_end() + flush_begin() in effect"""
        check(lib().hvq_flush_next(self._h))

    def sync(self) -> None:
        check(lib().hvq_sync(self._h))

    def replay(self, reps: int) -> float:
        ms = C.c_float(0)
        check(lib().hvq_replay(self._h, reps, C.byref(ms)))
        return float(ms.value)

    def replay_stage(self, reps: int, what: int = 1) -> float:
        """hvq_replay_stage: what = 1 queue build + reconstruction per repetition, 2 queue build only, 0 = replay"""
        ms = C.c_float(0)
        check(lib().hvq_replay_stage(self._h, reps, what, C.byref(ms)))
        return float(ms.value)

    def pic_bytes(self, sid: int) -> int:
        return int(lib().hvq_stream_pic_bytes(self._h, sid))

    def read_picture(self, sid: int, ordinal: int) -> np.ndarray:
        out = np.empty(self.pic_bytes(sid), dtype=np.uint8)
        check(lib().hvq_read_picture(self._h, sid, ordinal, out.ctypes.data, out.nbytes))
        return out

    def read_picture_rgb(self, sid: int, ordinal: int, width: int, height: int) -> np.ndarray:
        """RGB24 (h, w, 3) of a resident picture: the reference player's dumpRGB on the GPU"""
        out = np.empty(width * height * 3, dtype=np.uint8)
        check(lib().hvq_read_picture_rgb(self._h, sid, ordinal, out.ctypes.data, out.nbytes))
        return out.reshape(height, width, 3)

    def convert_yuv420_rgb(self, yuv: np.ndarray, width: int, height: int) -> np.ndarray:
        """the reference player's dumpRGB (h4m:897-926) on a host picture (Y|U|V 4:2:0) -> RGB24 (h, w, 3)"""
        yuv = np.ascontiguousarray(yuv, dtype=np.uint8)
        assert yuv.size == width * height * 3 // 2
        out = np.empty(width * height * 3, dtype=np.uint8)
        check(lib().hvq_convert_yuv420_rgb(self._h, yuv.ctypes.data, width, height, out.ctypes.data))
        return out.reshape(height, width, 3)

    def read_pictures(self, sids, ordinals, out: Optional[np.ndarray] = None) -> np.ndarray:
        """bulk readback (hvq_read_pictures): one synchronisation for all of them -> uint8[n, pic_bytes] (all streams of one size)"""
        n = len(sids)
        nb = self.pic_bytes(sids[0]) if n else 0
        if out is None:
            out = np.empty((n, nb), dtype=np.uint8)
        a_s = (C.c_int * n)(*sids)
        a_o = (C.c_int * n)(*ordinals)
        a_d = (C.c_void_p * n)(*[out.ctypes.data + i * out.strides[0] for i in range(n)])
        check(lib().hvq_read_pictures(self._h, n, a_s, a_o, a_d))
        return out

    def pinned_array(self, shape) -> np.ndarray:
        """uint8 array in pinned host memory (hvq_pinned_alloc).  The memory lives as long as the array or any view of it: it is
        freed when the last of them is collected, not when the context closes."""
        nbytes = int(np.prod(shape))
        p = lib().hvq_pinned_alloc(nbytes)
        if not p:
            raise MemoryError("hvq_pinned_alloc failed: " + lib().hvq_last_error_string().decode(errors="replace"))
        buf = (C.c_uint8 * nbytes).from_address(p)
        weakref.finalize(buf, lib().hvq_pinned_free, p)         # numpy views keep `buf` alive through their base
        return np.frombuffer(buf, dtype=np.uint8).reshape(shape)

    def picture_device_ptr(self, sid: int, ordinal: int) -> int:
        ptr = C.c_void_p()
        check(lib().hvq_picture_device_ptr(self._h, sid, ordinal, C.byref(ptr)))
        return int(ptr.value or 0)

    def rgb_bench(self, reps: int):
        """-> (gpu_ms, bytes_per_rep, pictures): batched display epilogue over the newest picture of every stream"""
        ms, by, n = C.c_float(0), C.c_uint64(0), C.c_uint32(0)
        check(lib().hvq_rgb_bench(self._h, reps, C.byref(ms), C.byref(by), C.byref(n)))
        return float(ms.value), int(by.value), int(n.value)

    def h2d_probe(self, nbytes: int, reps: int = 8) -> float:
        """GB/s of pinned-host -> device copies of `nbytes` on the copy stream (the PCIe bound of streaming from host memory)"""
        g = C.c_double(0)
        check(lib().hvq_h2d_probe(self._h, int(nbytes), int(reps), C.byref(g)))
        return float(g.value)

    def stats(self) -> HvqStats:
        st = HvqStats()
        check(lib().hvq_get_stats(self._h, C.byref(st)))
        return st

    def close(self):
        if self._h:
            lib().hvq_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_clip(ctx: Context, data: bytes, nslots: Optional[int] = None, gpu_parse: bool = False,
                flush_every: Optional[int] = None) -> np.ndarray:
    """Decode a whole .h4m through the batched path; returns uint8[n_pictures, pic_bytes] (decode order).
    gpu_parse: entropy-parse on the GPU (hvq_submit_many_device); flush_every: flush after that many pictures."""
    from .container import parse_header, video_pictures
    hdr = parse_header(data)
    pics = list(video_pictures(data))
    sid = ctx.open_stream(hdr.width, hdr.height, hdr.h_samp, hdr.v_samp, hdr.is15,
                          nslots if nslots is not None else len(pics) + 3)
    step = flush_every or len(pics)
    for at in range(0, len(pics), step):
        part = pics[at:at + step]
        if gpu_parse:
            ctx.submit_many_device([sid] * len(part), [ft for ft, _d, _p in part], [bytes(p) for _f, _d, p in part])
        else:
            for ft, _disp, pic in part:
                ctx.submit(sid, ft, pic)
        ctx.flush()
    out = np.stack([ctx.read_picture(sid, i) for i in range(len(pics))])
    ctx.close_stream(sid)
    return out


class PairedContexts:
    """The streaming subset of Context over TWO contexts of one GPU, driven by one thread: streams are dealt to the contexts in turn, a
    submit splits its pictures by their stream's context, flush_next() advances context A, then context B.  While the call waits for
    one context's parse results the other context's kernels run, and the two half-size parse kernels stay out of phase (one context's
    pictures in the scalar chains while the other's are in the all-thread passes or being reconstructed): 169-173 Gpixel/s against
    158-160 for the same 128 dense streams in one context (profiles/r05_flush_next.txt 6; INTEGRATION.md "Two contexts per GPU" is the
    same recipe in C).  Stream ids are the pair's own."""

    def __init__(self, device: int = 0):
        self._ctx = [Context(device), Context(device)]
        for c in self._ctx:
            c.set_launch_queues(1)                         # the two contexts are each other's second queue
        self._where = []                                   # pair stream id -> (context index, stream id inside it)

    def open_stream(self, width: int, height: int, h_samp: int = 2, v_samp: int = 2, is15: bool = True, nslots: int = 4) -> int:
        k = len(self._where) & 1
        self._where.append((k, self._ctx[k].open_stream(width, height, h_samp, v_samp, is15, nslots)))
        return len(self._where) - 1

    def submit_many_device(self, sids, frame_types, pictures, defer: bool = False):
        """as Context.submit_many_device; the returned ordinals are per stream, as there"""
        part = ([], [], [], []), ([], [], [], [])
        for i, sid in enumerate(sids):
            k, inner = self._where[sid]
            part[k][0].append(inner); part[k][1].append(frame_types[i]); part[k][2].append(pictures[i]); part[k][3].append(i)
        ords = [0] * len(sids)
        for k in (0, 1):
            if part[k][0]:
                for i, o in zip(part[k][3], self._ctx[k].submit_many_device(part[k][0], part[k][1], part[k][2], defer=defer)):
                    ords[i] = o
        return ords

    def flush_begin(self) -> None:
        for c in self._ctx:
            c.flush_begin()

    def flush_next(self) -> None:
        for c in self._ctx:
            c.flush_next()

    def flush_end(self) -> None:
        for c in self._ctx:
            c.flush_end()

    def flush(self) -> None:
        self.flush_begin()
        self.flush_end()

    def sync(self) -> None:
        for c in self._ctx:
            c.sync()

    def read_picture(self, sid: int, ordinal: int) -> np.ndarray:
        k, inner = self._where[sid]
        return self._ctx[k].read_picture(inner, ordinal)

    def stats(self):
        return [c.stats() for c in self._ctx]

    def close(self) -> None:
        for c in self._ctx:
            c.close()

