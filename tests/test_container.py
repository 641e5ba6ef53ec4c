"""CPU: .h4m demux -- same acceptance rules as the reference's load_header / block loop."""
import struct

import pytest

from hvqm4_amd.container import parse_header, video_pictures
from tests import clips


def test_header_fields_and_picture_slicing():
    clip = clips.get(clips.SMALL[14])        # three GOPs
    h = parse_header(clip.data)
    assert (h.width, h.height, h.h_samp, h.v_samp, h.version) == (64, 48, 2, 2, "1.5")
    assert h.blocks == 3 and h.video_frames == clip.n_pictures and h.pic_bytes == 64 * 48 * 3 // 2
    pics = list(video_pictures(clip.data))
    assert [p[0] for p in pics] == clip.kinds
    assert [p[2] for p in pics] == clip.pictures
    assert sorted(p[1] for p in pics[:4]) == [0, 1, 2, 3]        # display ids cover the GOP


@pytest.mark.parametrize("mutate,msg", [
    (lambda d: b"HVQM4 9.9" + d[9:], "HVQM4"),
    (lambda d: d[:0x10] + struct.pack(">I", 0x40) + d[0x14:], "header size"),
    (lambda d: d[:0x18] + struct.pack(">I", 0) + d[0x1C:], "zero blocks"),
    (lambda d: d[:0x2C] + struct.pack(">I", 1) + d[0x30:], "header field"),
    (lambda d: d[:0x44 + 16] + struct.pack(">I", 0) + d[0x44 + 20:], "block marker"),
    (lambda d: d[:0x1C] + struct.pack(">I", 99) + d[0x20:], "frame count"),
])
def test_malformed_containers_are_rejected(mutate, msg):
    data = mutate(clips.get(clips.SMALL[3]).data)
    with pytest.raises(ValueError, match=msg):
        list(video_pictures(data))


def test_display_order_sorts_by_gop_start_plus_disp_id():
    from hvqm4_amd.container import display_order
    clip = clips.get(clips.SMALL[14])        # three GOPs of I P B B: display I B B P
    order = display_order(clip.data)
    assert sorted(order) == list(range(clip.n_pictures))
    assert order[:4] == [0, 2, 3, 1] and order[4:8] == [4, 6, 7, 5]


def _c_demux(data):
    """the C library's demux (hvq_h4m_*): -> (info, [(type, disp, picture bytes)], final rc)"""
    import ctypes as C
    from hvqm4_amd._lib import HvqH4mInfo, HvqH4mIter, lib
    l = lib()
    info = HvqH4mInfo()
    rc = l.hvq_h4m_header(data, len(data), C.byref(info))
    if rc:
        return info, [], rc
    it = HvqH4mIter()
    l.hvq_h4m_begin(C.byref(it))
    out = []
    ft, disp, pic, ln = C.c_int(0), C.c_uint32(0), C.c_void_p(0), C.c_size_t(0)
    base = C.cast(C.c_char_p(data), C.c_void_p).value
    while True:
        rc = l.hvq_h4m_next(data, len(data), C.byref(it), C.byref(ft), C.byref(disp), C.byref(pic), C.byref(ln))
        if rc != 1:
            return info, out, rc
        off = pic.value - base
        out.append((ft.value, disp.value, data[off:off + ln.value]))


def test_c_demux_equals_python_demux():
    from hvqm4_amd.container import display_order
    for case in (clips.SMALL[3], clips.SMALL[14], clips.SMALL[9]):
        clip = clips.get(case)
        info, recs, rc = _c_demux(clip.data)
        assert rc == 0
        h = parse_header(clip.data)
        assert (info.width, info.height, info.h_samp, info.v_samp, bool(info.is_1_5), info.pic_bytes) == \
               (h.width, h.height, h.h_samp, h.v_samp, h.is15, h.pic_bytes)
        py = list(video_pictures(clip.data))
        assert [(t, p) for t, _d, p in recs] == [(t, p) for t, _d, p in py]
        order = sorted(range(len(recs)), key=lambda i: recs[i][1])
        assert order == display_order(clip.data)


@pytest.mark.parametrize("mutate", [
    lambda d: b"HVQM4 9.9" + d[9:],
    lambda d: d[:0x10] + struct.pack(">I", 0x40) + d[0x14:],
    lambda d: d[:0x18] + struct.pack(">I", 0) + d[0x1C:],
    lambda d: d[:0x2C] + struct.pack(">I", 1) + d[0x30:],
    lambda d: d[:0x44 + 16] + struct.pack(">I", 0) + d[0x44 + 20:],
    lambda d: d[:0x1C] + struct.pack(">I", 99) + d[0x20:],
    lambda d: d[:len(d) - 40],
])
def test_c_demux_rejects_what_the_reference_exits_on(mutate):
    from hvqm4_amd._lib import HVQ_E_CONTAINER
    data = mutate(clips.get(clips.SMALL[3]).data)
    _info, _recs, rc = _c_demux(data)
    assert rc == HVQ_E_CONTAINER
