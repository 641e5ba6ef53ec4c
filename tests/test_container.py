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
