"""`.h4m` container demux (reference: load_header h4m_audio_decode.c:2175-2247,
block/frame loop h4m:2427-2537).  Host-side, not on the hot path: it only slices frame records."""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Iterator, List, Tuple

MAGIC_13 = b"HVQM4 1.3".ljust(16, b"\0")
MAGIC_15 = b"HVQM4 1.5".ljust(16, b"\0")


@dataclass
class H4MHeader:
    version: str
    header_size: int
    body_size: int
    blocks: int
    video_frames: int
    audio_frames: int
    usec_per_frame: int
    max_frame_size: int
    width: int
    height: int
    h_samp: int
    v_samp: int
    video_mode: int

    @property
    def is15(self) -> bool:
        return self.version == "1.5"

    @property
    def pic_bytes(self) -> int:
        ss = self.h_samp * self.v_samp
        return self.width * self.height * (ss + 2) // ss        # h4m:2343-2345


def parse_header(data: bytes) -> H4MHeader:
    """Same acceptance rules as load_header (h4m:2175-2247): magic, header size 0x44, non-zero
    block count, zero words at 0x2C / 0x3B, video mode 0 or 0x12."""
    if len(data) < 0x44:
        raise ValueError("truncated HVQM4 header")
    if data[:16] == MAGIC_13:
        ver = "1.3"
    elif data[:16] == MAGIC_15:
        ver = "1.5"
    else:
        raise ValueError("does not appear to be a HVQM4 file")
    (hsz, body, blocks, vframes, aframes, usec, maxf, unk2c, _afs) = struct.unpack(">9I", data[0x10:0x34])
    w, h, hs, vs, mode, unk3b = struct.unpack(">HHBBBB", data[0x34:0x3C])
    if hsz != 0x44:
        raise ValueError(f"expected header size 0x44, got {hsz:#x}")
    if blocks == 0:
        raise ValueError("zero blocks")
    if unk2c != 0 or unk3b != 0 or mode not in (0, 0x12):
        raise ValueError("unexpected header field")
    return H4MHeader(ver, hsz, body, blocks, vframes, aframes, usec, maxf, w, h, hs, vs, mode)


def video_pictures(data: bytes) -> Iterator[Tuple[int, int, bytes]]:
    """Yield (frame_type, disp_id, picture_data) per video record in decode order.  picture_data
    starts after the 4-byte disp_id (what HVQM4Decode*pic receive, h4m:2100)."""
    hdr = parse_header(data)
    pos = 0x44
    total_v = 0
    for _ in range(hdr.blocks):
        _prev, bsize, vcount, acount, marker = struct.unpack(">5I", data[pos:pos + 20])
        if marker != 0x01000000:
            raise ValueError(f"bad block marker at {pos + 16:#x}")
        pos += 20
        start = pos
        v = a = 0
        while v < vcount or a < acount:
            id1, id2, size = struct.unpack(">HHI", data[pos:pos + 8])
            pos += 8
            if id1 == 1:
                if id2 not in (0x10, 0x20, 0x30):
                    raise ValueError(f"unknown video frame type {id2:#x}")
                disp = struct.unpack(">I", data[pos:pos + 4])[0]
                yield id2, disp, data[pos + 4:pos + size]
                v += 1
            elif id1 == 0:
                a += 1
            else:
                raise ValueError(f"unexpected frame id {id1:#06x} {id2:#06x} at {pos - 8:#x}")
            pos += size
        if pos != start + bsize:
            raise ValueError("block size mismatch")
        total_v += v
    if total_v != hdr.video_frames:
        raise ValueError("total frame count mismatch")


def display_order(data: bytes):
    """Decode-order indices of the video pictures sorted for display.  The reference names its output
    `gop_start + disp_id` (h4m:2085, 2121-2122): gop_start = pictures before the GOP, disp_id from the record."""
    hdr = parse_header(data)
    keys = []
    pos = 0x44
    idx = 0
    for _ in range(hdr.blocks):
        _prev, bsize, vcount, acount, _m = struct.unpack(">5I", data[pos:pos + 20])
        pos += 20
        gop_start = idx
        v = a = 0
        while v < vcount or a < acount:
            id1, _id2, size = struct.unpack(">HHI", data[pos:pos + 8])
            pos += 8
            if id1 == 1:
                keys.append((gop_start + struct.unpack(">I", data[pos:pos + 4])[0], idx))
                idx += 1
                v += 1
            else:
                a += 1
            pos += size
    return [i for _k, i in sorted(keys)]
