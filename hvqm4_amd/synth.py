"""Synthetic HVQM4 1.3/1.5 (.h4m) stream writer.

The reference ships no clips and no golden vectors, so every
test and benchmark input is produced here.  The writer walks blocks in the
decoder's exact consumption order (h4m_audio_decode.c:1073-1164 for I pictures,
1742-1776 pass 1 and 1919-1967 pass 2 for P/B pictures), records per-stream
operations (Huffman leaf | raw bits | bytes), then builds one prefix tree per tree
group (wiring at h4m:977-999), serialises it pre-order into the carrier stream
(h4m:607-630) and emits the codes.  Wire format: SURVEY.md Appendix B.

Legality (keeps the *reference* in-bounds, Appendix C): every motion-compensated
read and every MC-nest window address stays inside the Y|U|V picture buffer; no
future-referencing macroblocks in P pictures; P/B kind symbols <= 15.
"""
from __future__ import annotations

import heapq
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

I_FRAME, P_FRAME, B_FRAME = 0x10, 0x20, 0x30

# stream indices -------------------------------------------------------------
BN0, BN1, BNR0, BNR1 = 0, 1, 2, 3
DC0, BT0, FX0 = 4, 7, 10          # +plane
RLE0 = 13                         # +plane (I only)
MVH, MVV, MTYPE, MPROC = 16, 17, 18, 19
NSTREAMS = 20

# tree groups: (carrier, members)  -- h4m:977-999
GROUPS = {
    "bn": (BN0, (BN0, BN1)),
    "run": (BNR0, (BNR0, BNR1, RLE0, RLE0 + 1, RLE0 + 2)),
    "dc": (DC0, (DC0, DC0 + 1, DC0 + 2)),
    "bt": (BT0, (BT0, BT0 + 1, BT0 + 2)),
    "mv": (MVH, (MVH, MVV)),
    "mcb": (MTYPE, (MTYPE, MPROC)),
}
STREAM_GROUP = {}
for _g, (_c, _m) in GROUPS.items():
    for _s in _m:
        STREAM_GROUP[_s] = _g

# section order in the picture header offset tables (h4m:1979-1993, 2030-2044)
I_SECTIONS = [BN0, BNR0, BN1, BNR1,
              DC0, BT0, FX0, DC0 + 1, BT0 + 1, FX0 + 1, DC0 + 2, BT0 + 2, FX0 + 2,
              RLE0, RLE0 + 1, RLE0 + 2]
PB_SECTIONS = [BN0, BNR0, BN1, BNR1,
               DC0, BT0, FX0, DC0 + 1, BT0 + 1, FX0 + 1, DC0 + 2, BT0 + 2, FX0 + 2,
               MVH, MVV, MTYPE, MPROC]


@dataclass
class SynthConfig:
    width: int = 64
    height: int = 48
    version: str = "1.5"              # "1.3" | "1.5"
    gop: str = "IPBBPBB"              # decode-order picture kinds of one GOP
    n_gops: int = 1
    repeat_gops: int = 1              # the n_gops generated GOP blocks are written this many times (long clips, cheaply)
    nest_overhang: int = 0            # I pictures: nest window this many columns over the right edge of the block map (<= 2 stays
                                      # inside the reference's bordered array: border entries and the next row's first are read)
    p_future_refs: bool = False       # P pictures may carry type-2 ("future") macroblocks: the reference then reads the
                                      # picture being written (h4m:2058-2061); decoded like the reference since round 3
                                      # (side buffer + raster-order walk, hvq_selfref_kernel)
    seed: int = 0
    preset: str = "dense"             # "dense"  | "realistic" | "flat" | "natural"
    dc_shifts: Sequence[int] = (0, 1, 2)
    unk_shifts: Sequence[int] = (6, 7, 8, 9)
    mv_res_bits: Sequence[int] = (0, 1, 2)
    runoff_prob: float = 0.05         # MVs whose reads run off the row edge (linear addressing, H4)
    weird_kinds: bool = False         # also emit kinds 7, 9..15 (and I-luma bytes > 15)
    max_predi_bases: int = 6
    literal_weight: float = 1.0
    usec_per_frame: int = 33366
    sampling: str = "420"             # "420" (h_samp = v_samp = 2) | "444" (1, 1: four chroma blocks per macroblock) |
                                      # "422" (2, 1: two, one above the other)
    long_escape: int = 0              # I pictures: one luma DC delta written as this many overflow symbols (h4m:654-664 sums
                                      # for as long as the stream says) -- legal by format, far beyond what an encoder emits
    long_escape_pb: int = 0           # P/B pictures: the first intra DC delta written as this many overflow symbols
    predi_big: float = 0.0            # P/B pictures: probability that a scalar of an MC-residual block (h4m:1405-1406) lies beyond 16 bits
    p_proc1: float = -1.0             # >= 0: probability of a run of plain-MC (proc = 1) macroblocks, overriding the preset's
    p_zero: float = -1.0              # >= 0: probability of a zero-kind run start, overriding the preset's (small: nearly every block coded)


class _Ops:
    """Per-stream operation recorder: (code kind, value, nbits)."""
    __slots__ = ("sym", "kinds", "vals", "nbits", "bytes")

    def __init__(self):
        self.kinds: List[int] = []    # 0 = huffman leaf byte, 1 = raw bits
        self.vals: List[int] = []
        self.nbits: List[int] = []
        self.bytes = bytearray()      # fixvl streams only

    def leaf(self, b: int):
        self.kinds.append(0); self.vals.append(b & 0xFF); self.nbits.append(0)

    def raw(self, v: int, n: int):
        if n:
            self.kinds.append(1); self.vals.append(v); self.nbits.append(n)


def _huffman(freq: Dict[int, int]):
    """-> (tree_bits list[(val,n)], codes {leaf: (code,len)}).  Pre-order serialisation."""
    if not freq:
        return [], {}
    heap = [(f, i, ("L", s)) for i, (s, f) in enumerate(sorted(freq.items()))]
    heapq.heapify(heap)
    cnt = len(heap)
    while len(heap) > 1:
        f0, _, a = heapq.heappop(heap)
        f1, _, b = heapq.heappop(heap)
        heapq.heappush(heap, (f0 + f1, cnt, ("N", a, b)))
        cnt += 1
    root = heap[0][2]
    bits: List[Tuple[int, int]] = []
    codes: Dict[int, Tuple[int, int]] = {}

    def walk(node, code, ln):
        stack = [(node, code, ln)]
        while stack:
            nd, c, l = stack.pop()
            if nd[0] == "L":
                bits.append((nd[1], 9))          # '0' + 8-bit leaf byte
                codes[nd[1]] = (c, l)
            else:
                bits.append((1, 1))
                stack.append((nd[2], (c << 1) | 1, l + 1))
                stack.append((nd[1], (c << 1), l + 1))
    walk(root, 0, 0)
    return bits, codes


def _pack_bits(vals: np.ndarray, lens: np.ndarray) -> Tuple[bytes, int]:
    """MSB-first concatenation of (value, length) pairs."""
    total = int(lens.sum())
    if total == 0:
        return b"", 0
    ends = np.cumsum(lens)
    starts = ends - lens
    idx = np.repeat(np.arange(len(lens)), lens)
    pos = np.arange(total) - starts[idx]
    sh = (lens[idx] - 1 - pos).astype(np.int64)
    bits = ((vals[idx].astype(np.int64) >> sh) & 1).astype(np.uint8)
    return np.packbits(bits).tobytes(), total


def _sovf_leaves(t: int) -> List[int]:
    """Leaf bytes (int8 domain) encoding total t via the signed-overflow grammar (h4m:654-664)."""
    out = []
    while t >= 127:
        out.append(127); t -= 127
    while t <= -128:
        out.append(-128 & 0xFF); t += 128
    out.append(t & 0xFF)
    return out


def _uovf_leaves(n: int) -> List[int]:
    out = []
    while n >= 255:
        out.append(255); n -= 255
    out.append(n)
    return out


class _Picture:
    def __init__(self):
        self.ops = [_Ops() for _ in range(NSTREAMS)]
        self.header = b""
        self.kind = I_FRAME

    def assemble(self) -> Tuple[bytes, List[int]]:
        """-> (picture data, expected reader cursor per stream (relative to data start, -1 empty))."""
        sections = I_SECTIONS if self.kind == I_FRAME else PB_SECTIONS
        # 1. per group: alphabet, tree
        payload: Dict[int, bytes] = {}
        consumed: Dict[int, int] = {}
        for g, (carrier, members) in GROUPS.items():
            members = [m for m in members if m in sections]
            if not members:
                continue
            freq: Dict[int, int] = {}
            for m in members:
                o = self.ops[m]
                for k, v in zip(o.kinds, o.vals):
                    if k == 0:
                        freq[v] = freq.get(v, 0) + 1
            tree_bits, codes = _huffman(freq)
            for m in members:
                o = self.ops[m]
                vals: List[int] = []
                lens: List[int] = []
                if m == carrier:
                    for v, n in tree_bits:
                        vals.append(v); lens.append(n)
                for k, v, n in zip(o.kinds, o.vals, o.nbits):
                    if k == 0:
                        c, l = codes[v]
                        vals.append(c); lens.append(l)
                    else:
                        vals.append(v); lens.append(n)
                if not lens:
                    payload[m] = b""; consumed[m] = -1
                    continue
                data, nb = _pack_bits(np.asarray(vals, dtype=np.int64), np.asarray(lens, dtype=np.int64))
                words = (nb + 31) // 32
                if words == 0:
                    # a tree-less/zero-length-code stream: non-empty section so ptr != NULL
                    payload[m] = b"\0" * 8; consumed[m] = 0
                else:
                    payload[m] = data + b"\0" * (words * 4 - len(data)) + b"\0" * 4
                    consumed[m] = words * 4
        for p in range(3):
            s = FX0 + p
            bts = bytes(self.ops[s].bytes)
            if bts:
                payload[s] = bts + b"\0" * ((-len(bts)) % 4 + 4)
                consumed[s] = len(bts)
            else:
                payload[s] = b""; consumed[s] = -1
        # 2. layout
        table_sz = 4 * len(sections)
        body = bytearray()
        offsets = []
        cursors = [-1] * NSTREAMS
        for s in sections:
            offsets.append(len(body))
            pl = payload.get(s, b"")
            body += struct.pack(">I", len(pl))
            if pl:
                start = 8 + table_sz + len(body)
                cursors[s] = start + consumed[s]
            body += pl
        data = self.header + b"".join(struct.pack(">I", o) for o in offsets) + bytes(body)
        return data, cursors


class _Gen:
    def __init__(self, cfg: SynthConfig):
        self.cfg = cfg
        self.rng = np.random.default_rng(cfg.seed)
        w, h = cfg.width, cfg.height
        assert w % 8 == 0 and h % 8 == 0 and w >= 8 and h >= 8
        self.w, self.h = w, h
        self.hb, self.vb = w // 4, h // 4
        self.cws, self.chs = {"420": (1, 1), "444": (0, 0), "422": (1, 0)}[cfg.sampling]   # chroma plane shifts (h4m:845-848)
        self.cblk = (2 >> self.cws) * (2 >> self.chs)          # chroma blocks per macroblock (h4m:852-854)
        self.chb, self.cvb = (w >> self.cws) // 4, (h >> self.chs) // 4
        self.landscape = w >= h
        self.nest_w, self.nest_h = (70, 38) if self.landscape else (38, 70)
        self.picsize = w * h + 2 * (w >> self.cws) * (h >> self.chs)
        self.is15 = cfg.version == "1.5"
        self.smooth_mv = cfg.preset == "natural"
        p = cfg.preset
        if p == "dense":
            self.p_zero, self.run_mean = 0.35, 3.0
            self.p_dc_zero, self.dc_run_mean = 0.25, 3.0
            self.mcb_run_mean, self.proc_run_mean = 4.0, 3.0
            self.p_proc1 = 0.5
        elif p == "realistic":
            self.p_zero, self.run_mean = 0.6, 12.0
            self.p_dc_zero, self.dc_run_mean = 0.5, 8.0
            self.mcb_run_mean, self.proc_run_mean = 24.0, 12.0
            self.p_proc1 = 0.7
        elif p == "natural":
            # realistic sparsity + a coherent vector field: vectors follow a slow random walk from macroblock to
            # macroblock (what delta-coded vectors, h4m:1846-1860, are designed for), not independent draws
            self.p_zero, self.run_mean = 0.6, 12.0
            self.p_dc_zero, self.dc_run_mean = 0.5, 8.0
            self.mcb_run_mean, self.proc_run_mean = 24.0, 12.0
            self.p_proc1 = 0.7
        elif p == "flat":
            self.p_zero, self.run_mean = 0.9, 40.0
            self.p_dc_zero, self.dc_run_mean = 0.8, 30.0
            self.mcb_run_mean, self.proc_run_mean = 300.0, 300.0
            self.p_proc1 = 0.9
        else:
            raise ValueError(p)
        if cfg.p_zero >= 0:
            self.p_zero = cfg.p_zero
        if cfg.p_proc1 >= 0:
            self.p_proc1 = cfg.p_proc1

    # -- helpers ------------------------------------------------------------
    def _run(self, mean: float, lo: int = 0, hi: int = 255) -> int:
        n = int(self.rng.geometric(1.0 / (mean + 1.0))) - 1 + lo
        return min(n, hi)

    def _delta(self) -> int:
        r = self.rng.random()
        if r < 0.02:
            return int(self.rng.integers(-400, 401))      # exercises overflow symbols
        return int(np.rint(self.rng.laplace(0, 6)))

    def _intra_kind(self, luma_I: bool) -> int:
        """non-zero kind for an intra-coded block"""
        cfg = self.cfg
        if cfg.weird_kinds and self.rng.random() < 0.05:
            if luma_I and self.rng.random() < 0.3:
                return int(self.rng.choice([16, 17, 0x20, 0x46, 0x80, 0x88, 200]))
            return int(self.rng.choice([7, 9, 10, 12, 15]))
        ws = np.array([1, 1, 1, 1, 1, 1, cfg.literal_weight], dtype=float)
        k = int(self.rng.choice([8, 1, 2, 3, 4, 5, 6], p=ws / ws.sum()))
        return k

    def _inter_kind(self, predi_ok: bool) -> int:
        cfg = self.cfg
        if not predi_ok:
            return 6
        if cfg.weird_kinds and self.rng.random() < 0.05:
            return int(self.rng.choice([8, 9, 12, 15]))
        if self.rng.random() < 0.15 * cfg.literal_weight:
            return 6
        k = int(self.rng.integers(1, cfg.max_predi_bases + 2))   # bases = k-1
        if k == 6:
            k = 7
        return min(k, 15)

    def _emit_payload_intra(self, pic: _Picture, plane: int, kind: int):
        """literal or `kind` bases into fixvl/bufTree0 of `plane`."""
        if kind == 6:
            pic.ops[FX0 + plane].bytes += self.rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
        else:
            self._emit_bases(pic, plane, kind)

    def _emit_bases(self, pic: _Picture, plane: int, n: int):
        if n <= 0:
            return
        words = self.rng.integers(0, 65536, n)
        coefs = self.rng.integers(0, 256, n)
        fx = pic.ops[FX0 + plane].bytes
        bt = pic.ops[BT0 + plane]
        for wv, c in zip(words, coefs):
            fx += struct.pack(">H", int(wv))
            bt.leaf(int(c))

    # -- I picture (h4m:1970-2016) -----------------------------------------
    def gen_I(self) -> _Picture:
        rng = self.rng
        pic = _Picture()
        pic.kind = I_FRAME
        dc_shift = int(rng.choice(self.cfg.dc_shifts))
        unk_shift = int(rng.choice(self.cfg.unk_shifts))
        nest_x = int(rng.integers(0, self.hb - self.nest_w + 1)) if self.hb >= self.nest_w else 0
        if self.cfg.nest_overhang and self.hb >= self.nest_w:
            nest_x = self.hb - self.nest_w + self.cfg.nest_overhang
        nest_y = int(rng.integers(0, self.vb - self.nest_h + 1)) if self.vb >= self.nest_h else 0
        pic.header = struct.pack(">BBHHH", dc_shift, unk_shift, 0, nest_x, nest_y)
        # kinds: luma then chroma (Ipic_BasisNumDec)
        nY = self.hb * self.vb
        kY = np.zeros(nY, dtype=np.int32)
        i = 0
        while i < nY:
            if rng.random() < self.p_zero:
                pic.ops[BN0].leaf(0)
                n = self._run(self.run_mean)
                pic.ops[BNR0].leaf(n)
                i += 1 + n
            else:
                k = self._intra_kind(True)
                pic.ops[BN0].leaf(k)
                kY[i] = k
                i += 1
        nC = self.chb * self.cvb
        kU = np.zeros(nC, dtype=np.int32)
        kV = np.zeros(nC, dtype=np.int32)
        i = 0
        while i < nC:
            if rng.random() < self.p_zero:
                pic.ops[BN1].leaf(0)
                n = self._run(self.run_mean)
                pic.ops[BNR1].leaf(n)
                i += 1 + n
            else:
                while True:
                    u = self._intra_kind(False) if rng.random() > 0.3 else 0
                    v = self._intra_kind(False) if rng.random() > 0.3 else 0
                    if u | v:
                        break
                pic.ops[BN1].leaf((u & 0xF) | ((v & 0xF) << 4))
                kU[i], kV[i] = u & 0xF, v & 0xF
                i += 1
        # DC deltas (IpicDcvDec / getDeltaDC)
        for p, n in ((0, nY), (1, nC), (2, nC)):
            i = 0
            while i < n:
                if rng.random() < self.p_dc_zero:
                    for lf in _sovf_leaves(0):
                        pic.ops[DC0 + p].leaf(lf)
                    r = self._run(self.dc_run_mean)
                    pic.ops[RLE0 + p].leaf(r)
                    i += 1 + r
                else:
                    t = self._delta()
                    if t == 0:
                        t = 1
                    if self.cfg.long_escape and p == 0 and i == n // 2:
                        t = 127 * (self.cfg.long_escape - 1) + 5          # long_escape symbols: 127, 127, ..., 5
                    for lf in _sovf_leaves(t):
                        pic.ops[DC0 + p].leaf(lf)
                    i += 1
        # payloads, plane raster (IpicPlaneDec)
        for p, ks in ((0, kY), (1, kU), (2, kV)):
            for k in ks:
                k = int(k)
                if k == 0 or k == 8:
                    continue
                self._emit_payload_intra(pic, p, k)
        return pic

    # -- P/B picture (h4m:2018-2056) ---------------------------------------
    def _mc_extent(self, rx: int, ry: int):
        """largest in-plane coordinates touched by the MC reads of one macroblock:
        (luma col, luma row, chroma col, chroma row); half-pel rule per version (h4m:1329-1343)"""
        pdx, pdy = rx >> self.cws, ry >> self.chs
        hx, hy = (pdx & 1, pdy & 1) if self.is15 else (rx & 1, ry & 1)
        cex, cey = (8 >> self.cws) - 1, (8 >> self.chs) - 1    # a chroma macroblock is (8 >> shift) samples wide / high
        return ((rx >> 1) + 7 + (rx & 1), (ry >> 1) + 7 + (ry & 1),
                (pdx >> 1) + cex + hx, (pdy >> 1) + cey + hy)

    def _mv_legal(self, rx: int, ry: int, predi: bool) -> bool:
        """every MC read stays inside its own plane's storage (rows may run off the right
        edge into the next row: linear addressing) and the MC-nest window is in-buffer"""
        w, h = self.w, self.h
        if rx < 0 or ry < 0:
            return False
        lc, lr, cc, cr = self._mc_extent(rx, ry)
        cw, ch = w >> self.cws, h >> self.chs
        if lr * w + lc > w * h - 1 or cr * cw + cc > cw * ch - 1:
            return False
        if predi:
            if self.landscape:
                o = (rx // 2) + (ry // 2 - 16) * w - 32
                mx = o + 37 * w + 69
            else:
                o = (rx // 2) + (ry // 2 - 32) * w - 16
                mx = o + 69 * w + 37
            if o < 0 or mx > self.picsize - 1:
                return False
        return True

    def _mv_sane(self, rx: int, ry: int) -> bool:
        """reads stay inside the picture rectangle of every plane"""
        if rx < 0 or ry < 0:
            return False
        lc, lr, cc, cr = self._mc_extent(rx, ry)
        return lc <= self.w - 1 and lr <= self.h - 1 and cc <= (self.w >> self.cws) - 1 and cr <= (self.h >> self.chs) - 1

    def gen_PB(self, kind: int) -> _Picture:
        rng = self.rng
        cfg = self.cfg
        pic = _Picture()
        pic.kind = kind
        dc_shift = int(rng.choice(cfg.dc_shifts))
        unk_shift = int(rng.choice(cfg.unk_shifts))
        res = [int(rng.choice(cfg.mv_res_bits)) for _ in range(4)]    # h0 v0 h1 v1
        pic.header = struct.pack(">BBBBBBBB", dc_shift, unk_shift, res[0], res[1], res[2], res[3], 0, 0)
        mw, mh = self.w // 8, self.h // 8
        nm = mw * mh
        # 1. MCB type runs
        types = np.zeros(nm, dtype=np.int32)
        p_two = kind == P_FRAME and not cfg.p_future_refs
        allowed = (0, 1) if p_two else (0, 1, 2)
        wts = np.array([0.25, 0.75] if p_two else [0.2, 0.4, 0.4])
        runs: List[Tuple[int, int]] = []
        i = 0
        cur = int(rng.choice(allowed, p=wts))
        while i < nm:
            n = self._run(self.mcb_run_mean, lo=1, hi=100000)
            runs.append((cur, n))
            types[i:i + n] = cur
            i += n
            nxt = [a for a in allowed if a != cur]
            cur = int(rng.choice(nxt))
        ot = pic.ops[MTYPE]
        ot.raw(runs[0][0], 2)
        for lf in _uovf_leaves(runs[0][1]):
            ot.leaf(lf)
        for (pv, _), (cv, cn) in zip(runs[:-1], runs[1:]):
            ot.raw(0 if cv == (pv + 1) % 3 else 1, 1)
            for lf in _uovf_leaves(cn):
                ot.leaf(lf)
        # 2. proc runs over non-intra MCBs
        n_inter = int((types != 0).sum())
        procs = np.zeros(nm, dtype=np.int32)
        if n_inter:
            seq = np.zeros(n_inter, dtype=np.int32)
            pruns: List[Tuple[int, int]] = []
            i = 0
            cur = 1 if rng.random() < self.p_proc1 else 0
            while i < n_inter:
                n = self._run(self.proc_run_mean, lo=1, hi=100000)
                pruns.append((cur, n))
                seq[i:i + n] = cur
                i += n
                cur ^= 1
            procs[types != 0] = seq
            op = pic.ops[MPROC]
            op.raw(pruns[0][0], 1)
            for _, n in pruns:
                for lf in _uovf_leaves(n):
                    op.leaf(lf)
        # 3. motion vectors (absolute half-pel position), chosen before kinds so Predi legality is known
        ref_xy = np.zeros((nm, 2), dtype=np.int64)
        predi_ok = np.zeros(nm, dtype=bool)
        prev_ref = -1
        mvh = mvv = 0
        mv_ops: List[Tuple[int, int, int, int]] = []   # (sym_h,res_h, sym_v,res_v) + bits
        for m in range(nm):
            t = int(types[m])
            if t == 0:
                continue
            ref = t - 1
            if ref != prev_ref:
                prev_ref = ref
                mvh = mvv = 0
            rh, rv = res[2 * ref], res[2 * ref + 1]
            Rh, Rv = 1 << (rh + 5), 1 << (rv + 5)
            x, y = (m % mw) * 8, (m // mw) * 8
            want_predi = procs[m] == 0
            for attempt in range(24):
                if attempt == 23:
                    nh, nv = 0, 0
                elif self.smooth_mv and attempt < 8:
                    # random walk around the previous vector of this reference, occasional jump
                    if rng.random() < 0.03:
                        nh = int(rng.integers(-Rh, Rh)); nv = int(rng.integers(-Rv, Rv))
                    else:
                        nh = int(np.clip(mvh + rng.integers(-1, 2), -Rh, Rh - 1))
                        nv = int(np.clip(mvv + rng.integers(-1, 2), -Rv, Rv - 1))
                else:
                    nh = int(rng.integers(-Rh, Rh)); nv = int(rng.integers(-Rv, Rv))
                    if attempt > 8:
                        nh //= 4; nv //= 4
                rx, ry = 2 * x + nh, 2 * y + nv
                runoff = rng.random() < cfg.runoff_prob
                ok = self._mv_legal(rx, ry, False) and (runoff or self._mv_sane(rx, ry))
                if ok:
                    break
            predi_ok[m] = want_predi and self._mv_legal(rx, ry, True)
            ref_xy[m] = (rx, ry)
            for (new, old, r, R, stream) in ((nh, mvh, rh, Rh, MVH), (nv, mvv, rv, Rv, MVV)):
                d = new - old
                d = (d + R) % (2 * R) - R
                sym = d >> r
                rbits = d - (sym << r)
                pic.ops[stream].leaf(sym & 0xFF)
                pic.ops[stream].raw(rbits, r)
            mvh, mvv = nh, nv
        # 4. pass 1 (spread_PB_descMap): DC + kinds ; pass 2 payload ops are queued per MCB
        rleY = rleC = 0
        nc = self.cblk
        kinds = np.zeros((nm, 4 + 2 * nc), dtype=np.int32)   # Y TL,BL,BR,TR, then U blocks, then V blocks
        for m in range(nm):
            t = int(types[m])
            intra = t == 0
            if intra:
                for p, cnt in ((0, 4), (1, nc), (2, nc)):
                    for _ in range(cnt):
                        d = self._delta() if rng.random() > self.p_dc_zero else 0
                        if self.cfg.long_escape_pb and not getattr(pic, "_long_done", False):
                            d = 127 * (self.cfg.long_escape_pb - 1) + 5
                            pic._long_done = True
                        for lf in _sovf_leaves(d):
                            pic.ops[DC0 + p].leaf(lf)
            elif procs[m] == 1:
                continue
            pok = bool(predi_ok[m])
            for j in range(4):
                if rleY:
                    rleY -= 1
                    continue
                if rng.random() < self.p_zero:
                    pic.ops[BN0].leaf(0)
                    rleY = self._run(self.run_mean)
                    pic.ops[BNR0].leaf(rleY)
                else:
                    k = self._intra_kind(False) if intra else self._inter_kind(pok)
                    k &= 0xF
                    pic.ops[BN0].leaf(k)
                    kinds[m, j] = k
            for jc in range(nc):
                if rleC:
                    rleC -= 1
                elif rng.random() < self.p_zero:
                    pic.ops[BN1].leaf(0)
                    rleC = self._run(self.run_mean)
                    pic.ops[BNR1].leaf(rleC)
                else:
                    while True:
                        u = (self._intra_kind(False) if intra else self._inter_kind(pok)) if rng.random() > 0.3 else 0
                        v = (self._intra_kind(False) if intra else self._inter_kind(pok)) if rng.random() > 0.3 else 0
                        if (u & 0xF) | (v & 0xF):
                            break
                    pic.ops[BN1].leaf((u & 0xF) | ((v & 0xF) << 4))
                    kinds[m, 4 + jc], kinds[m, 4 + nc + jc] = u & 0xF, v & 0xF
        # 5. pass 2 payloads (BpicPlaneDec second loop): per MCB, per plane, per block
        for m in range(nm):
            t = int(types[m])
            if t == 0:
                for j in range(4 + 2 * nc):
                    p = 0 if j < 4 else 1 if j < 4 + nc else 2
                    k = int(kinds[m, j])
                    if k == 0 or k == 8:
                        continue
                    self._emit_payload_intra(pic, p, k)
            elif procs[m] == 0:
                for j in range(4 + 2 * nc):
                    p = 0 if j < 4 else 1 if j < 4 + nc else 2
                    k = int(kinds[m, j])
                    if k == 6:
                        pic.ops[FX0 + p].bytes += rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
                    elif k != 0:
                        self._emit_bases(pic, p, k - 1)
                        for _ in range(2):
                            tval = int(np.rint(rng.laplace(0, 10)))
                            if rng.random() < 0.02:
                                tval = int(rng.integers(-300, 301))
                            if self.cfg.predi_big and rng.random() < self.cfg.predi_big:
                                tval = int(rng.integers(33000, 50000)) * (1 if rng.random() < 0.5 else -1)
                            for lf in _sovf_leaves(tval):
                                pic.ops[DC0 + p].leaf(lf)
        return pic


def _display_ids(gop: str) -> List[int]:
    """decode order -> display index inside the GOP (B pictures precede the anchor decoded before them)."""
    ids = [0] * len(gop)
    disp = 0
    i = 0
    n = len(gop)
    while i < n:
        # an anchor followed by its B pictures: the Bs display first
        j = i + 1
        while j < n and gop[j] == "B":
            j += 1
        nb = j - i - 1
        if gop[i] == "I" and i == 0:
            ids[i] = disp; disp += 1
            for k in range(nb):
                ids[i + 1 + k] = disp; disp += 1
        else:
            for k in range(nb):
                ids[i + 1 + k] = disp; disp += 1
            ids[i] = disp; disp += 1
        i = j
    return ids


@dataclass
class SynthClip:
    data: bytes
    width: int
    height: int
    version: str
    kinds: List[int]                       # frame type per picture, decode order
    cursors: List[List[int]]               # expected reader cursor per picture per stream
    pictures: List[bytes] = field(default_factory=list)   # picture data (after disp_id), decode order
    samp: int = 2                          # h_samp (= v_samp for 4:2:0 and 4:4:4)
    samp_v: int = 0                        # v_samp; 0: same as samp

    @property
    def samp_h(self) -> int:
        return self.samp

    def __post_init__(self):
        if not self.samp_v:
            self.samp_v = self.samp

    @property
    def picsize(self) -> int:
        ss = self.samp * self.samp_v
        return self.width * self.height * (ss + 2) // ss

    @property
    def n_pictures(self) -> int:
        return len(self.kinds)


def make_clip(cfg: SynthConfig) -> SynthClip:
    g = _Gen(cfg)
    assert cfg.gop[0] == "I", "streams must start with an I picture"
    assert "B" not in cfg.gop[:2] or len(cfg.gop) < 2, "a B picture needs two decoded anchors"
    body = bytearray()
    kinds: List[int] = []
    cursors: List[List[int]] = []
    pictures: List[bytes] = []
    max_frame = 0
    disp = _display_ids(cfg.gop)
    for _ in range(cfg.n_gops):
        frames = bytearray()
        for ch, d in zip(cfg.gop, disp):
            if ch == "I":
                pic = g.gen_I(); ft = I_FRAME
            elif ch == "P":
                pic = g.gen_PB(P_FRAME); ft = P_FRAME
            else:
                pic = g.gen_PB(B_FRAME); ft = B_FRAME
            data, cur = pic.assemble()
            payload = struct.pack(">I", d) + data
            frames += struct.pack(">HHI", 1, ft, len(payload)) + payload
            max_frame = max(max_frame, len(payload))
            kinds.append(ft); cursors.append(cur); pictures.append(data)
        body += struct.pack(">IIIII", 0, len(frames), len(cfg.gop), 0, 0x01000000) + frames
    if cfg.repeat_gops > 1:
        body = body * cfg.repeat_gops
        kinds = kinds * cfg.repeat_gops; cursors = cursors * cfg.repeat_gops; pictures = pictures * cfg.repeat_gops
    magic = (b"HVQM4 1.5" if cfg.version == "1.5" else b"HVQM4 1.3").ljust(16, b"\0")
    hdr = magic + struct.pack(">IIIIIIIII", 0x44, len(body), cfg.n_gops * cfg.repeat_gops, len(kinds), 0,
                              cfg.usec_per_frame, max_frame, 0, 0)
    samp, samp_v = {"420": (2, 2), "444": (1, 1), "422": (2, 1)}[cfg.sampling]
    hdr += struct.pack(">HHBBBBBBBBI", cfg.width, cfg.height, samp, samp_v, 0, 0, 0, 0, 0, 0, 0)
    assert len(hdr) == 0x44, len(hdr)
    return SynthClip(bytes(hdr + body), cfg.width, cfg.height, cfg.version, kinds, cursors, pictures, samp, samp_v)
