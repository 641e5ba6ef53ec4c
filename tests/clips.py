"""Seeded synthetic clip catalogue shared by the CPU and GPU suites."""
from hvqm4_amd.synth import SynthConfig, make_clip

# (name, config) -- small enough for the scalar oracle to finish in well under a second each
SMALL = [
    ("i16", SynthConfig(width=16, height=16, gop="I", seed=1)),
    ("ip8", SynthConfig(width=8, height=8, gop="IPP", seed=2)),
    ("ipb32", SynthConfig(width=32, height=32, gop="IPB", seed=3)),
    ("gop64x48_15", SynthConfig(width=64, height=48, gop="IPBBPBB", seed=4, version="1.5")),
    ("gop64x48_13", SynthConfig(width=64, height=48, gop="IPBBPBB", seed=5, version="1.3")),
    ("portrait48x64", SynthConfig(width=48, height=64, gop="IPBB", seed=6)),
    ("portrait152x280", SynthConfig(width=152, height=280, gop="IPB", seed=7)),
    ("nest_exact280x152", SynthConfig(width=280, height=152, gop="IPB", seed=8)),
    ("wide296x160", SynthConfig(width=296, height=160, gop="IPBB", seed=9, runoff_prob=0.5)),
    ("weird128x96", SynthConfig(width=128, height=96, gop="IPPBB", seed=10, weird_kinds=True, version="1.3")),
    ("weird64x64", SynthConfig(width=64, height=64, gop="IPBBP", seed=11, weird_kinds=True)),
    ("realistic128x96", SynthConfig(width=128, height=96, gop="IPBBPBB", seed=12, preset="realistic")),
    ("flat128x96", SynthConfig(width=128, height=96, gop="IPBBPBB", seed=13, preset="flat")),
    ("ragged24x40", SynthConfig(width=24, height=40, gop="IPBB", seed=14)),
    ("twogops64x48", SynthConfig(width=64, height=48, gop="IPBB", n_gops=3, seed=15)),
    ("natural128x96", SynthConfig(width=128, height=96, gop="IPBBPBB", seed=17, preset="natural")),
    ("bigshift64x48", SynthConfig(width=64, height=48, gop="IPBB", seed=16, dc_shifts=(3, 4), unk_shifts=(4, 10, 12))),
    # 4:4:4 (h_samp = v_samp = 1): four chroma blocks per macroblock, chroma planes as large as luma
    ("yuv444_64x48", SynthConfig(width=64, height=48, gop="IPBBPBB", seed=18, sampling="444")),
    ("yuv444_13_portrait48x64", SynthConfig(width=48, height=64, gop="IPBB", seed=19, sampling="444", version="1.3", runoff_prob=0.3)),
    ("yuv444_296x160", SynthConfig(width=296, height=160, gop="IPB", seed=20, sampling="444", weird_kinds=True)),
    # 4:2:2 as the reference spells it (h_samp 2, v_samp 1): two chroma blocks per macroblock, one above the other (h4m:852-870)
    ("yuv422_64x48", SynthConfig(width=64, height=48, gop="IPBBPBB", seed=43, sampling="422")),
    ("yuv422_13_portrait48x64", SynthConfig(width=48, height=64, gop="IPBB", seed=44, sampling="422", version="1.3", runoff_prob=0.3)),
    ("yuv422_296x160", SynthConfig(width=296, height=160, gop="IPB", seed=45, sampling="422", weird_kinds=True)),
    # fills the last cells of tests/test_fixture_coverage.py: inter kind 8 in a chroma plane, past (P) and future (B)
    ("weird160x128", SynthConfig(width=160, height=128, gop="IPPBBPBB", seed=34, weird_kinds=True)),
    # one DC delta of 300 overflow symbols (the reference sums for as long as the stream says, h4m:654-664): decoded like the
    # reference; beyond the fast parsers' cap (4096 symbols) the host parser follows the run to its end (round 5, tests/test_gpu_reject.py)
    ("longescape64x48", SynthConfig(width=64, height=48, gop="IPB", seed=35, long_escape=300)),
    # P pictures with future-referencing (type 2) macroblocks: the reference passes the picture being written as `future`
    # (h4m:2058-2061), so they read it in raster-order-dependent states and, where nothing has been written yet, what the player's
    # third buffer held before: nothing (first P), the last B picture (P after B), the picture of three anchors ago (P after P)
    ("pselfref64x48_15", SynthConfig(width=64, height=48, gop="IPBBPBPP", seed=36, p_future_refs=True)),
    ("pselfref64x48_13", SynthConfig(width=64, height=48, gop="IPPPBP", seed=37, p_future_refs=True, version="1.3")),
    ("pselfref444_48x64", SynthConfig(width=48, height=64, gop="IPBPP", seed=38, p_future_refs=True, sampling="444")),
    ("pselfref422_64x48", SynthConfig(width=64, height=48, gop="IPBPP", seed=46, p_future_refs=True, sampling="422")),
    # nearly every coded block a literal (kind 6, h4m:543-549): more literal blocks in a pair of tiles than a workgroup has lanes
    # MC-residual scalars beyond 16 bits (long overflow runs in the DC sections): the item records carry the payload offset instead
    ("bigscalars64x64", SynthConfig(width=64, height=64, gop="IPB", seed=42, predi_big=0.03, dc_shifts=(0,))),
    ("literals96x96", SynthConfig(width=96, height=96, gop="IPB", seed=40, literal_weight=400.0, p_zero=0.02)),
]

# Clips that once caught a defect; compared with the oracle like the others, not part of the golden manifest.
# longrun88x96: clip 1708 of the randomized sweep with seed 6006 -- a macroblock-type run "to the end of the picture" written as
# seventeen 0xFF symbols, one more than the parsers' cap: both parsers flagged HVQ_F_CAPPED and the back end refused a picture the
# reference decodes (round 4; a capped run length is exact, hvq_parse.c sym_uovf)
REGRESSION = [
    ("longrun88x96", SynthConfig(width=88, height=96, version="1.3", gop="IPBB", n_gops=2, seed=782179942, preset="flat",
                                 dc_shifts=(2, 0), unk_shifts=(9, 8), mv_res_bits=(1, 2), weird_kinds=True)),
]

MEDIUM = [
    ("c2_320x240_I", SynthConfig(width=320, height=240, gop="I", n_gops=4, seed=20)),
    ("c3_640x480", SynthConfig(width=640, height=480, gop="IPBBPBB", seed=21)),
    ("qvga_13", SynthConfig(width=320, height=240, gop="IPBBPBB", seed=22, version="1.3")),
    ("vga_realistic", SynthConfig(width=640, height=480, gop="IPBBPBB", seed=23, preset="realistic")),
    # nest windows that overlap the border column of the block map by one and by two entries: the reference indexes the bordered
    # map flat (h4m:1169), so it reads border values and the next row's first entries -- deterministic, reproduced exactly
    ("nest_border1_320x240", SynthConfig(width=320, height=240, gop="IPB", seed=24, nest_overhang=1)),
    ("nest_border2_296x160", SynthConfig(width=296, height=160, gop="IPB", seed=25, nest_overhang=2)),
    # up to 14 bases per MC-residual block, nearly every block coded: tiles with more (item, basis) pairs than a pair list holds
    # (1024) -- their items walk their bases themselves
    ("manybases320x240", SynthConfig(width=320, height=240, gop="IPB", seed=41, max_predi_bases=14, p_zero=0.02, p_proc1=0.0, literal_weight=0.1)),
    ("pselfref320x240", SynthConfig(width=320, height=240, gop="IPBBPBBP", seed=39, p_future_refs=True)),
]

# config C4 -- the per-GPU share at 8 GPUs: clips 0, 8, ..., 56 of the 64 (4 x 320x240 + 4 x 640x480, HVQM4 1.3
# and 1.5 alternating, seed = clip number), one 16-picture GOP each here (bench.py plays the GOP four times)
def _c4(i):
    small, v13 = (i // 8) % 2 == 0, ((i // 16) + i) % 2 == 0
    return (f"c4_clip{i:02d}", SynthConfig(width=320 if small else 640, height=240 if small else 480, version="1.3" if v13 else "1.5",
                                          gop="IPBBPBBPBBPBBPBB", seed=i))


C4_SHARE = [_c4(i) for i in range(0, 64, 8)]

_cache = {}


def get(name_cfg):
    name, cfg = name_cfg
    if name not in _cache:
        _cache[name] = make_clip(cfg)
    return _cache[name]
