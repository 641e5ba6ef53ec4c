"""CPU: which (picture kind x block kind x plane x reference x proc) and which half-sample cases the clips of the parity
suites actually hit.  An empty cell fails: a path nobody decodes is a path nobody compares with the oracle."""
import os

from tests import clips, coverage

HERE = os.path.dirname(os.path.abspath(__file__))


def _catalogue():
    for case in clips.SMALL + clips.MEDIUM:
        yield case[0], clips.get(case)


def test_every_cell_of_the_path_is_hit_by_some_parity_clip():
    total = None
    per_clip = {}
    for name, cl in _catalogue():
        h = coverage.histogram(cl)
        per_clip[name] = h
        total = h if total is None else total + h
    missing = [c for c in coverage.required_cells() if total[c] == 0]
    assert not missing, f"{len(missing)} cells are never exercised, e.g. {missing[:12]}"
    # proc = 1 macroblocks carry arbitrary kind nibbles in the map only as zeros (the writer and the parsers agree)
    assert all(c[4] == "k0" for c in total if c[0] == "blk" and c[3].endswith("/proc"))


def test_committed_golden_clips_cover_both_versions_and_all_picture_kinds():
    """the .h4m files under tests/golden (hashed by the reference itself) on their own"""
    from hvqm4_amd.container import parse_header, video_pictures
    from hvqm4_amd.synth import SynthClip
    total = None
    for f in sorted(os.listdir(os.path.join(HERE, "golden"))):
        if not f.endswith(".h4m"):
            continue
        data = open(os.path.join(HERE, "golden", f), "rb").read()
        hd = parse_header(data)
        pics = list(video_pictures(data))
        cl = SynthClip(data, hd.width, hd.height, "1.5" if hd.is15 else "1.3", [p[0] for p in pics], [], [bytes(p[2]) for p in pics], hd.h_samp)
        h = coverage.histogram(cl)
        total = h if total is None else total + h
    for pic in "IPB":
        assert any(c[0] == "blk" and c[1] == pic for c in total)
    for version in ("1.3", "1.5"):
        for hx in (0, 1):
            for hy in (0, 1):
                assert sum(v for c, v in total.items() if c[0] == "mc" and c[1] == version and c[4:] == (hx, hy)) > 0, (version, hx, hy)
