"""CPU: what CAN be pinned of the `.osu` parser without the `slider` package ("parity for parsing is unpinned").

The reference turns a beatmap into sequence columns in data_loading.py:65-124 -- row 0-1 position, row 2 time in ms, rows 3.. a one-hot
type -- from `slider`'s objects.  Three maps, expectations written out BY HAND from the file text and the published file format (osu!
wiki: hit-object type bits, slider duration = length x spans / (100 x SliderMultiplier x SV) beats): the first ten columns of the
shipped toy beatmap (circles with and without new combo, a Bezier slider whose doubled anchor makes it two straight segments: head,
corner anchor, last anchor, end), and two small maps written for this test (a linear slider with two spans, a spinner, a slider under an
inherited timing point, a collinear "perfect" slider that falls back to linear, a semicircular perfect curve cut at its quarter, a
Catmull anchor, four and five spans).  Nothing here is computed by the code under test or by the reference: every number is literal.
"""
import os

import pytest
import torch

from osu_diffusion_amd.beatmap import Beatmap, beatmap_to_sequence

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(seq, rows, tol=1e-3):
    assert seq.shape[0] == 19
    for i, (x, y, t, kind) in enumerate(rows):
        col = seq[:, i]
        assert abs(float(col[0]) - x) <= tol and abs(float(col[1]) - y) <= tol, (i, col[:3].tolist(), (x, y))
        assert abs(float(col[2]) - t) <= 2e-3, (i, float(col[2]), t)
        onehot = torch.zeros(16)
        onehot[kind] = 1.0
        assert torch.equal(col[3:], onehot), (i, col[3:].tolist(), kind)


def test_shipped_toy_beatmap_first_columns():
    """tests/golden/toy_beatmap.osu: red timing point at 458 ms with 300 ms per beat, SV 1, SliderMultiplier 1.8.
        120,237,458,5   circle, new combo (type bits 1 + 4)                     -> type 1
        34,355,608,1 ... 120,383,1058,1   plain circles                          -> type 0
        392,384,1508,6,0,B|433:341|433:341|433:220,1,180   slider, new combo     -> head type 5
          duration 180 / (100 x 1.8 x 1) x 300 = 300 ms; four control points, the inner two identical: the first of the pair is a corner
          (type 9) a third of the span in (1608), its twin is skipped; last anchor (433, 220) at the end of the span (type 10); the body
          is (392,384)-(433,341) = sqrt(41^2 + 43^2) = 59.4138 px, then straight up: 180 px of arc end 120.5862 px above (433,341)
          -> end (433, 220.4138), one span -> type 11."""
    seq = beatmap_to_sequence(Beatmap.from_path(os.path.join(GOLDEN, "toy_beatmap.osu")))
    _check(seq, [(120, 237, 458, 1), (34, 355, 608, 0), (172, 310, 758, 0), (34, 265, 908, 0), (120, 383, 1058, 0),
                 (392, 384, 1508, 5), (433, 341, 1608, 9), (433, 220, 1808, 10), (433, 220.4138, 1808, 11), (432, 96, 1958, 0)])


MAP_B = """osu file format v14

[General]
Mode: 0

[Difficulty]
SliderMultiplier:1.4
SliderTickRate:1

[TimingPoints]
1000,500,4,2,0,100,1,0
3000,-50,4,2,0,100,0,0

[HitObjects]
100,100,1000,1,0,0:0:0:0:
200,100,1500,2,0,L|300:100,2,100
256,192,2500,12,0,2900,0:0:0:0:
50,300,3000,2,0,L|50:200,1,80
400,50,4000,6,0,L|400:150|400:250,1,140
"""


def test_linear_sliders_spinner_and_an_inherited_timing_point():
    """500 ms per beat, SliderMultiplier 1.4.
        200,100,1500,2,0,L|300:100,2,100: two spans of 100 px at SV 1: 100 x 2 / 140 x 500 = 714.2857 ms; head type 4, no inner anchor,
          last anchor (300,100) at the end of the FIRST span (1857.1429, type 10), end = the body's end (300,100) at 2214.2857, two spans -> type 12
        256,192,2500,12,...,2900: spinner with new combo -> (2500, type 2), (2900, type 3)
        50,300,3000,2,0,L|50:200,1,80: the green point at 3000 (-50 -> SV 2) applies: 80 / (140 x 2) x 500 = 142.857 ms; the body is 100 px
          long but only 80 px are used: end (50, 220), type 11
        400,50,4000,6,0,L|400:150|400:250,1,140: new combo head (type 5), one inner anchor of a linear path (type 9) halfway through the span
          (140 / 280 x 500 = 250 ms -> 4125), last anchor (400,250) at 4250, end 140 px down the body (400,190)."""
    seq = beatmap_to_sequence(Beatmap.parse(MAP_B))
    assert seq.shape == (19, 13)
    _check(seq, [(100, 100, 1000, 0),
                 (200, 100, 1500, 4), (300, 100, 1857.1429, 10), (300, 100, 2214.2857, 12),
                 (256, 192, 2500, 2), (256, 192, 2900, 3),
                 (50, 300, 3000, 4), (50, 200, 3142.8571, 10), (50, 220, 3142.8571, 11),
                 (400, 50, 4000, 5), (400, 150, 4125, 9), (400, 250, 4250, 10), (400, 190, 4250, 11)])


MAP_C = """osu file format v14

[General]
Mode: 0

[Difficulty]
SliderMultiplier:1
SliderTickRate:1

[TimingPoints]
0,400,4,1,0,100,1,0

[HitObjects]
100,200,0,2,0,P|200:100|300:200,1,157.0796
10,10,1000,2,0,P|20:20|30:30,3,28.284271
100,300,2000,2,0,C|150:250|200:300,4,100
300,300,4000,2,0,B|350:300,5,50
"""


def test_perfect_curve_catmull_and_many_spans():
    """400 ms per beat, SliderMultiplier 1: a slider of L px x n spans lasts L n / 100 x 400 ms.
        P|200:100|300:200 from (100,200): the circle through the three points has centre (200,200), radius 100; the path is its upper
          half (pi x 100 = 314.16 px), cut at 157.0796 = a quarter: end (200,100); inner anchor of a perfect curve: type 7 at half the
          span (157.0796 / 100 x 400 = 628.3184 ms -> 314.1592), last anchor (300,200) at 628.3184, end type 11
        P|20:20|30:30 from (10,10): collinear -> a straight line (the fallback `slider` applies too): inner anchor type 9; three spans of
          28.284271 px (= the whole diagonal): 339.4113 ms; first span ends at 1113.1371; end (30,30), type 13
        C|150:250|200:300, four spans of 100 px -> 1600 ms: Catmull anchor type 8 at 2200, last anchor at 2400, end type 14 (even, >= 4)
          -- its position depends on the spline's arc length and is not asserted
        B|350:300, five spans of 50 px -> 1000 ms: no inner anchor, last anchor (350,300) at 4200, end (350,300) type 15 (odd, >= 5)."""
    seq = beatmap_to_sequence(Beatmap.parse(MAP_C))
    assert seq.shape == (19, 15)
    _check(seq[:, :3], [(100, 200, 0, 4), (200, 100, 314.1592, 7), (300, 200, 628.3184, 10)])
    _check(seq[:, 3:4], [(200, 100, 628.3184, 11)], tol=0.25)  # (a point ON the arc: the path is a polyline within the flattening tolerance of the circle)
    _check(seq[:, 4:8], [(10, 10, 1000, 4), (20, 20, 1056.5685, 9), (30, 30, 1113.1371, 10), (30, 30, 1339.4113, 13)], tol=2e-3)
    _check(seq[:, 8:11], [(100, 300, 2000, 4), (150, 250, 2200, 8), (200, 300, 2400, 10)])
    assert float(seq[2, 11]) == pytest.approx(3600.0, abs=2e-3) and float(seq[3 + 14, 11]) == 1.0
    _check(seq[:, 12:], [(300, 300, 4000, 4), (350, 300, 4200, 10), (350, 300, 5000, 15)])
