"""CPU (-m "not gpu"): `.osu` text -> hit objects -> (19, L) sequence (osu_diffusion_amd.beatmap, restating
data_loading.py:32-135 over an own parser) and sequence -> `.osu` (osu_diffusion_amd.export, restating
export/create_beatmap.py:22-147).

The reference reads / writes files through the third-party `slider` package (absent, unpinned), so there is no golden
output for these two steps: the column layout is checked against hand-computed values from the reference's text, the
geometry is pinned separately (test_curves.py), and the two directions are checked against each other by round trips —
on a hand-written map covering every object / anchor type and on the toy beatmap the reference keeps under
testing/toy_datasets (copied as data: tests/golden/toy_beatmap.osu)."""
import os

import numpy as np
import pytest
import torch

from osu_diffusion_amd import beatmap as B
from osu_diffusion_amd import windows as W
from osu_diffusion_amd.curves import BEZIER, CATMULL, LINEAR, PERFECT
from osu_diffusion_amd.export import create_beatmap
from tests.helpers import GOLDEN, T, load

HAND = """osu file format v14

[General]
AudioFilename: audio.mp3
Mode: 0

[Metadata]
Title:hand made
Artist:nobody
Creator:tests
Version:all types
BeatmapID:123456
BeatmapSetID:99

[Difficulty]
HPDrainRate:5
CircleSize:4
SliderMultiplier:1.6
SliderTickRate:1

[Events]
//Background and Video events
0,0,"bg.png",0,0

[TimingPoints]
0,400,4,2,1,50,1,0
3000,-50,4,2,1,50,0,0
6000,500,4,2,1,60,1,1

[HitObjects]
100,100,400,1,0,0:0:0:0:
200,120,800,5,0,0:0:0:0:
256,192,1200,12,0,2000,0:0:0:0:
50,300,2400,2,0,L|150:300,1,100
300,50,3200,6,0,P|350:100|300:150,2,160
100,200,4000,2,0,B|150:150|200:200|200:200|260:260,1,180
400,300,5000,2,0,C|420:200|480:250,3,120
64,64,6400,2,0,P|96:64|128:64,4,64
32,32,7400,2,0,B|64:64,5,40
300,300,8000,1,0,0:0:0:0:
"""


def types_of(seq):
    return seq[3:].argmax(0).tolist()


def test_hand_made_map_layout():
    bm = B.Beatmap.parse(HAND)
    assert (bm.beatmap_id, bm.title, bm.artist, bm.slider_multiplier, bm.format_version) == (123456, "hand made", "nobody", 1.6, 14)
    hos = bm.hit_objects(stacking=False)
    assert [type(h).__name__ for h in hos] == ["Circle", "Circle", "Spinner"] + ["Slider"] * 6 + ["Circle"]
    assert [tp.parent is None for tp in bm.timing_points] == [True, False, True]
    assert bm.timing_points[1].parent is bm.timing_points[0]
    # slider kinds after osu!'s fallbacks: collinear "P" -> linear
    assert [h.kind for h in hos[3:9]] == [LINEAR, PERFECT, BEZIER, CATMULL, LINEAR, BEZIER]
    # durations: length * spans / (100 * SM * SV) * beat
    assert hos[3].end_time == pytest.approx(2400 + 100 / 160 * 400)                 # red point, SV 1
    assert hos[4].end_time == pytest.approx(3200 + 160 * 2 / (160 * 2.0) * 400)     # green point: SV 2, parent beat 400
    assert hos[6].end_time == pytest.approx(5000 + 120 * 3 / (160 * 2.0) * 400)
    assert hos[7].end_time == pytest.approx(6400 + 64 * 4 / 160 * 500)              # second red point resets SV
    seq = B.beatmap_to_sequence(bm)
    assert seq.dtype == torch.float32 and seq.shape[0] == 19
    assert torch.equal(seq[3:].sum(0), torch.ones(seq.shape[1]))                    # exactly one type per column
    expect = [0, 1, 2, 3,
              4, 10, 11,                 # linear, 2 points: head, last anchor, end (1 span)
              5, 7, 10, 12,              # perfect, new combo, 2 spans
              4, 6, 9, 10, 11,           # bezier: plain anchor, corner (its twin skipped), last
              4, 8, 10, 13,              # catmull, 3 spans
              4, 9, 10, 14,              # collinear perfect -> linear anchors; 4 spans -> "even"
              4, 10, 15,                 # 5 spans -> "odd"
              0]
    assert types_of(seq) == expect
    cols = seq.T.tolist()
    assert cols[0][:3] == [100, 100, 400] and cols[3][:3] == [256, 192, 2000]       # spinner end keeps the position
    # perfect slider: anchor at 1/2 of the first span, last anchor at the end of the first span, end at end_time
    assert cols[8][:3] == [350, 100, 3300] and cols[9][:3] == [300, 150, 3400] and cols[10][2] == 3600
    # bezier anchors are spread over the first span by control-point index (the skipped twin keeps its slot)
    assert [c[2] for c in cols[11:16]] == pytest.approx([4000, 4000 + 225 / 4, 4000 + 2 * 225 / 4, 4225, 4225])  # SV 2: 225 ms
    # end positions: `length` px of arc from the head
    assert cols[6][:2] == pytest.approx([150, 300])
    end = np.array(cols[10][:2])
    assert np.linalg.norm(end - [300, 100]) == pytest.approx(50, abs=0.5)   # circle of radius 50 around (300, 100) (+ 3 px run-out)
    assert cols[26][:2] == pytest.approx([32 + 40 / np.sqrt(2), 32 + 40 / np.sqrt(2)], abs=1e-4)


def test_repeat_type_and_kind_fallbacks():
    assert [B.repeat_type(r) for r in (1, 2, 3, 4, 5, 6, 7)] == [0, 1, 2, 3, 4, 3, 4]
    assert B.slider_kind("P", [(0, 0), (1, 1), (2, 0), (3, 3)]) == BEZIER
    assert B.slider_kind("P", [(0, 0), (1, 1), (2, 2)]) == LINEAR
    assert B.slider_kind("P", [(0, 0), (1, 1), (2, 0)]) == PERFECT
    assert B.slider_kind("c", [(0, 0), (1, 1)]) == CATMULL


def test_sliders_with_100_or_more_points_count_as_circles():
    pts = "|".join(f"{i}:{i % 7}" for i in range(1, 100))
    text = HAND.split("[HitObjects]")[0] + f"[HitObjects]\n0,0,1000,2,0,B|{pts},1,300\n"
    seq = B.beatmap_to_sequence(B.Beatmap.parse(text))
    assert seq.shape[1] == 1 and types_of(seq) == [0]


def test_old_format_timing_points_and_errors():
    text = "osu file format v5\n\n[Difficulty]\nSliderMultiplier:1\n\n[TimingPoints]\n100,500\n2000,-200\n\n[HitObjects]\n10,10,2500,2,0,L|110:10,1,100\n"
    bm = B.Beatmap.parse(text)
    assert bm.timing_points[1].parent is bm.timing_points[0]
    assert bm.hit_objects()[0].end_time == pytest.approx(2500 + 100 / (100 * 0.5) * 500)
    with pytest.raises(ValueError):
        B.Beatmap.parse("[HitObjects]\n1,2,3,1,0\n")
    with pytest.raises(ValueError):
        B.Beatmap.parse("osu file format v14\n[HitObjects]\n1,2,3\n")
    assert B.beatmap_to_sequence(B.Beatmap.parse("osu file format v14\n[HitObjects]\n")).shape == (19, 0)


def round_trip(bm, tmp_path):
    seq = B.beatmap_to_sequence(bm)
    norm = seq.clone()
    norm[0] /= 512
    norm[1] /= 384
    out = create_beatmap(norm, bm, "round trip")
    path = os.path.join(tmp_path, "out.osu")
    out.write_path(path)
    back = B.Beatmap.from_path(path)
    return seq, out, back, B.beatmap_to_sequence(back)


@pytest.mark.parametrize("source", ["hand", "toy"])
def test_sequence_to_osu_and_back(source, tmp_path):
    bm = B.Beatmap.parse(HAND) if source == "hand" else B.Beatmap.from_path(os.path.join(GOLDEN, "toy_beatmap.osu"))
    seq, out, back, seq2 = round_trip(bm, str(tmp_path))
    assert back.version == "round trip" and back.beatmap_id == 0 and back.title == bm.title
    assert back.sections["Difficulty"] == bm.sections["Difficulty"] and back.raw.get("Events") == bm.raw.get("Events")
    assert len(back.hit_objects()) == len(bm.hit_objects())
    assert seq2.shape == seq.shape and types_of(seq2) == types_of(seq)
    # times: whole-ms rounding of heads + span arithmetic.  Inner anchors are excluded: their times are spread by control-
    # point index, and a linear slider comes back as a Bezier with doubled corner points (the reference's scheme too)
    inner = torch.tensor([6 <= k <= 9 for k in types_of(seq)])
    assert (seq2[2, ~inner] - seq[2, ~inner]).abs().max() <= 1.0
    heads = [j for j, k in enumerate(types_of(seq)) if k in (4, 5)]
    for j in heads:
        k = j + 1
        while types_of(seq)[k] < 10:
            assert seq[2, j] <= seq2[2, k] <= seq2[2, k + 1] + 1e-3
            k += 1
    ends = torch.tensor([k >= 11 for k in types_of(seq)])
    assert (seq2[:2, ~ends] - seq[:2, ~ends].round()).abs().max() == 0   # written positions are the rounded pixels
    # slider ends: the length comes from the reference's fixed-step search for the path point nearest the rounded pixel
    err = (seq2[:2, ends] - seq[:2, ends]).norm(dim=0)
    assert float(err.max()) <= 5.0 and float(err.median()) <= 1.0
    # the red timing points survive, every slider gets exactly one inherited point
    reds = [tp for tp in bm.timing_points if tp.parent is None]
    assert [tp.offset for tp in back.timing_points if tp.parent is None] == [tp.offset for tp in reds]
    n_sliders = sum(isinstance(h, B.Slider) for h in bm.hit_objects())
    assert sum(tp.parent is not None for tp in back.timing_points) == n_sliders
    # a second trip is a fixed point (idempotence)
    seq3 = round_trip(back, str(tmp_path))[3]
    assert types_of(seq3) == types_of(seq2) and (seq3[:2] - seq2[:2]).abs().max() <= 5.0


def test_toy_beatmap_structure():
    bm = B.Beatmap.from_path(os.path.join(GOLDEN, "toy_beatmap.osu"))
    hos = bm.hit_objects()
    sliders = [h for h in hos if isinstance(h, B.Slider)]
    assert (len(hos), len(sliders), sum(isinstance(h, B.Spinner) for h in hos), len(bm.timing_points)) == (395, 138, 1, 248)
    seq = B.beatmap_to_sequence(bm)
    per_slider = sum(len(B.hit_object_columns(s)) for s in sliders)
    assert seq.shape[1] == (len(hos) - len(sliders) - 1) + 2 + per_slider
    heads = torch.tensor([k in (0, 1, 2, 4, 5) for k in types_of(seq)])
    assert (seq[2, heads][1:] >= seq[2, heads][:-1]).all()        # objects in time order
    assert (seq[2][1:] >= seq[2][:-1] - 1e-3).all()              # and every column inside its object
    for s in sliders:                                            # slider bodies are as long as the file says
        assert s.path().get_distance() == pytest.approx(s.length, abs=1e-6)
        assert s.end_time > s.time
    # first slider of the file, by hand: 180 px at SV 1, SliderMultiplier 1.8, beat 300 ms -> 300 ms
    assert (sliders[0].time, sliders[0].end_time) == (1508, pytest.approx(1808))
    # the sequence feeds the window contract like a parsed reference beatmap would
    (x, o, c), n = W.split_and_process_sequence_no_augment(seq)
    assert x.shape == (2, n) and o.shape == (n,) and c.shape == (144, n) and float(x.max()) <= 1.1   # a few objects sit just outside the playfield


def test_loader_over_osu_files(tmp_path):
    """train.py --data-path layout: <root>/TrackNNNNN/beatmaps/<6-digit id> name.osu (data_loading.py:327-347, :255)."""
    src = os.path.join(GOLDEN, "toy_beatmap.osu")
    for track, ident in ((0, "000123"), (1, "000456")):
        folder = tmp_path / f"Track{track:05d}" / "beatmaps"
        folder.mkdir(parents=True)
        (folder / f"{ident} toy.osu").write_text(open(src, encoding="utf-8").read(), encoding="utf-8")
    factory = W.WindowIterableFactory(128, 64, open_fn=B.open_beatmap_sequence)
    loader = W.get_data_loader(B.track_catalogue(str(tmp_path)), 0, 2, factory, batch_size=4, num_workers=0, drop_last=True)
    (x, o, c), y = next(iter(loader))
    assert x.shape == (4, 2, 128) and o.shape == (4, 128) and c.shape == (4, 144, 128)
    assert set(y.tolist()) <= {123, 456}
    labels = [int(v) for (_, yy) in loader for v in yy]
    assert set(labels) == {123, 456}


def test_file_dialects(tmp_path):
    """CRLF line ends, a UTF-8 byte-order mark, comments, blank lines, hold notes (mania) and unknown keys survive."""
    text = HAND.replace("[HitObjects]\n", "[HitObjects]\n// a comment\n\n64,192,300,128,0,900:0:0:0:0:\n")
    text = text.replace("Mode: 0", "Mode: 3\nSpecialStyle: 1")
    path = tmp_path / "crlf.osu"
    path.write_bytes(b"\xef\xbb\xbf" + text.replace("\n", "\r\n").encode("utf-8"))
    bm = B.Beatmap.from_path(str(path))
    assert bm.mode == 3 and bm.sections["General"]["SpecialStyle"] == "1"
    hos = bm.hit_objects()
    assert isinstance(hos[0], B.HoldNote) and hos[0].end_time == 900 and len(hos) == 11
    seq = B.beatmap_to_sequence(bm)
    assert types_of(seq)[0] == 0 and seq[2, 0] == 300            # a hold note counts as a circle (data_loading.py:121-125)
    again = B.Beatmap.parse(bm.pack())
    assert again.sections == bm.sections and again.raw == bm.raw
    assert [h.pack() for h in again.hit_objects()] == [h.pack() for h in hos]
    assert [tp.pack() for tp in again.timing_points] == [tp.pack() for tp in bm.timing_points]


def test_timing_point_lookup_and_velocity_clamp():
    bm = B.Beatmap.parse(HAND)
    assert bm.timing_point_at(-50) is bm.timing_points[0]         # before the first point: the first point
    assert bm.timing_point_at(2999.9) is bm.timing_points[0] and bm.timing_point_at(3000) is bm.timing_points[1]
    assert bm.timing_point_at(1e9) is bm.timing_points[2]
    fast = B.Beatmap.parse(HAND.replace("3000,-50,", "3000,-1,"))  # SV 100 is clamped to 10 (osu!'s rule)
    assert fast.slider_duration(3200, 160, 1) == pytest.approx(160 / (100 * 1.6 * 10.0) * 400)
    slow = B.Beatmap.parse(HAND.replace("3000,-50,", "3000,-5000,"))
    assert slow.slider_duration(3200, 160, 1) == pytest.approx(160 / (100 * 1.6 * 0.1) * 400)


def test_create_beatmap_matches_the_reference_export():
    """fixture g13_export: export/create_beatmap.py:22-147 of the reference run on a jittered copy of the toy beatmap's sequence
    (tests/golden/make_golden.py::g13_export).  Same objects in the same order: kinds, rounded pixel positions, times, combo flags;
    per slider the control points, path letter, span count and length (path length x nearest progress, 1e-9 px); and the
    slider-velocity timing points (the reference keeps times in whole microseconds, after an fp32 division, so velocities agree to 2e-4 relative)."""
    from osu_diffusion_amd.export import create_beatmap

    fx = load("g13_export")
    src = B.Beatmap.from_path(os.path.join(GOLDEN, "toy_beatmap.osu"))
    out = create_beatmap(T(fx["seq"]), src, "golden")
    objs = out._hit_objects
    assert len(objs) == len(fx["kind"])
    kinds = [0 if isinstance(o, B.Circle) else (1 if isinstance(o, B.Spinner) else 2) for o in objs]
    assert kinds == list(fx["kind"])
    p0 = 0
    for i, o in enumerate(objs):
        assert (o.x, o.y) == tuple(fx["xy"][i]) and bool(o.new_combo) == bool(fx["new_combo"][i]), i
        assert abs(o.time - fx["time"][i]) < 5e-3, i   # the reference divides the fp32 time by 1000 in fp32 and rounds to microseconds
        if kinds[i]:
            assert abs(o.end_time - fx["end_time"][i]) < 5e-3, i
        if kinds[i] == 2:
            n = int(fx["n_points"][i])
            assert [tuple(map(float, q)) for q in o.points] == [tuple(q) for q in fx["points"][p0:p0 + n]], i
            p0 += n
            assert o.repeat == int(fx["repeat"][i]) and o.letter == str(fx["letter"][i]), i
            assert abs(o.length - fx["length"][i]) <= 1e-9 * max(1.0, fx["length"][i]), i
    n_red = len([tp for tp in src.timing_points if tp.parent is None])
    new_tp = out.timing_points[n_red:]
    assert len(out.timing_points) == int(fx["n_timing_points"]) and len(new_tp) == len(fx["tp_offset"])
    for tp, off, mpb, par in zip(new_tp, fx["tp_offset"], fx["tp_ms_per_beat"], fx["tp_parent_ms_per_beat"]):
        assert abs(tp.offset - off) <= 0.5 + 1e-9                            # written on the whole ms of the slider head (export.py)
        assert abs(tp.ms_per_beat - mpb) <= 2e-4 * abs(mpb), (tp.ms_per_beat, mpb)  # span times differ by up to 4e-3 ms (see above)
        assert tp.parent.ms_per_beat == par
