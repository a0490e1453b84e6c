"""Sampled sequence -> playable `.osu` difficulty (the reference's export/create_beatmap.py:22-147,173-213).

`create_beatmap(seq, ref_beatmap, version)` walks the (19, L) sequence — rows 0-1 positions NORMALISED to [0, 1] as
the sampler returns them, row 2 ms, rows 3.. type scores — and rebuilds circles, spinners and sliders:

* positions are rounded to whole osu! pixels;
* a slider collects its anchors (a corner anchor is written twice, a perfect-curve / Catmull anchor switches the
  path type), takes the time of its "last anchor" column as the length of one span, and on its end column gets
  - length  = (arc length of the full control path) x (progress of the path point nearest the sampled end position),
  - spans   = 1..3 from the end type, else round(duration / span),
  - an inherited timing point whose slider velocity makes the slider last exactly that long;
* only the uninherited timing points of the source beatmap are kept besides those.

The new difficulty copies the source file's General / Editor / Metadata / Difficulty / Events / Colours sections with
`Version` replaced and `BeatmapID` zeroed.  The reference writes the file through `slider`'s Beatmap.pack, whose exact
text is not reproducible here (package absent): the file written by osu_diffusion_amd.beatmap is standard v14 text
and is checked by re-reading it (tests/test_beatmap.py round trips).  Plotting / animation are out of scope.
"""
from __future__ import annotations

import copy
from typing import List

import numpy as np
import torch

from .beatmap import Beatmap, Circle, HitObject, Slider, Spinner, TimingPoint
from .curves import BEZIER, CATMULL, PERFECT, SliderPath, position_to_progress

__all__ = ["create_beatmap", "new_difficulty", "position_to_progress"]


def _blank_slider(x: int, y: int, time: float, new_combo: bool) -> Slider:
    return Slider(x=x, y=y, time=time, new_combo=new_combo, kind=BEZIER, points=[(x, y)], repeat=0, length=0.0, end_time=time)


def create_beatmap(seq: torch.Tensor, ref_beatmap: Beatmap, version: str) -> Beatmap:
    seq = torch.as_tensor(seq)
    hit_objects: List[HitObject] = []
    timing_points = [tp for tp in ref_beatmap.timing_points if tp.parent is None]
    current = None
    anchors: list = []
    path_type = None
    span_ms = 0.0
    for j in range(seq.shape[1]):
        x = int(round(float(seq[0, j] * 512)))
        y = int(round(float(seq[1, j] * 384)))
        time = float(seq[2, j])
        kind = int(torch.argmax(seq[3:, j]))
        sliding = isinstance(current, Slider)
        if kind == 0 or kind == 1:
            hit_objects.append(Circle(x=x, y=y, time=time, new_combo=kind == 1))
        elif kind == 2:
            current = Spinner(x=x, y=y, time=time, new_combo=True, end_time=time)
        elif kind == 3 and isinstance(current, Spinner):
            current.end_time = time
            hit_objects.append(current)
        elif kind == 4 or kind == 5:
            current = _blank_slider(x, y, time, kind == 5)
            anchors, path_type = [(x, y)], BEZIER
        elif kind == 6 and sliding:
            anchors.append((x, y))
        elif kind == 7 and sliding:
            anchors.append((x, y))
            path_type = PERFECT
        elif kind == 8 and sliding:
            anchors.append((x, y))
            path_type = CATMULL
        elif kind == 9 and sliding:
            anchors += [(x, y), (x, y)]
        elif kind == 10 and sliding:
            anchors.append((x, y))
            span_ms = time - current.time
        elif sliding:
            # slider end (types 11..15; like the reference's catch-all this also takes a stray type-3 column)
            full = SliderPath(path_type, np.array(anchors, dtype=float))
            length = full.get_distance() * position_to_progress(full, np.array((x, y), dtype=float))
            current.kind, current.letter = _resolved_kind(path_type, anchors), path_type[0]
            current.points = list(anchors)
            current.length = float(length)
            current.end_time = time
            duration = time - current.time
            current.repeat = int(round(duration / span_ms)) if kind > 13 else kind - 10
            current.edge_sounds = [0] * current.repeat
            current.edge_additions = ["0:0"] * current.repeat
            hit_objects.append(current)
            # inherited timing point: the slider velocity that makes `length` px take `span_ms` (offset on the whole ms the
            # slider head is written at, so that it is in force for the slider when the file is read back)
            tp = ref_beatmap.timing_point_at(current.time)
            parent = tp.parent if tp.parent is not None else tp
            sv = length * parent.ms_per_beat / (100 * ref_beatmap.slider_multiplier * span_ms)
            timing_points.append(TimingPoint(float(round(current.time)), -100 / sv if sv > 0 else -100, tp.meter, tp.sample_type,
                                             tp.sample_set, tp.volume, parent, tp.kiai_mode))
    return new_difficulty(ref_beatmap, version, hit_objects, timing_points)


def _resolved_kind(path_type: str, anchors) -> str:
    from .beatmap import slider_kind
    return slider_kind(path_type[0], anchors)


def new_difficulty(ref_beatmap: Beatmap, version: str, hit_objects: List[HitObject], timing_points: List[TimingPoint]) -> Beatmap:
    bm = Beatmap()
    bm.format_version = ref_beatmap.format_version
    bm.sections = copy.deepcopy(ref_beatmap.sections)
    bm.raw = copy.deepcopy(ref_beatmap.raw)
    bm.sections["Metadata"]["Version"] = version
    bm.sections["Metadata"]["BeatmapID"] = "0"
    bm.timing_points = list(timing_points)
    bm._hit_objects = list(hit_objects)
    return bm
