"""`.osu` files <-> the (19, L) hit-object sequence the DiT path consumes, without the `slider` package.

The reference reads beatmaps through the third-party `slider` package (unpinned: requirements.txt:5) and turns its
objects into sequence columns in data_loading.py:32-135.  `slider` is absent here and the reference holds no fixture
of its output, so PARITY OF PARSING IS UNPINNED: this module follows the published osu! file format (v3..v14+) and
osu!'s own slider rules (duration = length * slides / (100 * SliderMultiplier * SV) beats, SV = clamp(-100 / beatLength,
0.1, 10); end position = the point at `length` px of arc along the path, osu_diffusion_amd/curves.py), and restates
the reference's column layout exactly:

    row 0-1 position (osu! px), row 2 time (ms), rows 3.. one-hot type
      0 circle              1 circle, new combo      2 spinner start        3 spinner end
      4 slider head         5 slider head, new combo
      6 Bezier anchor       7 perfect-curve anchor   8 Catmull anchor       9 red (corner) anchor
     10 last anchor (time = end of the first span)
     11/12/13 slider end, 1/2/3 spans     14 end, even spans >= 4     15 end, odd spans >= 5

What IS pinned here: the column layout against data_loading.py's text, the geometry against the reference's export
classes (fixture g10_curves), and sequence -> beatmap -> file -> sequence round trips (tests/test_beatmap.py).
"""
from __future__ import annotations

import os
import pickle
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .curves import BEZIER, CATMULL, LINEAR, PERFECT, SliderPath

feature_size = 19
_KIND_OF_LETTER = {"B": BEZIER, "L": LINEAR, "C": CATMULL, "P": PERFECT}
_LETTER_OF_KIND = {v: k for k, v in _KIND_OF_LETTER.items()}


# ----------------------------------------------------------------------------------------------- objects
@dataclass
class TimingPoint:
    """One [TimingPoints] line.  `parent` is None for an uninherited (red) point; an inherited (green) point carries
    its slider-velocity as `ms_per_beat` = -100 / SV and points at the red point it inherits the beat length from."""
    offset: float
    ms_per_beat: float
    meter: int = 4
    sample_type: int = 0
    sample_set: int = 0
    volume: int = 100
    parent: Optional["TimingPoint"] = None
    kiai_mode: int = 0

    def pack(self) -> str:
        return ",".join([_num(self.offset), _num(self.ms_per_beat), str(self.meter), str(self.sample_type),
                         str(self.sample_set), str(self.volume), "1" if self.parent is None else "0", str(int(self.kiai_mode))])


@dataclass
class HitObject:
    x: float
    y: float
    time: float            # ms
    new_combo: bool = False
    hitsound: int = 0
    combo_skip: int = 0
    addition: str = "0:0:0:0:"

    @property
    def position(self) -> Tuple[float, float]:
        return (self.x, self.y)

    def _type_bits(self, base: int) -> int:
        return base | (4 if self.new_combo else 0) | ((self.combo_skip & 7) << 4)


@dataclass
class Circle(HitObject):
    def pack(self) -> str:
        return f"{_num(self.x)},{_num(self.y)},{_ms(self.time)},{self._type_bits(1)},{self.hitsound},{self.addition}"


@dataclass
class Spinner(HitObject):
    end_time: float = 0.0

    def pack(self) -> str:
        return f"{_num(self.x)},{_num(self.y)},{_ms(self.time)},{self._type_bits(8)},{self.hitsound},{_ms(self.end_time)},{self.addition}"


@dataclass
class HoldNote(HitObject):
    end_time: float = 0.0

    def pack(self) -> str:
        return f"{_num(self.x)},{_num(self.y)},{_ms(self.time)},{self._type_bits(128)},{self.hitsound},{_ms(self.end_time)}:{self.addition}"


@dataclass
class Slider(HitObject):
    """`points` includes the head; `kind` is the path type after osu!'s fallbacks (a "P" slider that does not have exactly
    three points is a Bezier, one with collinear points is linear)."""
    kind: str = BEZIER
    points: List[Tuple[float, float]] = field(default_factory=list)
    repeat: int = 1
    length: float = 0.0
    end_time: float = 0.0
    edge_sounds: List[int] = field(default_factory=list)
    edge_additions: List[str] = field(default_factory=list)
    letter: Optional[str] = None    # the letter as written in the file (kept for a faithful rewrite)

    def path(self) -> SliderPath:
        return SliderPath(self.kind, np.asarray(self.points, dtype=float), self.length)

    def end_position(self) -> np.ndarray:
        """Where the slider body ends: `length` px of arc from the head (the reference asks `slider` for curve(1))."""
        return self.path().position_at(1)

    def pack(self) -> str:
        letter = self.letter or _LETTER_OF_KIND[self.kind]
        body = "|".join([letter] + [f"{_num(px)}:{_num(py)}" for px, py in self.points[1:]])
        out = f"{_num(self.x)},{_num(self.y)},{_ms(self.time)},{self._type_bits(2)},{self.hitsound},{body},{self.repeat},{_num(self.length)}"
        if self.edge_sounds or self.edge_additions:
            out += "," + "|".join(str(s) for s in self.edge_sounds) + "," + "|".join(self.edge_additions) + "," + self.addition
        return out


def _num(v: float) -> str:
    """Integers without a fraction, everything else with up to 15 significant digits (as editors write them)."""
    f = float(v)
    return str(int(f)) if f == int(f) and abs(f) < 1e15 else repr(round(f, 12))


def _ms(t: float) -> str:
    return str(int(round(float(t))))


def _collinear(p) -> bool:
    (ax, ay), (bx, by), (cx, cy) = p
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax) == 0


def slider_kind(letter: str, points: Sequence[Tuple[float, float]]) -> str:
    """Path type osu! actually uses for a slider written with `letter` (the fallbacks `slider`'s
    Curve.from_kind_and_points applies too, which decide the anchor types of data_loading.py:78-95)."""
    kind = _KIND_OF_LETTER.get(letter.upper(), BEZIER)
    if kind == PERFECT:
        if len(points) != 3:
            return BEZIER
        if _collinear(points):
            return LINEAR
    return kind


# ----------------------------------------------------------------------------------------------- the file
_KEYED = ("General", "Editor", "Metadata", "Difficulty")


class Beatmap:
    """A parsed `.osu` file.  Keyed sections are kept as ordered dicts of strings (unknown keys survive a rewrite),
    [Events] / [Colours] verbatim, timing points and hit objects as objects."""

    def __init__(self):
        self.format_version = 14
        self.sections: Dict[str, Dict[str, str]] = {k: {} for k in _KEYED}
        self.raw: Dict[str, List[str]] = {}
        self.timing_points: List[TimingPoint] = []
        self._hit_objects: List[HitObject] = []

    # -- reading -----------------------------------------------------------------------------------
    @classmethod
    def from_path(cls, path) -> "Beatmap":
        with open(path, "r", encoding="utf-8-sig", errors="replace") as f:
            return cls.parse(f.read())

    @classmethod
    def parse(cls, text: str) -> "Beatmap":
        bm = cls()
        lines = text.replace("\r\n", "\n").replace("\r", "\n").split("\n")
        head = next((ln for ln in lines if ln.strip()), "")
        if "file format v" not in head:
            raise ValueError("not an .osu file: missing 'osu file format vN' header")
        bm.format_version = int(head.strip().split("file format v")[1])
        section = None
        body: Dict[str, List[str]] = {}
        for ln in lines[lines.index(head) + 1:]:
            s = ln.strip()
            if s.startswith("[") and s.endswith("]"):
                section = s[1:-1]
                body.setdefault(section, [])
            elif section is not None and s and not (s.startswith("//") and section not in ("Events",)):
                body[section].append(ln.rstrip())
        for name in _KEYED:
            for ln in body.pop(name, []):
                if ":" in ln:
                    k, v = ln.split(":", 1)
                    bm.sections[name][k.strip()] = v.strip()
        tp_lines = body.pop("TimingPoints", [])
        ho_lines = body.pop("HitObjects", [])
        bm.raw = body
        bm.timing_points = _parse_timing_points(tp_lines)
        bm._hit_objects = [_parse_hit_object(ln, bm) for ln in ho_lines]
        return bm

    # -- fields the path uses ------------------------------------------------------------------------
    def _get(self, section, key, default, cast=str):
        v = self.sections[section].get(key)
        try:
            return default if v is None or v == "" else cast(v)
        except ValueError:
            return default

    beatmap_id = property(lambda self: self._get("Metadata", "BeatmapID", 0, int))
    beatmap_set_id = property(lambda self: self._get("Metadata", "BeatmapSetID", -1, int))
    title = property(lambda self: self._get("Metadata", "Title", ""))
    artist = property(lambda self: self._get("Metadata", "Artist", ""))
    creator = property(lambda self: self._get("Metadata", "Creator", ""))
    version = property(lambda self: self._get("Metadata", "Version", ""))
    mode = property(lambda self: self._get("General", "Mode", 0, int))
    slider_multiplier = property(lambda self: self._get("Difficulty", "SliderMultiplier", 1.4, float))
    slider_tick_rate = property(lambda self: self._get("Difficulty", "SliderTickRate", 1.0, float))
    circle_size = property(lambda self: self._get("Difficulty", "CircleSize", 5.0, float))

    def hit_objects(self, stacking: bool = False) -> List[HitObject]:
        """All hit objects in file order.  Stacking offsets are a display matter the reference switches off
        (data_loading.py:129), so `stacking=True` is not offered."""
        if stacking:
            raise NotImplementedError("stacked positions are not used by the diffusion path")
        return list(self._hit_objects)

    def timing_point_at(self, time_ms: float) -> TimingPoint:
        """The timing point in force at `time_ms`: the last one (file order) whose offset is <= the time, the first one
        for times before it."""
        if not self.timing_points:
            raise ValueError("beatmap has no timing points")
        current = self.timing_points[0]
        for tp in self.timing_points:
            if tp.offset <= time_ms:
                current = tp
            else:
                break
        return current

    def slider_duration(self, time_ms: float, length: float, repeat: int) -> float:
        """ms a slider of `length` px x `repeat` spans takes when it starts at `time_ms` (osu! stable's rule)."""
        tp = self.timing_point_at(time_ms)
        if tp.parent is not None:
            sv = min(max(-100.0 / tp.ms_per_beat, 0.1), 10.0) if tp.ms_per_beat < 0 else 1.0
            beat = tp.parent.ms_per_beat
        else:
            sv, beat = 1.0, tp.ms_per_beat
        return length * repeat / (100.0 * self.slider_multiplier * sv) * beat

    # -- writing -----------------------------------------------------------------------------------
    def pack(self) -> str:
        out = [f"osu file format v{self.format_version}", ""]
        for name in ("General", "Editor", "Metadata", "Difficulty"):
            out.append(f"[{name}]")
            sep = ": " if name in ("General", "Editor") else ":"
            out += [f"{k}{sep}{v}" for k, v in self.sections[name].items()]
            out.append("")
        out.append("[Events]")
        out += self.raw.get("Events", [])
        out.append("")
        out.append("[TimingPoints]")
        order = sorted(range(len(self.timing_points)), key=lambda i: (self.timing_points[i].offset,
                                                                      self.timing_points[i].parent is not None, i))
        out += [self.timing_points[i].pack() for i in order]
        out.append("")
        for name, lines in self.raw.items():
            if name != "Events":
                out += [f"[{name}]"] + lines + [""]
        out.append("[HitObjects]")
        out += [ho.pack() for ho in self._hit_objects]
        return "\n".join(out) + "\n"

    def write_path(self, path) -> None:
        with open(path, "w", encoding="utf-8", newline="\r\n") as f:
            f.write(self.pack())


def _parse_timing_points(lines: Sequence[str]) -> List[TimingPoint]:
    out: List[TimingPoint] = []
    last_red: Optional[TimingPoint] = None
    for ln in lines:
        f = [v.strip() for v in ln.split(",")]
        if len(f) < 2:
            continue
        beat = float(f[1])
        geti = lambda i, d: int(float(f[i])) if len(f) > i and f[i] != "" else d  # noqa: E731
        uninherited = (geti(6, 1) == 1) if len(f) > 6 else beat > 0
        tp = TimingPoint(float(f[0]), beat, geti(2, 4), geti(3, 0), geti(4, 0), geti(5, 100), None, geti(7, 0))
        if uninherited or last_red is None:
            last_red = tp
        else:
            tp.parent = last_red
        out.append(tp)
    return out


def _parse_hit_object(line: str, bm: Beatmap) -> HitObject:
    f = line.split(",")
    if len(f) < 5:
        raise ValueError(f"malformed hit object: {line!r}")
    x, y, t = float(f[0]), float(f[1]), float(f[2])
    kind_bits, hitsound = int(f[3]), int(f[4])
    common = dict(x=x, y=y, time=t, new_combo=bool(kind_bits & 4), hitsound=hitsound, combo_skip=(kind_bits >> 4) & 7)
    if kind_bits & 2:
        if len(f) < 8:
            raise ValueError(f"malformed slider: {line!r}")
        parts = f[5].split("|")
        letter = parts[0]
        pts = [(x, y)]
        for p in parts[1:]:
            if ":" in p:
                px, py = p.split(":")[:2]
                pts.append((float(px), float(py)))
        repeat, length = int(f[6]), float(f[7])
        edge_sounds = [int(v) for v in f[8].split("|") if v != ""] if len(f) > 8 else []
        edge_additions = [v for v in f[9].split("|") if v != ""] if len(f) > 9 else []
        addition = f[10] if len(f) > 10 else "0:0:0:0:"
        s = Slider(**common, addition=addition, kind=slider_kind(letter, pts), points=pts, repeat=repeat, length=length,
                   edge_sounds=edge_sounds, edge_additions=edge_additions, letter=letter)
        s.end_time = t + bm.slider_duration(t, length, repeat)
        return s
    if kind_bits & 8:
        return Spinner(**common, end_time=float(f[5]) if len(f) > 5 else t, addition=f[6] if len(f) > 6 else "0:0:0:0:")
    if kind_bits & 128:
        tail = f[5].split(":", 1) if len(f) > 5 else [str(t)]
        return HoldNote(**common, end_time=float(tail[0]), addition=tail[1] if len(tail) > 1 else "0:0:0:0:")
    return Circle(**common, addition=f[5] if len(f) > 5 else "0:0:0:0:")


# ------------------------------------------------------------------------------------- beatmap -> sequence
def repeat_type(repeat: int) -> int:
    """Slider-end type offset by span count: 1, 2, 3 spans -> 0, 1, 2; more -> 3 (even) / 4 (odd)  (data_loading.py:42-48)."""
    if repeat < 4:
        return repeat - 1
    return 3 if repeat % 2 == 0 else 4


def _column(x: float, y: float, time_ms: float, kind: int) -> List[float]:
    col = [0.0] * feature_size
    col[0], col[1], col[2] = float(x), float(y), float(time_ms)
    col[kind + 3] = 1.0
    return col


def hit_object_columns(ho: HitObject) -> List[List[float]]:
    """Sequence columns of one hit object (get_data, data_loading.py:65-125)."""
    if isinstance(ho, Slider) and len(ho.points) < 100:
        assert ho.repeat >= 1
        cols = [_column(ho.x, ho.y, ho.time, 5 if ho.new_combo else 4)]
        span = (ho.end_time - ho.time) / ho.repeat
        n = len(ho.points)
        for i in range(1, n - 1):          # inner anchors, spread evenly over the first span
            t = ho.time + i / (n - 1) * span
            p = ho.points[i]
            if ho.kind == LINEAR:
                cols.append(_column(*p, t, 9))
            elif ho.kind == CATMULL:
                cols.append(_column(*p, t, 8))
            elif ho.kind == PERFECT:
                cols.append(_column(*p, t, 7))
            elif p == ho.points[i + 1]:    # Bezier: the first of a doubled point is a corner ...
                cols.append(_column(*p, t, 9))
            elif p != ho.points[i - 1]:    # ... its twin is skipped, anything else is a plain anchor
                cols.append(_column(*p, t, 6))
        cols.append(_column(*ho.points[-1], ho.time + span, 10))
        end = ho.end_position()
        cols.append(_column(end[0], end[1], ho.end_time, 11 + repeat_type(ho.repeat)))
        return cols
    if isinstance(ho, Spinner):
        return [_column(ho.x, ho.y, ho.time, 2), _column(ho.x, ho.y, ho.end_time, 3)]
    return [_column(ho.x, ho.y, ho.time, 1 if ho.new_combo else 0)]


def beatmap_to_sequence(beatmap: Beatmap) -> torch.Tensor:
    """(19, L) float32 sequence of a beatmap (data_loading.py:127-135)."""
    cols = [c for ho in beatmap.hit_objects(stacking=False) for c in hit_object_columns(ho)]
    if not cols:
        return torch.zeros(feature_size, 0)
    return torch.tensor(cols, dtype=torch.float64).T.contiguous().float()


# -------------------------------------------------------------------------------------------- catalogues
def get_beatmap_idx(path: str) -> Dict[int, int]:
    """{beatmap id -> class index} pickle (data_loading.py:378-382; the reference resolves the name next to its sources)."""
    with open(path, "rb") as f:
        return pickle.load(f)


def track_catalogue(dataset_path: str):
    """`catalogue(start, end)` for windows.WindowDataset over the reference's dataset layout
    `<dataset>/TrackNNNNN/beatmaps/*` (data_loading.py:327-347; listing order = os.listdir, as there)."""
    def catalogue(start: int, end: int) -> List[str]:
        files = []
        for i in range(start, end):
            folder = os.path.join(dataset_path, "Track" + str(i).zfill(5), "beatmaps")
            files += [os.path.join(folder, name) for name in os.listdir(folder)]
        return files
    return catalogue


def open_beatmap_sequence(path: str) -> torch.Tensor:
    """`open_fn` for windows.WindowIterable when sources are `.osu` paths."""
    return beatmap_to_sequence(Beatmap.from_path(path))
