"""Slider geometry: control points -> piecewise-linear path -> positions by arc length.

The reference carries a numpy port of osu!lazer's path approximator for its export step
(export/path_approximator.py:10-253, export/slider_path.py:26-208).  This module restates it
array-at-a-time (de Casteljau levels and Catmull-Rom pieces are evaluated as whole arrays) with the
same constants and the same floating-point expression order, so the flattened vertices are bit-equal
for Bezier / Catmull / linear paths and agree to 1e-9 px for circular arcs (libm sin/cos);
tests/golden/g10_curves.npz pins this against the reference's own classes.

The same geometry serves `.osu` parsing (osu_diffusion_amd/beatmap.py): a slider's end position is
`SliderPath(kind, points, pixel_length).position_at(1)`.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

BEZIER_TOLERANCE = 0.25      # path_approximator.py:3
CATMULL_DETAIL = 50          # :4
CIRCULAR_ARC_TOLERANCE = 0.1  # :5

BEZIER, LINEAR, CATMULL, PERFECT = "Bezier", "Linear", "Catmull", "PerfectCurve"


def _halve(cp: np.ndarray):
    """de Casteljau split at 1/2 of one Bezier piece -> (left, right) control polygons (:187-204)."""
    n = len(cp)
    left, right = np.empty_like(cp), np.empty_like(cp)
    level = cp
    for i in range(n):
        left[i] = level[0]
        right[n - 1 - i] = level[-1]
        level = (level[:-1] + level[1:]) / 2
    return left, right


def _flat(cp: np.ndarray) -> bool:
    """All second differences shorter than 2 * tolerance (:177-184)."""
    if len(cp) < 3:
        return True
    d = cp[:-2] - 2 * cp[1:-1] + cp[2:]
    return not bool(np.any(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] > BEZIER_TOLERANCE * BEZIER_TOLERANCE * 4))


def _emit(cp: np.ndarray, out: List[np.ndarray]) -> None:
    """Vertices contributed by one flat-enough piece: its first control point, then the smoothed
    even-indexed points of its once-subdivided polygon (:207-228)."""
    n = len(cp)
    out.append(cp[:1].copy())
    if n > 2:
        left, right = _halve(cp)
        poly = np.concatenate([left, right[1:]], 0)
        even = np.arange(1, n - 1) * 2
        out.append(0.25 * (poly[even - 1] + 2 * poly[even] + poly[even + 1]))


def flatten_bezier(control_points: np.ndarray) -> np.ndarray:
    """Adaptive flattening of one Bezier piece, left to right (:10-86 with p = 0)."""
    control_points = np.asarray(control_points, dtype=float)
    if len(control_points) == 0:
        return np.zeros((0, 2))
    out: List[np.ndarray] = []
    pending = [control_points.copy()]
    while pending:
        piece = pending.pop()
        if _flat(piece):
            _emit(piece, out)
            continue
        left, right = _halve(piece)
        pending.append(right)
        pending.append(left)
    out.append(control_points[-1:].copy())
    return np.concatenate(out, 0)


def flatten_catmull(control_points: np.ndarray) -> np.ndarray:
    """Catmull-Rom through the control points, CATMULL_DETAIL pieces per span (:89-103, 231-253).  The
    reference emits both ends of every piece; the shared ends are bit-identical and SliderPath drops
    consecutive duplicates, so emitting each parameter once yields the same path."""
    cp = np.asarray(control_points, dtype=float)
    t = (np.arange(CATMULL_DETAIL + 1) / CATMULL_DETAIL)[:, None]
    t2 = t * t
    t3 = t * t2
    out = []
    for i in range(len(cp) - 1):
        v1 = cp[i - 1] if i > 0 else cp[i]
        v2 = cp[i]
        v3 = cp[i + 1]
        v4 = cp[i + 2] if i < len(cp) - 2 else v3 + v3 - v2
        out.append(0.5 * (2 * v2 + (-v1 + v3) * t + (2 * v1 - 5 * v2 + 4 * v3 - v4) * t2
                          + (-v1 + 3 * v2 - 3 * v3 + v4) * t3))
    return np.concatenate(out, 0) if out else np.zeros((0, 2))


def flatten_arc(control_points: np.ndarray) -> np.ndarray:
    """Circular arc through three points; empty when they are (nearly) collinear or coincide (:106-165)."""
    a, b, c = (np.asarray(p, dtype=float) for p in control_points[:3])
    sq = lambda v: float(np.inner(v, v))  # noqa: E731
    a_sq, b_sq, c_sq = sq(b - c), sq(a - c), sq(a - b)
    if np.isclose(a_sq, 0) or np.isclose(b_sq, 0) or np.isclose(c_sq, 0):
        return np.zeros((0, 2))
    s = a_sq * (b_sq + c_sq - a_sq)
    t = b_sq * (a_sq + c_sq - b_sq)
    u = c_sq * (a_sq + b_sq - c_sq)
    total = s + t + u
    if np.isclose(total, 0):
        return np.zeros((0, 2))
    centre = (s * a + t * b + u * c) / total
    d_a, d_c = a - centre, c - centre
    r = np.linalg.norm(d_a)
    theta_start = np.arctan2(d_a[1], d_a[0])
    theta_end = np.arctan2(d_c[1], d_c[0])
    while theta_end < theta_start:
        theta_end += 2 * np.pi
    direction = 1
    theta_range = theta_end - theta_start
    chord = c - a
    if np.dot(np.array([chord[1], -chord[0]]), b - a) < 0:  # b on the other side: go the long way round
        direction = -1
        theta_range = 2 * np.pi - theta_range
    if 2 * r <= CIRCULAR_ARC_TOLERANCE:
        count = 2
    else:
        count = int(max(2, np.ceil(theta_range / (2 * np.arccos(1 - CIRCULAR_ARC_TOLERANCE / r)))))
    fract = np.arange(count) / (count - 1)
    theta = theta_start + direction * fract * theta_range
    return centre + np.stack([np.cos(theta), np.sin(theta)], 1) * r


def flatten_linear(control_points: np.ndarray) -> np.ndarray:
    return np.array(control_points, dtype=float).reshape(-1, 2)


class SliderPath:
    """Piecewise-linear slider path with cumulative arc length (export/slider_path.py:26-208).

    `path_type`: "Bezier" | "Linear" | "Catmull" | "PerfectCurve"; a repeated control point starts a new
    piece.  With `expected_distance` the path is cut, or its last segment extended, to that length.
    `calculated_path` is (V, 2), `cumulative_length` (V,).

    Difference from the reference kept out on purpose: when the reference truncates, a slicing slip
    (slider_path.py:158) can leave stale vertices *behind* the cut in `calculated_path` (they are
    never reached through `cumulative_length`); here the path ends at the cut.
    """

    def __init__(self, path_type: str, control_points, expected_distance: Optional[float] = None):
        self.path_type = path_type
        self.control_points = np.asarray(control_points if control_points is not None else [], dtype=float).reshape(-1, 2)
        self.expected_distance = expected_distance
        self.calculated_path = self._flatten()
        self.cumulative_length = self._measure()

    # -- construction ------------------------------------------------------------------------------
    def _piece(self, cp: np.ndarray) -> np.ndarray:
        if self.path_type == LINEAR:
            return flatten_linear(cp)
        if self.path_type == PERFECT:
            if len(self.control_points) != 3 or len(cp) != 3:
                return flatten_bezier(cp)
            arc = flatten_arc(cp)
            return arc if len(arc) else flatten_bezier(cp)
        if self.path_type == CATMULL:
            return flatten_catmull(cp)
        return flatten_bezier(cp)

    def _flatten(self) -> np.ndarray:
        cps = self.control_points
        n = len(cps)
        if n == 0:
            return np.zeros((0, 2))
        breaks = [i + 1 for i in range(n) if i == n - 1 or bool((cps[i] == cps[i + 1]).all())]
        pieces, start = [], 0
        for end in breaks:
            pieces.append(self._piece(cps[start:end]))
            start = end
        verts = np.concatenate([p for p in pieces if len(p)], 0) if any(len(p) for p in pieces) else np.zeros((0, 2))
        if len(verts) > 1:  # drop consecutive duplicates
            keep = np.concatenate([[True], (verts[1:] != verts[:-1]).any(1)])
            verts = verts[keep]
        return verts

    def _measure(self) -> np.ndarray:
        path = self.calculated_path
        if len(path) == 0:
            return np.zeros(1)
        want = self.expected_distance
        cum = [0.0]
        length = 0.0
        for i in range(len(path) - 1):
            diff = path[i + 1] - path[i]
            d = float(np.linalg.norm(diff))
            if want is not None and want - length < d:  # cut inside this segment
                path[i + 1] = path[i] + diff * (want - length) / d
                self.calculated_path = path = path[: i + 2]
                cum.append(want)
                return np.asarray(cum, dtype=float)
            length += d
            cum.append(length)
        if want is not None and length < want and len(path) > 1:  # extend the last segment
            diff = path[-1] - path[-2]
            d = float(np.linalg.norm(diff))
            if d > 0:
                path[-1] = path[-1] + diff * (want - cum[-1]) / d
                cum[-1] = want
        return np.asarray(cum, dtype=float)

    # -- queries -----------------------------------------------------------------------------------
    def get_distance(self) -> float:
        return float(self.cumulative_length[-1]) if len(self.cumulative_length) else 0.0

    def progress_to_distance(self, progress) -> float:
        return float(np.clip(progress, 0, 1)) * self.get_distance()

    def _vertex_at(self, i: int, d: float) -> np.ndarray:
        path, cum = self.calculated_path, self.cumulative_length
        if len(path) == 0:
            return np.zeros(2)
        if i <= 0:
            return path[0]
        if i >= len(path):
            return path[-1]
        d0, d1 = cum[i - 1], cum[i]
        if np.isclose(d0, d1):
            return path[i - 1]
        return path[i - 1] + (path[i] - path[i - 1]) * ((d - d0) / (d1 - d0))

    def position_at(self, progress) -> np.ndarray:
        d = self.progress_to_distance(progress)
        return self._vertex_at(int(np.searchsorted(self.cumulative_length, d, side="left")), d)

    def path_to_progress(self, p0: float, p1: float) -> np.ndarray:
        """Vertices of the sub-path between two progress values (get_path_to_progress, :60-78)."""
        d0, d1 = self.progress_to_distance(p0), self.progress_to_distance(p1)
        cum, path = self.cumulative_length, self.calculated_path
        i = int(np.searchsorted(cum, d0, side="left"))
        j = max(i, int(np.searchsorted(cum, d1, side="left")))
        return np.vstack([self._vertex_at(i, d0)[None], path[i:j], self._vertex_at(j, d1)[None]])


def position_to_progress(path: SliderPath, pos: Sequence[float]) -> float:
    """Progress in [0, 1] of the path point nearest `pos`, by the reference's fixed-step descent from the end
    (export/create_beatmap.py:156-170): step = distance(t) - distance(t - 1e-4), at most 100 steps, stop when the
    step is 0 or t leaves [0, 1]."""
    pos = np.asarray(pos, dtype=float)
    eps = 1e-4
    t = 1
    for _ in range(100):
        step = np.linalg.norm(path.position_at(t) - pos) - np.linalg.norm(path.position_at(t - eps) - pos)
        t -= step
        if step == 0 or t < 0 or t > 1:
            break
    return float(np.clip(t, 0, 1))
