#!/usr/bin/env python3
"""One steady-state step cut out of a rocprofv3 rocpd kernel trace: the dispatches between the last two launches of a marker
kernel (default: train_loss_kernel for training, sampler_step_kernel for sampling), summed per kernel.
    python tools/step_cut.py trace_results.db [marker-substring]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"osud::", "", name).replace("unsigned short", "bf16")
    return re.sub(r"\(.*$", "", name)[:90]


def main(path, marker):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namec}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 3:
        raise SystemExit(f"fewer than 3 launches of '{marker}' in the trace")
    a, b = marks[-3], marks[-2]
    seg = rows[a:b]
    agg = {}
    for n, s, e in seg:
        k = short(n)
        c, t = agg.get(k, (0, 0))
        agg[k] = (c + 1, t + (e - s))
    busy = sum(t for _, t in agg.values())
    span = seg[-1][2] - seg[0][1]
    print(f"{len(seg)} dispatches, summed kernel time {busy / 1e6:.3f} ms, first start to last end {span / 1e6:.3f} ms\n")
    print("| launches | total µs | avg µs | kernel |\n|---:|---:|---:|---|")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {c} | {t / 1e3:.1f} | {t / 1e3 / c:.1f} | `{k}` |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "train_loss_kernel")
