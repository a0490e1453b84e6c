#!/usr/bin/env python3
"""For every dispatch of kernels matching <pattern> in a rocpd trace: its duration, and which kernels ran at the same time.
    python tools/overlap_probe.py trace.db adamw"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
rows = db.execute("select name, start, end from kernels order by start").fetchall()
last = rows[-len(rows) // 3:]  # the tail of the run (steady state)
for i, (n, s, e) in enumerate(last):
    if pat in n:
        others = [(m[:40], max(s, s2), min(e, e2)) for (m, s2, e2) in last if m != n and s2 < e and e2 > s]
        ov = sum(b - a for _, a, b in others)
        print(f"{n[:30]:30s} {(e - s) / 1e3:8.1f} us  overlapped {ov / 1e3:8.1f} us with {[o[0][:28] for o in others][:4]}")
