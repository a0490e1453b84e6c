#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace as a per-kernel table (what `--stats` prints as CSV):
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/xxx.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"osud::", "", name)
    name = name.replace("unsigned short", "bf16")
    return name if len(name) < 110 else name[:107] + "..."


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namec}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       f"from kernels group by {namec} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of `{path}`\n")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, s, a, mn, mx in rows:
        print(f"| `{short(n)}` | {c} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * s / total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
