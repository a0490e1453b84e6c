#!/usr/bin/env python3
"""The scalar atomics of wgrad_phased_kernel's chunk queue return into an SGPR that inline asm defines "at once" as far as the compiler knows;
the value really arrives with the next s_waitcnt lgkmcnt(0).  This check compiles csrc/wgrad.hip to a listing and verifies that NOTHING reads
the destination register of any s_atomic_add between the instruction and the first lgkmcnt(0) wait on ANY path from it (both sides of every
conditional branch are followed; a compiler-made copy there would copy the increment, not the ticket).  Run after touching the kernel or changing the toolchain:
    python tools/check_satomic.py        (exit code 0 = clean)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "osu_diffusion_amd", "csrc", "wgrad.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "wgrad.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-mllvm",
                           "-amdgpu-atomic-optimizer-strategy=None", "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", l)
    if m:
        labels[m.group(1)] = i


def walk(start, reg):
    """Every path from line `start` to the first lgkmcnt(0) wait: returns a list of offending (line, text).  Both sides of a conditional branch
    are followed unless its direction is known: hipcc guards the "no claim this slot" code that sits between the asm and the join with
    `s_mov_b64 s[a:b], 0 | -1` ... `s_andn2_b64 vcc, exec, s[a:b]` ... `s_cbranch_vccnz / vccz`, so pairs set to 0 / -1 on the path and the vcc made
    from them are tracked.  Bounded: a path that has not met a wait after 3000 instructions counts as offending."""
    bad, seen, work = [], set(), [(start, 0, {}, None)]
    while work:
        j, depth, pairs, vcc = work.pop()
        pairs = dict(pairs)
        while j < len(lines):
            if (j, vcc) in seen:
                break
            seen.add((j, vcc))
            t = lines[j].strip()
            j += 1
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            depth += 1
            if depth > 3000:
                bad.append((j, "no wait within 3000 instructions"))
                break
            if "s_waitcnt" in t and "lgkmcnt(0)" in t:
                break
            if reg(t):
                bad.append((j, t))
                break
            m = re.match(r"s_mov_b64 (s\[\d+:\d+\]), (0|-1)$", t)
            if m:
                pairs[m.group(1)] = int(m.group(2))
            else:
                m = re.match(r"s_andn2_b64 vcc, exec, (s\[\d+:\d+\])$", t)
                if m:
                    vcc = {0: "nz", -1: "z"}.get(pairs.get(m.group(1)))
                elif re.match(r"\S+\s+(s\[\d+:\d+\]|vcc)\b", t):  # any other write to a tracked pair / vcc: forget it
                    w = re.match(r"\S+\s+(s\[\d+:\d+\]|vcc)\b", t).group(1)
                    if w == "vcc":
                        vcc = None
                    else:
                        pairs.pop(w, None)
            m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\S+)", t)
            if m:
                tgt = labels.get(m.group(2))
                if tgt is None:
                    bad.append((j, "unknown branch target " + t))
                    break
                kind = m.group(1)
                taken = None
                if kind == "s_cbranch_vccnz" and vcc is not None:
                    taken = vcc == "nz"
                if kind == "s_cbranch_vccz" and vcc is not None:
                    taken = vcc == "z"
                if kind == "s_branch" or taken is True:
                    j = tgt
                    continue
                if taken is None:
                    work.append((tgt, depth, pairs, vcc))
            if t.startswith(("s_endpgm", "s_setpc")):
                bad.append((j, "leaves the kernel before a wait: " + t))
                break
    return bad


def reg_pattern(dest):
    """A regex matching any mention of the SGPRs in `dest` ("s37" or "s[8:15]"), alone or inside a range operand."""
    m = re.match(r"s\[(\d+):(\d+)\]", dest)
    lo, hi = (int(m.group(1)), int(m.group(2))) if m else (int(dest[1:]), int(dest[1:]))
    singles = "|".join(f"s{r}" for r in range(lo, hi + 1))

    def hits(text):
        if re.search(r"\b(" + singles + r")\b", text):
            return True
        for a, b in re.findall(r"s\[(\d+):(\d+)\]", text):
            if int(a) <= hi and int(b) >= lo:
                return True
        return False
    return hits


n = bad = 0
for i, l in enumerate(lines):
    m = re.search(r"(s_atomic_add|s_load_dwordx8) (s\d+|s\[\d+:\d+\]),", l)
    if not m or "glc" not in l:
        continue
    n += 1
    off = walk(i + 1, reg_pattern(m.group(2)))
    if off:
        bad += 1
        for j, t in off:
            print(f"line {i + 1}: {l.strip()} -- before the wait, line {j}: {t}")
    else:
        print(f"line {i + 1}: {l.strip()} -- no path touches {m.group(2)} before an s_waitcnt lgkmcnt(0)")
print(f"{n} scalar memory operations with late results, {bad} unsafe")
sys.exit(1 if bad or n == 0 else 0)
