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


ALL_SOURCES = ("s_cmp", "s_bitcmp", "s_cbranch", "global_store", "ds_write", "buffer_store", "scratch_store", "s_waitcnt", "s_barrier", "s_setprio", "s_nop")


def touches(t, hit):
    """How instruction text `t` uses the register(s) `hit` matches: "read" (a source operand), "write" (destination only) or None."""
    ops = t.split(None, 1)
    if len(ops) < 2:
        return None
    operands = [o.strip() for o in ops[1].split(",")]
    srcs = operands if ops[0].startswith(ALL_SOURCES) else operands[1:]
    if any(hit(o) for o in srcs):
        return "read"
    return "write" if hit(operands[0]) else None


def walk(start, hit, landing=None):
    """Every path from line `start` until the result has landed: a scalar result with the first s_waitcnt lgkmcnt(0), a vector claim (landing =
    its register name) with the asm statement `s_waitcnt vmcnt(N)` + `v_readfirstlane_b32 sX, vN`.  The compiler believes an asm result is
    there at once, so the hazard is a COPY: any instruction that READS the register in front of the landing fails (a spill, a v_mov / s_mov
    of it -- also one that is copied back later).  A path on which the register is WRITTEN first is a path on which no operation is in
    flight (both sides of the protocol's branches are laid out between the two statements; the compiler never overwrites a value it
    holds live) and ends there.  Both sides of conditional branches are followed."""
    bad, seen, work = [], set(), [start]
    while work:
        j = work.pop()
        steps = 0
        while j < len(lines):
            if j in seen:
                break
            seen.add(j)
            t = lines[j].strip()
            j += 1
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            steps += 1
            if steps > 8000:
                bad.append((j, "no landing within 8000 instructions"))
                break
            if landing is None and "s_waitcnt" in t and "lgkmcnt(0)" in t:
                break
            if landing is not None and re.match(r"v_readfirstlane_b32 s\d+, " + landing + r"$", t):
                k = j - 2
                while k >= 0 and (not lines[k].strip() or lines[k].strip().startswith(";")):
                    k -= 1
                if not re.match(r"s_waitcnt vmcnt\((\d+)\)", lines[k].strip()):
                    bad.append((j, "read without its counted wait: " + t))
                break
            use = touches(t, hit)
            if use == "read":
                bad.append((j, t))
                break
            if use == "write":
                break
            m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\S+)", t)
            if m:
                tgt = labels.get(m.group(2))
                if tgt is None:
                    bad.append((j, "unknown branch target " + t))
                    break
                if m.group(1) == "s_branch":
                    j = tgt
                    continue
                work.append(tgt)
            if t.startswith(("s_endpgm", "s_setpc")):
                break
    return bad


def reg_pattern(dest):
    """A predicate matching any mention of the registers in `dest` ("s37", "s[8:15]", "v131"), alone or inside a range operand."""
    kind = dest[0]
    m = re.match(kind + r"\[(\d+):(\d+)\]", dest)
    lo, hi = (int(m.group(1)), int(m.group(2))) if m else (int(dest[1:]), int(dest[1:]))
    singles = "|".join(f"{kind}{r}" for r in range(lo, hi + 1))

    def hits(text):
        if re.search(r"\b(" + singles + r")\b", text):
            return True
        for a, b in re.findall(kind + r"\[(\d+):(\d+)\]", text):
            if int(a) <= hi and int(b) >= lo:
                return True
        return False
    return hits


n = bad = 0
in_phased = False
for i, l in enumerate(lines):
    if re.match(r"^_ZN.*wgrad_phased_kernel.*:", l):
        in_phased = True
    if ".end_amdhsa_kernel" in l:
        in_phased = False
    m = re.search(r"(s_atomic_add|s_load_dwordx8) (s\d+|s\[\d+:\d+\]),", l)
    if m and "glc" in l:
        n += 1
        off = walk(i + 1, reg_pattern(m.group(2)))
        if off:
            bad += 1
            for j, t in off:
                print(f"line {i + 1}: {l.strip()} -- before the wait, line {j}: {t}")
        else:
            print(f"line {i + 1}: {l.strip()} -- no path touches {m.group(2)} before an s_waitcnt lgkmcnt(0)")
    m = re.search(r"global_atomic_add (v\d+), v\d+, v\d+, s\[\d+:\d+\] sc0$", l.strip())
    if m and in_phased and lines[i - 1].strip() == "s_mov_b64 exec, 1":  # (the protocol's own: the compiler's atomics carry the compiler's waits)
        n += 1
        off = walk(i + 1, reg_pattern(m.group(1)), landing=m.group(1))
        if off:
            bad += 1
            for j, t in off:
                print(f"line {i + 1}: {l.strip()} -- line {j}: {t}")
        else:
            print(f"line {i + 1}: {l.strip()} -- {m.group(1)} is not read before its s_waitcnt vmcnt(N) + v_readfirstlane")
print(f"{n} operations with late results, {bad} unsafe")
sys.exit(1 if bad or n == 0 else 0)
