#!/usr/bin/env python3
"""Register / occupancy / spill / LDS table of every kernel in csrc (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
A streaming kernel that silently lands at one wave per SIMD runs at 70 % of the copy rate -- this table is how that was found.

  python tools/kernel_resources.py [file.hip ...] [--all]      # default: only kernels below 4 waves per SIMD or with spills
"""
import os, re, subprocess, sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "osu_diffusion_amd", "csrc")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def main():
    show_all = "--all" in sys.argv
    files = [a for a in sys.argv[1:] if a.endswith(".hip")] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    mk = open(os.path.join(CSRC, "Makefile")).read()
    rows = []
    for f in files:
        stem = os.path.splitext(os.path.basename(f))[0]
        m = re.search(rf"^FLAGS_{stem} *:= *(.*)$", mk, re.M)
        extra = m.group(1).split() if m else []
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", *extra, "-c", os.path.basename(f), "-o", "/dev/null",
               "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == "Function Name":
                cur = {"file": stem, "name": v}
                rows.append(cur)
            elif cur is not None:
                cur[k.split(" [")[0]] = int(v)
    names = demangle([r["name"] for r in rows])
    print(f"{'file':14s} {'VGPR':>5s} {'AGPR':>5s} {'waves':>5s} {'spill':>5s} {'LDS':>7s}  kernel")
    for r, n in zip(rows, names):
        if not show_all and r.get("Occupancy", 8) >= 4 and r.get("VGPRs Spill", 0) == 0:
            continue
        n = re.sub(r"\(anonymous namespace\)::|osud::", "", n)
        n = re.sub(r"\(.*", "", n)
        print(f"{r['file']:14s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('Occupancy', 0):5d} {r.get('VGPRs Spill', 0):5d} {r.get('LDS Size', 0):7d}  {n}")


if __name__ == "__main__":
    main()
