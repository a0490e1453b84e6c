#!/usr/bin/env python3
"""Turn one tools/collect_profiles.sh output directory into the round's profiles/ files (headers + the bench line of the profiled run
+ the kernel table).   python tools/assemble_profiles.py gpurun_out/r5q r05"""
import json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = "--no-cpu-baseline --no-family-table --no-roofline"
jobs = [
    ("train_trace_one_stream.md", "train_one.log", f"{tag}_train_kernel_trace.md",
     "rocprofv3 kernel trace of `python bench.py --mode train` on ONE stream (option wgrad_side_stream = 0; MI355X, bf16 tier, phased GEMM / weight-gradient loops)",
     f"OSUD_OPTIONS=wgrad_side_stream=0 rocprofv3 --kernel-trace -d /tmp/tr1 -o t -- python3 bench.py --mode train {B} --steps 50 --warmup 10"),
    ("train_trace_two_streams.md", "train_two.log", f"{tag}_train_kernel_trace_two_streams.md",
     "rocprofv3 kernel trace of `python bench.py --mode train`, default options (weight gradients on the side stream: kernels of the two streams overlap, so per-kernel durations are longer than on one stream while the step is shorter)",
     f"rocprofv3 --kernel-trace -d /tmp/tr2 -o t -- python3 bench.py --mode train {B} --steps 50 --warmup 10"),
    ("sample_trace_bf16.md", "sample_bf16.log", f"{tag}_sample_kernel_trace.md",
     "rocprofv3 kernel trace of `python bench.py --mode sample` (bf16 tier: outside the 1e-3 tolerance, reported under `sampling.also`)",
     f"rocprofv3 --kernel-trace -d /tmp/ts1 -o t -- python3 bench.py --mode sample {B} --no-parity-tier --steps 100 --warmup 10"),
    ("sample_trace_h8.md", "sample_h8.log", f"{tag}_sample_h8_kernel_trace.md",
     "rocprofv3 kernel trace of `python bench.py --mode sample --precision fp16f8` (the tolerance tier: fp16 + e4m3 operands on the phased loop, streamed split-bf16 attention)",
     f"rocprofv3 --kernel-trace -d /tmp/ts2 -o t -- python3 bench.py --mode sample --precision fp16f8 {B} --no-parity-tier --steps 100 --warmup 10"),
]
for table, log, out, title, cmd in jobs:
    line = ""
    for l in open(os.path.join(src, log), errors="replace"):
        if l.startswith('{"metric"'):
            line = l.strip()
    body = open(os.path.join(src, table)).read().split("\n", 2)[2]  # drop the summary's own title
    with open(os.path.join(root, "profiles", out), "w") as f:
        f.write(f"# Round {int(tag[1:])} -- {title}\n\nCommand: `{cmd}`\n(summarised from the rocpd database by tools/rocpd_summary.py; collected by "
                f"tools/collect_profiles.sh, assembled by tools/assemble_profiles.py)\n\nbench line of the profiled run:\n\n```\n{line}\n```\n\n{body.lstrip()}")
    print("wrote", out)
for name in ("pmc_mfma.json",):
    if os.path.isfile(os.path.join(src, name)):
        open(os.path.join(root, "profiles", f"{tag}_{name}"), "w").write(open(os.path.join(src, name)).read())
        print("wrote", f"{tag}_{name}")
