#!/usr/bin/env python3
"""Drift of the tolerance-tier candidates from the fp32 tier over the 1000-step CFG-4 loop at the bench shape (the inputs of
bench.py's `parity_tier_and_drift`: same windows, noise and seeded weights), next to their step time -- one line per candidate:

    python tools/tier_drift.py fp16f8 fp16w8 fp16m8:3 fp16m8:7 ...      # fp16m8:<mask> sets option f16m8_forms (bit i = GEMM i on w8_t)
    SEEDS=3 python tools/tier_drift.py ...                               # more than one (windows, noise) draw: the max is a tail statistic

Used to choose which of a block's four GEMMs may drop the activation's residual (HISTORY.md section 2 "fp16w8")."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib  # noqa: E402
from osu_diffusion_amd.diffusion import create_diffusion  # noqa: E402
from osu_diffusion_amd.models import DiT_models  # noqa: E402
from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows  # noqa: E402

dev = torch.device("cuda:0")
num_classes, n, T, S = 52670, 64, 128, int(os.environ.get("S", "1000"))
cands = sys.argv[1:] or ["fp16f8", "fp16w8"]
diffusion = create_diffusion(str(S), noise_schedule="squaredcos_cap_v2")


def run(prec, z, kw, noise):
    name, _, mask = prec.partition(":")
    if mask:
        _lib.set_option("f16m8_forms", int(mask))
    torch.manual_seed(4321)
    model = randomize_zero_init(DiT_models["DiT-B"](num_classes=num_classes, context_size=19 - 3 + 128, precision=name).to(dev), seed=0).eval()
    model.reserve(2 * n, T)
    st = z.clone()
    diffusion.run_steps(model.forward_with_cfg, st, kw, first_step=S - 1, last_step=S - 2, step_noise=noise[:2])
    st = z.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    diffusion.run_steps(model.forward_with_cfg, st, kw, first_step=S - 1, last_step=0, step_noise=noise)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del model
    torch.cuda.empty_cache()
    return st[:n].clone(), dt / S * 1e3


worst = {c: [0.0, 0.0, 0.0, 0.0] for c in cands}
for seed in range(int(os.environ.get("SEEDS", "1"))):
    (x, o, c), y = synthetic_windows(n, T, num_classes, seed=1000 + seed, train_offsets=False)
    o, c = torch.cat([o, o]).to(dev), torch.cat([c, c]).to(dev)
    y = torch.cat([y, torch.full_like(y, num_classes)]).to(dev)
    kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=None)
    g = torch.Generator(device=dev).manual_seed(1234 + seed)
    z = torch.randn(n, 2, T, device=dev, generator=g)
    z = torch.cat([z, z])
    noise = torch.randn(S, 2 * n, 2, T, device=dev, generator=g)
    ref, ms32 = run("fp32", z, kw, noise)
    print(f"seed {seed}: fp32 tier {ms32:.2f} ms/step", flush=True)
    for cnd in cands:
        got, ms = run(cnd, z, kw, noise)
        d = (got - ref).abs().flatten().double()
        mx, p99, p999, mean = float(d.max()), float(torch.quantile(d, 0.99)), float(torch.quantile(d, 0.999)), float(d.mean())
        w = worst[cnd]
        w[0], w[1], w[2], w[3] = max(w[0], mx), max(w[1], p999), max(w[2], p99), ms
        print(f"  {cnd:12s} {ms:6.3f} ms/step = {1e3 / ms:6.1f} steps/s   drift max {mx:.3e}  p99.9 {p999:.3e}  p99 {p99:.3e}  mean {mean:.3e}", flush=True)
print("worst over the seeds:")
for cnd, (mx, p999, p99, ms) in worst.items():
    print(f"  {cnd:12s} {1e3 / ms:6.1f} steps/s   max {mx:.3e}  p99.9 {p999:.3e}  p99 {p99:.3e}")
