"""How sensitive is the noise prediction of a randomly initialised DiT-B to its input coordinates?  (GPU, fp32 tier.)
Explains bench.py's bf16_drift yardstick: eps moves by delta_eps for a 1e-6 move of x; x0 = A x - B eps with B ~ 2e4 at t = 999."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from osu_diffusion_amd.models import DiT_models
from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows

dev = torch.device("cuda:0")
n, T = 64, 128
(x, o, c), y = synthetic_windows(n, T, 52670, seed=1000, train_offsets=False)
o, c = torch.cat([o, o]).to(dev), torch.cat([c, c]).to(dev)
y = torch.cat([y, torch.full_like(y, 52670)]).to(dev)
g = torch.Generator(device=dev).manual_seed(1234)
z = torch.randn(n, 2, T, device=dev, generator=g); z = torch.cat([z, z])
pert = torch.randn(n, 2, T, device=dev, generator=g); pert = torch.cat([pert, pert])
for pos_gain in (0.1, 0.01):
    for prec in ("fp32", "bf16"):
        m = randomize_zero_init(DiT_models["DiT-B"](num_classes=52670, context_size=144, precision=prec).to(dev), seed=0, pos_gain=pos_gain).eval()
        for tval in (999, 500, 10):
            t = torch.full((2 * n,), tval, device=dev)
            a = m.forward_with_cfg(z, t, o, c, y, 4.0)
            a2 = m.forward_with_cfg(z, t, o, c, y, 4.0)
            row = [f"pos_gain={pos_gain} {prec} t={tval}: |eps| rms {a[:, :2].pow(2).mean().sqrt():.3f}, rerun diff {float((a - a2).abs().max()):.1e}"]
            for e in (1e-6, 1e-4, 1e-2):
                b = m.forward_with_cfg(z + e * pert, t, o, c, y, 4.0)
                d = (b - a)[:, :2].abs()
                row.append(f"dx={e:g}: d_eps max {float(d.max()):.2e} mean {float(d.mean()):.2e}")
            print(" | ".join(row), flush=True)
        del m
