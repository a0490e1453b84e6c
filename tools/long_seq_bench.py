"""Sampling step time for ONE long beatmap (sample.py's real shape): n variants x 2 (CFG), T tokens, banded mask."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT_models
from osu_diffusion_amd.synthetic import banded_attn_mask, randomize_zero_init, synthetic_windows
dev = "cuda:0"
T_, n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 4
prec = os.environ.get("PRECISION", "bf16")  # bf16 | bf16x3 | fp16f8 | fp32 | fp8
m = randomize_zero_init(DiT_models["DiT-B"](num_classes=52670, context_size=144, precision=prec).to(dev), seed=0).eval()
d = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
(x, o, c), y = synthetic_windows(1, T_, 52670, seed=1)
z = torch.randn(n, 2, T_, device=dev); z = torch.cat([z, z]); o = o.repeat(2 * n, 1).to(dev); c = c.repeat(2 * n, 1, 1).to(dev)
y = torch.cat([torch.arange(n), torch.full((n,), 52670)]).to(dev)
for name, mask in (("banded", banded_attn_mask(T_, 128).to(dev)), ("no mask (T^2)", None)):
    kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=mask)
    K = 50
    d.run_steps(m.forward_with_cfg, z.clone(), kw, 999, 990, seed=1)
    torch.cuda.synchronize(); t0 = time.time()
    d.run_steps(m.forward_with_cfg, z.clone(), kw, 999, 1000 - K, seed=1)
    torch.cuda.synchronize(); dt = (time.time() - t0) / K
    print(f"[{prec}] T={T_} rows={2 * n} {name:14s}: {dt * 1e3:7.3f} ms/step = {1 / dt:6.1f} steps/s")
