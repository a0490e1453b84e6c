"""One beatmap, one variant (sample.py defaults): rows = 2, T tokens, banded mask — where does a sampling step go?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT_models
from osu_diffusion_amd.synthetic import banded_attn_mask, randomize_zero_init, synthetic_windows
dev = "cuda:0"
T_, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = randomize_zero_init(DiT_models["DiT-B"](num_classes=52670, context_size=144, precision="bf16").to(dev), seed=0).eval()
d = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
(x, o, c), y = synthetic_windows(1, T_, 52670, seed=1)
z = torch.randn(n, 2, T_, device=dev); z = torch.cat([z, z]); o = o.repeat(2 * n, 1).to(dev); c = c.repeat(2 * n, 1, 1).to(dev)
y = torch.cat([torch.arange(n), torch.full((n,), 52670)]).to(dev)
kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=banded_attn_mask(T_, 128).to(dev))
d.run_steps(m.forward_with_cfg, z.clone(), kw, 999, 990, seed=1)
torch.cuda.synchronize(); t0 = time.time()
d.run_steps(m.forward_with_cfg, z.clone(), kw, 999, 900, seed=1)
torch.cuda.synchronize(); dt = (time.time() - t0) / 100
print(f"T={T_} rows={2 * n}: {dt * 1e3:.3f} ms/step = {1 / dt:.1f} steps/s")
