"""Why does a 1e-6 move of x_999 change x_998 by O(1e-3)?  One fp32 sampler step at t = 999 from z and from z + 1e-6 * pert."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT_models
from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows

dev = torch.device("cuda:0")
n, T = 64, 128
(x, o, c), y = synthetic_windows(n, T, 52670, seed=1000, train_offsets=False)
o, c = torch.cat([o, o]).to(dev), torch.cat([c, c]).to(dev)
y = torch.cat([y, torch.full_like(y, 52670)]).to(dev)
g = torch.Generator(device=dev).manual_seed(1234)
z = torch.randn(n, 2, T, device=dev, generator=g); z = torch.cat([z, z])
noise = torch.randn(1, 2 * n, 2, T, device=dev, generator=g)
pert = torch.randn(n, 2, T, device=dev, generator=g); pert = torch.cat([pert, pert])
kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=None)
d = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
m = randomize_zero_init(DiT_models["DiT-B"](num_classes=52670, context_size=144, precision="fp32").to(dev), seed=0).eval()
t = torch.full((2 * n,), 999, device=dev)
outs = []
for e in (0.0, 1e-6, 0.0):
    st = (z + e * pert).clone()
    eps = m.forward_with_cfg(st, t, o, c, y, 4.0)[:, :2]
    A = float(d.sqrt_recip_alphas_cumprod[999]); B = float(d.sqrt_recipm1_alphas_cumprod[999])
    x0 = A * st - B * eps
    print(f"e={e:g}: eps rms {float(eps.pow(2).mean().sqrt()):.3f}, |x - eps| rms {float((st - eps).pow(2).mean().sqrt()):.3e}, "
          f"x0 in (-1, 2): {float(((x0 > -1) & (x0 < 2)).float().mean()):.4f}, x0 rms {float(x0.pow(2).mean().sqrt()):.3e}")
    d.run_steps(m.forward_with_cfg, st, kw, first_step=999, last_step=999, step_noise=noise)
    outs.append((st.clone(), x0.clone(), eps.clone()))
for name, i in (("x_998", 0), ("x0 pre-clamp", 1), ("eps", 2)):
    dd = (outs[0][i] - outs[1][i]).abs()
    print(f"{name}: moved-vs-plain max {float(dd.max()):.3e} mean {float(dd.mean()):.3e}; rerun max {float((outs[0][i] - outs[2][i]).abs().max()):.1e}")
dc = (outs[0][1].clamp(-1, 2) - outs[1][1].clamp(-1, 2)).abs()
print(f"clamped x0: max {float(dc.max()):.3e} mean {float(dc.mean()):.3e}, c1 = {float(d.posterior_mean_coef1[999]):.5f}")
