"""Sanity: 60 training steps of DiT-S on a fixed stream of synthetic windows, bf16 tier vs fp32 tier — the loss curves must
track each other (catches subtle bugs in the fused backward that a single-step gradient check could miss)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT_models
from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows
from osu_diffusion_amd.training import NativeTrainer
curves = {}
for prec in ("fp32", "bf16"):
    torch.manual_seed(0)
    m = randomize_zero_init(DiT_models["DiT-S"](num_classes=100, context_size=144, class_dropout_prob=0.2, precision=prec).to("cuda:0"), seed=0).train()
    tr = NativeTrainer(m, create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True), lr=2e-4)
    out = []
    for it in range(60):
        (x, o, c), y = synthetic_windows(32, 128, 100, seed=1000 + it % 8)
        g = torch.Generator().manual_seed(it)
        t = torch.randint(0, 1000, (32,), generator=g)
        noise = torch.randn(32, 2, 128, generator=g)
        drop = (torch.rand(32, generator=g) < 0.2).long()
        out.append(float(tr.step(x, o, c, y, t=t, noise=noise, drop_ids=drop)[2].mean()))
    curves[prec] = out
for i in range(0, 60, 6):
    print(f"step {i:3d}: fp32 {curves['fp32'][i]:.4f}  bf16 {curves['bf16'][i]:.4f}")
d = max(abs(a - b) for a, b in zip(curves["fp32"], curves["bf16"]))
print("max |fp32 - bf16| loss over 60 steps:", d, " first/last fp32:", curves["fp32"][0], curves["fp32"][-1])
