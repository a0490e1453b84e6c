"""Full-size training step: report which gradients / losses are non-finite."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT_models
from osu_diffusion_amd.training import NativeTrainer
from osu_diffusion_amd.synthetic import synthetic_windows, randomize_zero_init
DEV = "cuda:0"
torch.manual_seed(0)
m = DiT_models["DiT-B"](num_classes=52670, context_size=144, class_dropout_prob=0.2, precision="bf16").to(DEV)
m = randomize_zero_init(m, seed=0).train()
tr = NativeTrainer(m, create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True), lr=1e-4)
(x, o, c), y = synthetic_windows(256, 128, 52670, seed=1)
t = torch.randint(1, 1000, (256,), generator=torch.Generator().manual_seed(2))
noise = torch.randn(256, 2, 128, generator=torch.Generator().manual_seed(3))
for it in range(3):
    terms = tr.step(x, o, c, y, t=t, noise=noise)
    torch.cuda.synchronize()
    print("step", it, "loss", float(terms[0].mean()), float(terms[2].mean()))
    bad = [(k, int((~torch.isfinite(v)).sum()), v.numel()) for k, v in tr.arena.grad_views().items() if not torch.isfinite(v).all()]
    print("non-finite grads:", bad[:3], len(bad))
    print("finite:", [k for k, v in tr.arena.grad_views().items() if torch.isfinite(v).all()])
    if bad: break
