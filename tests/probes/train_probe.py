import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dit_oracle as mo
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT
from osu_diffusion_amd.training import NativeTrainer
from tests.helpers import T, load, weights_for
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
fx = load("g7_train_l1")
shape, sd = weights_for(fx)
m = DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads, context_size=144, num_classes=shape.num_classes, class_dropout_prob=0.2, precision=prec)
m.load_state_dict(sd); m = m.to("cuda:0").eval()
tr = NativeTrainer(m, create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True))
terms = tr.step(T(fx["x"]), T(fx["o"]), T(fx["c"]), T(fx["y"]), t=T(fx["t"]), noise=T(fx["noise"]), drop_ids=T(fx["drop"]).long())
torch.cuda.synchronize()
print("terms", terms.cpu())
gv = {k: v.cpu() for k, v in tr.arena.grad_views().items()}
norms = dict(zip((str(s) for s in fx["grad_keys"]), fx["grad_norms"]))
for k, n in norms.items():
    got = float(gv[k].double().norm())
    flag = "" if abs(got - n) <= 2e-3 * max(n, 1e-4) else "   <<<<<< MISMATCH"
    print(f"{k:50s} ref {n:.6e} got {got:.6e}{flag}")
for k in fx:
    if k.startswith("grad:"):
        print(k, "max|d|", float((gv[k[5:]] - T(fx[k])).abs().max()), "max|ref|", float(T(fx[k]).abs().max()))
