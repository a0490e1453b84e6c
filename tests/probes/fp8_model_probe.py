"""fp8 tier of the DiT forward: deviation from the fp32 oracle next to the bf16 tier's, and sampling speed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dit_oracle as mo
from osu_diffusion_amd.models import DiT
from osu_diffusion_amd.synthetic import synthetic_windows
shape = mo.DitShape(depth=4, hidden=384, heads=6, num_classes=10)
sd = mo.seeded_state_dict(shape, 77)
(x, o, c), y = synthetic_windows(4, 128, 10, seed=5)
t = torch.tensor([999, 500, 30, 0])
ref = mo.forward(sd, shape, x, t, o, c, y)
for prec in ("bf16", "fp8"):
    m = DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads, context_size=144, num_classes=shape.num_classes, precision=prec)
    m.load_state_dict(sd); m = m.to("cuda:0").eval()
    with torch.no_grad():
        got = m(x.cuda(), t.cuda(), o.cuda(), c.cuda(), y.cuda()).cpu()
    err = (got - ref)
    print(f"{prec}: max|d| {float(err.abs().max()):.4e}  rms {float(err.pow(2).mean().sqrt()):.4e}  (ref rms {float(ref.pow(2).mean().sqrt()):.3f}, max {float(ref.abs().max()):.3f})")
