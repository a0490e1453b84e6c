"""How many host threads give the best CPU-oracle step time on this box (sizes bench.py's cpu_baseline)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dit_oracle as mo
from osu_diffusion_amd.synthetic import synthetic_windows
shape = mo.shape_of("DiT-B", num_classes=100)
sd = mo.seeded_state_dict(shape, 0)
(x, o, c), y = synthetic_windows(128, 128, 100, seed=0)
t = torch.full((128,), 999)
print("cpu_count", os.cpu_count())
for n in (8, 16, 32, 64, 128):
    if n > (os.cpu_count() or 1): break
    torch.set_num_threads(n)
    with torch.no_grad():
        mo.forward_with_cfg(sd, shape, x[:8], t[:8], o[:8], c[:8], y[:8], 4.0)
        t0 = time.perf_counter(); mo.forward_with_cfg(sd, shape, x, t, o, c, y, 4.0); dt = time.perf_counter() - t0
    print(f"threads {n}: {dt:.2f} s / forward_with_cfg (batch 128 x 128 tokens)", flush=True)
