"""Target of the rocprofv3 --pmc passes: a few DiT-B training steps and sampling steps at the bench shapes (bf16 tier), so that
every hot kernel (forward / dgrad GEMMs, wgrad, attention forward / backward, the HBM-bound kernels) shows up with its counters.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o run -- \
        python3 tools/pmc_step.py [train_steps] [sample_steps]
then  python tools/pmc_summary.py gpurun_out/pmc_mfma > profiles/rNN_pmc_mfma.json
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd.diffusion import create_diffusion  # noqa: E402
from osu_diffusion_amd.models import DiT_models  # noqa: E402
from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows  # noqa: E402
from osu_diffusion_amd.training import NativeTrainer  # noqa: E402

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
if n_train:
    model = randomize_zero_init(DiT_models["DiT-B"](num_classes=52670, context_size=144, class_dropout_prob=0.2, precision="bf16").to(dev), seed=0).train()
    tr = NativeTrainer(model, create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True), lr=1e-4)
    (x, o, c), y = synthetic_windows(256, 128, 52670, seed=1)
    x, o, c, y = x.to(dev), o.to(dev), c.to(dev), y.to(dev)
    for _ in range(n_train):
        tr.step(x, o, c, y)
    torch.cuda.synchronize()
    del tr, model
    torch.cuda.empty_cache()
if n_sample:
    from osu_diffusion_amd import _lib as _l
    _l.set_option("sample_graph", 0)  # plain launches: one counter record per kernel
    model = randomize_zero_init(DiT_models["DiT-B"](num_classes=52670, context_size=144, precision="bf16").to(dev), seed=0).eval()
    (x, o, c), y = synthetic_windows(64, 128, 52670, seed=2, train_offsets=False)
    o, c = torch.cat([o, o]).to(dev), torch.cat([c, c]).to(dev)
    y = torch.cat([y, torch.full_like(y, 52670)]).to(dev)
    z = torch.randn(128, 2, 128, device=dev)
    d = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
    d.run_steps(model.forward_with_cfg, z, dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=None), first_step=999, last_step=999 - n_sample + 1,
                step_noise=torch.randn(n_sample, 128, 2, 128, device=dev))
    torch.cuda.synchronize()
print("done")
