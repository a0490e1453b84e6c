"""Launch only the top-time kernel of a training step -- the fc1 weight gradient (wgrad_kernel + splitk_reduce_kernel, bf16) -- a few
times: target of the rocprofv3 --pmc passes that give roofline.traffic (HBM bytes per launch) in bench.py."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
Ny, Nx = 3072, 768
P = (torch.randn(M, Ny, device=dev) * 0.05).to(torch.bfloat16); Q = torch.randn(M, Nx, device=dev).to(torch.bfloat16)
out = torch.empty(Ny, Nx, device=dev); ws = torch.empty(16 * Ny * Nx, device=dev)
for _ in range(10):
    _lib.check(L.osud_op_wgrad(_lib.ptr(P), Ny, _lib.ptr(Q), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None))
torch.cuda.synchronize()
print("done", M)
