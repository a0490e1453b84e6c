"""Launch only the dominant kernel (fc1 GEMM, bias+GELU epilogue, bf16) a few times — target of the
rocprofv3 --pmc passes that give roofline.traffic (HBM bytes per launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N, K = 3072, 768
Yf = torch.randn(M, K, device=dev); Xf = torch.randn(N, K, device=dev) / K ** 0.5
Y = torch.empty(M, K, dtype=torch.bfloat16, device=dev); X = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
L.osud_op_convert(0, _lib.ptr(Yf), _lib.ptr(Y), Yf.numel(), None); L.osud_op_convert(0, _lib.ptr(Xf), _lib.ptr(X), Xf.numel(), None)
out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); bias = torch.randn(N, device=dev) * 0.02
for _ in range(10):
    _lib.check(L.osud_op_gemm(0, _lib.EPI_BIAS_GELU_TE, _lib.ptr(Y), K, _lib.ptr(X), K, M, N, K, _lib.ptr(out), N, _lib.ptr(bias), None, 0, 0, 0, None))
torch.cuda.synchronize()
print("done", M)
