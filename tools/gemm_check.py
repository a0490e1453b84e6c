"""Write the bf16 outputs of a few big-shape GEMM launches to a file (for bit comparisons between builds / env settings):
    python tools/gemm_check.py out.pt      then      python tools/gemm_check.py out2.pt --compare out.pt"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib

L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
res = {}
for name, epi, M, N, K in (("fc1", _lib.EPI_BIAS_GELU_TE, 8192, 3072, 768), ("plain", _lib.EPI_BIAS_TE, 4096, 1024, 4096), ("k64", _lib.EPI_BIAS_TE, 2048, 512, 64),
                           ("k192", _lib.EPI_BIAS_TE, 2048, 768, 192)):
    Yf = torch.randn(M, K, device=dev, generator=g); Xf = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    Y = torch.empty(M, K, dtype=torch.bfloat16, device=dev); X = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    L.osud_op_convert(0, _lib.ptr(Yf), _lib.ptr(Y), Yf.numel(), None); L.osud_op_convert(0, _lib.ptr(Xf), _lib.ptr(X), Xf.numel(), None)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); bias = torch.randn(N, device=dev, generator=g)
    for _ in range(3):
        _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, M, N, K, _lib.ptr(out), N, _lib.ptr(bias), None, 0, 0, 0, None))
    torch.cuda.synchronize()
    ref = Y.float() @ X.float().t() + bias
    if epi == _lib.EPI_BIAS_GELU_TE:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    print(name, "max|d| vs fp32 reference", float((out.float() - ref).abs().max()), "of", float(ref.abs().max()))
    res[name] = out.cpu()
torch.save(res, sys.argv[1])
if len(sys.argv) > 3:
    other = torch.load(sys.argv[3])
    for k in res:
        print(k, "bit-identical to", sys.argv[3], bool(torch.equal(res[k], other[k])))
