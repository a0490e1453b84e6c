"""Per-phase cycle breakdown of the GEMM main loop (needs ab/libosud_timing.so built with -DOSUD_GEMM_TIMING)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run(name, epi, My, Nx, K, f32out=False, pad=0):
    Y = torch.randn(My, K + pad, device=dev).to(torch.bfloat16); X = (torch.randn(Nx, K + pad, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(My, Nx, dtype=torch.float32 if f32out else torch.bfloat16, device=dev)
    bias = torch.randn(max(My, Nx), device=dev) * 0.02
    dbg = torch.zeros(16 * 8 * 8, device=dev)
    def go():
        _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K + pad, _lib.ptr(X), K + pad, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias), _lib.ptr(dbg), Nx, 128, max(1, My // 128), None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    go(); torch.cuda.synchronize(); e0.record(); go(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    for _ in range(1):
        _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K + pad, _lib.ptr(X), K + pad, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias), _lib.ptr(dbg), Nx, 128, max(1, My // 128), None))
    torch.cuda.synchronize()
    d = dbg.view(16, 8, 8).cpu()
    print(name, My, Nx, K, f"kernel {us:.1f} us; whole-kernel ticks (wave 0) {float(d[0,0,7]):.0f} -> {float(d[0,0,7]) / us / 1e3:.3f} ticks/ns")
    for w in (0, 4):
        v = d[0, w]
        n = max(1.0, float(v[4]))
        print(f"  wave {w}: per slab: wait_vm {v[0]/n:7.0f}  barrier {v[1]/n:7.0f}  dma_issue {v[2]/n:7.0f}  compute {v[3]/n:7.0f} | slabs {n:.0f}  epilogue/tile {v[5]/(n/ (K*2/128)):8.0f} (drain wait {v[6]/(n/ (K*2/128)):6.0f})")
D = 768
pad = int(os.environ.get("PAD", "0"))
run("fc1 plain bf16", _lib.EPI_BIAS_TE, 32768, 4 * D, D, pad=pad)
run("fc1 gelu bf16", _lib.EPI_BIAS_GELU_TE, 32768, 4 * D, D, pad=pad)
run("fc2-shape bf16", _lib.EPI_BIAS_TE, 32768, D, 4 * D, pad=pad)
run("square 8192", _lib.EPI_BIAS_TE, 8192, 8192, 8192, pad=pad)
