#!/usr/bin/env python3
"""Per-segment cycle stamps of the phased GEMM loop (needs a -DOSUD_PH_TIMING build: tools/build_gemm_variants.sh tm "-DOSUD_PH_TIMING";
run with OSUD_LIB=ab/libosud_tm.so).  Prints, per shape and wave group, cycles per phase spent in each segment, the epilogue, and the
shader clock the kernel ran at (stamped cycles / event-timed duration)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
M = int(os.environ.get("M", "32768")); D = 768
LOOP = int(os.environ.get("LOOP", "1"))
_lib.set_option("gemm_loop", LOOP)
SH = [("fc2 fwd  256x192", _lib.EPI_BIAS_TE, M, D, 4 * D), ("fc1 fwd  256x256 gelu", _lib.EPI_BIAS_GELU_TE, M, 4 * D, D),
      ("qkv dgrad 256x192", _lib.EPI_NONE_TE, M, D, 3 * D), ("fc2 dgrad 256x256", _lib.EPI_NONE_TE, M, 4 * D, D), ("4096^3 256x256", _lib.EPI_NONE_TE, 4096, 4096, 4096)]
H8 = os.environ.get("PREC") == "h8"  # the tolerance tier's operand form (fp16 + e4m3 rows), sampling shapes
if H8:
    M = int(os.environ.get("M", "16384"))
    SH = [("h8 fc2 256x192", _lib.EPI_NONE_F32, M, D, 4 * D), ("h8 fc1 256x256", _lib.EPI_BIAS_GELU_TE, M, 4 * D, D), ("h8 qkv 256x192", _lib.EPI_NONE_F32, M, 3 * D, D)]


def pack_h8(t, weight):
    R, C = t.shape
    o = torch.empty(R, 4 * C, dtype=torch.uint8, device=dev)
    _lib.check(L.osud_op_pack_h8(_lib.ptr(t), C, C, _lib.ptr(o), C, R, 1 if weight else 0, None))
    return o


TRAIN = os.environ.get("TRAIN")  # "code" / "rows": the fc1 launch of a TRAINING step -- bias + GELU with the saved derivative as the 8-bit code / as bf16 rows
if TRAIN:
    SH = [("fc1 fwd 256x256 gelu + derivative (" + TRAIN + ")", _lib.EPI_BIAS_GELU_TE, M, 4 * D, D)]
for name, epi, My, Nx, K in SH:
    Yf = torch.randn(My, K, device=dev); Xf = torch.randn(Nx, K, device=dev) / K ** 0.5
    if os.environ.get("ZERO") == "1":  # all-zero operands: what the same instruction stream costs when nothing toggles
        Yf.zero_(); Xf.zero_()
    if H8:
        Y, X = pack_h8(Yf, False), pack_h8(Xf, True)
        out = torch.zeros(My, 4 * Nx if epi == _lib.EPI_BIAS_GELU_TE else Nx, dtype=torch.uint8 if epi == _lib.EPI_BIAS_GELU_TE else torch.float32, device=dev)
    else:
        Y, X = Yf.to(torch.bfloat16), Xf.to(torch.bfloat16)
        out = torch.zeros(My, Nx, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(max(My, Nx), device=dev) * 0.02
    dbg = torch.zeros(32 * 8 * 8, device=dev)
    go = lambda: _lib.check(L.osud_op_gemm(_lib.PREC_F16F8 if H8 else 0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias), _lib.ptr(dbg), 0, 0, 0, None))
    if TRAIN:  # (a timing build of dit.hip hands the stamps out through osud_op_gemm_ex's colpart pointer)
        out2 = torch.zeros(My * Nx, dtype=torch.uint8, device=dev) if TRAIN == "code" else torch.zeros(My, Nx, dtype=torch.bfloat16, device=dev)
        go = lambda: _lib.check(L.osud_op_gemm_ex(0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias), _lib.ptr(out2), None,
                                                  1 if TRAIN == "code" else 0, _lib.ptr(dbg), None))
    for _ in range(10): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    d = dbg.view(32, 8, 8).cpu().double()
    tot = d[:, :, 6].mean().item()
    print(f"{name}: {us:.1f} us per launch, {2.0 * My * Nx * K / us / 1e6:.0f} TF; kernel {tot:.0f} cycles -> {tot / us / 1e3:.2f} GHz shader clock")
    for grp in (0, 1):
        w = d[:, 4 * grp:4 * grp + 4, :].reshape(-1, 8)
        nph = w[:, 4].mean().item()
        if nph == 0:  # the slab loop: the kernel total is all there is
            break
        if w[:, 0].sum().item() == 0:  # level-1 build: no per-segment stamps
            print(f"   group {grp}: phases {nph:.0f}  epilogue total {w[:, 5].mean().item():.0f}  main loop {(tot - w[:, 5].mean().item()) / nph:.0f} cycles per phase")
            continue
        seg = [w[:, i].mean().item() / nph for i in range(4)]
        print(f"   group {grp}: per phase: issue {seg[0]:.0f}  barrier-1 {seg[1]:.0f}  lgkm+cluster {seg[2]:.0f}  barrier-2 {seg[3]:.0f}  = {sum(seg):.0f} cycles;"
              f"  phases {nph:.0f}  epilogue total {w[:, 5].mean().item():.0f}  main loop total {sum(seg) * nph:.0f}")
