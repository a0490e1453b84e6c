#!/usr/bin/env python3
"""Epilogue experiments on the dominant GEMM shapes (needs ab/libosud_exp.so built with -DOSUD_GEMM_EXP; run with
OSUD_LIB=ab/libosud_exp.so).  Every variant is a fresh process (the knobs are read once per process):
    python tools/gemm_exp.py            # driver: runs itself once per variant
"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [("baseline", {}), ("no stores", {"OSUD_GEMM_ORDER": "16"}), ("stores, no math", {"OSUD_GEMM_ORDER": "32"}),
            ("reads, no MFMA", {"OSUD_LIB": "ab/libosud_exp64.so"}), ("no LDS-DMA", {"OSUD_LIB": "ab/libosud_exp128.so"}),
            ("MFMA only (no reads)", {"OSUD_LIB": "ab/libosud_exp256.so"}), ("MFMA only, no DMA", {"OSUD_LIB": "ab/libosud_exp384.so"}),
            ("DMA only", {"OSUD_LIB": "ab/libosud_exp512.so"}),
            ("zero operands", {"OSUD_EXP_ZERO": "1"})]


def child():
    from osu_diffusion_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda:0")
    res = []
    for name, epi, M, N, K, f32 in (("fc1 gelu", _lib.EPI_BIAS_GELU_TE, 32768, 3072, 768, False), ("fc1 gelu", _lib.EPI_BIAS_GELU_TE, 16384, 3072, 768, False),
                                    ("qkv bias", _lib.EPI_BIAS_TE, 32768, 2304, 768, False), ("fc2 bias", _lib.EPI_BIAS_TE, 32768, 768, 3072, False),
                                    ("fc2 gate", _lib.EPI_GATE_RES, 16384, 768, 3072, True), ("sq4096", _lib.EPI_NONE_TE, 4096, 4096, 4096, False)):
        Yf = torch.randn(M, K, device=dev); Xf = torch.randn(N, K, device=dev) / K ** 0.5
        if os.environ.get("OSUD_EXP_ZERO"):
            Yf.zero_(); Xf.zero_()
        Y = torch.empty(M, K, dtype=torch.bfloat16, device=dev); X = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
        L.osud_op_convert(0, _lib.ptr(Yf), _lib.ptr(Y), Yf.numel(), None); L.osud_op_convert(0, _lib.ptr(Xf), _lib.ptr(X), Xf.numel(), None)
        out = torch.zeros(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        bias = torch.randn(N, device=dev) * 0.02
        gate = torch.randn(M // 128, N, device=dev)

        def go():
            _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, M, N, K, _lib.ptr(out), N, _lib.ptr(bias), _lib.ptr(gate), N, 128, M // 128, None))
        for _ in range(5):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(40):
            go()
        e1.record(); torch.cuda.synchronize()
        res.append(f"{name} M={M}: {e0.elapsed_time(e1) * 1e3 / 40:7.1f} us")
    print(" | ".join(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rep in range(2):
            for name, env in VARIANTS:
                r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True)
                print(f"{name:16s} {r.stdout.strip() or r.stderr[-300:]}", flush=True)
