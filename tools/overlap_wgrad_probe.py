#!/usr/bin/env python3
"""Feasibility probe: a block's backward chain (data-gradient GEMMs with HBM-bound kernels between them) on one stream, its weight
gradients on a second stream sized for OSUD_WGRAD_CUS compute units -- against everything on one stream.  Stand-ins: the chain's
HBM-bound kernels are device copies of the LayerNorm backward's bytes (453 MB) and the attention backward's (403 MB).

  OSUD_WGRAD_CUS=96 OSUD_GEMM_DYNAMIC=1 python tools/overlap_wgrad_probe.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
M, D = 32768, 768
bf = torch.bfloat16
def rnd(r, c): return (torch.randn(r, c, device=dev)).to(bf)
dbr, dz1, du, dqkv, dbr2 = rnd(M, D), rnd(M, 4 * D), rnd(M, D), rnd(M, 3 * D), rnd(M, D)
u2, g, ao, u1 = rnd(M + 1, D), rnd(M + 1, 4 * D), rnd(M + 1, D), rnd(M + 1, D)
w2t, w1t, wot, wqt = rnd(4 * D, D), rnd(D, 4 * D), rnd(D, D), rnd(D, 3 * D)
zp = rnd(M, 4 * D)
ws = [torch.empty(16 * 4 * D * D, device=dev) for _ in range(2)]
dW = [torch.empty(4 * D, D, device=dev), torch.empty(D, 4 * D, device=dev), torch.empty(D, D, device=dev), torch.empty(3 * D, D, device=dev)]
ca, cb = torch.empty(453 * 1000 * 1000 // 8, device=dev), torch.empty(453 * 1000 * 1000 // 8, device=dev)
da, db_ = torch.empty(403 * 1000 * 1000 // 8, device=dev), torch.empty(403 * 1000 * 1000 // 8, device=dev)
EPI_NONE_TE, EPI_GELUGRAD = 7, 9

def gemm(epi, Y, X, My, Nx, K, out, st, aux=None):
    _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, None, None, 0, 0, 0, st))

def wgrad(P, ldp, Q, ldq, Ny, Nx, out, wsb, st):
    _lib.check(L.osud_op_wgrad(_lib.ptr(P), ldp, _lib.ptr(Q), ldq, Ny, Nx, M, _lib.ptr(out), _lib.ptr(wsb), wsb.numel(), st))

def block(sa, sb, concurrent):
    """one block of the backward pass; sa: chain stream, sb: weight-gradient stream (== sa when serial)"""
    pa, pb = sa.cuda_stream, sb.cuda_stream
    ev = lambda s: (lambda e: (e.record(s), e)[1])(torch.cuda.Event())
    with torch.cuda.stream(sa):
        gemm(EPI_NONE_TE, dbr, w2t, M, 4 * D, D, dz1, pa)            # dgrad fc2 (stand-in for the gelu-grad epilogue)
    eA = ev(sa)
    if concurrent: sb.wait_event(eA)
    with torch.cuda.stream(sb):
        wgrad(dz1, 4 * D, u2, D, 4 * D, D, dW[0], ws[1 if concurrent else 0], pb)
        wgrad(dbr, D, g, 4 * D, D, 4 * D, dW[1], ws[1 if concurrent else 0], pb)
    with torch.cuda.stream(sa):
        gemm(EPI_NONE_TE, dz1, w1t, M, D, 4 * D, du, pa)             # dgrad fc1
        cb.copy_(ca)                                                   # LN2 backward
        gemm(EPI_NONE_TE, dbr2, wot, M, D, D, du, pa)                # dgrad out_proj
    eB = ev(sa)
    if concurrent: sb.wait_event(eB)
    with torch.cuda.stream(sb):
        wgrad(dbr2, D, ao, D, D, D, dW[2], ws[1 if concurrent else 0], pb)
    with torch.cuda.stream(sa):
        db_.copy_(da)                                                  # attention backward
    eC = ev(sa)
    if concurrent: sb.wait_event(eC)
    with torch.cuda.stream(sb):
        wgrad(dqkv, 3 * D, u1, D, 3 * D, D, dW[3], ws[1 if concurrent else 0], pb)
    with torch.cuda.stream(sa):
        gemm(EPI_NONE_TE, dqkv, wqt, M, D, 3 * D, du, pa)            # dgrad in_proj
    if concurrent:
        sa.wait_event(ev(sb))                                          # the branch gradients are overwritten from here on
    with torch.cuda.stream(sa):
        cb.copy_(ca)                                                   # LN1 backward

def run(concurrent, blocks=12, reps=5):
    sa = torch.cuda.Stream(); sb = torch.cuda.Stream() if concurrent else sa
    for _ in range(2):
        for _ in range(blocks): block(sa, sb, concurrent)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(sa)
    for _ in range(reps):
        for _ in range(blocks): block(sa, sb, concurrent)
    if concurrent: sa.wait_event((lambda e: (e.record(sb), e)[1])(torch.cuda.Event()))
    e1.record(sa); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print(f"[WGRAD_CUS={os.environ.get('OSUD_WGRAD_CUS', 'all')} DYN={os.environ.get('OSUD_GEMM_DYNAMIC', '0')}] 12 blocks: serial {run(False):.2f} ms, two streams {run(True):.2f} ms", flush=True)
