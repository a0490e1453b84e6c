#!/usr/bin/env python3
"""Headline benchmark of the native DiT path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode sample|train]

N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(one rank per GPU).  Prints ONE JSON line on rank 0, at most 4 KB (`compact_line`: the contract fields, `roofline`,
`cpu_baseline`, a `sampling` summary); the full report goes to bench_detail.json next to this file and to stderr
(DESIGN.md §5 says how each number is obtained).

mode train (BASELINE.json configs[1], the configuration the headline metric is quoted on): DiT-B,
  seq-len 128, per-GPU batch 256 synthetic windows, bf16 MFMA tier, one "step" = everything in the
  reference's train.py:243-261 — randint t, randn noise, label dropout, q_sample, forward, L1+vb loss,
  backward, gradient all-reduce (RCCL, N > 1), AdamW, EMA, re-pack of the bf16 weight copies.  Weak
  scaling: the per-GPU batch is fixed, global batch = 256 * N.
mode both (default): the train line above as the primary metric plus a "sampling" object holding the
  sampling measurement below.
mode sample (BASELINE.json configs[3]): DiT-B, 64 synthetic beatmap windows of 128 tokens doubled
  for classifier-free guidance (batch 128), cfg-scale 4.0, the 1000-step squaredcos schedule.  A
  "step" is one sampling step = forward_with_cfg + the p_sample update; K steps starting at t=999
  are timed, per-step Gaussian noise for the K steps is generated inside the timed region.
  Sampling shards by rows with no collective: every rank runs its own 64 windows (weak scaling).
"""
import argparse
import json
import math
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
# What the matrix pipe sustains on FULL-ENTROPY bf16 operands with nothing else running (tools/probes/mfma_peak.hip: back-to-back
# v_mfma_f32_32x32x16_bf16 from registers; zero operands reach 2340): profiles/r03_mfma_peak_probe.md.  Reported beside `frac`.
MFMA_RANDOM_OPERAND_CEILING_TFLOPS = 1735.0
FLOP_PER_TOKEN_FWD = 176.10e6  # DiT-B, T=128
FLOP_PER_TOKEN_TRAIN = 528.3e6  # forward + backward = 3x forward


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=["both", "sample", "train", "xl"], default=os.environ.get("OSUD_BENCH_MODE", "both"))
    ap.add_argument("--no-xl", action="store_true", help="mode both: skip the DiT-XL seq-len 256 line (BASELINE configs[4] shape)")
    ap.add_argument("--xl-precision", choices=["bf16", "fp8"], default=None, help="tier of the DiT-XL line (default: both, bf16 first)")
    ap.add_argument("--batch", type=int, default=256, help="training windows per GPU")
    ap.add_argument("--sample-steps", type=int, default=None, help="timed sampling steps in mode both (default 1000)")
    ap.add_argument("--precision", choices=["bf16", "fp16", "fp32", "fp8", "bf16x3", "fp16f8", "fp16w8", "fp16m8"], default="bf16")
    ap.add_argument("--maps", type=int, default=64, help="beatmap windows per GPU (CFG doubles the batch)")
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--model", default="DiT-B")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--zero1", action="store_true", help="N > 1: sharded optimizer between reduce-scatter and all-gather instead of all-reduce")
    ap.add_argument("--grad-wire", choices=["fp32", "bf16"], default="fp32", help="--zero1: dtype of the gradients on the wire")
    ap.add_argument("--native-comm", action="store_true", help="N > 1: the timed exchange goes through the library's own RCCL communicator (C ABI)")
    ap.add_argument("--no-schedule-selection", action="store_true", help="N > 1: time the all-reduce schedule as given instead of measuring both exchange schedules first and keeping the faster")
    ap.add_argument("--no-exchange-ab", action="store_true", help="N > 1: skip the in-run A/B of the exchange schedules (multi_gpu.schedules)")
    ap.add_argument("--watchdog-s", type=float, default=900.0, help="N > 1: if the collective diagnostics after the timed region have not returned after this "
                                                                    "many seconds, rank 0 prints the line with what is measured and every rank exits")
    ap.add_argument("--simulate-hang", action="store_true", help=argparse.SUPPRESS)  # (test of the watchdog: the first diagnostic never returns)
    ap.add_argument("--no-compute-floor", action="store_true", help="N = 1: skip the compute floor of the multi-GPU schedule (phased backward, queued tiles, stubbed exchange)")
    ap.add_argument("--no-family-table", action="store_true", help="skip the torch.profiler pass (use under rocprofv3)")
    ap.add_argument("--no-parity-tier", action="store_true", help="skip the fp32 parity-tier throughput and the bf16-vs-fp32 drift run")
    ap.add_argument("--drift-steps", type=int, default=1000, help="length of the CFG-4 loop the bf16 drift is measured on")
    ap.add_argument("--h2d", action="store_true", help="also time the training steps with every batch copied from pinned host "
                                                         "memory inside the step (reported as pcie_inclusive, never as value)")
    return ap.parse_args()


WATCHDOG_EXIT = 3  # exit code of every rank when the watchdog fires (distinct from Python's 1 and from a signal's 128 + n)


class Watchdog:
    """N > 1 only.  Everything after the timed region of a multi-GPU run is collective (the exchange-schedule legs, sharded sampling, the
    DiT-XL line) and none of it has run on more than one GPU before the first such run: a collective that never returns on some node
    must not take the timed result with it.  Armed once the line's contract fields exist; if `main` has not disarmed it by the deadline,
    rank 0 prints the line as it stands -- plus `diagnostics_incomplete` naming the stage -- and every rank leaves with
    os._exit(WATCHDOG_EXIT): a hung collective is a FAILED run to torchrun, the driver and CI (the partial line is there to be read,
    not to be mistaken for a complete one)."""

    def __init__(self):
        self.lock = threading.Lock()
        self.res, self.rank, self.deadline, self.done, self.stage = None, 0, None, False, "after the timed region"

    def arm(self, res, rank, seconds, stage=None):
        self.res, self.rank, self.deadline = res, rank, time.monotonic() + seconds
        self.seconds = seconds
        self.done = False
        if stage is not None:
            self.stage = stage
        threading.Thread(target=self._run, daemon=True).start()

    def disarm(self):
        with self.lock:
            self.done = True

    def _run(self):
        while time.monotonic() < self.deadline:
            time.sleep(0.25)
            if self.done:
                return
        with self.lock:
            if self.done:
                return
            if self.rank == 0:
                out = {k: v for k, v in self.res.items()}
                out["diagnostics_incomplete"] = {"stage": self.stage, "after_s": self.seconds,
                                                 "note": "a diagnostic after the timed region did not return; value / ms_per_step are the completed timed region"}
                emit(out)
            else:
                time.sleep(2.0)  # (rank 0's line first)
            os._exit(WATCHDOG_EXIT)


WATCHDOG = Watchdog()
WATCHDOG_PRE = Watchdog()  # (the schedule selection in FRONT of the timed region is collective too: a hang there must still end the run)


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # OSUD_DIST_BACKEND=gloo + OSUD_SINGLE_DEVICE=1: functional check of the N>1 path on a one-GPU box (all ranks on GPU 0)
        if os.environ.get("OSUD_SINGLE_DEVICE", "0") == "1":
            local = 0
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get("OSUD_DIST_BACKEND", "nccl"))  # nccl = RCCL on ROCm
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    return world, rank, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    if torch.cuda.is_available():  # (the world-8 schema test of multi_gpu_report runs this file's exchange report on CPU tensors over gloo)
        torch.cuda.synchronize()


def max_over_ranks(value, world, dev):
    if world == 1:
        return value
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gemm_roofline(M, N, K, dev, iters=50):
    """Live HIP-event timing of the dominant kernel — the fc1 GEMM (bias + GELU epilogue) at the
    bench shape — on the current stream, random operands."""
    from osu_diffusion_amd import _lib

    L = _lib.lib()
    Yf = torch.randn(M, K, device=dev)
    Xf = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev) * 0.02
    Y = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
    X = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    _lib.check(L.osud_op_convert(0, _lib.ptr(Yf), _lib.ptr(Y), Yf.numel(), _lib.stream_ptr(dev)))
    _lib.check(L.osud_op_convert(0, _lib.ptr(Xf), _lib.ptr(X), Xf.numel(), _lib.stream_ptr(dev)))

    def launch():
        _lib.check(L.osud_op_gemm(0, _lib.EPI_BIAS_GELU_TE, _lib.ptr(Y), K, _lib.ptr(X), K, M, N, K, _lib.ptr(out), N,
                                  _lib.ptr(bias), None, 0, 0, 0, _lib.stream_ptr(dev)))

    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / 1e3 / iters
    flops = 2.0 * M * N * K
    traffic, src = None, None
    try:  # HBM bytes per launch of this kernel from the committed rocprofv3 --pmc passes (not measurable in-process)
        if M == 32768:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")))["fc1_forward"]["hbm_bytes_per_launch"]
            src = "profiles/r06_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, read side x2 per the gfx950 note)"
        else:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_fc1.json")))[f"M={M}"]["hbm_bytes_per_launch"]
            src = "profiles/r02_pmc_fc1.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, read side x2 per the gfx950 note)"
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "gemm_kernel<bf16, EPI_BIAS_GELU_TE> (fc1 %dx%dx%d)" % (M, N, K),
            "achieved": round(flops / sec / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flops / sec / 1e12 / PEAK_BF16_TFLOPS, 4),
            "frac_of_random_operand_ceiling": round(flops / sec / 1e12 / MFMA_RANDOM_OPERAND_CEILING_TFLOPS, 4),
            "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes": 2 * (M * K + N * K + M * N),
            "flop_per_launch": flops, "avg_launch_us": round(sec * 1e6, 2)}


def wgrad_roofline(M, D, dev, iters=8):
    """Live HIP-event timing of the kernel with the largest share of a training step's device time -- `wgrad_kernel`, the
    transpose-free weight-gradient product (csrc/wgrad.hip), with the deterministic combine of its split-K partial slabs
    (`splitk_reduce_kernel`) -- at the four shapes it runs at in every block (in_proj 3D x D, out_proj D x D, fc1 4D x D, fc2
    D x 4D over M tokens), random bf16 operands, on the current stream.  As in the step, the gradient operand of a launch has
    just been written by the kernel in front of it (here: a device copy, outside the timed events), the activation operand
    has not.  Each launch sits between its own pair of events; `avg_launch_us` is the mean over all of them -- the number the
    kernel trace's `wgrad_kernel` + `splitk_reduce_kernel` rows average to (profiles/r03_train_kernel_trace.md)."""
    from osu_diffusion_amd import _lib

    L = _lib.lib()
    shapes = [("in_proj", 3 * D, D), ("out_proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)]
    bufs = []
    for _, Ny, Nx in shapes:
        Psrc = (torch.randn(M, Ny, device=dev) * 0.05).to(torch.bfloat16)
        bufs.append((Psrc, torch.empty_like(Psrc), torch.randn(M, Nx, device=dev).to(torch.bfloat16), torch.empty(Ny, Nx, device=dev)))
    ws = torch.empty(16 * 4 * D * D, device=dev)
    pairs = []
    for it in range(iters + 2):
        for (name, Ny, Nx), (Psrc, P, Q, out) in zip(shapes, bufs):
            P.copy_(Psrc)  # the producer's write (not timed)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.osud_op_wgrad(_lib.ptr(P), Ny, _lib.ptr(Q), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
            e1.record()
            if it >= 2:
                pairs.append((name, e0, e1))
    torch.cuda.synchronize()
    per = {}
    for name, e0, e1 in pairs:
        per.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
    per_us = {k: sum(v) / len(v) for k, v in per.items()}
    avg_us = sum(per_us.values()) / len(per_us)
    flops = sum(2.0 * M * Ny * Nx for _, Ny, Nx in shapes) / len(shapes)  # per launch, averaged like the time
    ach = flops / (avg_us * 1e-6) / 1e12
    traffic, src = None, None
    try:  # HBM bytes per launch (kernel + combine pass, fc1's shape) from the committed rocprofv3 --pmc passes (not measurable in-process)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")))["wgrad_fc1"]
        if pmc["M"] == M and D == 768:
            traffic = pmc["hbm_bytes_per_launch"]
            src = ("profiles/r06_pmc_traffic.json (fc1's shape: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/wgrad_only.py, separate passes, "
                   "read side x2 per the gfx950 note)")
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "wgrad_phased_kernel + splitk_reduce_kernel (a block's four weight gradients, %d tokens, D = %d)" % (M, D),
            "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
            "frac_of_random_operand_ceiling": round(ach / MFMA_RANDOM_OPERAND_CEILING_TFLOPS, 4),
            "random_operand_ceiling": {"value": MFMA_RANDOM_OPERAND_CEILING_TFLOPS, "unit": "TFLOP/s",
                                       "source": "profiles/r03_mfma_peak_probe.md: back-to-back bf16 MFMAs from registers on random operands "
                                                 "(zero operands: 2340); the chip's power management paces the pipe by operand content"},
            "traffic": traffic, "traffic_source": src,
            "algorithmic_bytes": (2 * M * (4 * D + D) + 4 * 4 * D * D) if traffic else None,
            "flop_per_launch": flops, "avg_launch_us": round(avg_us, 2), "per_shape_us": {k: round(v, 2) for k, v in per_us.items()}}


def train_roofline(wg_alone, fc1_alone, pf):
    """The bench line's `roofline` of a training run.  Top level = the kernel with the largest share of the step's device time (the
    weight-gradient kernel with the fixed-order combine of its partial slabs) at its IN-STEP duration: the mean device duration of its
    launches inside training steps of this very run (torch.profiler / roctracer over steps behind the timed region, single-stream
    schedule) -- the figure a `rocprofv3 --kernel-trace` of the same command reproduces (profiles/r05_train_kernel_trace.md: FLOP per
    launch / (avg of the wgrad kernel + avg of splitk_reduce_kernel)).  The stand-alone, HIP-event-timed launch on cold operands that
    earlier rounds reported here stays as `stand_alone`.  `lowest_fraction_top_family` names the family with the largest distance to its
    roofline among those with >= 10 % of the step (the forward GEMMs), with its heaviest kernel (fc1 + GELU) at its in-step duration."""
    out = dict(wg_alone)
    stand = {k: out.pop(k) for k in ("achieved", "frac", "frac_of_random_operand_ceiling", "avg_launch_us", "per_shape_us") if k in out}
    stand["what"] = "each launch alone between its own pair of HIP events, operands not cache-warm (bench.py: wgrad_roofline)"
    fc1 = dict(fc1_alone)
    fc1_stand = {k: fc1.pop(k) for k in ("achieved", "frac", "frac_of_random_operand_ceiling", "avg_launch_us") if k in fc1}
    ok = isinstance(pf, dict) and "mfma_kernels" in pf
    wg_us, wg_name = in_step_kernel(pf, r"^wgrad(_phased)?_kernel") if ok else (None, None)
    sk_us, _ = in_step_kernel(pf, r"splitk_reduce") if ok else (None, None)
    if wg_us:
        us = wg_us + (sk_us or 0.0)
        ach = out["flop_per_launch"] / (us * 1e-6) / 1e12
        out.update(achieved=round(ach, 2), frac=round(ach / PEAK_BF16_TFLOPS, 4), frac_of_random_operand_ceiling=round(ach / MFMA_RANDOM_OPERAND_CEILING_TFLOPS, 4),
                   avg_launch_us=round(us, 2), avg_launch_us_parts={wg_name: wg_us, "splitk_reduce_kernel": sk_us},
                   measured="in-step: mean device duration of the kernel's launches inside training steps of this run (torch.profiler / roctracer, "
                            "8 steps behind the timed region, single-stream schedule)")
    else:
        out.update(stand, measured="stand-alone launches (no per-kernel table in this run: --no-family-table)")
    out["stand_alone"] = stand
    f1_us, f1_name = in_step_kernel(pf, r"^gemm(_phased)?_kernel<bf16, 4,") if ok else (None, None)
    if f1_us:
        a = fc1["flop_per_launch"] / (f1_us * 1e-6) / 1e12
        fc1.update(achieved=round(a, 2), frac=round(a / PEAK_BF16_TFLOPS, 4), avg_launch_us=f1_us, kernel_in_step=f1_name,
                   measured="in-step (as the top level; the launch also writes the saved GELU derivative: 201 MB more than the stand-alone launch)")
    else:
        fc1.update(fc1_stand, measured="stand-alone")
    fc1["stand_alone"] = fc1_stand
    out["fc1_forward"] = fc1
    if ok:
        big = {k: v for k, v in pf["families"].items() if v.get("bound") == "mfma" and v.get("share", 0) >= 0.10}
        if big:
            fam = min(big, key=lambda k: big[k]["frac"])
            out["lowest_fraction_top_family"] = {"family": fam, "ms_per_step": big[fam]["ms_per_step"], "share": big[fam]["share"], "frac": big[fam]["frac"],
                                                 "heaviest_kernel": fc1.get("kernel_in_step") if fam == "gemm_fwd" else big[fam]["top_kernel"],
                                                 "heaviest_kernel_frac": fc1.get("frac") if fam == "gemm_fwd" else None}
    if pf is not None:
        out["per_family"] = pf
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# Per-family roofline table.  A few steps AFTER the timed region are run under torch.profiler (roctracer sees every kernel
# the process launches, libosud's included); kernel device times are summed per family and set against the family's
# algorithmic work (2*M*N*K per Linear, 4*T*D per token and block for the attention core; elementwise work is not
# counted) or, for the HBM-bound families, its algorithmic bytes.
PEAK_HBM_GBS = 8000.0  # spec; ~6300 GB/s is what a float4 copy reaches (MI355X_MICROARCH.md)

_EPI_FWD = {"0", "1", "2", "3", "4", "5"}  # csrc/gemm.h: bias / silu / gelu / gated-residual epilogues = forward Linears


def _family_of(name):
    import re

    if "gemm_kernel<" in name or "gemm_phased_kernel<" in name:
        m = re.search(r"gemm(?:_phased)?_kernel<[^,]+,\s*\(?(?:osud::)?(?:GemmEpilogue\)?)?\s*(\d+)", name)
        epi = m.group(1) if m else "?"
        if epi in _EPI_FWD:
            return "gemm_fwd"
        return "gemm_dgrad" if epi in ("7", "9") else "gemm_other"
    if "wgrad_kernel" in name or "wgrad_phased_kernel" in name:
        return "gemm_wgrad"
    if "attn_bwd" in name:
        return "attention_bwd"
    if "attn_" in name:
        return "attention_fwd"
    if "adamw_ema" in name:
        return "hbm_adamw_ema"
    if "ln_mod_bwd" in name or "gate_bwd" in name or "final_bwd" in name:
        return "hbm_layernorm_bwd"
    if "ln_mod" in name or "final_kernel" in name:
        return "hbm_layernorm_fwd"
    if any(k in name for k in ("splitk_reduce", "colsum", "seg_kernel", "transpose", "convert_kernel", "pack_rows", "unpad_rows",
                               "mask_rows", "fillBuffer", "copyBuffer")):
        return "hbm_housekeeping"
    return "other"


def _short_kernel(name):
    import re

    name = re.sub(r"\(anonymous namespace\)::|osud::|^void ", "", name).replace("unsigned short", "bf16")
    return re.sub(r"\(.*$", "", name)[:80]


def in_step_kernel(pf, pattern):
    """(avg_us, name) of the kernel of the family table's per-kernel rows whose short name matches `pattern` with the most device time."""
    import re

    best = None
    for k, v in (pf.get("mfma_kernels") or {}).items():
        if re.search(pattern, k):
            t = v["avg_us"] * v["calls_per_step"]
            if best is None or t > best[2]:
                best = (v["avg_us"], k, t)
    return (best[0], best[1]) if best else (None, None)


def family_table(run_steps, n_steps, work, dev):
    """run_steps(n) executes n steps; work = {family: (amount per step, "flop" | "byte")}.  Returns the table or an
    {"error": ...} stub (the profiler is best effort: the headline numbers never depend on it)."""
    try:
        from torch.profiler import ProfilerActivity, profile

        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            run_steps(n_steps)
            torch.cuda.synchronize()
        fam, top, kern = {}, {}, {}
        for ev in prof.key_averages():
            t_us = getattr(ev, "device_time_total", None)
            if t_us is None:
                t_us = getattr(ev, "cuda_time_total", 0.0)
            if not t_us or ev.key.startswith("Memcpy") or ev.key.startswith("Memset"):
                continue
            f = _family_of(ev.key)
            fam[f] = fam.get(f, 0.0) + t_us
            if f.startswith("gemm") or "splitk_reduce" in ev.key:  # per-kernel rows of the MFMA kernels (and the split-K combine): `roofline` is made from them
                kern[_short_kernel(ev.key)] = {"family": f, "calls_per_step": round(ev.count / n_steps, 2), "avg_us": round(t_us / max(1, ev.count), 2)}
            if t_us > top.get(f, ("", 0.0))[1]:
                top[f] = (ev.key[:96], t_us)
        if not fam:
            return {"error": "the profiler returned no kernel records"}
        total = sum(fam.values())
        rows = {}
        for f, t_us in sorted(fam.items(), key=lambda kv: -kv[1]):
            ms = t_us / n_steps / 1e3
            row = {"ms_per_step": round(ms, 4), "share": round(t_us / total, 4), "top_kernel": top[f][0]}
            if f in work and ms > 0:
                amount, kind = work[f]
                if kind == "flop":
                    ach = amount / (ms * 1e-3) / 1e12
                    row.update(bound="mfma", achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4))
                else:
                    ach = amount / (ms * 1e-3) / 1e9
                    row.update(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4))
            rows[f] = row
        return {"source": f"torch.profiler (roctracer) over {n_steps} steps after the timed region", "kernel_ms_per_step": round(total / n_steps / 1e3, 4),
                "mfma_ms_per_step": round(sum(v for k, v in fam.items() if k.startswith("gemm") or k.startswith("attention")) / n_steps / 1e3, 4),
                "non_mfma_ms_per_step": round(sum(v for k, v in fam.items() if not (k.startswith("gemm") or k.startswith("attention"))) / n_steps / 1e3, 4),
                "top_family": next(iter(rows)), "families": rows, "mfma_kernels": kern}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def dit_work(D, L, M, T, Kfirst=528, training=True, n_params=None):
    """Algorithmic work per step of a DiT with hidden D, L blocks, M tokens of T-token windows."""
    lin = L * 24.0 * M * D * D + 2.0 * M * Kfirst * D           # qkv + proj + fc1 + fc2 per block, + the first linear
    # (the attention core at T = 128 is HBM-bound -- 0.39 of the 14.55 MFLOP per token and block -- and is priced by its bytes)
    w = {"gemm_fwd": (lin, "flop"), "attention_fwd": (L * 8.0 * M * D, "byte"),   # reads Q|K|V (3 x 2 B), writes O (2 B) per element
         # LN+modulate forward: read h (4 B) write u (2 B) per element and launch, 2 launches per block (+ training: the branch in,
         # the updated residual out)
         "hbm_layernorm_fwd": (L * 2.0 * M * D * ((4 + 2 + 4 + 2) if training else (4 + 2)), "byte")}
    if training:
        w.update({"gemm_dgrad": (lin - 2.0 * M * Kfirst * D, "flop"), "gemm_wgrad": (lin, "flop"),
                  "attention_bwd": (L * 16.0 * M * D, "byte"),                      # reads Q|K|V, dO, O; writes dQ|dK|dV
                  # LN backward + the gate step riding in it: h, du, dh_skip, br in; dh, dbr out
                  "hbm_layernorm_bwd": (L * 2.0 * M * D * (4 + 2 + 4 + 2 + 4 + 2), "byte")})
        if n_params:
            w["hbm_adamw_ema"] = (9.0 * 4 * n_params, "byte")
    return w


def cpu_baseline_sample(model, args, windows, steps=2):
    """Oracle (CPU restatement of the reference) timed on this host's cores on a bounded sample:
    `steps` p_sample steps at the SAME shapes (batch 128 x 128 tokens, DiT-B, cfg 4)."""
    from oracle import diffusion_oracle as do
    from oracle import dit_oracle as mo

    (x, o, c), y = windows
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    shape = mo.shape_of(args.model, num_classes=model.y_embedder.num_classes)
    sch = do.create_schedule("1000", "squaredcos_cap_v2")
    cores = min(32, os.cpu_count() or 1)  # measured on the GPU host: 32 threads is the fastest setting (tools/cpu_threads_probe.py)
    torch.set_num_threads(cores)
    xx = torch.randn(x.shape)
    t = torch.full((x.shape[0],), 999, dtype=torch.long)
    tmap = torch.from_numpy(sch.timestep_map)
    with torch.no_grad():
        mo.forward_with_cfg(sd, shape, xx[:4], tmap[t[:4]], o[:4], c[:4], y[:4], 4.0)  # warm the allocator
        t0 = time.perf_counter()
        for k in range(steps):
            out = mo.forward_with_cfg(sd, shape, xx, tmap[t - k], o, c, y, 4.0)
            xx = do.p_sample_step(sch, out, xx, t - k, torch.randn(x.shape))["sample"]
        dt = time.perf_counter() - t0
    return {"value": round(steps / dt, 5), "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} p_sample steps (t=999..) of the same workload, fp32 torch-CPU oracle, {dt:.1f} s"}


def cpu_baseline_train(model, args, batch, steps=1, cpu_batch=32):
    """Oracle training step (fp32 torch-CPU restatement + autograd + torch AdamW + EMA) on a bounded
    sample: `steps` steps at batch `cpu_batch` x seq_len (the GPU step uses batch 256)."""
    from oracle import diffusion_oracle as do
    from oracle import dit_oracle as mo

    (x, o, c), y = batch
    x, o, c, y = x[:cpu_batch].cpu(), o[:cpu_batch].cpu(), c[:cpu_batch].cpu(), y[:cpu_batch].cpu()
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    shape = mo.shape_of(args.model, num_classes=model.y_embedder.num_classes)
    sd = {k: v.detach().float().cpu().clone().requires_grad_(k != "xoc_embedder.playfield_size")
          for k, v in model.state_dict().items()}
    ema = {k: v.detach().clone() for k, v in sd.items()}
    opt = torch.optim.AdamW([v for v in sd.values() if v.requires_grad], lr=1e-4, weight_decay=0)
    sch = do.create_schedule("", "squaredcos_cap_v2")
    T = x.shape[2]
    t0 = time.perf_counter()
    for _ in range(steps):
        t = torch.randint(0, 1000, (cpu_batch,))
        drop = torch.rand(cpu_batch) < 0.2
        fn = lambda xx, tt: mo.forward(sd, shape, xx, tt, o, c, y, drop_mask=drop)  # noqa: E731
        terms = do.training_losses(sch, fn, x, t, torch.randn_like(x), loss="l1")
        terms["loss"].mean().backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            for k in ema:
                ema[k].mul_(0.9999).add_(sd[k].detach(), alpha=1e-4)
    dt = time.perf_counter() - t0
    return {"value": round(steps * cpu_batch * T / dt, 2), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{steps} full training step(s) at batch {cpu_batch} x {T} tokens (GPU: batch {len(batch[1])}), fp32 "
                      f"torch-CPU oracle + autograd + AdamW + EMA, {dt:.1f} s"}


def multi_gpu_report(args, world, rank, dev, model, diffusion, batches, timed_ms, timed_name, trainer_factory=None):
    """What the first multi-GPU run needs to be read (no such node was available while this was written): did RCCL see every rank,
    how much of the step is exposed communication, how many bytes travel, and which exchange schedule is fastest -- measured in this
    very run, every schedule on the same model, batches and boxes:
      allreduce        per-slice async all-reduces under the phased backward (the default; reference: DDP's bucketed reducer, train.py:152,257)
      zero1            reduce-scatter -> AdamW / EMA on the own 1 / world shard -> all-gather of the masters under the next forward
      zero1_no_overlap the same, all-gathers joined before the next step
      native_comm      the all-reduce schedule through the library's own RCCL communicator (C ABI: osud_allreduce_grads)
      no_exchange      the all-reduce schedule with every collective left out (NOT a training run: the compute floor)
    exposed_comm_ms_per_step = allreduce - no_exchange.
    `trainer_factory` (default: NativeTrainer) is what a leg builds its trainer with: tests/test_distributed_cpu.py runs this report at
    world size 8 over gloo on CPU tensors with a trainer whose compute is a sleep and whose exchange is the real slice schedule."""
    import torch.distributed as dist

    from osu_diffusion_amd.training import overlap_slices

    on_gpu = torch.device(dev).type == "cuda"
    if trainer_factory is None:
        from osu_diffusion_amd.training import NativeTrainer as trainer_factory

    if args.simulate_hang:
        time.sleep(1e9)
    backend = dist.get_backend()
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    rccl_version = None
    if on_gpu:
        from osu_diffusion_amd import _lib

        rccl_version = int(_lib.lib().osud_comm_rccl_version())
    rep = {"rccl": {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "world_size": world, "ranks_seen": int(round(float(ones.item()))),
                    "version": rccl_version,
                    "torch_nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None}}
    B, T = args.batch, args.seq_len
    K, W = (6, 2) if args.precision == "fp32" else (max(4, min(args.steps or 20, 20)), 3)

    def leg(**kw):
        tr = trainer_factory(model, diffusion, lr=1e-4, **kw)
        for i in range(W):
            (x, o, c), y = batches[i % 4]
            tr.step(x, o, c, y)
        barrier(world)
        t0 = time.perf_counter()
        for i in range(K):
            (x, o, c), y = batches[i % 4]
            tr.step(x, o, c, y)
        tr.finish_exchange()
        barrier(world)
        ms = max_over_ranks(time.perf_counter() - t0, world, dev) / K * 1e3
        del tr
        if on_gpu:
            torch.cuda.empty_cache()
        return round(ms, 3)

    legs = {"allreduce": dict(), "zero1": dict(shard_optimizer=True), "zero1_no_overlap": dict(shard_optimizer=True, overlap_gather=False),
            "native_comm": dict(native_comm=True), "no_exchange": dict(stub_exchange=True)}
    sched = {}
    for name, kw in legs.items():
        if name == "native_comm" and backend != "nccl":  # (the one-GPU rehearsal over gloo: two ranks of ONE device cannot form an RCCL communicator)
            sched[name] = {"skipped": "needs one GPU per rank (backend nccl)"}
            continue
        WATCHDOG.stage = f"multi_gpu.schedules.{name}"
        try:
            sched[name] = {"ms_per_step": leg(**kw)}
        except Exception as e:  # a schedule that does not run here (e.g. no librccl for the native communicator) must not sink the line
            sched[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    for name, v in sched.items():
        if "ms_per_step" in v:
            v["tokens_per_s"] = round(world * B * T / v["ms_per_step"] * 1e3, 1)
    rep["schedules"] = sched
    rep["schedules_measured_over"] = {"steps": K, "warmup": W, "note": "shorter than the timed region: compare the legs with each other, not with `value`"}
    ok = {n: v["ms_per_step"] for n, v in sched.items() if "ms_per_step" in v and n != "no_exchange"}
    if ok:
        rep["fastest_schedule"] = min(ok, key=ok.get)
    if "ms_per_step" in sched.get("allreduce", {}) and "ms_per_step" in sched.get("no_exchange", {}):
        rep["exposed_comm_ms_per_step"] = round(sched["allreduce"]["ms_per_step"] - sched["no_exchange"]["ms_per_step"], 3)
    rep["timed_schedule"] = {"name": timed_name, "ms_per_step": round(timed_ms, 3)}
    # bytes one rank hands to the exchange per step, and what a ring moves over its links for them
    arena = model._arena
    blocks, tail = overlap_slices(arena, model.depth)
    dense = sum(hi - lo for _, _, lo, hi in blocks) + sum(hi - lo for kind, _, lo, hi in tail if kind in ("tail", "final"))
    D = model.hidden_size
    rows_bytes = world * B * (D * 4 + 8)  # all-gathered label rows + indices
    rep["wire_bytes_per_step"] = {"dense_slices_payload": int(dense * 4), "class_table_rows_allgather": int(rows_bytes),
                                  "class_table_dense_would_be": int(next(hi - lo for kind, _, lo, hi in tail if kind == "table") * 4),
                                  "ring_bytes_sent_per_gpu": int(2 * (world - 1) / world * dense * 4 + (world - 1) / world * rows_bytes),
                                  "slices": len(blocks) + 1 + sum(1 for kind, *_ in tail if kind == "tail")}
    rep["predicted_comm_ms_per_step"] = predicted_comm_ms(dense * 4, rows_bytes, world)
    return rep


XGMI_LINK_GBS = 153.0  # per link and direction; 7 links per GPU, point to point (MI355X_MICROARCH.md / the task's hardware sheet)


def predicted_comm_ms(dense_bytes, rows_bytes, world):
    """What the gradient exchange of one step costs ON THE WIRE if nothing overlaps, from the payload and the xGMI link rate -- the number
    the first 8-GPU run's `exposed_comm_ms_per_step` is to be judged against (DESIGN.md section 6 has the derivation):
      ring all-reduce          2 (W - 1) / W x payload over ONE link per direction (a ring uses one outgoing link per GPU)
      direct (mesh) exchange   reduce-scatter + all-gather with every peer at once: each of the W - 1 links carries payload / W per phase,
                               two phases -> 2 x payload / W per link
    With the phased backward the exchange of all but the last slice runs under compute: exposed = what is left behind the first block's
    backward (the last slice to be reduced), so both figures are upper bounds of `exposed_comm_ms_per_step`."""
    if world < 2:
        return None
    link = XGMI_LINK_GBS * 1e9
    ring = (2.0 * (world - 1) / world * dense_bytes + (world - 1) / world * rows_bytes) / link * 1e3
    mesh = (2.0 * dense_bytes / world + rows_bytes / world) / link * 1e3
    return {"link_GBps": XGMI_LINK_GBS, "ring_allreduce_unoverlapped": round(ring, 3), "mesh_reduce_scatter_allgather_unoverlapped": round(mesh, 3),
            "note": "upper bounds of exposed_comm_ms_per_step: the phased backward hides every slice but the last under compute"}


def bench_train(args, world, rank, dev):
    from osu_diffusion_amd.diffusion import create_diffusion
    from osu_diffusion_amd.models import DiT_models
    from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows
    from osu_diffusion_amd.training import NativeTrainer

    K = args.steps if args.steps is not None else (50 if args.precision != "fp32" else 4)
    W = args.warmup if args.warmup is not None else (10 if args.precision != "fp32" else 1)
    num_classes = 52670
    seed = 0 * world + rank  # train.py:113: global_seed * world_size + rank
    torch.manual_seed(seed)
    model = DiT_models[args.model](num_classes=num_classes, context_size=19 - 3 + 128, class_dropout_prob=0.2,
                                   precision=args.precision)
    model = randomize_zero_init(model.to(dev), seed=0).train()  # train(): label dropout on (train.py:199)
    diffusion = create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True)  # train.py defaults
    B, T = args.batch, args.seq_len
    batches = []
    for i in range(4):  # a few distinct resident batches, cycled
        (x, o, c), y = synthetic_windows(B, T, num_classes, seed=10_000 * rank + i, train_offsets=True)
        batches.append(((x.to(dev), o.to(dev), c.to(dev)), y.to(dev)))
    # N > 1: the exchange schedule of the timed region is MEASURED first (3 steps each of the all-reduce and the sharded-optimizer form, the
    # faster one is kept on every rank) unless a flag names it: the first multi-GPU line must not depend on RCCL's own algorithm choice
    selected = None
    if world > 1 and not args.zero1 and not args.native_comm and not args.no_schedule_selection:
        from osu_diffusion_amd.training import select_exchange_schedule
        WATCHDOG_PRE.arm({"metric": f"{args.model} seq{args.seq_len} train tokens/sec", "value": None, "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None},
                         rank, args.watchdog_s, stage="schedule selection (before the timed region)")
        selected = select_exchange_schedule(lambda shard_optimizer: NativeTrainer(model, diffusion, lr=1e-4, shard_optimizer=shard_optimizer),
                                            batches, steps=3, warmup=1, device=dev)
        WATCHDOG_PRE.disarm()
        args.zero1 = selected["name"] == "zero1"
    trainer = NativeTrainer(model, diffusion, lr=1e-4, shard_optimizer=args.zero1,
                            wire_dtype=torch.bfloat16 if args.grad_wire == "bf16" else None, native_comm=args.native_comm)
    terms = None
    for i in range(W):
        (x, o, c), y = batches[i % 4]
        terms = trainer.step(x, o, c, y)
    barrier(world)
    t0 = time.perf_counter()
    for i in range(K):
        (x, o, c), y = batches[i % 4]
        terms = trainer.step(x, o, c, y)
    trainer.finish_exchange()  # (sharded optimizer: the last step's master all-gather belongs to the timed region)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    loss = float(terms[2].mean())
    assert math.isfinite(loss), "non-finite training loss"
    tokens_per_s = world * B * T * K / dt
    res = {
        "metric": f"{args.model} seq{args.seq_len} train tokens/sec (whole job; per-GPU = value / n_gpus)", "value": round(tokens_per_s, 1),
        "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp32": "f32", "fp8": "fp8 (e4m3 operands, delayed per-tensor scaling: in_proj / out_proj / fc1 / fc2 forward, data-gradient and weight-gradient products) + bf16 (attention, adaLN, embedders)"}[args.precision], "data": "synthetic",
        "config": {"workload": f"train.py step: {args.model} seq-len {T}, per-GPU batch {B} synthetic windows (global {B * world}), "
                               f"L1+vb loss, AdamW lr 1e-4, EMA 0.9999, label dropout 0.2, squaredcos_cap_v2 1000 steps",
                   "per_gpu_batch": B, "global_batch": B * world, "seq_len": T,
                   "parallelism": (f"dp{world}: flat fp32 gradient arena, " + (f"per-slice RCCL reduce-scatter ({args.grad_wire} wire) -> AdamW/EMA on the own "
                                                                                 f"1/{world} shard -> all-gather of the masters" if trainer.shard_optimizer else
                                                                                 "per-slice RCCL all-reduces") + " overlapped with the phased backward") if world > 1 else "single GPU",
                   "last_loss": round(loss, 4)},
        "per_gpu_tokens_per_s": round(tokens_per_s / world, 1),
    }
    if args.model == "DiT-B" and T == 128:
        per_gpu = tokens_per_s / world
        res["end_to_end"] = {"flop_per_token": FLOP_PER_TOKEN_TRAIN, "achieved_tflops_per_gpu": round(per_gpu * FLOP_PER_TOKEN_TRAIN / 1e12, 1),
                             "mfma_frac": round(per_gpu * FLOP_PER_TOKEN_TRAIN / 1e12 / PEAK_BF16_TFLOPS, 4)}
    if world > 1:
        WATCHDOG.arm(res, rank, args.watchdog_s)
    if args.h2d:  # the loader's hand-over (train.py:244-248): pinned host batch -> device, every step
        host = [tuple(v.cpu().pin_memory() for v in (x, o, c, y)) for (x, o, c), y in batches]
        barrier(world)
        t0 = time.perf_counter()
        for i in range(K):
            x, o, c, y = (v.to(dev, non_blocking=True) for v in host[i % 4])
            trainer.step(x, o, c, y)
        barrier(world)
        dt_h = max_over_ranks(time.perf_counter() - t0, world, dev)
        res["pcie_inclusive"] = {"value": round(world * B * T * K / dt_h, 1), "unit": "tokens/s", "ms_per_step": round(dt_h / K * 1e3, 3),
                                 "bytes_per_step": sum(v.numel() * v.element_size() for v in host[0])}
    if rank == 0 and not args.no_roofline and args.precision == "bf16":
        D = model.hidden_size
        # The dominant kernel of a training step by device time is the weight-gradient kernel (19 % of the step): it is
        # the top-level `roofline`; the forward fc1 GEMM -- the heaviest single forward launch, round 1's and 2's entry -- stays
        # next to it.  Both are timed live with HIP events at the step's shapes.
        fc1_alone = gemm_roofline(B * T, 4 * D, D, dev)
        wg_alone = wgrad_roofline(B * T, D, dev)
        pf = None
        if not args.no_family_table and world == 1:
            def more(n):
                for i in range(n):
                    (x, o, c), y = batches[i % 4]
                    trainer.step(x, o, c, y)
            # (the timed region runs a block's weight gradients on a side stream next to its data-gradient chain; kernels that
            #  run side by side report stretched durations, so the per-kernel table is taken on the single-stream schedule)
            from osu_diffusion_amd import _lib as _l
            with _l.option("wgrad_side_stream", 0):
                more(3)  # (the schedule has just changed: let the clocks settle before the profiled steps)
                pf = family_table(more, 8, dit_work(D, model.depth, B * T, T, training=True, n_params=trainer.arena.total), dev)
            if isinstance(pf, dict) and "source" in pf:
                pf["source"] += ("; single-stream schedule (option wgrad_side_stream = 0): the timed region overlaps a block's weight "
                                 "gradients with its data-gradient chain on a second stream, which stretches the durations of "
                                 "kernels that run side by side")
        res["roofline"] = train_roofline(wg_alone, fc1_alone, pf)
    if world == 1 and args.precision == "bf16" and not args.no_compute_floor:
        # What the data-parallel schedule costs in compute alone, on this one GPU: the phased backward (a slice per block, as under the
        # exchange), early AdamW on finished slices, GEMM tiles / weight-gradient K-chunks / attention heads drawn from the ticket queues
        # (osud_set_gemm_dynamic_tiles(1), what a trainer switches on when world > 1), every collective left out.  To be read against
        # ms_per_step: the first multi-GPU run starts from THIS compute floor, not from the single-GPU step.
        from osu_diffusion_amd import _lib as _l
        del trainer
        torch.cuda.empty_cache()
        _l.check(_l.lib().osud_set_gemm_dynamic_tiles(1))
        try:
            tr2 = NativeTrainer(model, diffusion, lr=1e-4, force_phased=True, stub_exchange=True)
            for i in range(min(W, 5) or 1):
                (x, o, c), y = batches[i % 4]
                tr2.step(x, o, c, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                (x, o, c), y = batches[i % 4]
                tr2.step(x, o, c, y)
            tr2.finish_exchange()
            torch.cuda.synchronize()
            floor_ms = (time.perf_counter() - t0) / K * 1e3
            res["multi_gpu_schedule_compute_floor_ms"] = round(floor_ms, 3)
            res["multi_gpu_schedule_compute_floor"] = {
                "ms_per_step": round(floor_ms, 3), "vs_single_gpu_step": round(floor_ms / (dt / K * 1e3), 4), "steps": K,
                "what": "phased backward + early AdamW + ticket-queued tiles / K-chunks / heads (dynamic tiles on), collectives stubbed, one GPU"}
            del tr2
        except Exception as e:  # noqa: BLE001  (a diagnostic leg must not cost the line its headline)
            res["multi_gpu_schedule_compute_floor"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        finally:
            _l.check(_l.lib().osud_set_gemm_dynamic_tiles(-1))
        trainer = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_train(model, args, batches[0])
    if world > 1 and args.no_exchange_ab and selected is not None:
        res["multi_gpu"] = {"schedule_selected": selected}
    if world > 1 and not args.no_exchange_ab:  # (every rank takes part: the legs are collective)
        timed_name = ("zero1" if args.zero1 else "allreduce") + (" + native_comm" if args.native_comm else "")
        del trainer
        trainer = None
        torch.cuda.empty_cache()
        res["multi_gpu"] = multi_gpu_report(args, world, rank, dev, model, diffusion, batches, dt / K * 1e3, timed_name)
        if selected is not None:
            res["multi_gpu"]["schedule_selected"] = selected
    del trainer, model
    torch.cuda.empty_cache()
    from osu_diffusion_amd import _lib
    _lib.check(_lib.lib().osud_set_gemm_dynamic_tiles(-1))  # the data-parallel trainer queues GEMM tiles; sampling has no collectives
    return res


def bench_sample(args, world, rank, dev):
    from osu_diffusion_amd.diffusion import create_diffusion
    from osu_diffusion_amd.models import DiT_models
    from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows

    K = args.steps if args.steps is not None else 1000
    W = args.warmup if args.warmup is not None else 50
    assert 1 <= K <= 1000 and 0 <= W <= 1000
    num_classes = 52670
    torch.manual_seed(rank)
    model = DiT_models[args.model](num_classes=num_classes, context_size=19 - 3 + 128, precision=args.precision)
    model = randomize_zero_init(model.to(dev), seed=0).eval()
    n, T = args.maps, args.seq_len
    (x, o, c), y = synthetic_windows(n, T, num_classes, seed=1000 + rank, train_offsets=False)
    x, o, c = torch.cat([x, x]), torch.cat([o, o]), torch.cat([c, c])
    y = torch.cat([y, torch.full_like(y, num_classes)])  # uncond half uses the null class (sample.py:106-107)
    windows = ((x, o, c), y)
    diffusion = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
    kw = dict(o=o.to(dev), c=c.to(dev), y=y.to(dev), cfg_scale=4.0, attn_mask=None)
    model.reserve(2 * n, T)
    z = torch.randn(2 * n, 2, T, device=dev)

    def run(k_steps, state):
        noise = torch.randn(k_steps, *state.shape, device=dev)  # per-step noise, generated in the timed region
        diffusion.run_steps(model.forward_with_cfg, state, kw, first_step=999, last_step=999 - k_steps + 1,
                            step_noise=noise)

    if W:
        run(W, z.clone())
    state = z.clone()
    barrier(world)
    t0 = time.perf_counter()
    run(K, state)
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    assert torch.isfinite(state).all(), "non-finite samples"

    steps_per_s = K / dt
    M = 2 * n * T
    flop_step = M * FLOP_PER_TOKEN_FWD if args.model == "DiT-B" and T == 128 else None
    res = {
        "metric": f"1000-step CFG sample steps/sec ({args.model} seq{args.seq_len})", "value": round(world * steps_per_s, 3),
        "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp32": "f32", "fp8": "fp8(e4m3 GEMM operands)+bf16", "bf16x3": "bf16x3(split-bf16 operands, 3 MFMAs per product)",
                  "fp16f8": "fp16f8(split-bf16 tier with the four big GEMMs on fp16 + e4m3-residual operands)",
                  "fp16w8": "fp16w8(fp16f8 with fp16 activation operands: only the weights carry the e4m3 residual)",
                  "fp16m8": "fp16m8(fp16w8's operand form for in_proj / out_proj / fc2, fp16f8's for fc1)",
                  "fp16": "fp16(IEEE half MFMA operands: the 11 significand bits of the reference's TF32 sampling matmuls)"}[args.precision], "data": "synthetic",
        "config": {"workload": f"sample.py path: {args.model} seq-len {T}, {n} synthetic beatmap windows x2 (CFG) per GPU, "
                               f"cfg-scale 4.0, 1000-step squaredcos schedule, steps t=999..{999 - K + 1}",
                   "rows_per_gpu": 2 * n, "seq_len": T, "sharding": "rows per rank, no collective"},
    }
    if flop_step:
        res["end_to_end"] = {"flop_per_step": flop_step, "achieved_tflops": round(steps_per_s * flop_step / 1e12, 1),
                             "mfma_frac": round(steps_per_s * flop_step / 1e12 / PEAK_BF16_TFLOPS, 4)}
    if rank == 0 and not args.no_roofline and args.precision == "bf16":
        D = model.hidden_size
        alone = gemm_roofline(M, 4 * D, D, dev)
        pf = family_table(lambda k: run(k, z.clone()), 20, dit_work(D, model.depth, M, T, training=False), dev) if (not args.no_family_table and world == 1) else None
        rf = dict(alone)
        stand = {k: rf.pop(k) for k in ("achieved", "frac", "frac_of_random_operand_ceiling", "avg_launch_us") if k in rf}
        us, kname = in_step_kernel(pf, r"^gemm(_phased)?_kernel<bf16, 4,") if isinstance(pf, dict) and "mfma_kernels" in pf else (None, None)
        if us:
            a = rf["flop_per_launch"] / (us * 1e-6) / 1e12
            rf.update(achieved=round(a, 2), frac=round(a / PEAK_BF16_TFLOPS, 4), frac_of_random_operand_ceiling=round(a / MFMA_RANDOM_OPERAND_CEILING_TFLOPS, 4),
                      avg_launch_us=us, kernel_in_step=kname,
                      measured="in-step: mean device duration of the kernel's launches inside sampler steps of this run (torch.profiler / roctracer, 20 steps behind the timed region)")
        else:
            rf.update(stand, measured="stand-alone launches (no per-kernel table in this run)")
        rf["stand_alone"] = stand
        if pf is not None:
            rf["per_family"] = pf
        res["roofline"] = rf
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_sample(model, args, windows)
    return res


def parity_tier_and_drift(args, dev):
    """(1) Throughput of the parity tier (precision="fp32": exact-f32 MFMA v_mfma_f32_32x32x2_f32, peak 157.3 TFLOP/s) -- the
    tier that carries the 1e-3 end-to-end claim -- on the same two workloads.  (2) How far the benchmarked bf16 tier ends from
    it: the SAME window batch, initial noise and per-step noise through the full CFG-4 p_sample loop in both tiers; the deviation
    of the final normalised (x, y) of the conditional rows is reported (max, 99th percentile, mean)."""
    from osu_diffusion_amd.diffusion import create_diffusion
    from osu_diffusion_amd.models import DiT_models
    from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows

    out = {}
    # (1a) training tokens/s
    targs = argparse.Namespace(**vars(args))
    targs.precision, targs.steps, targs.warmup, targs.no_roofline, targs.no_cpu_baseline, targs.h2d = "fp32", 4, 1, True, True, False
    targs.no_compute_floor = True
    tr = bench_train(targs, 1, 0, dev)
    out["train"] = {"value": tr["value"], "unit": "tokens/s", "ms_per_step": tr["ms_per_step"], "steps": tr["steps"],
                    "mfma_frac_of_f32_peak": round(tr["value"] * FLOP_PER_TOKEN_TRAIN / 1e12 / 157.3, 4) if args.model == "DiT-B" and args.seq_len == 128 else None}
    # (1b) + (2): the full loop in both tiers
    num_classes, n, T, S = 52670, args.maps, args.seq_len, max(1, min(1000, args.drift_steps))
    (x, o, c), y = synthetic_windows(n, T, num_classes, seed=1000, train_offsets=False)
    o, c = torch.cat([o, o]).to(dev), torch.cat([c, c]).to(dev)
    y = torch.cat([y, torch.full_like(y, num_classes)]).to(dev)
    kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=None)
    g = torch.Generator(device=dev).manual_seed(1234)
    z = torch.randn(n, 2, T, device=dev, generator=g)
    z = torch.cat([z, z])
    noise = torch.randn(S, 2 * n, 2, T, device=dev, generator=g)
    diffusion = create_diffusion(str(S), noise_schedule="squaredcos_cap_v2")
    # Three runs over the same windows and noise: the bf16 tier, the fp32 tier, and the fp32 tier again from an initial state
    # moved by 1e-6 (a few fp32 ulps of a coordinate).  The third run is the yardstick: it shows how far the sampler map itself
    # carries a rounding-sized difference on these (random, untrained) weights, i.e. what ANY two implementations may differ by.
    marks = [k for k in (1, 10, 50, 100, 250, 500, 1000) if k < S] + [S]
    runs = (("bf16", "bf16", 0.0), ("fp16", "fp16", 0.0), ("bf16x3", "bf16x3", 0.0), ("fp16f8", "fp16f8", 0.0), ("fp16w8", "fp16w8", 0.0), ("fp16m8", "fp16m8", 0.0),
            ("fp32", "fp32", 0.0), ("fp32_moved", "fp32", 1e-6))
    states, sec = {}, {}
    pert = torch.randn(n, 2, T, device=dev, generator=g)
    for name, prec, eps in runs:
        torch.manual_seed(4321)  # the trunk's Xavier / normal initialisation draws from the global generator: same weights every time
        model = DiT_models[args.model](num_classes=num_classes, context_size=19 - 3 + 128, precision=prec)
        model = randomize_zero_init(model.to(dev), seed=0).eval()
        model.reserve(2 * n, T)
        st = z.clone()
        diffusion.run_steps(model.forward_with_cfg, st, kw, first_step=S - 1, last_step=S - 2, step_noise=noise[:2])  # warm-up
        st = z.clone()
        st[:n] += eps * pert
        st[n:] += eps * pert
        snaps, done = {}, 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in marks:  # steps S-1-done .. S-k
            diffusion.run_steps(model.forward_with_cfg, st, kw, first_step=S - 1 - done, last_step=S - k, step_noise=noise[done:k])
            snaps[k] = st[:n].clone()
            done = k
        torch.cuda.synchronize()
        sec[name] = time.perf_counter() - t0
        states[name] = snaps
        del model
        torch.cuda.empty_cache()

    def dev_stats(a, b):
        d = (a - b).abs().flatten().double()
        return {"max": float(f"{float(d.max()):.3g}"), "p99": float(f"{float(torch.quantile(d, 0.99)):.3g}"), "mean": float(f"{float(d.mean()):.3g}")}

    M = 2 * n * T
    out["sample"] = {"value": round(S / sec["fp32"], 3), "unit": "steps/s", "ms_per_step": round(sec["fp32"] / S * 1e3, 4), "steps": S,
                     "mfma_frac_of_f32_peak": round(S / sec["fp32"] * M * FLOP_PER_TOKEN_FWD / 1e12 / 157.3, 4) if args.model == "DiT-B" and T == 128 else None}
    out["tier"] = "precision=fp32: exact-f32 MFMA (v_mfma_f32_32x32x2_f32, peak 157.3 TFLOP/s), fp32 storage; carries the 1e-3 claim (tests)"
    drift = dict(dev_stats(states["bf16"][S], states["fp32"][S]))
    drift.update({
        "unit": "normalised playfield coordinates (1 = 512 px in x, 384 px in y)", "steps": S, "rows": n,
        "what": f"|x_t(bf16 tier) - x_t(fp32 tier)| of the conditional rows along the {S}-step CFG-4 p_sample loop: identical windows, "
                f"initial and per-step noise, {args.model} seq-len {T}, seeded random (untrained) weights",
        "after_steps": {str(k): dev_stats(states["bf16"][k], states["fp32"][k]) for k in marks},
        "yardstick_fp32_vs_fp32_moved_by_1e-6": {str(k): dev_stats(states["fp32_moved"][k], states["fp32"][k]) for k in marks},
        "reading": "the yardstick is the fp32 tier against itself from an initial state moved by 1e-6: where it reaches O(1) the "
                   "sampler map on these weights amplifies rounding-sized differences to full scale, and the end-to-end distance "
                   "between two arithmetic tiers stops measuring their accuracy (the teacher-forced per-step error does: tests)",
        "bf16_loop_ms_per_step": round(sec["bf16"] / S * 1e3, 4)})
    # The tiers that meet the 1e-3 tolerance at MFMA speed: split-bf16 operands everywhere (bf16x3: hi + lo planes, three bf16 MFMAs
    # per product), and the same tier with the four big GEMMs of a block on fp16 + e4m3-residual operands (fp16f8: fp16 hi product +
    # ONE block-scaled e4m3 MFMA for both cross terms = 2/3 of the matrix-pipe passes).  `tolerance_tier` is the faster one that meets
    # the bound on this run; the other is listed under "also".
    DTYPES = {
        "bf16x3": "bf16x3 (split-bf16 operands: v = hi + lo, three v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate / residual / statistics)",
        "fp16f8": "fp16f8 (the bf16x3 tier with in_proj / out_proj / fc1 / fc2 on v = fp16 hi + 2^-12 e4m3 lo: two v_mfma_f32_32x32x16_f16 + one "
                  "v_mfma_scale_f32_32x32x64_f8f6f4 per 32 k, fp32 accumulate / residual / statistics)",
        "fp16w8": "fp16w8 (the fp16f8 tier with the ACTIVATION operand of in_proj / out_proj / fc1 / fc2 rounded to fp16 and only the weight "
                  "carrying its e4m3 residual: eight v_mfma_f32_32x32x16_f16 + two v_mfma_scale_f32_32x32x64_f8f6f4 per 128 k, fp32 accumulate / "
                  "residual / statistics)",
        "fp16m8": "fp16m8 (per-GEMM mix of the two forms above: in_proj / out_proj / fc2 on fp16-activation operands, fc1 on fp16 + e4m3 operands -- "
                  "option f16m8_forms = 11)"}
    TIERS = {
        "bf16x3": "precision=bf16x3: 16 significand bits per GEMM / attention operand (the reference's sampling matmuls are TF32: 11 bits, "
                  "sample.py:25-26); meets the 1e-3 tolerance on final coordinates (tests/test_gpu_x3.py) at a third of the bf16 tier's MFMA rate",
        "fp16f8": "precision=fp16f8: 15 significand bits per operand of the four big GEMMs of a block (16 elsewhere; the reference's sampling "
                  "matmuls are TF32: 11 bits, sample.py:25-26); meets the 1e-3 tolerance on final coordinates (tests/test_gpu_h8.py) at "
                  "2/3 of the bf16x3 tier's matrix-pipe passes",
        "fp16w8": "precision=fp16w8: 15 significand bits per WEIGHT of the four big GEMMs of a block, 11 per activation operand (an activation's "
                  "rounding is a fresh error per token and step and averages out over the loop; a weight's repeats in every product): 3-8x closer to "
                  "the reference than TF32-class arithmetic on both operands (the fp16 tier), the bulk of the coordinates inside 1e-3 (p99.9 3e-4), "
                  "isolated ones not (1.1-1.3e-3 here, 2.3e-2 on the CLI fixture): not a tolerance tier",
        "fp16m8": "precision=fp16m8: per-GEMM mix of fp16f8's and fp16w8's operand forms (fc1 keeps the activation's residual, the other three big "
                  "GEMMs drop it): inside 1e-3 on both draws of this loop (tools/tier_drift.py) but NOT on the CLI fixture (one coordinate of 256 at 2.6e-2): "
                  "not a tolerance tier"}
    PASSES = {"bf16x3": 3.0, "fp16f8": 2.0, "fp16w8": 1.5, "fp16m8": 1.5 + 0.5 / 3}  # matrix-pipe passes per product, in units of one bf16 MFMA pass (the big GEMMs)

    def tier_record(name):
        tol = dict(dev_stats(states[name][S], states["fp32"][S]))
        fl = S / sec[name] * M * FLOP_PER_TOKEN_FWD / 1e12 if args.model == "DiT-B" and T == 128 else None
        return {
            "value": round(S / sec[name], 3), "unit": "steps/s", "ms_per_step": round(sec[name] / S * 1e3, 4), "steps": S,
            "dtype": DTYPES[name], "tier": TIERS[name],
            "drift_vs_fp32_tier": {"max": tol["max"], "p99": tol["p99"], "mean": tol["mean"], "steps": S,
                                   "after_steps": {str(k): dev_stats(states[name][k], states["fp32"][k]) for k in marks},
                                   "what": f"same loop, windows and noise as bf16_drift: |x_t({name} tier) - x_t(fp32 tier)| of the conditional rows"},
            "meets_1e-3": bool(tol["max"] <= 1e-3),
            # algorithmic FLOPs (one product per multiply-add of the model) against the bf16 MFMA peak, and the matrix-pipe work
            # actually issued (in bf16-MFMA passes: 3 per product for split-bf16 operands, 2 for fp16 + e4m3) against the same peak
            "algorithmic_tflops": round(fl, 1) if fl else None,
            "mfma_frac_algorithmic": round(fl / PEAK_BF16_TFLOPS, 4) if fl else None,
            "mfma_frac_issued": round(PASSES[name] * fl / PEAK_BF16_TFLOPS, 4) if fl else None}

    # the fast tier on half operands (11 significand bits = the reference's TF32 sampling matmuls) on the same loop
    f16 = dev_stats(states["fp16"][S], states["fp32"][S])
    out["fp16_tier"] = {"value": round(S / sec["fp16"], 3), "unit": "steps/s", "ms_per_step": round(sec["fp16"] / S * 1e3, 4), "steps": S,
                        "dtype": "fp16 (IEEE half MFMA operands, fp32 accumulate / residual / statistics)",
                        "drift_vs_fp32_tier": dict(f16, after_steps={str(k): dev_stats(states["fp16"][k], states["fp32"][k]) for k in marks}),
                        "meets_1e-3": bool(f16["max"] <= 1e-3)}
    recs = {name: tier_record(name) for name in ("bf16x3", "fp16f8", "fp16m8", "fp16w8")}
    # A tier can be THE tolerance tier only if its 1e-3 claim is also held by every reference fixture of the test suite (the reference's
    # 1000-step DiT-B loop, the 20- / 250-step loops, both CLI re-enactments).  The fp16-activation forms are not: on the CLI fixture one
    # coordinate of 256 ends 2-3e-2 away (tests/test_gpu_scripts.py) -- the sampler has coordinates that amplify their ten times larger
    # rounding a hundredfold -- whatever they show on this run's draw.  They are reported, never selected.
    VERIFIED = ("bf16x3", "fp16f8")
    for name, r in recs.items():
        r["held_by_every_reference_fixture"] = name in VERIFIED
    meeting = [name for name in VERIFIED if recs[name]["meets_1e-3"]] or ["bf16x3"]
    best = max(meeting, key=lambda name: recs[name]["value"])
    out["tolerance_tier"] = dict(recs[best], name=best, also={name: r for name, r in recs.items() if name != best})
    return out, drift


XL_TIERS = ["bf16", "fp8"]
PEAK_FP8_TFLOPS = 5000.0
# share of the step's algorithmic FLOPs the fp8 training tier runs on e4m3 operands: in_proj, out_proj, fc1, fc2 -- forward, data-gradient AND
# weight-gradient products; the attention core, the adaLN product, the embedders and the final layer stay bf16
FLOP_PER_TOKEN_TRAIN_XL = 2783.5e6  # DiT-XL, T=256
# the four Linear products of a block are 24 D^2 FLOP per token and block forward (x 3 for training, like everything else): 24 x 1152^2 x 28 =
# 891.8 of DiT-XL's 927.84 MFLOP per token; the rest -- attention core 33.0, adaLN 1.7, first linear 1.2, final layer -- runs in bf16
F8_SHARE = round(3 * 24 * 1152 ** 2 * 28 / FLOP_PER_TOKEN_TRAIN_XL, 4)


def bench_xl(args, world, rank, dev, precision="bf16", steps=None, warmup=None):
    """BASELINE configs[4]'s shape on this job's GPUs: DiT-XL (D=1152, 28 blocks, 16 heads of 72), seq-len 256, 128 windows per
    GPU, the whole train.py step.  A secondary line (the headline metric is DiT-B): few steps, no CPU leg."""
    xa = argparse.Namespace(**vars(args))
    xa.model, xa.seq_len, xa.batch, xa.precision = "DiT-XL", 256, 128, precision
    xa.steps, xa.warmup = (steps if steps is not None else 6), (warmup if warmup is not None else 2)
    xa.no_roofline, xa.no_cpu_baseline, xa.h2d, xa.no_family_table, xa.no_compute_floor = True, True, False, True, True
    r = bench_train(xa, world, rank, dev)
    per_gpu = r["value"] / world
    # Peaks (MI355X_MICROARCH.md): bf16 2.5 PFLOP/s dense, fp8 5 PFLOP/s dense (block-scaled K = 64 MFMA).  The fp8 tier runs the
    # share F8_SHARE of its GEMM FLOPs on e4m3 operands and the rest in bf16, so besides the fraction of the plain fp8 peak SURVEY 8d
    # names it is quoted against the FLOP-weighted peak of that mix: 1 / (share / 5000 + (1 - share) / 2500).
    ach = per_gpu * FLOP_PER_TOKEN_TRAIN_XL / 1e12
    out = {k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "per_gpu_tokens_per_s")}
    if precision == "fp8":
        mixed = 1.0 / (F8_SHARE / PEAK_FP8_TFLOPS + (1.0 - F8_SHARE) / PEAK_BF16_TFLOPS)
        out["end_to_end"] = {"flop_per_token": FLOP_PER_TOKEN_TRAIN_XL, "achieved_tflops_per_gpu": round(ach, 1),
                             "peak_tflops": PEAK_FP8_TFLOPS, "mfma_frac": round(ach / PEAK_FP8_TFLOPS, 4),
                             "fp8_flop_share": F8_SHARE, "mixed_peak_tflops": round(mixed, 1), "mfma_frac_of_mixed_peak": round(ach / mixed, 4),
                             "mfma_frac_of_bf16_peak": round(ach / PEAK_BF16_TFLOPS, 4)}
    else:
        out["end_to_end"] = {"flop_per_token": FLOP_PER_TOKEN_TRAIN_XL, "achieved_tflops_per_gpu": round(ach, 1),
                             "peak_tflops": PEAK_BF16_TFLOPS, "mfma_frac": round(ach / PEAK_BF16_TFLOPS, 4)}
    return out


LINE_LIMIT = 4000  # bytes: the driver reads the LAST stdout line from a bounded tail (round 5's 22.5 KB report was not parsed)
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _cut(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def compact_line(res):
    """The ONE stdout line: the contract fields + `roofline` + `cpu_baseline` + a six-field `sampling` summary (+ `multi_gpu` for
    N > 1), every string cut short, <= LINE_LIMIT bytes by construction (tests/test_bench_line.py).  Everything else a run measures
    (per-family table, per-kernel rows, stand-alone timings, the other tiers, drift tables, the fp32 tier, DiT-XL) is written to
    bench_detail.json next to this file and echoed on stderr."""
    out = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    out["vs_baseline"] = res.get("vs_baseline")
    out.update(_pick(res, ("dtype", "data")))
    out["metric"] = _cut(out.get("metric"), 80)
    out["dtype"] = _cut(out.get("dtype"), 24)
    cfg = dict(res.get("config") or {})
    for k, n in (("workload", 170), ("parallelism", 150), ("sharding", 40)):
        if k in cfg:
            cfg[k] = _cut(cfg[k], n)
    out["config"] = cfg
    if "end_to_end" in res:
        out["end_to_end"] = _pick(res["end_to_end"], ("achieved_tflops_per_gpu", "achieved_tflops", "mfma_frac", "mfma_frac_of_mixed_peak"))
    rf = res.get("roofline")
    if isinstance(rf, dict):
        r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "flop_per_launch", "avg_launch_us", "measured"))
        r["kernel"] = _cut(r.get("kernel"), 90)
        r["measured"] = _cut(r.get("measured"), 60)
        r.setdefault("traffic", None)
        low = rf.get("lowest_fraction_top_family")
        if isinstance(low, dict):
            r["lowest_fraction_top_family"] = _pick(low, ("family", "ms_per_step", "share", "frac", "heaviest_kernel", "heaviest_kernel_frac"))
        pf = rf.get("per_family")
        if isinstance(pf, dict) and "families" in pf:
            r["non_mfma_ms_per_step"] = pf.get("non_mfma_ms_per_step")
            r["family_frac"] = {k: v["frac"] for k, v in pf["families"].items() if "frac" in v and v.get("share", 0) >= 0.04}
        fc1 = rf.get("fc1_forward")
        if isinstance(fc1, dict):
            r["fc1_forward"] = _pick(fc1, ("avg_launch_us", "frac", "traffic", "algorithmic_bytes"))
        out["roofline"] = r
    if "cpu_baseline" in res:
        cb = dict(res["cpu_baseline"])
        cb["sample"] = _cut(cb.get("sample"), 150)
        out["cpu_baseline"] = cb
    sp = res.get("sampling")
    if isinstance(sp, dict):
        tol = (sp.get("tolerance") or {}).get("this_line") or {}
        s = _pick(sp, ("value", "unit", "tier", "ms_per_step", "steps"))
        s["meets_1e-3"] = tol.get("meets", (sp.get("tolerance") or {}).get("meets_1e-3"))
        if "held_by_every_reference_fixture" in tol:
            s["held_by_every_reference_fixture"] = tol["held_by_every_reference_fixture"]
            s["max_drift_vs_fp32_tier"] = tol.get("max_drift_vs_fp32_tier")
        s["mfma_frac"] = (sp.get("end_to_end") or {}).get("mfma_frac")
        cb = sp.get("cpu_baseline") or ((sp.get("also") or {}).get("bf16") or {}).get("cpu_baseline")
        if cb:
            s["cpu_baseline"] = dict(_pick(cb, ("value", "unit", "cores", "kind")), sample=_cut(cb.get("sample"), 90))
        also = {}
        for name, v in (sp.get("also") or {}).items():
            if isinstance(v, dict) and "value" in v:
                m = v.get("meets_1e-3", (v.get("tolerance") or {}).get("meets_1e-3"))
                also[name] = {"value": v["value"], "meets_1e-3": m} if m is not None else {"value": v["value"]}
        if also:
            s["also"] = also
        out["sampling"] = s
    if "parity_tier" in res:
        pt = res["parity_tier"]
        out["parity_tier"] = {k: _pick(pt[k], ("value", "unit", "mfma_frac_of_f32_peak")) for k in ("train", "sample") if k in pt}
    if "xl" in res:
        out["xl"] = {p: dict(_pick(v, ("value", "unit", "ms_per_step")), mfma_frac=(v.get("end_to_end") or {}).get("mfma_frac")) for p, v in res["xl"].items()}
    if "pcie_inclusive" in res:
        out["pcie_inclusive"] = _pick(res["pcie_inclusive"], ("value", "unit", "ms_per_step"))
    mg = res.get("multi_gpu")
    if isinstance(mg, dict):
        m = _pick(mg, ("rccl", "fastest_schedule", "schedule_selected", "exposed_comm_ms_per_step", "timed_schedule", "wire_bytes_per_step", "multi_gpu_schedule_compute_floor_ms"))
        m["schedules"] = {k: ({kk: _cut(vv, 80) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in (mg.get("schedules") or {}).items()}
        pc = mg.get("predicted_comm_ms_per_step")
        if isinstance(pc, dict):
            m["predicted_comm_ms_per_step"] = {k: v for k, v in pc.items() if k != "note"}
        out["multi_gpu"] = m
    if "multi_gpu_schedule_compute_floor_ms" in res:
        out["multi_gpu_schedule_compute_floor_ms"] = res["multi_gpu_schedule_compute_floor_ms"]
    if "diagnostics_incomplete" in res:
        out["diagnostics_incomplete"] = _pick(res["diagnostics_incomplete"], ("stage", "after_s"))
    out["detail"] = "bench_detail.json (next to bench.py; also echoed on stderr)"
    line = json.dumps(out, default=str)
    # by construction the line is well below the limit; should a future field outgrow it, drop the optional objects, never the contract fields
    for k in ("pcie_inclusive", "parity_tier", "xl", "end_to_end", "multi_gpu"):
        if len(line) <= LINE_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out, default=str)
    return line


def emit(res):
    """Rank 0: the full report to bench_detail.json and stderr, the compact line LAST on stdout."""
    full = json.dumps(res, default=str)
    try:
        with open(DETAIL_FILE, "w") as f:
            f.write(full + "\n")
    except OSError as e:  # (a read-only tree must not cost the line)
        print(f"bench_detail.json not written: {e}", file=sys.stderr, flush=True)
    print("bench_detail " + full, file=sys.stderr, flush=True)
    print(compact_line(res), flush=True)


def main():
    args = parse()
    world, rank, local = dist_setup(args)
    dev = torch.device("cuda", local if world > 1 else 0)
    if args.mode == "xl":
        res = bench_xl(args, world, rank, dev, args.xl_precision or "bf16", args.steps, args.warmup)
        res.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
    elif args.mode == "train":
        res = bench_train(args, world, rank, dev)
    elif args.mode == "sample":
        res = bench_sample(args, world, rank, dev)
    else:
        res = bench_train(args, world, rank, dev)
        WATCHDOG.stage = "sampling"
        sargs = argparse.Namespace(**vars(args))
        sargs.steps, sargs.warmup = args.sample_steps, None
        samp = bench_sample(sargs, world, rank, dev)
        res["sampling"] = {k: samp[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "end_to_end",
                                                "roofline", "cpu_baseline") if k in samp}
        if world == 1 and args.precision == "bf16" and not args.no_parity_tier:
            res["parity_tier"], res["bf16_drift"] = parity_tier_and_drift(args, dev)
            tt = res["parity_tier"].pop("tolerance_tier")
            fp16_tier = res["parity_tier"].pop("fp16_tier")
            # Which number answers "1000-step CFG sampling steps/s, matching the reference within 1e-3": the fastest tier whose 1000-step
            # drift from the exact-f32 tier stays below the bound on this very workload AND whose claim is held by every reference fixture
            # of the test suite.  THAT is `sampling.value`; the bf16 tier's (faster, outside the tolerance) line moves to `also.bf16`.
            bd = res["bf16_drift"]
            bf16_line = res["sampling"]
            bf16_line["tolerance"] = {"max_drift_vs_fp32_tier": bd["max"], "p99": bd["p99"], "meets_1e-3": bool(bd["max"] <= 1e-3)}
            credited = tt if tt["meets_1e-3"] else dict(res["parity_tier"]["sample"], name="fp32", dtype="f32", **{"meets_1e-3": True})
            res["sampling"] = {
                "metric": bf16_line["metric"], "value": credited["value"], "unit": "steps/s", "steps": credited["steps"], "warmup": 2,
                "ms_per_step": credited["ms_per_step"], "tier": credited.get("name"), "dtype": credited.get("dtype"),
                "config": bf16_line["config"],
                "tolerance": {
                    "bound": 1e-3, "unit": "normalised playfield coordinates, max over the conditional rows after the full loop",
                    "this_line": {"tier": credited.get("name"), "max_drift_vs_fp32_tier": (credited.get("drift_vs_fp32_tier") or {}).get("max", 0.0), "meets": True,
                                  "held_by_every_reference_fixture": credited.get("held_by_every_reference_fixture", True)},
                    "reference_fixture": "tests/golden/g6_loop_p1000_dit_b.npz: the reference's own 1000-step CFG-4 DiT-B loop; tests/test_gpu_x3.py, "
                                         "tests/test_gpu_h8.py (fp32 tier 9.8e-5, bf16x3 1.2e-4, fp16f8 1.2e-4, bf16 8.8e-3 from it)"},
                "end_to_end": {"flop_per_step": bf16_line.get("end_to_end", {}).get("flop_per_step"), "achieved_tflops": credited.get("algorithmic_tflops"),
                               "mfma_frac": credited.get("mfma_frac_algorithmic"), "mfma_frac_issued_passes": credited.get("mfma_frac_issued")},
                "tolerance_tier": tt,
                "also": {"bf16": bf16_line, "fp16": fp16_tier},
            }
            # the fp8 inference tier on the same sampling workload (reduced precision: 0.7 % rms from the fp32 oracle, tests/test_gpu_fp8.py)
            fargs = argparse.Namespace(**vars(args))
            fargs.precision, fargs.steps, fargs.warmup, fargs.no_roofline, fargs.no_cpu_baseline = "fp8", 300, 30, True, True
            f8 = bench_sample(fargs, world, rank, dev)
            res["sampling"]["also"]["fp8"] = {"value": f8["value"], "unit": "steps/s", "ms_per_step": f8["ms_per_step"], "steps": f8["steps"], "dtype": f8["dtype"]}
        if not args.no_xl and args.precision == "bf16":
            WATCHDOG.stage = "xl"
            res["xl"] = {p: bench_xl(args, world, rank, dev, p) for p in ([args.xl_precision] if args.xl_precision else XL_TIERS)}
    WATCHDOG.disarm()
    if rank == 0:
        emit(res)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
