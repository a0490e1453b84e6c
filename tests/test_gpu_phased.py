"""GPU (-m gpu): the phased main loops (csrc/gemm_phased.h, csrc/wgrad.hip: wgrad_phased_kernel; option gemm_loop = 1, the default)
against the slab loops they re-schedule (gemm_loop = 0).

The phased kernels run the same staging, fragment reads, MFMAs and epilogue; every accumulator sees its k in the same order.  So the
bar is BIT equality -- of single launches in every operand form the loop is built for (bf16, fp16, fp16 + e4m3 rows, e4m3), on both
256-row tile geometries, with even and odd slab counts (the buffer parity alternates between tiles), one-tile and multi-round grids --
and of whole-model results: a forward per sampling tier, a sampler loop, a training step's gradients in the bf16 and the fp8 tier.
What each form computes is pinned elsewhere against the reference (tests/test_gpu_forward.py, test_gpu_h8.py, test_gpu_train.py, ...):
those suites run on the phased loops by default, this file ties the two schedules together.
"""
import pytest
import torch

from oracle import dit_oracle as mo
from osu_diffusion_amd import _lib
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT
from osu_diffusion_amd.synthetic import synthetic_windows
from osu_diffusion_amd.training import NativeTrainer

# every test here compares two schedules of THIS library with each other: `selfcheck` (collected behind every reference / oracle test, tests/conftest.py)
pytestmark = [pytest.mark.gpu, pytest.mark.selfcheck]
DEV = "cuda:0"
EPI = {"bias_f32": _lib.EPI_BIAS_F32, "bias": _lib.EPI_BIAS_TE, "silu": _lib.EPI_BIAS_SILU_TE, "rowbias": _lib.EPI_ROWBIAS_TE,
       "gelu": _lib.EPI_BIAS_GELU_TE, "gate": _lib.EPI_GATE_RES, "none_f32": _lib.EPI_NONE_F32, "none": _lib.EPI_NONE_TE,
       "accum": _lib.EPI_ACCUM_F32}
F32OUT = {"bias_f32", "gate", "none_f32", "accum"}


def _gemm(prec, epi, Y, ldy, X, ldx, M, N, K, out, bias, gate):
    _lib.check(_lib.lib().osud_op_gemm(prec, EPI[epi], _lib.ptr(Y), ldy, _lib.ptr(X), ldx, M, N, K, _lib.ptr(out), N, _lib.ptr(bias),
                                       _lib.ptr(gate) if gate is not None else None, N if gate is not None else 0, 128 if gate is not None else 0,
                                       M // 128 if gate is not None else 0, None))


def _both_loops(osud_option, run):
    res = []
    for loop in (0, 1):
        osud_option("gemm_loop", loop)
        res.append(run())
        torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("tile", [256, 192])
@pytest.mark.parametrize("epi", sorted(EPI))
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_plain_operand_gemm_is_bit_identical_to_the_slab_loop(osud_option, prec, epi, tile):
    """512 x 768 outputs (2 x 3 / 2 x 4 tiles: one tile per workgroup), K = 768 (12 slabs) and K = 192 (3 slabs).  A workgroup that crosses a tile
    boundary with an odd slab count -- where the stream cursor and the consumer's buffer parity continue flipped into the next tile -- is in
    test_many_rounds_and_the_shortest_stream."""
    dt, pc = (torch.bfloat16, _lib.PREC_BF16) if prec == "bf16" else (torch.float16, _lib.PREC_F16)
    if prec == "fp16" and epi in ("accum", "rowbias", "none"):
        pytest.skip("the fp16 tier builds the forward pass's epilogues only (csrc/gemm_f16.hip)")
    osud_option("gemm_tile", tile)
    g = torch.Generator(device=DEV).manual_seed(5)
    M, N = 512, 768
    for K in (768, 192):
        Y = torch.randn(M, K, device=DEV, generator=g).to(dt)
        X = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(dt)
        bias = torch.randn(max(M, N), device=DEV, generator=g)
        gate = torch.randn(M // 128, N, device=DEV, generator=g) if epi == "gate" else None
        init = torch.randn(M, N, device=DEV, generator=g)

        def run():
            out = init.clone() if epi in F32OUT else init.to(dt)
            _gemm(pc, epi, Y, K, X, K, M, N, K, out, bias, gate)
            return out

        a, b = _both_loops(osud_option, run)
        assert torch.equal(a, b), (prec, epi, tile, K)
        if epi == "none_f32":  # ... and it is the product
            ref = Y.float() @ X.float().t()
            assert float((b - ref).abs().max()) < 2e-3 * max(1.0, float(ref.abs().max()))


def test_many_rounds_and_the_shortest_stream(osud_option):
    """More tiles than compute units (the stream runs across tile boundaries: 2048 x 3072 = 96 / 128 tiles on <= 256 workgroups is one round, so
    also 8192 x 3072 = 384 tiles), and K = 128 = two slabs, the shortest stream the loop takes;
    also K = 192 / 320 (3 / 5 slabs: odd) on multi-round grids, where a workgroup enters its next tile with the buffer parity flipped."""
    g = torch.Generator(device=DEV).manual_seed(6)
    for M, N, K, tile in ((8192, 3072, 768, 256), (8192, 2304, 256, 192), (256, 768, 128, 256), (256, 768, 128, 192), (1024, 256, 320, 256),
                          (8192, 3072, 192, 256), (8192, 2304, 320, 192), (16384, 3072, 320, 256)):
        osud_option("gemm_tile", tile)
        Y = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
        X = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV, generator=g)

        def run():
            out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            _gemm(_lib.PREC_BF16, "gelu", Y, K, X, K, M, N, K, out, bias, None)
            return out

        a, b = _both_loops(osud_option, run)
        assert torch.equal(a, b), (M, N, K, tile)


def test_e4m3_operands_on_a_multi_round_grid_with_an_odd_slab_count(osud_option):
    """DiT-XL's fp8 shape class: e4m3 operands (128 k per 128-byte slab), K = 1152 = 9 slabs, 16384 x 1152 outputs = 384 tiles of 256 x 192 on
    <= 256 workgroups: a workgroup enters its second tile with the stream cursor and the consumer's buffer parity flipped.  Bit equality of
    the two loops, static and queued tile order."""
    g = torch.Generator(device=DEV).manual_seed(8)
    M, N, K = 16384, 1152, 1152
    Y = torch.randn(M, K, device=DEV, generator=g).to(torch.float8_e4m3fn)
    X = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5 * 8).to(torch.float8_e4m3fn)
    bias = torch.randn(N, device=DEV, generator=g)
    L = _lib.lib()
    outs = []
    try:
        for loop, dyn in ((0, 0), (1, 0), (1, 1)):
            osud_option("gemm_loop", loop)
            _lib.check(L.osud_set_gemm_dynamic_tiles(dyn))
            out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            _gemm(_lib.PREC_FP8, "bias", Y, K, X, K, M, N, K, out, bias, None)
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        _lib.check(L.osud_set_gemm_dynamic_tiles(-1))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = Y.float() @ X.float().t() + bias
    assert float((outs[1].float() - ref).abs().max()) < 1e-2 * float(ref.abs().max())


@pytest.mark.parametrize("shape", [(256, 256, 128), (512, 256, 1024), (768, 3072, 8192), (1152, 1152, 4096), (384, 256, 2048),
                                   (3456, 1152, 4096), (1152, 4608, 4096), (4608, 1152, 2112)])  # DiT-XL: the 256 x 192 and 192 x 256 geometries (ragged splits: 33 stages)
def test_weight_gradient_kernel_is_bit_identical_to_the_slab_loop(osud_option, shape):
    """out = P^T Q over M tokens: split over the token axis into partial slabs + the fixed-order combine; odd multiples of 128 (half-empty edge
    tiles) and token counts that leave the splits ragged included."""
    Ny, Nx, M = shape
    g = torch.Generator(device=DEV).manual_seed(7)
    P = torch.randn(M + 1, Ny, device=DEV, generator=g).to(torch.bfloat16)[:M]  # (one spare row: the padded tiles' over-read stays inside the allocation)
    Q = torch.randn(M + 1, Nx, device=DEV, generator=g).to(torch.bfloat16)[:M]
    ws = torch.empty(32 * Ny * Nx, device=DEV)

    def run():
        out = torch.full((Ny, Nx), float("nan"), device=DEV)
        ws.fill_(float("nan"))
        _lib.check(_lib.lib().osud_op_wgrad(_lib.ptr(P), Ny, _lib.ptr(Q), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None))
        return out

    a, b = _both_loops(osud_option, run)
    assert torch.equal(a, b)
    ref = P.float().t() @ Q.float()
    assert float((b - ref).abs().max()) < 1e-3 * float(ref.abs().max())


def test_ticket_queues_in_the_phased_loops(osud_option):
    """Shared-GPU mode (osud_set_gemm_dynamic_tiles(1), what a data-parallel trainer switches on) INSIDE the phased loops: multi-round GEMM
    launches draw their tiles from the per-XCD ticket queues (every tile is computed the same way whoever takes it: bit-identical to the
    static order, in both loops, also with an odd slab count), and the split-K weight-gradient kernel draws K-chunks per tile through a
    scalar atomic (the partial sums then depend on who took which chunk: equal to rounding).  Each case twice: the second launch runs on
    counters the first one re-armed."""
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(11)
    osud_option("gemm_loop", 1)
    try:
        for M, N, K, tile, epi in ((8192, 3072, 768, 256, "gelu"), (8192, 2304, 320, 192, "bias"), (16384, 768, 3072, 192, "bias"), (16384, 3072, 192, 256, "gelu")):
            osud_option("gemm_tile", tile)
            Y = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
            X = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
            bias = torch.randn(N, device=DEV, generator=g)
            outs = []
            for mode in (0, 1, 1):
                _lib.check(L.osud_set_gemm_dynamic_tiles(mode))
                out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
                _gemm(_lib.PREC_BF16, epi, Y, K, X, K, M, N, K, out, bias, None)
                torch.cuda.synchronize()
                outs.append(out)
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (M, N, K, tile)
        osud_option("gemm_tile", 0)
        for Ny, Nx, M in ((768, 768, 32768), (3072, 768, 32768), (768, 2304, 16384)):  # chunks of >= 4 stages: the phased kernel takes the queued launch
            P = torch.randn(M, Ny, device=DEV, generator=g).to(torch.bfloat16)
            Q = torch.randn(M, Nx, device=DEV, generator=g).to(torch.bfloat16)
            ws = torch.empty(32 * Ny * Nx, device=DEV)
            outs = []
            for mode in (0, 1, 1):
                _lib.check(L.osud_set_gemm_dynamic_tiles(mode))
                out = torch.full((Ny, Nx), float("nan"), device=DEV)
                ws.fill_(float("nan"))
                _lib.check(L.osud_op_wgrad(_lib.ptr(P), Ny, _lib.ptr(Q), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None))
                torch.cuda.synchronize()
                outs.append(out)
            scale = float(outs[0].abs().max())
            assert float((outs[1] - outs[0]).abs().max()) < 2e-5 * scale and float((outs[2] - outs[0]).abs().max()) < 2e-5 * scale, (Ny, Nx, M)
            ref = P.float().t() @ Q.float()
            assert float((outs[2] - ref).abs().max()) < 1e-3 * float(ref.abs().max())
    finally:
        _lib.check(L.osud_set_gemm_dynamic_tiles(-1))


def _model(shape, sd, precision, train=False):
    m = DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads, context_size=shape.context, num_classes=shape.num_classes,
            class_dropout_prob=0.2, precision=precision)
    m.load_state_dict(sd)
    m = m.to(DEV)
    return m.train() if train else m.eval()


@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp16f8", "fp8"])
def test_forward_of_every_tier_is_bit_identical_under_both_loops(osud_option, precision):
    """DiT-B's block geometry (hidden 768, 12 heads), 2 blocks, 8 windows of 128 tokens = 1024 rows: with the 256-row tiles forced, in_proj /
    fc1 run the 256 x 256 geometry and out_proj / fc2 the 256 x 192 one in each tier's operand form (fp16f8: fp16 + e4m3 rows, fp8: e4m3)."""
    shape = mo.DitShape(depth=2, hidden=768, heads=12, num_classes=8)
    sd = mo.seeded_state_dict(shape, 17)
    (x, o, c), y = synthetic_windows(8, 128, 8, seed=2)
    t = torch.tensor([0, 1, 10, 100, 500, 800, 998, 999])
    outs = []
    for tile in (256, 192):
        osud_option("gemm_tile", tile)
        for loop in (0, 1):
            osud_option("gemm_loop", loop)
            m = _model(shape, sd, precision)
            with torch.no_grad():
                outs.append(m(x, t, o, c, y).clone())
            torch.cuda.synchronize()
        assert torch.equal(outs[-2], outs[-1]), (precision, tile)
    assert torch.isfinite(outs[-1]).all()


def test_sampler_loop_is_bit_identical_under_both_loops(osud_option):
    """20 p_sample steps with CFG 4 in the tolerance tier on given noise, through the captured graph: the graph is rebuilt when the option
    changes (the option epoch is in its key), so loop 0 / 1 / 0 must give A / A / A -- not a replay of the first capture."""
    shape = mo.DitShape(depth=2, hidden=768, heads=12, num_classes=8)
    sd = mo.seeded_state_dict(shape, 18)
    (x, o, c), y = synthetic_windows(4, 128, 8, seed=3)
    osud_option("gemm_tile", 256)
    m = _model(shape, sd, "fp16f8")
    diff = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    z = torch.randn(4, 2, 128, generator=torch.Generator().manual_seed(9)).to(DEV)
    z = torch.cat([z, z], 0)
    kw = dict(o=torch.cat([o, o]).to(DEV), c=torch.cat([c, c]).to(DEV), y=torch.cat([y, torch.full_like(y, shape.num_classes)]).to(DEV), cfg_scale=4.0)
    noise = torch.randn(20, 8, 2, 128, generator=torch.Generator().manual_seed(10)).to(DEV)
    res = []
    for loop in (0, 1, 0):
        osud_option("gemm_loop", loop)
        res.append(diff.p_sample_loop(m.forward_with_cfg, z.shape, z.clone(), clip_denoised=True, model_kwargs=kw, device=DEV, step_noise=noise).clone())
        torch.cuda.synchronize()
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
    assert torch.isfinite(res[0]).all()


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_training_step_gradients_are_bit_identical_under_both_loops(osud_option, precision):
    """Forward, data gradients (incl. the GELU' epilogue with its column sums) and weight gradients of a training step: every gradient tensor,
    the loss terms and the stepped masters equal bit for bit between the two schedules."""
    shape = mo.DitShape(depth=2, hidden=768, heads=12, num_classes=8)
    sd = mo.seeded_state_dict(shape, 19)
    (x, o, c), y = synthetic_windows(8, 128, 8, seed=4)
    t = torch.tensor([0, 3, 50, 200, 500, 700, 900, 999])
    noise = torch.randn(8, 2, 128, generator=torch.Generator().manual_seed(3))
    osud_option("gemm_tile", 256)
    res = []
    for loop in (0, 1):
        osud_option("gemm_loop", loop)
        tr = NativeTrainer(_model(shape, sd, precision, train=True), create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True))
        for _ in range(2):  # (fp8: the second step runs on the first one's delayed scales)
            terms = tr.step(x, o, c, y, t=t, noise=noise, drop_ids=torch.zeros(8, dtype=torch.long))
        torch.cuda.synchronize()
        res.append((terms.clone(), {k: v.clone() for k, v in tr.arena.grad_views().items()}, tr.arena.flat.clone()))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    assert torch.equal(res[0][2], res[1][2])


def test_race_screen_under_memory_load(osud_option):
    """The staging rules of the phased loops (a piece is read one phase after the wait that retires it; a region is restaged two phases after
    its last read) are checked at compile time, but what they guard against -- a fragment read that overtakes its LDS-DMA piece, a piece
    that overwrites rows still being read -- only shows when a piece lands late or early.  So: 150 launches of the GEMM and of the
    weight-gradient kernel while a second stream keeps HBM and the L2s busy with device copies of changing size (DMA latencies move by
    factors), every result compared with the slab loop's bits; shapes with many rounds, odd slab counts and ragged splits."""
    g = torch.Generator(device=DEV).manual_seed(21)
    side = torch.cuda.Stream()
    junk_a = torch.empty(96 << 20, dtype=torch.uint8, device=DEV)
    junk_b = torch.empty_like(junk_a)
    cases = []
    for M, N, K, tile in ((4096, 3072, 768, 256), (4096, 768, 3072, 192), (2048, 2304, 192, 192), (1024, 512, 320, 256)):
        Y = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
        X = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV, generator=g)
        cases.append((M, N, K, tile, Y, X, bias))
    P = torch.randn(8192, 768, device=DEV, generator=g).to(torch.bfloat16)
    Q = torch.randn(8192, 1024, device=DEV, generator=g).to(torch.bfloat16)
    ws = torch.empty(32 * 768 * 1024, device=DEV)

    def gemm(case):
        M, N, K, tile, Y, X, bias = case
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        _gemm(_lib.PREC_BF16, "gelu", Y, K, X, K, M, N, K, out, bias, None)
        return out

    def wgrad():
        out = torch.empty(768, 1024, device=DEV)
        _lib.check(_lib.lib().osud_op_wgrad(_lib.ptr(P), 768, _lib.ptr(Q), 1024, 768, 1024, 8192, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None))
        return out

    osud_option("gemm_loop", 0)
    want = []
    for case in cases:
        osud_option("gemm_tile", case[3])
        want.append(gemm(case))
    want_w = wgrad()
    torch.cuda.synchronize()
    osud_option("gemm_loop", 1)
    bad = 0
    for it in range(150):
        n = (1 + (it * 7) % 13) << 22  # 4 .. 52 MiB per copy, a different size every iteration
        with torch.cuda.stream(side):
            for _ in range(3):
                junk_b[:n].copy_(junk_a[:n], non_blocking=True)
        k = it % len(cases)
        osud_option("gemm_tile", cases[k][3])
        got = gemm(cases[k])
        got_w = wgrad() if it % 3 == 0 else None
        bad += int(not torch.equal(got, want[k]))
        if got_w is not None:
            bad += int(not torch.equal(got_w, want_w))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} launches of the phased loops differ from the slab loop under memory load"
