"""GPU (-m gpu): the 8-bit block code of the saved GELU derivative (option gelu_code; csrc/common.h: gelu_code4 / gelu_decode4) at operator level,
through the C ABI (osud_op_gemm_ex), against a plain PyTorch fp32 restatement of nn.GELU(approximate="tanh")'s derivative
(/root/reference models.py:138: the MLP's activation; autograd's derivative of it is what the reference's backward multiplies with).

Producer: the fc1 epilogue (bias + GELU, epilogue 4) writes the derivative as codes; decoded on the host with the layout the header documents
they must sit within the code's step of the exact derivative -- 2.5e-3 + the bf16 GEMM's own error in z -- and be EXACT where the derivative is
saturated.  Consumer: the fc2 data-gradient epilogue (epilogue 9) multiplies its product with the decoded codes.  Both on the two 256-row tile
geometries and the 128 x 128 one (every geometry shares the epilogue's lane order, which is what makes the block layout geometry-free)."""
import math

import pytest
import torch

from osu_diffusion_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gelu_tanh_grad(z):
    """d/dz of 0.5 z (1 + tanh(k (z + 0.044715 z^3))), k = sqrt(2 / pi), in fp64."""
    z = z.double()
    k = math.sqrt(2.0 / math.pi)
    u = k * (z + 0.044715 * z ** 3)
    t = torch.tanh(u)
    return 0.5 * (1 + t) + 0.5 * z * (1 - t * t) * k * (1 + 3 * 0.044715 * z * z)


def decode(code, M, N):
    """uint8 [M * N] in 32 x 32 blocks of 1 KiB -> float [M][N]: lane l = 4 (row & 15) + ((col & 31) >> 3) holds 16 bytes at 16 l,
    columns (col & ~7) .. + 7 of row (row & 15), then of row 16 + (row & 15) (include/osud.h: osud_op_gemm_ex)."""
    b = code.view(M // 32, N // 32, 16, 4, 2, 8)          # block row, block col, lane >> 2, lane & 3, half (row + 16), byte
    rows = b.permute(0, 4, 2, 1, 3, 5).reshape(M, N)       # row = 32 by + 16 half + (lane >> 2); col = 32 bx + 8 (lane & 3) + byte
    return (rows.float() - 26.0) / 200.0


def _ex(prec, epi, Y, X, M, N, K, out, bias=None, out2=None, aux=None, aux_code=0):
    _lib.check(_lib.lib().osud_op_gemm_ex(prec, epi, _lib.ptr(Y), K, _lib.ptr(X), K, M, N, K, _lib.ptr(out), N,
                                          _lib.ptr(bias) if bias is not None else None, _lib.ptr(out2) if out2 is not None else None,
                                          _lib.ptr(aux) if aux is not None else None, aux_code, None, None))


@pytest.mark.parametrize("tile", [256, 192, 128])
def test_code_written_by_the_gelu_epilogue_and_read_by_the_data_gradient_epilogue(osud_option, tile):
    osud_option("gemm_tile", tile)
    g = torch.Generator(device=DEV).manual_seed(21 + tile)
    M, N, K = 1024, 768, 256
    Y = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    X = (torch.randn(N, K, device=DEV, generator=g) * 2.5 / K ** 0.5).to(torch.bfloat16)   # pre-activations out to |z| ~ 10: both saturated ends occur
    bias = torch.randn(N, device=DEV, generator=g) * 0.5
    z = Y.float() @ X.float().t() + bias                      # what the epilogue sees, up to the MFMA's accumulation order
    want = gelu_tanh_grad(z).float()
    # ---- producer
    act_rows, act_code = (torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    d_rows = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    d_code = torch.full((M * N,), 255, dtype=torch.uint8, device=DEV)
    _ex(_lib.PREC_BF16, _lib.EPI_BIAS_GELU_TE, Y, X, M, N, K, act_rows, bias, out2=d_rows)
    _ex(_lib.PREC_BF16, _lib.EPI_BIAS_GELU_TE, Y, X, M, N, K, act_code, bias, out2=d_code, aux_code=1)
    torch.cuda.synchronize()
    assert torch.equal(act_rows, act_code)                   # the GELU output does not depend on how its derivative is saved
    got = decode(d_code, M, N)
    err_code, err_rows = float((got - want).abs().max()), float((d_rows.float() - want).abs().max())
    print(f"MEASURED gelu code [tile {tile}]: max |decoded - exact derivative| {err_code:.2e} (code step 5e-3: bound 2.5e-3 + z's own error); "
          f"bf16 rows {err_rows:.2e}")
    assert err_code < 3.5e-3 and err_rows < 4.5e-3
    sat0, sat1 = z < -9.0, z > 9.0                            # saturated ends: exactly 0 and 1 in both forms
    assert int(sat0.sum()) > 0 and int(sat1.sum()) > 0
    assert float(got[sat0].abs().max()) == 0.0 and float((got[sat1] - 1.0).abs().max()) == 0.0
    assert int(d_code.max()) <= 252 and int(d_code.min()) >= 0   # (the derivative's range: codes 0 .. 252)
    # ---- consumer: out = (P . W^T) * derivative, with the derivative as codes / as the bf16 rows holding the DECODED values
    Kd = 256
    P = torch.randn(M, Kd, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, Kd, device=DEV, generator=g) / Kd ** 0.5).to(torch.bfloat16)
    o_code, o_rows = (torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    _ex(_lib.PREC_BF16, _lib.EPI_GELUGRAD_TE, P, W, M, N, Kd, o_code, aux=d_code, aux_code=1)
    _ex(_lib.PREC_BF16, _lib.EPI_GELUGRAD_TE, P, W, M, N, Kd, o_rows, aux=got.to(torch.bfloat16))
    torch.cuda.synchronize()
    ref = (P.float() @ W.float().t()) * got
    scale = float(ref.abs().max())
    assert float((o_code.float() - ref).abs().max()) < 6e-3 * scale      # bf16 rounding of the output
    # the decoded value goes into the product at fp32 in the code form and as a bf16 row in the other: they differ by that rounding only
    assert float((o_code.float() - o_rows.float()).abs().max()) < 8e-3 * scale


def test_code_is_rejected_where_it_does_not_exist():
    M = N = K = 128
    Y = torch.zeros(M, K, device=DEV)
    out = torch.zeros(M, N, device=DEV)
    with pytest.raises(Exception, match="8-bit code"):
        _ex(_lib.PREC_F32, _lib.EPI_BIAS_GELU_TE, Y, Y, M, N, K, out, torch.zeros(N, device=DEV), out2=torch.zeros(M, N, device=DEV), aux_code=1)
