"""GPU (-m gpu): the native row exchange of the class-table gradient (csrc/exchange.hip, C ABI osud_table_rows_pack / _apply)
against the tensor-library formulation it replaces (training.exchange_table_rows' CPU path) and against the dense all-reduce the
reference's DDP performs (train.py:257) -- with 8 simulated ranks, duplicate labels inside a rank, overlaps across ranks."""
import pytest
import torch

from osu_diffusion_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("W,B,rows,D", [(8, 256, 52671, 768), (2, 16, 12, 128), (1, 4, 11, 384), (8, 512, 700, 128)])
def test_row_exchange_kernels_equal_rank_ordered_dense_sum(W, B, rows, D):
    L = _lib.lib()
    g = torch.Generator().manual_seed(W * 1000 + B)
    labels = [torch.randint(0, rows, (B,), generator=g) for _ in range(W)]
    if B >= 16:
        labels[0][:8] = labels[0][0]       # heavy duplication inside a rank
        labels[-1][:4] = labels[0][0]      # ... and the same row touched by another rank
    dense = []
    for r in range(W):  # what each rank's backward leaves: only its label rows non-zero
        t = torch.zeros(rows, D)
        t.index_add_(0, labels[r], torch.randn(B, D, generator=g))
        dense.append(t)
    want = torch.zeros(rows, D)
    for t in dense:  # rank order
        want += t
    all_idx = torch.empty(W, B, dtype=torch.int64, device=DEV)
    all_rows = torch.empty(W, B, D, device=DEV)
    for r in range(W):
        tg = dense[r].to(DEV)
        _lib.check(L.osud_table_rows_pack(_lib.ptr(tg), rows, D, _lib.ptr(labels[r].to(DEV)), B, _lib.ptr(all_idx[r]), _lib.ptr(all_rows[r]), None))
    torch.cuda.synchronize()
    # pack == sorted labels + first-occurrence rows
    for r in range(W):
        ys, _ = torch.sort(labels[r])
        first = torch.ones(B, dtype=torch.bool)
        first[1:] = ys[1:] != ys[:-1]
        assert torch.equal(all_idx[r].cpu(), ys)
        assert torch.equal(all_rows[r].cpu(), dense[r][ys] * first.unsqueeze(1))
    scratch = torch.empty(W * B + 1, dtype=torch.int64, device=DEV)
    outs = []
    for r in (0, W - 1):  # the result on two different ranks (each starts from its OWN dense gradient)
        tg = dense[r].to(DEV)
        _lib.check(L.osud_table_rows_apply(_lib.ptr(tg), rows, D, _lib.ptr(all_idx), _lib.ptr(all_rows), W, B, _lib.ptr(scratch), None))
        torch.cuda.synchronize()
        outs.append(tg.cpu())
    assert torch.equal(outs[0], outs[1])  # replicas bit-identical
    assert torch.equal(outs[0], want)     # == the rank-ordered dense sum, bit for bit (untouched rows stay zero)


def test_row_exchange_rejects_oversized_unions():
    L = _lib.lib()
    t = torch.zeros(16, 128, device=DEV)
    idx = torch.zeros(8 * 1024, dtype=torch.int64, device=DEV)
    rc = L.osud_table_rows_apply(_lib.ptr(t), 16, 128, _lib.ptr(idx), _lib.ptr(t), 8, 1024, _lib.ptr(idx), None)
    assert rc == _lib.ERR_ARG and "4096" in _lib.last_error()
