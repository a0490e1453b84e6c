"""Synthetic beatmap windows with the reference's tensor contract.

A training/sampling sample is ``((x, o, c), y)`` with x (2,T) positions / (512,384),
o (T) time offsets in ms, c (144,T) = [cos|sin embedding of the distance to the
previous object (128) ; one-hot object type (16)], y = beatmap class index
(reference: data_loading.py:146-203).  Real windows come from ``.osu`` files through
the third-party ``slider`` parser, which is out of scope here; the
bench and the tests use windows of the same shape, dtype, layout and value ranges
built from a seeded generator (SURVEY.md §8d "Synthetic inputs").
"""
from __future__ import annotations

import math

import torch


def _emb128(v: torch.Tensor) -> torch.Tensor:
    """cos|sin embedding with 64 frequencies 1e4^(-k/64) (positional_embedding.py:29-49)."""
    f = torch.exp(-math.log(10000) * torch.arange(0, 64, dtype=torch.float32) / 64)
    a = v[..., None].float() * f
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def synthetic_windows(n: int, seq_len: int, num_classes: int = 52670, seed: int = 0,
                      train_offsets: bool = True):
    """Return ((x (n,2,T), o (n,T), c (n,144,T)), y (n,)) float32 / int64 CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 2, seq_len, generator=g)
    gaps = torch.randint(50, 601, (n, seq_len), generator=g).float()
    o = torch.cumsum(gaps, dim=1)
    o = o - o[:, :1]
    if train_offsets:  # data_loading.py:198-200: + U(0,1) * 1e5 ms per window
        o = o + torch.rand(n, 1, generator=g) * 100000
    px = x * torch.tensor([512.0, 384.0]).view(1, 2, 1)
    prev = torch.roll(px, 1, 2)
    prev[:, 0, 0] = 256  # data_loading.py:146-151
    prev[:, 1, 0] = 192
    dist = torch.linalg.vector_norm(px - prev, ord=2, dim=1)  # (n,T)
    types = torch.randint(0, 16, (n, seq_len), generator=g)
    onehot = torch.nn.functional.one_hot(types, 16).float().transpose(1, 2)  # (n,16,T)
    c = torch.cat([_emb128(dist).transpose(1, 2), onehot], dim=1).contiguous()
    y = torch.randint(0, num_classes, (n,), generator=g)
    return (x.contiguous(), o.contiguous(), c), y


def banded_attn_mask(seq_len: int, window: int) -> torch.Tensor:
    """(T,T) bool, True = masked; reference sample.py:81-84 (query r may attend key i
    iff r - W < i <= r + W; slightly asymmetric, reproduced deliberately)."""
    m = torch.full((seq_len, seq_len), True, dtype=torch.bool)
    for i in range(seq_len):
        m[max(0, i - window): min(seq_len, i + window), i] = False
    return m


def randomize_zero_init(model, seed: int = 0, std: float = 0.02, pos_gain: float = 0.1):
    """Give a freshly initialised DiT non-degenerate weights for benches / smoke runs.

    The reference zero-initialises every adaLN modulation and the output projection
    (models.py:295-304), so a fresh model is the identity and outputs exactly 0.  This fills
    those tensors (and all biases) with N(0, std) from a seeded generator on the parameters'
    device and scales the position-feature columns of the first linear by `pos_gain`
    (see oracle.dit_oracle.seeded_state_dict for why).  There are no checkpoints to download."""
    dev = model.xoc_embedder.playfield_size.device
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("playfield_size"):
                continue
            if "adaLN" in name or name.startswith("final_layer") or name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g, device=dev) * std)
        model.xoc_embedder.mlp[0].weight[:, : model.in_channels * 128] *= pos_gain
    return model
