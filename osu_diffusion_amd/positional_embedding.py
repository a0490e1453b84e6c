"""Host-side sinusoidal embeddings (used when building context tensors, e.g. the distance
embedding inside ``c``).  Reference: positional_embedding.py:29-77.  The model's own token /
timestep embeddings are computed inside the native kernels, not here."""
import math

import torch


def timestep_embedding(t, dim, max_period=10000):
    """(N,) possibly fractional values -> (N, dim) as [cos | sin] of t * max_period^(-k/half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def offset_sequence_embedding(t, dim, max_period=10000):
    """(N, T) time offsets -> (N, T, dim)."""
    n, length = t.shape
    return timestep_embedding(t.reshape(-1), dim, max_period).reshape(n, length, dim)


def position_sequence_embedding(t, dim, max_period=10000):
    """(N, T, P) positions -> (N, T, P * dim)."""
    n, length, p = t.shape
    return timestep_embedding(t.reshape(-1), dim, max_period).reshape(n, length, p * dim)
