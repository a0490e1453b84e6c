"""Row sharding of the sampling path across ranks: the n variants of a sampling run are independent given
their own noise, so they are split contiguously over the ranks — each conditional row stays with its unconditional twin on the
same GPU — with no collective while sampling and one small gather of the final coordinates at the end.  (The reference is
single-GPU: sample.py:43,97-108.)"""
from __future__ import annotations

import math

import torch


def shard_rows(n: int, rank: int, world: int):
    """[lo, hi) of the n rows owned by `rank`: contiguous blocks of ceil(n / world), the rule train.py:165-169 uses for tracks."""
    per = int(math.ceil(n / float(world)))
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def gather_rows(rows: torch.Tensor, n_total: int, rank: int, world: int, group=None):
    """Collect every rank's (n_r, ...) rows on rank 0 in variant order; other ranks get None.  Ranks may own different
    numbers of rows (the last ones fewer, possibly none): shards are padded to the common block size for the all_gather."""
    if world == 1:
        return rows
    import torch.distributed as dist

    per = int(math.ceil(n_total / float(world)))
    pad = torch.zeros(per, *rows.shape[1:], dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    if rank != 0:
        return None
    out = []
    for r, part in enumerate(parts):
        lo, hi = shard_rows(n_total, r, world)
        out.append(part[: hi - lo])
    return torch.cat(out, 0)
