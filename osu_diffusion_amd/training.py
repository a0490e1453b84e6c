"""Training path on top of libosud.so.

Two entry levels, both running the native forward/backward kernels:

* ``loss.backward()`` compatibility: with grad enabled, ``DiT.forward`` goes through
  ``dit_forward_autograd`` (a ``torch.autograd.Function``), so the reference's training loop
  (train.py:243-261: ``diffusion.training_losses(model, ...)`` -> ``loss.backward()`` ->
  ``opt.step()``) runs unmodified against this package.
* ``NativeTrainer``: the whole step of train.py:243-261 fused natively — q_sample, forward,
  fused loss forward+backward, backward, gradient all-reduce over RCCL on ONE flat fp32 arena,
  AdamW + EMA in one pass — with the reference's checkpoint layout
  ``{"model", "ema", "opt", "scaler", "args"}`` (train.py:287-293).
"""
from __future__ import annotations

import copy
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib


# ------------------------------------------------------------------------------ sharding rules
def shard_range(data_start: int, data_end: int, rank: int, world_size: int):
    """Per-rank contiguous track range (train.py:165-169)."""
    per_rank = int(np.ceil((data_end - data_start) / float(world_size)))
    start = data_start + rank * per_rank
    return start, min(start + per_rank, data_end)


def worker_range(start: int, end: int, worker_id: int, num_workers: int):
    """Per-DataLoader-worker sub-range (data_loading.py:366-376)."""
    per_worker = int(math.ceil((end - start) / float(num_workers)))
    s = start + worker_id * per_worker
    return s, min(s + per_worker, end)


def allreduce_mean_(flat: torch.Tensor, group=None) -> float:
    """DDP's gradient averaging (train.py:152,257) on one flat buffer: SUM all-reduce, the 1/world
    factor is returned so the optimizer kernel applies it for free.  NCCL backend == RCCL."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 1.0
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)


# ------------------------------------------------------------------------------ flat arenas
class ParamArena:
    """Re-homes a module's parameters into ONE flat fp32 tensor (views keep the Parameter objects,
    names and order), plus a same-layout gradient arena bound to the native backward."""

    def __init__(self, model):
        self.model = model
        params = list(model.named_parameters())
        dev = params[0][1].device
        self.names = [n for n, _ in params]
        self.sizes = [p.numel() for _, p in params]
        # every tensor starts on a 16-byte boundary (4 fp32 elements): the 2-element playfield_size comes first, and unpadded it
        # left every later offset at 2 mod 4 -- shards, moments and EMA 8 bytes off the (aligned) scatter buffer, which put the
        # whole sharded optimizer step on the kernel's scalar path.  The pad elements stay zero (zero gradient: AdamW keeps them)
        self.align = 4
        starts, pos = [], 0
        for n in self.sizes:
            starts.append(pos)
            pos += -(-n // self.align) * self.align
        self.offsets = np.asarray(starts + [pos], dtype=np.int64)
        self.total = int(pos)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for (name, p), off, n in zip(params, self.offsets[:-1], self.sizes):
            view = self.flat[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
        model._uploaded = {}

    def view(self, flat, name):
        i = self.names.index(name)
        shape = dict(self.model.named_parameters())[name].shape
        return flat[self.offsets[i]:self.offsets[i] + self.sizes[i]].view(shape)

    def grad_views(self):
        shapes = {n: p.shape for n, p in self.model.named_parameters()}
        return {n: self.grads[o:o + s].view(shapes[n]) for n, o, s in zip(self.names, self.offsets[:-1], self.sizes)}

    def bind(self, handle):
        L = _lib.lib()
        for n, g in self.grad_views().items():
            if n.endswith("playfield_size"):
                continue
            _lib.check(L.osud_dit_bind_grad(handle, n.encode(), _lib.ptr(g)))

    def frozen_range(self):
        i = self.names.index("xoc_embedder.playfield_size")
        return int(self.offsets[i]), int(self.offsets[i] + self.sizes[i])


def _arena_of(model) -> ParamArena:
    a = getattr(model, "_arena", None)
    if a is None or a.flat.device != model._device():
        a = ParamArena(model)
        model._arena = a
        model._arena_bound = None
    return a


def _train_ready(model, N, T):
    """Handle with training workspaces, uploaded parameters and bound gradient buffers."""
    arena = _arena_of(model)
    h = model.native_handle()
    with torch.cuda.device(model._device()):
        _lib.check(_lib.lib().osud_dit_reserve(h, int(N), int(T), 1))
        if getattr(model, "_arena_bound", None) != (id(h), h.value):
            arena.bind(h)
            model._arena_bound = (id(h), h.value)
    return h, arena


def native_forward_train(model, x, t, o, c, y):
    N, T = model._check_inputs(x, t, o, c, y, None)
    h, _ = _train_ready(model, N, T)
    x, t, o, c, y, _ = model._prep(x, t, o, c, y, None)
    out = torch.empty(N, model.out_channels, T, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().osud_dit_forward_train(h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(o), _lib.ptr(c), _lib.ptr(y), N, T,
                                                     _lib.ptr(out), _lib.stream_ptr(x.device)))
    model._train_keep = (x, t, o, c, y)  # the backward pass reads y again
    return out


def native_backward(model, dout, phases=None):
    """Runs the native backward of the last training forward (all of it, or phases [lo, hi]); gradients
    land in model._arena.grads."""
    h = model._handle
    dout = dout.contiguous().float()
    with torch.cuda.device(dout.device):
        if phases is None:
            _lib.check(_lib.lib().osud_dit_backward(h, _lib.ptr(dout), _lib.stream_ptr(dout.device)))
        else:
            _lib.check(_lib.lib().osud_dit_backward_phases(h, _lib.ptr(dout), int(phases[0]), int(phases[1]),
                                                           _lib.stream_ptr(dout.device)))
    return dout


def overlap_slices(arena, depth):
    """Element ranges of the flat gradient arena in the order they become final during the phased backward:
    ("final", depth, lo, hi) right after phase 0 (final linear + its adaLN pair), ("block", l, lo, hi) right after block l's
    phase — its 8 attention/MLP tensors and its adaLN pair are contiguous (the library differentiates the adaLN slice of a
    block inside that block's phase when it is driven phase by phase) — then what needs the last phase: ("tail", ...) =
    the embedders, and ("table", ...) = the class table, whose gradient is row-sparse and is exchanged as rows
    (exchange_table_rows) instead of densely."""
    off = {n: (int(o), int(o + s)) for n, o, s in zip(arena.names, arena.offsets[:-1], arena.sizes)}
    blocks = []
    for l in range(depth):
        lo = off[f"blocks.{l}.attn.in_proj_weight"][0]
        hi = off[f"blocks.{l}.adaLN_modulation.1.bias"][1]
        blocks.append(("block", l, lo, hi))
    t_lo, t_hi = off["y_embedder.embedding_table.weight"]
    first_block = off["blocks.0.attn.in_proj_weight"][0]
    tail = [("tail", -1, 0, t_lo), ("table", -1, t_lo, t_hi)]
    if t_hi < first_block:
        tail.append(("tail", -1, t_hi, first_block))
    tail.append(("final", depth, off["final_layer.linear.weight"][0], arena.total))
    covered = sorted((lo, hi) for _, _, lo, hi in blocks + tail)
    assert covered[0][0] == 0 and covered[-1][1] == arena.total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    return blocks, tail


def exchange_table_rows(table_grad, labels, group=None):
    """SUM the class-table gradient over ranks by exchanging only the touched rows.

    The table gradient (52 671 x D, 162 MB for DiT-B — a quarter of the whole gradient) is non-zero only in the rows of
    the step's (post-dropout) labels, at most B per rank (models.py:56-74; the reference's DDP all-reduces it densely).
    Each rank contributes its labels sorted, with one copy of every touched row (duplicates zeroed), via all_gather
    (2 x W x B x D floats in total); every rank then rebuilds the touched rows by adding the contributions in RANK ORDER,
    so replicas stay bit-identical (a scatter with atomics in arbitrary order would not guarantee that)."""
    import torch.distributed as dist

    W = dist.get_world_size(group)
    if table_grad.is_cuda and labels.numel() * W <= 4096 and W <= 64:  # (osud_table_rows_apply merges at most 64 rank lists)
        # GPU: two native launches per side (csrc/exchange.hip) instead of ~25 tensor-library ones -- same sums, same order
        L = _lib.lib()
        rows_n, D = table_grad.shape
        B = labels.numel()
        lab = labels.to(device=table_grad.device, dtype=torch.int64).reshape(-1).contiguous()
        ys = torch.empty(B, dtype=torch.int64, device=table_grad.device)
        rows = torch.empty(B, D, dtype=torch.float32, device=table_grad.device)
        st = _lib.stream_ptr(table_grad.device)
        with torch.cuda.device(table_grad.device):
            _lib.check(L.osud_table_rows_pack(_lib.ptr(table_grad), rows_n, D, _lib.ptr(lab), B, _lib.ptr(ys), _lib.ptr(rows), st))
        all_idx = torch.empty(W, B, dtype=torch.int64, device=table_grad.device)
        all_rows = torch.empty(W, B, D, dtype=torch.float32, device=table_grad.device)
        dist.all_gather_into_tensor(all_idx, ys, group=group) if dist.get_backend(group) == "nccl" else dist.all_gather(list(all_idx.unbind(0)), ys, group=group)
        dist.all_gather_into_tensor(all_rows, rows, group=group) if dist.get_backend(group) == "nccl" else dist.all_gather(list(all_rows.unbind(0)), rows, group=group)
        scratch = torch.empty(W * B + 1, dtype=torch.int64, device=table_grad.device)
        with torch.cuda.device(table_grad.device):
            _lib.check(L.osud_table_rows_apply(_lib.ptr(table_grad), rows_n, D, _lib.ptr(all_idx), _lib.ptr(all_rows), W, B,
                                               _lib.ptr(scratch), _lib.stream_ptr(table_grad.device)))
        return table_grad
    ys, _ = torch.sort(labels.to(torch.int64).reshape(-1))
    first = torch.ones_like(ys, dtype=torch.bool)
    first[1:] = ys[1:] != ys[:-1]
    rows = table_grad[ys] * first.unsqueeze(1).to(table_grad.dtype)
    all_idx = [torch.empty_like(ys) for _ in range(W)]
    all_rows = [torch.empty_like(rows) for _ in range(W)]
    dist.all_gather(all_idx, ys, group=group)
    dist.all_gather(all_rows, rows, group=group)
    for idx in all_idx:
        table_grad.index_fill_(0, idx, 0.0)
    for idx, r in zip(all_idx, all_rows):  # within one rank's list a row index carries at most one non-zero row
        table_grad.index_add_(0, idx, r)
    return table_grad


def backward_with_overlapped_allreduce(model, dout, group=None, force=False, on_blocks_reduced=None, comm=None, stub=False):
    """Backward in phases; each block's gradient slice is SUM-all-reduced (async, RCCL's own stream) as soon
    as its phase is enqueued, overlapping the exchange with the remaining backward compute — the role of
    DDP's bucketed reducer (train.py:152,257).  `on_blocks_reduced()` (optional) is called once every block slice has
    been reduced, while the tail exchange is still in flight (the optimizer uses that window).
    `comm` (a comm.NativeComm): the slices travel through the C ABI's osud_allreduce_grads instead of torch.distributed.
    `stub`: the same phased schedule with every collective left out (measurement only: bench.py's exposed-communication figure is
    the step time with the exchange minus the step time of this).
    Returns the 1/world factor for the optimizer."""
    import torch.distributed as dist

    inited = dist.is_available() and dist.is_initialized()
    active = inited and (dist.get_world_size(group) > 1 or force)  # force: exercise RCCL even on one rank
    if not (active or force):
        native_backward(model, dout)
        return 1.0
    arena, depth = model._arena, model.depth
    blocks, tail = overlap_slices(arena, depth)
    world_scale = 1.0 / dist.get_world_size(group) if active else 1.0
    if stub:
        active = False
    if comm is not None and active:  # the library's own RCCL communicator (C ABI: osud_allreduce_grads on its side stream)
        reduce = lambda t: comm.all_reduce_(t, async_op=True)  # noqa: E731
    else:
        reduce = (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)) if active else (lambda t: None)
    handles = []
    dout = native_backward(model, dout, phases=(0, 0))
    _, _, f_lo, f_hi = next(s for s in tail if s[0] == "final")
    handles.append(reduce(arena.grads[f_lo:f_hi]))
    for p in range(1, depth + 1):
        native_backward(model, dout, phases=(p, p))
        _, _, lo, hi = blocks[depth - p]
        handles.append(reduce(arena.grads[lo:hi]))
    native_backward(model, dout, phases=(depth + 1, depth + 1))
    tail_handles = []
    for kind, _, lo, hi in tail:
        if kind == "tail":
            tail_handles.append(reduce(arena.grads[lo:hi]))
    for h in handles:
        if h is not None:
            h.wait()
    if on_blocks_reduced is not None:
        on_blocks_reduced([(lo, hi) for _, _, lo, hi in blocks] + [(f_lo, f_hi)])
    if comm is not None:
        # the library's communicator and torch's process group are two RCCL communicators on one device: their kernels must not be
        # in flight together (ranks could schedule them in different orders and deadlock), so the tail reductions are joined
        # BEFORE torch's group runs the row all-gathers below
        for h in tail_handles:
            if h is not None:
                h.wait()
        tail_handles = []
    if active:
        _, _, t_lo, t_hi = next(s for s in tail if s[0] == "table")
        rows = dict(model.named_parameters())["y_embedder.embedding_table.weight"].shape[0]
        exchange_table_rows(arena.grads[t_lo:t_hi].view(rows, -1), model._train_keep[4], group)
    for h in tail_handles:
        if h is not None:
            h.wait()
    return world_scale


def select_exchange_schedule(make_trainer, batches, group=None, steps=3, warmup=1, device=None):
    """world > 1: which gradient exchange this job should run -- decided by MEASUREMENT at start-up, on this node, this RCCL build and
    this topology, instead of by RCCL's per-call algorithm choice: `steps` training steps each of `allreduce` (per-slice all-reduces under
    the phased backward) and `zero1` (reduce-scatter -> AdamW / EMA on the own shard -> all-gather of the masters), each behind `warmup`
    untimed steps and bracketed by barriers; a schedule's time is the MAX over ranks, so that every rank keeps the same one.  On xGMI a ring
    all-reduce moves 2 (W - 1) / W of the payload over ONE link per direction, the mesh form payload / W per link over all of them
    (5.98 ms vs 0.85 ms per DiT-B step at 8 ranks, unoverlapped) -- the first multi-GPU run must not land on the
    slow one by default.  `make_trainer(shard_optimizer=...)` builds a trainer (the steps run on the caller's model: a few real
    optimisation steps, like a warm-up); `batches`: [((x, o, c), y), ...].
    Returns {"name": "allreduce" | "zero1", "allreduce_ms": ..., "zero1_ms": ..., "steps": ..., "world_size": ...}; world 1:
    {"name": "allreduce", "skipped": ...} without running anything."""
    import time

    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return {"name": "allreduce", "skipped": "one rank: nothing to exchange"}
    world = dist.get_world_size(group)
    on_gpu = device is not None and torch.device(device).type == "cuda"

    def fence():
        dist.barrier(group=group)
        if on_gpu:
            torch.cuda.synchronize(device)

    out = {"steps": int(steps), "world_size": world}
    for name, shard in (("allreduce", False), ("zero1", True)):
        tr = None
        try:
            tr = make_trainer(shard_optimizer=shard)
            for i in range(warmup):
                (x, o, c), y = batches[i % len(batches)]
                tr.step(x, o, c, y)
            tr.finish_exchange()
            fence()
            t0 = time.perf_counter()
            for i in range(steps):
                (x, o, c), y = batches[i % len(batches)]
                tr.step(x, o, c, y)
            tr.finish_exchange()
            fence()
            ms = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64, device=device if on_gpu else "cpu")
            dist.all_reduce(ms, op=dist.ReduceOp.MAX, group=group)
            out[name + "_ms"] = round(float(ms.item()), 3)
        except Exception as e:  # noqa: BLE001
            # a schedule that does not run on this node (raised on every rank alike: a missing library, an unsupported collective) must not take
            # the job with it: the other one is kept.  (A failure on ONE rank leaves the others inside a collective: the caller's watchdog.)
            if name == "allreduce":
                raise
            out[name + "_error"] = f"{type(e).__name__}: {e}"[:200]
        finally:
            del tr
            if on_gpu:
                torch.cuda.empty_cache()
    # (ties and near-ties -- within 2 % -- keep the all-reduce: same results to the last bit as the single-call reference path)
    out["name"] = "zero1" if "zero1_ms" in out and out["zero1_ms"] < 0.98 * out["allreduce_ms"] else "allreduce"
    return out


# ------------------------------------------------------------------------------ ZeRO-1 exchange (reduce-scatter / all-gather)
def shard_plan(lo, hi, world, align=4):
    """How one contiguous gradient slice [lo, hi) of the flat arena is exchanged when the optimizer is sharded: a bulk of
    world x per elements (per a multiple of `align`: 16-byte aligned shards) that is reduce-scattered -- rank r owns
    [lo + r*per, lo + (r+1)*per) -- and a remainder of fewer than world x align elements that is all-reduced and updated by every
    rank.  Returns (per, bulk_hi)."""
    per = ((hi - lo) // world) // align * align
    return per, lo + per * world


def _reduce_scatter_sum(out, inp, group, comm=None):
    """SUM reduce-scatter of `inp` (world x out.numel()) into `out`; async handle.  RCCL: one reduce_scatter_tensor; backends
    without it (gloo, used by the CPU / one-GPU tests): all-reduce, then keep the own shard.  `comm`: through the C ABI."""
    import torch.distributed as dist

    if comm is not None:
        return comm.reduce_scatter(out, inp, async_op=True), None
    if dist.get_backend(group) == "nccl":
        return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group, async_op=True), None
    h = dist.all_reduce(inp, op=dist.ReduceOp.SUM, group=group, async_op=True)
    r, n = dist.get_rank(group), out.numel()
    return h, (lambda: out.copy_(inp[r * n:(r + 1) * n]))


def _all_gather_into(full, shard, group, comm=None):
    """all-gather equal shards into `full` (shard may be full's own slice: in place); async handle + finisher."""
    import torch.distributed as dist

    if comm is not None:
        return comm.all_gather(full, shard, async_op=True), None
    if dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(full, shard, group=group, async_op=True), None
    W = dist.get_world_size(group)
    parts = [torch.empty_like(shard) for _ in range(W)]
    h = dist.all_gather(parts, shard.clone(), group=group, async_op=True)
    n = shard.numel()

    def finish():
        for r, part in enumerate(parts):
            full[r * n:(r + 1) * n].copy_(part)
    return h, finish


class _DiTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, t, o, c, y, *params):
        ctx.model = model
        ctx.n_params = len(params)
        return native_forward_train(model, x, t, o, c, y)

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        native_backward(model, dout)
        gv = model._arena.grad_views()
        grads = []
        for name, p in model.named_parameters():
            grads.append(gv[name].clone() if p.requires_grad else None)
        return (None, None, None, None, None, None, *grads)


def dit_forward_autograd(model, x, t, o, c, y, attn_mask):
    if attn_mask is not None:
        raise _lib.NativeError("the native training path has no attention mask (the reference trains without one)")
    model.native_handle()  # raises if the module is not on a GPU
    return _DiTFunction.apply(model, x, t, o, c, y, *model.parameters())


# ------------------------------------------------------------------------------ fused trainer
def _split_at(ranges, cut):
    """`ranges` with every range split at the boundaries of `cut` (so that no piece straddles it)."""
    out = []
    for lo, hi in ranges:
        pts = [lo] + [c for c in cut if lo < c < hi] + [hi]
        out += list(zip(pts[:-1], pts[1:]))
    return out


def _complement(ranges, total):
    """Element ranges of [0, total) not covered by `ranges`."""
    out, pos = [], 0
    for lo, hi in sorted(ranges):
        if lo > pos:
            out.append((pos, lo))
        pos = max(pos, hi)
    if pos < total:
        out.append((pos, total))
    return out


class NativeTrainer:
    """The step body of train.py:243-261, natively.  `model` is trained in place; `ema` (a deepcopy
    made here unless given) tracks it with decay 0.9999."""

    def __init__(self, model, diffusion, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_decay=0.9999,
                 ema=None, group=None, broadcast_init=True, shard_optimizer=False, wire_dtype=None, force_phased=False,
                 overlap_gather=True, native_comm=False, stub_exchange=False):
        """shard_optimizer: ZeRO-1 exchange (reduce-scatter / sharded AdamW + EMA / all-gather) instead of the all-reduce.
        wire_dtype=torch.bfloat16: gradients rounded to bf16 for the reduce-scatter.  overlap_gather: with the sharded optimizer,
        the master all-gather runs under the next step's forward.  native_comm: the exchange goes through the library's own RCCL
        communicator (C ABI) instead of torch.distributed.  force_phased: run the multi-GPU schedule (phased backward, per-slice
        collectives) even on one rank -- tests and one-GPU rehearsals of the distributed path."""
        from .diffusion import gaussian_diffusion as gd

        assert diffusion.model_mean_type == gd.ModelMeanType.EPSILON
        assert diffusion.model_var_type == gd.ModelVarType.LEARNED_RANGE
        assert diffusion.loss_type in (gd.LossType.L1, gd.LossType.MSE), "native loss kernel: L1 or MSE (+vb)"
        self.use_l1 = int(diffusion.loss_type == gd.LossType.L1)
        self.model, self.diffusion, self.group = model, diffusion, group
        self.lr, self.betas, self.eps, self.weight_decay, self.ema_decay = lr, betas, eps, weight_decay, ema_decay
        self.arena = _arena_of(model)
        self.ema = ema if ema is not None else copy.deepcopy(model)
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self.ema_arena = ParamArena(self.ema)
        self.ema.eval()
        self.exp_avg = torch.zeros_like(self.arena.flat)
        self.exp_avg_sq = torch.zeros_like(self.arena.flat)
        # torch's AdamW keeps a step counter PER PARAMETER.  Two counters cover every reference recipe: the trunk's, and the
        # class table's own absolute one -- ahead of the trunk after --embed-only-epochs (train.py:223-241: only the table
        # trains while `embed_only` is set), behind it after --relearn-embeds (train.py:212-215 deletes optimizer state 7, so
        # the table restarts at step 1 with fresh moments)
        self.step_count = 0
        self.table_step = 0
        self.embed_only = False
        self.force_phased = bool(force_phased)
        self.stub_exchange = bool(stub_exchange)  # measurement only: the multi-GPU schedule with its collectives left out
        self._side = None
        # ZeRO-1 between the two halves of the gradient exchange: every finished slice is reduce-SCATTERED
        # (rank r receives the sum of its 1/world shard), AdamW + EMA run on that shard only (1/world of the 6.1 GB the optimizer
        # streams per step), and the updated master weights are all-GATHERED.  Same bytes on the wire as the all-reduce, same
        # results (the sum of a shard is formed once instead of world times).
        # wire_dtype=torch.bfloat16 halves the reduce-scatter bytes: gradients are rounded to bf16 for the exchange and summed by
        # RCCL in bf16 (reduced precision of the AVERAGED gradient, 2^-9 relative: opt-in).
        self.shard_optimizer = bool(shard_optimizer)
        self.wire_dtype = wire_dtype
        self._shard_buf = None
        self._ema_stale = False
        # sharded optimizer: gather the updated master shards and re-pack them block by block UNDER the next step's forward
        # (False: gather everything, then re-pack, then go on)
        self.overlap_gather = bool(overlap_gather)
        self._gate_events = []
        self._gates_set = False
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            # collectives will share the compute units with the backward: let multi-round GEMM launches queue their tiles
            _lib.check(_lib.lib().osud_set_gemm_dynamic_tiles(1))
        # native_comm: the gradient / parameter exchange goes through the library's own RCCL communicator (C ABI: osud_comm_init,
        # osud_allreduce_grads, ...), bootstrapped once over torch's process group; the class-table row exchange and the scalar
        # bookkeeping stay on torch.distributed.  (world 1 with force_phased exercises every call.)
        self.comm = None
        if native_comm and dist.is_available() and dist.is_initialized():
            from .comm import NativeComm

            self.comm = NativeComm.from_torch_distributed(group, device=self.arena.flat.device)
        if broadcast_init and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if self.comm is not None:
                self.comm.broadcast_(self.arena.flat, 0)
            else:
                dist.broadcast(self.arena.flat, 0, group=group)  # DDP ctor: rank 0's init wins (train.py:152)
        self.ema_arena.flat.copy_(self.arena.flat)  # update_ema(ema, model, decay=0), train.py:194-198
        self._tmap = torch.from_numpy(np.asarray(diffusion._model_timestep_map)).to(self.arena.flat.device)

    def step(self, x, o, c, y, t=None, noise=None, drop_ids=None, loss_weights=None):
        """One optimisation step on a batch of windows; returns the (3, B) tensor [l1|mse; vb; loss]
        (device tensor, no host sync).  `loss_weights` (B,): per-sample factors of the objective, as a schedule sampler
        returns them with its timesteps (diffusion/timestep_sampler.py: loss = mean(loss_b * w_b)); the returned terms stay
        unweighted."""
        model, d = self.model, self.diffusion
        dev = self.arena.flat.device
        L = _lib.lib()
        B, _, T = x.shape
        x0 = x.to(dev, torch.float32).contiguous()
        if t is None:
            t = torch.randint(0, d.num_timesteps, (B,), device=dev)  # train.py:248
        t = t.to(dev, torch.int64).contiguous()
        if noise is None:
            noise = torch.randn_like(x0)  # gaussian_diffusion.py:799-800
        noise = noise.to(dev, torch.float32).contiguous()
        y = y.to(dev, torch.int64)
        if drop_ids is not None:
            drop_ids = drop_ids.to(dev)
        if model.training and model.y_embedder.dropout_prob > 0:
            y = model.y_embedder.token_drop(y, drop_ids)  # models.py:56-72
        elif drop_ids is not None:
            y = model.y_embedder.token_drop(y, drop_ids)
        st = _lib.stream_ptr(dev)
        with torch.cuda.device(dev):
            x_t = torch.empty_like(x0)
            _lib.check(L.osud_q_sample(d._sched.handle, _lib.ptr(x0), _lib.ptr(t), _lib.ptr(noise), B, T, _lib.ptr(x_t), st))
            out = native_forward_train(model, x_t, self._tmap[t], o, c, y)
            terms = torch.empty(3, B, device=dev, dtype=torch.float32)
            dout = torch.empty_like(out)
            _lib.check(L.osud_train_loss(d._sched.handle, self.use_l1, _lib.ptr(out), _lib.ptr(x0), _lib.ptr(x_t), _lib.ptr(noise),
                                         _lib.ptr(t), B, T, _lib.ptr(terms), _lib.ptr(dout), st))
            if loss_weights is not None:  # d(mean_b w_b loss_b)/d(out): the fused loss kernel's rows scale with their sample's weight
                dout.mul_(loss_weights.to(dev, torch.float32).view(B, 1, 1))
            if self.embed_only:
                self._embed_only_update(dout, y)
                return terms
            done = []

            def early(ranges, _self=self):  # block slices are final: update them while the tail exchange is in flight
                import torch.distributed as dist
                _self._advance()
                _self._adamw(ranges, 1.0 / dist.get_world_size(_self.group) if dist.is_initialized() else 1.0)
                done.extend(ranges)

            import torch.distributed as dist
            single = not (dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1)
            if self.shard_optimizer and (not single or self.force_phased) and dist.is_available() and dist.is_initialized():
                self._backward_sharded(dout)
                return terms
            scale = backward_with_overlapped_allreduce(model, dout, self.group, force=self.force_phased, on_blocks_reduced=early,
                                                       comm=self.comm, stub=self.stub_exchange)
            if done:
                self._adamw(_complement(done, self.arena.total), scale)
                self._refresh()
            else:
                self.optimizer_step(scale)
        return terms

    def _backward_sharded(self, dout):
        """Phased backward with the sharded optimizer in the middle of the gradient exchange:
            phase done -> async reduce-scatter of that slice (own shard lands in self._shard_buf)
            all phases done -> small tail all-reduced, class-table rows exchanged (both updated by every rank)
            AdamW + EMA on the own shards (gradient = the scattered sum x 1/world) and on the replicated parts
            async all-gather of the updated master shards, slice by slice -> re-pack of the low-precision copies."""
        import torch.distributed as dist

        model, arena, group = self.model, self.arena, self.group
        W, r = dist.get_world_size(group), dist.get_rank(group)
        depth = model.depth
        blocks, tail = overlap_slices(arena, depth)
        _, _, f_lo, f_hi = next(s for s in tail if s[0] == "final")
        order = [(f_lo, f_hi)] + [(blocks[depth - p][2], blocks[depth - p][3]) for p in range(1, depth + 1)]  # as the phases finish
        plans = [(lo, hi) + shard_plan(lo, hi, W) for lo, hi in order]
        need = sum(per for _, _, per, _ in plans)
        wire = self.wire_dtype
        if self._shard_buf is None or self._shard_buf.numel() < need or self._shard_buf.dtype != (wire or torch.float32):
            self._shard_buf = torch.empty(need, dtype=wire or torch.float32, device=arena.flat.device)
        pending, off = [], 0
        dout = native_backward(model, dout, phases=(0, 0))
        for p, (lo, hi, per, bulk_hi) in enumerate(plans):
            if p > 0:
                native_backward(model, dout, phases=(p, p))
            out = self._shard_buf[off:off + per]
            if per > 0:
                src = arena.grads[lo:bulk_hi] if wire is None else arena.grads[lo:bulk_hi].to(wire)
                pending.append(_reduce_scatter_sum(out, src, group, self.comm) + (src,))  # (handle, finisher, keep-alive)
            if bulk_hi < hi:
                pending.append((self._allreduce_async(arena.grads[bulk_hi:hi]), None, None))
            off += per
        native_backward(model, dout, phases=(depth + 1, depth + 1))
        for kind, _, lo, hi in tail:
            if kind == "tail":
                pending.append((self._allreduce_async(arena.grads[lo:hi]), None, None))
        for h, fin, _keep in pending:
            h.wait()
            if fin is not None:
                fin()
        _, _, t_lo, t_hi = next(s for s in tail if s[0] == "table")
        rows = dict(model.named_parameters())["y_embedder.embedding_table.weight"].shape[0]
        exchange_table_rows(arena.grads[t_lo:t_hi].view(rows, -1), model._train_keep[4], group)
        # ---- optimizer: own shards from the scatter buffer, everything else replicated.  Every update is enqueued before the first
        # gather: collectives of one communicator complete in issue order, so the gathers below are issued in FORWARD order (block 0
        # .. L-1, then the final layer) and phase p's weights depend on the gathers up to p only
        self._advance()
        scale, off, own, shards = 1.0 / W, 0, [], {}
        phase_of = [depth + 1] + [depth - p + 1 for p in range(1, depth + 1)]  # plans[i] -> phase: final layer, then blocks L-1 .. 0
        for i, (lo, hi, per, bulk_hi) in enumerate(plans):
            if per > 0:
                a, b = lo + r * per, lo + (r + 1) * per
                g = self._shard_buf[off:off + per]
                self._adamw_range(a, b, g if wire is None else g.float(), scale)
                own.append((lo, bulk_hi))
                shards[phase_of[i]] = (lo, bulk_hi, a, b)
            off += per
        self._adamw(_complement(own, arena.total), scale)
        self._ema_stale = True  # every rank's EMA is current only on its own shards (and the replicated parts)
        forward_order = list(range(1, depth + 1)) + [depth + 1]
        by_phase = {}
        for ph in forward_order:
            if ph in shards:
                lo, bulk_hi, a, b = shards[ph]
                by_phase[ph] = _all_gather_into(arena.flat[lo:bulk_hi], arena.flat[a:b], group, self.comm)
        if not self.overlap_gather:
            for h, fin in by_phase.values():
                h.wait()
                if fin is not None:
                    fin()
            self._refresh()
            return
        # ---- the all-gather of the updated masters under the NEXT step's forward: each gather's completion and the re-pack of its
        # phase go to a side stream, and the next forward waits per phase (osud_dit_forward_gate): block 0's kernels start as soon as
        # block 0's weights are in place while the later blocks' shards are still on the wire.  Phase 0 (embedders, conditioning path,
        # class table: replicated, updated by every rank above) is re-packed on the compute stream right away.
        dev = arena.flat.device
        L_ = _lib.lib()
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        h_model = self.model._handle
        with torch.cuda.device(dev):
            _lib.check(L_.osud_dit_refresh_phases(h_model, 0, 0, _lib.stream_ptr(dev)))
        self._side.wait_stream(main)  # the optimizer's writes (own shards + replicated parts) precede every re-pack
        self._gate_events = []
        with torch.cuda.stream(self._side):
            for ph in forward_order:
                g = by_phase.get(ph)
                if g is not None:
                    h, fin = g
                    h.wait()  # (orders the side stream behind the collective; no host sync)
                    if fin is not None:
                        fin()
                with torch.cuda.device(dev):
                    _lib.check(L_.osud_dit_refresh_phases(h_model, ph, ph, C.c_void_p(self._side.cuda_stream)))
                ev = torch.cuda.Event()
                ev.record(self._side)
                self._gate_events.append(ev)  # (kept alive until the gates are cleared: finish_exchange / the next exchange)
                _lib.check(L_.osud_dit_forward_gate(h_model, ph, C.c_void_p(ev.cuda_event)))
        self._gates_set = True
        self.ema._uploaded = {}  # its masters changed behind torch's back

    def _allreduce_async(self, t):
        import torch.distributed as dist

        if self.comm is not None:
            return self.comm.all_reduce_(t, async_op=True)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _adamw_range(self, lo, hi, grads, grad_scale):
        """AdamW + EMA on arena elements [lo, hi) with the gradient taken from `grads` (hi - lo elements, e.g. a scattered shard)."""
        a, dev = self.arena, self.arena.flat.device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().osud_adamw_ema_step(
                _lib.ptr(a.flat[lo:hi]), _lib.ptr(grads), _lib.ptr(self.exp_avg[lo:hi]), _lib.ptr(self.exp_avg_sq[lo:hi]),
                _lib.ptr(self.ema_arena.flat[lo:hi]), hi - lo, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                max(self.step_count, 1), self.ema_decay, 0, 0, float(grad_scale), _lib.stream_ptr(dev)))

    def finish_exchange(self):
        """Order the current stream behind a master all-gather / re-pack still running on the side stream (sharded optimizer with
        the gather under the next forward).  The next step does this per block through the forward's gates; anything else that
        reads the masters or the packed weights -- checkpoints, evaluation with the model, tests -- calls this first."""
        if self._side is not None:
            torch.cuda.current_stream(self.arena.flat.device).wait_stream(self._side)
        self._clear_gates()

    def _clear_gates(self):
        """The forward gates hold raw event handles owned by self._gate_events: once the stream is ordered behind the side stream
        (or the trainer goes away) they are taken out of the model handle, so that no later forward -- evaluation, sampling, another
        stream -- waits on an event that has been destroyed."""
        if getattr(self, "_gates_set", False):
            h = getattr(self.model, "_handle", None)
            if h is not None:
                L_ = _lib.lib()
                for ph in range(0, self.model.depth + 2):
                    L_.osud_dit_forward_gate(h, ph, None)
            self._gates_set = False
        self._gate_events = []

    def __del__(self):
        try:
            if getattr(self, "_gates_set", False):
                if self._side is not None:
                    self._side.synchronize()
                self._clear_gates()
        except Exception:  # interpreter shutdown: the library may be gone already
            pass

    def sync_sharded_state(self):
        """Sharded optimizer: bring the moments and the EMA of every shard to every rank (checkpoints and EMA evaluation need the
        whole state; during training nobody reads the other ranks' shards)."""
        import torch.distributed as dist

        self.finish_exchange()

        if not (self.shard_optimizer and self._ema_stale and dist.is_available() and dist.is_initialized()):
            return
        W, r = dist.get_world_size(self.group), dist.get_rank(self.group)
        blocks, tail = overlap_slices(self.arena, self.model.depth)
        _, _, f_lo, f_hi = next(s for s in tail if s[0] == "final")
        for lo, hi in [(f_lo, f_hi)] + [(b[2], b[3]) for b in blocks]:
            per, bulk_hi = shard_plan(lo, hi, W)
            if per == 0:
                continue
            for buf in (self.exp_avg, self.exp_avg_sq, self.ema_arena.flat):
                h, fin = _all_gather_into(buf[lo:bulk_hi], buf[lo + r * per:lo + (r + 1) * per], self.group, self.comm)
                h.wait()
                if fin is not None:
                    fin()
        self._ema_stale = False
        self.ema._uploaded = {}

    def _table_range(self):
        i = self.arena.names.index("y_embedder.embedding_table.weight")
        return int(self.arena.offsets[i]), int(self.arena.offsets[i] + self.arena.sizes[i])

    def _embed_only_update(self, dout, labels):
        """Frozen trunk (train.py:223-225, requires_grad_non_embed): the backward runs, but only the class table is
        exchanged (as rows) and updated; every other parameter gets just its EMA update."""
        import torch.distributed as dist

        native_backward(self.model, dout)
        t_lo, t_hi = self._table_range()
        scale = 1.0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            rows = dict(self.model.named_parameters())["y_embedder.embedding_table.weight"].shape[0]
            exchange_table_rows(self.arena.grads[t_lo:t_hi].view(rows, -1), self.model._train_keep[4], self.group)
            scale = 1.0 / dist.get_world_size(self.group)
        self.table_step += 1
        self._adamw([(0, t_lo), (t_hi, self.arena.total)], 1.0, frozen=True)
        self._adamw([(t_lo, t_hi)], scale)
        self._refresh()

    def _adamw(self, ranges, grad_scale, frozen=False):
        """AdamW + EMA on element ranges of the arenas (step_count already advanced); frozen: EMA only."""
        a, dev = self.arena, self.arena.flat.device
        f_lo, f_hi = a.frozen_range()
        t_lo, t_hi = self._table_range()
        L = _lib.lib()
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            for lo, hi in _split_at(ranges, (t_lo, t_hi)):
                if hi <= lo:
                    continue
                fl, fh = max(f_lo, lo) - lo, min(f_hi, hi) - lo  # frozen playfield_size, relative to this range
                if fh <= fl:
                    fl = fh = 0
                if frozen:
                    fl, fh = 0, hi - lo
                step = self.table_step if (lo >= t_lo and hi <= t_hi) else self.step_count
                _lib.check(L.osud_adamw_ema_step(_lib.ptr(a.flat[lo:hi]), _lib.ptr(a.grads[lo:hi]), _lib.ptr(self.exp_avg[lo:hi]),
                                                 _lib.ptr(self.exp_avg_sq[lo:hi]), _lib.ptr(self.ema_arena.flat[lo:hi]), hi - lo,
                                                 self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                                 max(step, 1), self.ema_decay, fl, fh, float(grad_scale), st))

    def _refresh(self):
        dev = self.arena.flat.device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().osud_dit_refresh(self.model._handle, _lib.stream_ptr(dev)))
        self.ema._uploaded = {}  # its masters changed behind torch's back

    def _advance(self):
        """One optimizer step begins: every parameter's AdamW step counter moves on."""
        self.step_count += 1
        self.table_step += 1

    @property
    def table_extra_steps(self):  # how far the class table's counter runs ahead of (> 0) or behind (< 0) the trunk's
        return self.table_step - self.step_count

    def optimizer_step(self, grad_scale=1.0):
        """AdamW(lr, betas, eps, wd) + EMA (train.py:258-261) + re-pack of the low-precision copies."""
        self._advance()
        self._adamw([(0, self.arena.total)], grad_scale)
        self._refresh()

    # ---- checkpoint layout of train.py:287-293 ------------------------------------------------
    def opt_state_dict(self):
        """torch.optim.AdamW-format state: parameter i = i-th entry of model.parameters() (so the frozen
        playfield_size is index 0 with no state and the class table is index 7, train.py:212-215)."""
        state = {}
        for i, name in enumerate(self.arena.names):
            if name.endswith("playfield_size"):
                continue
            step = self.table_step if name == "y_embedder.embedding_table.weight" else self.step_count
            state[i] = {"step": torch.tensor(float(step)),
                        "exp_avg": self.arena.view(self.exp_avg, name).clone(),
                        "exp_avg_sq": self.arena.view(self.exp_avg_sq, name).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "foreach": None, "maximize": False, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(self.arena.names)))}
        return {"state": state, "param_groups": [group]}

    def load_opt_state_dict(self, sd):
        self.lr = sd["param_groups"][0]["lr"]
        # a parameter without an entry has no optimizer state: torch creates it lazily at step 0 with zero moments (this is
        # what --relearn-embeds relies on for the class table, train.py:212-215)
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count = self.table_step = 0
        for i, st in sd["state"].items():
            name = self.arena.names[int(i)]
            self.arena.view(self.exp_avg, name).copy_(st["exp_avg"])
            self.arena.view(self.exp_avg_sq, name).copy_(st["exp_avg_sq"])
            if name == "y_embedder.embedding_table.weight":
                self.table_step = int(float(st["step"]))
            else:
                self.step_count = int(float(st["step"]))

    def checkpoint(self, args=None):
        self.finish_exchange()
        return self._checkpoint(args)

    def _checkpoint(self, args=None):
        """The reference's checkpoint dict (train.py:287-293).  With the sharded optimizer the state of the other ranks' shards
        must have been gathered by `sync_sharded_state()` -- a collective that EVERY rank has to enter -- before one rank alone
        calls this; a stale state here is an error, not something to fix up with a collective only this rank would join."""
        if self.shard_optimizer and self._ema_stale:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
                raise RuntimeError("NativeTrainer.checkpoint(): sharded optimizer state is stale -- call sync_sharded_state() on "
                                   "EVERY rank first (it all-gathers moments and EMA), then checkpoint() on the saving rank")
            self.sync_sharded_state()  # one rank: a local no-op gather
        scaler = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                  "_growth_tracker": 0}  # bf16 needs no loss scaling; key kept for layout compatibility
        return {"model": {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                "ema": {k: v.detach().cpu().clone() for k, v in self.ema.state_dict().items()},
                "opt": self.opt_state_dict(), "scaler": scaler, "args": args}

    def load_checkpoint(self, ckpt, lr=None, relearn_embeds=False):
        """train.py:203-221 incl. --relearn-embeds (drops the class table and optimizer state 7)."""
        ckpt = {k: (dict(v) if isinstance(v, dict) else v) for k, v in ckpt.items()}
        if lr is not None:
            ckpt["opt"]["param_groups"][0]["lr"] = lr
        if relearn_embeds:
            del ckpt["model"]["y_embedder.embedding_table.weight"]
            del ckpt["ema"]["y_embedder.embedding_table.weight"]
            ckpt["opt"] = dict(ckpt["opt"], state={k: v for k, v in ckpt["opt"]["state"].items() if int(k) != 7})
        self.model.load_state_dict(ckpt["model"], strict=not relearn_embeds)
        self.ema.load_state_dict(ckpt["ema"], strict=not relearn_embeds)
        self.load_opt_state_dict(ckpt["opt"])
        self.model._uploaded = {}
        self.ema._uploaded = {}
