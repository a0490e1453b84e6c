"""CPU, world_size 2 over gloo: the N > 1 pieces of the training path that do not need a GPU — the
per-rank data sharding rule (train.py:165-170, data_loading.py:366-376) and the flat-arena gradient
averaging that replaces DDP's bucketed all-reduce (train.py:152,257)."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from osu_diffusion_amd.training import allreduce_mean_, shard_range, worker_range


def test_shard_rules():
    assert [shard_range(0, 13402, r, 8) for r in (0, 1, 7)] == [(0, 1676), (1676, 3352), (11732, 13402)]
    got = [shard_range(5, 105, r, 3) for r in range(3)]
    assert got[0][0] == 5 and got[-1][1] == 105 and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    assert [worker_range(0, 10, w, 4) for w in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # ranks are seeded differently (train.py:113-114)
        flat = torch.randn(1000)
        local = flat.clone()
        scale = allreduce_mean_(flat)
        gathered = [torch.empty(1000) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = torch.stack(gathered).sum(0)
        ok = bool(torch.allclose(flat, want)) and abs(scale - 1.0 / world) < 1e-12
        # init broadcast (DDP ctor): rank 0's parameters win
        params = torch.full((8,), float(rank))
        dist.broadcast(params, 0)
        ok = ok and bool((params == 0).all())
        out[rank] = ok
    finally:
        dist.destroy_process_group()


def test_allreduce_mean_two_ranks_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_allreduce_mean_without_process_group_is_identity():
    t = torch.ones(4)
    assert allreduce_mean_(t) == 1.0 and torch.equal(t, torch.ones(4))


def test_overlap_slices_cover_the_arena_exactly_once():
    from osu_diffusion_amd.models import DiT
    from osu_diffusion_amd.training import ParamArena, overlap_slices

    m = DiT(depth=3, hidden_size=128, num_heads=2, context_size=144, num_classes=4)
    arena = ParamArena(m)
    blocks, tail = overlap_slices(arena, 3)
    kinds = [t[0] for t in tail]
    assert [b[1] for b in blocks] == [0, 1, 2] and kinds[:2] == ["tail", "table"] and kinds[-1] == "final"
    seen = torch.zeros(arena.total, dtype=torch.int32)
    for _, _, lo, hi in blocks + tail:
        seen[lo:hi] += 1
    assert bool((seen == 1).all())
    names = dict(zip(arena.names, zip(arena.offsets[:-1], arena.sizes)))
    for key in ("blocks.1.mlp.fc1.weight", "blocks.1.adaLN_modulation.1.weight", "blocks.1.adaLN_modulation.1.bias"):
        lo, n = names[key]  # a block's adaLN pair travels with the block (differentiated inside the block's phase)
        assert blocks[1][2] <= lo and lo + n <= blocks[1][3]
    lo, n = names["final_layer.adaLN_modulation.1.weight"]
    assert tail[-1][2] <= lo and lo + n <= tail[-1][3]
    lo, n = names["y_embedder.embedding_table.weight"]
    assert tail[1] == ("table", -1, int(lo), int(lo + n))  # exchanged as rows, not densely


def _table_worker(rank, world, port, out):
    import torch.distributed as dist
    from osu_diffusion_amd.training import exchange_table_rows

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7 + rank)
    labels = torch.randint(0, 12, (16,), generator=g)  # duplicates inside a rank and overlaps across ranks
    dense = torch.zeros(12, 8)
    dense.index_add_(0, labels, torch.randn(16, 8, generator=g))  # what the backward leaves: only label rows non-zero
    ref = dense.clone()
    dist.all_reduce(ref)  # the reference's dense DDP all-reduce
    got = exchange_table_rows(dense.clone(), labels)
    out[rank] = (ref, got)
    dist.destroy_process_group()


def test_row_exchange_of_the_class_table_equals_dense_allreduce():
    import torch.multiprocessing as mp

    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_table_worker, args=(2, port, out), nprocs=2, join=True)
        (ref0, got0), (ref1, got1) = out[0], out[1]
    assert torch.equal(got0, got1)  # replicas bit-identical
    assert float((got0 - ref0).abs().max()) <= 1e-6 and float((ref0 - ref1).abs().max()) == 0.0


def test_sampling_row_shards():
    from osu_diffusion_amd.sharding import shard_rows

    assert [shard_rows(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    assert [shard_rows(3, r, 2) for r in range(2)] == [(0, 2), (2, 3)]
    assert [shard_rows(1, r, 4) for r in range(4)] == [(0, 1), (1, 1), (1, 1), (1, 1)]  # more ranks than variants: empty shards
    for n, w in ((5, 3), (8, 8), (9, 4)):
        got = [shard_rows(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n and all(a[1] == b[0] for a, b in zip(got, got[1:]))


def _gather_worker(rank, world, port, n_total, out):
    import torch.distributed as dist
    from osu_diffusion_amd.sharding import gather_rows, shard_rows

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(n_total * 6, dtype=torch.float32).view(n_total, 2, 3)
    lo, hi = shard_rows(n_total, rank, world)
    got = gather_rows(full[lo:hi].clone(), n_total, rank, world)
    out[rank] = None if got is None else bool(torch.equal(got, full))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [3, 1, 4])
def test_gather_rows_two_ranks_gloo(n_total):
    """sample.py under torchrun: ragged shards (3 variants on 2 ranks), an empty shard (1 variant on 2 ranks), even shards."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_gather_worker, args=(2, port, n_total, out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: None}


def test_shard_plan_partitions_a_slice():
    from osu_diffusion_amd.training import shard_plan

    for lo, hi, W in ((0, 1000, 8), (17, 17 + 7_077_888, 8), (5, 5 + 30, 8), (100, 103, 2), (0, 4096, 1)):
        per, bulk_hi = shard_plan(lo, hi, W)
        assert per % 4 == 0 and bulk_hi == lo + per * W and bulk_hi <= hi and hi - bulk_hi < 4 * W + W


def _zero1_worker(rank, world, port, out):
    """The exchange of the sharded optimizer on CPU tensors over gloo (reduce-scatter falls back to all-reduce + own shard):
    scatter -> local update of the own shard -> gather must equal all-reduce -> update everywhere."""
    from osu_diffusion_amd.training import _all_gather_into, _reduce_scatter_sum, shard_plan

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(3 + rank)
        lo, hi = 16, 16 + 1003                      # a slice that does not divide by world * 4
        grads = torch.randn(1100, generator=g)
        params = torch.arange(1100, dtype=torch.float32) / 100
        ref_g = grads.clone()
        dist.all_reduce(ref_g)
        want = params.clone()
        want[lo:hi] -= 0.1 * ref_g[lo:hi] / world
        per, bulk_hi = shard_plan(lo, hi, world)
        shard = torch.empty(per)
        h, fin = _reduce_scatter_sum(shard, grads[lo:bulk_hi], None)
        h2 = dist.all_reduce(grads[bulk_hi:hi], async_op=True)
        h.wait(); fin and fin(); h2.wait()
        a, b = lo + rank * per, lo + (rank + 1) * per
        params[a:b] -= 0.1 * shard / world
        params[bulk_hi:hi] -= 0.1 * grads[bulk_hi:hi] / world
        h, fin = _all_gather_into(params[lo:bulk_hi], params[a:b], None)
        h.wait(); fin and fin()
        out[rank] = float((params - want).abs().max())
    finally:
        dist.destroy_process_group()


def test_sharded_exchange_two_ranks_gloo():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_zero1_worker, args=(2, port, out), nprocs=2, join=True)
        assert max(out.values()) <= 1e-6, dict(out)


def _sampler_worker(rank, world, port, out):
    """LossAwareSampler.update_with_local_losses over gloo: ranks contribute batches of different sizes; afterwards every rank
    holds the same history, equal to feeding rank 0's pairs then rank 1's."""
    import numpy as np
    from osu_diffusion_amd.diffusion import create_diffusion
    from osu_diffusion_amd.diffusion.timestep_sampler import create_named_schedule_sampler

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
        s = create_named_schedule_sampler("loss-second-moment", d)
        g = torch.Generator().manual_seed(50 + rank)
        n = 5 + 3 * rank
        ts = torch.randint(0, 20, (n,), generator=g)
        losses = torch.rand(n, generator=g)
        s.update_with_local_losses(ts, losses)
        out[rank] = (s._loss_counts.copy(), np.sort(s._loss_history, axis=1), ts, losses)
    finally:
        dist.destroy_process_group()


def test_loss_aware_sampler_synchronises_over_gloo():
    import numpy as np
    from osu_diffusion_amd.diffusion import create_diffusion
    from osu_diffusion_amd.diffusion.timestep_sampler import create_named_schedule_sampler

    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_sampler_worker, args=(2, port, out), nprocs=2, join=True)
        (c0, h0, t0, l0), (c1, h1, t1, l1) = out[0], out[1]
    assert np.array_equal(c0, c1) and np.array_equal(h0, h1)
    ref = create_named_schedule_sampler("loss-second-moment", create_diffusion("20", noise_schedule="squaredcos_cap_v2"))
    ref.update_with_all_losses(t0.tolist() + t1.tolist(), l0.tolist() + l1.tolist())
    assert np.array_equal(ref._loss_counts, c0) and np.allclose(np.sort(ref._loss_history, axis=1), h0)


# ------------------------------------------------------------------ world = 8 (the node the job targets): shard arithmetic
def test_param_arena_offsets_are_16_byte_aligned():
    """Every tensor of the flat arena starts on a 4-element boundary (the 2-element playfield_size comes first): the sharded
    optimizer's shards, moments and EMA are then aligned like its (aligned) scatter buffer and take the 16-byte path."""
    from osu_diffusion_amd.models import DiT
    from osu_diffusion_amd.training import ParamArena, overlap_slices, shard_plan

    m = DiT(depth=3, hidden_size=128, num_heads=2, context_size=144, num_classes=5)
    want = {n: p.detach().clone() for n, p in m.named_parameters()}
    arena = ParamArena(m)
    assert all(int(o) % 4 == 0 for o in arena.offsets) and arena.total % 4 == 0
    assert arena.total >= sum(arena.sizes) and arena.total - sum(arena.sizes) < 4 * len(arena.sizes)
    for n, p in m.named_parameters():  # re-homed, values and Parameter objects intact
        assert torch.equal(p.detach(), want[n]) and p.data_ptr() == arena.view(arena.flat, n).data_ptr()
    blocks, tail = overlap_slices(arena, 3)
    _, _, f_lo, f_hi = next(s for s in tail if s[0] == "final")
    for W in (2, 8):
        for lo, hi in [(f_lo, f_hi)] + [(b[2], b[3]) for b in blocks]:
            per, bulk_hi = shard_plan(lo, hi, W)
            assert lo % 4 == 0 and per % 4 == 0 and all((lo + r * per) % 4 == 0 for r in range(W))
            assert lo + W * per == bulk_hi <= hi and hi - bulk_hi < 4 * W


def _zero1_world8_worker(rank, world, port, out):
    """_backward_sharded's exchange arithmetic at world = 8 on a real arena layout (CPU tensors, gloo): every slice of the phased
    backward is reduce-scattered (bulk) + all-reduced (remainder), the own shard and the replicated parts are updated, the shards
    are gathered back; the class table travels as rows.  Must equal: dense all-reduce of everything, update everywhere."""
    from osu_diffusion_amd.models import DiT
    from osu_diffusion_amd.training import (ParamArena, _all_gather_into, _complement, _reduce_scatter_sum, exchange_table_rows,
                                            overlap_slices, shard_plan)

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(5)
        m = DiT(depth=2, hidden_size=128, num_heads=2, context_size=144, num_classes=37)  # 38 table rows
        arena = ParamArena(m)
        g = torch.Generator().manual_seed(11 + rank)
        grads = torch.randn(arena.total, generator=g)
        t_lo, t_hi = [(lo, hi) for k, _, lo, hi in overlap_slices(arena, 2)[1] if k == "table"][0]
        rows = (t_hi - t_lo) // 128
        labels = torch.randint(0, rows, (6,), generator=g)  # duplicates inside a rank, overlaps across ranks, untouched rows
        tg = torch.zeros(rows, 128)
        tg.index_add_(0, labels, torch.randn(6, 128, generator=g))
        grads[t_lo:t_hi] = tg.reshape(-1)
        params = arena.flat.clone()
        ref = grads.clone()
        dist.all_reduce(ref)
        want = params - 0.1 * ref / world
        blocks, tail = overlap_slices(arena, 2)
        _, _, f_lo, f_hi = next(s for s in tail if s[0] == "final")
        order = [(f_lo, f_hi)] + [(b[2], b[3]) for b in reversed(blocks)]
        plans = [(lo, hi) + shard_plan(lo, hi, world) for lo, hi in order]
        buf = torch.empty(sum(p[2] for p in plans))
        pending, off = [], 0
        for lo, hi, per, bulk_hi in plans:
            if per > 0:
                pending.append(_reduce_scatter_sum(buf[off:off + per], grads[lo:bulk_hi], None))
            if bulk_hi < hi:
                pending.append((dist.all_reduce(grads[bulk_hi:hi], async_op=True), None))
            off += per
        for kind, _, lo, hi in tail:
            if kind == "tail":
                pending.append((dist.all_reduce(grads[lo:hi], async_op=True), None))
        for h, fin in pending:
            h.wait()
            if fin is not None:
                fin()
        exchange_table_rows(grads[t_lo:t_hi].view(rows, 128), labels)
        off, own, gathers = 0, [], []
        for lo, hi, per, bulk_hi in plans:
            if per > 0:
                a, b = lo + rank * per, lo + (rank + 1) * per
                params[a:b] -= 0.1 * buf[off:off + per] / world
                own.append((lo, bulk_hi))
                gathers.append(_all_gather_into(params[lo:bulk_hi], params[a:b], None))
            off += per
        for lo, hi in _complement(own, arena.total):
            params[lo:hi] -= 0.1 * grads[lo:hi] / world
        for h, fin in gathers:
            h.wait()
            if fin is not None:
                fin()
        out[rank] = (float((params - want).abs().max()), params)
    finally:
        dist.destroy_process_group()


def test_sharded_exchange_eight_ranks_gloo():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_zero1_world8_worker, args=(8, port, out), nprocs=8, join=True)
        res = dict(out)
    assert sorted(res) == list(range(8))
    assert max(v[0] for v in res.values()) <= 2e-6, {k: v[0] for k, v in res.items()}
    assert all(torch.equal(res[0][1], res[r][1]) for r in range(1, 8))  # replicas bit-identical after the gather


class _StubTrainer:
    """What bench.multi_gpu_report needs of a trainer, with the compute replaced by a sleep and the exchange by the REAL slice schedule on
    CPU tensors: the arena's slices of the phased backward (training.overlap_slices) are all-reduced one by one (schedule `allreduce`), or
    reduce-scattered and the masters gathered back (`zero1`), or skipped (`stub_exchange`: the compute floor)."""

    def __init__(self, model, diffusion, lr=1e-4, shard_optimizer=False, overlap_gather=True, native_comm=False, stub_exchange=False):
        from osu_diffusion_amd.training import overlap_slices

        self.arena, self.zero1, self.stub, self.world = model._arena, shard_optimizer, stub_exchange, dist.get_world_size()
        blocks, tail = overlap_slices(self.arena, model.depth)
        self.slices = [(lo, hi) for k, _, lo, hi in tail if k in ("final", "tail")] + [(b[2], b[3]) for b in reversed(blocks)]
        self.grads = torch.zeros(self.arena.total)
        self.steps = 0

    def step(self, x, o, c, y):
        import time

        from osu_diffusion_amd.training import _all_gather_into, _reduce_scatter_sum, shard_plan

        time.sleep(0.002)  # "compute"
        self.grads.fill_(1.0)
        if not self.stub:
            pending = []
            for lo, hi in self.slices:
                if self.zero1:
                    per, bulk_hi = shard_plan(lo, hi, self.world)
                    if per > 0:
                        buf = torch.empty(per)
                        pending.append(_reduce_scatter_sum(buf, self.grads[lo:bulk_hi], None))
                        pending.append(_all_gather_into(self.arena.flat[lo:bulk_hi], self.arena.flat[lo + dist.get_rank() * per:lo + (dist.get_rank() + 1) * per], None))
                    if bulk_hi < hi:
                        pending.append((dist.all_reduce(self.grads[bulk_hi:hi], async_op=True), None))
                else:
                    pending.append((dist.all_reduce(self.grads[lo:hi], async_op=True), None))
            for h, fin in pending:
                h.wait()
                if fin is not None:
                    fin()
            if not self.zero1:
                lo, hi = self.slices[0]
                assert float(self.grads[lo]) == float(self.world)  # every rank's ones arrived
        self.steps += 1

    def finish_exchange(self):
        pass


def _bench_report_worker(rank, world, port, out):
    import argparse
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from osu_diffusion_amd.models import DiT
        from osu_diffusion_amd.training import ParamArena

        torch.manual_seed(0)
        m = DiT(depth=2, hidden_size=128, num_heads=2, context_size=144, num_classes=37)
        m._arena = ParamArena(m)
        args = argparse.Namespace(batch=4, seq_len=64, precision="bf16", steps=4, simulate_hang=False)
        batches = [((None, None, None), None)] * 4
        rep = bench.multi_gpu_report(args, world, rank, torch.device("cpu"), m, None, batches, 12.5, "allreduce", trainer_factory=_StubTrainer)
        # ... and the start-up choice of the exchange schedule, as bench_train records it (multi_gpu.schedule_selected)
        from osu_diffusion_amd.training import select_exchange_schedule

        rep["schedule_selected"] = select_exchange_schedule(lambda shard_optimizer: _StubTrainer(m, None, shard_optimizer=shard_optimizer), batches, steps=3, warmup=1)
        out[rank] = rep
    finally:
        dist.destroy_process_group()


def test_bench_multi_gpu_report_schema_at_world_8():
    """bench.py's N > 1 report (`multi_gpu`) at the world size the driver's scaling run uses, over gloo on CPU tensors with stubbed compute:
    every rank is seen, every schedule leg runs and is timed, the exposed-communication figure and the wire bytes exist and are
    consistent with the arena, and the unoverlapped wire times the first 8-GPU line is to be judged against are there."""
    world = 8
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_bench_report_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert sorted(res) == list(range(world))
    rep = res[0]
    assert rep["rccl"]["ranks_seen"] == world and rep["rccl"]["world_size"] == world and rep["rccl"]["backend"] == "gloo"
    sched = rep["schedules"]
    assert set(sched) == {"allreduce", "zero1", "zero1_no_overlap", "native_comm", "no_exchange"}
    for name in ("allreduce", "zero1", "zero1_no_overlap", "no_exchange"):
        assert sched[name]["ms_per_step"] > 0 and sched[name]["tokens_per_s"] > 0, (name, sched[name])
    assert "skipped" in sched["native_comm"]  # (needs one GPU per rank)
    assert rep["fastest_schedule"] in ("allreduce", "zero1", "zero1_no_overlap")
    assert abs(rep["exposed_comm_ms_per_step"] - (sched["allreduce"]["ms_per_step"] - sched["no_exchange"]["ms_per_step"])) < 2e-3
    assert rep["timed_schedule"] == {"name": "allreduce", "ms_per_step": 12.5}
    wb = rep["wire_bytes_per_step"]
    from osu_diffusion_amd.models import DiT
    from osu_diffusion_amd.training import ParamArena, overlap_slices

    torch.manual_seed(0)
    m = DiT(depth=2, hidden_size=128, num_heads=2, context_size=144, num_classes=37)
    arena = ParamArena(m)
    blocks, tail = overlap_slices(arena, 2)
    table = next(hi - lo for k, _, lo, hi in tail if k == "table")
    assert wb["dense_slices_payload"] == 4 * (arena.total - table) and wb["class_table_dense_would_be"] == 4 * table
    assert wb["class_table_rows_allgather"] == world * 4 * (128 * 4 + 8)
    assert wb["ring_bytes_sent_per_gpu"] == int(2 * 7 / 8 * wb["dense_slices_payload"] + 7 / 8 * wb["class_table_rows_allgather"])
    pc = rep["predicted_comm_ms_per_step"]
    assert pc["link_GBps"] == 153.0 and pc["ring_allreduce_unoverlapped"] > pc["mesh_reduce_scatter_allgather_unoverlapped"] > 0
    # DiT-B's figures, as DESIGN.md section 6 quotes them: 170 370 054 parameters less the 52 671 x 768 class table = 519.7 MB of dense fp32
    # slices -> 5.98 ms over one ring link, 0.85 ms over the mesh (the table travels as 8 x 256 rows: 6.3 MB)
    import bench

    p = bench.predicted_comm_ms(4 * (170_370_054 - 52_671 * 768), 8 * 256 * (768 * 4 + 8), 8)
    assert abs(p["ring_allreduce_unoverlapped"] - 5.98) < 0.03 and abs(p["mesh_reduce_scatter_allgather_unoverlapped"] - 0.854) < 0.01
    sel = rep["schedule_selected"]
    assert sel["name"] in ("allreduce", "zero1") and sel["allreduce_ms"] > 0 and sel["zero1_ms"] > 0 and sel["steps"] == 3 and sel["world_size"] == world
    for r in range(1, world):  # every rank computed the same report skeleton -- and KEPT THE SAME SCHEDULE (the times are maxima over ranks)
        assert res[r]["rccl"]["ranks_seen"] == world and set(res[r]["schedules"]) == set(sched)
        assert res[r]["schedule_selected"] == sel
    line = json.loads(bench.compact_line({"metric": "m", "value": 1.0, "unit": "tokens/s", "n_gpus": world, "steps": 4, "warmup": 1, "ms_per_step": 12.5,
                                          "config": {"workload": "w"}, "multi_gpu": rep}))
    assert line["multi_gpu"]["schedule_selected"] == sel and len(json.dumps(line)) < bench.LINE_LIMIT


class _SlowExchangeTrainer:
    """A trainer whose step costs `base` seconds plus `extra[schedule]`: the selection logic must keep the faster schedule, the same on every rank
    -- also when the ranks disagree about which one was faster locally (rank 1's all-reduce steps are made slow: the MAX over ranks decides)."""

    def __init__(self, shard_optimizer, extra):
        self.delay = 0.004 + extra["zero1" if shard_optimizer else "allreduce"]
        self.t = torch.zeros(4)

    def step(self, x, o, c, y):
        import time

        time.sleep(self.delay)
        dist.all_reduce(self.t)

    def finish_exchange(self):
        pass


def _selection_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from osu_diffusion_amd.training import select_exchange_schedule

        batches = [((None, None, None), None)] * 2
        res = {}
        # (a) the sharded form is slower everywhere -> allreduce; (b) the all-reduce is slow on rank 1 only -> zero1, on BOTH ranks;
        # (c) a near-tie (1 %) keeps the all-reduce
        for case, extra in (("a", {"allreduce": 0.0, "zero1": 0.02}), ("b", {"allreduce": 0.03 if rank == 1 else 0.0, "zero1": 0.01}),
                            ("c", {"allreduce": 0.0500, "zero1": 0.0495})):
            res[case] = select_exchange_schedule(lambda shard_optimizer, e=extra: _SlowExchangeTrainer(shard_optimizer, e), batches, steps=3, warmup=1)

        # (d) the sharded form cannot be built on this node (raised on every rank alike): the all-reduce is kept, the error is on record
        def only_allreduce(shard_optimizer):
            if shard_optimizer:
                raise RuntimeError("no reduce-scatter here")
            return _SlowExchangeTrainer(False, {"allreduce": 0.0, "zero1": 0.0})

        res["d"] = select_exchange_schedule(only_allreduce, batches, steps=2, warmup=1)
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_exchange_schedule_is_selected_by_measurement_and_agreed_by_every_rank():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_selection_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert res[0] == res[1]  # same times (maxima over ranks), same choice
    assert res[0]["a"]["name"] == "allreduce" and res[0]["a"]["zero1_ms"] > res[0]["a"]["allreduce_ms"]
    assert res[0]["b"]["name"] == "zero1" and res[0]["b"]["allreduce_ms"] > 30.0  # rank 1's slow steps decide for both
    assert res[0]["c"]["name"] == "allreduce"
    assert res[0]["d"]["name"] == "allreduce" and "no reduce-scatter here" in res[0]["d"]["zero1_error"] and "zero1_ms" not in res[0]["d"]
    from osu_diffusion_amd.training import select_exchange_schedule

    assert select_exchange_schedule(None, [])["name"] == "allreduce"  # no process group: nothing is run
