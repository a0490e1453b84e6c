"""The library's own RCCL communicator (C ABI: osud_comm_* / osud_allreduce_grads ... in include/osud.h) for the
data-parallel gradient exchange of train.py (reference: train.py:106 init_process_group, :152 DDP's parameter broadcast,
:257 the gradient all-reduce).

`NativeComm.from_torch_distributed()` bootstraps it inside a torchrun job: rank 0 makes the 128-byte unique id, torch's
process group (any backend) carries it to the other ranks once, and from then on the heavy traffic -- the per-slice
all-reduce / reduce-scatter / all-gather of the flat gradient and parameter arenas -- goes through libosud's RCCL calls on a
side stream, overlapped with the backward phases.  `NativeTrainer(native_comm=True)` (train.py --native-comm) uses it.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

WIRE = {torch.float32: 0, torch.bfloat16: 1}


class _Pending:
    """What an async collective returns: wait() orders the CURRENT stream behind it (no host sync), like torch's Work.wait()."""

    def __init__(self, event, keep=None):
        self.event, self.keep = event, keep

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


class NativeComm:
    def __init__(self, rank: int, world: int, uid: bytes, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.rank, self.world = int(rank), int(world)
        assert len(uid) == 128, "the RCCL unique id is 128 bytes"
        h = C.c_void_p()
        buf = C.create_string_buffer(uid, 128)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().osud_comm_init(self.rank, self.world, buf, C.byref(h)))
        self._h = h
        self.stream = torch.cuda.Stream(device=self.device)  # collectives run here, beside the compute stream

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().osud_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def rccl_version() -> int:
        return int(_lib.lib().osud_comm_rccl_version())

    @classmethod
    def from_torch_distributed(cls, group=None, device=None):
        """Inside an initialised torch.distributed job (torchrun): rank 0 creates the id, the process group carries it."""
        import torch.distributed as dist

        assert dist.is_available() and dist.is_initialized(), "from_torch_distributed needs an initialised process group"
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0], device)

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.lib().osud_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- collectives.  async_op=True: enqueued on the communicator's side stream behind everything the current stream has
    # queued so far; the returned handle's wait() orders the current stream behind the collective.
    def _run(self, fn, async_op, keep=None):
        cur = torch.cuda.current_stream(self.device)
        st = self.stream if async_op else cur
        if async_op:
            st.wait_stream(cur)
        with torch.cuda.device(self.device):
            _lib.check(fn(C.c_void_p(st.cuda_stream)))
        if not async_op:
            return None
        ev = torch.cuda.Event()
        ev.record(st)
        return _Pending(ev, keep)

    def all_reduce_(self, t: torch.Tensor, async_op=False):
        """SUM in place (fp32, or bf16 for gradients rounded for the wire)."""
        assert t.is_contiguous() and t.dtype in WIRE
        if async_op:
            t.record_stream(self.stream)
        return self._run(lambda s: _lib.lib().osud_allreduce_grads(self._h, _lib.ptr(t), t.numel(), WIRE[t.dtype], s), async_op, t)

    def broadcast_(self, t: torch.Tensor, root=0, async_op=False):
        assert t.is_contiguous() and t.dtype == torch.float32
        return self._run(lambda s: _lib.lib().osud_broadcast_params(self._h, _lib.ptr(t), t.numel(), int(root), s), async_op, t)

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, async_op=False):
        assert out.is_contiguous() and inp.is_contiguous() and inp.dtype == out.dtype and inp.numel() == out.numel() * self.world
        return self._run(lambda s: _lib.lib().osud_reduce_scatter_grads(self._h, _lib.ptr(inp), _lib.ptr(out), out.numel(), WIRE[out.dtype], s),
                         async_op, (out, inp))

    def all_gather(self, full: torch.Tensor, shard: torch.Tensor, async_op=False):
        assert full.is_contiguous() and shard.is_contiguous() and full.dtype == shard.dtype == torch.float32
        assert full.numel() == shard.numel() * self.world
        return self._run(lambda s: _lib.lib().osud_allgather_params(self._h, _lib.ptr(shard), _lib.ptr(full), shard.numel(), s),
                         async_op, (full, shard))
