"""ctypes binding of libosud.so (C ABI declared in include/osud.h).

The shared library is the product; this module only loads it and converts status codes
into Python exceptions.  There is deliberately NO fallback: if the library is missing or a
symbol cannot be resolved, importing the native path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OSUD_LIB", os.path.join(_HERE, "libosud.so"))

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = 0, 1, 2, 3, 4
PREC_BF16, PREC_F32 = 0, 1
SAMPLER_P, SAMPLER_DDIM = 0, 1
PREC_FP8 = 2  # inference only: bf16 tier with e4m3 operands in the four big per-block GEMMs
PREC_BF16X3 = 3  # inference only: split-bf16 operands (hi + lo planes), three bf16 MFMAs per product -- the tier that meets the
#                  reference's 1e-3 tolerance at MFMA speed (the reference's own sampling matmuls are TF32: sample.py:25-26)
PREC_F16F8 = 4  # inference only: the split-bf16 tier with the four big GEMMs of a block on fp16 + e4m3-residual operands (15
#                 significand bits per operand; fp16 hi product + ONE block-scaled e4m3 MFMA for both cross terms: 2/3 of the passes)
PREC_F16 = 5  # inference only: the bf16 tier's kernels on IEEE half operands -- 11 significand bits, the precision of the TF32 matmuls of
#               the reference's own sampling path (sample.py:25-26) -- at the bf16 tier's speed
PREC_F16W8 = 6  # inference only: fp16f8 with the ACTIVATION operand of the four big GEMMs in plain fp16 (the weight keeps its e4m3
#                 residual: 15 bits where the error would repeat in every product): 3/4 of fp16f8's matrix-pipe passes
PREC_F16M8 = 7  # inference only: per-GEMM mix of the fp16f8 and fp16w8 operand forms (option "f16m8_forms", read when the handle is made)
PRECISIONS = {"bf16": PREC_BF16, "fp32": PREC_F32, "f32": PREC_F32, "fp8": PREC_FP8, "bf16x3": PREC_BF16X3, "fp16f8": PREC_F16F8,
              "fp16": PREC_F16, "fp16w8": PREC_F16W8, "fp16m8": PREC_F16M8}

# gemm epilogue codes (csrc/gemm.h)
EPI_BIAS_F32, EPI_BIAS_TE, EPI_BIAS_SILU_TE, EPI_ROWBIAS_TE, EPI_BIAS_GELU_TE, EPI_GATE_RES = range(6)
EPI_NONE_F32, EPI_NONE_TE, EPI_ACCUM_F32, EPI_GELUGRAD_TE = 6, 7, 8, 9


class DitCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden", "depth", "heads", "context", "in_channels", "table_rows",
                                          "learn_sigma", "precision")]


class NativeError(RuntimeError):
    """A libosud call failed (HIP error, missing parameter, unsupported configuration)."""


_lib = None

_vp, _i, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t
_SIGNATURES = {
    "osud_last_error": (C.c_char_p, []),
    "osud_version": (_i, []),
    "osud_build_arch": (C.c_char_p, []),
    "osud_dit_create": (_i, [C.POINTER(DitCfg), C.POINTER(_vp)]),
    "osud_dit_destroy": (None, [_vp]),
    "osud_dit_set_param": (_i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i, _vp]),
    "osud_dit_missing_params": (_i, [_vp]),
    "osud_dit_reserve": (_i, [_vp, _i, _i, _i]),
    "osud_dit_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "osud_dit_calibrate_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    "osud_sched_create": (_i, [C.POINTER(C.c_double), _i, C.POINTER(C.c_int64), _i, C.POINTER(_vp)]),
    "osud_sched_destroy": (None, [_vp]),
    "osud_sched_num_timesteps": (_i, [_vp]),
    "osud_sched_table": (_i, [_vp, C.c_char_p, C.POINTER(C.c_double), _i]),
    "osud_sched_timestep_map": (_i, [_vp, C.POINTER(C.c_int64), _i]),
    "osud_sampler_step": (_i, [_vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp]),
    "osud_sample_loop": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i, _vp, _u64, _vp]),
    "osud_sampler_step_inpaint": (_i, [_vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "osud_sample_loop_inpaint": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i, _vp, _u64, _vp, _vp]),
    "osud_sample_repeat": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i, _vp, _u64, _vp, _vp]),
    "osud_dit_bind_grad": (_i, [_vp, C.c_char_p, _vp]),
    "osud_dit_refresh": (_i, [_vp, _vp]),
    "osud_dit_refresh_phases": (_i, [_vp, _i, _i, _vp]),
    "osud_dit_forward_gate": (_i, [_vp, _i, _vp]),
    "osud_dit_forward_train": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "osud_dit_backward": (_i, [_vp, _vp, _vp]),
    "osud_dit_backward_phases": (_i, [_vp, _vp, _i, _i, _vp]),
    "osud_q_sample": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "osud_train_loss": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "osud_adamw_ema_step": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _f, _sz, _sz, _f, _vp]),
    "osud_comm_unique_id": (_i, [_vp]),
    "osud_comm_init": (_i, [_i, _i, _vp, C.POINTER(_vp)]),
    "osud_comm_destroy": (None, [_vp]),
    "osud_comm_rank": (_i, [_vp]),
    "osud_comm_world": (_i, [_vp]),
    "osud_comm_rccl_version": (_i, []),
    "osud_allreduce_grads": (_i, [_vp, _vp, _sz, _i, _vp]),
    "osud_broadcast_params": (_i, [_vp, _vp, _sz, _i, _vp]),
    "osud_reduce_scatter_grads": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "osud_allgather_params": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "osud_table_rows_pack": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "osud_table_rows_apply": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "osud_op_gemm": (_i, [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "osud_op_gemm_ex": (_i, [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "osud_op_convert": (_i, [_i, _vp, _vp, _sz, _vp]),
    "osud_op_pack_h8": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "osud_op_pack_w8": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "osud_set_gemm_dynamic_tiles": (_i, [_i]),
    "osud_set_option": (_i, [C.c_char_p, _i]),
    "osud_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "osud_op_attention": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "osud_op_wgrad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "osud_op_wgrad8": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "osud_op_attention_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
}


def lib():
    """Load libosud.so once; raise (never fall back) if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (libamdhip64.so.7); load it FIRST so libosud.so binds to the
    # same runtime instance as the tensors and streams it is handed (two runtimes in one process
    # see no common device state).
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise NativeError(
            f"libosud.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C osu_diffusion_amd/csrc`). There is no CPU/PyTorch fallback for the DiT path.")
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def last_error() -> str:
    return (lib().osud_last_error() or b"").decode()


def check(rc: int):
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_ARG:  # the reference raises AssertionError / IndexError on bad shapes
        raise AssertionError(msg)
    raise NativeError(f"libosud error {rc}: {msg}")


def set_option(name: str, value: int) -> None:
    """osud_set_option (include/osud.h): the library's one table of run-time switches; value -1 = the default."""
    check(lib().osud_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib().osud_get_option(name.encode(), C.byref(v)))
    return int(v.value)


class option:
    """`with _lib.option("sample_graph", 0): ...` -- an option set for a block and put back afterwards."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False


def ptr(t):
    """Device (or host) pointer of a torch tensor / None."""
    return None if t is None else C.c_void_p(t.data_ptr())


class InPaintStruct(C.Structure):
    """`osud_inpaint` of include/osud.h: device pointers to the (N,2,T) keep mask (uint8) and known values (f32)."""
    _fields_ = [("keep", C.c_void_p), ("known", C.c_void_p)]


def stream_ptr(device=None):
    import torch

    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
