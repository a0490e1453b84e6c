"""`DiT_models` registry and the DiT module, backed by the native gfx950 library.

Mirrors the reference's model surface (/root/reference/models.py) so callers switch by
changing one import:

* ``DiT_models[name](num_classes=, context_size=, class_dropout_prob=, learn_sigma=True, ...)``
  (models.py:410-431, ctor :243-273) returns an ``nn.Module`` whose parameter tree —
  names, shapes, dtypes, ``parameters()`` order and seeded initial values — equals the
  reference's, so ``load_state_dict(find_model(ckpt))`` works strict on existing ``.pt`` files.
* ``forward(x, t, o, c, y, attn_mask=None)`` (:306-325) and
  ``forward_with_cfg(x, t, o, c, y, cfg_scale, attn_mask=None)`` (:327-343).

The torch modules below only *hold* parameters (torch is plumbing: device memory, streams,
state dicts).  All arithmetic runs in libosud.so; there is no eager fallback — a module that
is not on a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn as nn

from . import _lib

DEFAULT_PRECISION = os.environ.get("OSUD_PRECISION", "bf16")


class _FirstLayer(nn.Module):
    """Parameter holder for the token embedder (reference FirstLayer, models.py:199-225)."""

    def __init__(self, hidden_size, context_size, in_channels, frequency_embedding_size=128):
        super().__init__()
        fan_in = in_channels * frequency_embedding_size + frequency_embedding_size + context_size
        self.mlp = nn.Sequential(nn.Linear(fan_in, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size
        self.playfield_size = nn.Parameter(torch.tensor((512, 384), dtype=torch.float32), requires_grad=False)


class _TimestepEmbedder(nn.Module):
    """models.py:21-33."""

    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size


class _LabelEmbedder(nn.Module):
    """models.py:41-54; the extra last row is the null class used for CFG."""

    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes + int(dropout_prob > 0), hidden_size)
        self.num_classes = num_classes
        self.dropout_prob = dropout_prob

    def token_drop(self, labels, force_drop_ids=None):
        """models.py:56-67 — the only host-side randomness of the model (train-time label dropout)."""
        if force_drop_ids is None:
            drop_ids = torch.rand(labels.shape[0], device=labels.device) < self.dropout_prob
        else:
            drop_ids = force_drop_ids == 1
        return torch.where(drop_ids, self.num_classes, labels)


class _Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features, bias=True)
        self.fc2 = nn.Linear(hidden_features, in_features, bias=True)


class _Block(nn.Module):
    """Parameter holder for one adaLN-Zero block (models.py:122-149)."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.attn = nn.MultiheadAttention(hidden_size, num_heads=num_heads, batch_first=True)
        self.mlp = _Mlp(hidden_size, int(hidden_size * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))


class _FinalLayer(nn.Module):
    """models.py:178-190."""

    def __init__(self, hidden_size, out_channels):
        super().__init__()
        self.linear = nn.Linear(hidden_size, out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))


class DiT(nn.Module):
    """Diffusion transformer over beatmap token windows; native forward on MI355X."""

    def __init__(self, in_channels=2, context_size=142, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4.0,
                 class_dropout_prob=0.1, num_classes=1000, learn_sigma=True, precision=None):
        super().__init__()
        if mlp_ratio != 4.0:
            raise NotImplementedError("the native path is built for mlp_ratio=4.0 (all reference configs)")
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.context_size = context_size
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.num_heads = num_heads
        self.hidden_size = hidden_size
        self.depth = depth
        self.precision = precision or DEFAULT_PRECISION
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {self.precision!r}")

        self.xoc_embedder = _FirstLayer(hidden_size, context_size, in_channels)
        self.t_embedder = _TimestepEmbedder(hidden_size)
        self.y_embedder = _LabelEmbedder(num_classes, hidden_size, class_dropout_prob)
        self.blocks = nn.ModuleList([_Block(hidden_size, num_heads, mlp_ratio) for _ in range(depth)])
        self.final_layer = _FinalLayer(hidden_size, self.out_channels)
        self.initialize_weights()

        self._handle = None       # osud_dit*
        self._handle_key = None   # (device index, precision)
        self._uploaded = {}       # param name -> (data_ptr, version)
        self._grad_ctx = None

    # ------------------------------------------------------------------ init (models.py:275-304)
    def initialize_weights(self):
        def _basic(mod):  # every Linear (incl. MHA's out_proj): Xavier-uniform weight, zero bias
            if isinstance(mod, nn.Linear):
                nn.init.xavier_uniform_(mod.weight)
                if mod.bias is not None:
                    nn.init.constant_(mod.bias, 0)
        self.apply(_basic)
        nn.init.normal_(self.xoc_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.y_embedder.embedding_table.weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        zero = [blk.adaLN_modulation[-1] for blk in self.blocks]
        zero += [self.final_layer.adaLN_modulation[-1], self.final_layer.linear]
        for lin in zero:  # adaLN-Zero: a fresh model is the identity and outputs exactly 0
            nn.init.constant_(lin.weight, 0)
            nn.init.constant_(lin.bias, 0)

    # ------------------------------------------------------------------ native handle
    def _device(self):
        return self.xoc_embedder.playfield_size.device

    def native_handle(self):
        """Create (once per device/precision) the libosud handle and upload changed parameters."""
        dev = self._device()
        if dev.type != "cuda":
            raise _lib.NativeError(
                "osu_diffusion_amd.DiT runs only on an AMD GPU through libosud.so; move the module to the GPU "
                "(`model.to('cuda')`). There is no CPU fallback.")
        L = _lib.lib()
        key = (dev.index or 0, self.precision)
        if self._handle is None or self._handle_key != key:
            self._free_handle()
            cfg = _lib.DitCfg(self.hidden_size, self.depth, self.num_heads, self.context_size, self.in_channels,
                              self.y_embedder.embedding_table.weight.shape[0], int(self.learn_sigma),
                              _lib.PRECISIONS[self.precision])
            h = C.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(L.osud_dit_create(C.byref(cfg), C.byref(h)))
            self._handle, self._handle_key, self._uploaded = h, key, {}
            # frequency tables exactly as torch evaluates positional_embedding.py:39-44
            for name, half in (("const.freqs64", 64), ("const.freqs128", 128)):
                f = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
                self._set(name, f)
        with torch.cuda.device(dev):
            for name, p in self.named_parameters():
                stamp = (p.data_ptr(), p._version)
                if self._uploaded.get(name) != stamp:
                    self._set(name, p.detach())
                    self._uploaded[name] = stamp
        return self._handle

    def _set(self, name, tensor):
        t = tensor.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        _lib.check(_lib.lib().osud_dit_set_param(self._handle, name.encode(), _lib.ptr(t), shape, t.dim(),
                                                 _lib.stream_ptr(t.device)))

    def _free_handle(self):
        if getattr(self, "_handle", None) is not None:
            try:
                _lib.lib().osud_dit_destroy(self._handle)
            except Exception:
                pass
            self.__dict__["_handle"] = None  # not through nn.Module.__setattr__: it may be half torn down at interpreter exit

    def __del__(self):
        try:
            self._free_handle()
        except Exception:  # interpreter shutdown: modules / torch internals may already be gone
            pass

    def __getstate__(self):  # deepcopy (EMA copy, train.py:147) / pickling: never share a native handle
        state = self.__dict__.copy()
        state.update(_handle=None, _handle_key=None, _uploaded={}, _grad_ctx=None, _arena=None, _arena_bound=None,
                     _train_keep=None)
        return state

    def _apply(self, fn, *a, **k):  # .to()/.cuda() moves storage: packed copies are stale
        out = super()._apply(fn, *a, **k)
        self._uploaded = {}
        return out

    def reserve(self, max_batch, max_seq_len, training=False):
        """Pre-allocate native workspaces (otherwise done on the first call with a new shape)."""
        h = self.native_handle()
        with torch.cuda.device(self._device()):
            _lib.check(_lib.lib().osud_dit_reserve(h, int(max_batch), int(max_seq_len), int(training)))

    def calibrate_fp8(self, x, t, o, c, y, cfg_scale=None, attn_mask=None, accumulate=False):
        """fp8 tier, inference: measure the e4m3 activation scales on this batch (bf16 forward recording each block's LayerNorm /
        attention / GELU output amax; scale = 448 / (2 amax)) instead of the built-in constants.  Call it on a few timesteps with
        `accumulate=True` after the first to cover the schedule.  No effect on the other tiers' arithmetic."""
        if self.precision != "fp8":
            raise ValueError("calibrate_fp8 applies to precision='fp8' models")
        N, T = self._check_inputs(x, t, o, c, y, attn_mask)
        h = self.native_handle()
        x, t, o, c, y, m = self._prep(x, t, o, c, y, attn_mask)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().osud_dit_calibrate_fp8(h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(o), _lib.ptr(c), _lib.ptr(y), _lib.ptr(m), N, T,
                                                         float(-1.0 if cfg_scale is None else cfg_scale), int(bool(accumulate)),
                                                         _lib.stream_ptr(x.device)))

    # ------------------------------------------------------------------ forward
    def _check_inputs(self, x, t, o, c, y, attn_mask):
        assert x.dim() == 3 and x.shape[1] == self.in_channels, f"x must be (N, {self.in_channels}, T), got {tuple(x.shape)}"
        N, _, T = x.shape
        assert t.shape == (N,), f"t must be (N,), got {tuple(t.shape)}"
        assert o.shape == (N, T), f"o must be (N, T), got {tuple(o.shape)}"
        assert c.shape == (N, self.context_size, T), f"c must be (N, {self.context_size}, T), got {tuple(c.shape)}"
        assert y.shape == (N,), f"y must be (N,), got {tuple(y.shape)}"
        if attn_mask is not None:
            assert attn_mask.shape == (T, T) and attn_mask.dtype == torch.bool, "attn_mask must be a (T, T) bool tensor"
        return N, T

    def _prep(self, x, t, o, c, y, attn_mask):
        dev = self._device()
        f = lambda v: v.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        i = lambda v: v.to(device=dev, dtype=torch.int64).contiguous()  # noqa: E731
        m = None if attn_mask is None else attn_mask.to(device=dev).to(torch.uint8).contiguous()
        return f(x), i(t), f(o), f(c), i(y), m

    def _run(self, x, t, o, c, y, attn_mask, cfg_scale):
        N, T = self._check_inputs(x, t, o, c, y, attn_mask)
        h = self.native_handle()
        x, t, o, c, y, m = self._prep(x, t, o, c, y, attn_mask)
        out = torch.empty(N, self.out_channels, T, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().osud_dit_forward(h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(o), _lib.ptr(c), _lib.ptr(y),
                                                   _lib.ptr(m), N, T, float(cfg_scale), _lib.ptr(out),
                                                   _lib.stream_ptr(x.device)))
        return out

    def forward(self, x, t, o, c, y, attn_mask=None):
        """x (N,C,T) positions / playfield, t (N) timesteps, o (N,T) ms offsets, c (N,E,T) context,
        y (N) class labels -> (N, out_channels, T).  Reference: models.py:306-325."""
        if self.training and self.y_embedder.dropout_prob > 0:
            y = self.y_embedder.token_drop(y)  # models.py:69-72
        if attn_mask is None and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .training import dit_forward_autograd  # backward through the native kernels

            return dit_forward_autograd(self, x, t, o, c, y, attn_mask)
        # Inference path.  A masked forward is always inference: the reference trains without a mask (train.py:255) and the native
        # backward has none.  Asked for with grad mode on, a training-mode module refuses (a masked fine-tune would otherwise fail
        # later at backward() with an unrelated autograd error, or train on silently missing gradients); an eval-mode module --
        # sampling code that forgot no_grad -- gets its result with a one-time warning that it carries no grad_fn.
        if attn_mask is not None and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if self.training:
                raise NotImplementedError("DiT.forward(attn_mask=...) has no native backward: the masked forward is inference only "
                                          "(call it under torch.no_grad(), or train without a mask as the reference does)")
            if not getattr(self, "_warned_masked_grad", False):
                import warnings

                warnings.warn("DiT.forward(attn_mask=...) with grad mode on returns a tensor without grad_fn (the native path has no "
                              "masked backward); wrap sampling in torch.no_grad()", stacklevel=2)
                self.__dict__["_warned_masked_grad"] = True
        return self._run(x, t, o, c, y, attn_mask, -1.0)

    def forward_with_cfg(self, x, t, o, c, y, cfg_scale, attn_mask=None):
        """Batched cond/uncond forward with classifier-free guidance on the eps channels
        (models.py:327-343).  The caller passes the doubled batch [cond; uncond]."""
        assert len(x) % 2 == 0, "forward_with_cfg expects the batch doubled as [cond; uncond]"
        assert cfg_scale >= 0, "cfg_scale must be >= 0"
        with torch.no_grad():
            return self._run(x, t, o, c, y, attn_mask, float(cfg_scale))


# ---------------------------------------------------------------------------- configs (models.py:410-431)
def DiT_XL(**kwargs) -> DiT:
    return DiT(depth=28, hidden_size=1152, num_heads=16, **kwargs)


def DiT_L(**kwargs) -> DiT:
    return DiT(depth=24, hidden_size=1024, num_heads=16, **kwargs)


def DiT_B(**kwargs) -> DiT:
    return DiT(depth=12, hidden_size=768, num_heads=12, **kwargs)


def DiT_S(**kwargs) -> DiT:
    return DiT(depth=12, hidden_size=384, num_heads=6, **kwargs)


DiT_models = {"DiT-XL": DiT_XL, "DiT-L": DiT_L, "DiT-B": DiT_B, "DiT-S": DiT_S}


def find_model(ckpt_path):
    """Load a checkpoint written by train.py (takes its "ema" weights) or a bare state dict
    (reference sample.py:31-36)."""
    assert os.path.isfile(ckpt_path), f"Could not find DiT checkpoint at {ckpt_path}"
    checkpoint = torch.load(ckpt_path, map_location=lambda storage, loc: storage, weights_only=False)
    if "ema" in checkpoint:
        checkpoint = checkpoint["ema"]
    return checkpoint
