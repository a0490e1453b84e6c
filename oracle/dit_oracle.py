"""Oracle: functional CPU restatement of the reference DiT forward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch-CPU tensor math on a
state dict that uses the reference's parameter names; no nn.Module, no kernels.
Written from the algorithm spec; every function cites the
reference lines it restates.  All citations are relative to /root/reference.

The restatement is differentiable (plain torch ops), so ``torch.autograd`` over it is
the gradient oracle for the training path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

# name -> (depth, hidden, heads)                       models.py:410-431
CONFIGS = {
    "DiT-XL": (28, 1152, 16),
    "DiT-L": (24, 1024, 16),
    "DiT-B": (12, 768, 12),
    "DiT-S": (12, 384, 6),
}


@dataclass(frozen=True)
class DitShape:
    depth: int
    hidden: int
    heads: int
    context: int = 144  # feature_size - 3 + 128              sample.py:71, train.py:143
    in_channels: int = 2
    num_classes: int = 52670  # table has num_classes + 1 rows when dropout > 0  models.py:48-52
    learn_sigma: bool = True

    @property
    def out_channels(self) -> int:  # models.py:259
        return self.in_channels * 2 if self.learn_sigma else self.in_channels

    @property
    def first_in(self) -> int:  # models.py:213-217
        return self.in_channels * 128 + 128 + self.context


def shape_of(name: str, **kw) -> DitShape:
    d, h, nh = CONFIGS[name]
    return DitShape(depth=d, hidden=h, heads=nh, **kw)


def param_shapes(s: DitShape, table_rows: int | None = None) -> "dict[str, tuple]":
    """Reference state-dict keys and shapes, in ``named_parameters()`` order
    (models.py:262-272)."""
    D = s.hidden
    rows = s.num_classes + 1 if table_rows is None else table_rows
    out = {
        "xoc_embedder.playfield_size": (2,),
        "xoc_embedder.mlp.0.weight": (D, s.first_in),
        "xoc_embedder.mlp.0.bias": (D,),
        "t_embedder.mlp.0.weight": (D, 256),
        "t_embedder.mlp.0.bias": (D,),
        "t_embedder.mlp.2.weight": (D, D),
        "t_embedder.mlp.2.bias": (D,),
        "y_embedder.embedding_table.weight": (rows, D),
    }
    for i in range(s.depth):
        p = f"blocks.{i}."
        out[p + "attn.in_proj_weight"] = (3 * D, D)
        out[p + "attn.in_proj_bias"] = (3 * D,)
        out[p + "attn.out_proj.weight"] = (D, D)
        out[p + "attn.out_proj.bias"] = (D,)
        out[p + "mlp.fc1.weight"] = (4 * D, D)
        out[p + "mlp.fc1.bias"] = (4 * D,)
        out[p + "mlp.fc2.weight"] = (D, 4 * D)
        out[p + "mlp.fc2.bias"] = (D,)
        out[p + "adaLN_modulation.1.weight"] = (6 * D, D)
        out[p + "adaLN_modulation.1.bias"] = (6 * D,)
    out["final_layer.linear.weight"] = (s.out_channels, D)
    out["final_layer.linear.bias"] = (s.out_channels,)
    out["final_layer.adaLN_modulation.1.weight"] = (2 * D, D)
    out["final_layer.adaLN_modulation.1.bias"] = (2 * D,)
    return out


def seeded_state_dict(s: DitShape, seed: int, table_rows: int | None = None, mod_std: float = 0.02,
                      bias_std: float = 0.02, gain: float = 1.0,
                      pos_gain: float = 0.1) -> "dict[str, torch.Tensor]":
    """Deterministic *non-degenerate* weights for tests and benches.

    The reference zero-inits every adaLN and the final projection (models.py:295-304),
    which makes a fresh model output exactly 0 and every block an identity — useless
    for parity.  This is "reference init, perturbed": trunk Linear weights are Xavier-
    uniform (models.py:277-283) times ``gain``; embedder weights and the class table are
    N(0, 0.02) (:286-293); the zero-init adaLN / final layers get N(0, mod_std) and all
    biases N(0, bias_std).  Drawn from a seeded torch CPU generator in key order, so the
    golden generator, the tests and the bench rebuild identical weights without storing
    them (fixtures pin a checksum).

    ``pos_gain`` scales the first-layer columns that read the sin/cos *position*
    features (arguments up to 512 rad per unit x, so d(feature)/dx ~ 512).  With
    random weights at pos_gain=1 the sampler map is expansive: the reference's own
    fp32 path and an fp64 evaluation of the same weights end 0.19 apart after a
    20-step CFG-4 run, i.e. end-to-end parity is ill-posed for ANY implementation.
    pos_gain=0.1 gives a non-expansive denoiser (fp32 vs fp64: 3e-5 after 20 or 100
    steps) on which the 1e-3 end-to-end bound is meaningful; pos_gain=1 ("rough")
    is used for single-forward / teacher-forced tests.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in param_shapes(s, table_rows).items():
        if k.endswith("playfield_size"):
            sd[k] = torch.tensor((512.0, 384.0))
        elif len(shp) == 1:
            sd[k] = torch.randn(shp, generator=g) * bias_std
        elif "adaLN" in k or k.startswith("final_layer"):
            sd[k] = torch.randn(shp, generator=g) * mod_std
        elif "embedding_table" in k or "embedder" in k:
            sd[k] = torch.randn(shp, generator=g) * 0.02
            if k == "xoc_embedder.mlp.0.weight":
                sd[k][:, : s.in_channels * 128] *= pos_gain
        else:
            a = gain * math.sqrt(6.0 / (shp[0] + shp[1]))
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * a
    return sd


# --------------------------------------------------------------------------- embeddings


def freqs(half: int) -> torch.Tensor:
    """positional_embedding.py:39-44 — same fp32 op order (python float * arange / half)."""
    return torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)


def sincos_embedding(v: torch.Tensor, dim: int) -> torch.Tensor:
    """positional_embedding.py:29-49 for any leading shape: [..] -> [.., dim], cos first."""
    f = freqs(dim // 2).to(v.dtype)
    args = v[..., None] * f
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def first_layer(sd, x_ntc, o_nt, c_nte):
    """models.py:227-235 (+ positional_embedding.py:52-77).
    x_ntc (N,T,2) in [0,1] units, o (N,T) ms, c (N,T,E)."""
    dt = sd["xoc_embedder.mlp.0.weight"].dtype
    xp = x_ntc * sd["xoc_embedder.playfield_size"]  # models.py:229
    N, T, C = xp.shape
    x_freq = sincos_embedding(xp.reshape(-1), 128)  # positional_embedding.py:75-76
    x_freq = x_freq.reshape(N, T, C * 128)  # positional_embedding.py:74-77
    o_freq = sincos_embedding(o_nt / 10, 128)  # models.py:232
    xoc = torch.cat((x_freq, o_freq, c_nte), -1)  # models.py:233
    return xoc @ sd["xoc_embedder.mlp.0.weight"].T + sd["xoc_embedder.mlp.0.bias"]


def silu(v):
    return v * torch.sigmoid(v)


def t_embed(sd, t):
    """models.py:35-38."""
    dt = sd["t_embedder.mlp.0.weight"].dtype
    e = sincos_embedding(t.to(dt), 256)  # t[:, None].float() * freqs, positional_embedding.py:45
    h = silu(e @ sd["t_embedder.mlp.0.weight"].T + sd["t_embedder.mlp.0.bias"])
    return h @ sd["t_embedder.mlp.2.weight"].T + sd["t_embedder.mlp.2.bias"]


def y_embed(sd, y, drop_mask=None, num_classes=None):
    """models.py:56-74.  ``drop_mask`` (bool, N) restates token_drop with the random
    draw made by the caller (rand(N) < p); None = eval mode."""
    if drop_mask is not None:
        y = torch.where(drop_mask, torch.full_like(y, num_classes), y)
    return sd["y_embedder.embedding_table.weight"][y]


def layer_norm(h, eps=1e-6):
    """nn.LayerNorm(elementwise_affine=False, eps=1e-6): biased variance. models.py:129,136,185."""
    mu = h.mean(-1, keepdim=True)
    var = ((h - mu) ** 2).mean(-1, keepdim=True)
    return (h - mu) / torch.sqrt(var + eps)


def modulate(v, shift, scale):
    """models.py:12-13."""
    return v * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def gelu_tanh(z):
    """nn.GELU(approximate='tanh'), models.py:138."""
    return 0.5 * z * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (z + 0.044715 * z ** 3)))


def attention(sd, p, u, heads, attn_mask):
    """nn.MultiheadAttention(batch_first=True) self-attention, models.py:130-135,164-170.
    Packed in_proj rows are [Wq;Wk;Wv]; q scaled by hd**-0.5; bool mask True = masked."""
    N, T, D = u.shape
    hd = D // heads
    qkv = u @ sd[p + "attn.in_proj_weight"].T + sd[p + "attn.in_proj_bias"]
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(N, T, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.reshape(N, T, heads, hd).transpose(1, 2)
    v = v.reshape(N, T, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)  # (N,H,T,T)
    if attn_mask is not None:
        s = s.masked_fill(attn_mask, float("-inf"))
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(N, T, D)
    return a @ sd[p + "attn.out_proj.weight"].T + sd[p + "attn.out_proj.bias"]


def block(sd, i, h, b, heads, attn_mask):
    """models.py:151-175."""
    p = f"blocks.{i}."
    ada = silu(b) @ sd[p + "adaLN_modulation.1.weight"].T + sd[p + "adaLN_modulation.1.bias"]
    sh1, sc1, g1, sh2, sc2, g2 = ada.chunk(6, dim=1)
    h = h + g1.unsqueeze(1) * attention(sd, p, modulate(layer_norm(h), sh1, sc1), heads, attn_mask)
    u2 = modulate(layer_norm(h), sh2, sc2)
    m = gelu_tanh(u2 @ sd[p + "mlp.fc1.weight"].T + sd[p + "mlp.fc1.bias"])
    m = m @ sd[p + "mlp.fc2.weight"].T + sd[p + "mlp.fc2.bias"]
    return h + g2.unsqueeze(1) * m


def final_layer(sd, h, b):
    """models.py:192-196."""
    ada = silu(b) @ sd["final_layer.adaLN_modulation.1.weight"].T + sd["final_layer.adaLN_modulation.1.bias"]
    sh, sc = ada.chunk(2, dim=1)
    return modulate(layer_norm(h), sh, sc) @ sd["final_layer.linear.weight"].T + sd["final_layer.linear.bias"]


def forward(sd, s: DitShape, x, t, o, c, y, attn_mask=None, drop_mask=None):
    """DiT.forward, models.py:306-325.  x (N,2,T), t (N) int, o (N,T), c (N,E,T), y (N) int.
    Returns (N, out_channels, T)."""
    dt = sd["xoc_embedder.mlp.0.weight"].dtype
    xs = x.to(dt).swapaxes(1, 2)  # models.py:315
    cs = c.to(dt).swapaxes(1, 2)  # models.py:316
    h = first_layer(sd, xs, o.to(dt), cs)  # models.py:317
    b = t_embed(sd, t) + y_embed(sd, y, drop_mask, s.num_classes)  # models.py:318-320
    for i in range(s.depth):  # models.py:321-322
        h = block(sd, i, h, b, s.heads, attn_mask)
    out = final_layer(sd, h, b)  # models.py:323
    return out.swapaxes(1, 2)  # models.py:324


def forward_with_cfg(sd, s: DitShape, x, t, o, c, y, cfg_scale, attn_mask=None):
    """DiT.forward_with_cfg, models.py:327-343."""
    half = x[: len(x) // 2]
    combined = torch.cat([half, half], dim=0)
    out = forward(sd, s, combined, t, o, c, y, attn_mask)
    C = s.in_channels
    eps, rest = out[:, :C], out[:, C:]
    cond_eps, uncond_eps = torch.split(eps, len(eps) // 2, dim=0)
    half_eps = uncond_eps + cfg_scale * (cond_eps - uncond_eps)
    eps = torch.cat([half_eps, half_eps], dim=0)
    return torch.cat([eps, rest], dim=1)


def to_dtype(sd, dtype):
    return {k: v.to(dtype) for k, v in sd.items()}
