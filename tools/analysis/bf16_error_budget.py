#!/usr/bin/env python3
"""Where does the fast tier's end-to-end drift come from?  CPU emulation with the oracle (test infrastructure, not product):
bf16 rounding (round-to-nearest-even, fp32 accumulate) is switched on for ONE class of GEMM operands at a time and the
deviation from the all-fp32 oracle is measured (a) on one forward_with_cfg output, teacher-forced, and (b) on the final (x, y)
after a chained CFG-4 p_sample loop with identical noise.

    python tools/analysis/bf16_error_budget.py [--model tiny|small] [--steps 20] [--pos-gain 0.1]

Result (committed in HISTORY.md §2): every GEMM class injects about the same ~2^-9 relative error, the loop amplifies whatever
is injected; no single layer is "the" source.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import diffusion_oracle as do  # noqa: E402
from oracle import dit_oracle as mo  # noqa: E402
from osu_diffusion_amd.synthetic import synthetic_windows  # noqa: E402

ROUND = set()


DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("DTYPE", "bf16")]  # operand format that is emulated


def r(v, cls):
    return v.to(DTYPE).float() if cls in ROUND else v


def forward(sd, s, x, t, o, c, y):
    """oracle.dit_oracle.forward with per-class operand rounding hooks (same op order)."""
    xs, cs = x.swapaxes(1, 2), c.swapaxes(1, 2)
    xp = xs * sd["xoc_embedder.playfield_size"]
    N, T, C = xp.shape
    feats = torch.cat((mo.sincos_embedding(xp.reshape(-1), 128).reshape(N, T, C * 128), mo.sincos_embedding(o / 10, 128), cs), -1)
    h = r(feats, "first") @ r(sd["xoc_embedder.mlp.0.weight"], "first").T + sd["xoc_embedder.mlp.0.bias"]
    e = mo.sincos_embedding(t.float(), 256)
    th = mo.silu(r(e, "temb") @ r(sd["t_embedder.mlp.0.weight"], "temb").T + sd["t_embedder.mlp.0.bias"])
    tv = r(th, "temb") @ r(sd["t_embedder.mlp.2.weight"], "temb").T + sd["t_embedder.mlp.2.bias"]
    b = tv + sd["y_embedder.embedding_table.weight"][y]
    sb = mo.silu(b)
    for i in range(s.depth):
        p = f"blocks.{i}."
        ada = r(sb, "ada") @ r(sd[p + "adaLN_modulation.1.weight"], "ada").T + sd[p + "adaLN_modulation.1.bias"]
        sh1, sc1, g1, sh2, sc2, g2 = ada.chunk(6, dim=1)
        u = mo.modulate(mo.layer_norm(h), sh1, sc1)
        D, H = s.hidden, s.heads
        hd = D // H
        qkv = r(u, "qkv") @ r(sd[p + "attn.in_proj_weight"], "qkv").T + sd[p + "attn.in_proj_bias"]
        qkv = r(qkv, "attn")
        q, k, v = qkv.split(D, dim=-1)
        q = q.reshape(N, T, H, hd).transpose(1, 2) * hd ** -0.5
        k = k.reshape(N, T, H, hd).transpose(1, 2)
        v = v.reshape(N, T, H, hd).transpose(1, 2)
        pr = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
        a = (r(pr, "attn") @ v).transpose(1, 2).reshape(N, T, D)
        a = r(a, "proj") @ r(sd[p + "attn.out_proj.weight"], "proj").T + sd[p + "attn.out_proj.bias"]
        h = h + g1.unsqueeze(1) * a
        u2 = mo.modulate(mo.layer_norm(h), sh2, sc2)
        z = r(u2, "fc1") @ r(sd[p + "mlp.fc1.weight"], "fc1").T + sd[p + "mlp.fc1.bias"]
        m = r(mo.gelu_tanh(z), "fc2") @ r(sd[p + "mlp.fc2.weight"], "fc2").T + sd[p + "mlp.fc2.bias"]
        h = h + g2.unsqueeze(1) * m
    ada = r(sb, "ada") @ r(sd["final_layer.adaLN_modulation.1.weight"], "ada").T + sd["final_layer.adaLN_modulation.1.bias"]
    sh, sc = ada.chunk(2, dim=1)
    out = mo.modulate(mo.layer_norm(h), sh, sc) @ sd["final_layer.linear.weight"].T + sd["final_layer.linear.bias"]
    return out.swapaxes(1, 2)


def with_cfg(sd, s, x, t, o, c, y, scale):
    half = x[: len(x) // 2]
    out = forward(sd, s, torch.cat([half, half]), t, o, c, y)
    eps, rest = out[:, :2], out[:, 2:]
    ce, ue = eps.chunk(2)
    he = ue + scale * (ce - ue)
    return torch.cat([torch.cat([he, he]), rest], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="tiny")
    ap.add_argument("--steps", default="20")
    ap.add_argument("--pos-gain", type=float, default=0.1)
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--T", type=int, default=64)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    shape = mo.DitShape(depth=2, hidden=128, heads=2, num_classes=10) if a.model == "tiny" else mo.shape_of(
        {"small": "DiT-S", "base": "DiT-B"}[a.model], num_classes=10)
    sd = mo.seeded_state_dict(shape, 11, pos_gain=a.pos_gain)
    (x, o, c), y = synthetic_windows(a.n, a.T, 10, seed=1, train_offsets=False)
    x, o, c = torch.cat([x, x]), torch.cat([o, o]), torch.cat([c, c])
    y = torch.cat([y, torch.full_like(y, 10)])
    sch = do.create_schedule(a.steps, "squaredcos_cap_v2")
    torch.manual_seed(0)
    z = torch.randn(a.n, 2, a.T)
    z = torch.cat([z, z])
    noises = torch.randn(sch.num_timesteps, *z.shape)
    fn = lambda xx, tt: with_cfg(sd, shape, xx, tt, o, c, y, 4.0)  # noqa: E731
    tm = torch.from_numpy(sch.timestep_map)
    t_mid = torch.full((len(z),), sch.num_timesteps // 2, dtype=torch.long)
    ROUND.clear()
    ref_out = fn(z, tm[t_mid])
    ref_fin = do.sample_loop(sch, fn, z, noises)
    print(f"model={a.model} steps={a.steps} pos_gain={a.pos_gain}: max|out|={ref_out.abs().max():.3f}")
    print(f"{'bf16 operands in':<22}{'one forward max|d|':>20}{'rel':>10}{'loop final max|d|':>20}{'mean|d|':>12}")
    classes = ["first", "temb", "ada", "qkv", "attn", "proj", "fc1", "fc2"]
    only = os.environ.get("ONLY")  # ONLY=fast: just the fast tier's configuration (everything but the split first linear)
    cfgs = [classes[1:]] if only == "fast" else [[c_] for c_ in classes] + [["qkv", "attn", "proj", "fc1", "fc2"], classes]
    for cfg in cfgs:
        ROUND.clear()
        ROUND.update(cfg)
        out = fn(z, tm[t_mid])
        fin = do.sample_loop(sch, fn, z, noises)
        d1 = (out - ref_out).abs().max().item()
        df = (fin - ref_fin).abs()
        print(f"{'+'.join(cfg) if len(cfg) < 4 else ('trunk' if len(cfg) == 5 else ('all but first' if len(cfg) == 7 else 'all')):<22}{d1:>20.3e}{d1 / ref_out.abs().max().item():>10.1e}"
              f"{df.max().item():>20.3e}{df.mean().item():>12.3e}")


if __name__ == "__main__":
    main()
