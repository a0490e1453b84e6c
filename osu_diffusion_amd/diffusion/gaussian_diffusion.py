"""Gaussian diffusion process with the reference's API (diffusion/gaussian_diffusion.py in
the osu-diffusion repository) on top of libosud.so.

* The fp64 schedule tables come from the library's host code (``osud_sched_create``).
* When the model is the native DiT (``model.forward`` / ``model.forward_with_cfg`` / the module)
  and no Python hook (``denoised_fn`` / ``cond_fn``) is given, ``p_sample`` / ``ddim_sample`` run
  the fused native sampler update and the ``*_sample_loop`` methods run the whole loop as one
  replayed hipGraph (``osud_sample_loop``).
* Anything else (an arbitrary callable model, in-paint ``denoised_fn`` masks, ``cond_fn``,
  non-default mean/variance types) takes the generic path: same formulas with torch tensor ops
  and device-resident coefficient tables (the reference re-uploads numpy scalars every call,
  gaussian_diffusion.py:951-963).
"""
from __future__ import annotations

import ctypes as C
import enum
import math

import numpy as np
import torch as th

from .. import _lib
from .diffusion_utils import discretized_gaussian_log_likelihood, normal_kl


def mean_flat(tensor):
    """Mean over all non-batch dimensions."""
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()
    L1 = enum.auto()
    RESCALED_L1 = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


# ------------------------------------------------------------------------------ beta schedules
def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """Legacy schedule names (gaussian_diffusion.py:71-109)."""
    n = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif beta_schedule in ("warmup10", "warmup50"):
        frac = 0.1 if beta_schedule == "warmup10" else 0.5
        betas = beta_end * np.ones(n, dtype=np.float64)
        w = int(n * frac)
        betas[:w] = np.linspace(beta_start, beta_end, w, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(n, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (n,)
    return betas


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """Discretise a cumulative-alpha function (gaussian_diffusion.py:139-155)."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """gaussian_diffusion.py:112-136."""
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return get_beta_schedule("linear", beta_start=scale * 0.0001, beta_end=scale * 0.02,
                                 num_diffusion_timesteps=num_diffusion_timesteps)
    if schedule_name == "squaredcos_cap_v2":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


# ------------------------------------------------------------------------------ native schedule
class _NativeSchedule:
    """Owns an ``osud_sched*`` (fp64 tables computed by the library's host code)."""

    TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
              "log_betas"]

    def __init__(self, base_betas, use_timesteps):
        L = _lib.lib()
        base = np.ascontiguousarray(base_betas, dtype=np.float64)
        use = np.ascontiguousarray(sorted(use_timesteps), dtype=np.int64)
        h = C.c_void_p()
        _lib.check(L.osud_sched_create(base.ctypes.data_as(C.POINTER(C.c_double)), len(base),
                                       use.ctypes.data_as(C.POINTER(C.c_int64)), len(use), C.byref(h)))
        self.handle = h
        self.n = L.osud_sched_num_timesteps(h)

    def table(self, name):
        out = np.empty(self.n, dtype=np.float64)
        _lib.check(_lib.lib().osud_sched_table(self.handle, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)),
                                               self.n))
        return out

    def timestep_map(self):
        out = np.empty(self.n, dtype=np.int64)
        _lib.check(_lib.lib().osud_sched_timestep_map(self.handle, out.ctypes.data_as(C.POINTER(C.c_int64)), self.n))
        return out

    def __del__(self):
        try:
            if self.handle is not None:
                _lib.lib().osud_sched_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class InPaintMask:
    """`denoised_fn` for in-painting: keep the prediction where `mask` is True, force `known` elsewhere — what the
    reference builds as a closure (testing/test_toy.py:56-62: `torch.where(mask, x2, x)`).  Any callable still works
    as `denoised_fn` (generic Python path); an instance of this class is also understood by the fused native loops,
    which apply it inside the sampler-update kernel."""

    def __init__(self, mask, known):
        assert mask.shape == known.shape, "mask and known values must have the shape of the samples"
        self.mask, self.known = mask.bool(), known

    def __call__(self, x0):
        return th.where(self.mask, x0, self.known)

    def native(self, like):
        """(ctypes struct, tensors to keep alive) for the C ABI's `osud_inpaint`."""
        import ctypes as C

        keep = self.mask.to(device=like.device, dtype=th.uint8).expand(like.shape).contiguous()
        known = self.known.to(device=like.device, dtype=th.float32).expand(like.shape).contiguous()
        st = _lib.InPaintStruct(_lib.ptr(keep), _lib.ptr(known))
        return C.cast(C.pointer(st), C.c_void_p), (st, keep, known)


def _native_target(model):
    """(dit, uses_cfg) when `model` is the native DiT module or one of its two forward methods."""
    from ..models import DiT

    if isinstance(model, DiT):
        return model, False
    owner = getattr(model, "__self__", None)
    func = getattr(model, "__func__", None)
    if isinstance(owner, DiT):
        if func is DiT.forward_with_cfg:
            return owner, True
        if func is DiT.forward:
            return owner, False
    return None, False


class GaussianDiffusion:
    """Training and sampling utilities; `betas` are this process's own betas.
    Reference: gaussian_diffusion.py:158-211 for the tables."""

    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, _sched=None, _timestep_map=None):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        betas = np.array(betas, dtype=np.float64)
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self._sched = _sched if _sched is not None else _NativeSchedule(betas, range(len(betas)))
        s = self._sched
        self.betas = s.table("betas")
        self.num_timesteps = int(self.betas.shape[0])
        self.alphas_cumprod = s.table("alphas_cumprod")
        self.alphas_cumprod_prev = s.table("alphas_cumprod_prev")
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = s.table("sqrt_alphas_cumprod")
        self.sqrt_one_minus_alphas_cumprod = s.table("sqrt_one_minus_alphas_cumprod")
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = s.table("sqrt_recip_alphas_cumprod")
        self.sqrt_recipm1_alphas_cumprod = s.table("sqrt_recipm1_alphas_cumprod")
        self.posterior_variance = s.table("posterior_variance")
        self.posterior_log_variance_clipped = (s.table("posterior_log_variance_clipped")
                                               if self.num_timesteps > 1 else np.array([]))
        self.posterior_mean_coef1 = s.table("posterior_mean_coef1")
        self.posterior_mean_coef2 = s.table("posterior_mean_coef2")
        self._dev_tables = {}
        self._model_timestep_map = (np.arange(self.num_timesteps, dtype=np.int64) if _timestep_map is None
                                    else np.asarray(_timestep_map, dtype=np.int64))

    # ---- coefficient gather: fp64 table -> device once, gathered per call (then .float()) ----
    def _extract(self, arr, timesteps, broadcast_shape):
        key = (id(arr), timesteps.device)
        tab = self._dev_tables.get(key)
        if tab is None or tab[0] is not arr:
            tab = (arr, th.from_numpy(np.ascontiguousarray(arr)).to(timesteps.device))
            self._dev_tables[key] = tab
        res = tab[1][timesteps].float()
        while len(res.shape) < len(broadcast_shape):
            res = res[..., None]
        return res + th.zeros(broadcast_shape, device=timesteps.device)

    def _native_ok(self, model, x, denoised_fn, cond_fn):
        dit, use_cfg = _native_target(getattr(model, "model", model))
        ok = (dit is not None and (denoised_fn is None or isinstance(denoised_fn, InPaintMask)) and cond_fn is None and x.is_cuda
              and self.model_mean_type == ModelMeanType.EPSILON and self.model_var_type == ModelVarType.LEARNED_RANGE
              and dit.learn_sigma)
        return (dit, use_cfg) if ok else (None, False)

    # ------------------------------------------------------------------ forward process
    def q_mean_variance(self, x_start, t):
        mean = self._extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
        variance = self._extract(1.0 - self.alphas_cumprod, t, x_start.shape)
        log_variance = self._extract(self.log_one_minus_alphas_cumprod, t, x_start.shape)
        return mean, variance, log_variance

    def q_sample(self, x_start, t, noise=None):
        """x_t ~ q(x_t | x_0) (gaussian_diffusion.py:231-247)."""
        if noise is None:
            noise = th.randn_like(x_start)
        assert noise.shape == x_start.shape
        return (self._extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + self._extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        assert x_start.shape == x_t.shape
        mean = (self._extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + self._extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        var = self._extract(self.posterior_variance, t, x_t.shape)
        logvar = self._extract(self.posterior_log_variance_clipped, t, x_t.shape)
        assert mean.shape[0] == var.shape[0] == logvar.shape[0] == x_start.shape[0]
        return mean, var, logvar

    # ------------------------------------------------------------------ reverse process (generic)
    def _call_model(self, model, x, t, model_kwargs):
        """Apply the model at the ORIGINAL process's timestep (respace.py:127-132)."""
        tmap = self._extract_map(t)
        return model(x, tmap, **model_kwargs)

    def _extract_map(self, t):
        key = ("tmap", t.device)
        m = self._dev_tables.get(key)
        if m is None:
            m = th.from_numpy(self._model_timestep_map).to(t.device)
            self._dev_tables[key] = m
        return m.to(t.dtype)[t]

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """p(x_{t-1} | x_t) and the x_0 prediction (gaussian_diffusion.py:273-369)."""
        if model_kwargs is None:
            model_kwargs = {}
        B, Cn = x.shape[:2]
        assert t.shape == (B,)
        model_output = self._call_model(model, x, t, model_kwargs)
        extra = None
        if isinstance(model_output, tuple):
            model_output, extra = model_output
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            assert model_output.shape == (B, Cn * 2, *x.shape[2:])
            model_output, var_values = th.split(model_output, Cn, dim=1)
            if self.model_var_type == ModelVarType.LEARNED:
                log_variance = var_values
            else:
                min_log = self._extract(self.posterior_log_variance_clipped, t, x.shape)
                max_log = self._extract(self._log_betas(), t, x.shape)
                frac = (var_values + 1) / 2
                log_variance = frac * max_log + (1 - frac) * min_log
            variance = th.exp(log_variance)
        else:
            if self.model_var_type == ModelVarType.FIXED_LARGE:
                v = np.append(self.posterior_variance[1], self.betas[1:])
                variance, log_variance = v, np.log(v)
            else:
                variance, log_variance = self.posterior_variance, self.posterior_log_variance_clipped
            variance = self._extract(variance, t, x.shape)
            log_variance = self._extract(log_variance, t, x.shape)

        def process_xstart(v):
            if denoised_fn is not None:
                v = denoised_fn(v)
            return v.clamp(-1, 2) if clip_denoised else v  # the reference clamps to [-1, 2] (:345)

        if self.model_mean_type == ModelMeanType.START_X:
            pred_xstart = process_xstart(model_output)
        else:
            pred_xstart = process_xstart(self._predict_xstart_from_eps(x_t=x, t=t, eps=model_output))
        mean, _, _ = self.q_posterior_mean_variance(x_start=pred_xstart, x_t=x, t=t)
        assert mean.shape == log_variance.shape == pred_xstart.shape == x.shape
        return {"mean": mean, "variance": variance, "log_variance": log_variance, "pred_xstart": pred_xstart,
                "extra": extra}

    def _log_betas(self):
        if not hasattr(self, "_log_betas_arr"):
            self._log_betas_arr = self._sched.table("log_betas")
        return self._log_betas_arr

    def _predict_xstart_from_eps(self, x_t, t, eps):
        assert x_t.shape == eps.shape
        return (self._extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - self._extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * eps)

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        return ((self._extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart)
                / self._extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape))

    def condition_mean(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        gradient = cond_fn(x, self._extract_map(t), **(model_kwargs or {}))
        return p_mean_var["mean"].float() + p_mean_var["variance"] * gradient.float()

    def condition_score(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        alpha_bar = self._extract(self.alphas_cumprod, t, x.shape)
        eps = self._predict_eps_from_xstart(x, t, p_mean_var["pred_xstart"])
        eps = eps - (1 - alpha_bar).sqrt() * cond_fn(x, self._extract_map(t), **(model_kwargs or {}))
        out = p_mean_var.copy()
        out["pred_xstart"] = self._predict_xstart_from_eps(x, t, eps)
        out["mean"], _, _ = self.q_posterior_mean_variance(x_start=out["pred_xstart"], x_t=x, t=t)
        return out

    # ------------------------------------------------------------------ native single step
    def _native_step(self, dit, use_cfg, mode, eta, x, t, clip_denoised, model_kwargs, noise=None, inpaint=None):
        kw = dict(model_kwargs or {})
        cfg_scale = kw.pop("cfg_scale", None)
        fwd = dit.forward_with_cfg if use_cfg else dit.forward
        if use_cfg:
            kw["cfg_scale"] = cfg_scale
        with th.no_grad():
            model_out = fwd(x, self._extract_map(t), **kw).contiguous()
        if noise is None:
            noise = th.randn_like(x)
        x = x.contiguous().float()
        sample = th.empty_like(x)
        x0 = th.empty_like(x)
        N, _, T = x.shape
        ip, ip_keep = inpaint.native(x) if inpaint is not None else (None, None)
        with th.cuda.device(x.device):
            _lib.check(_lib.lib().osud_sampler_step_inpaint(
                self._sched.handle, mode, float(eta), _lib.ptr(model_out), _lib.ptr(x), _lib.ptr(t.to(th.int64).contiguous()),
                _lib.ptr(noise.contiguous()), N, T, -1.0, int(bool(clip_denoised)), ip, _lib.ptr(sample), _lib.ptr(x0),
                _lib.stream_ptr(x.device)))
        self._keepalive_step = ip_keep
        return {"sample": sample, "pred_xstart": x0}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        """One ancestral step x_t -> x_{t-1} (gaussian_diffusion.py:420-467)."""
        dit, use_cfg = self._native_ok(model, x, denoised_fn, cond_fn)
        if dit is not None and not th.is_grad_enabled():
            return self._native_step(dit, use_cfg, _lib.SAMPLER_P, 0.0, x, t, clip_denoised, model_kwargs, inpaint=denoised_fn)
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                   model_kwargs=model_kwargs)
        noise = th.randn_like(x)
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        if cond_fn is not None:
            out["mean"] = self.condition_mean(cond_fn, out, x, t, model_kwargs=model_kwargs)
        sample = out["mean"] + nonzero_mask * th.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        """One DDIM step (gaussian_diffusion.py:563-610)."""
        dit, use_cfg = self._native_ok(model, x, denoised_fn, cond_fn)
        if dit is not None and not th.is_grad_enabled():
            return self._native_step(dit, use_cfg, _lib.SAMPLER_DDIM, eta, x, t, clip_denoised, model_kwargs, inpaint=denoised_fn)
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                   model_kwargs=model_kwargs)
        if cond_fn is not None:
            out = self.condition_score(cond_fn, out, x, t, model_kwargs=model_kwargs)
        eps = self._predict_eps_from_xstart(x, t, out["pred_xstart"])
        alpha_bar = self._extract(self.alphas_cumprod, t, x.shape)
        alpha_bar_prev = self._extract(self.alphas_cumprod_prev, t, x.shape)
        sigma = eta * th.sqrt((1 - alpha_bar_prev) / (1 - alpha_bar)) * th.sqrt(1 - alpha_bar / alpha_bar_prev)
        noise = th.randn_like(x)
        mean_pred = out["pred_xstart"] * th.sqrt(alpha_bar_prev) + th.sqrt(1 - alpha_bar_prev - sigma ** 2) * eps
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        return {"sample": mean_pred + nonzero_mask * sigma * noise, "pred_xstart": out["pred_xstart"]}

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                            eta=0.0):
        """x_t -> x_{t+1} along the deterministic DDIM ODE (gaussian_diffusion.py:612-651)."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                   model_kwargs=model_kwargs)
        if cond_fn is not None:
            out = self.condition_score(cond_fn, out, x, t, model_kwargs=model_kwargs)
        eps = self._predict_eps_from_xstart(x, t, out["pred_xstart"])
        alpha_bar_next = self._extract(self.alphas_cumprod_next, t, x.shape)
        mean_pred = out["pred_xstart"] * th.sqrt(alpha_bar_next) + th.sqrt(1 - alpha_bar_next) * eps
        return {"sample": mean_pred, "pred_xstart": out["pred_xstart"]}

    # ------------------------------------------------------------------ loops
    def _native_loop(self, dit, use_cfg, mode, eta, img, clip_denoised, model_kwargs, step_noise, seed,
                     first_step=None, last_step=0, in_place=False, inpaint=None, repeat=None):
        """Steps first_step..last_step in the library: one captured hipGraph replayed per step.  repeat=k: the step at index
        first_step k times over (osud_sample_repeat) instead of a descending run."""
        kw = dict(model_kwargs or {})
        cfg_scale = float(kw.pop("cfg_scale")) if use_cfg else -1.0
        mask = kw.pop("attn_mask", None)
        o, c, y = kw.pop("o"), kw.pop("c"), kw.pop("y")
        assert not kw, f"unexpected model kwargs {sorted(kw)}"
        first_step = self.num_timesteps - 1 if first_step is None else int(first_step)
        n_steps = first_step - int(last_step) + 1 if repeat is None else int(repeat)
        x = img if in_place else img.detach().clone().float().contiguous()
        assert x.is_cuda and x.dtype == th.float32 and x.is_contiguous()
        N, T = dit._check_inputs(x, th.zeros(x.shape[0], dtype=th.long, device=x.device), o, c, y, mask)
        handle = dit.native_handle()
        _, _, o, c, y, m = dit._prep(x, th.zeros(N, dtype=th.long), o, c, y, mask)
        if step_noise is not None:
            assert step_noise.shape == (n_steps, *x.shape), "step_noise must be (number of steps, *shape)"
            step_noise = step_noise.to(device=x.device, dtype=th.float32).contiguous()
        ip, ip_keep = inpaint.native(x) if inpaint is not None else (None, None)
        keep = (x, o, c, y, m, step_noise, ip_keep)  # alive until the stream has consumed them
        with th.cuda.device(x.device):
            if repeat is None:
                _lib.check(_lib.lib().osud_sample_loop_inpaint(
                    handle, self._sched.handle, mode, float(eta), _lib.ptr(x), _lib.ptr(o), _lib.ptr(c), _lib.ptr(y), _lib.ptr(m),
                    N, T, cfg_scale, int(bool(clip_denoised)), first_step, int(last_step), _lib.ptr(step_noise), int(seed), ip,
                    _lib.stream_ptr(x.device)))
            else:
                _lib.check(_lib.lib().osud_sample_repeat(
                    handle, self._sched.handle, mode, float(eta), _lib.ptr(x), _lib.ptr(o), _lib.ptr(c), _lib.ptr(y), _lib.ptr(m),
                    N, T, cfg_scale, int(bool(clip_denoised)), first_step, int(repeat), _lib.ptr(step_noise), int(seed), ip,
                    _lib.stream_ptr(x.device)))
        self._keepalive = keep
        return x

    def p_sample_repeat(self, model, x, iters, t=0, clip_denoised=True, denoised_fn=None, model_kwargs=None, step_noise=None, seed=0):
        """`iters` applications of p_sample at the SAME step index t, i.e. the reference's refine pass (sample.py:186-205:
        `for _ in range(refine_iters): img = diffusion.p_sample(model.forward_with_cfg, img, t=0, ...)["sample"]`, run after the
        weights of --refine-ckpt were loaded) as ONE native call: the captured sampler step replayed with the step counter held
        still.  Returns the new image; `x` is left alone.  Falls back to the per-call loop where the native path does not apply
        (a Python denoised_fn other than InPaintMask, a module that is not the native DiT)."""
        dit, use_cfg = self._native_ok(model, x, denoised_fn, None)
        if dit is None or th.is_grad_enabled():
            img = x
            for _ in range(int(iters)):
                tt = th.full((img.shape[0],), int(t), device=img.device, dtype=th.long)
                img = self.p_sample(model, img, tt, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)["sample"]
            return img
        return self._native_loop(dit, use_cfg, _lib.SAMPLER_P, 0.0, x, clip_denoised, model_kwargs, step_noise, seed or 0,
                                 first_step=int(t), inpaint=denoised_fn, repeat=int(iters))

    def run_steps(self, model, x, model_kwargs, first_step, last_step=0, sampler="p", eta=0.0, clip_denoised=True,
                  step_noise=None, seed=None, denoised_fn=None):
        """Extension: run sampler steps first_step, first_step-1, ..., last_step on `x` IN PLACE with the
        native DiT (the building block of the loops; also what bench.py times).  `step_noise` is
        (n_steps, *x.shape) or None (then `seed` keys the in-kernel Philox stream)."""
        dit, use_cfg = self._native_ok(model, x, denoised_fn, None)
        if dit is None:
            raise _lib.NativeError("run_steps needs the native DiT (model.forward / model.forward_with_cfg) on a GPU")
        mode = {"p": _lib.SAMPLER_P, "ddim": _lib.SAMPLER_DDIM}[sampler]
        return self._native_loop(dit, use_cfg, mode, eta, x, clip_denoised, model_kwargs, step_noise, seed or 0,
                                 first_step=first_step, last_step=last_step, in_place=True, inpaint=denoised_fn)

    def _loop(self, step_fn, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device, progress,
              **extra):
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else th.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm

            indices = tqdm(indices)
        for i in indices:
            t = th.tensor([i] * shape[0], device=device)
            with th.no_grad():
                out = step_fn(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                              model_kwargs=model_kwargs, **extra)
                yield out
                img = out["sample"]

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, step_noise=None, seed=None):
        """Generate samples (gaussian_diffusion.py:469-512).  Extensions: ``step_noise``
        (num_timesteps, *shape) supplies the per-step noise in execution order (parity runs);
        ``seed`` selects the in-kernel Philox stream instead of torch's generator."""
        if noise is None and device is None:
            device = next(model.parameters()).device
        img = noise if noise is not None else th.randn(*shape, device=device)
        dit, use_cfg = self._native_ok(model, img, denoised_fn, cond_fn)
        if dit is not None and not progress:
            if step_noise is None and seed is None:
                step_noise = th.randn(self.num_timesteps, *img.shape, device=img.device)
            return self._native_loop(dit, use_cfg, _lib.SAMPLER_P, 0.0, img, clip_denoised, model_kwargs, step_noise,
                                     seed or 0, inpaint=denoised_fn)
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=img, clip_denoised=clip_denoised,
                                                     denoised_fn=denoised_fn, cond_fn=cond_fn,
                                                     model_kwargs=model_kwargs, device=device, progress=progress):
            final = sample
        return final["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        """Yield every intermediate step (gaussian_diffusion.py:514-561)."""
        yield from self._loop(self.p_sample, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                              device, progress)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, step_noise=None, seed=None):
        """gaussian_diffusion.py:653-684."""
        if noise is None and device is None:
            device = next(model.parameters()).device
        img = noise if noise is not None else th.randn(*shape, device=device)
        dit, use_cfg = self._native_ok(model, img, denoised_fn, cond_fn)
        if dit is not None and not progress:
            if step_noise is None and seed is None:
                step_noise = th.randn(self.num_timesteps, *img.shape, device=img.device)
            return self._native_loop(dit, use_cfg, _lib.SAMPLER_DDIM, eta, img, clip_denoised, model_kwargs,
                                     step_noise, seed or 0, inpaint=denoised_fn)
        final = None
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=img, clip_denoised=clip_denoised,
                                                        denoised_fn=denoised_fn, cond_fn=cond_fn,
                                                        model_kwargs=model_kwargs, device=device, progress=progress,
                                                        eta=eta):
            final = sample
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0):
        """gaussian_diffusion.py:686-733."""
        yield from self._loop(self.ddim_sample, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                              device, progress, eta=eta)

    # ------------------------------------------------------------------ training
    def _vb_terms_bpd(self, model, x_start, x_t, t, clip_denoised=True, model_kwargs=None):
        """One term of the variational bound in bits/dim (gaussian_diffusion.py:735-783)."""
        true_mean, _, true_lv = self.q_posterior_mean_variance(x_start=x_start, x_t=x_t, t=t)
        out = self.p_mean_variance(model, x_t, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
        kl = mean_flat(normal_kl(true_mean, true_lv, out["mean"], out["log_variance"])) / np.log(2.0)
        nll = -discretized_gaussian_log_likelihood(x_start, means=out["mean"], log_scales=0.5 * out["log_variance"])
        assert nll.shape == x_start.shape
        nll = mean_flat(nll) / np.log(2.0)
        return {"output": th.where((t == 0), nll, kl), "pred_xstart": out["pred_xstart"]}

    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """Per-sample training loss terms (gaussian_diffusion.py:785-874)."""
        if model_kwargs is None:
            model_kwargs = {}
        if noise is None:
            noise = th.randn_like(x_start)
        x_t = self.q_sample(x_start, t, noise=noise)
        terms = {}
        lt = self.loss_type
        if lt in (LossType.KL, LossType.RESCALED_KL):
            terms["loss"] = self._vb_terms_bpd(model=model, x_start=x_start, x_t=x_t, t=t, clip_denoised=False,
                                               model_kwargs=model_kwargs)["output"]
            if lt == LossType.RESCALED_KL:
                terms["loss"] *= self.num_timesteps
            return terms
        if lt not in (LossType.MSE, LossType.RESCALED_MSE, LossType.L1, LossType.RESCALED_L1):
            raise NotImplementedError(lt)
        model_output = self._call_model(model, x_t, t, model_kwargs)
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            B, Cn = x_t.shape[:2]
            assert model_output.shape == (B, Cn * 2, *x_t.shape[2:])
            model_output, var_values = th.split(model_output, Cn, dim=1)
            frozen = th.cat([model_output.detach(), var_values], dim=1)  # vb trains the variance only
            frozen_model = _Frozen(frozen)
            terms["vb"] = self._vb_terms_bpd(model=frozen_model, x_start=x_start, x_t=x_t, t=t,
                                             clip_denoised=False)["output"]
            if lt in (LossType.RESCALED_MSE, LossType.RESCALED_L1):
                terms["vb"] *= self.num_timesteps / 1000.0
        target = {ModelMeanType.PREVIOUS_X: self.q_posterior_mean_variance(x_start=x_start, x_t=x_t, t=t)[0],
                  ModelMeanType.START_X: x_start, ModelMeanType.EPSILON: noise}[self.model_mean_type]
        assert model_output.shape == target.shape == x_start.shape
        if lt in (LossType.L1, LossType.RESCALED_L1):
            terms["l1"] = mean_flat(th.abs(target - model_output))
            main = terms["l1"]
        else:
            terms["mse"] = mean_flat((target - model_output) ** 2)
            main = terms["mse"]
        terms["loss"] = main + terms["vb"] if "vb" in terms else main
        return terms

    def _prior_bpd(self, x_start):
        batch_size = x_start.shape[0]
        t = th.tensor([self.num_timesteps - 1] * batch_size, device=x_start.device)
        qt_mean, _, qt_log_variance = self.q_mean_variance(x_start, t)
        kl_prior = normal_kl(mean1=qt_mean, logvar1=qt_log_variance, mean2=0.0, logvar2=0.0)
        return mean_flat(kl_prior) / np.log(2.0)

    def calc_bpd_loop(self, model, x_start, clip_denoised=True, model_kwargs=None):
        """Full variational bound, step by step (gaussian_diffusion.py:895-948)."""
        device = x_start.device
        batch_size = x_start.shape[0]
        vb, xstart_mse, mse = [], [], []
        for t in list(range(self.num_timesteps))[::-1]:
            t_batch = th.tensor([t] * batch_size, device=device)
            noise = th.randn_like(x_start)
            x_t = self.q_sample(x_start=x_start, t=t_batch, noise=noise)
            with th.no_grad():
                out = self._vb_terms_bpd(model, x_start=x_start, x_t=x_t, t=t_batch, clip_denoised=clip_denoised,
                                         model_kwargs=model_kwargs)
            vb.append(out["output"])
            xstart_mse.append(mean_flat((out["pred_xstart"] - x_start) ** 2))
            eps = self._predict_eps_from_xstart(x_t, t_batch, out["pred_xstart"])
            mse.append(mean_flat((eps - noise) ** 2))
        vb, xstart_mse, mse = th.stack(vb, dim=1), th.stack(xstart_mse, dim=1), th.stack(mse, dim=1)
        prior_bpd = self._prior_bpd(x_start)
        return {"total_bpd": vb.sum(dim=1) + prior_bpd, "prior_bpd": prior_bpd, "vb": vb, "xstart_mse": xstart_mse,
                "mse": mse}


class _Frozen:
    """Stand-in "model" returning a precomputed output (reference uses a lambda, :835)."""

    def __init__(self, out):
        self.out = out

    def __call__(self, *args, **kwargs):
        return self.out
