"""Timestep respacing (reference: diffusion/respace.py).  The beta re-derivation itself is
done by the library's host code (osud_sched_create walks the base alphas_cumprod)."""
import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion, _NativeSchedule


def space_timesteps(num_timesteps, section_counts):
    """Pick which of the base process's steps to keep (respace.py:11-61): a list / comma string of
    per-section counts, or "ddimN" for the fixed integer stride that yields exactly N steps."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == desired:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(v) for v in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that keeps only `use_timesteps` of a base process (respace.py:64-117)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        base_betas = np.array(kwargs["betas"], dtype=np.float64)
        self.original_num_steps = len(base_betas)
        sched = _NativeSchedule(base_betas, [i for i in range(len(base_betas)) if i in self.use_timesteps])
        self.timestep_map = [int(v) for v in sched.timestep_map()]
        kwargs["betas"] = sched.table("betas")
        super().__init__(_sched=sched, _timestep_map=self.timestep_map, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.original_num_steps)

    def _call_model(self, model, x, t, model_kwargs):
        if isinstance(model, _WrappedModel):  # already maps step index -> original timestep
            return model(x, t, **model_kwargs)
        return super()._call_model(model, x, t, model_kwargs)

    def _scale_timesteps(self, t):
        return t


class _WrappedModel:
    """Callable that feeds the model the ORIGINAL process's timestep (respace.py:120-132); the map
    tensor is cached per device instead of being rebuilt on every call."""

    def __init__(self, model, timestep_map, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.original_num_steps = original_num_steps
        self._maps = {}

    def __call__(self, x, ts, **kwargs):
        key = (ts.device, ts.dtype)
        m = self._maps.get(key)
        if m is None:
            m = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
            self._maps[key] = m
        return self.model(x, m[ts], **kwargs)
