"""Timestep samplers for the training objective (the reference's diffusion/timestep_sampler.py — unused by its scripts, kept
for interface parity): ``create_named_schedule_sampler("uniform" | "loss-second-moment", diffusion)`` returns an object with

* ``sample(batch_size, device) -> (timesteps, weights)``: importance sampling with ``numpy.random.choice`` on the normalised
  ``weights()`` and the unbiasing factors ``1 / (T * p[t])`` (timestep_sampler.py:41-56), and, for the loss-aware kind,
* ``update_with_local_losses(local_ts, local_losses)`` (all ranks' (t, loss) pairs gathered, timestep_sampler.py:67-96) /
  ``update_with_all_losses(ts, losses)`` (:115-150): per timestep the last ``history_per_term`` losses; once every timestep has
  a full history the sampling weight of t is the root-mean-square of its history, mixed with ``uniform_prob`` of uniform mass.

``NativeTrainer.step(..., t=ts, loss_weights=w)`` consumes the pair.  The history is a ring per timestep here (the reference
shifts an array): the weights only depend on which losses are in the window, not on their order.
"""
from __future__ import annotations

import numpy as np
import torch as th


class ScheduleSampler:
    """A distribution over diffusion steps; subclasses provide ``weights()`` (positive, not necessarily normalised)."""

    def weights(self):
        raise NotImplementedError

    def sample(self, batch_size, device):
        w = np.asarray(self.weights(), dtype=np.float64)
        p = w / np.sum(w)
        picked = np.random.choice(len(p), size=(batch_size,), p=p)
        unbias = 1 / (len(p) * p[picked])
        return th.from_numpy(picked).long().to(device), th.from_numpy(unbias).float().to(device)


class UniformSampler(ScheduleSampler):
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


class LossAwareSampler(ScheduleSampler):
    def update_with_local_losses(self, local_ts, local_losses):
        """Every rank contributes its batch of (timestep, loss) pairs; all ranks then apply the same update in rank order."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return self.update_with_all_losses(local_ts.tolist(), local_losses.tolist())
        world = dist.get_world_size()
        sizes = [th.zeros(1, dtype=th.int32, device=local_ts.device) for _ in range(world)]
        dist.all_gather(sizes, th.tensor([len(local_ts)], dtype=th.int32, device=local_ts.device))
        sizes = [int(s.item()) for s in sizes]
        width = max(sizes)
        pad_t = th.zeros(width, dtype=local_ts.dtype, device=local_ts.device)
        pad_l = th.zeros(width, dtype=local_losses.dtype, device=local_losses.device)
        pad_t[: len(local_ts)] = local_ts
        pad_l[: len(local_losses)] = local_losses
        all_t = [th.zeros_like(pad_t) for _ in range(world)]
        all_l = [th.zeros_like(pad_l) for _ in range(world)]
        dist.all_gather(all_t, pad_t)
        dist.all_gather(all_l, pad_l)
        ts = [int(v) for row, n in zip(all_t, sizes) for v in row[:n].tolist()]
        losses = [float(v) for row, n in zip(all_l, sizes) for v in row[:n].tolist()]
        self.update_with_all_losses(ts, losses)

    def update_with_all_losses(self, ts, losses):
        raise NotImplementedError


class LossSecondMomentResampler(LossAwareSampler):
    def __init__(self, diffusion, history_per_term=10, uniform_prob=0.001):
        self.diffusion = diffusion
        self.history_per_term = history_per_term
        self.uniform_prob = uniform_prob
        self._loss_history = np.zeros([diffusion.num_timesteps, history_per_term], dtype=np.float64)
        self._loss_counts = np.zeros([diffusion.num_timesteps], dtype=np.int64)
        self._next = np.zeros([diffusion.num_timesteps], dtype=np.int64)  # ring position of the oldest entry

    def _warmed_up(self):
        return bool((self._loss_counts == self.history_per_term).all())

    def weights(self):
        n = self.diffusion.num_timesteps
        if not self._warmed_up():
            return np.ones([n], dtype=np.float64)
        w = np.sqrt(np.mean(self._loss_history ** 2, axis=-1))
        w /= np.sum(w)
        w *= 1 - self.uniform_prob
        w += self.uniform_prob / len(w)
        return w

    def update_with_all_losses(self, ts, losses):
        for t, loss in zip(ts, losses):
            t = int(t)
            self._loss_history[t, self._next[t]] = loss
            self._next[t] = (self._next[t] + 1) % self.history_per_term
            if self._loss_counts[t] < self.history_per_term:
                self._loss_counts[t] += 1


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    if name == "loss-second-moment":
        return LossSecondMomentResampler(diffusion)
    raise NotImplementedError(f"unknown schedule sampler: {name}")
