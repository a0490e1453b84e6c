"""Gaussian KL / discretised log-likelihood helpers used by the variational-bound term of the
training loss (reference: diffusion/diffusion_utils.py:9-89).  Operation order is kept exactly (the generic Python
path must round like the reference)."""
import numpy as np
import torch as th

_SQRT_2_OVER_PI = np.sqrt(2.0 / np.pi)
_BIN = 1.0 / 255.0   # half width of a discretisation bin (image-era constant kept by the reference)
_EDGE = 0.999        # beyond +-_EDGE the tails are open
_FLOOR = 1e-12


def _as_tensor_like(value, like):
    return value if isinstance(value, th.Tensor) else th.tensor(value).to(like)


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL(N(mean1, e^logvar1) || N(mean2, e^logvar2)), broadcasting scalars."""
    anchor = next((v for v in (mean1, logvar1, mean2, logvar2) if isinstance(v, th.Tensor)), None)
    assert anchor is not None, "at least one argument must be a Tensor"
    lv1, lv2 = _as_tensor_like(logvar1, anchor), _as_tensor_like(logvar2, anchor)
    return 0.5 * (-1.0 + lv2 - lv1 + th.exp(lv1 - lv2) + ((mean1 - mean2) ** 2) * th.exp(-lv2))


def approx_standard_normal_cdf(x):
    """tanh approximation of the standard normal CDF."""
    return 0.5 * (1.0 + th.tanh(_SQRT_2_OVER_PI * (x + 0.044715 * th.pow(x, 3))))


def continuous_gaussian_log_likelihood(x, *, means, log_scales):
    z = (x - means) * th.exp(-log_scales)
    return th.distributions.Normal(th.zeros_like(x), th.ones_like(x)).log_prob(z)


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    """log P(x) under a Gaussian discretised into 1/255-wide bins with open tails beyond +-0.999
    (the reference keeps these image-era constants although x are coordinates)."""
    assert x.shape == means.shape == log_scales.shape
    delta, inv_std = x - means, th.exp(-log_scales)
    upper = approx_standard_normal_cdf(inv_std * (delta + _BIN))
    lower = approx_standard_normal_cdf(inv_std * (delta - _BIN))
    inside = th.log((upper - lower).clamp(min=_FLOOR))
    left_tail = th.log(upper.clamp(min=_FLOOR))
    right_tail = th.log((1.0 - lower).clamp(min=_FLOOR))
    out = th.where(x < -_EDGE, left_tail, th.where(x > _EDGE, right_tail, inside))
    assert out.shape == x.shape
    return out
