"""Gaussian KL / discretised log-likelihood helpers used by the variational-bound term of the
training loss (reference: diffusion/diffusion_utils.py:9-89)."""
import numpy as np
import torch as th


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL(N(mean1, e^logvar1) || N(mean2, e^logvar2)), broadcasting scalars."""
    tensor = next((v for v in (mean1, logvar1, mean2, logvar2) if isinstance(v, th.Tensor)), None)
    assert tensor is not None, "at least one argument must be a Tensor"
    logvar1, logvar2 = (v if isinstance(v, th.Tensor) else th.tensor(v).to(tensor) for v in (logvar1, logvar2))
    return 0.5 * (-1.0 + logvar2 - logvar1 + th.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * th.exp(-logvar2))


def approx_standard_normal_cdf(x):
    """tanh approximation of the standard normal CDF."""
    return 0.5 * (1.0 + th.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * th.pow(x, 3))))


def continuous_gaussian_log_likelihood(x, *, means, log_scales):
    centered = x - means
    normalized = centered * th.exp(-log_scales)
    return th.distributions.Normal(th.zeros_like(x), th.ones_like(x)).log_prob(normalized)


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    """log P(x) under a Gaussian discretised into 1/255-wide bins with open tails beyond +-0.999
    (the reference keeps these image-era constants although x are coordinates)."""
    assert x.shape == means.shape == log_scales.shape
    centered = x - means
    inv_std = th.exp(-log_scales)
    cdf_plus = approx_standard_normal_cdf(inv_std * (centered + 1.0 / 255.0))
    cdf_min = approx_standard_normal_cdf(inv_std * (centered - 1.0 / 255.0))
    log_cdf_plus = th.log(cdf_plus.clamp(min=1e-12))
    log_one_minus_cdf_min = th.log((1.0 - cdf_min).clamp(min=1e-12))
    cdf_delta = cdf_plus - cdf_min
    log_probs = th.where(x < -0.999, log_cdf_plus,
                         th.where(x > 0.999, log_one_minus_cdf_min, th.log(cdf_delta.clamp(min=1e-12))))
    assert log_probs.shape == x.shape
    return log_probs
