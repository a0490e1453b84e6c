"""`create_diffusion` — the reference's factory (diffusion/__init__.py:10-47): same keyword arguments, same defaults,
same resulting object (a `SpacedDiffusion` over the respaced schedule)."""
from . import gaussian_diffusion as gd
from .gaussian_diffusion import InPaintMask
from .respace import SpacedDiffusion, space_timesteps

__all__ = ["create_diffusion", "SpacedDiffusion", "space_timesteps", "InPaintMask"]


def _loss_type(use_kl: bool, rescale_learned_sigmas: bool, use_l1: bool):
    """KL wins over everything; otherwise (rescaled?) x (L1 | MSE)."""
    if use_kl:
        return gd.LossType.RESCALED_KL
    table = {(True, True): gd.LossType.RESCALED_L1, (True, False): gd.LossType.RESCALED_MSE,
             (False, True): gd.LossType.L1, (False, False): gd.LossType.MSE}
    return table[(bool(rescale_learned_sigmas), bool(use_l1))]


def _var_type(learn_sigma: bool, sigma_small: bool):
    if learn_sigma:
        return gd.ModelVarType.LEARNED_RANGE
    return gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False,
                     predict_xstart=False, learn_sigma=True, rescale_learned_sigmas=False, diffusion_steps=1000,
                     use_l1=False):
    respacing = timestep_respacing if timestep_respacing not in (None, "") else [diffusion_steps]
    kept = space_timesteps(diffusion_steps, respacing)
    mean_type = gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON
    return SpacedDiffusion(use_timesteps=kept, betas=gd.get_named_beta_schedule(noise_schedule, diffusion_steps),
                           model_mean_type=mean_type, model_var_type=_var_type(learn_sigma, sigma_small),
                           loss_type=_loss_type(use_kl, rescale_learned_sigmas, use_l1))
