"""`create_diffusion` — the reference's factory (diffusion/__init__.py:10-47), same keyword
arguments and defaults."""
from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False,
                     predict_xstart=False, learn_sigma=True, rescale_learned_sigmas=False, diffusion_steps=1000,
                     use_l1=False):
    betas = gd.get_named_beta_schedule(noise_schedule, diffusion_steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_L1 if use_l1 else gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.L1 if use_l1 else gd.LossType.MSE
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    if learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(diffusion_steps, timestep_respacing),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON,
        model_var_type=var_type,
        loss_type=loss_type,
    )


__all__ = ["create_diffusion", "SpacedDiffusion", "space_timesteps"]
