"""osu-diffusion's DiT denoising path, native on AMD MI355X (gfx950).

Public surface (mirrors the reference repository's modules):

    from osu_diffusion_amd.models import DiT_models, find_model
    from osu_diffusion_amd.diffusion import create_diffusion
"""
__all__ = ["models", "diffusion", "synthetic"]
