"""Reward estimation around the sampler (reference: reward_utils.py:284-341, `do_sample`; driver reward.py:225-250):
`ensemble_size` sampling runs from the same conditioning with fresh noise; the reward is exp(-mean over all latent elements
of the unbiased ensemble variance) -- low disagreement between the imagined futures = high reward.

Pure re-use of the hot path: the ensemble members run through the same sampler/denoiser objects;
only the variance reduction is new (one fixed-order HIP reduction, vk_ensemble_variance_sum). `model` is duck-typed like in
vista_amd.sample_utils (VistaPipeline or the reference engine); `get_condition` / `noise_fn` are the same optional hooks.
"""
import math

import torch

from . import ops
from .modules.diffusionmodules.denoiser import Denoiser
from .modules.diffusionmodules.sampling import FusedDenoiser


@torch.no_grad()
def do_sample(images, model, sampler, value_dict, num_frames, ensemble_size: int = 5, force_uc_zero_embeddings=None,
              initial_cond_indices=None, device="cuda", get_condition=None, noise_fn=None, fused=True):
    """-> (images, reward) with reward a 0-dim CPU tensor, like the reference."""
    if ensemble_size < 2:
        raise ValueError("reward estimation needs at least two ensemble members (unbiased variance)")
    initial_cond_indices = [0] if initial_cond_indices is None else initial_cond_indices
    force_uc_zero_embeddings = [] if force_uc_zero_embeddings is None else force_uc_zero_embeddings
    get_condition = get_condition or getattr(model, "condition_fn", None)
    if get_condition is None:
        raise ValueError("do_sample: no conditioner -- pass get_condition=")
    noise_fn = noise_fn or torch.randn_like

    def denoiser(x, sigma, cond, cond_mask):
        return model.denoiser(model.model, x, sigma, cond, cond_mask)
    if fused and isinstance(model.denoiser, Denoiser):
        denoiser = FusedDenoiser(model.denoiser, model.model)

    with model.ema_scope("Sampling"):
        z = model.encode_first_stage(images)
        cond_mask = torch.zeros(num_frames, device=device)
        cond_mask[initial_cond_indices] = 1
        c, uc = get_condition(model, value_dict, num_frames, force_uc_zero_embeddings, device)
        members = []
        for _ in range(ensemble_size):
            sample = sampler(denoiser, noise_fn(z), cond=c, uc=uc, cond_frame=z, cond_mask=cond_mask)
            sample[0] = z[0]
            members.append(sample.float())
        stacked = torch.stack(members).contiguous()                     # (E, T, 4, h, w)
        var_sum = ops.ensemble_variance_sum(stacked)                    # sum over elements of sum_e (x_e - mean_e)^2 / (E - 1)
        reward = torch.tensor(math.exp(-var_sum / stacked[0].numel()))  # exp(-variance.mean())
    return images, reward
