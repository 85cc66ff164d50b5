"""EDM-preconditioned denoiser (reference: vwm/modules/diffusionmodules/denoiser.py:10-35)."""
from typing import Dict

import torch
import torch.nn as nn

from ... import ops
from ...util import instantiate_from_config
from .denoiser_scaling import DenoiserScaling


class Denoiser(nn.Module):
    def __init__(self, scaling_config: Dict, num_frames: int = 25):
        super().__init__()
        self.scaling: DenoiserScaling = instantiate_from_config(scaling_config)
        self.num_frames = num_frames

    def possibly_quantize_sigma(self, sigma):
        return sigma

    def possibly_quantize_c_noise(self, c_noise):
        return c_noise

    @torch.no_grad()
    def forward(self, network: nn.Module, noised_input: torch.Tensor, sigma: torch.Tensor, cond: Dict, cond_mask: torch.Tensor):
        """network(x*c_in, c_noise, cond, cond_mask, num_frames)*c_out + x*c_skip, with per-image coefficients."""
        sigma = self.possibly_quantize_sigma(sigma)
        c_skip, c_out, c_in, c_noise = self.scaling(sigma.float())
        c_noise = self.possibly_quantize_c_noise(c_noise)
        x = noised_input.float()
        net = network(ops.scale_rows(x, c_in), c_noise, cond, cond_mask, self.num_frames)
        return ops.denoiser_combine(net.float(), x, c_out, c_skip)
