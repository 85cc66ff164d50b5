"""First-stage model container (reference: vwm/models/autoencoder.py:98-210, `AutoencodingEngine` minus everything Lightning):
`encode` = regularizer(encoder(x)), `decode` = decoder(z, **kwargs). State-dict names `encoder.*` / `decoder.*` as in
`first_stage_model.*` of a Vista checkpoint."""
import torch
import torch.nn as nn

from .. import ops


class DiagonalGaussianRegularizer(nn.Module):
    """vwm/modules/autoencoding/regularizers/__init__.py:24-43 over distributions.py:24-37: posterior sample (or mode) of the
    encoder's (mean | logvar) moments. `noise_fn(shape, device)` defaults to torch.randn, like the reference."""

    def __init__(self, sample: bool = True):
        super().__init__()
        self.sample = sample
        self.noise_fn = None

    def forward(self, moments, scale=1.0):
        n, c2, h, w = moments.shape
        noise = None
        if self.sample:
            make = self.noise_fn or (lambda shape, device: torch.randn(shape, device=device))
            noise = make((n, c2 // 2, h, w), moments.device).float()
        return ops.gaussian_sample(moments, noise, scale), {}


class AutoencodingEngine(nn.Module):
    def __init__(self, encoder=None, decoder=None, regularizer=None):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.regularization = regularizer if regularizer is not None else DiagonalGaussianRegularizer()

    def encode(self, x, return_reg_log=False, unregularized=False, scale=1.0):
        """autoencoder.py:188-204; `scale` folds encode_first_stage's `z * scale_factor` into the sampling kernel."""
        z = self.encoder(x)
        if unregularized:
            return z, {}
        z, reg_log = self.regularization(z, scale)
        return (z, reg_log) if return_reg_log else z

    def decode(self, z, **kwargs):
        """autoencoder.py:206-208"""
        return self.decoder(z, **kwargs)
