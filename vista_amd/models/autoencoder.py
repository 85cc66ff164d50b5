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


class AutoencoderKLModeOnly(nn.Module):
    """vwm/models/autoencoder.py:432-529 (AutoencodingEngineLegacy with a mode-only DiagonalGaussianRegularizer) as the conditioner uses
    it (configs/inference/vista.yaml:71-95, `VideoPredictionEmbedderWithEncoder(is_ae=True)`): encode(x) = mean of
    quant_conv(encoder(x)). The 1x1 `quant_conv` directly follows the encoder's 3x3 `conv_out` with nothing in between, so the two are
    ONE 3x3 convolution (W' = W_q W_c, b' = W_q b_c + b_q, composed in fp32 at pack time). The legacy class also owns a decoder and a
    post_quant_conv that inference never calls; they are not built (their checkpoint keys are reported as unexpected by
    load_state_dict(strict=False), like every unused key of the reference)."""

    def __init__(self, embed_dim, ddconfig, **ignored):
        super().__init__()
        from ..modules.attention import Packable  # noqa: F401
        from ..modules.diffusionmodules.model import Encoder
        from ..modules.diffusionmodules.util import ConvNd
        self.encoder = Encoder(**ddconfig)
        zc = (2 if ddconfig.get("double_z", True) else 1)
        self.quant_conv = ConvNd(zc * ddconfig["z_channels"], zc * embed_dim, (1, 1))
        self.embed_dim, self.double_z = embed_dim, bool(ddconfig.get("double_z", True))
        self._pk, self._pk_key = None, None
        # any load_state_dict on this module drops the composed conv_out x quant_conv pack and every pack underneath (inference
        # tensors carry no version counter, so the key below cannot see an in-place load into them)
        from ..modules.attention import _invalidate_after_load

        def _drop(module, incompatible_keys):
            module._pk = None
            _invalidate_after_load(module, incompatible_keys)
        self.register_load_state_dict_post_hook(_drop)

    def _packed(self):
        co, qc = self.encoder.conv_out, self.quant_conv
        ps = (co.weight, co.bias, qc.weight, qc.bias)
        key = tuple((id(p), p.device, 0 if p.is_inference() else p._version) for p in ps)
        if self._pk is None or self._pk_key != key:
            with torch.no_grad():
                wq = qc.weight.detach().float()[:, :, 0, 0]                                     # (out, 8)
                w = torch.einsum("om,mikl->oikl", wq, co.weight.detach().float())               # (out, C, 3, 3)
                b = wq @ co.bias.detach().float() + qc.bias.detach().float()
                self._pk, self._pk_key = ops.pack_conv3x3(w, b, device=co.weight.device), key
        return self._pk

    @ops.bf16_storage   # (the first stage stores bf16 in every process: ops.storage)
    def encode(self, x, return_reg_log=False, scale=1.0):
        h, H, W = self.encoder.features(x)
        n = h.shape[0]
        mom, _, _ = ops.conv3x3(h, self._packed(), n, H, W, out_f32=True)                      # (n, H*W, >= 2*embed) f32 moments, token-major
        nz = self.embed_dim
        z = ops.tokens_to_nchw(mom, n, nz, H, W)                                               # mode = the mean half (distributions.py:24-37)
        if scale != 1.0:
            z = ops.scale_rows(z, torch.full((n,), float(scale), device=z.device))
        return (z, {}) if return_reg_log else z
