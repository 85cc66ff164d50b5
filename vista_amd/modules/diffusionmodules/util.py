"""Parameter containers and small helpers (reference: vwm/modules/diffusionmodules/util.py:141-318).

The nn.Module classes here own parameters under the reference's names and shapes so `vista.safetensors` loads
unchanged; they carry no eager arithmetic. The compute is done by the parent modules through vista_amd.ops (HIP).
"""
import math

import torch
import torch.nn as nn

from ... import ops


class _Params(nn.Module):
    """Leaf parameter container; `forward` is intentionally absent (no eager fallback)."""

    def forward(self, *a, **k):
        raise RuntimeError(f"{self.__class__.__name__} is a parameter container; compute runs in the parent's HIP path")


class Linear(_Params):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        with torch.no_grad():
            self.weight.normal_(0, in_features ** -0.5)
            if bias:
                self.bias.zero_()


class ConvNd(_Params):
    """nn.Conv2d / nn.Conv3d weight layout [Cout][Cin][k...]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = tuple(kernel_size)
        self.stride, self.padding = stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        with torch.no_grad():
            self.weight.normal_(0, (in_channels * math.prod(self.kernel_size)) ** -0.5)
            self.bias.zero_()


class NormParams(_Params):
    """GroupNorm / LayerNorm affine parameters."""

    def __init__(self, channels, eps, num_groups=None):
        super().__init__()
        self.num_channels, self.eps, self.num_groups = channels, eps, num_groups
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))


def GroupNorm32(num_groups, channels):
    """util.py:196-216 (32 groups, eps 1e-5, fp32 statistics)."""
    return NormParams(channels, 1e-5, num_groups)


def normalization(channels):
    return GroupNorm32(32, channels)


def LayerNorm(channels):
    return NormParams(channels, 1e-5)


def conv_nd(dims, in_channels, out_channels, kernel_size, stride=1, padding=0, causal=False):
    """util.py:236-251"""
    if causal:
        raise NotImplementedError("causal temporal conv is not on the Vista inference path (video_model.py:51)")
    if isinstance(kernel_size, int):
        kernel_size = (kernel_size,) * dims
    if dims not in (2, 3):
        raise ValueError(f"Unsupported dimensions: {dims}")
    return ConvNd(in_channels, out_channels, kernel_size, stride, padding)


def linear(*args, **kwargs):
    return Linear(*args, **kwargs)


def zero_module(module):
    """util.py:168-175"""
    for p in module.parameters():
        p.detach().zero_()
    return module


class SiLU(nn.Module):
    """Placeholder so Sequential indices match the reference (`time_embed.0/.2`, `in_layers.0/.2`, `out_layers.0/.3`)."""

    def forward(self, x):
        raise RuntimeError("SiLU placeholder: fused into the HIP kernels of the parent module")


class Dropout(SiLU):
    def __init__(self, p=0.0):
        super().__init__()
        self.p = p


class AlphaBlender(nn.Module):
    """util.py:277-318. For 'learned_with_images' the reference's alpha is the scalar sigmoid(mix_factor)
    broadcast by the rearrange pattern; the blend itself is fused into GEMM epilogues of the parent."""
    strategies = ["learned", "fixed", "learned_with_images"]

    def __init__(self, alpha, merge_strategy, rearrange_pattern):
        super().__init__()
        assert merge_strategy in self.strategies, f"merge_strategy needs to be in {self.strategies}"
        self.merge_strategy = merge_strategy
        self.rearrange_pattern = rearrange_pattern
        if merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)], dtype=torch.float32))
        else:
            self.register_parameter("mix_factor", nn.Parameter(torch.tensor([float(alpha)], dtype=torch.float32)))

    def alpha_value(self):
        """Host float (one D2H read at pack time, never per step)."""
        m = float(self.mix_factor.detach().float().cpu().item())
        return m if self.merge_strategy == "fixed" else 1.0 / (1.0 + math.exp(-m))


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """util.py:141-165 on the device: (N,) f32 -> (N, dim) bf16 [cos | sin]."""
    if repeat_only:
        raise NotImplementedError
    return ops.timestep_embedding(timesteps.float(), dim, float(max_period))


def mlp_f32(x_bf16, pw0, pw2):
    """Linear-SiLU-Linear with f32 output (embedding MLPs; video_model.py:148-157,176-182; video_attention.py:227-231)."""
    h = ops.linear(x_bf16, pw0, out_f32=True)
    return ops.linear(ops.silu_to_bf16(h), pw2, out_f32=True)
