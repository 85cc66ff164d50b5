"""Classifier-free-guidance batching and combination (reference: vwm/modules/diffusionmodules/guiders.py).
`prepare_inputs` is pure tensor plumbing (cat); the combine runs as one HIP kernel (vk_cfg_combine)."""
from abc import ABC, abstractmethod
from typing import List, Literal, Optional, Union

import torch

from ... import ops
from ...util import default


class Guider(ABC):
    @abstractmethod
    def __call__(self, x: torch.Tensor, sigma: float) -> torch.Tensor:
        pass

    def prepare_inputs(self, x, s, c, cond_mask, uc):
        pass

    def frame_scales(self, num_frames):
        """Per-frame guidance scale (T,) f32 on the host; None = no CFG doubling (IdentityGuider)."""
        return None


def _cfg_prepare(x, s, c, cond_mask, uc, keys):
    """The CFG-doubled batch: [unconditional | conditional] along dim 0 for x, sigma, the mask and every conditioning entry in `keys`;
    entries outside `keys` must be the same object in both dicts and pass through (guiders.py:30-38,76-84)."""
    doubled = {}
    for name, value in c.items():
        if name in keys:
            doubled[name] = torch.cat((uc[name], value), 0)
        else:
            assert value == uc[name]
            doubled[name] = value
    twice = lambda t: torch.cat((t, t))  # noqa: E731
    return twice(x), twice(s), doubled, twice(cond_mask)


class VanillaCFG(Guider):
    def __init__(self, scale: float):
        self.scale = scale

    def frame_scales(self, num_frames):
        return torch.full((num_frames,), float(self.scale))

    def __call__(self, x, sigma):
        n = x.shape[0] // 2
        return ops.cfg_combine(x.float(), torch.full((n,), float(self.scale), device=x.device))

    def prepare_inputs(self, x, s, c, cond_mask, uc):
        return _cfg_prepare(x, s, c, cond_mask, uc, ["vector", "crossattn", "concat"])


class IdentityGuider(Guider):
    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, cond_mask, uc):
        return x, s, {k: c[k] for k in c}, cond_mask


class LinearPredictionGuider(Guider):
    def __init__(self, num_frames: int = 25, max_scale: float = 2.5, min_scale: float = 1.0,
                 additional_cond_keys: Optional[Union[List[str], str]] = None):
        self.min_scale, self.max_scale, self.num_frames = min_scale, max_scale, num_frames
        self.scale = torch.linspace(min_scale, max_scale, num_frames).unsqueeze(0)
        additional_cond_keys = default(additional_cond_keys, list())
        if isinstance(additional_cond_keys, str):
            additional_cond_keys = [additional_cond_keys]
        self.additional_cond_keys = additional_cond_keys

    def frame_scales(self, num_frames):
        assert num_frames == self.num_frames
        return self.scale[0].clone()

    def __call__(self, x, sigma):
        n = x.shape[0] // 2
        b = n // self.num_frames
        return ops.cfg_combine(x.float(), self.scale[0].repeat(b).to(x.device))

    def prepare_inputs(self, x, s, c, cond_mask, uc):
        return _cfg_prepare(x, s, c, cond_mask, uc, ["vector", "crossattn", "concat"] + self.additional_cond_keys)


class TrianglePredictionGuider(LinearPredictionGuider):
    def __init__(self, num_frames: int = 25, max_scale: float = 2.5, min_scale: float = 1.0, period: float = 1.0,
                 period_fusing: Literal["mean", "multiply", "max"] = "max",
                 additional_cond_keys: Optional[Union[List[str], str]] = None):
        super().__init__(num_frames, max_scale, min_scale, additional_cond_keys)
        ramp = torch.linspace(0, 1, num_frames)
        waves = torch.stack([self.triangle_wave(ramp, p) for p in ([period] if isinstance(period, float) else period)])
        if period_fusing == "mean":
            fused = sum(waves) / waves.shape[0]
        elif period_fusing == "multiply":
            fused = torch.prod(waves, dim=0)
        elif period_fusing == "max":
            fused = torch.max(waves, dim=0).values
        else:
            raise NotImplementedError
        self.scale = (fused * (max_scale - min_scale) + min_scale).unsqueeze(0)

    def triangle_wave(self, values, period):
        return 2 * (values / period - torch.floor(values / period + 0.5)).abs()
