"""Temporal VAE decoder (reference: vwm/modules/autoencoding/temporal_ae.py): the image decoder's ResnetBlocks followed by
3-D `time_stack` ResBlocks over the frames of a clip, blended with a learned scalar, and a time-mixing output conv.

Same classes / constructor arguments / state-dict keys as the reference (`first_stage_model.decoder.*` loads unchanged).
"""
import math
from typing import Callable, Iterable, Union

import torch
import torch.nn as nn

from ... import ops
from ...util import partialclass
from ..attention import Packable
from ..diffusionmodules.model import Decoder, ResnetBlock
from ..diffusionmodules.openaimodel import ResBlock
from ..diffusionmodules.util import ConvNd


class VideoResBlock(ResnetBlock):
    """temporal_ae.py:11-72"""

    def __init__(self, out_channels, *args, dropout=0.0, video_kernel_size=3, alpha=0.0, merge_strategy="learned", **kwargs):
        super().__init__(out_channels=out_channels, dropout=dropout, *args, **kwargs)
        if video_kernel_size is None:
            video_kernel_size = [3, 1, 1]
        self.time_stack = ResBlock(channels=out_channels, emb_channels=0, dropout=dropout, dims=3, use_scale_shift_norm=False,
                                   use_conv=False, up=False, down=False, kernel_size=video_kernel_size, use_checkpoint=False,
                                   skip_t_emb=True)
        self.merge_strategy = merge_strategy
        if self.merge_strategy == "fixed":
            self.register_buffer("mix_factor", torch.tensor([float(alpha)]))
        elif self.merge_strategy == "learned":
            self.register_parameter("mix_factor", nn.Parameter(torch.tensor([float(alpha)])))
        else:
            raise ValueError(f"Unknown merge strategy {self.merge_strategy}")
        self._alpha = None

    def invalidate_packed(self):
        super().invalidate_packed()
        self._alpha = None

    def get_alpha(self):
        """temporal_ae.py:47-53 as a host float (one D2H read per weight load, none per decode); re-read when mix_factor changes."""
        mf = self.mix_factor
        key = (mf.device, mf._version)
        if self._alpha is None or self._alpha[0] != key:
            m = float(mf.detach().float().cpu().item())
            self._alpha = (key, m if self.merge_strategy == "fixed" else 1.0 / (1.0 + math.exp(-m)))
        return self._alpha[1]

    def forward(self, x, temb, H, W, skip_video=False, timesteps=None):
        if timesteps is None:
            timesteps = self.timesteps
        x = super().forward(x, temb, H, W)
        if skip_video:
            return x
        # alpha*(x + h_t) + (1-alpha)*x == x + alpha*h_t, folded into the last temporal conv's epilogue
        return self.time_stack(x, None, H, W, T=timesteps, out_alpha=self.get_alpha())


class AE3DConv(ConvNd, Packable):
    """temporal_ae.py:75-97: Conv2d(in -> out) then Conv3d(out -> out, video_kernel_size) across the clip.
    The 3 image channels travel between the two convs in a zero-padded 64-wide token buffer (one K block of the temporal conv)."""

    def __init__(self, in_channels, out_channels, video_kernel_size=3, kernel_size=3, stride=1, padding=1):
        if kernel_size != 3 or stride != 1 or padding != 1:
            raise NotImplementedError("AE3DConv: conv_out is 3x3, stride 1, pad 1 (model.py:652)")
        super().__init__(in_channels, out_channels, (3, 3), stride, padding)
        if isinstance(video_kernel_size, Iterable):
            vks = tuple(video_kernel_size)
            pad = [int(k // 2) for k in vks]
        else:
            vks = (video_kernel_size,) * 3
            pad = int(video_kernel_size // 2)
        if vks not in ((3, 1, 1), (3, 3, 3)):
            raise NotImplementedError("AE3DConv: video_kernel_size must be [3,1,1] or 3")
        self.full3d = vks == (3, 3, 3)
        self.time_mix_conv = ConvNd(out_channels, out_channels, vks, padding=pad)

    def _pack(self, dev):
        pt = ops.pack_conv3d if self.full3d else ops.pack_conv_t3
        return {"conv": ops.pack_conv3x3(self.weight, self.bias, device=dev),
                "mix": pt(self.time_mix_conv.weight, self.time_mix_conv.bias, device=dev, cin_pad=64)}

    def forward(self, x, H, W, timesteps, skip_video=False):
        pk = self.packed()
        n_img = x.shape[0]
        if skip_video:
            out, _, _ = ops.conv3x3(x, pk["conv"], n_img, H, W, out_f32=True)
            return ops.tokens_to_nchw(out, n_img, self.out_channels, H, W)
        mid = torch.zeros((n_img * H * W, 64), dtype=torch.bfloat16, device=x.device)
        ops.conv3x3(x, pk["conv"], n_img, H, W, out=mid[:, :pk["conv"].N])
        mid = mid.view(n_img, H * W, 64)
        if self.full3d:
            out = ops.conv3d(mid, pk["mix"], timesteps, H, W, out_f32=True)
        else:
            out = ops.conv_t3(mid, pk["mix"], timesteps, H * W, out_f32=True)
        return ops.tokens_to_nchw(out, n_img, self.out_channels, H, W)


class VideoDecoder(Decoder):
    """temporal_ae.py:105-152 (time_mode 'conv-only', the only mode Vista's config uses)."""
    available_time_modes = ["all", "conv-only", "attn-only"]

    def __init__(self, *args, video_kernel_size: Union[int, list] = 3, alpha: float = 0.0, merge_strategy: str = "learned",
                 time_mode: str = "conv-only", **kwargs):
        self.video_kernel_size = video_kernel_size
        self.alpha = alpha
        self.merge_strategy = merge_strategy
        self.time_mode = time_mode
        assert self.time_mode in self.available_time_modes, f"time_mode parameter has to be in {self.available_time_modes}"
        if time_mode != "conv-only":
            raise NotImplementedError("VideoDecoder: only time_mode='conv-only' (vista.yaml) is built")
        super().__init__(*args, **kwargs)

    def get_last_layer(self, skip_time_mix=False, **kwargs):
        return self.conv_out.time_mix_conv.weight if not skip_time_mix else self.conv_out.weight

    def _make_conv(self) -> Callable:
        return partialclass(AE3DConv, video_kernel_size=self.video_kernel_size)

    def _make_resblock(self) -> Callable:
        return partialclass(VideoResBlock, video_kernel_size=self.video_kernel_size, alpha=self.alpha, merge_strategy=self.merge_strategy)
