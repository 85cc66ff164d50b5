"""UNet primitives (reference: vwm/modules/diffusionmodules/openaimodel.py), MI355X-native.

  ResBlock._forward (openaimodel.py:258-284):
      h = conv(silu(GN32(x)))          vk_groupnorm_silu_bf16 -> implicit-GEMM 3x3 (or 3x1x1) conv
      h += Linear(silu(emb))[n]        per-image row vector added in that conv's epilogue
      h = conv(silu(GN32(h)))          second conv; `skip(x) + h` is its residual epilogue
  Upsample (openaimodel.py:86-103): nearest x2 fused into the conv's gather; Downsample (:136): stride-2 gather.
"""
from typing import Iterable

import torch.nn as nn

from ... import ops
from ..attention import FP8, Packable
from .util import Dropout, SiLU, conv_nd, linear, normalization, timestep_embedding, zero_module


class TimestepBlock(nn.Module):
    """Any module where forward() takes timestep embeddings as a second argument."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """openaimodel.py:27-53: passes (emb | context | num_frames) to the children that take them.
    Activations are token-major: x (n_img, S, C) bf16 with the spatial size (H, W) carried alongside."""

    def forward(self, x, emb_silu, context=None, frame_idx=None, num_frames=None, H=None, W=None, shard=None, full=None):
        """shard / full: multi-GPU frame sharding (vista_amd.parallel.FrameShard) and the replicated full-window
        tensors {"emb_silu": (B*T, E), "ctx": (B*T, ctx)} the pixel-sharded temporal halves need."""
        from ..video_attention import SpatialVideoTransformer
        from .video_model import VideoResBlock
        layers = list(self)
        x_gn = None  # GroupNorm statistics of x, emitted by the convolution that produced it (ops.GnPartials), for the layer that opens with a norm of x
        for i, layer in enumerate(layers):
            if isinstance(layer, VideoResBlock):
                # the block's last convolution (temporal conv2 + blend) emits the statistics of the transformer's opening GroupNorm
                x_gn = ops.GnPartials() if i + 1 < len(layers) and isinstance(layers[i + 1], SpatialVideoTransformer) else None
                x = layer(x, emb_silu, num_frames, H, W, shard=shard, full=full, out_gn=x_gn)
            elif isinstance(layer, SpatialVideoTransformer):
                x = layer(x, context, frame_idx, num_frames, H, W, shard=shard, full=full, x_gn=x_gn)
                x_gn = None
            elif isinstance(layer, (Upsample, Downsample)):
                x, H, W = layer(x, H, W)
            else:
                raise TypeError(f"unexpected layer {type(layer).__name__} in TimestepEmbedSequential")
        return x, H, W


class Upsample(nn.Module, Packable):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_up=False, kernel_size=3, scale_factor=2):
        super().__init__()
        if dims != 2 or not use_conv or kernel_size != 3 or scale_factor != 2 or padding != 1:
            raise NotImplementedError("Vista uses Upsample(dims=2, use_conv=True, k=3, x2)")
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv, self.dims = use_conv, dims
        self.conv = conv_nd(dims, self.channels, self.out_channels, kernel_size, padding=padding)

    def _pack(self, dev):
        return ops.pack_conv3x3(self.conv.weight, self.conv.bias, device=dev)

    def forward(self, x, H, W):
        assert x.shape[-1] == self.channels
        out, Ho, Wo = ops.conv3x3(x, self.packed(), x.shape[0], H, W, ups=2)
        return out, Ho, Wo


class Downsample(nn.Module, Packable):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1, third_down=False):
        super().__init__()
        if dims != 2 or not use_conv or padding != 1:
            raise NotImplementedError("Vista uses Downsample(dims=2, use_conv=True)")
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv, self.dims = use_conv, dims
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def _pack(self, dev):
        return ops.pack_conv3x3(self.op.weight, self.op.bias, device=dev)

    def forward(self, x, H, W):
        assert x.shape[-1] == self.channels
        out, Ho, Wo = ops.conv3x3(x, self.packed(), x.shape[0], H, W, stride=2)
        return out, Ho, Wo


class ResBlock(TimestepBlock, Packable):
    """openaimodel.py:146-284. dims=2: 3x3 convs; dims=3 with kernel (3,1,1): the temporal `time_stack`."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, up=False, down=False, kernel_size=3, exchange_temb_dims=False, skip_t_emb=False, causal=False):
        super().__init__()
        if up or down or use_scale_shift_norm or use_conv or causal:
            raise NotImplementedError("resblock_updown / scale-shift norm / causal are not used by Vista")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_checkpoint = use_checkpoint
        self.exchange_temb_dims = exchange_temb_dims and not skip_t_emb
        self.skip_t_emb = skip_t_emb
        self.dims = dims
        if isinstance(kernel_size, Iterable):
            kernel_size = tuple(kernel_size)
            padding = [k // 2 for k in kernel_size]
        else:
            padding = kernel_size // 2
        if dims == 2 and kernel_size not in (3, (3, 3)):
            raise NotImplementedError("2-D ResBlock kernel must be 3x3")
        if dims == 3 and kernel_size not in (3, (3, 3, 3), (3, 1, 1)):
            raise NotImplementedError("3-D ResBlock kernel must be (3,1,1) (video_kernel_size: [3,1,1]) or 3x3x3")
        self.full3d = dims == 3 and kernel_size != (3, 1, 1)
        self.in_layers = nn.Sequential(normalization(channels), SiLU(), conv_nd(dims, channels, self.out_channels, kernel_size, padding=padding))
        self.updown = False
        # skip_t_emb (openaimodel.py:216-220): the VAE decoder's time_stack has no embedding branch
        self.emb_layers = None if skip_t_emb else nn.Sequential(SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), SiLU(), Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, kernel_size, padding=padding)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def _pack(self, dev):
        pc = ops.pack_conv3x3 if self.dims == 2 else (ops.pack_conv3d if self.full3d else ops.pack_conv_t3)
        pk = {"conv1": pc(self.in_layers[2].weight, self.in_layers[2].bias, device=dev),
              "conv2": pc(self.out_layers[3].weight, self.out_layers[3].bias, device=dev)}
        if self.emb_layers is not None:
            pk["emb"] = ops.pack_linear(self.emb_layers[1].weight, self.emb_layers[1].bias, dev)
        if not isinstance(self.skip_connection, nn.Identity):
            pk["skip"] = ops.pack_linear(self.skip_connection.weight, self.skip_connection.bias, dev)
        if FP8["conv"] and not self.full3d and self.emb_layers is not None:  # the UNet's ResBlocks only (the VAE decoder's time_stack has no embedding)
            pc8 = ops.pack_conv3x3_fp8 if self.dims == 2 else ops.pack_conv_t3_fp8
            pk["conv1_8"] = pc8(self.in_layers[2].weight, self.in_layers[2].bias, device=dev)
            pk["conv2_8"] = pc8(self.out_layers[3].weight, self.out_layers[3].bias, device=dev)
        return pk

    def forward(self, x, emb_silu, H, W, T=None, out_alpha=1.0, shard=None, T_global=None, x_gn=None, out_gn=None):
        """x (n_img, S, C) bf16, or a pair (h, skip) standing for their channel concat (the `torch.cat([h, hs.pop()], dim=1)` of
        the UNet's output blocks, video_model.py:493, read in place by the norm and the 1x1 skip conv -- never materialised);
        emb_silu (n_img, emb_channels) bf16 = silu(emb).
        dims=2: returns skip(x) + h.  dims=3 (time_stack): statistics/conv span the T frames of each clip and the result is
        blend + out_alpha*(conv2 + bias) with blend = x, i.e. AlphaBlender(x_spatial=x, x_temporal=x+h) folded in.
        Multi-GPU (`shard`): x holds T = t_local frames of a T_global-frame clip; the norms all-reduce their partial sums and
        the convs read the neighbour ranks' boundary frames (halo exchange).
        x_gn / out_gn (ops.GnPartials or None): the GroupNorm statistics of x as its producer's epilogue emitted them (the first norm then runs
        no statistics pass), and a holder the LAST convolution of this block fills with the statistics of the result for whoever norms it next;
        the second norm always takes its statistics from the first convolution's epilogue when that launch can emit them (ops.GN_EPI)."""
        pk = self.packed()
        xb = None
        if isinstance(x, tuple):
            x, xb = x
            if self.dims != 2 or "skip" not in pk:
                raise ValueError("a concatenated input needs the 2-D ResBlock with a 1x1 skip convolution")
        n_img, S, _ = x.shape
        gn1, gn2 = self.in_layers[0], self.out_layers[0]
        fpg = 1 if self.dims == 2 else T
        # (n_img, Cout): `emb_layers(emb)[..., None, None]`; skip_t_emb adds zeros (openaimodel.py:268-269)
        src = getattr(self, "_emb_src", None)  # (thread-local table, column offset, width) when a VideoUNet owns this block
        table = getattr(src[0], "table", None) if src is not None else None
        table = table.get("emb") if table is not None else None
        if self.emb_layers is None:
            emb_out = None
        elif table is not None and table.shape[0] == n_img:
            emb_out = table[:, src[1]:src[1] + src[2]]  # the UNet already projected silu(emb) for every block in one GEMM
        else:
            emb_out = ops.linear(emb_silu, pk["emb"], out_f32=True)

        def gnorm(t, gn, part=None):
            if shard is None or self.dims == 2:
                return ops.groupnorm(t, gn.weight, gn.bias, gn.eps, silu=True, frames_per_group=fpg, gn=part)
            # frame-sharded temporal norm: statistics span (C/32, ALL T frames, H*W) -> all-reduce the local partial sums
            cnt = float(t.shape[-1] // 32) * float(S) * float(T_global)
            return ops.groupnorm_sharded(t, gn.weight, gn.bias, gn.eps, True, fpg, shard.all_reduce_sum, cnt, gn=part)

        def halo(t):
            return shard.halo_exchange(t) if (shard is not None and self.dims == 3) else (None, None)
        if FP8["conv"] and shard is None and not self.full3d and self.emb_layers is not None:
            # BASELINE config 5: both convolutions in fp8 e4m3. The GroupNorm+SiLU pass writes e4m3 with one scale per image (per clip for the
            # temporal norm) -- half the bytes it wrote before -- and the implicit-GEMM loaders stream those bytes; no quantisation pass.
            if "conv1_8" not in pk:  # the switch was flipped after the bf16 pack was built
                self.invalidate_packed()
                pk = self.packed()
            h8, hs = ops.groupnorm_fp8(x, gn1.weight, gn1.bias, gn1.eps, True, fpg, x2=xb)
            if self.dims == 2:
                h = ops.conv3x3_fp8(h8, hs, pk["conv1_8"], n_img, H, W, rowvec=emb_out)
                h8, hs = ops.groupnorm_fp8(h, gn2.weight, gn2.bias, gn2.eps, True)
                skip = x if "skip" not in pk else ops.linear(x, pk["skip"], x2=xb)
                return ops.conv3x3_fp8(h8, hs, pk["conv2_8"], n_img, H, W, res1=skip)
            h = ops.conv_t3_fp8(h8, hs, pk["conv1_8"], T, S, rowvec=emb_out)
            h8, hs = ops.groupnorm_fp8(h, gn2.weight, gn2.bias, gn2.eps, True, fpg)
            return ops.conv_t3_fp8(h8, hs, pk["conv2_8"], T, S, alpha=out_alpha, res2=x, beta=1.0)
        h = gnorm(x, gn1, x_gn) if xb is None else ops.groupnorm_cat(x, xb, gn1.weight, gn1.bias, gn1.eps, silu=True)
        mid_gn = ops.GnPartials()  # statistics of the first convolution's output, from its epilogue (stays empty when that launch cannot emit them)
        if self.dims == 2:
            h, _, _ = ops.conv3x3(h, pk["conv1"], n_img, H, W, rowvec=emb_out, gn=mid_gn)
            h = ops.groupnorm(h, gn2.weight, gn2.bias, gn2.eps, silu=True, gn=mid_gn)
            skip = x if "skip" not in pk else ops.linear(x, pk["skip"], x2=xb)
            out, _, _ = ops.conv3x3(h, pk["conv2"], n_img, H, W, res1=skip, gn=out_gn)
            return out
        if self.full3d:  # 3x3x3 time_stack (VAE decoder with video_kernel_size=3); single-GPU only
            if shard is not None or emb_out is not None:
                raise NotImplementedError("3x3x3 time_stack: no frame sharding / embedding branch")
            h = ops.conv3d(h, pk["conv1"], T, H, W)
            h = gnorm(h, gn2)
            return ops.conv3d(h, pk["conv2"], T, H, W, alpha=out_alpha, res2=x, beta=1.0)
        prev, nxt = halo(h)
        h = ops.conv_t3(h, pk["conv1"], T, S, rowvec=emb_out, halo_prev=prev, halo_next=nxt, gn=mid_gn)
        h = gnorm(h, gn2, mid_gn)
        prev, nxt = halo(h)
        return ops.conv_t3(h, pk["conv2"], T, S, alpha=out_alpha, res2=x, beta=1.0, halo_prev=prev, halo_next=nxt, gn=out_gn)


class Timestep(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        return timestep_embedding(t, self.dim)
