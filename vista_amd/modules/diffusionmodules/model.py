"""The image-VAE decoder building blocks behind `first_stage_model.decoder` (reference: vwm/modules/diffusionmodules/model.py).

Same class names, constructor arguments and state-dict keys as the reference; activations are token-major bf16
(n_img, H*W, C) and every op is a HIP launch through vista_amd.ops -- there is no eager / CPU path.
"""
from typing import Callable

import torch
import torch.nn as nn

from ... import ops
from ..attention import Packable
from .util import ConvNd, NormParams

CIN_PAD = 64  # the 4 latent channels are zero-padded to one 64-wide K block of the implicit-GEMM conv


def Normalize(in_channels, num_groups=32):
    """model.py:51-52: GroupNorm(32, eps=1e-6, affine)"""
    return NormParams(in_channels, 1e-6, num_groups)


class Upsample(nn.Module, Packable):
    """model.py:55-66: nearest x2 then conv3x3 -- one implicit-GEMM launch reading the source at (y>>1, x>>1)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("Vista's decoder uses resamp_with_conv=True")
        self.with_conv = with_conv
        self.conv = ConvNd(in_channels, in_channels, (3, 3), stride=1, padding=1)

    def _pack(self, dev):
        return ops.pack_conv3x3(self.conv.weight, self.conv.bias, device=dev)

    def forward(self, x, H, W):
        return ops.conv3x3(x, self.packed(), x.shape[0], H, W, ups=2)


class Downsample(nn.Module, Packable):
    """model.py:69-84: F.pad(x, (0,1,0,1)) then conv3x3 stride 2 pad 0 -- one implicit-GEMM launch with bottom/right padding."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("Vista's encoder uses resamp_with_conv=True")
        self.with_conv = with_conv
        self.conv = ConvNd(in_channels, in_channels, (3, 3), stride=2, padding=0)

    def _pack(self, dev):
        return ops.pack_conv3x3(self.conv.weight, self.conv.bias, device=dev)

    def forward(self, x, H, W):
        return ops.conv3x3(x, self.packed(), x.shape[0], H, W, stride=2, asym_pad=True)


class ResnetBlock(nn.Module, Packable):
    """model.py:87-135 with temb_channels=0 (the decoder passes temb=None)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        if conv_shortcut or temb_channels > 0:
            raise NotImplementedError("decoder ResnetBlock: nin_shortcut and no timestep embedding (model.py:610-624)")
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = ConvNd(in_channels, out_channels, (3, 3), padding=1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = ConvNd(out_channels, out_channels, (3, 3), padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = ConvNd(in_channels, out_channels, (1, 1))

    def _pack(self, dev):
        pk = {"conv1": ops.pack_conv3x3(self.conv1.weight, self.conv1.bias, device=dev),
              "conv2": ops.pack_conv3x3(self.conv2.weight, self.conv2.bias, device=dev)}
        if self.in_channels != self.out_channels:
            pk["nin"] = ops.pack_linear(self.nin_shortcut.weight, self.nin_shortcut.bias, dev)
        return pk

    def forward(self, x, temb, H, W):
        assert temb is None
        pk = self.packed()
        n_img = x.shape[0]
        h = ops.groupnorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, silu=True)  # swish == SiLU (model.py:46-48)
        h, _, _ = ops.conv3x3(h, pk["conv1"], n_img, H, W)
        h = ops.groupnorm(h, self.norm2.weight, self.norm2.bias, self.norm2.eps, silu=True)
        skip = ops.linear(x, pk["nin"]) if "nin" in pk else x
        out, _, _ = ops.conv3x3(h, pk["conv2"], n_img, H, W, res1=skip)
        return out


class AttnBlock(nn.Module, Packable):
    """model.py:147-176: single-head self-attention over the H*W positions of each frame with head dim = channels (512).
    Per frame: scores = q.k^T/sqrt(C) (GEMM, fp32 out) -> row softmax (bf16) -> P.v (GEMM against v^T, which the v
    projection writes transposed) ; proj_out adds the residual in its epilogue."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = ConvNd(in_channels, in_channels, (1, 1))
        self.k = ConvNd(in_channels, in_channels, (1, 1))
        self.v = ConvNd(in_channels, in_channels, (1, 1))
        self.proj_out = ConvNd(in_channels, in_channels, (1, 1))

    def _pack(self, dev):
        return {n: ops.pack_linear(getattr(self, n).weight, getattr(self, n).bias, dev) for n in ("q", "k", "v", "proj_out")}

    def forward(self, x, H, W, **kwargs):
        pk = self.packed()
        n_img, S, C = x.shape
        if S % 64 or C % 64:
            raise ValueError(f"AttnBlock: H*W ({S}) and channels ({C}) must be multiples of 64")
        h = ops.groupnorm(x, self.norm.weight, self.norm.bias, self.norm.eps, silu=False)
        slack = 320  # weight-operand rows are read up to the next block-tile boundary
        q = ops.linear(h, pk["q"]).view(n_img, S, C)
        kbuf = torch.empty((n_img * S + slack, C), dtype=torch.bfloat16, device=x.device)
        k = ops.linear(h, pk["k"], out=kbuf[:n_img * S])
        vbuf = torch.empty((n_img * C + slack, S), dtype=torch.bfloat16, device=x.device)
        ops.linear_vt(h, pk["v"], S, out=vbuf[:n_img * C].view(n_img, C, S))
        o = torch.empty((n_img, S, C), dtype=torch.bfloat16, device=x.device)
        scores = torch.empty((S, S), dtype=torch.float32, device=x.device)
        prob = torch.empty((S, S), dtype=torch.bfloat16, device=x.device)
        scale = float(C) ** -0.5  # F.scaled_dot_product_attention default (model.py:167)
        for f in range(n_img):
            ops.linear(q[f], ops.pack_rows_as_weight(kbuf[f * S:], S, C), out=scores, alpha=scale)
            ops.softmax_rows(scores, out=prob)
            ops.linear(prob, ops.pack_rows_as_weight(vbuf[f * C:], C, S), out=o[f])
        return ops.linear(o, pk["proj_out"], res1=x).view(n_img, S, C)


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None):
    """model.py:244-271: vista.yaml selects `vanilla` for the first stage and `vanilla-xformers` for the conditioner's encoder copy
    (MemoryEfficientAttnBlock, model.py:179-232: the same parameters -- norm, q, k, v, proj_out -- and the same single-head
    softmax(q k^T / sqrt(C)) v, only computed by xformers there): one class serves both."""
    if attn_type not in ("vanilla", "vanilla-xformers"):
        raise NotImplementedError(f"attn_type {attn_type!r}: Vista uses 'vanilla' / 'vanilla-xformers'")
    assert attn_kwargs is None
    return AttnBlock(in_channels)


class Decoder(nn.Module, Packable):
    """model.py:560-694. forward(z NCHW fp32 latents, **kwargs) -> NCHW fp32 images."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, use_linear_attn=False, attn_type="vanilla",
                 **ignorekwargs):
        super().__init__()
        if use_linear_attn or give_pre_end or tanh_out:
            raise NotImplementedError("linear attention / give_pre_end / tanh_out are not used by Vista's first stage")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels, self.out_ch, self.z_channels = resolution, in_channels, out_ch, z_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        make_attn_cls = self._make_attn()
        make_resblock_cls = self._make_resblock()
        make_conv_cls = self._make_conv()
        self.conv_in = ConvNd(z_channels, block_in, (3, 3), padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = make_resblock_cls(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn_cls(block_in, attn_type=attn_type)
        self.mid.block_2 = make_resblock_cls(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks + 1):
                block.append(make_resblock_cls(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn_cls(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = make_conv_cls(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def _make_attn(self) -> Callable:
        return make_attn

    def _make_resblock(self) -> Callable:
        return ResnetBlock

    def _make_conv(self) -> Callable:
        return _Conv2dOut

    def get_last_layer(self, **kwargs):
        return self.conv_out.weight

    def _pack_params(self):  # only what _pack reads (the children pack themselves)
        return [self.conv_in.weight, self.conv_in.bias]

    def _pack(self, dev):
        return {"conv_in": ops.pack_conv3x3(self.conv_in.weight, self.conv_in.bias, cin_pad=CIN_PAD, device=dev)}

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        for m in self.modules():
            if isinstance(m, Packable):
                m.invalidate_packed()
        return r

    @ops.bf16_storage   # (the first stage stores bf16 in every process: ops.storage)
    def forward(self, z, **kwargs):
        if z.device.type != "cuda":
            raise ops._lib.VistaHipError("Decoder: latents must be on the MI355X (vista_amd has no CPU path)")
        n_img, _, H, W = z.shape
        h = ops.nchw_to_tokens(z.float(), CIN_PAD)
        h, _, _ = ops.conv3x3(h, self.packed()["conv_in"], n_img, H, W)
        h = self.mid.block_1(h, None, H, W, **kwargs)
        h = self.mid.attn_1(h, H, W, **kwargs)
        h = self.mid.block_2(h, None, H, W, **kwargs)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block](h, None, H, W, **kwargs)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block](h, H, W, **kwargs)
            if i_level != 0:
                h, H, W = self.up[i_level].upsample(h, H, W)
        h = ops.groupnorm(h, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps, silu=True)
        return self.conv_out(h, H, W, **kwargs)


class Encoder(nn.Module, Packable):
    """model.py:445-558. forward(x NCHW fp32 images in [-1, 1]) -> NCHW fp32 moments (2*z_channels when double_z)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, double_z=True, use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn:
            raise NotImplementedError("linear attention is not used by Vista's first stage")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = ConvNd(in_channels, self.ch, (3, 3), padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = _Conv2dOut(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def _pack_params(self):  # only what _pack reads (the children pack themselves)
        return [self.conv_in.weight, self.conv_in.bias]

    def _pack(self, dev):
        return {"conv_in": ops.pack_conv3x3(self.conv_in.weight, self.conv_in.bias, cin_pad=CIN_PAD, device=dev)}

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        for m in self.modules():
            if isinstance(m, Packable):
                m.invalidate_packed()
        return r

    @ops.bf16_storage
    def forward(self, x):
        h, H, W = self.features(x)
        return self.conv_out(h, H, W)

    @ops.bf16_storage
    def features(self, x):
        """Everything up to (and including) norm_out + swish: ((n, H'*W', C) bf16 tokens, H', W'). `conv_out` follows; the mode-only
        autoencoder of the conditioner composes its 1x1 quant_conv into that convolution (models/autoencoder.AutoencoderKLModeOnly)."""
        if x.device.type != "cuda":
            raise ops._lib.VistaHipError("Encoder: images must be on the MI355X (vista_amd has no CPU path)")
        n_img, _, H, W = x.shape
        h = ops.nchw_to_tokens(x.float(), CIN_PAD)
        h, _, _ = ops.conv3x3(h, self.packed()["conv_in"], n_img, H, W)
        for i_level in range(self.num_resolutions):  # the reference's `hs` list only ever reads its last element
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block](h, None, H, W)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block](h, H, W)
            if i_level != self.num_resolutions - 1:
                h, H, W = self.down[i_level].downsample(h, H, W)
        h = self.mid.block_1(h, None, H, W)
        h = self.mid.attn_1(h, H, W)
        h = self.mid.block_2(h, None, H, W)
        h = ops.groupnorm(h, self.norm_out.weight, self.norm_out.bias, self.norm_out.eps, silu=True)
        return h, H, W


class _Conv2dOut(ConvNd, Packable):
    """nn.Conv2d conv_out of the plain image Decoder (model.py:652): tokens -> NCHW fp32."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1):
        super().__init__(in_channels, out_channels, (kernel_size, kernel_size), stride, padding)

    def _pack(self, dev):
        return ops.pack_conv3x3(self.weight, self.bias, device=dev)

    def forward(self, x, H, W, **kwargs):
        out, _, _ = ops.conv3x3(x, self.packed(), x.shape[0], H, W, out_f32=True)
        return ops.tokens_to_nchw(out, x.shape[0], self.out_channels, H, W)
