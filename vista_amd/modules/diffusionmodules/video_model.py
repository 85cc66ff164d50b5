"""VideoUNet (reference: vwm/modules/diffusionmodules/video_model.py), MI355X-native.

Constructor arguments, forward signature and state-dict names are the reference's (video_model.py:79-114,442-451), so
the class is selected by changing only `network_config.target` in configs/inference/vista.yaml and
`ckpts/vista.safetensors` loads unchanged. Internally activations are token-major bf16 ((b t), H*W, C); the NCHW fp32
boundary tensors are converted by HIP layout kernels at entry and exit.
"""
from typing import List, Optional, Union

import threading

import torch
import torch.nn as nn

from ... import ops
from ...util import default, repeat_as_img_seq
from ..attention import Packable, _invalidate_after_load as _drop_packs_after_load
from ..video_attention import SpatialVideoTransformer
from .openaimodel import Downsample, ResBlock, Timestep, TimestepEmbedSequential, Upsample  # noqa: F401
from .util import AlphaBlender, SiLU, conv_nd, linear, mlp_f32, normalization, timestep_embedding, zero_module

CIN_PAD = 64  # the 8 input channels are zero-padded to one 64-wide K-step of the implicit GEMM


class _ThreadTable:
    """Per-thread slot for the batched projections of the forward in flight (a dict: "emb" = every ResBlock's emb_layers output,
    "ctx_s" / "ctx_t" = every spatial / temporal cross-attention's context vector (the multi-rank tests drive one model from
    several threads). Unlike threading.local it survives copy.deepcopy / pickling of the module that owns it (a fresh, empty table)."""

    def __init__(self):
        self._slots = {}

    @property
    def table(self):
        return self._slots.get(threading.get_ident())

    @table.setter
    def table(self, value):
        if value is None:
            self._slots.pop(threading.get_ident(), None)
        else:
            self._slots[threading.get_ident()] = value

    def __deepcopy__(self, memo):
        return _ThreadTable()

    def __reduce__(self):
        return (_ThreadTable, ())


class VideoResBlock(ResBlock):
    """video_model.py:9-75: 2-D ResBlock, then the (3,1,1) temporal ResBlock over `b c t h w`, blended by AlphaBlender."""

    def __init__(self, channels, emb_channels, dropout, video_kernel_size=3, merge_strategy="fixed", merge_factor=0.5,
                 out_channels=None, use_conv=False, use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__(channels, emb_channels, dropout, out_channels=out_channels, use_conv=use_conv,
                         use_scale_shift_norm=use_scale_shift_norm, dims=dims, use_checkpoint=use_checkpoint, up=up, down=down)
        oc = default(out_channels, channels)
        self.time_stack = ResBlock(oc, emb_channels, dropout=dropout, dims=3, out_channels=oc, use_scale_shift_norm=False,
                                   use_conv=False, up=False, down=False, kernel_size=video_kernel_size, use_checkpoint=use_checkpoint,
                                   exchange_temb_dims=True, causal=False)
        self.time_mixer = AlphaBlender(alpha=merge_factor, merge_strategy=merge_strategy, rearrange_pattern="b t -> b 1 t 1 1")

    def _pack_params(self):  # the spatial ResBlock's own tensors + the blend factor; time_stack packs itself
        return [p for n, p in self.named_parameters() if not n.startswith("time_stack.")]

    def _pack(self, dev):
        pk = super()._pack(dev)
        pk["alpha"] = self.time_mixer.alpha_value()
        return pk

    def forward(self, x, emb_silu, num_frames, H, W, shard=None, full=None, out_gn=None):
        """out_gn (ops.GnPartials or None): filled by the last temporal convolution with the GroupNorm statistics of the result (the opening
        norm of the SpatialVideoTransformer that follows). The temporal ResBlock's own first norm takes its statistics from the epilogue of the
        spatial ResBlock's last convolution in the same way."""
        mid = ops.GnPartials()
        x = super().forward(x, emb_silu, H, W, out_gn=mid)
        alpha = self.packed()["alpha"]
        # alpha*x + (1-alpha)*(x + h_t) == x + (1-alpha)*h_t, fused into the last temporal conv's epilogue
        if shard is None:
            return self.time_stack(x, emb_silu, H, W, T=num_frames, out_alpha=1.0 - alpha, x_gn=mid, out_gn=out_gn)
        # frame-sharded: the temporal ResBlock keeps this rank's frames (halo exchange + stats all-reduce inside)
        return self.time_stack(x, emb_silu, H, W, T=shard.t_local, out_alpha=1.0 - alpha, shard=shard, T_global=num_frames, x_gn=mid, out_gn=out_gn)


class VideoUNet(nn.Module, Packable):
    def __init__(
            self,
            in_channels: int,
            model_channels: int,
            out_channels: int,
            num_res_blocks: int,
            attention_resolutions: int,
            dropout: float = 0.0,
            channel_mult: List[int] = (1, 2, 4, 8),
            conv_resample: bool = True,
            dims: int = 2,
            num_classes: Optional[int] = None,
            use_checkpoint: bool = False,
            num_heads: int = -1,
            num_head_channels: int = -1,
            num_heads_upsample: int = -1,
            use_scale_shift_norm: bool = False,
            resblock_updown: bool = False,
            transformer_depth: Union[List[int], int] = 1,
            transformer_depth_middle: Optional[int] = None,
            context_dim: Optional[int] = None,
            time_downup: bool = False,
            time_context_dim: Optional[int] = None,
            extra_ff_mix_layer: bool = False,
            use_spatial_context: bool = False,
            merge_strategy: str = "learned_with_images",
            merge_factor: float = 0.5,
            spatial_transformer_attn_type: str = "softmax",
            video_kernel_size: Union[int, List[int]] = 3,
            use_linear_in_transformer: bool = False,
            adm_in_channels: Optional[int] = None,
            disable_temporal_crossattention: bool = False,
            max_ddpm_temb_period: int = 10000,
            add_lora: bool = False,
            action_control: bool = False
    ):
        super().__init__()
        assert context_dim is not None
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1
        if num_head_channels == -1:
            assert num_heads != -1
        if dims != 2 or resblock_updown or time_downup or not conv_resample or use_scale_shift_norm:
            raise NotImplementedError("only the shipped Vista topology options are implemented (dims=2, conv resample)")
        if num_classes != "sequential":
            raise NotImplementedError('num_classes must be "sequential" (vista.yaml:22)')
        if in_channels > CIN_PAD or in_channels % 4:
            raise NotImplementedError("in_channels must be <= 64 and a multiple of 4")
        if model_channels % 64:
            raise NotImplementedError("model_channels must be a multiple of 64 (GEMM K-step)")

        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        transformer_depth_middle = default(transformer_depth_middle, transformer_depth[-1])
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.max_ddpm_temb_period = max_ddpm_temb_period

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), SiLU(), linear(time_embed_dim, time_embed_dim))
        self.cond_time_stack_embed = nn.Sequential(linear(model_channels, time_embed_dim), SiLU(), linear(time_embed_dim, time_embed_dim))
        assert adm_in_channels is not None
        self.label_emb = nn.Sequential(nn.Sequential(linear(adm_in_channels, time_embed_dim), SiLU(), linear(time_embed_dim, time_embed_dim)))

        self.input_blocks = nn.ModuleList([_InputConv(in_channels, model_channels)])
        self._feature_size = model_channels
        input_block_chans = [model_channels]
        ch = model_channels
        ds = 1

        def get_attention_layer(ch, num_heads, dim_head, depth=1):
            return SpatialVideoTransformer(
                ch, num_heads, dim_head, depth=depth, context_dim=context_dim, time_context_dim=time_context_dim, dropout=dropout,
                ff_in=extra_ff_mix_layer, use_spatial_context=use_spatial_context, merge_strategy=merge_strategy,
                merge_factor=merge_factor, use_checkpoint=use_checkpoint, use_linear=use_linear_in_transformer,
                attn_mode=spatial_transformer_attn_type, disable_self_attn=False,
                disable_temporal_crossattention=disable_temporal_crossattention, max_time_embed_period=max_ddpm_temb_period,
                add_lora=add_lora, action_control=action_control)

        def get_resblock(ch, out_ch):
            return VideoResBlock(merge_factor=merge_factor, merge_strategy=merge_strategy, video_kernel_size=video_kernel_size,
                                 channels=ch, emb_channels=time_embed_dim, dropout=dropout, out_channels=out_ch, dims=dims,
                                 use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm)

        def heads_for(ch):
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [get_resblock(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    nh, dh = heads_for(ch)
                    layers.append(get_attention_layer(ch, nh, dh, depth=transformer_depth[level]))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                ds *= 2
                out_ch = ch
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=out_ch, third_down=time_downup)))
                ch = out_ch
                input_block_chans.append(ch)
                self._feature_size += ch

        nh, dh = heads_for(ch)
        self.middle_block = TimestepEmbedSequential(get_resblock(ch, None), get_attention_layer(ch, nh, dh, depth=transformer_depth_middle),
                                                    get_resblock(ch, None))
        self._feature_size += ch

        self.output_blocks = nn.ModuleList(list())
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = input_block_chans.pop()
                layers = [get_resblock(ch + ich, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    nh, dh = heads_for(ch)
                    layers.append(get_attention_layer(ch, nh, dh, depth=transformer_depth[level]))
                if level and i == num_res_blocks:
                    out_ch = ch
                    ds //= 2
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=out_ch, third_up=time_downup))
                self.output_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch

        self.out = nn.Sequential(normalization(ch), SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        self._frame_idx_cache = {}
        # Every ResBlock (spatial and time_stack) projects the SAME silu(emb) through its own emb_layers Linear: 44 launches of a
        # 7..50-row GEMM per forward. forward_tokens runs them as ONE GEMM over the row-concatenated weights and hands each block
        # its column slice through this thread-local table (thread-local: the multi-rank tests drive one model from several threads).
        self._emb_tls = _ThreadTable()
        off = 0
        self._emb_blocks = []
        for m in self.modules():
            if isinstance(m, ResBlock) and m.emb_layers is not None:
                object.__setattr__(m, "_emb_src", (self._emb_tls, off, m.out_channels))
                self._emb_blocks.append(m)
                off += m.out_channels
        # Likewise the 32 one-token cross-attentions (attn2 of every spatial / temporal transformer block): each is ONE affine map of the
        # context token (MemoryEfficientCrossAttention.context_map), so all of a kind run as one GEMM over the row-concatenated maps
        # (64 launches of 1..50-row GEMMs with K = 3456 per forward -> 2; they were 1.5 % of a single-GPU step and 7 % of an 8-GPU rank's).
        from ..attention import BasicTransformerBlock
        from ..video_attention import VideoTransformerBlock
        # belt and braces next to the packs' own (identity, version) keys: a load through THIS module drops every pack under it
        self.register_load_state_dict_post_hook(_drop_packs_after_load)
        self._ctx_layers = {"ctx_s": [], "ctx_t": []}
        offs = {"ctx_s": 0, "ctx_t": 0}
        for m in self.modules():
            key = "ctx_s" if isinstance(m, BasicTransformerBlock) else ("ctx_t" if isinstance(m, VideoTransformerBlock) and m.has_cross else None)
            if key is not None:
                object.__setattr__(m.attn2, "_ctx_src", (self._emb_tls, key, offs[key]))
                self._ctx_layers[key].append(m.attn2)
                offs[key] += m.attn2.query_dim

    # ---- weights ----
    # No load_state_dict override: every Packable keys its packed weights on its parameters' version counters, so ANY load
    # (this module's, a parent container's, an in-place EMA swap) is picked up at the next forward.
    def invalidate_all_packed(self):
        for m in self.modules():
            if isinstance(m, Packable):
                m.invalidate_packed()

    def _pack_params(self):  # only what _pack reads (the children pack themselves)
        ps = [p for seq in (self.time_embed, self.cond_time_stack_embed, self.label_emb, self.out) for p in seq.parameters()]
        for m in self._emb_blocks:
            ps += list(m.emb_layers.parameters())
        for layers in self._ctx_layers.values():
            for a in layers:
                ps += [a.to_v.weight, a.to_out[0].weight, a.to_out[0].bias] + ([a.v_adapter_action_control.weight] if a.action_control else [])
        return ps

    def _pack(self, dev):
        def mlp(seq):
            return (ops.pack_linear(seq[0].weight, seq[0].bias, dev), ops.pack_linear(seq[2].weight, seq[2].bias, dev))
        emb_w = torch.cat([m.emb_layers[1].weight.detach().float() for m in self._emb_blocks], 0)
        emb_b = torch.cat([m.emb_layers[1].bias.detach().float() for m in self._emb_blocks], 0)
        ctx = {}
        for key, layers in self._ctx_layers.items():
            if not layers:  # disable_temporal_crossattention=True: no temporal cross-attention layers to batch
                continue
            maps = [a.context_map() for a in layers]
            ctx[key] = ops.pack_linear(torch.cat([w for w, _ in maps], 0), torch.cat([b for _, b in maps], 0), dev)
        return {"emb_cat": ops.pack_linear(emb_w, emb_b, dev), **ctx,
                "time_embed": mlp(self.time_embed), "cond": mlp(self.cond_time_stack_embed), "label": mlp(self.label_emb[0]),
                "out": ops.pack_conv3x3(self.out[2].weight, self.out[2].bias, device=dev)}

    def _frame_idx(self, n_img, T, device):
        key = (n_img, T, str(device))
        if key not in self._frame_idx_cache:
            self._frame_idx_cache[key] = torch.arange(T, dtype=torch.float32).repeat(n_img // T).to(device)
        return self._frame_idx_cache[key]

    # ---- forward ----
    @torch.no_grad()
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, context: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None,
                time_context: Optional[torch.Tensor] = None, cond_mask: Optional[torch.Tensor] = None, num_frames: Optional[int] = None):
        """x (N, in_channels, H, W) NCHW, N = b*T with the frame index fastest; timesteps (N,); context (N or N/T, 1, ctx);
        y (N or N/T, adm); cond_mask (N,) 0/1; returns (N, out_channels, H, W) in x.dtype."""
        assert (y is not None) == (self.num_classes is not None), "Must specify y if and only if the model is class-conditional"
        if time_context is not None:
            raise NotImplementedError("separate time_context is not used with use_spatial_context=True")
        n_img, _, H, W = x.shape
        T = int(num_frames)
        assert n_img % T == 0
        tokens = ops.nchw_to_tokens(x.float(), CIN_PAD)
        out_tok = self.forward_tokens(tokens, timesteps, context, y, cond_mask, T, H, W)
        return ops.tokens_to_nchw(out_tok, n_img, self.out_channels, H, W).to(x.dtype)

    @torch.no_grad()
    def forward_tokens(self, tokens, timesteps, context, y, cond_mask, T, H, W, shard=None):
        """Token-major entry used by the fused sampler path: tokens (N, H*W, 64) bf16 (channels >= in_channels zero);
        returns (N, H*W, out_channels) f32. With `shard` (vista_amd.parallel.FrameShard) `tokens` holds only this rank's
        frames ((b, t_local) order) while timesteps / context / y / cond_mask are the replicated full-window tensors."""
        pk = self.packed()
        n_img = timesteps.shape[0]  # images of the full window (B*T)
        dev = tokens.device
        t_emb = timestep_embedding(timesteps, self.model_channels, max_period=10000)
        te = mlp_f32(t_emb, *pk["time_embed"])
        if context.shape[0] != n_img:
            assert context.shape[0] == n_img // T, f"{context.shape} {n_img}"
            context = repeat_as_img_seq(context, T)
        if y.shape[0] != n_img:
            assert y.shape[0] == n_img // T, f"{y.shape} {n_img}"
            y = repeat_as_img_seq(y, T)
        le = mlp_f32(ops.cast_to_bf16(y.float()), *pk["label"])
        if cond_mask is not None:
            # masked frames take the cond-frame time embedding (video_model.py:457-461). The reference branches on
            # cond_mask.any() (a host sync); blending with an all-zero mask is the identical arithmetic (a*0 + b*1).
            ce = mlp_f32(t_emb, *pk["cond"])
            emb, emb_silu = ops.emb_combine(ce, te, le, cond_mask.float().contiguous())
        else:
            emb, emb_silu = ops.emb_combine(None, te, le, None)
        if context.dim() != 3 or context.shape[1] != 1:
            raise NotImplementedError("Vista's cross-attention context is one token per image (crossattn: (N, 1, 3456))")
        ctx = ops.cast_to_bf16(context.float().reshape(n_img, -1))
        frame_idx = self._frame_idx(n_img, T, dev)
        full = None
        if shard is not None:  # spatial halves see this rank's rows; temporal halves get the replicated full tensors
            full = {"emb_silu": emb_silu, "ctx": ctx}
            emb_silu = shard.take_local_rows(emb_silu)
            ctx = shard.take_local_rows(ctx)
        kw = dict(frame_idx=frame_idx, num_frames=T, shard=shard, full=full)

        clip_ctx = (ctx if full is None else full["ctx"]).view(n_img // T, T, -1)[:, 0]  # context[::T]: first frame of every clip
        self._emb_tls.table = {"emb": ops.linear(emb_silu, pk["emb_cat"], out_f32=True),   # all emb_layers projections of this forward
                               "ctx_s": ops.linear(ctx, pk["ctx_s"], out_f32=True) if "ctx_s" in pk else None,        # all spatial cross-attention context vectors
                               "ctx_t": ops.linear(clip_ctx, pk["ctx_t"], out_f32=True) if "ctx_t" in pk else None}   # all temporal ones (one per clip)
        try:
            hs = []
            h = tokens
            for module in self.input_blocks:
                h, H, W = module(h, emb_silu, context=ctx, H=H, W=W, **kw)
                hs.append(h)
            h, H, W = self.middle_block(h, emb_silu, context=ctx, H=H, W=W, **kw)
            for module in self.output_blocks:
                # `torch.cat([h, hs.pop()], dim=1)` (video_model.py:493) is never materialised: the ResBlock's first norm and its
                # 1x1 skip conv read the two tensors in place
                h, H, W = module((h, hs.pop()), emb_silu, context=ctx, H=H, W=W, **kw)
        finally:
            self._emb_tls.table = None
        gn = self.out[0]
        h = ops.groupnorm(h, gn.weight, gn.bias, gn.eps, silu=True)
        out, _, _ = ops.conv3x3(h, pk["out"], tokens.shape[0], H, W, out_f32=True)
        return out


class _InputConv(TimestepEmbedSequential, Packable):
    """input_blocks.0 = TimestepEmbedSequential(conv3x3(in_channels -> model_channels)); key `input_blocks.0.0.weight`."""

    def __init__(self, in_channels, model_channels):
        super().__init__(conv_nd(2, in_channels, model_channels, 3, padding=1))

    def _pack(self, dev):
        return ops.pack_conv3x3(self[0].weight, self[0].bias, cin_pad=CIN_PAD, device=dev)

    def forward(self, x, emb_silu, context=None, frame_idx=None, num_frames=None, H=None, W=None, shard=None, full=None):
        out, H, W = ops.conv3x3(x, self.packed(), x.shape[0], H, W)
        return out, H, W
