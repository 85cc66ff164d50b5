"""Temporal transformer and the interleaved spatial/temporal block (reference: vwm/modules/video_attention.py).

Token-major layout makes the reference's `(b t) s c <-> (b s) t c` rearranges (video_attention.py:116,140) free:
LayerNorm, the GEGLU feed-forwards and the q/k/v/out projections are per-token, so they run on the (b t s) row order
unchanged, and only the 25x25 attention core gathers across frames (vk_attn_temporal_bf16 reads rows with stride S).

  VideoTransformerBlock._forward (video_attention.py:111-141), per pixel over T frames:
      x = ff_in(norm_in(x)) + x        x_mix = x + frame-pos-emb leaves the spatial block's last GEMM epilogue
      x = attn1(norm1(x)) + x          fused q|k|v GEMM -> temporal attention -> out GEMM (+res)
      x = attn2(norm2(x), ctx) + x     one-token context (first frame's, video_attention.py:252-257): per-clip constant
                                       vector, added as a row vector in the attn1 out-projection epilogue
      x = ff(norm3(x)) + x             the AlphaBlender mix with the spatial branch is fused into this GEMM's epilogue

Every LayerNorm is folded into the GEMM that consumes it (include/vista_hip.h, VkGemmDesc.ln_*): the producing GEMM's epilogue
emits per-row (sum, sum of squares) partials of its bf16 output and the consumer applies rstd*(acc - mean*colsum) in its own
epilogue, so no normalised tensor is written or read (7 full HBM passes per block pair less).
"""
import torch
import torch.nn as nn

from .. import ops
from .attention import FeedForward, MemoryEfficientCrossAttention, Packable, SpatialTransformer
from .diffusionmodules.util import AlphaBlender, LayerNorm, Linear, SiLU, mlp_f32, timestep_embedding


class VideoTransformerBlock(nn.Module, Packable):
    ATTENTION_MODES = {"softmax": MemoryEfficientCrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, use_checkpoint=False, timesteps=None,
                 ff_in=False, inner_dim=None, attn_mode="softmax", disable_self_attn=False, disable_temporal_crossattention=False,
                 switch_temporal_ca_to_sa=False, add_lora=False, action_control=False):
        super().__init__()
        attn_cls = self.ATTENTION_MODES[attn_mode]
        self.ff_in = ff_in or inner_dim is not None
        if inner_dim is None:
            inner_dim = dim
        assert int(n_heads * d_head) == inner_dim
        self.is_res = inner_dim == dim
        if not self.is_res:
            raise NotImplementedError("inner_dim != dim is not the Vista configuration")
        if disable_self_attn or switch_temporal_ca_to_sa:
            raise NotImplementedError("disable_self_attn / switch_temporal_ca_to_sa are not used by Vista")
        if self.ff_in:
            self.norm_in = LayerNorm(dim)
            self.ff_in = FeedForward(dim, dim_out=inner_dim, dropout=dropout, glu=gated_ff)
        self.timesteps = timesteps
        self.disable_self_attn = False
        self.attn1 = attn_cls(query_dim=inner_dim, heads=n_heads, dim_head=d_head, dropout=dropout, causal=False, add_lora=add_lora)
        self.attn1.temporal = True
        self.ff = FeedForward(inner_dim, dim_out=dim, dropout=dropout, glu=gated_ff)
        self.has_cross = not disable_temporal_crossattention
        if self.has_cross:
            self.norm2 = LayerNorm(inner_dim)
            self.attn2 = attn_cls(query_dim=inner_dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                                  add_lora=add_lora, action_control=action_control)
        self.norm1 = LayerNorm(inner_dim)
        self.norm3 = LayerNorm(inner_dim)
        self.switch_temporal_ca_to_sa = False
        self.use_checkpoint = use_checkpoint
        self.n_heads, self.dim = n_heads, dim

    def _pack(self, dev):
        # the four LayerNorms of the block, folded into the GEMMs that consume them (this block owns the norms)
        a = self.attn1
        return {"ffin_in": self.ff_in.pack_in_folded(self.norm_in, dev),
                "qkv": ops.pack_linear_cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], dev, ln=self.norm1),
                "ff_in": self.ff.pack_in_folded(self.norm3, dev)}

    def forward(self, x_mix, stats, emb_rows, clip_context, B, T, S, alpha, emit_stats=False):
        """x_mix: ((b t) s, dim) bf16 = spatial branch output + frame-position embedding (video_attention.py:283-284) with its
        RowStats; emb_rows: (b*t, dim) f32 that embedding; clip_context: (b, ctx_width) bf16 (context of each clip's first frame).
        Returns alpha*x_spatial + (1-alpha)*temporal_branch (AlphaBlender, util.py:311-318) with x_spatial = x_mix - emb recovered
        in the last epilogue (res2 + rowvec2), so the un-embedded tensor is never stored."""
        assert self.timesteps is None or self.timesteps == T
        if not self.ff_in:
            raise NotImplementedError("extra_ff_mix_layer=False is not the Vista configuration")
        pk = self.packed()
        a1 = self.attn1.packed()
        x, st = self.ff_in.forward_folded(x_mix, stats, pk["ffin_in"], self.norm_in, res1=x_mix, emit_stats=True)
        qkv = ops.linear(x, pk["qkv"], ln=st, alt_cols_from=2 * x.shape[-1])   # (fp16 build: the V block leaves as bf16; no-op in the bf16 build)
        att = ops.attn_temporal(qkv, B, T, S, self.n_heads, self.attn1.dim_head ** -0.5)
        if self.has_cross:
            cv = self.attn2.context_vector(clip_context)  # (b, dim) f32, constant over the clip's frames and pixels
            x, st = ops.linear(att, a1["out"], res1=x, rowvec=cv, rows_per_vec=T * S, emit_stats=True)
        else:
            x, st = ops.linear(att, a1["out"], res1=x, emit_stats=True)
        r = self.ff.forward_folded(x, st, pk["ff_in"], self.norm3, res1=x, alpha=1.0 - alpha, res2=x_mix, rowvec2=emb_rows.neg_rows,
                                   beta=alpha, rows_per_vec=S, emit_stats=emit_stats)
        return r if emit_stats else (r, None)


class _EmbRows:
    """Frame-position embedding rows (b*t, C) f32 and their negation (the blend epilogue subtracts them again)."""

    __slots__ = ("rows", "neg_rows")

    def __init__(self, rows):
        self.rows, self.neg_rows = rows, rows.neg()


class SpatialVideoTransformer(SpatialTransformer):
    """video_attention.py:147-296"""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, use_linear=False, context_dim=None,
                 use_spatial_context=False, timesteps=None, merge_strategy="fixed", merge_factor=0.5, time_context_dim=None,
                 ff_in=False, use_checkpoint=False, time_depth=1, attn_mode="softmax", disable_self_attn=False,
                 disable_temporal_crossattention=False, max_time_embed_period=10000, add_lora=False, action_control=False):
        super().__init__(in_channels, n_heads, d_head, depth=depth, dropout=dropout, attn_type=attn_mode, use_checkpoint=use_checkpoint,
                         context_dim=context_dim, use_linear=use_linear, disable_self_attn=disable_self_attn, add_lora=add_lora,
                         action_control=action_control)
        self.time_depth = time_depth
        self.depth = depth
        self.max_time_embed_period = max_time_embed_period
        inner_dim = n_heads * d_head
        if not use_spatial_context:
            raise NotImplementedError("use_spatial_context=False (separate time_context) is not the Vista configuration")
        time_context_dim = context_dim
        self.time_stack = nn.ModuleList([
            VideoTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=time_context_dim, timesteps=timesteps,
                                  use_checkpoint=use_checkpoint, ff_in=ff_in, inner_dim=inner_dim, attn_mode=attn_mode,
                                  disable_self_attn=disable_self_attn, disable_temporal_crossattention=disable_temporal_crossattention,
                                  add_lora=add_lora, action_control=action_control) for _ in range(self.depth)])
        assert len(self.time_stack) == len(self.transformer_blocks)
        self.use_spatial_context = use_spatial_context
        self.in_channels = in_channels
        time_embed_dim = in_channels * 4
        self.time_pos_embed = nn.Sequential(Linear(in_channels, time_embed_dim), SiLU(), Linear(time_embed_dim, in_channels))
        self.time_mixer = AlphaBlender(alpha=merge_factor, merge_strategy=merge_strategy, rearrange_pattern="b t -> (b t) 1 1")
        self.n_heads = n_heads

    def _pack(self, dev):
        return {"proj_in": ops.pack_linear(self.proj_in.weight, self.proj_in.bias, dev),
                "proj_out": ops.pack_linear(self.proj_out.weight, self.proj_out.bias, dev),
                "tpe0": ops.pack_linear(self.time_pos_embed[0].weight, self.time_pos_embed[0].bias, dev),
                "tpe2": ops.pack_linear(self.time_pos_embed[2].weight, self.time_pos_embed[2].bias, dev),
                "alpha": self.time_mixer.alpha_value()}

    def forward(self, x, context, frame_idx, T, H, W, shard=None, full=None, x_gn=None):
        """x: (n_img, S, C) bf16 tokens; context: (n_img, ctx_width) bf16 (one token per image); frame_idx: (B*T,) f32 frame
        index of every image of the window (arange(T) repeated per clip, video_attention.py:270-271).
        Multi-GPU (`shard`): x / context hold this rank's frames; the temporal block runs pixel-sharded between two
        all-to-alls and uses the replicated full["ctx"] for the clips' first-frame context."""
        pk = self.packed()
        n_img, S, C = x.shape
        x_in = x
        # (x_gn: the statistics of x from the epilogue of the temporal convolution that produced it, ops.GnPartials -- no statistics pass then)
        h = ops.groupnorm(x, self.norm.weight, self.norm.bias, self.norm.eps, silu=False, gn=x_gn)
        h, st = ops.linear(h, pk["proj_in"], emit_stats=True)                      # (n_img*S, C) + row sums for norm1
        # Frame-position embedding time_pos_embed(sinusoid(frame_idx)) (video_attention.py:270-276): a function of the WEIGHTS and the frame
        # indices only. The UNet hands every forward the same frame_idx tensor (VideoUNet._frame_idx), so the rows are kept next to the packed
        # weights and re-used while that very tensor is unchanged (identity + version counter; a repacked weight drops pk and the rows with
        # it): the sinusoid + two 25-row GEMMs + SiLU were 4 latency-bound launches x 16 transformers per step (1.2 ms of 187).
        tc = pk.get("_tpe_rows")
        fv = Packable._param_version(frame_idx)
        if tc is not None and tc[0] is frame_idx and tc[1] == fv:
            emb = tc[2]
        else:
            emb = _EmbRows(mlp_f32(timestep_embedding(frame_idx, self.in_channels, self.max_time_embed_period), pk["tpe0"], pk["tpe2"]))
            pk["_tpe_rows"] = (frame_idx, fv, emb)
        ctx_full = context if shard is None else full["ctx"]
        B = ctx_full.shape[0] // T
        clip_context = ctx_full.view(B, T, -1)[:, 0]                               # context[::T] (first frame of each clip)
        emb_local = emb.rows if shard is None else shard.take_local_rows(emb.rows)  # spatial half sees this rank's frames only
        last = len(self.transformer_blocks) - 1
        for i, (block, mix_block) in enumerate(zip(self.transformer_blocks, self.time_stack)):
            # spatial block; its last epilogue adds the frame-position embedding: h = x_spatial + emb
            h, st = block(h, st, context, n_img, S, out_rowvec=emb_local, emit_stats=shard is None)
            if shard is None:
                h, st = mix_block(h, st, emb, clip_context, B, T, S, alpha=pk["alpha"], emit_stats=i != last)
            else:
                hp = shard.to_pixels(h.view(n_img, S, C))                          # (B*T, S_r, C)
                s_r = hp.shape[1]
                nch = min(getattr(shard, "a2a_chunks", 1), min(shard.pixel_counts(S)))  # (the same on every rank: the narrowest slice decides)
                if nch <= 1:
                    hp = hp.view(-1, C)
                    hp, _ = mix_block(hp, ops.rowstats(hp), emb, clip_context, B, T, s_r, alpha=pk["alpha"])
                    h = shard.to_frames(hp.view(B * T, s_r, C), S).view(-1, C)
                else:
                    # opt-in overlap (VISTA_A2A_CHUNKS): the temporal block is pointwise in space, so it runs on pixel sub-ranges in turn and
                    # sub-range i travels back to the frame layout while sub-range i + 1 computes
                    out = torch.empty((n_img, S, C), dtype=hp.dtype, device=hp.device)
                    pending = []
                    for ci, (lo, hi) in enumerate(shard.pixel_chunks(S, nch)):
                        part = hp[:, lo:hi].contiguous().view(-1, C)
                        part, _ = mix_block(part, ops.rowstats(part), emb, clip_context, B, T, hi - lo, alpha=pk["alpha"])
                        pending.append(shard.to_frames_begin(part.view(B * T, hi - lo, C), S, nch, ci))
                    for pnd in pending:
                        shard.to_frames_end(pnd, out)
                    h = out.view(-1, C)
                st = ops.rowstats(h) if i != last else None
        out = ops.linear(h, pk["proj_out"], res1=x_in)
        return out.view(n_img, S, C)
