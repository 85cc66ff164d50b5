"""Spatial transformer of the VideoUNet (reference: vwm/modules/attention.py), MI355X-native.

Same classes, constructor arguments and state-dict names as the reference; `forward` works on token-major bf16
activations ((n_img*S, C), C contiguous) and runs entirely on the HIP kernels of libvista_hip.so:

  BasicTransformerBlock (attention.py:514-524)
      x += to_out(softmax(q k^T/8) v)        LN folded -> ONE fused q|k|v GEMM -> vk_attn_spatial_qkv_bf16 -> out GEMM(+res)
      x += attn2(norm2(x), context)          context is ONE token (CLIP (+) action embeddings), so softmax == 1 and the
                                             output is to_out(to_v(ctx) + v_adapter(ctx_act)) for every query
                                             (attention.py:341-353,400-421): computed per image by two tiny GEMMs and
                                             added as a per-image row vector in the attn1 out-projection epilogue.
                                             Bit-for-bit the same function of the inputs; q/k/norm2 weights cannot
                                             influence the result and are kept only for state-dict compatibility.
      x += FF(norm3(x))                      LN -> GEGLU GEMM (value*gelu(gate) fused in the epilogue) -> out GEMM(+res)
"""
import os

import torch
import torch.nn as nn

from .. import ops
from .diffusionmodules.util import Dropout, LayerNorm, Linear, NormParams, zero_module


def exists(val):
    return val is not None


def default(val, d):
    return val if exists(val) else (d() if callable(d) else d)


class Packable:
    """Lazy bf16 weight packing (ops.pack_*). The pack is keyed on the device, dtype, identity and in-place version counters of the
    parameters it was built from, so it is rebuilt after `.cuda()`, after ANY load_state_dict (also one issued on a parent
    container, which never reaches the children's own load_state_dict; also `assign=True`, which REPLACES the Parameter objects),
    after `module.weight = nn.Parameter(...)` and after in-place updates of the Parameters (`p.copy_()`, `p.mul_()` under no_grad).
    Replacement is detected through the owning modules' `_parameters` slots: every call checks that each slot still holds the very
    object the pack was built from (a dict lookup per parameter; re-walking `parameters()` would cost ~10 us x 600 packs per step).
    Writes through `p.data` (the reference's LitEma.copy_to, vwm/modules/ema.py) carry their own version counter by PyTorch's design
    and are invisible here, and so are in-place updates of INFERENCE tensors (parameters created or moved under torch.inference_mode()
    have no version counter at all): call `invalidate_packed(model)` after either. Every load_state_dict on a root that owns packs
    (VideoUNet, GeneralConditioner, FrozenOpenCLIPImageEmbedder, AutoencoderKLModeOnly) drops them through a post-hook regardless."""
    _pk = None
    _pk_key = None
    _pk_params = None
    _pack_device_types = ("cuda",)  # (the CPU tests of the cache-key logic widen this on a stub class; product classes never do)

    def _pack_params(self):
        """Parameters the pack depends on (default: every parameter under this module)."""
        return list(self.parameters())

    def _resolve_pack_params(self):
        """[(owner `_parameters` dict or None, name, tensor)] for every tensor of _pack_params(). The owner slot is what lets packed()
        notice a REPLACED Parameter (the cached tensor itself would keep its old version counter for ever)."""
        ps = self._pack_params()
        where = {}
        for m in self.modules():
            for n, q in m._parameters.items():
                if q is not None:
                    where.setdefault(id(q), (m._parameters, n))
        slots = [(*where.get(id(q), (None, None)), q) for q in ps]
        object.__setattr__(self, "_pk_params", slots)  # plain attribute: nn.Module.__setattr__ would try to register it
        return slots

    @staticmethod
    def _param_version(p):
        return 0 if p.is_inference() else p._version  # inference tensors (a model moved under torch.inference_mode()) have no counter

    def packed(self):
        slots = self._pk_params
        if slots is None:
            slots = self._resolve_pack_params()
        else:
            for d, n, q in slots:
                if d is not None and d.get(n) is not q:  # the Parameter object was replaced (assign=True load, re-assignment, parametrize)
                    slots = self._resolve_pack_params()
                    self._pk = None
                    break
        p0 = slots[0][2]
        dev = p0.device
        if dev.type not in self._pack_device_types:
            raise ops._lib.VistaHipError(f"{self.__class__.__name__}: parameters are on {dev}; move the model to the MI355X "
                                         "(.cuda()) -- vista_amd has no CPU path")
        key = (dev, p0.dtype, *[(id(q), self._param_version(q)) for _, _, q in slots])
        if self._pk is None or self._pk_key != key:
            with torch.no_grad():
                self._pk = self._pack(dev)
            self._pk_key = key
        return self._pk

    def invalidate_packed(self):
        self._pk = None
        object.__setattr__(self, "_pk_params", None)  # the parameter list is re-resolved on the next pack
        _PACK_GEN[0] += 1   # anything that captured pointers into the old pack (a hipGraph of the forward) is stale from here on


_PACK_GEN = [0]


def pack_generation():
    """Counts invalidate_packed() calls process-wide. A captured hipGraph of a forward holds raw pointers into packed weights; its cache
    (sampling.FusedLoop._graph_for) compares this number, so an EMA swap through p.data + invalidate_packed(model), an in-place load into
    inference tensors or any load_state_dict (post-hook) re-captures instead of replaying launches that read freed memory."""
    return _PACK_GEN[0]


def _invalidate_after_load(module, incompatible_keys):
    invalidate_packed(module)


def invalidate_packed(module):
    """Drop every packed weight under `module` (needed only after writes through `param.data`, e.g. an EMA swap)."""
    for m in module.modules():
        if isinstance(m, Packable):
            m.invalidate_packed()


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)


# BASELINE config 5 opt-in (off by default; the headline path is bf16): run the FeedForward GEMMs -- GEGLU in-projection and
# out-projection, 32 % of the UNet's FLOPs -- in fp8 e4m3 with per-token activation scales and per-channel weight scales.
# "conv": additionally the ResBlock convolutions (2-D 3x3 and temporal 3x1x1) on e4m3 GroupNorm output (openaimodel.py ResBlock).
# "attention": the spatial self-attention's score product in fp8 -- q | k leave the fused q|k|v projection's epilogue as MX fp8 (no pass) and
# S^T = K Q^T is one scaled 32x32x64 fp8 MFMA per 32x32 block; "proj": additionally the attention output leaves the kernel as MX fp8 and the
# attention-out projection (K = C) runs as an fp8 GEMM on it.
FP8_PROJ_MIN_WIDTH = 640   # config 5: attention-out projections in fp8 from this block width up (round-4 probe: no gain at 320)
FP8 = {"feedforward": os.environ.get("VISTA_FP8", "0") == "1", "conv": os.environ.get("VISTA_FP8_CONV", "0") == "1",
       "attention": os.environ.get("VISTA_FP8_ATTN", "0") == "1", "proj": os.environ.get("VISTA_FP8_PROJ", "0") == "1"}


FF_FUSED = os.environ.get("VISTA_FF_FUSED", "1") != "0"    # A/B hook: 0 = the level-0 FeedForward as two GEMM launches (rounds 1-3)
QKV_SPLIT = os.environ.get("VISTA_QKV_SPLIT", "0") == "1"  # A/B hook: spatial self-attention's projections as in round 2 (q|k + V^T GEMMs)
Q_LOG2 = os.environ.get("VISTA_ATTN_QLOG2", "1") != "0"   # A/B hook: 0 = unscaled query rows + the scale applied inside the attention kernel


class FeedForward(nn.Module, Packable):
    """attention.py:95-128 (glu=True on this path). net = [GEGLU, Dropout, Linear] -> keys net.0.proj.*, net.2.*"""

    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0, zero_init=False):
        super().__init__()
        if not glu:
            raise NotImplementedError("non-gated FeedForward is not on the Vista inference path")
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        self.net = nn.Sequential(GEGLU(dim, inner_dim), Dropout(dropout), Linear(inner_dim, dim_out))
        if zero_init:
            zero_module(self.net[-1])

    def _pack(self, dev):
        # "in" (the UNFOLDED GEGLU weight) is packed on first use by forward(): the UNet path only runs forward_folded, whose LN-folded
        # copy belongs to the owning block -- packing both would keep ~0.6 GB of dead bf16 weights on the device for the 1.65 B network
        # the bf16 out-projection operand is packed on first use as well (_out_pack): plain for vk_gemm_bf16, or -- level 0, width 320 -- in the
        # fused kernel's own layout; whichever form the switches select at call time, so that flipping them restores the other path bit for bit
        pk = {}
        if self._fp8():
            pk["in8"] = ops.pack_geglu_fp8(self.net[0].proj.weight, self.net[0].proj.bias, dev)
            pk["out8"] = ops.pack_linear_fp8(self.net[2].weight, self.net[2].bias, dev)
        return pk

    def _fp8(self):
        """BASELINE config 5 for THIS FeedForward? Not where the fused bf16 kernel runs (level 0, width 320): round-4 same-box probe
        (tools/fp8_vs_bf16_probe.py, 460800 tokens): fused bf16 1.28 ms at 2.7e-3 of fp32 against 1.38 ms at 4.0e-2 for the fp8 pair --
        fp8 there costs time AND accuracy; levels 1 / 2: 0.93 vs 1.17 ms and 0.70 vs 1.03 ms, fp8 stays."""
        if not FP8["feedforward"] or ops.ACT is not torch.bfloat16:   # (config 5 exists in the bf16 build only)
            return False
        w2 = self.net[2].weight
        return not (FF_FUSED and w2.shape[0] == ops.FF_FUSED_WIDTH and w2.shape[1] % 64 == 0 and 128 <= w2.shape[1] <= ops.FF_FUSED_MAX_HIDDEN)

    def _out_pack(self, pk):
        """(fused?, packed out-projection) for the bf16 path. Level-0 FeedForward (width 320): GEGLU + out-projection run as ONE kernel
        (vk_ff_fused_bf16), whose out-projection operand has its own layout (K permuted to the MFMA accumulator order, chunk-major)."""
        w2 = self.net[2].weight
        if FF_FUSED and w2.shape[0] == ops.FF_FUSED_WIDTH and w2.shape[1] % 64 == 0 and 128 <= w2.shape[1] <= ops.FF_FUSED_MAX_HIDDEN:
            if "out_fused" not in pk:
                pk["out_fused"] = ops.pack_ff_out(w2, self.net[2].bias, w2.device)
            return True, pk["out_fused"]
        if "out" not in pk:
            pk["out"] = ops.pack_linear(w2, self.net[2].bias, w2.device)
        return False, pk["out"]

    def pack_in_folded(self, norm, dev):
        """GEGLU in-projection with the preceding LayerNorm `norm` folded in (the owner block packs it: it owns the norm)."""
        return ops.pack_geglu(self.net[0].proj.weight, self.net[0].proj.bias, dev, ln=norm)

    def forward(self, y, **epilogue):
        """y: LN output (M, dim). Returns net(y) fused with the residual / blend epilogue given by the caller (unfolded form: used
        by the fp8 experiment and by callers that already hold a normalised input)."""
        pk = self.packed()
        if not self._fp8():
            if "in" not in pk:  # lives in the pack dict, so it is dropped with it whenever the parameters change
                pk["in"] = ops.pack_geglu(self.net[0].proj.weight, self.net[0].proj.bias, self.net[2].weight.device)
            fused, pw_out = self._out_pack(pk)
            if fused:
                return ops.ff_fused(y, pk["in"], pw_out, **epilogue)
            return ops.linear(ops.linear(y, pk["in"]), pw_out, **epilogue)
        if "in8" not in pk:  # the switch was flipped after the bf16 pack was built
            self.invalidate_packed()
            pk = self.packed()
        yq, ys = ops.quantize_rows_fp8(y)
        h = ops.linear_fp8(yq, ys, pk["in8"])
        hq, hs = ops.quantize_rows_fp8(h)
        return ops.linear_fp8(hq, hs, pk["out8"], **epilogue)

    def forward_folded(self, x, stats, pw_in, norm, **epilogue):
        """net(LayerNorm(x)) with the LayerNorm folded into the GEGLU GEMM: x (M, dim) is the un-normalised residual stream,
        `stats` its RowStats, `pw_in` = pack_in_folded(norm). No normalised tensor is ever written (attention.py:524)."""
        if self._fp8():
            # BASELINE config 5: both GEMMs in fp8 e4m3 with NO stand-alone quantisation pass. LayerNorm + per-row quantisation is one
            # kernel (the normalised rows exist only as fp8), the GEGLU epilogue quantises its own output to MX fp8 (a power-of-two scale
            # per 32 columns), and the out-projection applies those block scales inside v_mfma_scale_f32_32x32x64_f8f6f4.
            pk = self.packed()
            if "in8" not in pk:  # the switch was flipped after the bf16 pack was built
                self.invalidate_packed()
                pk = self.packed()
            yq, ys = ops.layernorm_quant_fp8(x, norm)
            h8, hs = ops.linear_fp8(yq, ys, pk["in8"], mx_out=True)
            return ops.linear_fp8(h8, None, pk["out8"], a_mx=hs, **epilogue)
        fused, pw_out = self._out_pack(self.packed())
        if fused:
            return ops.ff_fused(x, pw_in, pw_out, ln=stats, **epilogue)
        return ops.linear(ops.linear(x, pw_in, ln=stats), pw_out, **epilogue)


class MemoryEfficientCrossAttention(nn.Module, Packable):
    """attention.py:246-421. Parameter names identical; LoRA branches are inference-off (`add_lora: False`,
    vista.yaml:39; merged offline by bin_to_st.py) and rejected here."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, zero_init=False, causal=False,
                 add_lora=False, lora_rank=16, lora_scale=1.0, action_control=False, **kwargs):
        super().__init__()
        if add_lora:
            raise NotImplementedError("add_lora=True is a training-time option; merge LoRA weights offline (bin_to_st.py)")
        if causal:
            raise NotImplementedError("causal attention is not used by Vista")
        if dim_head != 64:
            raise NotImplementedError("the gfx950 attention kernels are specialised for head dim 64 (num_head_channels: 64)")
        inner_dim = dim_head * heads
        self.is_self = context_dim is None
        context_dim = default(context_dim, query_dim)
        self.heads, self.dim_head, self.inner_dim, self.query_dim = heads, dim_head, inner_dim, query_dim
        self.to_q = Linear(query_dim, inner_dim, bias=False)
        self.to_k = Linear(context_dim, inner_dim, bias=False)
        self.to_v = Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(Linear(inner_dim, query_dim), Dropout(dropout))
        if zero_init:
            zero_module(self.to_out[0])
        self.add_lora = False
        self.action_control = action_control
        self.context_dim = context_dim
        if action_control:
            self.k_adapter_action_control = zero_module(Linear(128 * 19, inner_dim, bias=False))
            self.v_adapter_action_control = zero_module(Linear(128 * 19, inner_dim, bias=False))

    def _pack(self, dev):
        if self.is_self:
            pk = {"out": ops.pack_linear(self.to_out[0].weight, self.to_out[0].bias, dev)}
            if FP8["proj"] and self.query_dim >= FP8_PROJ_MIN_WIDTH:   # (narrower blocks keep the bf16 out-projection: never read there)
                pk["out8"] = ops.pack_linear_fp8(self.to_out[0].weight, self.to_out[0].bias, dev)
            return pk
        w, b = self.context_map()
        return {"ctx": ops.pack_linear(w, b, dev)}

    def context_map(self):
        """Cross-attention against a ONE-token context is a single affine map of that token: softmax over one key == 1, so the output is
        to_out(to_v(ctx) + v_adapter(ctx_act)) for every query (attention.py:341-353,400-421), and with no non-linearity in between
        that is ctx @ (W_out [W_v | W_va])^T + b_out. Returns the composed (query_dim, context_dim [+ 2432]) weight and the bias (fp32)."""
        ws = [self.to_v.weight]
        if self.action_control:
            ws.append(self.v_adapter_action_control.weight)
        wv = torch.cat([w.detach().float() for w in ws], 1)                      # (inner, ctx_width): [to_v | v_adapter] along K
        return self.to_out[0].weight.detach().float() @ wv, self.to_out[0].bias.detach().float()

    def context_vector(self, ctx):
        """attn2(norm2(x), ctx) for a ONE-token context: (n_ctx, query_dim) f32, constant over the queries. ctx: (n_ctx, ctx_width) bf16.
        Inside a VideoUNet forward the value comes from the UNet's one batched GEMM over every cross-attention of the network (a
        column slice of its table); standalone, from this layer's own composed weight."""
        src = getattr(self, "_ctx_src", None)  # (per-thread table holder, key, column offset) when a VideoUNet owns this layer
        if src is not None:
            tab = src[0].table
            t = tab.get(src[1]) if tab is not None else None
            if t is not None and t.shape[0] == ctx.shape[0]:
                return t[:, src[2]:src[2] + self.query_dim]
        pk = self.packed()
        if ctx.shape[-1] != pk["ctx"].K:
            raise ValueError(f"context width {ctx.shape[-1]} does not match to_v (+ action adapter) width {pk['ctx'].K}")
        return ops.linear(ctx, pk["ctx"], out_f32=True)


class BasicTransformerBlock(nn.Module, Packable):
    ATTENTION_MODES = {"softmax": MemoryEfficientCrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, use_checkpoint=False,
                 disable_self_attn=False, attn_mode="softmax", sdp_backend=None, add_lora=False, action_control=False):
        super().__init__()
        assert attn_mode in self.ATTENTION_MODES
        if disable_self_attn:
            raise NotImplementedError("disable_self_attn is not used by Vista")
        attn_cls = self.ATTENTION_MODES[attn_mode]
        self.disable_self_attn = False
        self.attn1 = attn_cls(query_dim=dim, context_dim=None, heads=n_heads, dim_head=d_head, dropout=dropout, add_lora=add_lora)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                              add_lora=add_lora, action_control=action_control)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.norm3 = LayerNorm(dim)
        self.use_checkpoint = use_checkpoint
        self.n_heads, self.dim = n_heads, dim

    def _pack(self, dev):
        # LayerNorms folded into the GEMMs that consume them (norm1 -> the fused q|k|v projection, norm3 -> GEGLU): the block owns the
        # norms, so it packs those weights; attn1 / ff keep their own out-projections. norm2 only feeds the 1-token cross-attention's
        # query, which cannot influence the output (softmax over one key == 1).
        a = self.attn1
        # the query rows carry dim_head^-0.5 * log2(e) (attention.py:400-407: softmax(q k^T * scale)): one bf16 rounding of the scaled projection
        # instead of one of the unscaled one, and the attention kernel's zero-base path needs no per-score scale / base arithmetic
        wq = a.to_q.weight.detach().float() * (a.dim_head ** -0.5 * ops.LOG2E) if Q_LOG2 else a.to_q.weight
        pk = {"qkv": ops.pack_linear_cat([wq, a.to_k.weight, a.to_v.weight], dev, ln=self.norm1),
              "ff_in": self.ff.pack_in_folded(self.norm3, dev)}
        if QKV_SPLIT:  # same-box A/B hook only: the round-2 form (q|k GEMM + V^T GEMM, a second pass over x)
            pk["qk"] = ops.pack_linear_cat([a.to_q.weight, a.to_k.weight], dev, ln=self.norm1)
            pk["v"] = ops.pack_linear(a.to_v.weight, None, dev, ln=self.norm1)
        return pk

    def forward(self, x, stats, context, n_img, S, out_rowvec=None, emit_stats=True):
        """x: (n_img*S, dim) bf16 tokens with their RowStats; context: (n_img, ctx_width) bf16 (one context token per image).
        out_rowvec: optional (n_img, dim) f32 added per image to the block output (the frame-position embedding of
        video_attention.py:283-284, so the temporal block's input leaves this block's last epilogue). Returns (out, RowStats)."""
        pk = self.packed()
        a1 = self.attn1.packed()
        C = self.dim
        scale = self.attn1.dim_head ** -0.5
        att8 = None
        if ops.ACT is not torch.bfloat16 and (QKV_SPLIT or FP8["attention"]):
            raise ops._lib.VistaHipError("VISTA_ACT_DTYPE=fp16: the q|k + V^T split and the fp8 score product (BASELINE config 5) exist in the bf16 build only")
        if QKV_SPLIT:
            qk = ops.linear(x, pk["qk"], ln=stats)
            vt = ops.linear_vt(x, pk["v"], S, ln=stats)
            att = ops.attn_spatial(qk[:, :C], qk[:, C:], vt, n_img, self.n_heads, S, scale)
        elif FP8["attention"] and C % 320 == 0:
            # BASELINE config 5: the same ONE GEMM, whose epilogue writes the q | k blocks as MX fp8 (e4m3 + a power-of-two scale per row and 32
            # head-dim elements) and v as bf16; the score product runs in fp8 with the scales applied inside the MFMA. With "proj" the attention
            # output leaves as MX fp8 too and the out-projection below is an fp8 GEMM -- no quantisation pass anywhere.
            v, qk8, qks = ops.linear(x, pk["qkv"], ln=stats, mx8_cols=2 * C)
            nb = C // 32
            # the fp8 out-projection only from width 640 up: at level 0 the bf16 one (K = N = 320: 0.26 ms) is as fast as the fp8 GEMM and
            # 10x closer to fp32 (tools/fp8_vs_bf16_probe.py)
            proj8 = FP8["proj"] and C >= FP8_PROJ_MIN_WIDTH
            r8 = ops.attn_spatial_fp8qk(qk8[:, :C], qk8[:, C:], qks[:, :nb], qks[:, nb:], v, n_img, self.n_heads, S, 0.0 if Q_LOG2 else scale,
                                        mx_out=proj8)  # (pk["qkv"]'s query rows already carry scale * log2 e)
            att8, att = (r8, None) if proj8 else (None, r8)
        else:
            # ONE q|k|v GEMM (attention.py:344-346), LayerNorm(norm1) folded: x is read once and never as a normalised copy; the attention
            # kernel takes V as the third column block and transposes its tiles on the way out of LDS (no V^T tensor, no TRANS GEMM)
            qkv = ops.linear(x, pk["qkv"], ln=stats, alt_cols_from=2 * C)   # (fp16 build: the V block leaves as bf16; no-op in the bf16 build)
            att = ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n_img, self.n_heads, S, scale, v_rows=True, q_log2=Q_LOG2)
        cv = self.attn2.context_vector(context)  # attn2(norm2(x), context): constant over the image's tokens
        if att8 is not None:
            if "out8" not in a1:  # the switch was flipped after the bf16 pack was built
                self.attn1.invalidate_packed()
                a1 = self.attn1.packed()
            x, st = ops.linear_fp8(att8[0], None, a1["out8"], a_mx=att8[1], res1=x, rowvec=cv, rows_per_vec=S, emit_stats=True)
        else:
            x, st = ops.linear(att, a1["out"], res1=x, rowvec=cv, rows_per_vec=S, emit_stats=True)
        r = self.ff.forward_folded(x, st, pk["ff_in"], self.norm3, res1=x, rowvec=out_rowvec, rows_per_vec=S, emit_stats=emit_stats)
        return r if emit_stats else (r, None)


class SpatialTransformer(nn.Module, Packable):
    """attention.py:527-632 with use_linear=True (the Vista setting). Parameter layout only; the video subclass
    (video_attention.SpatialVideoTransformer) implements forward."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, disable_self_attn=False,
                 use_linear=False, attn_type="softmax", use_checkpoint=False, sdp_backend=None, add_lora=False,
                 action_control=False):
        super().__init__()
        if not use_linear:
            raise NotImplementedError("use_linear_in_transformer=False (1x1 conv projections) is not the Vista configuration")
        if exists(context_dim) and not isinstance(context_dim, (list, tuple)):
            context_dim = [context_dim]
        if exists(context_dim) and isinstance(context_dim, (list, tuple)):
            context_dim = list(context_dim)
            if depth != len(context_dim):
                assert all(c == context_dim[0] for c in context_dim), "Need homogenous context_dim to match depth automatically"
                context_dim = depth * [context_dim[0]]
        elif context_dim is None:
            context_dim = [None] * depth
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.norm = NormParams(in_channels, 1e-6, 32)  # Normalize(): GroupNorm(32, eps=1e-6), attention.py:141-142
        self.proj_in = Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                  disable_self_attn=disable_self_attn, attn_mode=attn_type, use_checkpoint=use_checkpoint,
                                  add_lora=add_lora, action_control=action_control) for d in range(depth)])
        self.proj_out = zero_module(Linear(inner_dim, in_channels))
        self.use_linear = use_linear
