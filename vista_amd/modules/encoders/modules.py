"""Conditioner of the sampling pipeline (reference: vwm/modules/encoders/modules.py), MI355X-native -- SURVEY.md 8f rank 2.

Same class names, constructor arguments, `forward` contracts and state-dict names as the reference, so `conditioner.*` of a Vista
checkpoint loads unchanged and `configs/inference/vista.yaml:42-140` instantiates this package through `instantiate_from_config`:

  GeneralConditioner                      modules.py:70-180   key routing ("vector" / "crossattn" / "concat"), zero embeddings for absent
                                                              action keys, force_zero_embeddings, get_unconditional_conditioning
  FrozenOpenCLIPImageEmbedder             modules.py:251-399  OpenCLIP ViT-H/14 image tower: antialiased bicubic resize to 224 + CLIP
                                                              mean/std (ONE HIP kernel, fused with the patch convolution's im2col), 32
                                                              pre-LN transformer blocks on the GEMM family (LayerNorms folded into the
                                                              q|k|v / c_fc GEMMs, GELU in the c_fc epilogue, residuals in the out_proj /
                                                              c_proj epilogues), 16 x 80-dim heads over 257 tokens by vk_attn_small_bf16,
                                                              ln_post + projection of the class token
  FrozenOpenCLIPImagePredictionEmbedder   modules.py:505-516
  ConcatTimestepEmbedderND                modules.py:402-425  fp32 sinusoids (vk_timestep_embedding_f32)
  VideoPredictionEmbedderWithEncoder      modules.py:428-503  the first-stage encoder in mode-only form (models/autoencoder.AutoencoderKLModeOnly)

`open_clip` is not importable offline and its laion2b weights cannot be downloaded: the tower is built from the published architecture
(ViT-H-14: width 1280, 32 layers, 16 heads, MLP 5120, patch 14, 224 px, 1024-d projection, exact-erf GELU) with open_clip's parameter
names; weights come from the checkpoint (`conditioner.embedders.0.open_clip.model.visual.*`). Parity is pinned against
`transformers.CLIPVisionModelWithProjection` -- the same published algorithm -- in oracle/clip_oracle.py.
Training-time options (ucg dropout, image crops, token outputs, sigma samplers) raise NotImplementedError with the option named.
"""

import torch
import torch.nn as nn

from ... import ops
from ...util import default, instantiate_from_config
from ..attention import Packable
from ..diffusionmodules.util import LayerNorm, Linear

OPENCLIP_VISION_GEOMETRY = {  # open_clip model_configs/ViT-H-14.json, vision_cfg (+ embed_dim)
    "ViT-H-14": dict(width=1280, layers=32, heads=16, mlp=5120, patch=14, image=224, embed=1024),
}


class AbstractEmbModel(nn.Module):
    """modules.py:27-67: an embedder carries `is_trainable`, `ucg_rate` and `input_key`, set by GeneralConditioner from its config entry."""

    def __init__(self):
        super().__init__()
        self.is_trainable = None
        self.ucg_rate = None
        self.input_key = None


class GeneralConditioner(nn.Module):
    """Runs every embedder on its batch entry and routes the result by rank: 2-D -> "vector", 3-D -> "crossattn", 4/5-D -> "concat";
    same-key outputs are concatenated (vector along 1, crossattn along 2, concat along 1)."""
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}

    def __init__(self, emb_models):
        super().__init__()
        built = []
        for cfg in emb_models:
            emb = instantiate_from_config(cfg)
            if not isinstance(emb, AbstractEmbModel):
                raise TypeError(f"embedder {type(emb).__name__} must derive from AbstractEmbModel")
            emb.is_trainable = cfg.get("is_trainable", False)
            emb.ucg_rate = cfg.get("ucg_rate", 0.0)
            if emb.is_trainable:
                raise NotImplementedError("is_trainable embedders are a training-time option")
            # ucg_rate / legacy_ucg_value are stored as the reference stores them (encoders/modules.py:85-104): its training YAMLs set
            # ucg_rate 0.15 and must instantiate here too. The dropout itself is training-time arithmetic on torch's RNG: forward() refuses
            # it, get_unconditional_conditioning() -- the only entry the sampling path uses -- zeroes the rates around its two passes exactly
            # like the reference (:172-180).
            for p_ in emb.parameters():
                p_.requires_grad = False
            emb.eval()
            if "input_key" in cfg:
                emb.input_key = cfg["input_key"]
            elif "input_keys" in cfg:
                emb.input_keys = cfg["input_keys"]
            else:
                raise KeyError(f"Need either `input_key` or `input_keys` for embedder {type(emb).__name__}")
            emb.legacy_ucg_val = cfg.get("legacy_ucg_value", None)
            built.append(emb)
        self.embedders = nn.ModuleList(built)
        # a load_state_dict issued on this container never reaches the children's own hooks: drop every packed weight underneath
        from ..attention import _invalidate_after_load
        self.register_load_state_dict_post_hook(_invalidate_after_load)

    def _embed(self, emb, batch):
        """One embedder's output list, or None when its key is absent and it adds no sequence entry (modules.py:121-133)."""
        key = getattr(emb, "input_key", None)
        if key is not None:
            if key in batch:
                out = emb(batch[key])
            elif getattr(emb, "add_sequence_dim", False):  # absent action: a zero token segment keeps the crossattn width fixed
                ref = batch["cond_aug"]
                out = torch.zeros((ref.shape[0], 1, emb.num_features * emb.outdim), device=ref.device)
            else:
                return None
        else:
            out = emb(*[batch[k] for k in emb.input_keys])
        return list(out) if isinstance(out, (list, tuple)) else [out]

    @torch.no_grad()
    def forward(self, batch, force_zero_embeddings=None):
        zeroed = set(default(force_zero_embeddings, list()))
        output = {}
        for emb in self.embedders:
            if emb.ucg_rate and emb.ucg_rate > 0.0:
                raise NotImplementedError("conditioning dropout (ucg_rate > 0) is applied only in training; sampling goes through "
                                          "get_unconditional_conditioning(), which disables it like the reference does")
            outs = self._embed(emb, batch)
            if outs is None:
                continue
            for e in outs:
                if not torch.is_tensor(e):
                    raise TypeError(f"Encoder outputs must be tensors or a sequence, but got {type(e)}")
                key = self.OUTPUT_DIM2KEYS[e.dim()]
                if getattr(emb, "input_key", None) in zeroed:
                    e = torch.zeros_like(e)
                if key not in output:
                    output[key] = e
                elif key == "vector" and e.shape[-1] == 768:   # (a full-width vector embedder adds instead of concatenating, :169-170)
                    output[key] = output[key] + e
                else:
                    output[key] = torch.cat((output[key], e), self.KEY2CATDIM[key])
        return output

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_cond_zero_embeddings=None, force_uc_zero_embeddings=None):
        rates = [emb.ucg_rate for emb in self.embedders]
        for emb in self.embedders:
            emb.ucg_rate = 0.0
        try:
            c = self(batch_c, force_cond_zero_embeddings)
            uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings)
        finally:
            for emb, r in zip(self.embedders, rates):
                emb.ucg_rate = r
        return c, uc


# ------------------------------------------------------------------------------------------------ OpenCLIP image tower
class _Attn(nn.Module):
    """nn.MultiheadAttention parameter layout: in_proj_weight (3C, C) = [q; k; v], in_proj_bias (3C), out_proj."""

    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * width, width) * width ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = Linear(width, width)


class _Mlp(nn.Module):
    def __init__(self, width, hidden):
        super().__init__()
        self.c_fc = Linear(width, hidden)
        self.c_proj = Linear(hidden, width)


class _ResBlock(nn.Module):
    def __init__(self, width, hidden):
        super().__init__()
        self.ln_1 = LayerNorm(width)
        self.attn = _Attn(width)
        self.ln_2 = LayerNorm(width)
        self.mlp = _Mlp(width, hidden)


class _Transformer(nn.Module):
    def __init__(self, width, layers, hidden):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(width, hidden) for _ in range(layers)])


class _PatchConv(nn.Module):
    def __init__(self, width, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(width, 3, patch, patch) * (3 * patch * patch) ** -0.5)  # bias=False in open_clip


class _Visual(nn.Module):
    """open_clip.transformer.VisionTransformer parameter names (conv1, class_embedding, positional_embedding, ln_pre, transformer.resblocks,
    ln_post, proj)."""

    def __init__(self, g):
        super().__init__()
        w, n_tok = g["width"], (g["image"] // g["patch"]) ** 2 + 1
        self.conv1 = _PatchConv(w, g["patch"])
        self.class_embedding = nn.Parameter(torch.randn(w) * w ** -0.5)
        self.positional_embedding = nn.Parameter(torch.randn(n_tok, w) * w ** -0.5)
        self.ln_pre = LayerNorm(w)
        self.transformer = _Transformer(w, g["layers"], g["mlp"])
        self.ln_post = LayerNorm(w)
        self.proj = nn.Parameter(torch.randn(w, g["embed"]) * w ** -0.5)
        self.output_tokens = False


class _OpenClipModel(nn.Module):
    def __init__(self, g):
        super().__init__()
        self.visual = _Visual(g)


class FrozenOpenCLIPImageEmbedder(AbstractEmbModel, Packable):
    """Uses the OpenCLIP vision transformer encoder for images (modules.py:251-399). `arch` may also be a geometry dict (tests)."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, antialias=True, ucg_rate=0.0,
                 unsqueeze_dim=False, repeat_to_max_len=False, num_image_crops=0, output_tokens=False, init_device=None):
        super().__init__()
        for name, val in (("ucg_rate", ucg_rate), ("repeat_to_max_len", repeat_to_max_len), ("num_image_crops", num_image_crops),
                          ("output_tokens", output_tokens)):
            if val:
                raise NotImplementedError(f"FrozenOpenCLIPImageEmbedder: {name}={val!r} is not used by Vista's inference configuration")
        self.geometry = dict(arch) if isinstance(arch, dict) else OPENCLIP_VISION_GEOMETRY[arch]
        g = self.geometry
        if g["width"] % g["heads"] or g["width"] // g["heads"] not in (64, 80, 128) or g["width"] % 64 or g["mlp"] % 64:
            raise NotImplementedError("vision tower geometry: head dim must be 64 / 80 / 128 and width, MLP multiples of 64")
        self.model = _OpenClipModel(g)  # weights: `conditioner.embedders.0.open_clip.model.visual.*` of vista.safetensors (no download here)
        from ..attention import _invalidate_after_load
        self.register_load_state_dict_post_hook(_invalidate_after_load)   # (also when this embedder is loaded on its own)
        self.max_crops, self.pad_to_max_len, self.repeat_to_max_len = 0, False, False
        self.device, self.max_length, self.antialias = device, max_length, antialias
        self.register_buffer("mean", torch.tensor(ops.CLIP_MEAN), persistent=False)
        self.register_buffer("std", torch.tensor(ops.CLIP_STD), persistent=False)
        self.ucg_rate, self.unsqueeze_dim, self.output_tokens = ucg_rate, unsqueeze_dim, False
        self.stored_batch = None
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p_ in self.parameters():
            p_.requires_grad = False

    # ---- packing: LayerNorms folded into the GEMMs that consume them; class / position embeddings as the patch GEMM's residual ----
    def _pack(self, dev):
        v, g = self.model.visual, self.geometry
        k_raw = 3 * g["patch"] ** 2
        wconv = torch.zeros(g["width"], ops.ceil_to(k_raw, 64))
        wconv[:, :k_raw] = v.conv1.weight.detach().float().reshape(g["width"], k_raw).cpu()
        table = v.positional_embedding.detach().float().clone()
        table[0] += v.class_embedding.detach().float()   # row 0 of every image = class token (its patch row is zero) + position 0
        blocks = []
        for b in v.transformer.resblocks:
            blocks.append({"qkv": ops.pack_linear(b.attn.in_proj_weight, b.attn.in_proj_bias, dev, ln=b.ln_1),
                           "out": ops.pack_linear(b.attn.out_proj.weight, b.attn.out_proj.bias, dev),
                           "fc1": ops.pack_linear(b.mlp.c_fc.weight, b.mlp.c_fc.bias, dev, ln=b.ln_2),
                           "fc2": ops.pack_linear(b.mlp.c_proj.weight, b.mlp.c_proj.bias, dev)})
        return {"conv1": ops.pack_linear(wconv, None, dev), "pos": table.to(device=dev, dtype=torch.bfloat16), "pos_rep": {},
                "blocks": blocks, "proj": ops.pack_linear(v.proj.detach().float().t().contiguous(), None, dev)}

    def preprocess(self, x):
        """(n, 3, H, W) in [-1, 1] -> the patch-embedding GEMM's A operand (resize + normalise + im2col in one kernel)."""
        g = self.geometry
        return ops.clip_preprocess_patches(x.float(), out_hw=g["image"], patch=g["patch"], antialias=self.antialias)

    def _visual(self, img):
        g, v = self.geometry, self.model.visual
        pk = self.packed()
        n = img.shape[0]
        n_tok, width, heads = (g["image"] // g["patch"]) ** 2 + 1, g["width"], g["heads"]
        if n not in pk["pos_rep"]:
            pk["pos_rep"] = {n: pk["pos"].repeat(n, 1)}  # (n*n_tok, width) bf16, kept for the batch size in use
        t = ops.linear(self.preprocess(img), pk["conv1"], res1=pk["pos_rep"][n])
        x = ops.layernorm(t, v.ln_pre.weight, v.ln_pre.bias, v.ln_pre.eps)   # the residual stream starts normalised (ln_pre)
        st = ops.rowstats(x)
        for blk in pk["blocks"]:
            qkv = ops.linear(x, blk["qkv"], ln=st)                              # ln_1 folded
            att = ops.attn_small(qkv, n, heads, n_tok, width // heads)
            x, st = ops.linear(att, blk["out"], res1=x, emit_stats=True)
            h = ops.linear(x, blk["fc1"], ln=st, act="gelu")                    # ln_2 folded, nn.GELU in the epilogue
            x, st = ops.linear(h, blk["fc2"], res1=x, emit_stats=True)
        cls = x.view(n, n_tok, width)[:, 0].contiguous()
        pooled = ops.layernorm(cls, v.ln_post.weight, v.ln_post.bias, v.ln_post.eps)
        return ops.linear(pooled, pk["proj"], out_f32=True)[:, :g["embed"]]

    @ops.bf16_storage   # (the conditioner stores bf16 in every process: ops.storage)
    def encode_with_vision_transformer(self, img):
        if img.dim() != 4:
            raise NotImplementedError("image crops (5-D input) are not used by Vista's inference configuration")
        if img.device.type != "cuda":
            raise ops._lib.VistaHipError("FrozenOpenCLIPImageEmbedder: images must be on the MI355X (vista_amd has no CPU path)")
        # sample_utils.get_batch repeats ONE conditioning frame num_frames times: equal images have equal embeddings (per-image arithmetic;
        # only the GEMM launcher's M-dependent tile / split-K choice could move an fp32 rounding), so the tower runs once when they all match
        n = img.shape[0]
        if n > 1 and bool((img[1:] == img[:1]).flatten(1).all()):
            return self._visual(img[:1]).expand(n, -1).contiguous()
        return self._visual(img)

    @torch.no_grad()
    def forward(self, image, no_dropout=False):
        z = self.encode_with_vision_transformer(image).to(image.dtype)
        return z[:, None] if self.unsqueeze_dim else z

    def encode(self, text):
        return self(text)


class FrozenOpenCLIPImagePredictionEmbedder(AbstractEmbModel):
    def __init__(self, open_clip_embedding_config, n_cond_frames, n_copies):
        super().__init__()
        self.n_cond_frames, self.n_copies = n_cond_frames, n_copies
        self.open_clip = instantiate_from_config(open_clip_embedding_config)

    def forward(self, vid):
        z = self.open_clip(vid)                                            # ((b t), d)
        z = z.view(-1, self.n_cond_frames, z.shape[-1])                    # (b, t, d)
        return z.repeat_interleave(self.n_copies, dim=0)                   # "b t d -> (b s) t d"


class ConcatTimestepEmbedderND(AbstractEmbModel):
    """Embeds each dimension independently (fp32 cos|sin of openaimodel.Timestep, max_period 1e4) and concatenates them."""

    def __init__(self, outdim, num_features=None, add_sequence_dim=False):
        super().__init__()
        self.outdim, self.num_features, self.add_sequence_dim = outdim, num_features, add_sequence_dim

    def forward(self, x):
        if x.ndim == 1:
            x = x[:, None]
        if x.ndim != 2:
            raise ValueError("ConcatTimestepEmbedderND expects (b,) or (b, d)")
        b, dims = x.shape
        assert dims == self.num_features or self.num_features is None
        emb = ops.timestep_embedding(x.reshape(-1).float(), self.outdim, out_f32=True).view(b, dims * self.outdim)
        return emb[:, None] if self.add_sequence_dim else emb


class VideoPredictionEmbedderWithEncoder(AbstractEmbModel):
    """modules.py:428-503: first-stage encoding of the conditioning frame(s) for the UNet's "concat" input."""

    def __init__(self, n_cond_frames, n_copies, encoder_config, sigma_sampler_config=None, sigma_cond_config=None, is_ae=False,
                 scale_factor=1.0, disable_encoder_autocast=False, en_and_decode_n_samples_a_time=None):
        super().__init__()
        if sigma_sampler_config is not None or sigma_cond_config is not None:
            raise NotImplementedError("sigma_sampler / sigma_cond are training-time options")
        self.n_cond_frames, self.n_copies = n_cond_frames, n_copies
        self.encoder = instantiate_from_config(encoder_config)
        self.sigma_sampler = self.sigma_cond = None
        self.is_ae, self.scale_factor = is_ae, scale_factor
        self.disable_encoder_autocast = disable_encoder_autocast
        self.en_and_decode_n_samples_a_time = en_and_decode_n_samples_a_time
        self.skip_encode = False

    @torch.no_grad()
    def forward(self, vid):
        if self.skip_encode:
            return vid
        full = vid.shape[0]
        if full > 1 and bool((vid[1:] == vid[:1]).flatten(1).all()):  # get_batch's N copies of one frame: encode it once (see the image tower)
            vid = vid[:1]
        n_samples = default(self.en_and_decode_n_samples_a_time, vid.shape[0])
        outs = []
        for i in range(0, vid.shape[0], n_samples):
            chunk = vid[i:i + n_samples]
            outs.append(self.encoder.encode(chunk, scale=self.scale_factor) if self.is_ae else self.encoder(chunk) * self.scale_factor)
        z = torch.cat(outs, dim=0)                                          # ((b t), c, h, w), already times scale_factor
        if z.shape[0] != full:
            z = z.expand(full, -1, -1, -1).contiguous()
        bt, c, h, w = z.shape
        z = z.view(bt // self.n_cond_frames, self.n_cond_frames * c, h, w)   # "(b t) c h w -> b () (t c) h w"
        return z.repeat_interleave(self.n_copies, dim=0)                    # "b 1 c h w -> (b t) c h w"
