"""CPU oracle of the conditioner's OpenCLIP image tower. TEST INFRASTRUCTURE ONLY: imported by tests/, never by vista_amd/.

The reference delegates this path to two third-party packages that are ABSENT from /root/reference and from this image
(vwm/modules/encoders/modules.py:5-8, requirements.txt:12,18): `open_clip` (open-clip-torch >= 2.20.0: `create_model_and_transforms("ViT-H-14")`
-> VisionTransformer) and `kornia` == 0.6.9 (`kornia.geometry.resize(..., antialias=True)`, `kornia.enhance.normalize`). So:

* the transformer is checked against `transformers.CLIPVisionModelWithProjection` (installed here) -- the same published architecture
  (class token + learned positions, ln_pre, pre-LN blocks with nn.MultiheadAttention-style attention and a c_fc / GELU / c_proj MLP, ln_post of
  the class token, bias-free projection) -- fed the open_clip-named state dict through the standard name map below. hidden_act = "gelu"
  (exact erf): open_clip's ViT-H-14 config has no quick_gelu.
* the preprocessing is a RESTATEMENT of kornia 0.6.9's published algorithm (kornia/geometry/transform/affwarp.py `resize`: when downscaling and
  antialias, gaussian_blur2d with sigma = max((factor - 1) / 2, 0.001) per axis and kernel size int(max(4 sigma, 3)) made odd, border "reflect";
  then F.interpolate(mode="bicubic", align_corners=True)), then (x + 1) / 2 and (x - mean) / std. kornia cannot be imported to pin it:
  **parity of the resize step is unpinned** (stated in DESIGN.md); the transformer itself is pinned to the HF implementation.
"""
import math

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _gauss1d(ks, sigma):
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2) / (2.0 * sigma * sigma))
    return g / g.sum()


def kornia_resize_bicubic_antialias(x, size, antialias=True):
    """kornia 0.6.9 geometry.transform.resize(x, size, interpolation="bicubic", align_corners=True, antialias=antialias) on (n, c, H, W)."""
    H, W = x.shape[-2:]
    fy, fx = H / size[0], W / size[1]
    if antialias and max(fy, fx) > 1:
        sig = (max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * s, 3)) for s in sig]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        ky, kx = _gauss1d(ks[0], sig[0]), _gauss1d(ks[1], sig[1])
        c = x.shape[1]
        xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        k2 = (ky[:, None] * kx[None, :])[None, None].repeat(c, 1, 1, 1)
        x = F.conv2d(xp, k2, groups=c)
    return F.interpolate(x, size=size, mode="bicubic", align_corners=True)


def preprocess(x, size=224, antialias=True):
    """FrozenOpenCLIPImageEmbedder.preprocess (modules.py:304-315): images in [-1, 1] -> CLIP-normalised size x size."""
    x = kornia_resize_bicubic_antialias(x.float(), (size, size), antialias)
    x = (x + 1.0) / 2.0
    mean, std = torch.tensor(CLIP_MEAN)[None, :, None, None], torch.tensor(CLIP_STD)[None, :, None, None]
    return (x - mean) / std


def hf_vision_model(sd, g, prefix="model.visual."):
    """transformers.CLIPVisionModelWithProjection carrying the open_clip-named weights `sd[prefix + ...]` (geometry dict g as in
    vista_amd.modules.encoders.modules.OPENCLIP_VISION_GEOMETRY)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    cfg = CLIPVisionConfig(hidden_size=g["width"], intermediate_size=g["mlp"], num_hidden_layers=g["layers"], num_attention_heads=g["heads"],
                           image_size=g["image"], patch_size=g["patch"], projection_dim=g["embed"], hidden_act="gelu", layer_norm_eps=1e-5,
                           attention_dropout=0.0)
    m = CLIPVisionModelWithProjection(cfg).eval()
    w = g["width"]
    t = {k[len(prefix):]: v.float() for k, v in sd.items() if k.startswith(prefix)}
    hf = {"vision_model.embeddings.class_embedding": t["class_embedding"],
          "vision_model.embeddings.patch_embedding.weight": t["conv1.weight"],
          "vision_model.embeddings.position_embedding.weight": t["positional_embedding"],
          "vision_model.pre_layrnorm.weight": t["ln_pre.weight"], "vision_model.pre_layrnorm.bias": t["ln_pre.bias"],
          "vision_model.post_layernorm.weight": t["ln_post.weight"], "vision_model.post_layernorm.bias": t["ln_post.bias"],
          "visual_projection.weight": t["proj"].t().contiguous()}
    for i in range(g["layers"]):
        s, d = f"transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
        for j, name in enumerate(("q_proj", "k_proj", "v_proj")):  # nn.MultiheadAttention packs [q; k; v] along dim 0
            hf[d + f"self_attn.{name}.weight"] = t[s + "attn.in_proj_weight"][j * w:(j + 1) * w]
            hf[d + f"self_attn.{name}.bias"] = t[s + "attn.in_proj_bias"][j * w:(j + 1) * w]
        hf[d + "self_attn.out_proj.weight"], hf[d + "self_attn.out_proj.bias"] = t[s + "attn.out_proj.weight"], t[s + "attn.out_proj.bias"]
        hf[d + "layer_norm1.weight"], hf[d + "layer_norm1.bias"] = t[s + "ln_1.weight"], t[s + "ln_1.bias"]
        hf[d + "layer_norm2.weight"], hf[d + "layer_norm2.bias"] = t[s + "ln_2.weight"], t[s + "ln_2.bias"]
        hf[d + "mlp.fc1.weight"], hf[d + "mlp.fc1.bias"] = t[s + "mlp.c_fc.weight"], t[s + "mlp.c_fc.bias"]
        hf[d + "mlp.fc2.weight"], hf[d + "mlp.fc2.bias"] = t[s + "mlp.c_proj.weight"], t[s + "mlp.c_proj.bias"]
    missing, unexpected = m.load_state_dict(hf, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m


@torch.no_grad()
def image_embed(sd, g, images, antialias=True, prefix="model.visual."):
    """FrozenOpenCLIPImageEmbedder.forward for the inference configuration: (n, 3, H, W) in [-1, 1] -> (n, embed) fp32."""
    m = hf_vision_model(sd, g, prefix)
    return m(pixel_values=preprocess(images, g["image"], antialias)).image_embeds.float()


def restated_visual(sd, g, pixel_values, prefix="model.visual."):
    """The same tower written out in plain torch (independent of transformers' module code): used to cross-check the name map."""
    t = {k[len(prefix):]: v.float() for k, v in sd.items() if k.startswith(prefix)}
    n, w, heads = pixel_values.shape[0], g["width"], g["heads"]
    x = F.conv2d(pixel_values, t["conv1.weight"], stride=g["patch"]).flatten(2).transpose(1, 2)             # (n, 256, w)
    x = torch.cat([t["class_embedding"][None, None].expand(n, 1, w), x], 1) + t["positional_embedding"][None]
    x = F.layer_norm(x, (w,), t["ln_pre.weight"], t["ln_pre.bias"], 1e-5)
    for i in range(g["layers"]):
        s = f"transformer.resblocks.{i}."
        y = F.layer_norm(x, (w,), t[s + "ln_1.weight"], t[s + "ln_1.bias"], 1e-5)
        q, k, v = (y @ t[s + "attn.in_proj_weight"].t() + t[s + "attn.in_proj_bias"]).chunk(3, dim=-1)
        q, k, v = (a.view(n, -1, heads, w // heads).transpose(1, 2) for a in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(w // heads), dim=-1) @ v
        x = x + a.transpose(1, 2).reshape(n, -1, w) @ t[s + "attn.out_proj.weight"].t() + t[s + "attn.out_proj.bias"]
        y = F.layer_norm(x, (w,), t[s + "ln_2.weight"], t[s + "ln_2.bias"], 1e-5)
        y = F.gelu(y @ t[s + "mlp.c_fc.weight"].t() + t[s + "mlp.c_fc.bias"])
        x = x + y @ t[s + "mlp.c_proj.weight"].t() + t[s + "mlp.c_proj.bias"]
    pooled = F.layer_norm(x[:, 0], (w,), t["ln_post.weight"], t["ln_post.bias"], 1e-5)
    return pooled @ t["proj"]
