"""Import shim for the REAL reference (OpenDriveLab/Vista at /root/reference) on CPU, fp32.

TEST INFRASTRUCTURE ONLY. Used in the build container (where /root/reference exists) to (a) validate the CPU
restatement in oracle/vista_oracle.py and (b) generate the golden vectors committed under tests/golden/ (see
oracle/make_golden.py). Nothing under vista_amd/ imports this, and nothing that runs on the GPU box needs it.

Recipe: the reference's package __init__ files pull in pytorch_lightning / kornia / open_clip,
which are absent here, so `vwm` and `vwm.modules` are pre-registered as empty namespace packages pointing at the
reference directories (their sub-packages have no __init__ of their own); `omegaconf` is stubbed with the two
names the hot path touches; `xformers.ops.memory_efficient_attention` is mapped to
F.scaled_dot_product_attention (same semantics for 3-D (B*h, N, d) inputs: softmax(q k^T / sqrt(d)) v).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("VISTA_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "vwm", "modules", "diffusionmodules"))


def install():
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    if "vwm" in sys.modules and getattr(sys.modules["vwm"], "_vista_shim", False):
        return
    import torch.nn.functional as F

    vwm = types.ModuleType("vwm")
    vwm.__path__ = [os.path.join(REF_ROOT, "vwm")]
    vwm._vista_shim = True
    sys.modules["vwm"] = vwm
    mods = types.ModuleType("vwm.modules")
    mods.__path__ = [os.path.join(REF_ROOT, "vwm", "modules")]
    sys.modules["vwm.modules"] = mods
    vwm.modules = mods

    oc = types.ModuleType("omegaconf")

    class ListConfig(list):
        pass

    class OmegaConf(dict):
        pass

    oc.ListConfig, oc.OmegaConf = ListConfig, OmegaConf
    sys.modules.setdefault("omegaconf", oc)

    xf = types.ModuleType("xformers")
    xops = types.ModuleType("xformers.ops")

    def memory_efficient_attention(q, k, v, attn_bias=None, op=None):
        assert attn_bias is None
        return F.scaled_dot_product_attention(q, k, v)

    class LowerTriangularMask:  # referenced only for causal=True, never on the hot path
        pass

    xops.memory_efficient_attention = memory_efficient_attention
    xops.LowerTriangularMask = LowerTriangularMask
    xf.ops = xops
    sys.modules.setdefault("xformers", xf)
    sys.modules.setdefault("xformers.ops", xops)


def ref_classes():
    """Returns a dict of the reference hot-path classes."""
    install()
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        from vwm.modules.diffusionmodules.video_model import VideoUNet
        from vwm.modules.diffusionmodules.sampling import EulerEDMSampler
        from vwm.modules.diffusionmodules.denoiser import Denoiser
        from vwm.modules.diffusionmodules.wrappers import OpenAIWrapper
        from vwm.modules.diffusionmodules import guiders, discretizer, denoiser_scaling, util
    return dict(VideoUNet=VideoUNet, EulerEDMSampler=EulerEDMSampler, Denoiser=Denoiser, OpenAIWrapper=OpenAIWrapper,
                guiders=guiders, discretizer=discretizer, denoiser_scaling=denoiser_scaling, util=util)


# the shipped network config (configs/inference/vista.yaml:20-40)
VISTA_UNET_KWARGS = dict(
    adm_in_channels=768, num_classes="sequential", use_checkpoint=False, in_channels=8, out_channels=4,
    model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
    num_head_channels=64, use_linear_in_transformer=True, transformer_depth=1, context_dim=1024,
    spatial_transformer_attn_type="softmax-xformers", extra_ff_mix_layer=True, use_spatial_context=True,
    merge_strategy="learned_with_images", video_kernel_size=[3, 1, 1], add_lora=False, action_control=True)


def unet_kwargs(model_channels=320, **over):
    kw = dict(VISTA_UNET_KWARGS)
    kw["model_channels"] = model_channels
    kw.update(over)
    return kw


def build_ref_unet(**kw):
    import io
    import contextlib
    c = ref_classes()
    with contextlib.redirect_stdout(io.StringIO()):
        net = c["VideoUNet"](**kw)
    return net.eval()
