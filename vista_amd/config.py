"""The shipped network configuration and config plumbing. Single source: configs/inference/vista_mi355x.yaml (the overlay of the
reference's configs/inference/vista.yaml:10-40 that swaps in this package's classes)."""
import copy
import os

CONFIG_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "inference", "vista_mi355x.yaml")

# What configs/inference/vista_mi355x.yaml (= the reference's vista.yaml:20-40) holds under model.params.network_config.params. Used when
# the YAML or PyYAML is not there (a vendored copy of the package without the sibling configs/ directory); tests/test_config_cpu.py
# holds the two equal.
_FALLBACK_UNET_KWARGS = {
    "use_checkpoint": False, "in_channels": 8, "out_channels": 4, "model_channels": 320, "attention_resolutions": [4, 2, 1],
    "num_res_blocks": 2, "channel_mult": [1, 2, 4, 4], "num_head_channels": 64, "use_linear_in_transformer": True,
    "transformer_depth": 1, "context_dim": 1024, "spatial_transformer_attn_type": "softmax-xformers", "extra_ff_mix_layer": True,
    "use_spatial_context": True, "merge_strategy": "learned_with_images", "video_kernel_size": [3, 1, 1], "add_lora": False,
    "action_control": True, "adm_in_channels": 768, "num_classes": "sequential",
}


def load_config(path=CONFIG_PATH):
    import yaml  # lazy: importing vista_amd.config must not require PyYAML
    with open(path) as f:
        return yaml.safe_load(f)


def overlay(base, over):
    """Recursive dict merge (what OmegaConf.merge does for mappings): `over` wins, lists are replaced."""
    out = copy.deepcopy(base)
    for k, v in over.items():
        out[k] = overlay(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else copy.deepcopy(v)
    return out


_UNET_KWARGS = None


def vista_unet_kwargs():
    """network_config.params of the shipped configuration (read once, on first use)."""
    global _UNET_KWARGS
    if _UNET_KWARGS is None:
        try:
            _UNET_KWARGS = load_config()["model"]["params"]["network_config"]["params"]
        except (ImportError, OSError):
            _UNET_KWARGS = copy.deepcopy(_FALLBACK_UNET_KWARGS)
    return _UNET_KWARGS


def __getattr__(name):  # VISTA_UNET_KWARGS stays importable, resolved lazily
    if name == "VISTA_UNET_KWARGS":
        return vista_unet_kwargs()
    raise AttributeError(name)


def unet_kwargs(model_channels=320, **over):
    kw = copy.deepcopy(vista_unet_kwargs())
    kw["model_channels"] = model_channels
    kw.update(over)
    return kw
