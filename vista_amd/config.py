"""The shipped network configuration and config plumbing. Single source: configs/inference/vista_mi355x.yaml (the overlay of the
reference's configs/inference/vista.yaml:10-40 that swaps in this package's classes)."""
import copy
import os

import yaml

CONFIG_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "inference", "vista_mi355x.yaml")


def load_config(path=CONFIG_PATH):
    with open(path) as f:
        return yaml.safe_load(f)


def overlay(base, over):
    """Recursive dict merge (what OmegaConf.merge does for mappings): `over` wins, lists are replaced."""
    out = copy.deepcopy(base)
    for k, v in over.items():
        out[k] = overlay(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else copy.deepcopy(v)
    return out


VISTA_UNET_KWARGS = load_config()["model"]["params"]["network_config"]["params"]


def unet_kwargs(model_channels=320, **over):
    kw = copy.deepcopy(VISTA_UNET_KWARGS)
    kw["model_channels"] = model_channels
    kw.update(over)
    return kw
