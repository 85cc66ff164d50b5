"""Config / tensor helpers on the hot path (reference vwm/util.py:63-75,131-135,154-188), same names and behaviour.
`instantiate_from_config` additionally maps the reference's `vwm.modules...` target strings onto this package, so the
hard-coded sampler / guider / discretizer dicts of the reference's sample_utils.py:148-229 resolve to the MI355X
classes without editing them."""
import importlib

import torch

_TARGET_PREFIXES = (("vwm.modules.", "vista_amd.modules."), ("vwm.models.", "vista_amd.models."), ("vwm.util", "vista_amd.util"))


def default(val, d):
    if val is not None:
        return val
    return d() if callable(d) and not isinstance(d, type) else d


def map_target(target):
    for old, new in _TARGET_PREFIXES:
        if target.startswith(old):
            return new + target[len(old):]
    return target


def get_obj_from_str(string, reload=False, invalidate_cache=True):
    module, cls = map_target(string).rsplit(".", 1)
    if invalidate_cache:
        importlib.invalidate_caches()
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        elif config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def append_zero(x):
    return torch.cat((x, x.new_zeros([1])))


def append_dims(x, target_dims):
    """Appends dimensions to the end of a tensor until it has target_dims dimensions."""
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"Input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def repeat_as_img_seq(x, num_frames):
    """'b 1 ... -> (b t) ...' (vwm/util.py:63-75)"""
    if x is None:
        return None
    if isinstance(x, list):
        new_x = list()
        for item_x in x:
            new_x += [item_x] * num_frames
        return new_x
    return x.repeat_interleave(num_frames, dim=0)


def partialclass(cls, *args, **kwargs):
    """vwm/util.py partialclass: a subclass whose __init__ has the given arguments pre-bound."""
    import functools

    class NewCls(cls):
        __init__ = functools.partialmethod(cls.__init__, *args, **kwargs)

    return NewCls
