"""Config / tensor helpers of the hot path. Same names and behaviour as the reference's helpers (vwm/util.py:63-75,131-135,154-188) so
that code written against them keeps working; `instantiate_from_config` additionally maps the reference's `vwm.modules...` /
`vwm.models...` target strings onto this package, which is how the hard-coded sampler / guider / discretizer dicts of the reference's
sample_utils.py:148-229 resolve to the MI355X classes without being edited."""
import functools
import importlib

_TARGET_PREFIXES = (("vwm.modules.", "vista_amd.modules."), ("vwm.models.", "vista_amd.models."), ("vwm.util", "vista_amd.util"))
_PLACEHOLDER_CONFIGS = ("__is_first_stage__", "__is_unconditional__")  # config strings that stand for "no object"


def default(val, d):
    """val unless it is None; otherwise d, called first if it is a plain callable (not a class)."""
    if val is None:
        return d() if (callable(d) and not isinstance(d, type)) else d
    return val


def map_target(target):
    """'vwm.modules.x.Y' -> 'vista_amd.modules.x.Y' (anything else is returned unchanged)."""
    for theirs, ours in _TARGET_PREFIXES:
        if target.startswith(theirs):
            return ours + target[len(theirs):]
    return target


def get_obj_from_str(string, reload=False, invalidate_cache=True):
    module_name, _, attr = map_target(string).rpartition(".")
    if invalidate_cache:
        importlib.invalidate_caches()
    module = importlib.import_module(module_name)
    if reload:
        module = importlib.reload(module)
    return getattr(module, attr)


def instantiate_from_config(config):
    """{'target': dotted.path, 'params': {...}} -> dotted.path(**params); the two placeholder strings give None."""
    if "target" in config:
        return get_obj_from_str(config["target"])(**config.get("params", {}))
    if config in _PLACEHOLDER_CONFIGS:
        return None
    raise KeyError("Expected key `target` to instantiate")


def append_zero(x):
    """(n,) -> (n + 1,) with a trailing 0 of x's dtype / device (the sigma = 0 the sampler steps onto)."""
    out = x.new_zeros(x.shape[0] + 1)
    out[:-1] = x
    return out


def append_dims(x, target_dims):
    """x with trailing singleton axes up to `target_dims` dimensions."""
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"Input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x.reshape(*x.shape, *([1] * extra))


def repeat_as_img_seq(x, num_frames):
    """Per-video value -> per-frame value, 'b 1 ... -> (b t) ...' (vwm/util.py:63-75); lists repeat element-wise, None stays None."""
    if x is None:
        return None
    if isinstance(x, list):
        return [item for item in x for _ in range(num_frames)]
    return x.repeat_interleave(num_frames, dim=0)


def partialclass(cls, *args, **kwargs):
    """A subclass of `cls` whose __init__ has the given arguments pre-bound (vwm/util.py partialclass)."""
    return type("NewCls", (cls,), {"__init__": functools.partialmethod(cls.__init__, *args, **kwargs)})
