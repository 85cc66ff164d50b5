"""Checkpoint ingest: Vista's on-disk formats -> the state dicts vista_amd modules load.

* `load_checkpoint(path)`            - `vista.safetensors` or a Lightning `.ckpt` (sample_utils.py:54-71)
* `convert_training_checkpoint(sd)`  - what `bin_to_st.py` does to a DeepSpeed `pytorch_model.bin`: merge LoRA adapters into
                                       the frozen projections (:10-31), strip the `_forward_module.` prefix (:33-36), replace
                                       weights by their EMA copies (:38-47)
* `split_by_component(sd)`           - `model.diffusion_model.*` -> VideoUNet keys, `first_stage_model.decoder.*` -> VideoDecoder keys
* `load_into(...)`                   - `load_state_dict(strict=False)` + the reference's missing / unexpected report (sample_utils.py:73-77)

Tensors stay torch CPU tensors here; the bf16 packing for the HIP GEMMs happens lazily in each module (`Packable.packed()`),
which `load_state_dict` invalidates.
"""
import torch

UNET_PREFIX = "model.diffusion_model."
DECODER_PREFIX = "first_stage_model.decoder."
_LORA = (("q_adapter_down", "q_adapter_up", "to_q"), ("k_adapter_down", "k_adapter_up", "to_k"), ("v_adapter_down", "v_adapter_up", "to_v"))


def load_checkpoint(path, trust_pickle=False):
    """-> flat {name: tensor}. `.safetensors` via safetensors.torch.load_file, `.ckpt` via torch.load()['state_dict'].
    A `.ckpt` is a pickle: it is read with weights_only=True (tensors and plain containers only -- enough for a Lightning checkpoint's
    state_dict); `trust_pickle=True` is the explicit opt-in to the reference's unrestricted torch.load (sample_utils.py:60-64), which
    executes whatever the file contains."""
    if path.endswith("safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    if path.endswith("ckpt"):
        blob = torch.load(path, map_location="cpu", weights_only=not trust_pickle)
        return blob["state_dict"]
    raise NotImplementedError("Please convert the checkpoint to safetensors first")


def merge_lora(sd):
    """W <- W + up @ down for every `{q,k,v,out}_adapter_{down,up}` pair; the adapter tensors are dropped. In place; returns sd.
    The EMA copies name the output projection `to_out0` instead of `to_out.0` (LitEma strips dots)."""
    for k in list(sd.keys()):
        if "adapter_down" not in k:
            continue
        for down, up, frozen in _LORA:
            if down in k:
                up_k, w_k = k.replace(down, up), k.replace(down, frozen)
                break
        else:
            up_k = k.replace("out_adapter_down", "out_adapter_up")
            w_k = k.replace("out_adapter_down", "to_out0" if "model_ema" in k else "to_out.0")
        delta = sd[up_k] @ sd[k]
        del sd[k], sd[up_k]
        sd[w_k] = sd[w_k] + delta
    return sd


def convert_training_checkpoint(sd):
    """DeepSpeed `pytorch_model.bin` contents -> the dict `bin_to_st.py` saves as ckpts/vista.safetensors."""
    sd = merge_lora(dict(sd))
    out = {}
    for k, v in sd.items():  # keep only module weights; drop the engine prefix and the EMA bookkeeping scalars
        if "_forward_module" in k and "decay" not in k and "num_updates" not in k:
            out[k.replace("_forward_module.", "")] = v
    # EMA shadow weights are stored as `model_ema.<param name without dots>` for params under `model.`
    flat = {kk[6:].replace(".", ""): kk for kk in out if "model_ema" not in kk}
    for k in [k for k in out if "model_ema" in k]:
        target = flat.get(k[10:])
        if target is None:
            raise KeyError(f"EMA tensor {k} has no live counterpart")
        out[target] = out.pop(k)
    return out


def split_by_component(sd):
    """-> {"unet": {...}, "decoder": {...}, "rest": {...}} with the component prefixes stripped."""
    parts = {"unet": {}, "decoder": {}, "rest": {}}
    for k, v in sd.items():
        if k.startswith(UNET_PREFIX):
            parts["unet"][k[len(UNET_PREFIX):]] = v
        elif k.startswith(DECODER_PREFIX):
            parts["decoder"][k[len(DECODER_PREFIX):]] = v
        else:
            parts["rest"][k] = v
    return parts


def load_into(sd, unet=None, decoder=None, verbose=True):
    """Loads the matching slices of a full Vista state dict into a vista_amd VideoUNet / VideoDecoder, `strict=False` like the
    reference, and returns {"unet": (missing, unexpected), "decoder": (...)} -- names drifting silently is how one gets
    "a sequence of blur" (docs/SAMPLING.md:33), so callers should assert both lists are empty."""
    parts = split_by_component(sd)
    report = {}
    for name, mod in (("unet", unet), ("decoder", decoder)):
        if mod is None:
            continue
        res = mod.load_state_dict(parts[name], strict=False)
        missing, unexpected = list(res.missing_keys), list(res.unexpected_keys)
        if verbose and missing:
            print(f"Missing keys: {missing}")
        if verbose and unexpected:
            print(f"Unexpected keys: {unexpected}")
        report[name] = (missing, unexpected)
    return report
