"""Golden for vista_amd.checkpoint.convert_training_checkpoint: runs the reference's bin_to_st.py AS IS (exec of the file's
source with torch.load / save_file / os.makedirs intercepted) on a seeded synthetic `pytorch_model.bin` dict and stores the
resulting names with per-tensor checksums in tests/golden/ckpt_convert.json.      python oracle/make_golden_ckpt.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vista_amd import synth  # noqa: E402

REF = os.environ.get("VISTA_REFERENCE", "/root/reference")


def synthetic_training_dict(seed=0):
    """Names as a DeepSpeed dump of the Lightning engine has them: `_forward_module.` prefix, LoRA adapters beside frozen
    projections, LitEma shadows (`model_ema.<dotless name>`) for trainable tensors, EMA bookkeeping scalars, one stray key."""
    p = "_forward_module.model.diffusion_model.input_blocks.1.1.transformer_blocks.0.attn1."
    e = "_forward_module.model_ema.diffusion_modelinput_blocks11transformer_blocks0attn1"
    shapes = {}
    for proj, n in (("to_q", "q"), ("to_k", "k"), ("to_v", "v")):
        shapes[p + proj + ".weight"] = (64, 64)
        shapes[p + f"{n}_adapter_down.weight"] = (8, 64)
        shapes[p + f"{n}_adapter_up.weight"] = (64, 8)
        shapes[e + proj + "weight"] = (64, 64)
        shapes[e + f"{n}_adapter_downweight"] = (8, 64)
        shapes[e + f"{n}_adapter_upweight"] = (64, 8)
    shapes[p + "to_out.0.weight"] = (64, 64)
    shapes[p + "to_out.0.bias"] = (64,)
    shapes[p + "out_adapter_down.weight"] = (8, 64)
    shapes[p + "out_adapter_up.weight"] = (64, 8)
    shapes[e + "to_out0weight"] = (64, 64)
    shapes[e + "to_out0bias"] = (64,)
    shapes[e + "out_adapter_downweight"] = (8, 64)
    shapes[e + "out_adapter_upweight"] = (64, 8)
    shapes["_forward_module.model.diffusion_model.out.2.weight"] = (4, 64, 3, 3)          # no EMA shadow: stays as is
    shapes["_forward_module.first_stage_model.decoder.conv_in.weight"] = (64, 4, 3, 3)
    shapes["_forward_module.model_ema.decay"] = ()
    shapes["_forward_module.model_ema.num_updates"] = ()
    shapes["optimizer_stray_key"] = (3,)
    return {k: synth.seeded_tensor(k, s, seed) if len(s) else torch.tensor(0.5) for k, s in shapes.items()}


def checksum(t):
    t = t.double()
    return [float(t.sum()), float(t.abs().sum()), list(t.shape)]


def main():
    src = open(os.path.join(REF, "bin_to_st.py")).read()
    captured = {}
    import safetensors.torch as st
    real_load, real_save, real_mk = torch.load, st.save_file, os.makedirs
    torch.load = lambda *a, **k: synthetic_training_dict()
    st.save_file = lambda d, path: captured.update(d)
    os.makedirs = lambda *a, **k: None
    try:
        exec(compile(src, "bin_to_st.py", "exec"), {"__name__": "__main__"})
    finally:
        torch.load, st.save_file, os.makedirs = real_load, real_save, real_mk
    out = {k: checksum(v) for k, v in sorted(captured.items())}
    with open(os.path.join(ROOT, "tests", "golden", "ckpt_convert.json"), "w") as f:
        json.dump(out, f, indent=0)
    print(len(out), "tensors:", *out.keys(), sep="\n  ")


if __name__ == "__main__":
    main()
