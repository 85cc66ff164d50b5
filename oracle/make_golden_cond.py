"""Goldens of the conditioner.   python oracle/make_golden_cond.py

1. tests/golden/cond_general.pt: the REAL reference GeneralConditioner + get_batch + get_condition (sample_utils.py:232-276, extracted with `ast`
   and executed unmodified) over the fixture of oracle/cond_fixture.py -> c / uc dicts.
2. tests/golden/clip_tiny.pt: a 2-layer, 4-head (head dim 80) image tower at the real token geometry (224 px, patch 14 -> 257 tokens) through
   transformers.CLIPVisionModelWithProjection with seeded weights (oracle/clip_oracle.py) -> preprocessed pixels and the image embedding.
"""
import ast
import contextlib
import io
import math
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clip_oracle as CO, cond_fixture as CF, ref_shim  # noqa: E402
from vista_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TINY = dict(width=320, layers=2, heads=4, mlp=1280, patch=14, image=224, embed=64)


def reference_conditioner():
    ref_shim.install()
    for name in ("kornia", "open_clip"):  # imported at module level by the reference, never called by the classes used here
        sys.modules.setdefault(name, types.ModuleType(name))
    with contextlib.redirect_stdout(io.StringIO()):
        from vwm.modules.encoders import modules as RM
    stub = types.ModuleType("cond_stub")

    class StubImageEmbedder(RM.AbstractEmbModel):
        def __init__(self, dim):
            super().__init__()
            self.dim = dim

        def forward(self, img):
            return CF.stub_image_embed(img, self.dim)

    class StubLatentEmbedder(RM.AbstractEmbModel):
        def forward(self, z):
            return z * 1.0

    stub.StubImageEmbedder, stub.StubLatentEmbedder = StubImageEmbedder, StubLatentEmbedder
    sys.modules["cond_stub"] = stub
    with contextlib.redirect_stdout(io.StringIO()):
        cond = RM.GeneralConditioner(CF.emb_models("cond_stub"))
    return cond


def reference_get_condition():
    """get_batch / get_condition of the reference's sample_utils.py, executed as written (device cpu, no model offloading)."""
    tree = ast.parse(open(os.path.join(ref_shim.REF_ROOT, "sample_utils.py")).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("get_batch", "get_condition")]
    from einops import repeat
    from typing import List, Union
    ns = {"torch": torch, "math": math, "repeat": repeat, "List": List, "Union": Union, "ListConfig": list,
          "load_model": lambda m: None, "unload_model": lambda m: None}
    src = ast.Module(body=fns, type_ignores=[])
    for f in fns:  # the hard-coded device="cuda" default of get_batch
        for d in f.args.defaults:
            if isinstance(d, ast.Constant) and d.value == "cuda":
                d.value = "cpu"
    exec(compile(src, "sample_utils.py", "exec"), ns)
    return ns["get_condition"]


def main():
    torch.set_grad_enabled(False)
    os.makedirs(GOLD, exist_ok=True)
    cond = reference_conditioner()
    get_condition = reference_get_condition()
    model = types.SimpleNamespace(conditioner=cond)
    c, uc = get_condition(model, CF.value_dict(), CF.N, CF.FORCE_UC_ZERO, "cpu")
    torch.save({"c": c, "uc": uc, "N": CF.N}, os.path.join(GOLD, "cond_general.pt"))
    print({k: tuple(v.shape) for k, v in c.items()})

    from vista_amd.modules.encoders.modules import FrozenOpenCLIPImageEmbedder
    emb = FrozenOpenCLIPImageEmbedder(arch=TINY)
    shapes = {k: tuple(v.shape) for k, v in emb.state_dict().items()}
    sd = synth.seeded_state_dict(shapes, 7)
    img = torch.tanh(synth.seeded_tensor("clip.img", (2, 3, 96, 160), 7) * 1.5)
    img[1] = img[1].flip(-1) * 0.7
    pix = CO.preprocess(img, TINY["image"])
    out = CO.image_embed(sd, TINY, img)
    chk = CO.restated_visual(sd, TINY, pix)
    print("HF vs plain-torch restatement:", float((out - chk).abs().max()), "out rms", float(out.pow(2).mean().sqrt()))
    assert (out - chk).abs().max() < 2e-4 * out.abs().max() + 1e-5
    torch.save({"geometry": TINY, "seed": 7, "digest": synth.shapes_digest(shapes), "img_shape": tuple(img.shape), "pixels_checksum": float(pix.double().sum()),
                "pixels_sample": pix[:, :, ::16, ::16].clone(), "embed": out}, os.path.join(GOLD, "clip_tiny.pt"))
    print("clip_tiny embed", tuple(out.shape), out[0, :4])


if __name__ == "__main__":
    main()
