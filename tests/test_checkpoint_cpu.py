"""Checkpoint ingest on CPU: vista_amd.checkpoint against the golden produced by executing the
reference's own bin_to_st.py (oracle/make_golden_ckpt.py), and a safetensors round trip into the vista_amd modules."""
import json
import os

import pytest
import torch

from oracle.make_golden_ckpt import checksum, synthetic_training_dict
from vista_amd import checkpoint, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_convert_training_checkpoint_matches_reference_script():
    gold = json.load(open(os.path.join(GOLD, "ckpt_convert.json")))
    out = checkpoint.convert_training_checkpoint(synthetic_training_dict())
    assert sorted(out) == sorted(gold), "names differ from what bin_to_st.py writes"
    for k, (s, a, shape) in gold.items():
        cs = checksum(out[k])
        assert cs[2] == shape and abs(cs[0] - s) <= 1e-9 * max(1.0, abs(a)) and abs(cs[1] - a) <= 1e-9 * max(1.0, a), k


def test_merge_lora_is_w_plus_up_down():
    sd = {"x.to_q.weight": torch.randn(6, 5), "x.q_adapter_down.weight": torch.randn(2, 5), "x.q_adapter_up.weight": torch.randn(6, 2),
          "x.to_out.0.weight": torch.randn(5, 6), "x.out_adapter_down.weight": torch.randn(2, 6), "x.out_adapter_up.weight": torch.randn(5, 2)}
    want_q = sd["x.to_q.weight"] + sd["x.q_adapter_up.weight"] @ sd["x.q_adapter_down.weight"]
    want_o = sd["x.to_out.0.weight"] + sd["x.out_adapter_up.weight"] @ sd["x.out_adapter_down.weight"]
    out = checkpoint.merge_lora(sd)
    assert sorted(out) == ["x.to_out.0.weight", "x.to_q.weight"]
    assert torch.equal(out["x.to_q.weight"], want_q) and torch.equal(out["x.to_out.0.weight"], want_o)


def test_safetensors_round_trip_into_modules(tmp_path):
    from safetensors.torch import save_file
    from oracle.make_golden_vae import TINY
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    unet, dec = VideoUNet(**unet_kwargs(64)), VideoDecoder(video_kernel_size=[3, 1, 1], **TINY)
    usd = synth.seeded_state_dict({k: tuple(v.shape) for k, v in unet.state_dict().items()}, 7)
    dsd = synth.seeded_state_dict({k: tuple(v.shape) for k, v in dec.state_dict().items()}, 8)
    full = {checkpoint.UNET_PREFIX + k: v for k, v in usd.items()}
    full.update({checkpoint.DECODER_PREFIX + k: v for k, v in dsd.items()})
    full["conditioner.embedders.0.dummy"] = torch.zeros(3)          # other engine components are ignored
    path = str(tmp_path / "vista.safetensors")
    save_file(full, path)
    sd = checkpoint.load_checkpoint(path)
    rep = checkpoint.load_into(sd, unet=unet, decoder=dec, verbose=False)
    assert rep == {"unet": ([], []), "decoder": ([], [])}
    assert all(torch.equal(v, usd[k]) for k, v in unet.state_dict().items())
    assert all(torch.equal(v, dsd[k]) for k, v in dec.state_dict().items())
    # a drifting name is reported, not swallowed
    sd2 = dict(sd)
    sd2[checkpoint.UNET_PREFIX + "out.2.weightX"] = sd2.pop(checkpoint.UNET_PREFIX + "out.2.weight")
    rep = checkpoint.load_into(sd2, unet=unet, verbose=False)
    assert rep["unet"] == (["out.2.weight"], ["out.2.weightX"])
    with pytest.raises(NotImplementedError):
        checkpoint.load_checkpoint("model.bin")


def test_training_dump_of_the_gpu_test_converts_back_to_its_target():
    """The DeepSpeed-style dump tests/test_checkpoint_gpu.py builds (LoRA on live + EMA families, EMA shadows, bookkeeping scalars) goes back to
    the target weights through convert_training_checkpoint -- here on the 64-channel network (same names) -- and, where the reference is
    mounted, through the reference's own bin_to_st.py with identical results."""
    from tests.test_checkpoint_gpu import training_dump
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    net = VideoUNet(**unet_kwargs(64))
    target = synth.seeded_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 0)
    dump, n_lora = training_dump(target)
    conv = checkpoint.convert_training_checkpoint(dump)
    parts = checkpoint.split_by_component(conv)
    assert n_lora == 12 and sorted(parts["unet"]) == sorted(target) and not parts["decoder"] and not parts["rest"]
    worst = max((parts["unet"][k] - v).abs().max().item() / max(v.abs().max().item(), 1e-12) for k, v in target.items())
    assert worst <= 1e-6, worst
    ref = os.path.join(os.environ.get("VISTA_REFERENCE", "/root/reference"), "bin_to_st.py")
    if os.path.exists(ref):
        import safetensors.torch as st
        captured = {}
        real_load, real_save, real_mk = torch.load, st.save_file, os.makedirs
        torch.load = lambda *a, **k: {kk: v.clone() for kk, v in dump.items()}
        st.save_file = lambda d, path: captured.update(d)
        os.makedirs = lambda *a, **k: None
        try:
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                exec(compile(open(ref).read(), "bin_to_st.py", "exec"), {"__name__": "__main__"})
        finally:
            torch.load, st.save_file, os.makedirs = real_load, real_save, real_mk
        assert sorted(captured) == sorted(conv) and all(torch.equal(captured[k], conv[k]) for k in conv)
