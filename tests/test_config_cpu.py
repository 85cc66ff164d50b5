"""The drop-in boundary at the config level: classes are selected by `target:` strings only."""
import os

import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_YAML = "/root/reference/configs/inference/vista.yaml"


def _digest(net):
    from vista_amd import synth
    return synth.shapes_digest({k: tuple(v.shape) for k, v in net.state_dict().items()})


def test_shipped_overlay_instantiates_the_mi355x_classes():
    from vista_amd import config
    from vista_amd.modules.diffusionmodules import denoiser, sampling, video_model
    from vista_amd.util import instantiate_from_config
    cfg = config.load_config()
    mp = cfg["model"]["params"]
    net = instantiate_from_config(mp["network_config"])
    assert type(net) is video_model.VideoUNet and len(net.state_dict()) == 1496
    assert _digest(net) == torch.load(os.path.join(GOLD, "unet_full_t5.pt"))["digest"], "state-dict names/shapes = the reference VideoUNet's"
    den = instantiate_from_config(mp["denoiser_config"])
    assert type(den) is denoiser.Denoiser and den.num_frames == 25
    s = instantiate_from_config(cfg["sampler"])
    assert type(s) is sampling.EulerEDMSampler and len(s.host_sigmas()) == 51


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference tree not mounted")
def test_reference_vista_yaml_builds_this_package_unchanged_and_overlay_differs_only_in_targets():
    """The reference's REAL configs/inference/vista.yaml, fed unmodified (vwm.* targets) to instantiate_from_config."""
    from vista_amd import config
    from vista_amd.modules.diffusionmodules import denoiser, video_model
    from vista_amd.util import instantiate_from_config
    ref = yaml.safe_load(open(REF_YAML))
    rp = ref["model"]["params"]
    assert rp["network_config"]["target"].startswith("vwm.") and rp["denoiser_config"]["target"].startswith("vwm.")
    net = instantiate_from_config(rp["network_config"])
    assert type(net) is video_model.VideoUNet
    assert _digest(net) == torch.load(os.path.join(GOLD, "unet_full_t5.pt"))["digest"]
    assert type(instantiate_from_config(rp["denoiser_config"])) is denoiser.Denoiser
    # the shipped overlay = the reference entries with only the target strings rewritten
    ours = config.load_config()["model"]["params"]
    assert ours["network_config"]["params"] == rp["network_config"]["params"]
    assert ours["denoiser_config"]["params"]["num_frames"] == rp["denoiser_config"]["params"]["num_frames"] == rp["num_frames"]

    def targets(d, pre=""):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict):
                out.update(targets(v, pre + k + "."))
            elif k == "target":
                out[pre + k] = v
        return out
    def strip(d):  # the config without its target strings
        if isinstance(d, dict):
            return {k: strip(v) for k, v in d.items() if k != "target"}
        return [strip(v) for v in d] if isinstance(d, list) else d

    def targets_l(d, pre=""):  # targets() that also walks the emb_models LIST of the conditioner
        out = {}
        for k, v in (d.items() if isinstance(d, dict) else enumerate(d)):
            if isinstance(v, (dict, list)):
                out.update(targets_l(v, f"{pre}{k}."))
            elif k == "target":
                out[pre + k] = v
        return out
    for key in ("network_config", "denoiser_config", "conditioner_config"):
        for path, t in targets_l(ours[key]).items():
            rt = targets_l(rp[key])[path]
            assert t == rt.replace("vwm.modules.", "vista_amd.modules.").replace("vwm.models.", "vista_amd.models."), (path, t, rt)
    assert strip(ours["conditioner_config"]) == strip(rp["conditioner_config"]), "the conditioner entry differs from vista.yaml:42-140 only in targets"
    merged = config.overlay(ref, {"model": config.load_config()["model"]})
    assert merged["model"]["params"]["first_stage_config"] == rp["first_stage_config"], "everything this package does not replace stays the reference's"
    assert merged["model"]["params"]["network_config"]["target"].startswith("vista_amd.")
