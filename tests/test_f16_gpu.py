"""The fp16-storage build (libvista_hip_f16.so, VISTA_ACT_DTYPE=fp16: the reference's own autocast width, sample_utils.py:301-303) against the same
goldens as the bf16 build. The storage type is fixed per process, so the checks run in a worker process (tests/_f16_worker.py) and this file
states the bounds.

Stated tolerance of the fp16 build = measured (round 6, profiles/r06_f16_first_contact.txt) + 25-40 %: per UNet forward rel-L2 <= 2.5e-3 (measured
1.59e-3 full width, 1.69 / 1.78e-3 at 64 channels, 1.67e-3 on the full-size CFG step; profiles/r06_error_budget.txt predicted 1.5e-3; the
reference's own fp16-autocast run is 2.7-3.1e-3 from its fp32 self, profiles/r05_reference_hosted.txt; VERDICT r5 item 2 asked for <= 6e-3),
3-step samplers <= 2.4e-3 (1.0-1.8e-3), the 10-step config-1 miniature <= 1.8e-3 (1.35e-3). The bf16 build measures 1.2-1.3e-2 on the same goldens."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def f16():
    from vista_amd import build
    if not os.path.exists(build.LIB_F16):
        pytest.fail("vista_amd/lib/libvista_hip_f16.so is missing: __graft_entry__.build() links both storage variants")
    env = dict(os.environ, VISTA_ACT_DTYPE="fp16")
    env.pop("VISTA_HIP_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_f16_worker.py"), "--full-size"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("F16_RESULT ")][-1]
    res = json.loads(line[len("F16_RESULT "):])
    print("[fp16 build]", json.dumps(res))
    return res


def test_f16_process_loads_the_f16_library(f16):
    assert f16["act_dtype"] == 1 and f16["lib"] == "libvista_hip_f16.so" and f16["ops_act"] == "torch.float16" and f16["linear_dtype"] == "torch.float16"


def test_f16_kernels_vs_torch_fp32(f16):
    """Same fp16-rounded inputs, fp32 torch reference: what is left is the output rounding (2^-11 relative, rms ~ 2^-12.3 = 2e-4) and, for the
    attention, the bf16 numerators P and bf16 V (the zero-base softmax keeps bf16's exponent range in both builds)."""
    assert f16["linear"] <= 3e-4 and f16["linear_alt_qk"] <= 3e-4, f16            # measured 2.07e-4
    assert f16["linear_alt_v"] <= 2.2e-3, f16          # the V block leaves as bf16 (2^-9 relative rounding: rms 1.65e-3 measured)
    assert f16["attn_spatial"] <= 2.2e-3 and f16["attn_temporal"] <= 1.6e-3, f16   # measured 1.66e-3 / 1.20e-3
    assert f16["groupnorm_silu"] <= 3e-4, f16       # measured 2.08e-4


def test_f16_unet_vs_reference_golden(f16):
    for k in ("unet_tiny_t5", "unet_tiny_t25", "unet_full_t5"):
        assert f16[k]["finite"] and f16[k]["rel_l2"] <= 2.5e-3 and f16[k]["max_rel"] <= 3.5e-3, (k, f16[k])


def test_f16_full_size_cfg_step_vs_oracle_checksums(f16):
    assert f16["full_size_step"]["finite"] and f16["full_size_step"]["rel_l2"] <= 2.2e-3, f16["full_size_step"]


def test_f16_samplers_vs_reference_golden(f16):
    for name, r in f16["sampler"].items():
        assert r <= 2.4e-3, (name, r)
    assert f16["config1_miniature"] <= 1.8e-3, f16["config1_miniature"]


def test_f16_process_runs_the_first_stage_in_bf16_storage(f16):
    """ops.storage (round 6): inside the fp16 process the VAE decoder / encoder forwards switch to the bf16 library for the duration of the call (the reference
    runs the first stage without autocast; fp16's exponent is not safe there) -- bitwise what a bf16 process computes, and the fp16 state is back afterwards."""
    import hashlib
    import torch
    from oracle.make_golden_vae import TINY, images, latents
    from tests.test_vae_gpu import GOLD, _decoder
    from vista_amd import synth
    from vista_amd.modules.diffusionmodules.model import Encoder
    v = f16["vae"]
    assert v["decoder_dtype"] == "torch.float32" and v["decoder_rel_l2"] < 4e-2 and v["encoder_rel_l2"] < 4e-2, v
    assert v["act_after"] == "torch.float16" and v["current_after"] == "fp16" and v["libs_loaded"] == ["bf16", "fp16"], v
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"))
    dec, _, _ = _decoder("k311")
    out = dec(latents(g["T"], g["H"], g["W"], g["seed_z"]).cuda(), timesteps=g["T"])
    enc = Encoder(**TINY)
    shapes = {k: tuple(t.shape) for k, t in enc.state_dict().items()}
    enc.load_state_dict(synth.seeded_state_dict(shapes, 0), strict=True)
    mom = enc.cuda()(images(5, 64, 128, 7).cuda())
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == v["decoder_sha"], "the decoder of an fp16 process differs from a bf16 process's"
    assert hashlib.sha256(mom.cpu().numpy().tobytes()).hexdigest() == v["encoder_sha"], "the encoder of an fp16 process differs from a bf16 process's"
