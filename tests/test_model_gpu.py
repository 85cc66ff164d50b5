"""End-to-end parity of the MI355X path against the REAL reference's outputs (tests/golden/, produced by
oracle/make_golden.py on CPU fp32) through the reference-compatible class API.

Stated tolerance of the bf16 build (bf16 storage of activations AND weights, fp32 accumulation, against the fp32 reference). The bounds are
the values measured in round 6 (profiles/r06_gpu_suite_parity_lines.txt; every reduction runs in a fixed order, so a run reproduces them bit for
bit) plus 25 %, one bound per case -- VERDICT r5 item 2 (iii): per UNet forward rel-L2 <= 1.65e-2 and max|err| <= 2.2e-2 max|ref|; 3-step samplers
per guider (BOUNDS below). Where the 1.2e-2 comes from: profiles/r06_error_budget.txt -- bf16 weights alone are 6.3e-3, the bf16 residual stream
8.5e-3, bf16 GEMM operands 5.3e-3 (root-sum-square 1.19e-2): it is the format's floor, not a kernel's. The fp16-storage build of the same kernels
(the reference's own autocast width) measures 1.6e-3 on the same goldens: tests/test_f16_gpu.py."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TRAJ = [0.5, 0, 1.0, 0, 1.5, 0.1, 2.0, 0.2]


def rel_l2(a, b):
    return ((a.float() - b.float()).pow(2).sum().sqrt() / b.float().pow(2).sum().sqrt()).item()


def build_unet(model_channels, seed=0):
    from vista_amd import synth
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    net = VideoUNet(**unet_kwargs(model_channels))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth.seeded_state_dict(shapes, seed), strict=True)
    return net.cuda().eval(), shapes


_CACHE = {}


def tiny_unet():
    if "tiny" not in _CACHE:
        _CACHE["tiny"] = build_unet(64)
    return _CACHE["tiny"]


def check_unet(net, shapes, g, tag):
    from oracle.make_golden import unet_inputs
    from vista_amd import synth
    assert synth.shapes_digest(shapes) == g["digest"], "state-dict names/shapes differ from the reference VideoUNet"
    x8, ts, ctx, y, mask = unet_inputs(g["T"], g["H"], g["W"], seed=g["seed_x"], sigma=g["sigma"])
    out = net(x8.cuda(), timesteps=ts.cuda(), context=ctx.cuda(), y=y.cuda(), cond_mask=mask.cuda(), num_frames=g["T"]).cpu()
    ref = g["out"]
    r, mx = rel_l2(out, ref), ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"[parity] {tag}: rel-L2 {r:.4e}  max-abs/max-ref {mx:.4e}")
    assert torch.isfinite(out).all()
    assert r <= 1.65e-2 and mx <= 2.2e-2, f"{tag}: rel-L2 {r:.3e}, max-rel {mx:.3e}"   # measured 1.21-1.34e-2 / 1.60-1.67e-2


@pytest.mark.parametrize("tag", ["t5", "t25"])
def test_unet_tiny_vs_reference_golden(tag):
    net, shapes = tiny_unet()
    check_unet(net, shapes, torch.load(os.path.join(GOLD, f"unet_tiny_{tag}.pt")), f"unet_tiny_{tag}")


def test_unet_full_width_vs_reference_golden():
    """The shipped 1.65 B-parameter configuration (vista.yaml) at latent 16x32, T=5, against the reference's own output."""
    net, shapes = build_unet(320)
    check_unet(net, shapes, torch.load(os.path.join(GOLD, "unet_full_t5.pt")), "unet_full_t5")
    del net
    torch.cuda.empty_cache()


def _sampler(guider_cfg, steps=3):
    from vista_amd.modules.diffusionmodules.sampling import EulerEDMSampler
    P = "vwm.modules.diffusionmodules."  # the reference's target strings resolve to this package (vista_amd/util.py)
    return EulerEDMSampler(num_steps=steps, discretization_config={"target": P + "discretizer.EDMDiscretization",
                                                                   "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
                           guider_config=guider_cfg, s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=False, device="cuda")


@pytest.mark.parametrize("halves", [False, True], ids=["one_forward_of_50", "two_forwards_of_25"])
def test_full_size_cfg_step_vs_oracle_checksums(halves):
    """(halves: the step as bench.py runs it since round 6 -- the uncond and the cond clip as two forwards of 25 images, FusedLoop(cfg_streams=True); the
    concurrent graph replays are bitwise these serial halves, test_cfg_streams_* / tools/two_stream_probe.py. Other tile and split-K choices at the deep
    levels than the 50-image launch, i.e. another rounding order: measured 1.335e-2 against the one-forward form's 1.287e-2.)
    BASELINE.json config 2 itself, once per test run: ONE CFG-doubled UNet forward (N = 50 images = uncond + cond clip of 25 frames, latent
    72 x 128, the shipped 1.65 B-parameter network with seeded non-zero weights) on the HIP path against the checksum set the CPU fp32 oracle
    produced for exactly these inputs (tools/make_full_size_checksums.py -> tests/golden/full_size_step_checksums.json: per frame mean, rms
    and 64 values at seeded positions; the oracle itself is pinned to the reference by tests/test_oracle_cpu.py). Stated tolerance, bf16
    storage / fp32 accumulation through ~100 layers: per frame |mean - ref| <= 1e-2 rms, rms within 1 %, every sampled value within
    8e-2 rms + 2e-2 |ref| (six standard deviations of the measured 1.3e-2 error level: 3200 samples), and the relative L2 error over all
    samples <= 1.65e-2 (the per-forward bound of this file). Measured in round 6: 1.287e-2, worst frame mean 4.8e-3 rms, rms 1.8e-3."""
    import json
    from oracle.make_golden import unet_inputs
    from tools.make_full_size_checksums import H, NS, SEED, SIGMA, T, W, sample_positions
    path = os.path.join(GOLD, "full_size_step_checksums.json")
    gold = json.load(open(path))
    assert (gold["T"], gold["H"], gold["W"], gold["seed"], gold["sigma"], gold["n_samples"]) == (T, H, W, SEED, SIGMA, NS)
    net, _ = build_unet(320)
    x8, ts, ctx, y, mask = unet_inputs(T, H, W, seed=SEED, sigma=SIGMA)
    with torch.no_grad():
        parts = [slice(0, T), slice(T, 2 * T)] if halves else [slice(0, 2 * T)]
        out = torch.cat([net(x8[sl].cuda(), timesteps=ts[sl].cuda(), context=ctx[sl].cuda(), y=y[sl].cuda(), cond_mask=mask[sl].cuda(), num_frames=T).float().cpu()
                         for sl in parts])
    del net
    torch.cuda.empty_cache()
    assert out.shape == (2 * T, 4, H, W) and torch.isfinite(out).all()
    num = den = 0.0
    worst = {"mean": 0.0, "rms": 0.0, "sample": 0.0}
    for f, rec in enumerate(gold["frames"]):
        o = out[f]
        rms = rec["rms"]
        p = sample_positions(f)
        got = o[p[:, 0], p[:, 1], p[:, 2]]
        ref = torch.tensor(rec["samples"])
        worst["mean"] = max(worst["mean"], abs(o.mean().item() - rec["mean"]) / rms)
        worst["rms"] = max(worst["rms"], abs(o.pow(2).mean().sqrt().item() / rms - 1.0))
        worst["sample"] = max(worst["sample"], ((got - ref).abs() / (8e-2 * rms + 2e-2 * ref.abs())).max().item())
        num += (got - ref).pow(2).sum().item()
        den += ref.pow(2).sum().item()
    rel = (num / den) ** 0.5
    print(f"[full-size CFG step] N=50 72x128 full width{' as two forwards of 25' if halves else ''}: sampled rel-L2 {rel:.3e}; worst frame mean {worst['mean']:.2e} rms, rms {worst['rms']:.2e}, "
          f"sample {worst['sample']:.2f} of its tolerance")
    assert worst["mean"] <= 1e-2 and worst["rms"] <= 1e-2 and worst["sample"] <= 1.0 and rel <= 1.65e-2, (rel, worst)


def test_sampler_and_denoiser_vs_reference_golden():
    from vista_amd import synth
    from vista_amd.modules.diffusionmodules import guiders
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import FusedDenoiser
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    g = torch.load(os.path.join(GOLD, "sampler_tiny.pt"))
    net, _ = tiny_unet()
    T, H, W = g["T"], g["H"], g["W"]
    w = synth.window_inputs(T=T, H=H, W=W, seed=g["seed_x"], n_cond=1, trajectory=TRAJ)

    def cuda(d):
        return {k: v.cuda() for k, v in d.items()}
    wrapper = OpenAIWrapper(net)
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    P = "vwm.modules.diffusionmodules.guiders."
    cfgs = {"vanilla": {"target": P + "VanillaCFG", "params": {"scale": 2.5}},
            "linear": {"target": P + "LinearPredictionGuider", "params": {"num_frames": T, "max_scale": 2.5, "min_scale": 1.0}},
            "triangle": {"target": P + "TrianglePredictionGuider", "params": {"num_frames": T, "max_scale": 2.5, "min_scale": 1.0}},
            "identity": {"target": P + "IdentityGuider"}}
    fused = FusedDenoiser(den, wrapper)

    def closure(x, sigma, cond, cond_mask):  # the reference's own closure shape (sample_utils.py:314-315)
        return den(wrapper, x, sigma, cond, cond_mask)

    # measured (round 6, both paths within 2e-4 of each other): vanilla 1.342e-2, linear 1.264 / 1.279e-2, triangle 1.270e-2, identity 7.80e-3
    BOUNDS = {"vanilla": 1.68e-2, "linear": 1.6e-2, "triangle": 1.59e-2, "identity": 9.8e-3}
    for name, cfg in cfgs.items():
        for path, dn in (("fused", fused), ("generic", closure)):
            noise = w["noise"].clone().cuda()
            out = _sampler(cfg)(dn, noise, cond=cuda(w["c"]), uc=cuda(w["uc"]), cond_frame=w["cond_frame"].cuda(),
                                cond_mask=w["cond_mask"].cuda()).cpu()
            r = rel_l2(out, g[name])
            print(f"[parity] sampler {name}/{path}: rel-L2 {r:.4e}")
            assert r <= BOUNDS[name], f"sampler {name}/{path}: rel-L2 {r:.3e} > {BOUNDS[name]:.3e}"
            # the reference scales the caller's noise tensor in place (sampling.py:36)
            assert torch.allclose(noise.cpu(), g[name + "_noise_after"], rtol=1e-5, atol=1e-5)
            # cond frames are replaced exactly at the end (sampling.py:122)
            assert torch.equal(out[0], w["cond_frame"][0])
    # rollout-style window (BASELINE config 4): 3 carried-over cond frames, triangle guidance, trajectory action embedding
    w3 = synth.window_inputs(T=T, H=H, W=W, seed=22, n_cond=3, trajectory=[1.0, 0.2, 2.0, 0.5, 3.0, 0.9, 4.0, 1.4])
    out = _sampler(cfgs["triangle"])(fused, w3["noise"].clone().cuda(), cond=cuda(w3["c"]), uc=cuda(w3["uc"]),
                                     cond_frame=w3["cond_frame"].cuda(), cond_mask=w3["cond_mask"].cuda()).cpu()
    r = rel_l2(out, g["rollout3"])
    print(f"[parity] sampler rollout3 (3 cond frames, triangle): rel-L2 {r:.4e}")
    assert r <= 1.0e-2 and torch.equal(out[:3], w3["cond_frame"][:3])   # measured 8.03e-3
    # plain Denoiser.forward boundary
    sig = torch.full((T,), 5.0)
    x2, s2, c2, m2 = guiders.VanillaCFG(2.5).prepare_inputs((w["noise"] * 5.0).cuda(), sig.cuda(), cuda(w["c"]), w["cond_mask"].cuda(),
                                                            cuda(w["uc"]))
    d = den(wrapper, x2, s2, c2, m2).cpu()
    r = rel_l2(d, g["denoiser_out"])
    print(f"[parity] denoiser: rel-L2 {r:.4e}")
    assert r <= 1.6e-2   # measured 1.284e-2


def test_stochastic_sampler_step_vs_reference_golden():
    """The gamma > 0 branch of EulerEDMSampler.sampler_step (reference sampling.py:78-83: sigma_hat = sigma (1 + gamma), x += eps * s_noise *
    sqrt(sigma_hat^2 - sigma^2)), which Vista's configs leave off but the sampler API carries: 4 steps, s_churn 1.2 (gamma 0.3) inside
    [s_tmin 0.05, s_tmax 400] -> steps 1 and 2 churn, 0 and 3 do not. The noise the REFERENCE drew (recorded by oracle/make_golden_churn.py) is
    injected through the sampler's `noise_fn` hook; the FusedDenoiser is passed, so this also checks that s_churn > 0 leaves the fused path."""
    from vista_amd import synth
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import EulerEDMSampler, FusedDenoiser
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    g = torch.load(os.path.join(GOLD, "sampler_churn_tiny.pt"))
    net, _ = tiny_unet()
    T, H, W, prm = g["T"], g["H"], g["W"], g["params"]
    w = synth.window_inputs(T=T, H=H, W=W, seed=g["seed_x"], n_cond=1, trajectory=TRAJ)
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}  # noqa: E731
    P = "vwm.modules.diffusionmodules."
    s = EulerEDMSampler(discretization_config={"target": P + "discretizer.EDMDiscretization", "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
                        guider_config={"target": P + "guiders.VanillaCFG", "params": {"scale": 2.5}}, verbose=False, device="cuda", **prm)
    draws = iter(d.float() for d in g["draws"])
    s.noise_fn = lambda x: next(draws)
    out = s(FusedDenoiser(den, OpenAIWrapper(net)), w["noise"].clone().cuda(), cond=cu(w["c"]), uc=cu(w["uc"]),
            cond_frame=w["cond_frame"].cuda(), cond_mask=w["cond_mask"].cuda()).cpu()
    assert next(draws, None) is None, "the sampler did not draw noise on exactly the steps the reference did"
    r = rel_l2(out, g["out"])
    print(f"[parity] stochastic sampler (s_churn {prm['s_churn']}, 4 steps, injected reference draws): rel-L2 {r:.4e}")
    assert torch.isfinite(out).all() and r <= 1.9e-2 and torch.equal(out[0], w["cond_frame"][0])   # measured 1.505e-2


def test_config1_miniature_25_frames_10_steps_vs_reference_golden():
    """BASELINE config 1 in miniature: 1 cond frame -> 25 frames, 10 EDM steps, VanillaCFG 2.5, against the real reference sampler's
    output (CPU fp32, stored fp16). Measured 1.04e-2 (ten Euler steps do not compound the per-step bf16 error: each step contracts towards the
    denoised estimate); tolerance rel-L2 <= 1.3e-2."""
    from vista_amd import synth
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import FusedDenoiser
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    g = torch.load(os.path.join(GOLD, "config1_tiny.pt"))
    net, _ = tiny_unet()
    T, H, W = g["T"], g["H"], g["W"]
    w = synth.window_inputs(T=T, H=H, W=W, seed=g["seed_x"], n_cond=1)
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}  # noqa: E731
    s = _sampler({"target": "vwm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 2.5}}, steps=g["steps"])
    out = s(FusedDenoiser(den, OpenAIWrapper(net)), w["noise"].clone().cuda(), cond=cu(w["c"]), uc=cu(w["uc"]),
            cond_frame=w["cond_frame"].cuda(), cond_mask=w["cond_mask"].cuda()).cpu()
    r = rel_l2(out, g["out"].float())
    print(f"[parity] config-1 miniature (25 frames, 10 steps, CFG 2.5): rel-L2 {r:.4e}")
    assert torch.isfinite(out).all() and r <= 1.3e-2 and torch.equal(out[0], w["cond_frame"][0])


def test_unet_properties_batch_and_determinism():
    """Size-independent properties: clips in a batch are independent (running [A;B] == running A and B) and repeated runs
    are bitwise identical (every reduction, including the GroupNorm statistics, runs in a fixed order: no atomics).
    bf16 rounding amplifies ANY low-order difference to the 1e-2 noise floor over ~100 layers, so these are strict."""
    from oracle.make_golden import unet_inputs
    from vista_amd import ops
    net, _ = tiny_unet()
    T, H, W = 5, 16, 32
    xa = [t.cuda() for t in unet_inputs(T, H, W, seed=101, sigma=2.0)]
    xb = [t.cuda() for t in unet_inputs(T, H, W, seed=202, sigma=30.0)]
    xs = [torch.cat([a, b], 0) for a, b in zip(xa, xb)]  # each input is a CFG pair = 2 clips; stacked -> 4 clips

    def run(x):
        return net(x[0], timesteps=x[1], context=x[2], y=x[3], cond_mask=x[4], num_frames=T)
    # the GEMM launcher may split K for small problems, and whether it does depends on the number of rows: bitwise batch
    # independence is a property of the un-split kernels (fixed-order reductions everywhere) ...
    saved, ops.SPLITK_WS_BYTES = ops.SPLITK_WS_BYTES, 0
    try:
        oa, ob, oab = run(xa), run(xb), run(xs)
        assert torch.equal(oab[:2 * T], oa) and torch.equal(oab[2 * T:], ob)
        assert torch.equal(run(xa), oa)
    finally:
        ops.SPLITK_WS_BYTES = saved
    # ... with split-K enabled every run is still bitwise repeatable, and batch composition changes results only at the bf16 noise floor
    sa, sab = run(xa), run(xs)
    assert torch.equal(run(xa), sa)
    assert rel_l2(sab[:2 * T], sa) <= 2.5e-2 and rel_l2(sa, oa) <= 2.5e-2


def test_unet_full_latent_size_tiny_width_vs_oracle():
    """BASELINE latent size (25 frames, 72x128 -> token counts 9216/2304/576/144 incl. ragged attention tiles) with the
    64-channel network, against the CPU oracle computed here (the oracle is pinned to the reference by tests/test_oracle_cpu.py)."""
    from oracle import vista_oracle as O
    from oracle.make_golden import unet_inputs
    from vista_amd import synth
    net, shapes = tiny_unet()
    sd = synth.seeded_state_dict(shapes, 0)
    T, H, W = 25, 72, 128
    x8, ts, ctx, y, mask = unet_inputs(T, H, W, seed=77, sigma=12.0)
    with torch.no_grad():
        ref = O.unet_forward(sd, x8, ts, ctx, y, mask, T)
    out = net(x8.cuda(), timesteps=ts.cuda(), context=ctx.cuda(), y=y.cuda(), cond_mask=mask.cuda(), num_frames=T).cpu()
    r, mx = rel_l2(out, ref), ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"[parity] tiny-width UNet at the full 25x72x128 latent vs oracle: rel-L2 {r:.4e} max-rel {mx:.4e}")
    assert torch.isfinite(out).all() and r <= 1.65e-2 and mx <= 2.3e-2   # measured 1.302e-2 / 1.84e-2


def test_product_path_refuses_cpu():
    """No CPU / eager fallback: a model left on the CPU must fail loudly instead of computing."""
    from vista_amd._lib import VistaHipError
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    net = VideoUNet(**unet_kwargs(64))
    with pytest.raises((VistaHipError, RuntimeError, TypeError)):
        net(torch.zeros(2, 8, 16, 32), timesteps=torch.zeros(2), context=torch.zeros(2, 1, 3456), y=torch.zeros(2, 768),
            cond_mask=torch.zeros(2), num_frames=2)


def oracle_50_step():
    """(window inputs, the CPU oracle's 50-step result) of the 64-channel network at T=5, latent 16x32, VanillaCFG 2.5 -- ~1 min on the host,
    so it is computed once per box (per-process dict + a file under the temp dir) and shared by the bf16 and fp8 tests."""
    import tempfile
    from oracle import vista_oracle as O
    from vista_amd import synth
    if "o50" in _CACHE:
        return _CACHE["o50"]
    net, shapes = tiny_unet()
    sd = synth.seeded_state_dict(shapes, 0)
    T, H, W, steps = 5, 16, 32, 50
    w = synth.window_inputs(T=T, H=H, W=W, seed=31, n_cond=1, trajectory=TRAJ)
    path = os.path.join(tempfile.gettempdir(), f"vista_oracle50_{synth.shapes_digest(shapes):08x}.pt")
    if os.path.exists(path):
        want = torch.load(path)
    else:
        with torch.no_grad():
            want = O.euler_edm_sample(lambda a, s, c, m: O.denoiser_forward(sd, a, s, c, m, T), w["noise"], w["c"], w["uc"], w["cond_frame"],
                                      w["cond_mask"], steps, scale=2.5)
        torch.save(want, path)
    _CACHE["o50"] = (w, want, T, steps)
    return _CACHE["o50"]


def run_50_step(net, w, T, steps):
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import FusedDenoiser
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}  # noqa: E731
    s = _sampler({"target": "vwm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 2.5}}, steps=steps)
    return s(FusedDenoiser(den, OpenAIWrapper(net)), w["noise"].clone().cuda(), cond=cu(w["c"]), uc=cu(w["uc"]),
             cond_frame=w["cond_frame"].cuda(), cond_mask=w["cond_mask"].cuda()).cpu()


def test_sampler_full_50_step_schedule_vs_oracle():
    """The WHOLE 50-step EDM schedule end to end (VERDICT r1 missing #6; the longest chain before was 10 steps): 64-channel network,
    T=5, latent 16x32, VanillaCFG 2.5, fused path, against the CPU oracle's 50-step run (computed here, ~1 min on the host).
    Each step is a contraction towards the denoised estimate (x <- x + (sigma_next/sigma - 1)(x - D(x))), so per-step bf16 noise does
    not grow without bound (measured 8.16e-3, below one forward's 1.3e-2); stated tolerance rel-L2 <= 1.02e-2 (measured + 25 %), measured value appended to gpurun_out/parity_50step.json."""
    import json
    net, _ = tiny_unet()
    w, want, T, steps = oracle_50_step()
    got = run_50_step(net, w, T, steps)
    r = rel_l2(got, want)
    print(f"[parity] 50-step CFG EulerEDM (tiny net, T=5, 16x32) vs oracle: rel-L2 {r:.4e}")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        json.dump({"steps": steps, "rel_l2": r}, open(os.path.join(d, "parity_50step.json"), "w"))
    assert torch.isfinite(got).all() and r <= 1.02e-2 and torch.equal(got[0], w["cond_frame"][0])


def test_hipgraph_replay_of_the_unet_forward_is_bitwise_the_eager_loop(monkeypatch):
    """VISTA_HIPGRAPH=1 / FusedLoop(graph=True): every step's UNet forward is a replay of ONE captured hipGraph (input latents and the noise
    level enter through device buffers, the per-step scalars stay in the sampler kernels outside the graph). Same kernels on the same data ->
    the 10-step result must equal the eager loop's bit for bit, also when the graph object is reused by a second window."""
    from vista_amd.modules.diffusionmodules.sampling import FusedLoop
    net, _ = tiny_unet()
    w, _, T, _ = _window_only()
    monkeypatch.delenv("VISTA_HIPGRAPH", raising=False)
    eager = run_50_step(net, w, T, 10)
    made = []
    orig = FusedLoop.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    monkeypatch.setattr(FusedLoop, "__init__", spy)
    monkeypatch.setenv("VISTA_HIPGRAPH", "1")
    graphed = run_50_step(net, w, T, 10)
    assert made and made[-1]._graph is not None and "graph" in made[-1]._graph, "the graph path did not run"
    assert torch.equal(graphed, eager)
    assert torch.equal(run_50_step(net, w, T, 10), eager)
    # the captured graph is cached on the UNet and shared by later FusedLoops of the same geometry (one capture, one warm-up
    # stream, one split-K workspace per device) ...
    assert len(made) >= 2 and made[-1]._graph is made[-2]._graph and len(net.__dict__["_hipgraph_cache"]) == 1
    from vista_amd import ops
    assert len(FusedLoop._WARM_STREAMS) == 1 and len(ops._GRAPH_TLS.ws) == 1 and made[-1]._graph["ws"] is ops.graph_workspace_tensor()
    # ... with the conditioning as static buffers: another window through the SAME graph equals its own eager run
    w2 = dict(w)
    w2["c"] = {k: (v * 0.5 if k == "crossattn" else v) for k, v in w["c"].items()}
    monkeypatch.setenv("VISTA_HIPGRAPH", "0")
    eager2 = run_50_step(net, w2, T, 4)
    monkeypatch.setenv("VISTA_HIPGRAPH", "1")
    assert torch.equal(run_50_step(net, w2, T, 4), eager2) and not torch.equal(eager2, run_50_step(net, w, T, 4))
    # ... and dropped when a parameter changes after the capture (its launches point at the old packed weights)
    with torch.no_grad():
        next(net.parameters()).mul_(1.0)
    g_old = made[-1]._graph
    run_50_step(net, w, T, 2)
    assert made[-1]._graph is not g_old


def test_cfg_streams_two_concurrent_half_graphs_are_bitwise_the_serial_halves(monkeypatch):
    """VISTA_CFG_STREAMS=1 / FusedLoop(cfg_streams=True), bench.py's default on one GPU: a step's two guidance halves run as two UNet
    forwards of n images -- eagerly one after the other on the launch stream, with graph replay as two hipGraphs replayed CONCURRENTLY on
    two streams (a split-K workspace each). Same kernels on the same data -> the two forms are equal bit for bit, also on a second window
    through the same graphs; against the one-forward form (2n images per launch: other tile / split-K choices at the deep levels) the result
    moves by bf16 rounding only and sits as close to the oracle's golden output (stated: <= 1.02e-2, the one-forward bound of this window)."""
    from vista_amd import ops
    from vista_amd.modules.diffusionmodules.sampling import FusedLoop
    net, _ = tiny_unet()
    w, want, T, _ = oracle_50_step()
    monkeypatch.delenv("VISTA_HIPGRAPH", raising=False)
    monkeypatch.delenv("VISTA_CFG_STREAMS", raising=False)
    one = run_50_step(net, w, T, 10)
    monkeypatch.setenv("VISTA_CFG_STREAMS", "1")
    serial = run_50_step(net, w, T, 10)
    made = []
    orig = FusedLoop.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    monkeypatch.setattr(FusedLoop, "__init__", spy)
    monkeypatch.setenv("VISTA_HIPGRAPH", "1")
    conc = run_50_step(net, w, T, 10)
    gs = made[-1]._graph
    assert isinstance(gs, list) and len(gs) == 2 and gs[0]["graph"] is not gs[1]["graph"], "the two-graph path did not run"
    assert gs[0]["ws"] is not gs[1]["ws"] and gs[0]["ws"].data_ptr() != gs[1]["ws"].data_ptr(), "concurrent graphs must not share a split-K workspace"
    assert torch.equal(conc, serial)
    assert torch.equal(run_50_step(net, w, T, 10), serial) and made[-1]._graph[0] is gs[0] and made[-1]._graph[1] is gs[1]   # cached on the UNet
    assert len(FusedLoop._CFG_STREAMS) == 1 and len(ops._GRAPH_TLS.ws) == 2
    r = rel_l2(conc, one)
    print(f"[parity] cfg_streams (two 5-image forwards) vs one 10-image forward, 10 steps: rel-L2 {r:.3e}")
    assert r <= 1.5e-2 and torch.equal(conc[0], w["cond_frame"][0])
    # the whole 50-step schedule in the two-stream form against the oracle
    full = run_50_step(net, w, T, 50)
    r50 = rel_l2(full, want)
    print(f"[parity] 50-step CFG EulerEDM, cfg_streams + graph replay, vs oracle: rel-L2 {r50:.4e}")
    assert torch.isfinite(full).all() and r50 <= 1.02e-2


def _window_only():
    from vista_amd import synth
    T = 5
    return synth.window_inputs(T=T, H=16, W=32, seed=31, n_cond=1, trajectory=TRAJ), None, T, 50
