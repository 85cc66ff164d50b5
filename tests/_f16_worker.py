"""Worker of tests/test_f16_gpu.py: runs in its OWN process with VISTA_ACT_DTYPE=fp16 (the storage type is fixed per process: vista_amd/_lib.py
opens libvista_hip_f16.so, ops.ACT is torch.float16) and prints one JSON line of measured errors against the same goldens / fp32 references the
bf16 tests use. No assertions here: the parent test states the bounds."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt()).item()


def main():
    assert os.environ.get("VISTA_ACT_DTYPE") == "fp16"
    from oracle.make_golden import unet_inputs
    from tests.test_model_gpu import TRAJ, _sampler, build_unet
    from vista_amd import _lib, ops, synth
    lib = _lib.load()
    res = {"act_dtype": lib.vk_act_dtype(), "lib": os.path.basename(_lib.LIB_PATH), "ops_act": str(ops.ACT)}

    # ---- kernels against torch fp32 on the SAME fp16-rounded inputs ----
    g = torch.Generator(device="cuda").manual_seed(5)
    M, K, N = 4096, 320, 960
    x = (torch.randn(M, K, device="cuda", generator=g)).to(ops.ACT)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    b = torch.randn(N, device="cuda", generator=g)
    pw = ops.pack_linear(w, b, "cuda")
    y = ops.linear(x, pw)
    res["linear_dtype"] = str(y.dtype)
    res["linear"] = rel_l2(y, x.float() @ w.to(ops.ACT).float().t() + b)
    # the V column block of a q|k|v projection leaves as bf16 bits (alt_cols_from)
    y2 = ops.linear(x, pw, alt_cols_from=640)
    ref2 = x.float() @ w.to(ops.ACT).float().t() + b
    res["linear_alt_qk"] = rel_l2(y2[:, :640], ref2[:, :640])
    res["linear_alt_v"] = rel_l2(y2[:, 640:].contiguous().view(torch.bfloat16), ref2[:, 640:])
    # spatial attention straight from such a projection output
    n_img, heads, S, C = 2, 5, 2304, 320
    q3 = torch.randn(n_img * S, 3 * C, device="cuda", generator=g)
    qkv = q3.to(ops.ACT)
    qkv[:, 2 * C:] = q3[:, 2 * C:].to(torch.bfloat16).view(ops.ACT)
    o = ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n_img, heads, S, 64 ** -0.5, v_rows=True)
    qf = qkv[:, :C].float().view(n_img, S, heads, 64).transpose(1, 2)
    kf = qkv[:, C:2 * C].float().view(n_img, S, heads, 64).transpose(1, 2)
    vf = qkv[:, 2 * C:].contiguous().view(torch.bfloat16).float().view(n_img, S, heads, 64).transpose(1, 2)
    ref = torch.nn.functional.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(n_img * S, C)
    res["attn_spatial"] = rel_l2(o.view(n_img * S, C), ref)
    # temporal attention
    B, T, S2 = 2, 25, 256
    q3 = torch.randn(B * T * S2, 3 * C, device="cuda", generator=g)
    qkv = q3.to(ops.ACT)
    qkv[:, 2 * C:] = q3[:, 2 * C:].to(torch.bfloat16).view(ops.ACT)
    o = ops.attn_temporal(qkv, B, T, S2, heads, 64 ** -0.5)

    def tview(t):   # [(b t) s (h d)] -> [(b s) h t d]
        return t.view(B, T, S2, heads, 64).permute(0, 2, 3, 1, 4).reshape(B * S2, heads, T, 64)
    ref = torch.nn.functional.scaled_dot_product_attention(tview(qkv[:, :C].float()), tview(qkv[:, C:2 * C].float()),
                                                           tview(qkv[:, 2 * C:].contiguous().view(torch.bfloat16).float()))
    ref = ref.view(B, S2, heads, T, 64).permute(0, 3, 1, 2, 4).reshape(B * T * S2, C)
    res["attn_temporal"] = rel_l2(o.view(B * T * S2, C), ref)
    # GroupNorm + SiLU
    xg = torch.randn(4, 1024, 320, device="cuda", generator=g).to(ops.ACT)
    gam, bet = torch.randn(320, device="cuda", generator=g), torch.randn(320, device="cuda", generator=g)
    yg = ops.groupnorm(xg, gam, bet, 1e-5, True)
    refg = torch.nn.functional.silu(torch.nn.functional.group_norm(xg.float().transpose(1, 2), 32, gam, bet, 1e-5)).transpose(1, 2)
    res["groupnorm_silu"] = rel_l2(yg, refg)

    # ---- the UNet against the reference's own outputs ----
    def unet_err(net, gold):
        x8, ts, ctx, yv, mask = unet_inputs(gold["T"], gold["H"], gold["W"], seed=gold["seed_x"], sigma=gold["sigma"])
        out = net(x8.cuda(), timesteps=ts.cuda(), context=ctx.cuda(), y=yv.cuda(), cond_mask=mask.cuda(), num_frames=gold["T"]).cpu()
        return {"rel_l2": rel_l2(out, gold["out"]), "max_rel": ((out - gold["out"]).abs().max() / gold["out"].abs().max()).item(),
                "finite": bool(torch.isfinite(out).all())}
    tiny, _ = build_unet(64)
    res["unet_tiny_t5"] = unet_err(tiny, torch.load(os.path.join(GOLD, "unet_tiny_t5.pt")))
    res["unet_tiny_t25"] = unet_err(tiny, torch.load(os.path.join(GOLD, "unet_tiny_t25.pt")))

    # ---- 3-step samplers (fused path), config-1 miniature ----
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import FusedDenoiser
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper
    gs = torch.load(os.path.join(GOLD, "sampler_tiny.pt"))
    T, H, W = gs["T"], gs["H"], gs["W"]
    win = synth.window_inputs(T=T, H=H, W=W, seed=gs["seed_x"], n_cond=1, trajectory=TRAJ)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}  # noqa: E731
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    fused = FusedDenoiser(den, OpenAIWrapper(tiny))
    P = "vwm.modules.diffusionmodules.guiders."
    cfgs = {"vanilla": {"target": P + "VanillaCFG", "params": {"scale": 2.5}},
            "linear": {"target": P + "LinearPredictionGuider", "params": {"num_frames": T, "max_scale": 2.5, "min_scale": 1.0}},
            "triangle": {"target": P + "TrianglePredictionGuider", "params": {"num_frames": T, "max_scale": 2.5, "min_scale": 1.0}},
            "identity": {"target": P + "IdentityGuider"}}
    res["sampler"] = {}
    for name, cfg in cfgs.items():
        out = _sampler(cfg)(fused, win["noise"].clone().cuda(), cond=cu(win["c"]), uc=cu(win["uc"]), cond_frame=win["cond_frame"].cuda(),
                            cond_mask=win["cond_mask"].cuda()).cpu()
        res["sampler"][name] = rel_l2(out, gs[name])
    g1 = torch.load(os.path.join(GOLD, "config1_tiny.pt"))
    w1 = synth.window_inputs(T=g1["T"], H=g1["H"], W=g1["W"], seed=g1["seed_x"], n_cond=1)
    den1 = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=g1["T"])
    s = _sampler({"target": P + "VanillaCFG", "params": {"scale": 2.5}}, steps=g1["steps"])
    out = s(FusedDenoiser(den1, OpenAIWrapper(tiny)), w1["noise"].clone().cuda(), cond=cu(w1["c"]), uc=cu(w1["uc"]),
            cond_frame=w1["cond_frame"].cuda(), cond_mask=w1["cond_mask"].cuda()).cpu()
    res["config1_miniature"] = rel_l2(out, g1["out"].float())
    del tiny, fused
    torch.cuda.empty_cache()

    # ---- the shipped 1.65 B configuration ----
    full, _ = build_unet(320)
    res["unet_full_t5"] = unet_err(full, torch.load(os.path.join(GOLD, "unet_full_t5.pt")))
    if "--full-size" in sys.argv:   # BASELINE config 2's own shape against the oracle's checksum set (as tests/test_model_gpu.py)
        from tools.make_full_size_checksums import H as FH, SEED, SIGMA, T as FT, W as FW, sample_positions
        gold = json.load(open(os.path.join(GOLD, "full_size_step_checksums.json")))
        x8, ts, ctx, yv, mask = unet_inputs(FT, FH, FW, seed=SEED, sigma=SIGMA)
        with torch.no_grad():
            out = full(x8.cuda(), timesteps=ts.cuda(), context=ctx.cuda(), y=yv.cuda(), cond_mask=mask.cuda(), num_frames=FT).float().cpu()
        num = dsum = 0.0
        for f, rec in enumerate(gold["frames"]):
            p = sample_positions(f)
            got, ref = out[f][p[:, 0], p[:, 1], p[:, 2]], torch.tensor(rec["samples"])
            num += (got - ref).pow(2).sum().item()
            dsum += ref.pow(2).sum().item()
        res["full_size_step"] = {"rel_l2": (num / dsum) ** 0.5, "finite": bool(torch.isfinite(out).all())}
    # ---- mixed storage: the first stage and the conditioner store bf16 inside this fp16 process (ops.storage) ----
    from oracle.make_golden_vae import TINY, images, latents
    from tests.test_vae_gpu import _decoder
    from vista_amd.modules.diffusionmodules.model import Encoder
    gv = torch.load(os.path.join(GOLD, "vae_tiny.pt"))
    dec, _, _ = _decoder("k311")
    z = latents(gv["T"], gv["H"], gv["W"], gv["seed_z"]).cuda()
    out = dec(z, timesteps=gv["T"])
    enc = Encoder(**TINY)
    eshapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(synth.seeded_state_dict(eshapes, 0), strict=True)
    mom = enc.cuda()(images(5, 64, 128, 7).cuda())
    import hashlib
    res["vae"] = {"decoder_rel_l2": rel_l2(out, gv["out_k311"]), "decoder_dtype": str(out.dtype), "encoder_rel_l2": rel_l2(mom, gv["enc_moments"]),
                  "act_after": str(ops.ACT), "current_after": _lib.CURRENT, "libs_loaded": sorted(_lib._libs),
                  "decoder_sha": hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest(), "encoder_sha": hashlib.sha256(mom.cpu().numpy().tobytes()).hexdigest()}
    print("F16_RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
