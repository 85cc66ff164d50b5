"""Checkpoint ingest end to end on the MI355X (reference: bin_to_st.py:10-54, sample_utils.py:54-80).

A DeepSpeed-style training dump of the shipped 1.65 B-parameter VideoUNet -- `_forward_module.` prefix, LoRA adapters beside frozen
projections, LitEma shadows for trainable tensors, EMA bookkeeping scalars, a stray key -- is built around the seeded weights of
tests/golden/unet_full_t5.pt, taken through `vista_amd.checkpoint.convert_training_checkpoint` (pinned to the reference's bin_to_st.py by
tests/test_checkpoint_cpu.py), written as `vista.safetensors`, read back with `load_checkpoint`, loaded `strict=False` into a VideoUNet that
ALREADY holds (and has packed) other weights, and run on the HIP path against the reference's own output for those weights:
  * bf16 path at the per-forward tolerance of tests/test_model_gpu.py (rel-L2 <= 2.5e-2, max <= 8e-2 max|ref|),
  * the fp8 (BASELINE config 5) packs of the same loaded weights at the config-5 tolerance of tests/test_fp8_gpu.py,
  * a captured hipGraph of the forward is NOT replayed across the load (the graph's launches point at freed packs)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
UP = "model.diffusion_model."


def rel_l2(a, b):
    return ((a.float() - b.float()).pow(2).sum().sqrt() / b.float().pow(2).sum().sqrt()).item()


def training_dump(target, seed=11):
    """`target`: {VideoUNet key: tensor} the converted checkpoint must reproduce. Returns the dict a DeepSpeed `pytorch_model.bin` of the
    Lightning engine would hold (names as oracle/make_golden_ckpt.py::synthetic_training_dict, which the reference's own bin_to_st.py
    accepts): every tensor under `_forward_module.model.diffusion_model.`;
      * two transformer blocks trained with LoRA AND tracked by LitEma: the live family (stale frozen projection + stale adapters) and the
        EMA family `model_ema.<dotless name>` (frozen = target - up @ down, adapters = down / up). bin_to_st.py:10-31 merges each family's
        adapters into its own projection, :38-47 then replaces the live tensors by the EMA ones -> target;
      * one block with live LoRA only (frozen = target - up @ down): the merge alone restores target;
      * two trainable tensors with an EMA shadow and no adapters; decay / num_updates scalars; a stray optimizer key."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *shape: torch.randn(*shape, generator=g)  # noqa: E731
    dump, fm, ema = {}, "_forward_module.", "_forward_module.model_ema."
    dot = lambda k: ("diffusion_model." + k).replace(".", "")  # noqa: E731  (LitEma strips the dots of the name under `model.`)
    for k, v in target.items():
        dump[fm + UP + k] = v.clone()
    n_lora = 0
    for blk, with_ema in (("input_blocks.1.1.transformer_blocks.0.attn1", True), ("middle_block.1.transformer_blocks.0.attn1", True),
                          ("output_blocks.11.1.time_stack.0.attn1", False)):
        for proj, short in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "out")):
            wk = f"{blk}.{proj}.weight"
            w = target[wk]
            down, up = rn(16, w.shape[1]) * w.shape[1] ** -0.5, rn(w.shape[0], 16) * 0.05
            if with_ema:
                dump[fm + UP + wk] = w + 0.05 * rn(*w.shape)                                   # live family: stale, and so are its adapters
                dump[fm + UP + f"{blk}.{short}_adapter_down.weight"] = rn(*down.shape) * 0.1
                dump[fm + UP + f"{blk}.{short}_adapter_up.weight"] = rn(*up.shape) * 0.1
                dump[ema + dot(wk)] = w - up @ down                                            # EMA family: the one that counts
                dump[ema + dot(f"{blk}.{short}_adapter_down.weight")] = down
                dump[ema + dot(f"{blk}.{short}_adapter_up.weight")] = up
            else:
                dump[fm + UP + wk] = w - up @ down
                dump[fm + UP + f"{blk}.{short}_adapter_down.weight"] = down
                dump[fm + UP + f"{blk}.{short}_adapter_up.weight"] = up
            n_lora += 1
        if with_ema:
            bk = f"{blk}.to_out.0.bias"
            dump[fm + UP + bk] = target[bk] + 0.1 * rn(*target[bk].shape)
            dump[ema + dot(bk)] = target[bk].clone()
    for k in ("out.2.weight", "input_blocks.4.0.in_layers.2.weight"):   # trainable tensors without adapters: EMA shadow replaces the live copy
        dump[fm + UP + k] = target[k] + 0.1 * rn(*target[k].shape)
        dump[ema + dot(k)] = target[k].clone()
    dump[ema + "decay"] = torch.tensor(0.9999)
    dump[ema + "num_updates"] = torch.tensor(1234)
    dump["optimizer_stray_key"] = torch.zeros(3)
    return dump, n_lora


def test_converted_safetensors_load_pack_forward_bf16_fp8_and_graph(tmp_path):
    from safetensors.torch import save_file
    from oracle.make_golden import unet_inputs
    from vista_amd import checkpoint, ops, synth
    from vista_amd.config import unet_kwargs
    from vista_amd.modules import attention as att
    from vista_amd.modules.diffusionmodules.denoiser import Denoiser
    from vista_amd.modules.diffusionmodules.sampling import EulerEDMSampler, FusedDenoiser
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    from vista_amd.modules.diffusionmodules.wrappers import OpenAIWrapper

    g = torch.load(os.path.join(GOLD, "unet_full_t5.pt"))
    net = VideoUNet(**unet_kwargs(320))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert synth.shapes_digest(shapes) == g["digest"]
    target = synth.seeded_state_dict(shapes, 0)          # the weights the reference produced g["out"] with
    other = synth.seeded_state_dict(shapes, 3)           # what the network holds BEFORE the checkpoint is loaded
    net.load_state_dict(other, strict=True)
    net = net.cuda().eval()
    T, H, W = g["T"], g["H"], g["W"]
    x8, ts, ctx, y, mask = unet_inputs(T, H, W, seed=g["seed_x"], sigma=g["sigma"])
    args = dict(timesteps=ts.cuda(), context=ctx.cuda(), y=y.cuda(), cond_mask=mask.cuda(), num_frames=T)

    def fwd():
        with torch.no_grad():
            return net(x8.cuda(), **args).float().cpu()
    before = fwd()                                       # packs every weight of the OTHER state dict
    assert rel_l2(before, g["out"]) > 0.5, "the pre-load weights must not already match the golden"

    # a short fused sampler run with the forward replayed from a hipGraph: captured on the OTHER weights
    w = synth.window_inputs(T=T, H=H, W=W, seed=6)
    cu = lambda d: {k: v.cuda() for k, v in d.items()}  # noqa: E731
    den = Denoiser(scaling_config={"target": "vwm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"}, num_frames=T)
    fd = FusedDenoiser(den, OpenAIWrapper(net))

    def sample(graph):
        from vista_amd.modules.diffusionmodules.sampling import FusedLoop
        s = EulerEDMSampler(num_steps=2, discretization_config={"target": "vwm.modules.diffusionmodules.discretizer.EDMDiscretization",
                                                                 "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
                            guider_config={"target": "vwm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 2.5}}, device="cuda")
        x, sigmas, _, c, uc = s.prepare_sampling_loop(w["noise"].clone().cuda(), cu(w["c"]), cu(w["uc"]))
        sig = [float(v) for v in sigmas]
        loop = FusedLoop(s, fd, x.float().clone(), c, uc, w["cond_frame"].cuda(), w["cond_mask"].cuda().float(), True, sig, graph=graph)
        for i in range(len(sig) - 1):
            loop.step(i)
        return loop.finish().float().cpu()
    g_other = sample(True)
    assert torch.equal(g_other, sample(False)), "graph replay and eager enqueue run the same kernels on the same buffers: bitwise equal"

    # ---- the checkpoint: training dump -> reference-format conversion -> vista.safetensors -> load_checkpoint -> load_into(strict=False) ----
    dump, n_lora = training_dump(target)
    conv = checkpoint.convert_training_checkpoint(dump)
    assert n_lora == 12 and not any("_adapter_down" in k or "_adapter_up" in k or "model_ema" in k or k.startswith("_forward_module") for k in conv)
    conv["conditioner.embedders.0.dummy"] = torch.zeros(3)     # other engine components travel in the same file and are ignored
    path = str(tmp_path / "vista.safetensors")
    save_file({k: v.contiguous() for k, v in conv.items()}, path)
    sd = checkpoint.load_checkpoint(path)
    rep = checkpoint.load_into(sd, unet=net, verbose=False)
    assert rep == {"unet": ([], [])}, rep
    worst = max((net.state_dict()[k].float().cpu() - v).abs().max().item() / max(v.abs().max().item(), 1e-12) for k, v in target.items())
    assert worst <= 1e-5, f"LoRA merge / EMA replacement did not restore the target weights (worst relative deviation {worst:.2e})"

    out = fwd()
    r, mx = rel_l2(out, g["out"]), ((out - g["out"]).abs().max() / g["out"].abs().max()).item()
    print(f"[checkpoint] converted safetensors -> load -> bf16 pack -> forward vs reference golden: rel-L2 {r:.3e}, max {mx:.3e}")
    assert torch.isfinite(out).all() and r <= 2.5e-2 and mx <= 8e-2, (r, mx)

    # the graph captured before the load must not be replayed: the graphed run on the NEW weights equals the eager run on the new weights
    g_new = sample(True)
    e_new = sample(False)
    assert torch.equal(g_new, e_new), "hipGraph replayed stale launches after load_state_dict"
    assert rel_l2(g_new, g_other) > 1e-2, "the sampler output did not change with the weights"

    # ---- BASELINE config 5: the fp8 packs are built from the loaded weights ----
    saved, saved_tile = dict(att.FP8), ops.TILE_CFG
    try:
        for k in ("feedforward", "conv", "attention", "proj"):
            att.FP8[k] = True
        out8 = fwd()
    finally:
        att.FP8.update(saved)
        ops.TILE_CFG = saved_tile
    r8 = rel_l2(out8, g["out"])
    print(f"[checkpoint] same weights, fp8 (config 5) packs: rel-L2 {r8:.3e}")
    assert torch.isfinite(out8).all() and r8 <= 6e-2, r8
    assert rel_l2(fwd(), g["out"]) == r, "switching config 5 off restores the bf16 path bit for bit"
