"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the host logic (schedules, guiders,
scalings, config plumbing, state-dict contract) matches the reference's known answers, and the product path refuses to
compute on the CPU (no fallback)."""
import ctypes
import json
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_abi_header_table_and_library_agree():
    from vista_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vista_hip.h")).read()
    declared = set(re.findall(r"^\s*int\s+(vk_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no entry points parsed from include/vista_hip.h"
    assert declared == set(_lib.SIGNATURES), f"header vs ctypes table: {declared ^ set(_lib.SIGNATURES)}"
    assert os.path.exists(_lib.LIB_PATH), "libvista_hip.so not built (python -m vista_amd.build)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} missing from libvista_hip.so"
    assert _lib.load().vk_abi_version() == _lib.ABI_VERSION


def test_gemm_desc_layout_matches_header():
    """Field order of the ctypes mirror must follow the C struct (a silent mismatch would corrupt every GEMM launch)."""
    from vista_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vista_hip.h")).read()
    start = hdr.index("typedef struct VkGemmDesc {") + len("typedef struct VkGemmDesc {")
    body = hdr[start:hdr.index("} VkGemmDesc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        decl = re.sub(r"^(const\s+)?(void|float|int32_t|int64_t)\s*\*?", "", decl).strip()
        names += [n.strip().lstrip("*") for n in decl.split(",")]
    assert names == [f[0] for f in _lib.VkGemmDesc._fields_], names


def test_fp8_args_layout_matches_header():
    """VkFp8Args (config 5 entry points): the ctypes mirror follows the C struct field for field."""
    from vista_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vista_hip.h")).read()
    start = hdr.index("typedef struct VkFp8Args {") + len("typedef struct VkFp8Args {")
    body = re.sub(r"/\*.*?\*/", "", hdr[start:hdr.index("} VkFp8Args;")], flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = re.sub(r"^(const\s+)?(void|float|int32_t|int64_t)\s*\*?", "", decl.strip()).strip()
        if decl:
            names.append(decl.lstrip("*"))
    assert names == [f[0] for f in _lib.VkFp8Args._fields_], names


def test_fp8_conv_weight_packing_and_groupnorm_scale_bound_on_cpu():
    """Host side of the fp8 ResBlock path (no launch). (1) pack_conv3x3_fp8 / pack_conv_t3_fp8: K order [Cout][Cin/64][tap][64] in
    bytes, rows padded to 128, per-output-channel scales that reproduce the weights to e4m3 precision. (2) the scale rule of
    vk_groupnorm_silu_fp8 (include/vista_hip.h): with y = a_c x + b_c the folded GroupNorm and max|x| of the image, (max_c |a_c| max|x| +
    |b_c|) bounds |SiLU(y)| for every element -- checked against a brute-force maximum, with and without SiLU."""
    from vista_amd import ops
    g = torch.Generator().manual_seed(3)
    cout, cin = 40, 128
    w = torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5
    pw = ops.pack_conv3x3_fp8(w, torch.zeros(cout), device="cpu")
    assert pw.K == 9 * cin and pw.Kp % 128 == 0 and pw.wt.dtype == torch.uint8 and pw.wt.shape[1] == pw.Kp
    deq = pw.wt[:cout, :pw.K].view(torch.float8_e4m3fn).float() * pw.scale[:cout, None]
    back = deq.view(cout, cin // 64, 9, 64).permute(0, 1, 3, 2).reshape(cout, cin, 3, 3)       # -> [Cout][Cin][ky][kx]
    assert (back - w).abs().max() <= 2 ** -4 * w.abs().amax() * 1.01
    assert torch.equal(pw.wt[:cout, pw.K:], torch.zeros(cout, pw.Kp - pw.K, dtype=torch.uint8)), "the K padding must be zero bytes"
    wt3 = torch.randn(cout, 64, 3, 1, 1, generator=g)
    pt = ops.pack_conv_t3_fp8(wt3, None, device="cpu")
    deq = pt.wt[:cout, :pt.K].view(torch.float8_e4m3fn).float() * pt.scale[:cout, None]
    assert pt.K == 192 and pt.Kp == 256 and (deq.view(cout, 1, 3, 64).permute(0, 1, 3, 2).reshape(cout, 64, 3) - wt3[:, :, :, 0, 0]).abs().max() <= 0.07 * wt3.abs().max()
    with pytest.raises(ValueError):
        ops.pack_conv3x3_fp8(torch.randn(8, 48, 3, 3), None, device="cpu")  # Cin % 64

    n, S, C = 3, 50, 64
    x = torch.randn(n, S, C, generator=g) * torch.tensor([0.3, 2.0, 9.0])[:, None, None] + 1.0
    x[1, 7, 5] = 40.0                                                                           # an outlier sets max|x|
    gamma, beta = 1 + 0.5 * torch.randn(C, generator=g), 0.4 * torch.randn(C, generator=g)
    xg = x.view(n, S, 32, C // 32)
    mean = xg.mean((1, 3), keepdim=True)
    rstd = (xg.var((1, 3), unbiased=False, keepdim=True) + 1e-5).rsqrt()
    a = (gamma.view(1, 1, 32, -1) * rstd).expand(n, 1, 32, C // 32)
    b = beta.view(1, 1, 32, -1) - mean * a
    y = xg * a + b
    xmax = x.abs().amax((1, 2))
    bound = (a.abs() * xmax.view(n, 1, 1, 1) + b.abs()).amax((1, 2, 3))
    assert (y.abs().amax((1, 2, 3)) <= bound * 1.0001).all()
    assert (torch.nn.functional.silu(y).abs().amax((1, 2, 3)) <= torch.maximum(bound, torch.tensor(0.2785)) * 1.0001).all()
    assert (bound <= 8 * y.abs().amax((1, 2, 3))).all(), "the bound should stay within a few octaves of the data"


def test_schedules_guiders_scalings_match_reference_kats():
    from vista_amd.modules.diffusionmodules import denoiser_scaling, discretizer, guiders
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    disc = discretizer.EDMDiscretization(sigma_min=0.002, sigma_max=700.0, rho=7.0)
    for n in (10, 50):
        assert torch.equal(disc(n), torch.tensor(kat[f"edm_sigmas_{n}"])), "sigma schedule must be bit-identical to the reference's"
    assert torch.equal(disc(7, do_append_zero=False), torch.tensor(kat["edm_sigmas_7_noappend"]))
    assert torch.equal(disc(5, flip=True), torch.flip(disc(5), (0,)))
    scal = denoiser_scaling.VScalingWithEDMcNoise()
    for s, ref in kat["vscaling"].items():
        got = torch.tensor([float(v) for v in scal(torch.tensor(float(s)))])
        assert torch.allclose(got, torch.tensor(ref), rtol=1e-6, atol=1e-9)
    lin = guiders.LinearPredictionGuider(num_frames=25, max_scale=2.5, min_scale=1.0)
    tri = guiders.TrianglePredictionGuider(num_frames=25, max_scale=2.5, min_scale=1.0)
    assert torch.allclose(lin.scale[0], torch.tensor(kat["linear_guider_25"]))
    assert torch.allclose(tri.scale[0], torch.tensor(kat["triangle_guider_25"]), atol=1e-6)
    assert torch.allclose(guiders.VanillaCFG(2.5).frame_scales(25), torch.full((25,), 2.5))
    assert guiders.IdentityGuider().frame_scales(25) is None


def test_guider_prepare_inputs_layout():
    from vista_amd.modules.diffusionmodules import guiders
    x, s, m = torch.randn(3, 4, 2, 2), torch.ones(3), torch.tensor([1.0, 0, 0])
    c = {"vector": torch.randn(3, 8), "crossattn": torch.randn(3, 1, 16), "concat": torch.randn(3, 4, 2, 2)}
    uc = {k: torch.zeros_like(v) for k, v in c.items()}
    x2, s2, c2, m2 = guiders.VanillaCFG(2.5).prepare_inputs(x, s, c, m, uc)
    assert x2.shape[0] == 6 and torch.equal(x2[:3], x) and torch.equal(x2[3:], x)
    for k in c:  # the uncond half comes FIRST (guiders.py:32)
        assert torch.equal(c2[k][:3], uc[k]) and torch.equal(c2[k][3:], c[k])
    assert torch.equal(m2, torch.cat([m, m])) and torch.equal(s2, torch.cat([s, s]))
    xi, si, ci, mi = guiders.IdentityGuider().prepare_inputs(x, s, c, m, uc)
    assert xi is x and mi is m


def test_instantiate_from_config_maps_reference_targets():
    from vista_amd.modules.diffusionmodules import discretizer, guiders, sampling
    from vista_amd.util import append_dims, instantiate_from_config
    g = instantiate_from_config({"target": "vwm.modules.diffusionmodules.guiders.VanillaCFG", "params": {"scale": 2.5}})
    assert isinstance(g, guiders.VanillaCFG)
    s = sampling.EulerEDMSampler(num_steps=10, discretization_config={"target": "vwm.modules.diffusionmodules.discretizer.EDMDiscretization",
                                                                      "params": {"sigma_min": 0.002, "sigma_max": 700.0, "rho": 7.0}},
                                 guider_config={"target": "vwm.modules.diffusionmodules.guiders.TrianglePredictionGuider",
                                                "params": {"num_frames": 25, "max_scale": 2.5, "min_scale": 1.0}},
                                 s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0, verbose=False, device="cuda")
    assert isinstance(s.discretization, discretizer.EDMDiscretization) and isinstance(s.guider, guiders.TrianglePredictionGuider)
    assert len(s.host_sigmas()) == 11 and s.host_sigmas()[-1] == 0
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})
    with pytest.raises(ValueError):
        append_dims(torch.zeros(2, 2), 1)


def test_state_dict_contract_matches_reference():
    """Names and shapes of every VideoUNet tensor equal the reference's (digest recorded from the real VideoUNet)."""
    from vista_amd import synth
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    g = torch.load(os.path.join(GOLD, "unet_tiny_t5.pt"))
    net = VideoUNet(**unet_kwargs(64))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert len(shapes) == 1496 and synth.shapes_digest(shapes) == g["digest"]
    for k in ("time_embed.0.weight", "input_blocks.0.0.weight", "input_blocks.1.0.time_stack.in_layers.2.weight",
              "input_blocks.1.1.transformer_blocks.0.attn2.k_adapter_action_control.weight", "input_blocks.3.0.op.weight",
              "middle_block.1.time_stack.0.ff_in.net.0.proj.weight", "output_blocks.2.1.conv.weight", "output_blocks.5.2.conv.weight",
              "input_blocks.1.0.time_mixer.mix_factor", "input_blocks.1.1.time_pos_embed.2.bias", "out.2.weight", "label_emb.0.2.bias"):
        assert k in shapes, k
    # unexpected / missing keys surface through load_state_dict(strict=True)
    sd = synth.seeded_state_dict(shapes, 0)
    net.load_state_dict(sd, strict=True)
    sd.pop("out.2.weight")
    with pytest.raises(RuntimeError):
        net.load_state_dict(sd, strict=True)


def test_seeded_tensors_are_reproducible_and_nonzero():
    from vista_amd import synth
    a = synth.seeded_tensor("input_blocks.1.0.out_layers.3.weight", (64, 64, 3, 3), 0)
    b = synth.seeded_tensor("input_blocks.1.0.out_layers.3.weight", (64, 64, 3, 3), 0)
    assert torch.equal(a, b) and a.abs().max() > 0
    assert not torch.equal(a, synth.seeded_tensor("input_blocks.1.0.out_layers.3.weight", (64, 64, 3, 3), 1))
    w = synth.window_inputs(T=5, H=4, W=8, seed=3, trajectory=[0.5, 0, 1.0, 0, 1.5, 0.1, 2.0, 0.2])
    assert w["c"]["crossattn"].shape == (5, 1, 3456) and w["c"]["vector"].shape == (5, 768) and w["c"]["concat"].shape == (5, 4, 4, 8)
    assert w["uc"]["crossattn"].abs().max() == 0 and torch.equal(w["uc"]["vector"], w["c"]["vector"])
    assert w["cond_mask"].tolist() == [1, 0, 0, 0, 0]


def test_product_path_refuses_cpu():
    from vista_amd import ops
    from vista_amd._lib import VistaHipError
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    with pytest.raises(VistaHipError):
        ops.layernorm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64), torch.zeros(64))
    net = VideoUNet(**unet_kwargs(64))
    with pytest.raises(VistaHipError):
        net(torch.zeros(2, 8, 16, 32), timesteps=torch.zeros(2), context=torch.zeros(2, 1, 3456), y=torch.zeros(2, 768),
            cond_mask=torch.zeros(2), num_frames=2)
    # parameter containers carry no eager arithmetic
    with pytest.raises(RuntimeError):
        net.time_embed[0](torch.zeros(1, 64))


def test_weight_packing_layouts_on_cpu():
    """Host-side packing (no kernel launch): GEGLU fragment-local row order, conv tap-major K order, zero padding, fp8 scales."""
    from vista_amd import ops
    nout, K = 64, 128
    perm = ops.geglu_perm(nout)
    assert sorted(perm.tolist()) == list(range(2 * nout)), "a permutation of the 2*nout rows"
    blocks = perm.view(-1, 32)
    assert torch.equal(blocks[:, :16] + nout, blocks[:, 16:]), "every 32-row fragment = [16 value rows | their 16 gate rows]"
    assert torch.equal(blocks[:, 0], torch.arange(0, nout, 16))
    w = torch.randn(2 * nout, K)
    b = torch.randn(2 * nout)
    pg = ops.pack_geglu(w, b, device="cpu")
    assert pg.geglu and pg.N == 2 * nout and pg.K == K and pg.wt.shape[0] == 320  # rows padded so that every block tile (<= 320 wide) reads whole tiles
    assert torch.equal(pg.wt[:2 * nout].float(), w[perm].to(torch.bfloat16).float()) and pg.wt[2 * nout:].abs().sum() == 0
    assert torch.equal(pg.bias[:2 * nout], b[perm])
    # conv3x3: [Cout][Cin slab of 64][ky][kx][64] (all taps of one channel slab on consecutive K-steps), Cin zero-padded to 64
    wc = torch.randn(8, 3, 3, 3)
    pc = ops.pack_conv3x3(wc, None, device="cpu")
    assert pc.N == 8 and pc.K == 9 * 64
    v = pc.wt[:8].float().view(8, 3, 3, 64)  # one slab: [Cout][ky][kx][64]
    assert torch.equal(v[..., :3], wc.permute(0, 2, 3, 1).to(torch.bfloat16).float()) and v[..., 3:].abs().sum() == 0
    w2 = torch.randn(4, 128, 3, 3)           # two slabs
    p2 = ops.pack_conv3x3(w2, None, device="cpu").wt[:4].float().view(4, 2, 3, 3, 64)
    assert torch.equal(p2[:, 1, 2, 0], w2[:, 64:, 2, 0].to(torch.bfloat16).float())
    wt3 = torch.randn(4, 128, 3, 1, 1)
    p3 = ops.pack_conv_t3(wt3, None, device="cpu").wt[:4].float().view(4, 2, 3, 64)
    assert torch.equal(p3[:, 1, 0], wt3[:, 64:, 0, 0, 0].to(torch.bfloat16).float())
    # odd Cout is rounded up to the 4-column epilogue quad with zero rows / bias
    pl = ops.pack_linear(torch.randn(3, 64), torch.randn(3), device="cpu")
    assert pl.N == 4 and pl.wt[3:].abs().sum() == 0 and pl.bias[3:].abs().sum() == 0
    # fp8: per-output-channel scales reproduce the weights to e4m3 precision (3 mantissa bits -> <= 2^-4 relative per element)
    w8 = torch.randn(40, 320) * torch.logspace(-2, 1, 40)[:, None]
    p8 = ops.pack_linear_fp8(w8, None, device="cpu")
    assert p8.Kp == 384 and p8.K == 320 and p8.wt.dtype == torch.uint8 and p8.wt[:, 320:].sum() == 0
    deq = p8.wt[:40, :320].view(torch.float8_e4m3fn).float() * p8.scale[:40, None]
    assert ((deq - w8).abs() <= w8.abs() * 2 ** -4 + p8.scale[:40, None] * 2 ** -9 + 1e-12).all()
    with pytest.raises(ValueError):
        ops.geglu_perm(24)


def test_ff_fused_out_projection_layout_on_cpu():
    """ops.ff_out_layout / pack_ff_out (the out-projection operand of vk_ff_fused_bf16, include/vista_hip.h): K permuted inside every 16-group to
    the order in which a lane of the in-projection's 32x32 MFMA accumulator holds the hidden units, then chunk-major [K/32][N][32].
    Emulates the kernel's dataflow on the host: lane (l31, lh) of a GEGLU fragment holds hidden units 8 g + 4 lh + e (g = 0, 1; e = 0..3) as the
    8 k-slots i = 4 g + e of MFMA half lh; the weight fragment's k-slot (lh, i) must then hold the SAME hidden unit."""
    from vista_amd import ops
    N, K = 320, 128
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K)
    lay = ops.ff_out_layout(w)
    assert lay.shape == (K // 32 * N, 32)
    lay3 = lay.view(K // 32, N, 32)
    for chunk in range(K // 32):
        for s in range(2):                      # k-substep (16-group) of the chunk
            for lh in range(2):
                for i in range(8):
                    hidden = 32 * chunk + 16 * s + 8 * (i >> 2) + 4 * lh + (i & 3)   # what the activation lane puts into k-slot (lh, i)
                    assert torch.equal(lay3[chunk, :, 16 * s + 8 * lh + i], w[:, hidden])
    # a permutation of the columns: every weight appears exactly once
    assert sorted(lay.flatten().tolist()) == sorted(w.flatten().tolist())
    pw = ops.pack_ff_out(torch.randn(N, 1280), torch.randn(N), device="cpu")
    assert pw.ffout and pw.N == N and pw.K == 1280 and pw.wt.dtype == torch.bfloat16 and pw.wt.numel() == N * 1280 and pw.bias.shape == (N,)
    pin = ops.pack_geglu(torch.randn(2560, 320), torch.randn(2560), device="cpu")
    assert ops.ff_fused_ok(pin, pw)
    assert not ops.ff_fused_ok(ops.pack_geglu(torch.randn(5120, 640), torch.randn(5120), device="cpu"), pw)     # only the width-320 level
    with pytest.raises(ValueError):
        ops.ff_out_layout(torch.zeros(320, 48))


def test_layernorm_fold_packing_algebra_on_cpu():
    """pack_*(..., ln=norm) must satisfy  LN(x) W^T + b == rstd * (x W'^T - mean * colsum) + bias'  (include/vista_hip.h, VkGemmDesc.ln_*),
    with colsum taken from the bf16-ROUNDED W' so that the mean term cancels exactly in the kernel's epilogue. Pure host check (no launch):
    the GEMM is emulated in fp32 from the packed operands; the GPU tests check the kernels against the same identity."""
    import torch.nn.functional as F
    from vista_amd import ops

    class Norm:
        pass
    g = torch.Generator().manual_seed(0)
    C, N, M = 128, 96, 40
    norm = Norm()
    norm.weight, norm.bias, norm.eps = 1.0 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g), 1e-5
    w, b = torch.randn(N, C, generator=g) * C ** -0.5, torch.randn(N, generator=g)
    x = (torch.randn(M, C, generator=g) * 2.0 + 5.0 * torch.randn(M, 1, generator=g)).to(torch.bfloat16).float()  # large row means

    def emulate(pw, rows):
        acc = rows @ pw.wt[:pw.N].float().t()
        mean = rows.mean(1, keepdim=True)
        rstd = (rows.var(1, unbiased=False, keepdim=True) + pw.ln_eps).rsqrt()
        return rstd * (acc - mean * pw.colsum[:pw.N]) + pw.bias[:pw.N]
    pw = ops.pack_linear(w, b, device="cpu", ln=norm)
    assert pw.colsum is not None and torch.equal(pw.colsum, pw.wt.float().sum(1)) and pw.ln_eps == 1e-5
    ref = F.layer_norm(x, (C,), norm.weight, norm.bias, norm.eps) @ w.t() + b
    got = emulate(pw, x)
    assert (got - ref).abs().max() <= 2e-2 * ref.abs().max(), "only the bf16 rounding of gamma (.) W separates the two forms"
    # exactness of the cancellation: a constant added to a row changes nothing (LayerNorm is shift invariant), to fp32 round-off
    shifted = emulate(pw, x + 64.0)
    assert (shifted - got).abs().max() <= 2e-3 * ref.abs().max()
    # GEGLU: same identity in the packed (value | gate) row order; cat: q|k|v share the fold
    wg, bg = torch.randn(2 * 64, C, generator=g) * C ** -0.5, torch.randn(2 * 64, generator=g)
    pg = ops.pack_geglu(wg, bg, device="cpu", ln=norm)
    perm = ops.geglu_perm(64)
    refg = (F.layer_norm(x, (C,), norm.weight, norm.bias, norm.eps) @ wg.t() + bg)[:, perm]
    assert (emulate(pg, x) - refg).abs().max() <= 2e-2 * refg.abs().max()
    pc = ops.pack_linear_cat([w, w * 0.5], device="cpu", ln=norm)
    assert pc.N == 2 * N and pc.bias is not None and torch.allclose(pc.bias[N:2 * N], (w * 0.5) @ norm.bias, atol=1e-6)
    assert ops.pack_linear(w, b, device="cpu").colsum is None


def test_video_unet_survives_deepcopy_and_pickle():
    """the per-thread table of the batched emb_layers projection must not make the module un-copyable (EMA copies, torch.save)."""
    import copy
    import pickle
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    net = VideoUNet(**unet_kwargs(64))
    twin = copy.deepcopy(net)
    assert twin._emb_tls is not net._emb_tls and all(m._emb_src[0] is twin._emb_tls for m in twin._emb_blocks)
    again = pickle.loads(pickle.dumps(net))
    assert all(m._emb_src[0] is again._emb_tls for m in again._emb_blocks)
    assert [k for k in again.state_dict()] == [k for k in net.state_dict()]


def test_gelu_logistic_quintic_formula():
    """csrc/common.h gelu_erf_f: x * Phi(x) with Phi as a logistic of an odd quintic. The same fp32 arithmetic here against the erf
    form (F.gelu, the reference's GEGLU, attention.py:92): |error| <= 2.6e-5 for every x, no NaN / inf at the extremes, and the
    constants in the header are the ones evaluated."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "vista_amd", "csrc", "common.h")).read()
    body = src[src.index("float gelu_erf_f(float x)"):]
    c2, c1, c0 = (float(v) for v in re.search(r"fmaf\(u, fmaf\(u, ([-0-9.e]+)f, ([-0-9.e]+)f\), ([-0-9.e]+)f\)", body).groups())
    clamp = float(re.search(r"fminf\(x \* x, ([0-9.]+)f\)", body).group(1))
    x = torch.cat([torch.linspace(-30, 30, 1_000_001), torch.tensor([-1e4, 1e4, -3e38, 3e38, 0.0])]).float()
    u = torch.minimum(x * x, torch.tensor(clamp))
    t = torch.addcmul(torch.tensor(c0), u, torch.addcmul(torch.tensor(c1), u, torch.tensor(c2)))
    got = x * (1.0 / (1.0 + torch.exp2(x * t)))
    ref = torch.nn.functional.gelu(x.double()).float()
    assert torch.isfinite(got).all()
    fin = x.abs() <= 1e4
    assert (got[fin] - ref[fin]).abs().max().item() <= 2.6e-5
    assert got[-2].item() == ref[-2].item() and got[-3].item() == 0.0   # +3e38 -> x, -3e38 -> -0


def test_packable_cache_key_sees_replaced_parameters_and_inference_tensors():
    """the pack cache must notice a REPLACED nn.Parameter (load_state_dict(assign=True), `m.weight = nn.Parameter()`),
    whose old object would keep its version counter for ever, and must not trip over inference-mode tensors (no `_version`)."""
    import torch.nn as nn
    from vista_amd.modules.attention import Packable

    class Stub(nn.Module, Packable):
        _pack_device_types = ("cpu", "cuda")  # test-only: the product classes pack on the GPU only

        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 4)
            self.packs = 0

        def _pack(self, dev):
            self.packs += 1
            return {"w": self.lin.weight.detach().clone()}

    m = Stub()
    pk0 = m.packed()
    assert m.packed() is pk0 and m.packs == 1
    # 1. in-place update: version counter
    with torch.no_grad():
        m.lin.weight.add_(1.0)
    pk1 = m.packed()
    assert pk1 is not pk0 and torch.equal(pk1["w"], m.lin.weight)
    # 2. the Parameter OBJECT is replaced: the cached tensor's version never changes
    m.lin.weight = nn.Parameter(torch.full((4, 4), 7.0))
    pk2 = m.packed()
    assert pk2 is not pk1 and float(pk2["w"][0, 0]) == 7.0
    # 3. load_state_dict(assign=True) on a PARENT container replaces the Parameters without touching the child's load_state_dict
    holder = nn.ModuleDict({"inner": m})
    sd = {k: torch.full_like(v, 3.0) for k, v in holder.state_dict().items()}
    holder.load_state_dict(sd, assign=True)
    pk3 = m.packed()
    assert pk3 is not pk2 and float(pk3["w"][0, 0]) == 3.0
    assert m.packed() is pk3
    # 4. inference-mode tensors have no version counter
    with torch.inference_mode():
        m2 = Stub()
        m2.lin.weight = nn.Parameter(torch.ones(4, 4), requires_grad=False)
    assert m2.lin.weight.is_inference()
    assert m2.packed() is m2.packed()
    # 5. a load through a VideoUNet drops every pack under it (post-hook), whatever the keys say
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    from vista_amd.config import unet_kwargs
    net = VideoUNet(**unet_kwargs(model_channels=64, channel_mult=[1], attention_resolutions=[1], num_res_blocks=1))
    packables = [x for x in net.modules() if isinstance(x, Packable)]
    for x in packables:
        x._pk = {"stale": True}
    from vista_amd.modules import attention as att
    gen0 = att.pack_generation()
    net.load_state_dict(net.state_dict())
    assert all(x._pk is None for x in packables)
    # 6. ... and moves the pack generation, which the hipGraph cache of the forward keys on (a captured graph holds raw pointers into
    #    the packs; version counters cannot see p.data writes / inference-tensor loads, invalidate_packed() is the one door they all use)
    assert att.pack_generation() >= gen0 + len(packables)
    gen1 = att.pack_generation()
    att.invalidate_packed(net)
    assert att.pack_generation() == gen1 + len(packables)


def test_config_fallback_equals_shipped_yaml_and_import_is_lazy():
    """`import vista_amd.config` neither needs PyYAML nor the sibling configs/ directory."""
    import vista_amd.config as cfg
    assert cfg.load_config()["model"]["params"]["network_config"]["params"] == cfg._FALLBACK_UNET_KWARGS
    assert cfg.VISTA_UNET_KWARGS == cfg._FALLBACK_UNET_KWARGS
    saved, cfg._UNET_KWARGS = cfg._UNET_KWARGS, None
    real = cfg.CONFIG_PATH
    try:
        cfg.CONFIG_PATH = "/nonexistent/vista.yaml"
        cfg.load_config.__defaults__ = (cfg.CONFIG_PATH,)
        assert cfg.unet_kwargs(64)["model_channels"] == 64 and cfg.unet_kwargs()["context_dim"] == 1024
    finally:
        cfg.CONFIG_PATH = real
        cfg.load_config.__defaults__ = (real,)
        cfg._UNET_KWARGS = saved


def test_rowstat_parts_answer_matches_the_launch_that_follows():
    """vk_gemm_rowstat_parts is asked BEFORE the caller can set rowstat_out (the buffer is sized from the answer): it must describe the launch
    with the pointer set. Round 4 regression: the streaming K = 320 kernel, which is not chosen for row-sum emitting launches, answered the
    query (1 slab) and the tiled kernel then wrote its 2 slabs into a buffer of one."""
    import ctypes as C
    from vista_amd import _lib, ops
    lib = _lib.load()
    one = C.c_void_p(4096)
    slabs = {1: (128, 2), 2: (128, 4), 3: (256, 4), 4: (320, 2), 5: (160, 1), 7: (320, 2)}   # tile variant -> (tile width, wave columns)
    for M, N, K in ((50 * 9216, 320, 320), (50 * 9216, 320, 1280), (50 * 2304, 640, 640), (50 * 576, 1280, 1280), (7 * 9216, 320, 320), (1440, 320, 320)):
        for res in (False, True):
            d = _lib.VkGemmDesc()
            d.A = d.Wt = d.out = one
            d.M, d.N, d.K, d.lda, d.ldc, d.alpha = M, N, K, K, N, 1.0
            d.amode, d.epi = ops.AMODE_DENSE, ops.EPI_LINEAR
            d.splitk_ws, d.splitk_ws_bytes = one, 160 << 20
            if res:
                d.res1, d.ld_res1 = one, N
            parts = lib.vk_gemm_rowstat_parts(C.byref(d))      # as ops._gemm asks: pointer not set yet
            d.rowstat_out = one
            cfg, ks = divmod(lib.vk_gemm_tile_choice(C.byref(d)), 16)
            assert ks in (1, 2) and cfg in slabs, (M, N, K, res, cfg, ks)
            bn, wn = slabs[cfg]
            assert parts == (N + bn - 1) // bn * wn, (M, N, K, res, cfg, parts)


def test_gemm_launch_rules_are_pinned():
    """vk_gemm_tile_choice: the launcher's (block tile, K slices) for the BASELINE shapes and for one rank of an 8-GPU run, as host arithmetic
    (no GPU). Pins what the same-box sweeps of rounds 1-3 chose (profiles/r03_tile5_sweep.txt, r03_gemm_sweep_rank7*.jsonl,
    r03_geglu_two_per_cu_experiment.txt): a changed rule must change this table knowingly."""
    import ctypes as C
    from vista_amd import _lib, ops
    lib = _lib.load()
    one = C.c_void_p(16)  # never dereferenced by the query: any non-null pointer passes validate()

    def choice(M, N, K, epi=ops.EPI_LINEAR, amode=ops.AMODE_DENSE, stats=False, ws=True, Cin=0, **kw):
        d = _lib.VkGemmDesc()
        d.A = d.Wt = d.out = one
        d.M, d.N, d.K, d.lda, d.ldc = M, N, K, K, (N // 2 if epi == ops.EPI_GEGLU else N)
        d.amode, d.epi, d.alpha = amode, epi, 1.0
        if stats:
            d.rowstat_out = one
        if ws:
            d.splitk_ws, d.splitk_ws_bytes = one, 160 << 20
        if amode != ops.AMODE_DENSE:
            d.Cin, d.H, d.Wd, d.Hout, d.Wout, d.stride, d.ups = Cin, kw["H"], kw["W"], kw["H"], kw["W"], 1, 1
        for k, v in kw.items():
            if k not in ("H", "W"):
                setattr(d, k, v)
        rc = lib.vk_gemm_tile_choice(C.byref(d))
        assert rc > 0, f"vk_gemm_tile_choice({M},{N},{K}) -> {rc}"
        return divmod(rc, 16)

    full, L1, L2 = 50 * 9216, 50 * 2304, 50 * 576
    # one GPU, 50 images
    assert choice(full, 320, 320, stats=True) == (5, 1)         # K = N projections: 128x160, two workgroups per CU
    assert choice(L1, 640, 640, stats=True) == (7, 1)           # round 4: the pipelined 256x320 kernel overtook the 128x160 tile at K = N = 640
    assert choice(L2, 1280, 1280, stats=True) == (7, 1)         # K = N = 1280: back on the big tile (7 = its eight-wave pipelined kernel, round 4)
    assert choice(full, 960, 320) == (7, 1)                     # q|k|v: N = 3K
    assert choice(full, 320, 1280, stats=True) == (7, 1)        # FeedForward out: K = 4N
    assert choice(full, 2560, 320, epi=ops.EPI_GEGLU) == (3, 1)
    assert choice(L1, 5120, 640, epi=ops.EPI_GEGLU) == (7, 1)   # K >= 640: the pipelined 256x320 kernel
    assert choice(L2, 10240, 1280, epi=ops.EPI_GEGLU) == (7, 1)
    assert choice(full, 320, 2880, amode=ops.AMODE_CONV3X3, Cin=320, H=72, W=128) == (7, 1)
    assert choice(full, 320, 2880, amode=ops.AMODE_CONV3X3, Cin=320, H=72, W=128, tile_cfg=4) == (4, 1)   # forced: the sixteen-wave kernel
    # one rank of 8 (7 images)
    r0, r1, r2 = 7 * 9216, 7 * 2304, 7 * 576
    assert choice(r0, 320, 1280, stats=True) == (7, 1)          # 252 tiles: one full round
    assert choice(r1, 640, 640, stats=True) == (5, 1)           # 126 tiles of 256x320 would fill half the chip
    assert choice(r1, 640, 2560, stats=True) == (5, 1)
    assert choice(r2, 1280, 5120, stats=True) == (5, 1)
    assert choice(r2, 10240, 1280, epi=ops.EPI_GEGLU) == (7, 1)  # 512 tiles of 256x320 = 2 rounds instead of 640 = 3
    cfg, ks = choice(r2, 1280, 11520, amode=ops.AMODE_CONV3X3, Cin=1280, H=18, W=32)
    assert cfg == 7 and ks >= 2, "small-M deep-K convolutions run split-K on the big tile (its pipelined kernel since round 4)"
    assert choice(r2, 1280, 11520, amode=ops.AMODE_CONV3X3, Cin=1280, H=18, W=32, ws=False) [1] == 1  # no workspace, no split


def test_gemm_tail_split_rule_is_pinned():
    """vk_gemm_tail_split: where vk_gemm_bf16 cuts a one-tile-per-workgroup launch into whole rounds + a 128x160 tail (round 5; measured without
    gain and therefore OFF unless VISTA_GEMM_TAIL=<percent> or tile_cfg bit 6 asks for it -- which this test does). Host arithmetic only. Level 0 of the BASELINE window is 460800 rows = 1800 row tiles of 256: 7.03 rounds at N = 320 (tail = the last 8 row tiles), 21.09 at N = 960."""
    import ctypes as C
    from vista_amd import _lib, ops
    lib = _lib.load()
    one = C.c_void_p(16)

    def split(M, N, K, amode=ops.AMODE_DENSE, stats=False, Cin=0, H=0, W=0, **kw):
        d = _lib.VkGemmDesc()
        d.A = d.Wt = d.out = one
        d.M, d.N, d.K, d.lda, d.ldc = M, N, K, K, N
        d.amode, d.epi, d.alpha = amode, ops.EPI_LINEAR, 1.0
        d.splitk_ws, d.splitk_ws_bytes = one, 160 << 20
        d.tile_cfg = 64   # ask for the rule
        if stats:
            d.rowstat_out = one
        if amode == ops.AMODE_CONV3X3:
            d.Cin, d.H, d.Wd, d.Hout, d.Wout, d.stride, d.ups = Cin, H, W, H, W, 1, 1
        if amode == ops.AMODE_TEMPORAL3:
            d.Cin, d.T, d.S = Cin, 25, H * W
        for k, v in kw.items():
            setattr(d, k, v)
        return lib.vk_gemm_tail_split(C.byref(d))

    full, L1, L2 = 50 * 9216, 50 * 2304, 50 * 576
    assert split(full, 320, 2880, amode=ops.AMODE_CONV3X3, Cin=320, H=72, W=128) == 1792 * 256       # 7 full rounds, then 8 row tiles
    assert split(full, 320, 960, amode=ops.AMODE_TEMPORAL3, Cin=320, H=72, W=128) == 1792 * 256
    assert split(full, 960, 320) == 1792 * 256                                                         # q|k|v: 5400 tiles = 21 rounds + 24
    assert split(full, 320, 1280, stats=True) == 1792 * 256
    assert split(full, 320, 320, stats=True) == 0                                                      # 128x160 tiles already (cfg 5)
    assert split(L1, 640, 5760, amode=ops.AMODE_CONV3X3, Cin=640, H=36, W=64) == 0                     # 900 tiles = 3.52 rounds: the tail is half a round
    assert split(L2, 1280, 1280, stats=True) == 0                                                      # 452 tiles = 1.77 rounds
    assert split(full, 320, 2880, amode=ops.AMODE_CONV3X3, Cin=320, H=72, W=128, tile_cfg=7) == 0      # a forced variant is never split
    assert split(full, 320, 2880, amode=ops.AMODE_CONV3X3, Cin=320, H=72, W=128, m_begin=256, m_end=full) == 256 + 1792 * 256   # 1799 tiles of a row range: 7 rounds + 7
    assert split(full, 320, 2880, amode=ops.AMODE_CONV3X3, Cin=320, H=72, W=128, m_begin=5, m_end=4) < 0      # bad range
    assert split(full, 960, 320, tile_cfg=0) == 0                                                      # not asked for: off by default


def test_gnstat_fit_rule_is_pinned():
    """vk_gemm_gnstat_fit (ABI v6): which launches can emit the GroupNorm statistics of their output from the epilogue (VkGemmDesc.gnstat_out) --
    the ResBlock convolutions of levels 0-2 at the BASELINE window (the pipelined 256x320 kernel, one launch) -- and which cannot and keep the
    statistics pass: split-K launches (level 3, one rank's deeper levels), fp32 output, two residuals, 144-row images, the tail-split option.
    Host arithmetic only; the answer does not depend on gnstat_out itself (the buffer is sized from it before the pointer exists)."""
    import ctypes as C
    from vista_amd import _lib, ops
    lib = _lib.load()
    one = C.c_void_p(16)

    def fit(n_img, H, W, Cin, N, amode=ops.AMODE_CONV3X3, **kw):
        d = _lib.VkGemmDesc()
        d.A = d.Wt = d.out = one
        S = H * W
        d.M, d.N, d.lda, d.ldc = n_img * S, N, Cin, N
        d.amode, d.epi, d.alpha, d.Cin = amode, ops.EPI_LINEAR, 1.0, Cin
        d.splitk_ws, d.splitk_ws_bytes = one, 160 << 20
        if amode == ops.AMODE_CONV3X3:
            d.K, d.H, d.Wd, d.Hout, d.Wout, d.stride, d.ups = 9 * Cin, H, W, H, W, 1, 1
        elif amode == ops.AMODE_TEMPORAL3:
            d.K, d.T, d.S = 3 * Cin, 25 if n_img % 25 == 0 else n_img, S
        else:
            d.K = Cin
        d.gn_rows = S
        for k, v in kw.items():
            setattr(d, k, v)
        a = lib.vk_gemm_gnstat_fit(C.byref(d))
        d.gnstat_out = one
        assert lib.vk_gemm_gnstat_fit(C.byref(d)) == a   # the pointer itself changes nothing
        return a

    t3 = ops.AMODE_TEMPORAL3
    assert fit(50, 72, 128, 320, 320) == 50 * 9216 // 64           # level-0 ResBlock convolution
    assert fit(50, 72, 128, 960, 320) == 7200                      # first convolution of an output block (concat input)
    assert fit(50, 72, 128, 320, 320, res1=one, ld_res1=320) == 7200
    assert fit(50, 36, 64, 640, 640) == 1800 and fit(50, 18, 32, 1280, 1280) == 450
    assert fit(50, 72, 128, 320, 320, amode=t3) == 7200 and fit(50, 36, 64, 640, 640, amode=t3) == 1800 and fit(50, 18, 32, 1280, 1280, amode=t3) == 450
    assert fit(50, 72, 128, 320, 320, amode=t3, res2=one, ld_res2=320, alpha=0.4, beta=1.0) == 7200   # temporal conv2 + blend
    assert fit(7, 72, 128, 320, 320) == 7 * 144                    # one rank of an 8-GPU run: level 0 still one launch of 252 tiles
    assert fit(50, 9, 16, 1280, 1280) == 0                         # level 3: 144 rows per image, and a split-K launch
    assert fit(7, 36, 64, 640, 640) == 0 and fit(7, 18, 32, 1280, 1280) == 0   # one rank's deeper levels: split-K
    assert fit(50, 72, 128, 320, 320, out_f32=1) == 0
    assert fit(50, 72, 128, 320, 320, res1=one, ld_res1=320, res2=one, ld_res2=320) == 0
    assert fit(50, 72, 128, 320, 320, act=1) == 0 and fit(50, 72, 128, 320, 320, rowstat_out=one) == 0
    assert fit(50, 72, 128, 320, 960) == 0                         # 30-channel groups do not tile 160-column wave tiles
    assert fit(50, 72, 128, 320, 320, amode=t3, halo_prev=one) == 0            # halo frames: the sixteen-wave kernel
    assert fit(50, 72, 128, 320, 320, tile_cfg=64) == 0            # the tail-split option would hand the last rows to another kernel
    assert fit(50, 72, 128, 320, 320, tile_cfg=4) == 0 and fit(3, 16, 16, 320, 320, tile_cfg=7) == 12  # forced variants
    assert fit(50, 72, 128, 320, 320, m_begin=256, m_end=512) == 0
    assert fit(50, 72, 128, 320, 320, amode=ops.AMODE_DENSE) == 0  # (dense producers are not built with the emitting bodies)
    assert fit(50, 72, 128, 320, 320, gn_rows=0) == 0 and fit(50, 72, 128, 320, 320, gn_rows=9216 + 64) == 0
