"""GroupNorm statistics emitted by the producing convolution's epilogue (VkGemmDesc.gnstat_out, ABI v6; vista_amd/csrc/gemm_common.h).

The reference's GroupNorm32 after a ResBlock convolution (openaimodel.py:195-199,227-234, video_model.py:38-52) is two passes over the tensor
in the HIP path (statistics, apply); with gnstat_out the statistics come out of the convolution's epilogue instead. Checked here:
  * the convolution's output is bitwise what it is without the option;
  * the folded sums equal those of vk_groupnorm_stats_bf16 on the same tensor (fp32 sums in another order: 2e-5 relative of the scale
    sqrt(count * sum of squares) -- the statistics pass is the reference, it is itself pinned to torch fp32 by test_kernels_gpu.py::test_groupnorm);
  * groupnorm(x, gn=partials) equals groupnorm(x) to bf16 rounding, for per-image (2-D) and per-clip (5-D, frames_per_group = T) norms;
  * run to run the partials are bitwise equal (fixed summation order, no atomics);
  * launches that cannot emit them (rows per image not a multiple of 64, fp32 output, split-K, two residuals) leave the holder empty and the
    norm falls back to its own statistics pass; setting the pointer on such a launch by hand is refused, not ignored;
  * a VideoResBlock + SpatialVideoTransformer pair gives the same result with ops.GN_EPI on and off.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def _ops():
    from vista_amd import ops
    return ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(BF16).cuda()


def stats_pass(ops, x, fpg):
    """[groups][64] raw sums of vk_groupnorm_stats_bf16 (the three-launch GroupNorm's statistics)."""
    from vista_amd import _lib
    n_img, S, Cc = x.shape
    sums = torch.empty((n_img // fpg) * 64, dtype=F32, device=x.device)
    part = torch.empty(n_img * ((S + 31) // 32) * 64, dtype=F32, device=x.device)
    _lib.check(_lib.load().vk_groupnorm_stats_bf16(ops._p(x), ops._p(sums), ops._p(part), n_img, S, Cc, fpg, ops._stream()), "vk_groupnorm_stats_bf16")
    return sums.view(-1, 64)


def fold(ops, gn, n_img, fpg):
    from vista_amd import _lib
    sums = torch.empty((n_img // fpg) * 64, dtype=F32, device=gn.t.device)
    part = gn.t.clone()   # (the fold of large groups works in place)
    _lib.check(_lib.load().vk_groupnorm_finalize_partials(ops._p(part), ops._p(sums), n_img, gn.nchunks, fpg, ops._stream()), "vk_groupnorm_finalize_partials")
    return sums.view(-1, 64)


def sums_close(a, b, count, name):
    """a, b: [groups][64] = 32 sums | 32 sums of squares. fp32 sums of `count` values in two different orders."""
    a, b = a.double().cpu(), b.double().cpu()
    q = b[:, 32:]
    scale_s = (count * q).sqrt() + 1e-6          # |sum| <= sqrt(count * sum of squares)
    es = ((a[:, :32] - b[:, :32]).abs() / scale_s).max().item()
    eq = ((a[:, 32:] - q).abs() / (q + 1e-6)).max().item()
    assert es <= 2e-5 and eq <= 2e-5, f"{name}: sums differ by {es:.3g} (of sqrt(n * sumsq)), sums of squares by {eq:.3g} relative"


CASES = [
    # kind, C, n_img, H, W, forced tile (0 = the launcher's own choice)
    ("conv+emb", 320, 3, 16, 16, 7),
    ("conv+res", 320, 3, 16, 16, 7),
    ("conv+emb", 640, 2, 16, 8, 7),
    ("conv+res", 640, 2, 16, 8, 7),
    ("conv+emb", 1280, 2, 16, 8, 7),     # (with a per-image row vector a 256-row tile may span at most 4 images: 64-row images go to the sixteen-wave kernel)
    ("conv+res", 1280, 3, 8, 8, 7),      # M = 192: one partial 256-row tile, the last 64-row block of it outside the problem
    ("t3+emb", 320, 6, 16, 16, 7),
    ("t3+blend", 320, 6, 16, 16, 7),
    ("t3+emb", 640, 4, 16, 8, 7),
    ("t3+blend", 1280, 4, 8, 8, 7),
    ("conv+emb", 320, 50, 32, 32, 0),    # 200 tiles of 256x320: the launcher picks the pipelined kernel by itself
    ("t3+blend", 320, 50, 32, 32, 0),
    ("conv+res", 640, 25, 48, 32, 0),    # 150 row tiles x 2 column tiles
]


def run_case(ops, kind, Cc, n, H, W, gn):
    S = H * W
    x = rnd(n, S, Cc, seed=3)
    rv = rnd(n, Cc, seed=4).float()
    res = rnd(n, S, Cc, seed=5)
    if kind.startswith("conv"):
        pw = ops.pack_conv3x3(rnd(Cc, Cc, 3, 3, scale=(9 * Cc) ** -0.5, seed=6), rnd(Cc, seed=7).float())
        if kind == "conv+emb":
            out, _, _ = ops.conv3x3(x, pw, n, H, W, rowvec=rv, gn=gn)
        else:
            out, _, _ = ops.conv3x3(x, pw, n, H, W, res1=res, gn=gn)
        return out
    T = n // 2
    pw = ops.pack_conv_t3(rnd(Cc, Cc, 3, 1, 1, scale=(3 * Cc) ** -0.5, seed=8), rnd(Cc, seed=9).float())
    if kind == "t3+emb":
        return ops.conv_t3(x, pw, T, S, rowvec=rv, gn=gn)
    return ops.conv_t3(x, pw, T, S, alpha=0.37, res2=res, beta=1.0, gn=gn)


@pytest.mark.parametrize("kind,Cc,n,H,W,force", CASES)
def test_conv_epilogue_gn_partials(kind, Cc, n, H, W, force):
    ops = _ops()
    assert ops.GN_EPI, "VISTA_GN_EPI=0 in the environment: this file tests the emitting path"
    S = H * W
    old = ops.TILE_CFG
    ops.TILE_CFG = force
    try:
        base = run_case(ops, kind, Cc, n, H, W, None)
        gn = ops.GnPartials()
        out = run_case(ops, kind, Cc, n, H, W, gn)
        gn2 = ops.GnPartials()
        out2 = run_case(ops, kind, Cc, n, H, W, gn2)
    finally:
        ops.TILE_CFG = old
    assert gn.t is not None and gn.nchunks == S // 64 and gn.t.numel() == n * S, f"{kind} C={Cc}: the launch did not emit its statistics"
    assert torch.equal(out, base), f"{kind} C={Cc}: the emitting epilogue changed the convolution's output"
    assert torch.equal(gn.t, gn2.t) and torch.equal(out, out2), "partials differ run to run"
    assert torch.isfinite(gn.t).all()
    out = out.view(n, S, Cc)
    for fpg in (1, n if kind.startswith("conv") else n // 2):
        sums_close(fold(ops, gn, n, fpg), stats_pass(ops, out, fpg), (Cc // 32) * S * fpg, f"{kind} C={Cc} n={n} fpg={fpg}")
    # the norm itself, both ways
    gamma, beta = 1.0 + 0.2 * rnd(Cc, seed=10).float(), 0.2 * rnd(Cc, seed=11).float()
    fpg = 1 if kind.startswith("conv") else n // 2
    ref = ops.groupnorm(out, gamma, beta, 1e-5, True, frames_per_group=fpg)
    got = ops.groupnorm(out, gamma, beta, 1e-5, True, frames_per_group=fpg, gn=gn)
    assert gn.t is None, "the partials are consumed by the norm"
    d = (got.float() - ref.float()).abs()
    # mean / rstd agree to ~1e-6: an output moves by at most one bf16 rounding step, and only where it sat on a rounding boundary
    assert d.max().item() <= 2.0 ** -6 * max(1.0, ref.float().abs().max().item()) and d.mean().item() <= 2e-4, \
        f"{kind} C={Cc}: GroupNorm from epilogue statistics differs from the three-launch form: max {d.max().item():.3g}, mean {d.mean().item():.3g}"


@pytest.mark.parametrize("why", ["rows_per_image", "f32_out", "two_residuals", "split_k", "switch_off"])
def test_unfit_launches_leave_the_holder_empty(why):
    ops = _ops()
    Cc, n, H, W = 320, 2, 16, 16
    kw, force = {}, 7
    if why == "rows_per_image":
        H, W = 10, 10
    elif why == "split_k":
        Cc, n, H, W, force = 1280, 16, 8, 8, 0   # 4 row tiles x 4 column tiles, K = 11520: the launcher runs 8 K slices of the pipelined kernel
    S = H * W
    x = rnd(n, S, Cc, seed=1)
    pw = ops.pack_conv3x3(rnd(Cc, Cc, 3, 3, scale=(9 * Cc) ** -0.5, seed=2), rnd(Cc, seed=3).float())
    if why == "f32_out":
        kw["out_f32"] = True
    if why == "two_residuals":
        kw.update(res1=rnd(n, S, Cc, seed=4), res2=rnd(n, S, Cc, seed=5), beta=0.5)
    old, old_sw = ops.TILE_CFG, ops.GN_EPI
    ops.TILE_CFG = force
    if why == "switch_off":
        ops.GN_EPI = 0
    try:
        gn = ops.GnPartials()
        out, _, _ = ops.conv3x3(x, pw, n, H, W, gn=gn, **kw)
        base, _, _ = ops.conv3x3(x, pw, n, H, W, **kw)
    finally:
        ops.TILE_CFG, ops.GN_EPI = old, old_sw
    assert gn.t is None, f"{why}: this launch cannot emit GroupNorm statistics"
    assert torch.equal(out, base)
    if why not in ("f32_out",):
        gamma, beta = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        o3 = out.view(n, S, Cc)
        assert torch.equal(ops.groupnorm(o3, gamma, beta, 1e-5, True, gn=gn), ops.groupnorm(o3, gamma, beta, 1e-5, True))   # falls back: the same launches


def test_pointer_on_an_unfit_launch_is_refused():
    """vk_gemm_bf16 with gnstat_out set on a problem vk_gemm_gnstat_fit answers 0 for returns an error instead of dropping the statistics."""
    ops = _ops()
    from vista_amd import _lib
    from vista_amd._lib import VkGemmDesc
    Cc, n, H, W = 320, 2, 10, 10
    x = rnd(n, H * W, Cc)
    pw = ops.pack_conv3x3(rnd(Cc, Cc, 3, 3, scale=0.02, seed=2), None)
    out = torch.empty((n * H * W, Cc), dtype=BF16, device="cuda")
    buf = torch.zeros(64 * 8, dtype=F32, device="cuda")
    d = VkGemmDesc()
    d.A, d.lda = ops._p(x), Cc
    d.amode, d.epi = ops.AMODE_CONV3X3, ops.EPI_LINEAR
    d.H, d.Wd, d.Cin, d.Hout, d.Wout, d.stride, d.ups = H, W, Cc, H, W, 1, 1
    ops._fill_epilogue(d, pw, out, n * H * W, None, H * W, None, None, 1.0, 0.0)
    d.gn_rows = H * W
    lib = _lib.load()
    assert lib.vk_gemm_gnstat_fit(C.byref(d)) == 0
    d.gnstat_out = ops._p(buf)
    assert lib.vk_gemm_bf16(C.byref(d), ops._stream()) < 0
    torch.cuda.synchronize()
    assert not buf.any()


def test_video_resblock_and_transformer_same_result_with_and_without_epilogue_statistics():
    """The block pair of input_blocks.1 at a reduced spatial size: every GroupNorm but the very first takes its statistics from an epilogue
    (ops.GN_EPI = 1) or from its own pass (0). Same weights, same input. The first norm that is fed from partials sees a bitwise equal input and
    must agree to rounding-boundary flips (measured: 0.002 % of its outputs move by one bf16 step, 2.8e-6 relative); after that the two runs are
    two bf16 computations that differ in a handful of roundings, and the block amplifies ANY such difference -- flipping the last mantissa bit
    of 0.1 % of the block's input moves its output by 4.7e-3 relative (tools/gnstat_block_diag.py) -- so the block outputs are held to twice
    that sensitivity, measured in the same test (3.8e-3 against 4.7e-3 when written)."""
    ops = _ops()
    from vista_amd.config import unet_kwargs
    from vista_amd.modules.diffusionmodules.video_model import VideoUNet
    torch.manual_seed(0)
    with torch.device("cuda"):
        net = VideoUNet(**unet_kwargs(320))
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():   # (every tensor of the pair non-zero: the default init zeroes the second convolutions and proj_out)
        for name, p in net.input_blocks[1].named_parameters():
            if name.endswith("mix_factor"):
                p.normal_(0, 0.5, generator=g)
            elif p.dim() >= 2:
                p.normal_(0, float(p[0].numel()) ** -0.5, generator=g)
            elif name.endswith(".weight"):
                p.normal_(1.0, 0.1, generator=g)
            else:
                p.normal_(0, 0.1, generator=g)
    net = net.eval()
    T, H, W = 4, 16, 16
    n = 2 * T
    x = rnd(n, H * W, 320, seed=20)
    emb = rnd(n, 1280, scale=0.7, seed=21)
    ctx = rnd(n, 3456, seed=22)
    frame_idx = torch.arange(T, dtype=F32, device="cuda").repeat(2)
    xp = x.clone()
    idx = torch.randperm(x.numel(), generator=torch.Generator().manual_seed(3))[: x.numel() // 1000].cuda()
    flat = xp.view(torch.int16).view(-1)
    flat[idx] = flat[idx] ^ 1   # the last mantissa bit of 0.1 % of the input elements
    blk = net.input_blocks[1]
    real_gn, log = ops.groupnorm, []

    def spy(xx, *a, **k):
        had = k.get("gn") is not None and k["gn"].t is not None
        out = real_gn(xx, *a, **k)
        log.append((had, out.clone()))
        return out
    old = (ops.GN_EPI, ops.TILE_CFG)
    runs = {}
    try:
        ops.groupnorm = spy
        ops.TILE_CFG = 7   # (small problem: force the pipelined kernel so that the emitting epilogues run)
        for key, sw, inp in (("epi", 1, x), ("pass", 0, x), ("perturbed", 0, xp)):
            ops.GN_EPI = sw
            log.clear()
            with torch.no_grad():
                o, _, _ = blk(inp, emb, ctx, frame_idx, T, H, W)
            runs[key] = (o.float(), list(log))
    finally:
        ops.groupnorm = real_gn
        ops.GN_EPI, ops.TILE_CFG = old
    rel = lambda a, b: ((a.float() - b.float()).pow(2).sum().sqrt() / b.float().pow(2).sum().sqrt()).item()  # noqa: E731
    fed = [had for had, _ in runs["epi"][1]]
    assert fed == [False, True, True, True, True], f"norms fed from epilogue partials: {fed} (in_layers, out_layers, time_stack in / out, transformer norm)"
    assert not any(had for had, _ in runs["pass"][1])
    first = rel(runs["epi"][1][1][1], runs["pass"][1][1][1])
    assert first <= 1e-4, f"the first GroupNorm fed from partials differs by {first:.3g} relative from the three-launch form on the same input"
    d_paths, d_sens = rel(runs["epi"][0], runs["pass"][0]), rel(runs["perturbed"][0], runs["pass"][0])
    assert d_paths <= 2.0 * d_sens + 1e-4, f"block output moved by {d_paths:.3g} between the two statistics paths; a one-ulp perturbation of 0.1 % of the input moves it by {d_sens:.3g}"


@pytest.mark.parametrize("n,S,Cc,fpg", [(4, 9216, 320, 1), (3, 2304, 640, 1), (5, 576, 1280, 1), (5, 144, 1280, 5), (10, 576, 1280, 5),
                                        (6, 130, 320, 3), (2, 72, 2560, 1), (50, 144, 1280, 25)])
def test_groupnorm_apply_folds_its_own_partials_bitwise(n, S, Cc, fpg):
    """ABI v7: the apply pass that folds the stage-1 slots itself (vk_groupnorm_silu_bf16 when frames_per_group * chunks <= vk_groupnorm_fold_max(),
    vk_groupnorm_apply_partials_bf16 on a convolution epilogue's slots) against the explicit stats -> finalize -> apply launches: bitwise equal."""
    from vista_amd import _lib
    ops = _ops()
    lib = _lib.load()
    assert lib.vk_groupnorm_fold_max() == 256
    x = rnd(n, S, Cc, seed=3)
    g, b = torch.randn(Cc, device="cuda"), torch.randn(Cc, device="cuda")
    got = ops.groupnorm(x, g, b, 1e-5, True, frames_per_group=fpg)          # folds when it may
    sums = stats_pass(ops, x, fpg).reshape(-1)
    want = torch.empty_like(x)
    count = float(Cc // 32) * S * fpg
    _lib.check(lib.vk_groupnorm_apply_bf16(ops._p(x), ops._p(want), ops._p(g), ops._p(b), ops._p(sums), n, S, Cc, fpg, count, 1e-5, 1, ops._stream()),
               "vk_groupnorm_apply_bf16")
    assert torch.equal(got, want)
    if S % 64 == 0:   # slots in the epilogue's geometry (one per 64 rows), here produced by summing the tensor on the host side of the test
        xf = x.float().view(n, S // 64, 64, 32, Cc // 32)
        part = torch.cat([xf.sum((2, 4)), (xf * xf).sum((2, 4))], -1).contiguous()   # [n][S/64][64]
        nch = S // 64
        out2 = torch.empty_like(x)
        rc = lib.vk_groupnorm_apply_partials_bf16(ops._p(x), ops._p(out2), ops._p(g), ops._p(b), ops._p(part), n, S, Cc, nch, fpg, count, 1e-5, 1, ops._stream())
        if fpg * nch > 256:
            assert rc == -22
        else:
            assert rc == 0
            s2 = torch.empty((n // fpg) * 64, dtype=F32, device="cuda")
            _lib.check(lib.vk_groupnorm_finalize_partials(ops._p(part.clone()), ops._p(s2), n, nch, fpg, ops._stream()), "vk_groupnorm_finalize_partials")
            want2 = torch.empty_like(x)
            _lib.check(lib.vk_groupnorm_apply_bf16(ops._p(x), ops._p(want2), ops._p(g), ops._p(b), ops._p(s2), n, S, Cc, fpg, count, 1e-5, 1, ops._stream()),
                       "vk_groupnorm_apply_bf16")
            assert torch.equal(out2, want2)
