"""fp8 (OCP e4m3) GEMM path of BASELINE config 5 on the MI355X.

Two kinds of checks, with their tolerances stated:
* kernel exactness: against an fp32 matmul of the DEQUANTISED operands (the very fp8 values the kernel multiplies) the only
  difference is fp32 accumulation order and the bf16 output rounding -> same tolerance as the bf16 GEMM tests;
* quantisation error (re-stated tolerance for config 5): per-row activation scales x per-channel weight scales, e4m3 has 3
  mantissa bits (relative step 2^-3 at the top of a binade, 2^-4 mean) -> relative L2 error of one projection vs the
  unquantised fp32 result <= 4.5e-2 on Gaussian data (measured ~3.6e-2 = sqrt(2) * 2.6e-2 for two quantised operands).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _ops():
    from vista_amd import ops
    return ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def deq(q_u8, scale):
    return q_u8.cpu().view(torch.float8_e4m3fn).float() * scale.cpu()[:, None]


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).pow(2).sum() / b.pow(2).sum()).sqrt().item()


@pytest.mark.parametrize("M,K", [(300, 320), (5, 64), (1000, 1280), (64, 5120), (257, 2432)])
def test_quantize_rows(M, K):
    ops = _ops()
    x = (rnd(M, K) * torch.logspace(-2, 2, M)[:, None]).to(BF16).cuda()  # rows of very different magnitude
    x[0] = 0                                                               # an all-zero row must not divide by zero
    q, s = ops.quantize_rows_fp8(x)
    xf = x.float().cpu()
    want_s = (xf.abs().amax(1) / 448.0).masked_fill(xf.abs().amax(1) == 0, 1.0)
    assert torch.allclose(s.cpu(), want_s, rtol=1e-6, atol=0)
    want_q = (xf / want_s[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    got = q.cpu().view(torch.float8_e4m3fn)
    # the kernel multiplies by 1/scale instead of dividing: a value on a rounding boundary may land one code apart
    diff = (got.float() - want_q.float()).abs()
    assert (diff > 0).float().mean().item() < 2e-3 and (diff <= want_q.float().abs() * 0.13 + 2 ** -9).all()
    assert torch.isfinite(got.float()).all()
    assert rel_l2(deq(q, s), xf) < 4e-2


@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (4608, 640, 1280), (1000, 960, 64), (128, 128, 128), (777, 1280, 2432), (260, 4, 320)])
@pytest.mark.parametrize("cfg", [0, 1, 3, 4])
def test_linear_fp8_exact_against_dequantised_operands(M, N, K, cfg):
    ops = _ops()
    x = rnd(M, K).to(BF16).cuda()
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    b = rnd(N, seed=2)
    pw = ops.pack_linear_fp8(w, b)
    xq, xs = ops.quantize_rows_fp8(x)
    r1 = rnd(M, N, seed=3).to(BF16).cuda()
    ops.TILE_CFG = cfg
    try:
        out = ops.linear_fp8(xq, xs, pw, res1=r1, alpha=0.5)
        out32 = ops.linear_fp8(xq, xs, pw, out_f32=True)
    finally:
        ops.TILE_CFG = 0
    wd = pw.wt[:N, :K].cpu().view(torch.float8_e4m3fn).float() * pw.scale[:N].cpu()[:, None]
    ref = deq(xq, xs) @ wd.t() + b
    n_out = (N + 3) // 4 * 4
    assert out32.shape == (M, n_out) and out.shape == (M, n_out)
    err = (out32[:, :N].cpu() - ref).abs().max().item()
    assert err <= 2e-3 * ref.pow(2).mean().sqrt().item() + 1e-5, f"fp8 GEMM f32 out: max err {err}"
    ref2 = 0.5 * (ref + r1.float().cpu()[:, :N])
    e2 = (out[:, :N].float().cpu() - ref2).abs()
    assert (e2 <= 2e-2 * ref2.pow(2).mean().sqrt() + 1.6e-2 * ref2.abs()).all()
    # and the re-stated config-5 tolerance against the UNQUANTISED projection
    full = x.float().cpu() @ w.t() + b
    assert rel_l2(out32[:, :N], full) <= 4.5e-2


def test_geglu_fp8():
    ops = _ops()
    M, K, nout = 520, 320, 1280
    x = rnd(M, K).to(BF16).cuda()
    w = rnd(2 * nout, K, scale=K ** -0.5, seed=1)
    b = rnd(2 * nout, seed=2)
    pw = ops.pack_geglu_fp8(w, b)
    xq, xs = ops.quantize_rows_fp8(x)
    out = ops.linear_fp8(xq, xs, pw)
    # dequantised reference in the ORIGINAL (value | gate) row order
    perm = ops.geglu_perm(nout)
    wd = torch.empty(2 * nout, K)
    wd[perm] = pw.wt[:2 * nout, :K].cpu().view(torch.float8_e4m3fn).float() * pw.scale[:2 * nout].cpu()[:, None]
    h = deq(xq, xs) @ wd.t() + b
    ref = h[:, :nout] * torch.nn.functional.gelu(h[:, nout:])
    e = (out.float().cpu() - ref).abs()
    assert out.shape == (M, nout) and (e <= 2e-2 * ref.pow(2).mean().sqrt() + 1.6e-2 * ref.abs()).all()


def _mx_deq(q_u8, s_u8):
    """MX fp8 -> f32: e4m3 codes x 2^(e - 127) per 32 consecutive columns."""
    v = q_u8.cpu().view(torch.float8_e4m3fn).float()
    sc = torch.exp2(s_u8.cpu().float() - 127.0)
    return v * sc.repeat_interleave(32, 1)


@pytest.mark.parametrize("rows,C", [(1000, 320), (257, 640), (129, 1280), (64, 64)])
def test_layernorm_quant_fp8(rows, C):
    """LayerNorm + per-row quantisation in one kernel == vk_layernorm's function followed by vk_quantize_rows_fp8's."""
    ops = _ops()
    x = (rnd(rows, C) * torch.logspace(-1, 1, rows)[:, None] + 0.7).to(BF16).cuda()
    norm = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        norm.weight.copy_(rnd(C, seed=4) * 0.3 + 1.0)
        norm.bias.copy_(rnd(C, seed=5) * 0.2)
    q, s = ops.layernorm_quant_fp8(x, norm)
    y = torch.nn.functional.layer_norm(x.float(), (C,), norm.weight, norm.bias, norm.eps).cpu()
    want_s = y.abs().amax(1) / 448.0
    assert torch.allclose(s.cpu(), want_s, rtol=2e-5, atol=0)
    got = q.cpu().view(torch.float8_e4m3fn).float()
    assert torch.isfinite(got).all()
    want_q = (y / want_s[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    diff = (got - want_q).abs()
    assert (diff > 0).float().mean().item() < 5e-3 and (diff <= want_q.abs() * 0.13 + 2 ** -9).all()
    assert rel_l2(got * s.cpu()[:, None], y) < 4e-2


@pytest.mark.parametrize("cfg", [0, 1, 3])
@pytest.mark.parametrize("M,K,nout", [(520, 320, 1280), (4608, 640, 2560), (130, 64, 256)])
def test_geglu_fp8_mx_output(M, K, nout, cfg):
    """GEGLU epilogue quantising its own output to MX fp8: per (row, 32-column block) the scale is the smallest power of two with
    max|h| / 2^e < 448, and every element is the e4m3 rounding of h / 2^e (<= 2^-4 relative, or half a subnormal step 2^-10 * 2^e)."""
    ops = _ops()
    x = rnd(M, K).to(BF16).cuda()
    w = rnd(2 * nout, K, scale=K ** -0.5, seed=1)
    b = rnd(2 * nout, seed=2)
    pw = ops.pack_geglu_fp8(w, b)
    xq, xs = ops.quantize_rows_fp8(x)
    ops.TILE_CFG = cfg
    try:
        h8, hs = ops.linear_fp8(xq, xs, pw, mx_out=True)
        hb = ops.linear_fp8(xq, xs, pw)  # the bf16-output kernel on the same operands
    finally:
        ops.TILE_CFG = 0
    assert h8.shape == (M, nout) and h8.dtype == torch.uint8 and hs.shape == (M, nout // 32)
    perm = ops.geglu_perm(nout)
    wd = torch.empty(2 * nout, K)
    wd[perm] = pw.wt[:2 * nout, :K].cpu().view(torch.float8_e4m3fn).float() * pw.scale[:2 * nout].cpu()[:, None]
    h = deq(xq, xs) @ wd.t() + b
    ref = h[:, :nout] * torch.nn.functional.gelu(h[:, nout:])
    codes = h8.cpu().view(torch.float8_e4m3fn).float()
    assert torch.isfinite(codes).all()
    cmax = codes.abs().view(M, nout // 32, 32).amax(2)
    assert (cmax <= 448).all() and (cmax >= 208).all(), "block scale is not the tightest power of two"  # 224 less one e4m3 step
    two_e = torch.exp2(hs.cpu().float() - 127.0).repeat_interleave(32, 1)
    e = (_mx_deq(h8, hs) - ref).abs()
    assert (e <= (2 ** -4) * 1.02 * ref.abs() + two_e * 2 ** -10 + 2e-3 * ref.pow(2).mean().sqrt()).all()
    assert rel_l2(_mx_deq(h8, hs), hb) < 4e-2


@pytest.mark.parametrize("cfg", [0, 1, 3, 4])
@pytest.mark.parametrize("M,N,K", [(300, 320, 1280), (4608, 640, 2560), (777, 1280, 5120), (130, 64, 256)])
def test_linear_fp8_mx_activations(M, N, K, cfg):
    """Out-projection consuming MX block-scaled activations (scales applied inside v_mfma_scale_f32_32x32x64_f8f6f4): exact against
    the dequantised operands, for block scales spread over 2^-8 .. 2^6 so that a wrong block / lane / op_sel mapping cannot hide."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 100).clamp(-448, 448).to(torch.float8_e4m3fn)
    a8 = a.view(torch.uint8).cuda()
    sc = torch.randint(119, 134, (M, K // 32), generator=g, dtype=torch.uint8).cuda()
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    b = rnd(N, seed=2)
    pw = ops.pack_linear_fp8(w, b)
    r1 = rnd(M, N, seed=3).to(BF16).cuda()
    ops.TILE_CFG = cfg
    try:
        out, st = ops.linear_fp8(a8, None, pw, a_mx=sc, res1=r1, emit_stats=True)
    finally:
        ops.TILE_CFG = 0
    wd = pw.wt[:N, :K].cpu().view(torch.float8_e4m3fn).float() * pw.scale[:N].cpu()[:, None]
    ref = _mx_deq(a8, sc) @ wd.t() + b + r1.float().cpu()
    e = (out[:, :N].float().cpu() - ref).abs()
    assert (e <= 2e-2 * ref.pow(2).mean().sqrt() + 1.6e-2 * ref.abs()).all(), f"max err {e.max().item()} rms {ref.pow(2).mean().sqrt().item()}"
    o = out[:, :N].float()
    tot = st.t.sum(0)
    assert torch.allclose(tot[:, 0], o.sum(1), rtol=1e-3, atol=1e-2 * o.abs().mean().item() * N ** 0.5)
    assert torch.allclose(tot[:, 1], o.pow(2).sum(1), rtol=1e-3)
    with pytest.raises(ValueError):
        ops.linear_fp8(a8, torch.ones(M, device="cuda"), pw, a_mx=sc)  # one kind of activation scale


@pytest.mark.parametrize("M,C", [(1000, 320), (2304 + 77, 640), (300, 1280)])
def test_qkv_projection_with_mx_fp8_q_and_k(M, C):
    """Fused LayerNorm-folded q|k|v projection whose q | k columns leave as MX fp8 (VkGemmDesc.mx8_*): the v block is BITWISE the plain
    GEMM's v columns; the q | k codes are the e4m3 rounding of the plain GEMM's fp32 value under the tightest power-of-two block scale
    (checked against the bf16 output of the plain GEMM: one extra bf16 rounding 2^-9 on the comparison side)."""
    ops = _ops()
    x = (rnd(M, C) * torch.logspace(-0.5, 0.5, M)[:, None] + 0.3).to(BF16).cuda()
    norm = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        norm.weight.copy_(rnd(C, seed=4) * 0.3 + 1.0)
        norm.bias.copy_(rnd(C, seed=5) * 0.2)
    ws = [rnd(C, C, scale=C ** -0.5, seed=10 + i).cuda() for i in range(3)]
    pw = ops.pack_linear_cat(ws, ln=norm)
    st = ops.rowstats(x)
    plain = ops.linear(x, pw, ln=st)
    v, q8, qs = ops.linear(x, pw, ln=st, mx8_cols=2 * C)
    assert v.shape == (M, C) and q8.shape == (M, 2 * C) and q8.dtype == torch.uint8 and qs.shape == (M, 2 * C // 32)
    assert torch.equal(v, plain[:, 2 * C:]), "the bf16 v block must not depend on the q|k output format"
    ref = plain[:, :2 * C].float().cpu()
    codes = q8.cpu().view(torch.float8_e4m3fn).float()
    assert torch.isfinite(codes).all()
    cmax = codes.abs().view(M, 2 * C // 32, 32).amax(2)
    assert (cmax <= 448).all() and (cmax >= 208).all(), "block scale is not the tightest power of two"
    two_e = torch.exp2(qs.cpu().float() - 127.0).repeat_interleave(32, 1)
    e = (_mx_deq(q8, qs) - ref).abs()
    assert (e <= ((2 ** -4) * 1.02 + 2 ** -8) * ref.abs() + two_e * 2 ** -10 + 1e-3 * ref.pow(2).mean().sqrt()).all()
    assert rel_l2(_mx_deq(q8, qs), ref) < 3e-2
    with pytest.raises(ValueError):
        ops.linear(x, pw, ln=st, mx8_cols=2 * C, emit_stats=True)
    with pytest.raises(ValueError):
        ops.linear(x, pw, ln=st, mx8_cols=96)


def _mx_quant_ref(t):
    """(M, C) f32 -> (codes u8, scales u8): the MX rule of the kernels (tightest power-of-two block scale, RNE e4m3)."""
    M, C = t.shape
    b = t.view(M, C // 32, 32)
    amax = b.abs().amax(2).clamp_min(2.0 ** -120)
    e = torch.ceil(torch.log2(amax / 448.0))
    e = torch.where(amax / torch.exp2(e) >= 448.0, e + 1, e)
    q = (b / torch.exp2(e)[..., None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(M, C).view(torch.uint8), (e + 127).to(torch.uint8)


@pytest.mark.parametrize("n_img,heads,S", [(3, 5, 144), (2, 10, 576), (2, 5, 2304), (1, 5, 9216), (2, 5, 4104), (3, 20, 200)])
def test_attn_spatial_fp8qk_exact_against_dequantised_operands(n_img, heads, S):
    """Spatial attention with the score product in MX fp8 (v_mfma_scale_f32_32x32x64_f8f6f4), q / k given as strided column blocks with
    block scales spread over 2^-3 .. 2^3: against fp32 softmax(q k^T) v on the DEQUANTISED q / k the only differences are the bf16 P
    and V of the second product -> the bf16 kernel's tolerance. S = 4104 and 200 are ragged (not multiples of the 64-key tile)."""
    ops = _ops()
    M, C = n_img * S, heads * 64
    g = torch.Generator().manual_seed(S + heads)
    qk = (torch.randn(M, 2 * C, generator=g) * 120).clamp(-448, 448).to(torch.float8_e4m3fn)
    qk8 = qk.view(torch.uint8).cuda()
    sc = torch.randint(124, 131, (M, 2 * C // 32), generator=g, dtype=torch.uint8).cuda()
    v = rnd(M, C, seed=3).to(BF16).cuda()
    nb = C // 32
    d = _mx_deq(qk8, sc).cuda()
    q = d[:, :C].view(n_img, S, heads, 64).transpose(1, 2)
    k = d[:, C:].view(n_img, S, heads, 64).transpose(1, 2)
    # logits of standard deviation 3 (heavy-tailed: the block-scale products span 2^-6 .. 2^6). Much sharper softmaxes (one-hot up to exact
    # ties) only measure the fp32 rounding of logits of magnitude 10^3 on both sides: tools/fp8attn_dbg.py sweeps the sharpness
    scale = 3.0 / (q[0, 0] @ k[0, 0].T).std().item()
    got = ops.attn_spatial_fp8qk(qk8[:, :C], qk8[:, C:], sc[:, :nb], sc[:, nb:], v, n_img, heads, S, scale=scale)
    vv = v.float().view(n_img, S, heads, 64).transpose(1, 2)
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, vv, scale=scale).transpose(1, 2).reshape(M, C)
    assert torch.isfinite(got).all()
    e = (got.float() - ref).abs()
    rms = ref.pow(2).mean().sqrt()
    assert (e <= 2e-2 * rms + 1.6e-2 * ref.abs()).all(), f"max err {e.max().item()} rms {rms.item()}"
    assert rel_l2(got, ref) < 6e-3
    # scale = 0: "q already carries scale * log2 e" (what BasicTransformerBlock packs): fold a power of two into q's block scales instead and
    # compare with softmax(q k^T * 2^sh * ln 2); S >= 4096 runs the zero-base 512-row kernel, shorter ones the general kernel with scale_log2 = 1
    import math
    sh = round(math.log2(scale * 1.4426950408889634))
    sc2 = sc.clone()
    sc2[:, :nb] = (sc2[:, :nb].to(torch.int16) + sh).clamp(1, 254).to(torch.uint8)
    got0 = ops.attn_spatial_fp8qk(qk8[:, :C], qk8[:, C:], sc2[:, :nb], sc2[:, nb:], v, n_img, heads, S, scale=0.0)
    ref0 = torch.nn.functional.scaled_dot_product_attention(q, k, vv, scale=2.0 ** sh * math.log(2.0)).transpose(1, 2).reshape(M, C)
    e0 = (got0.float() - ref0).abs()
    rms0 = ref0.pow(2).mean().sqrt()
    assert torch.isfinite(got0).all() and (e0 <= 2e-2 * rms0 + 1.6e-2 * ref0.abs()).all(), f"pre-scaled: max err {e0.max().item()} rms {rms0.item()}"
    assert rel_l2(got0, ref0) < 6e-3
    # MX fp8 output of the same kernel: the e4m3 rounding of the bf16 output under the tightest block scale, pad scale bytes = 2^0
    o8, osc = ops.attn_spatial_fp8qk(qk8[:, :C], qk8[:, C:], sc[:, :nb], sc[:, nb:], v, n_img, heads, S, scale=scale, mx_out=True)
    assert o8.shape == (M, C) and osc.shape[0] == M and osc.shape[1] % 4 == 0 and osc.shape[1] >= nb
    assert (osc[:, nb:] == 127).all()
    codes = o8.cpu().view(torch.float8_e4m3fn).float()
    assert torch.isfinite(codes).all()
    cmax = codes.abs().view(M, nb, 32).amax(2)
    assert (cmax <= 448).all() and (cmax >= 208).all()
    two_e = torch.exp2(osc[:, :nb].cpu().float() - 127.0).repeat_interleave(32, 1)
    e8 = (_mx_deq(o8, osc[:, :nb]) - got.float().cpu()).abs()
    assert (e8 <= ((2 ** -4) * 1.02 + 2 ** -8) * got.float().cpu().abs() + two_e * 2 ** -10 + 1e-3 * rms.item()).all()


@pytest.mark.parametrize("n_img,heads,S", [(2, 5, 2304), (3, 10, 576)])
def test_attention_branch_fp8_quantisation_error(n_img, heads, S):
    """Re-stated tolerance of config 5's attention branch: q and k each carry one e4m3 rounding (relative 2.6e-2 rms), so a logit -- a 64-term
    dot product -- carries an ABSOLUTE error of ~3.6e-2 x (its own standard deviation), and the softmax turns an absolute logit error eps into a
    relative weight error eps: the branch error scales with how sharp the softmax is. Unit-variance logits (this test): expected ~3.6e-2 of the
    deviation of the output from the mean of v -> rel-L2 <= 5e-2; at logit deviation 2.9 the same kernel measures 8.7e-2 (first version of
    this test), in the network (block parity, tests/test_blocks_gpu.py) the fp8 score product moves the block error by < 1e-3. V and P stay bf16."""
    ops = _ops()
    M, C = n_img * S, heads * 64
    q = rnd(M, C, seed=1)
    k = rnd(M, C, seed=2)
    v = rnd(M, C, seed=3).to(BF16).cuda()
    q8, qs = _mx_quant_ref(q)
    k8, ks = _mx_quant_ref(k)
    got = ops.attn_spatial_fp8qk(q8.cuda(), k8.cuda(), qs.cuda(), ks.cuda(), v, n_img, heads, S)
    sh = lambda t: t.float().cuda().view(n_img, S, heads, 64).transpose(1, 2)  # noqa: E731
    ref = torch.nn.functional.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(M, C)
    r = rel_l2(got, ref)
    print(f"fp8 QK^T attention S={S}: rel-L2 {r:.3e}")
    assert r < 5e-2


def test_feedforward_config5_keeps_the_fused_bf16_kernel_at_width_320():
    """Round 4: where the fused bf16 FeedForward kernel runs (level 0, width 320) config 5 keeps it -- it is faster than the fp8 pair there
    (1.28 vs 1.38 ms at the BASELINE shape) and 15x closer to fp32 (tools/fp8_vs_bf16_probe.py): the switch must not change that block's result."""
    ops = _ops()
    from vista_amd.modules import attention
    ff = attention.FeedForward(320, glu=True).cuda()
    norm = torch.nn.LayerNorm(320).cuda()
    x = (rnd(1200, 320) + 0.3).to(BF16).cuda()
    pw_in, st = ff.pack_in_folded(norm, x.device), ops.rowstats(x)
    plain = ff.forward_folded(x, st, pw_in, norm, res1=x)
    attention.FP8["feedforward"] = True
    try:
        assert not ff._fp8()
        cfg5 = ff.forward_folded(x, st, pw_in, norm, res1=x)
        wide = attention.FeedForward(640, glu=True).cuda()
        assert wide._fp8()
    finally:
        attention.FP8["feedforward"] = False
    assert torch.equal(plain, cfg5)


@pytest.mark.parametrize("dim,M", [(640, 1200), (1280, 300)])
def test_feedforward_fp8_no_quantisation_pass(dim, M):
    """FeedForward of config 5 (levels 1 / 2: width 640 / 1280): LN+quant -> fp8 GEGLU (MX out) -> fp8 out-projection (MX in), vs the fp32
    function. Re-stated tolerance on the branch output (residual excluded): four e4m3 roundings in series -- x and W1 (3.6e-2 together,
    module docstring), the value and the gate both carry that error into value*gelu(gate) (another ~3.6e-2), then h and W2 (2.6e-2 each):
    sqrt(2 * 3.6^2 + 2 * 2.6^2) e-2 = 6.3e-2 expected, 6.5e-2 measured -> rel-L2 <= 8e-2."""
    ops = _ops()
    from vista_amd.modules import attention
    ff = attention.FeedForward(dim, glu=True).cuda()
    norm = torch.nn.LayerNorm(dim).cuda()
    x = (rnd(M, dim) + 0.3).to(BF16).cuda()
    attention.FP8["feedforward"] = True
    try:
        out, st = ff.forward_folded(x, None, None, norm, emit_stats=True)
    finally:
        attention.FP8["feedforward"] = False
    with torch.no_grad():
        lin = torch.nn.functional.linear
        y = norm(x.float())
        a, g = lin(y, ff.net[0].proj.weight, ff.net[0].proj.bias).chunk(2, -1)
        ref = lin(a * torch.nn.functional.gelu(g), ff.net[2].weight, ff.net[2].bias)
    assert rel_l2(out, ref) <= 8e-2
    o = out.float()
    assert torch.allclose(st.t.sum(0)[:, 0], o.sum(1), rtol=1e-3, atol=1e-2 * o.abs().mean().item() * dim ** 0.5)


def _gn_ref(x, gamma, beta, eps, silu, fpg):
    n, S, Cc = x.shape
    xg = x.float().view(n // fpg, fpg * S, Cc).permute(0, 2, 1)
    y = torch.nn.functional.group_norm(xg, 32, gamma, beta, eps)
    if silu:
        y = torch.nn.functional.silu(y)
    return y.permute(0, 2, 1).reshape(n, S, Cc)


@pytest.mark.parametrize("n,S,C,C2,fpg,silu", [(4, 144, 320, 0, 1, True), (6, 100, 64, 0, 1, False), (6, 64, 192, 0, 3, True), (2, 576, 640, 320, 1, True),
                                               (50, 16, 1280, 1280, 1, True), (10, 300, 320, 0, 5, True)])
def test_groupnorm_fp8(n, S, C, C2, fpg, silu):
    """GroupNorm[+SiLU] with e4m3 output and one scale per image group: the scale is an upper BOUND taken from the statistics pass
    (no code may exceed 448 / be NaN), not looser than 8x the true maximum on this data (16x for a concat of two differently scaled
    tensors: max|x| is taken over both), and every element is the e4m3 rounding of the exact result at that scale (<= 2^-4 relative, or
    half a subnormal step)."""
    ops = _ops()
    x = (rnd(n, S, C).float() * torch.logspace(-1, 0.5, n)[:, None, None] * 1.5 + 0.7).to(BF16).cuda()
    x2 = (rnd(n, S, C2, seed=7).float() * 0.8 - 0.3).to(BF16).cuda() if C2 else None
    Ct = C + C2
    gamma = (rnd(Ct, seed=1) * 0.4 + 1.0).cuda()
    beta = (rnd(Ct, seed=2) * 0.3).cuda()
    y8, sc = ops.groupnorm_fp8(x, gamma, beta, 1e-5, silu, fpg, x2=x2)
    xin = x if x2 is None else torch.cat([x, x2], 2)
    ref = _gn_ref(xin, gamma, beta, 1e-5, silu, fpg).cpu()
    assert y8.shape == (n, S, Ct) and y8.dtype == torch.uint8 and sc.shape == (n // fpg,)
    codes = y8.cpu().view(torch.float8_e4m3fn).float()
    assert torch.isfinite(codes).all()
    s_img = sc.cpu().repeat_interleave(fpg)[:, None, None]
    true_amax = ref.reshape(n // fpg, -1).abs().amax(1)
    assert (sc.cpu() * 448.0 >= true_amax * 0.999).all(), "the scale must bound the data"
    assert (sc.cpu() * 448.0 <= true_amax * (16.0 if C2 else 8.0) + 0.3).all(), "the bound is uselessly loose"
    e = (codes * s_img - ref).abs()
    assert (e <= (2 ** -4) * 1.03 * ref.abs() + s_img * 2 ** -10 + 3e-3 * ref.abs().mean()).all()
    assert rel_l2(codes * s_img, ref) < 4e-2


def _deq_conv_weight(pw, cout, cin, taps):
    """packed e4m3 [Cout][Cin/64][tap][64] -> f32 [Cout][tap][Cin]"""
    w = pw.wt[:cout, :taps * cin].cpu().view(torch.float8_e4m3fn).float() * pw.scale[:cout].cpu()[:, None]
    return w.view(cout, cin // 64, taps, 64).permute(0, 2, 1, 3).reshape(cout, taps, cin)


def _rand_fp8(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * 60).clamp(-448, 448).to(torch.float8_e4m3fn)


@pytest.mark.parametrize("n,H,W,cin,cout,fps", [(3, 9, 16, 320, 320, 1), (2, 18, 32, 640, 640, 1), (2, 12, 20, 960, 320, 1), (2, 6, 10, 64, 128, 1),
                                                (4, 7, 9, 128, 320, 2), (1, 40, 33, 320, 640, 1)])
def test_conv3x3_fp8_exact_against_dequantised_operands(n, H, W, cin, cout, fps):
    """Implicit-GEMM 3x3 convolution on e4m3 activations with one scale per image (per `fps` images): exact against F.conv2d of the
    dequantised operands (odd and even (slab, tap) unit counts, Cout with and without the 320-wide tile, zero padding at the borders)."""
    ops = _ops()
    a = _rand_fp8((n, H * W, cin), n + H + cin)
    sc = (torch.rand(n // fps, generator=torch.Generator().manual_seed(5)) * 0.05 + 0.01)
    w = rnd(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=1)
    b = rnd(cout, seed=2)
    pw = ops.pack_conv3x3_fp8(w, b)
    r1 = rnd(n, H * W, cout, seed=3).to(BF16).cuda()
    rv = rnd(n, cout, seed=4).cuda()
    out = ops.conv3x3_fp8(a.view(torch.uint8).cuda(), sc.cuda(), pw, n, H, W, frames_per_scale=fps, rowvec=rv, res1=r1)
    wd = _deq_conv_weight(pw, cout, cin, 9).view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    xa = (a.float() * sc.repeat_interleave(fps)[:, None, None]).view(n, H, W, cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xa, wd, b, padding=1).permute(0, 2, 3, 1).reshape(n, H * W, cout)
    ref = ref + rv.cpu()[:, None, :] + r1.float().cpu()
    e = (out[..., :cout].float().cpu() - ref).abs()
    assert out.shape[:2] == (n, H * W)
    assert (e <= 2e-2 * ref.pow(2).mean().sqrt() + 1.6e-2 * ref.abs()).all(), f"max err {e.max().item()} rms {ref.pow(2).mean().sqrt().item()}"


@pytest.mark.parametrize("B,T,S,cin,cout", [(2, 5, 40, 320, 320), (1, 25, 16, 64, 64), (2, 7, 33, 640, 1280), (1, 3, 300, 128, 320)])
def test_conv_t3_fp8_exact_against_dequantised_operands(B, T, S, cin, cout):
    """3x1x1 temporal convolution on e4m3 activations with one scale per clip, zero padding at the clip ends."""
    ops = _ops()
    a = _rand_fp8((B * T, S, cin), B + T + S)
    sc = (torch.rand(B, generator=torch.Generator().manual_seed(6)) * 0.05 + 0.01)
    w = rnd(cout, cin, 3, 1, 1, scale=(3 * cin) ** -0.5, seed=1)
    b = rnd(cout, seed=2)
    pw = ops.pack_conv_t3_fp8(w, b)
    x0 = rnd(B * T, S, cout, seed=3).to(BF16).cuda()
    out = ops.conv_t3_fp8(a.view(torch.uint8).cuda(), sc.cuda(), pw, T, S, alpha=0.7, res2=x0, beta=1.0)
    wd = _deq_conv_weight(pw, cout, cin, 3).permute(0, 2, 1)                       # (Cout, Cin, 3)
    xa = (a.float() * sc.repeat_interleave(T)[:, None, None]).view(B, T, S, cin).permute(0, 2, 3, 1).reshape(B * S, cin, T)
    ref = torch.nn.functional.conv1d(xa, wd, b, padding=1).view(B, S, cout, T).permute(0, 3, 1, 2).reshape(B * T, S, cout)
    ref = 0.7 * ref + x0.float().cpu()
    e = (out[..., :cout].float().cpu() - ref).abs()
    assert (e <= 2e-2 * ref.pow(2).mean().sqrt() + 1.6e-2 * ref.abs()).all(), f"max err {e.max().item()}"


def test_video_resblock_fp8_convolutions():
    """VideoResBlock (2-D ResBlock -> temporal ResBlock -> blend) with all four convolutions in fp8 against its own bf16 path: four
    fp8 GEMMs on residual branches (~3.6e-2 each on the branch) -> rel-L2 of the block output <= 5e-2 (measured 2.7e-2); switching back
    is bit-exact."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_blocks_gpu import _seed, _tok, _rand, _emb, T
    from vista_amd.modules import attention
    from vista_amd.modules.diffusionmodules.video_model import VideoResBlock
    blk = VideoResBlock(channels=320, emb_channels=1280, dropout=0.0, out_channels=640, video_kernel_size=[3, 1, 1],
                        merge_strategy="learned_with_images", merge_factor=0.5, dims=2)
    _seed(blk, "blk", 3)
    blk = blk.cuda().eval()
    H, W = 18, 32
    x = _tok(_rand((T, 320, H, W), 10))
    emb_silu = torch.nn.functional.silu(_emb(5)).to(BF16).cuda()
    with torch.no_grad():
        base = blk(x, emb_silu, T, H, W).float().cpu()
        attention.FP8["conv"] = True
        try:
            out = blk(x, emb_silu, T, H, W).float().cpu()
        finally:
            attention.FP8["conv"] = False
        again = blk(x, emb_silu, T, H, W).float().cpu()
    r = rel_l2(out, base)
    print(f"[parity] VideoResBlock 320->640 fp8 convolutions vs bf16: rel-L2 {r:.3e}")
    assert torch.isfinite(out).all() and r <= 5e-2
    assert torch.equal(again, base)


def test_fp8_rejects_bad_arguments():
    ops = _ops()
    pw = ops.pack_linear_fp8(rnd(64, 64), None)
    xq, xs = ops.quantize_rows_fp8(rnd(8, 64).to(BF16).cuda())
    with pytest.raises(ValueError):
        ops.linear_fp8(xq[:, :32], xs, pw)
    with pytest.raises(TypeError):
        ops.linear_fp8(xq.float(), xs, pw)
    with pytest.raises(ValueError):
        ops.pack_linear_fp8(rnd(64, 72), None)  # K % 16


def test_unet_with_fp8_feedforward_vs_reference_golden():
    """BASELINE config 5: the FeedForward GEMMs of every transformer block in fp8, then additionally every ResBlock convolution.
    Re-stated tolerances against the fp32 reference golden: FeedForward only: relative L2 <= 7e-2 (measured 5.4e-2: 96 fp8 GEMMs with
    3-bit mantissas at ~3.6e-2 each on their residual branches; bf16 path: <= 2.5e-2, measured 1.4e-2); FeedForward + convolutions
    (another ~90 fp8 GEMMs): <= 1e-1 (measured 7.4e-2)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_model_gpu import GOLD, build_unet, rel_l2 as rl
    from oracle.make_golden import unet_inputs
    from vista_amd.modules import attention
    g = torch.load(os.path.join(GOLD, "unet_tiny_t5.pt"))
    net, _ = build_unet(64)
    x8, ts, ctx, y, mask = (t.cuda() for t in unet_inputs(g["T"], g["H"], g["W"], seed=g["seed_x"], sigma=g["sigma"]))
    base = net(x8, timesteps=ts, context=ctx, y=y, cond_mask=mask, num_frames=g["T"]).float().cpu()
    attention.FP8["feedforward"] = True
    try:
        out = net(x8, timesteps=ts, context=ctx, y=y, cond_mask=mask, num_frames=g["T"]).float().cpu()
    finally:
        attention.FP8["feedforward"] = False
    e8, eb = rl(out, g["out"]), rl(base, g["out"])
    print(f"[parity] UNet tiny, fp8 FeedForward: rel-L2 {e8:.3e} vs reference (bf16 path {eb:.3e}; fp8 vs bf16 {rl(out, base):.3e})")
    assert e8 <= 6.6e-2 and eb <= 2.5e-2   # measured 5.52e-2 (round 5, gpurun call 2) + 20 %
    attention.FP8["feedforward"] = attention.FP8["conv"] = True
    try:
        out_c = net(x8, timesteps=ts, context=ctx, y=y, cond_mask=mask, num_frames=g["T"]).float().cpu()
    finally:
        attention.FP8["feedforward"] = attention.FP8["conv"] = False
    e8c = rl(out_c, g["out"])
    print(f"[parity] UNet tiny, fp8 FeedForward + convolutions: rel-L2 {e8c:.3e} vs reference (vs bf16 {rl(out_c, base):.3e})")
    assert torch.isfinite(out_c).all() and e8c <= 8.8e-2   # measured 7.33e-2 + 20 %
    again = net(x8, timesteps=ts, context=ctx, y=y, cond_mask=mask, num_frames=g["T"]).float().cpu()
    assert torch.equal(again, base), "switching fp8 off must restore the bf16 path bit for bit"


ALL_FP8 = ("feedforward", "conv", "attention", "proj")  # every switch BASELINE config 5 names; a build that lacks one simply skips it


def _all_fp8_on():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_blocks_gpu import _fp8
    return _fp8(ALL_FP8)


def test_unet_full_width_fp8_vs_reference_golden():
    """BASELINE config 5 on the SHIPPED 1.65 B-parameter configuration (full widths 320/640/1280, T=5, latent 16x32)
    against the reference's own fp32 output (tests/golden/unet_full_t5.pt). Re-stated tolerance for fp8 e4m3 GEMM operands (round 5, tightened to
    what is measured + margin): rel-L2 <= 6e-2, max|err| <= 1.0e-1 max|ref| (the bf16 path: <= 2.5e-2 / 8e-2, measured 1.2e-2). Measured with FeedForwards + ResBlock
    convolutions in fp8: 6.1e-2 / 9.1e-2 -- the longer K-sums of the full widths average the e4m3 rounding only a little below the
    64-channel network's 7.4e-2, because the error is dominated by the ~190 fp8 GEMMs on residual branches, not by their width."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_model_gpu import GOLD, build_unet, rel_l2 as rl
    from oracle.make_golden import unet_inputs
    g = torch.load(os.path.join(GOLD, "unet_full_t5.pt"))
    net, _ = build_unet(320)
    x8, ts, ctx, y, mask = (t.cuda() for t in unet_inputs(g["T"], g["H"], g["W"], seed=g["seed_x"], sigma=g["sigma"]))
    run = lambda: net(x8, timesteps=ts, context=ctx, y=y, cond_mask=mask, num_frames=g["T"]).float().cpu()  # noqa: E731
    base = run()
    with _all_fp8_on():
        out = run()
    e8, eb = rl(out, g["out"]), rl(base, g["out"])
    mx = ((out - g["out"]).abs().max() / g["out"].abs().max()).item()
    print(f"[parity] full-width UNet (1.65 B), config 5 fp8: rel-L2 {e8:.3e} max-rel {mx:.3e} vs reference (bf16 path {eb:.3e}; fp8 vs bf16 {rl(out, base):.3e})")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        json.dump({"what": "unet_full_t5 golden, config 5 fp8", "rel_l2_fp8": e8, "max_rel_fp8": mx, "rel_l2_bf16": eb},
                  open(os.path.join(d, "parity_fp8_full_width.json"), "w"))
    assert torch.isfinite(out).all() and not torch.equal(out, base)
    assert e8 <= 6e-2 and mx <= 1.0e-1 and eb <= 2.5e-2   # measured 4.60e-2 / 6.8e-2 (rounds 4 and 5): the re-stated config-5 tolerance is 6e-2
    assert torch.equal(run(), base), "switching fp8 off must restore the bf16 path bit for bit"
    del net
    torch.cuda.empty_cache()


def test_sampler_full_50_step_schedule_fp8_vs_oracle():
    """the whole 50-step EDM schedule with BASELINE config 5 switched on, next to the bf16 figure (8.2e-3), against the
    same CPU oracle run. Each Euler step contracts towards the denoised estimate, so the per-forward fp8 error (7e-2 on this 64-channel
    network) does not accumulate linearly. Re-stated tolerance: rel-L2 <= 6e-2 end to end (measured 4.6e-2)."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_model_gpu import oracle_50_step, rel_l2 as rl, run_50_step, tiny_unet
    net, _ = tiny_unet()
    w, want, T, steps = oracle_50_step()
    with _all_fp8_on():
        got = run_50_step(net, w, T, steps)
    r = rl(got, want)
    print(f"[parity] 50-step CFG EulerEDM, config 5 fp8 (tiny net, T=5, 16x32) vs oracle: rel-L2 {r:.4e}")
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        json.dump({"steps": steps, "rel_l2_fp8": r}, open(os.path.join(d, "parity_50step_fp8.json"), "w"))
    assert torch.isfinite(got).all() and r <= 6e-2 and torch.equal(got[0], w["cond_frame"][0])   # measured 4.57e-2
