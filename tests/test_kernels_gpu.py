"""Per-kernel numerics of libvista_hip.so against plain PyTorch fp32 references of the same op (run on the GPU box).

Inputs are bf16-representable, the reference is computed in fp32 from those same values, so the only error left is
the kernel's bf16 output rounding + fp32 accumulation order.  Tolerance (stated): |out - ref| <= 2e-2*rms(ref) + 1.6e-2*|ref|
(bf16 has 8 bits of mantissa: 2^-8 = 3.9e-3 relative per rounding; attention adds the bf16 rounding of P).
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def _ops():
    from vista_amd import ops
    return ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(BF16).cuda()


def close(out, ref, name, rtol=1.6e-2, arel=2e-2):
    out = out.float()
    ref = ref.float()
    assert out.shape == ref.shape, f"{name}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{name}: non-finite output"
    rms = ref.pow(2).mean().sqrt().item()
    err = (out - ref).abs()
    tol = arel * rms + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} out of tolerance; max err {err.max().item():.4g} (rms ref {rms:.4g}); "
            f"first bad idx {idx}: out {out[tuple(idx)].item():.5g} ref {ref[tuple(idx)].item():.5g}")
    return err.max().item() / (rms + 1e-12)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (50, 1280, 768), (128, 128, 64), (1000, 4, 576), (257, 960, 2432), (4608, 640, 1280),
                                   (66597, 320, 64), (33300, 640, 128)])  # more tiles than CUs, ragged M
def test_linear_plain(M, N, K):
    ops = _ops()
    x = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    b = rnd(N, seed=2).float()
    pw = ops.pack_linear(w, b)
    out = ops.linear(x, pw)
    ref = x.float() @ w.float().t() + b
    close(out, ref, f"linear {M}x{N}x{K}")
    out32 = ops.linear(x, pw, out_f32=True)
    assert out32.dtype == F32
    close(out32, ref, f"linear f32 {M}x{N}x{K}", rtol=2e-3, arel=2e-3)


def test_linear_asymmetric_identity():
    """A = I against an asymmetric W catches a transposed accumulator layout (symmetric data would not)."""
    ops = _ops()
    K = 128
    x = torch.eye(K, dtype=BF16, device="cuda")
    w = (torch.arange(256 * K, dtype=F32).reshape(256, K) % 251 - 125).div(64).to(BF16).cuda()
    out = ops.linear(x, ops.pack_linear(w, None), out_f32=True)
    assert torch.equal(out, w.float().t().contiguous()), "identity GEMM must reproduce W^T exactly"


def test_linear_epilogue_full():
    ops = _ops()
    M, N, K, rpv = 600, 320, 640, 100
    x = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    b = rnd(N, seed=2).float()
    rv = rnd(M // rpv, N, seed=3).float()
    r1 = rnd(M, N, seed=4)
    r2 = rnd(M, N, seed=5)
    out = ops.linear(x, ops.pack_linear(w, b), rowvec=rv, rows_per_vec=rpv, res1=r1, res2=r2, alpha=0.3, beta=0.7)
    ref = 0.3 * (x.float() @ w.float().t() + b + rv.repeat_interleave(rpv, 0) + r1.float()) + 0.7 * r2.float()
    close(out, ref, "linear full epilogue")


def test_linear_strided_views():
    ops = _ops()
    M, N, K = 260, 320, 320
    big = rnd(M, 3 * K)
    x = big[:, K:2 * K]
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    outbuf = torch.zeros(M, 2 * N, dtype=BF16, device="cuda")
    ops.linear(x, ops.pack_linear(w, None), out=outbuf[:, N:])
    close(outbuf[:, N:], x.float() @ w.float().t(), "linear strided")
    assert outbuf[:, :N].abs().max().item() == 0


@pytest.mark.parametrize("M,C", [(200, 64), (460, 320), (130, 1280)])
def test_geglu(M, C):
    ops = _ops()
    x = rnd(M, C)
    w = rnd(8 * C, C, scale=C ** -0.5, seed=1)
    b = rnd(8 * C, seed=2).float()
    out = ops.linear(x, ops.pack_geglu(w, b))
    h = x.float() @ w.float().t() + b
    a, g = h.chunk(2, dim=-1)
    close(out, a * F.gelu(g), f"geglu {M}x{C}")


@pytest.mark.parametrize("n_img,S,C", [(3, 144, 320), (2, 576, 640), (5, 16, 64)])
def test_linear_vt(n_img, S, C):
    ops = _ops()
    x = rnd(n_img * S, C)
    w = rnd(C, C, scale=C ** -0.5, seed=1)
    vt = ops.linear_vt(x, ops.pack_linear(w, None), S)
    ref = (x.float() @ w.float().t()).view(n_img, S, C).transpose(1, 2)
    close(vt, ref, "linear_vt")


def _tok2nchw(x, n, H, W):
    return x.float().view(n, H, W, -1).permute(0, 3, 1, 2).contiguous()


def _nchw2tok(x):
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n, h * w, c)


@pytest.mark.parametrize("n,H,W,Cin,Cout,stride,ups", [
    (2, 9, 16, 64, 320, 1, 1), (3, 18, 32, 320, 320, 1, 1), (2, 18, 32, 128, 64, 2, 1), (2, 9, 16, 192, 128, 1, 2),
    (1, 72, 128, 64, 4, 1, 1), (2, 7, 5, 64, 64, 1, 1), (2, 8, 6, 64, 64, 2, 1),
    (8, 96, 88, 64, 320, 1, 1)])  # 264 row tiles: more than one round of workgroups on 256 CUs
def test_conv3x3(n, H, W, Cin, Cout, stride, ups):
    ops = _ops()
    x = rnd(n, H * W, Cin)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1)
    b = rnd(Cout, seed=2).float()
    out, Ho, Wo = ops.conv3x3(x, ops.pack_conv3x3(w, b), n, H, W, stride=stride, ups=ups)
    xi = _tok2nchw(x, n, H, W)
    if ups == 2:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    ref = F.conv2d(xi, w.float(), b, stride=stride, padding=1)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    close(out, _nchw2tok(ref), f"conv3x3 s{stride} u{ups}")


def test_conv3x3_epilogue_rowvec_res():
    ops = _ops()
    n, H, W, Cin, Cout = 4, 9, 16, 128, 192
    x = rnd(n, H * W, Cin)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1)
    b = rnd(Cout, seed=2).float()
    rv = rnd(n, Cout, seed=3).float()
    r1 = rnd(n, H * W, Cout, seed=4)
    out, _, _ = ops.conv3x3(x, ops.pack_conv3x3(w, b), n, H, W, rowvec=rv, res1=r1)
    ref = _nchw2tok(F.conv2d(_tok2nchw(x, n, H, W), w.float(), b, padding=1)) + rv[:, None, :] + r1.float()
    close(out, ref, "conv3x3 epilogue")


def test_conv_in_pad8():
    """UNet input conv: 8 real channels zero-padded to 64 (video_model.py:189)."""
    ops = _ops()
    n, H, W = 2, 9, 16
    x8 = rnd(n, 8, H, W).float()
    w = rnd(320, 8, 3, 3, scale=72 ** -0.5, seed=1)
    b = rnd(320, seed=2).float()
    xt = ops.nchw_to_tokens(x8, 64)
    out, _, _ = ops.conv3x3(xt, ops.pack_conv3x3(w, b, cin_pad=64), n, H, W)
    close(out, _nchw2tok(F.conv2d(x8, w.float(), b, padding=1)), "in conv")


@pytest.mark.parametrize("B,T,S,C", [(2, 25, 24, 64), (1, 25, 144, 320), (2, 5, 16, 128)])
def test_conv_t3(B, T, S, C):
    ops = _ops()
    x = rnd(B * T, S, C)
    w = rnd(C, C, 3, 1, 1, scale=(3 * C) ** -0.5, seed=1)
    b = rnd(C, seed=2).float()
    rv = rnd(B * T, C, seed=3).float()
    res = rnd(B * T, S, C, seed=4)
    out = ops.conv_t3(x, ops.pack_conv_t3(w, b), T, S, rowvec=rv, res2=res, alpha=0.4, beta=1.0)
    x5 = x.float().view(B, T, S, 1, C).permute(0, 4, 1, 2, 3)  # b c t s 1
    ref5 = F.conv3d(x5, w.float(), b, padding=(1, 0, 0))
    ref = ref5.permute(0, 2, 3, 4, 1).reshape(B * T, S, C)
    ref = 0.4 * (ref + rv[:, None, :]) + res.float()
    close(out, ref, "conv_t3")


# ------------------------------------------------------------------------------------------------ attention
def _sdpa(q, k, v):
    s = (q.float() @ k.float().transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return torch.softmax(s, -1) @ v.float()


@pytest.mark.parametrize("n_img,heads,S", [(2, 5, 144), (1, 2, 576), (3, 1, 64), (1, 5, 2304), (2, 3, 200), (1, 1, 9216),
                                           (1, 2, 2048), (2, 1, 2120), (1, 1, 2056), (1, 3, 4104)])  # >= 2048: ragged last q-block / key tile, odd and even tile counts
def test_attn_spatial(n_img, heads, S):
    ops = _ops()
    Cc = heads * 64
    x = rnd(n_img * S, Cc, scale=1.0)
    wq, wk, wv = (rnd(Cc, Cc, scale=1.5 * Cc ** -0.5, seed=s) for s in (1, 2, 3))
    qk = ops.linear(x, ops.pack_linear_cat([wq, wk]))
    vt = ops.linear_vt(x, ops.pack_linear(wv, None), S)
    o = ops.attn_spatial(qk[:, :Cc], qk[:, Cc:], vt, n_img, heads, S)
    q = qk[:, :Cc].float().view(n_img, S, heads, 64).permute(0, 2, 1, 3)
    k = qk[:, Cc:].float().view(n_img, S, heads, 64).permute(0, 2, 1, 3)
    v = vt.float().view(n_img, heads, 64, S).transpose(-1, -2)
    ref = _sdpa(q, k, v).permute(0, 2, 1, 3).reshape(n_img * S, Cc)
    close(o, ref, f"attn_spatial S={S}", rtol=2e-2, arel=3e-2)


@pytest.mark.parametrize("n_img,heads,S", [(2, 5, 144), (1, 2, 576), (3, 1, 64), (1, 5, 2304), (2, 3, 200), (1, 1, 9216),
                                           (1, 2, 2048), (2, 1, 2120), (1, 1, 2056), (1, 3, 4104)])
def test_attn_spatial_qkv_rows(n_img, heads, S):
    """V as the third column block of ONE fused q|k|v GEMM (round 3: no V^T tensor; the kernel transposes the V tile out of LDS with
    ds_read_b64_tr_b16) against fp32 SDPA, and BITWISE against the V^T form fed the transposed copy of the very same V values
    (same fragments, same MFMA order: only the way the tile reaches the registers differs)."""
    ops = _ops()
    Cc = heads * 64
    x = rnd(n_img * S, Cc, scale=1.0)
    wq, wk, wv = (rnd(Cc, Cc, scale=1.5 * Cc ** -0.5, seed=s) for s in (1, 2, 3))
    qkv = ops.linear(x, ops.pack_linear_cat([wq, wk, wv]))
    assert qkv.shape == (n_img * S, 3 * Cc)
    o = ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], n_img, heads, S, v_rows=True)
    q, k, v = (qkv[:, i * Cc:(i + 1) * Cc].float().view(n_img, S, heads, 64).permute(0, 2, 1, 3) for i in range(3))
    ref = _sdpa(q, k, v).permute(0, 2, 1, 3).reshape(n_img * S, Cc)
    close(o, ref, f"attn_spatial q|k|v rows S={S}", rtol=2e-2, arel=3e-2)
    vt = qkv[:, 2 * Cc:].reshape(n_img, S, Cc).transpose(1, 2).contiguous()   # (n_img, heads*64, S): the V^T image of the same values
    o_vt = ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], vt, n_img, heads, S)
    assert torch.equal(o, o_vt), "the transposing LDS read must deliver exactly the V^T fragments"
    with pytest.raises(ValueError):
        ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], vt, n_img, heads, S, v_rows=True)   # a V^T tensor passed as rows


@pytest.mark.parametrize("n_img,heads,S", [(2, 5, 144), (1, 2, 576), (1, 5, 2304), (2, 3, 200), (1, 1, 9216), (2, 1, 2120), (1, 3, 4104)])
def test_attn_spatial_qkv_log2_prescaled_query(n_img, heads, S):
    """The pre-scaled form (vk_attn_spatial_qkv_log2_bf16): the query weights carry dim_head^-0.5 * log2(e), a score is the base-2 exponent
    and rows with a maximum within +-60 octaves run against the base ZERO (no scale / base arithmetic per score). Against fp32 SDPA of the
    scaled projection the kernel actually saw, and against the plain kernel on the unscaled one."""
    ops = _ops()
    Cc = heads * 64
    x = rnd(n_img * S, Cc, scale=1.0)
    wq, wk, wv = (rnd(Cc, Cc, scale=1.5 * Cc ** -0.5, seed=s) for s in (1, 2, 3))
    c = 64 ** -0.5 * ops.LOG2E
    qkv = ops.linear(x, ops.pack_linear_cat([wq, wk, wv]))
    qkv2 = ops.linear(x, ops.pack_linear_cat([wq.float() * c, wk, wv]))
    assert torch.equal(qkv[:, Cc:], qkv2[:, Cc:])
    o = ops.attn_spatial(qkv2[:, :Cc], qkv2[:, Cc:2 * Cc], qkv2[:, 2 * Cc:], n_img, heads, S, v_rows=True, q_log2=True)
    q, k, v = (qkv2[:, i * Cc:(i + 1) * Cc].float().view(n_img, S, heads, 64).permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / ops.LOG2E, -1) @ v).permute(0, 2, 1, 3).reshape(n_img * S, Cc)   # exp2(q.k) = exp(q.k ln 2)
    close(o, ref, f"attn_spatial log2-prescaled q S={S}", rtol=2e-2, arel=3e-2)
    # against the plain kernel on the UNSCALED projection the two differ by which bf16 rounding of q they saw (one each, of the same fp32
    # value): relative L2 ~5e-3 on these unit-variance logits, i.e. the projection's own rounding noise, not the kernel's
    plain = ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], n_img, heads, S, v_rows=True).float()
    assert ((o.float() - plain).pow(2).sum() / plain.pow(2).sum()).sqrt().item() < 1.2e-2
    assert torch.equal(o, ops.attn_spatial(qkv2[:, :Cc], qkv2[:, Cc:2 * Cc], qkv2[:, 2 * Cc:], n_img, heads, S, v_rows=True, q_log2=True))


@pytest.mark.parametrize("S", [512, 4608])
@pytest.mark.parametrize("gain", [3.0, 12.0, 60.0, 400.0])
def test_attn_spatial_qkv_log2_leaves_the_zero_base_when_it_must(S, gain):
    """Zero-base edge cases of the pre-scaled kernel: exponents far above +60 (gain 12: ~140 octaves -> exp2 overflows to +inf against base 0,
    the row-sum test fails and the row re-bases on its true maximum), a row whose FIRST tile lies far below -60 (true first base), spikes
    that arrive in late tiles, an all-equal row, and rows that stay inside the band (gain 3 on most rows)."""
    ops = _ops()
    c = 64 ** -0.5 * ops.LOG2E
    q = rnd(S, 64)
    k = rnd(S, 64, seed=1)
    v = rnd(S, 64, seed=2)
    for j, row in enumerate(range(70, S, 197)):
        k[row] = q[(37 * j + 5) % S] * gain
    k[:64] = -q[300] * gain
    k[S - 1] = q[300] * gain
    q[11] = 0
    qs = (q.float() * c).to(BF16)
    buf = torch.cat([qs, k, v], 1).contiguous()
    o = ops.attn_spatial(buf[:, :64], buf[:, 64:128], buf[:, 128:], 1, 1, S, v_rows=True, q_log2=True)
    assert torch.isfinite(o.float()).all()
    ref = torch.softmax((qs.float() @ k.float().t()) / ops.LOG2E, -1) @ v.float()   # exp2(qs.k) = exp(qs.k * ln 2)
    close(o, ref, f"attn log2 zero-base S={S} gain={gain}", rtol=2e-2, arel=3e-2)


def test_attn_spatial_spike_forces_rescale():
    """One key row strongly aligned with one query row so the running max jumps mid-sequence (online-softmax rescale path)."""
    ops = _ops()
    S, heads = 512, 1
    q = rnd(S, 64)
    k = rnd(S, 64, seed=1)
    v = rnd(S, 64, seed=2)
    k[300] = q[7] * 4
    k[450] = q[100] * 6
    vt = v.t().contiguous().view(1, 64, S)
    o = ops.attn_spatial(q, k, vt, 1, heads, S)
    close(o, _sdpa(q[None], k[None], v[None])[0], "attn spike", rtol=2e-2, arel=3e-2)


@pytest.mark.parametrize("S", [512, 2304])
@pytest.mark.parametrize("gain", [3.0, 12.0, 60.0])
def test_attn_spatial_max_free_fast_path_falls_back(S, gain):
    """The kernel takes the exponentials of a tile against the EXISTING softmax base without computing the tile's maximum and
    validates the tile afterwards through its row sums (attention.hip, "max-free fast path"). Scores that jump far above the base
    in a LATE tile must send that tile through the slow path: gain 3 stays within the threshold band, 12 exceeds it, 60 overflows
    fp32 exp2 against the old base (|q|^2 ~ 64 -> 60*64/8*log2(e) ~ 690 octaves -> +inf row sums). Also a row block whose FIRST
    tile is far BELOW everything that follows (the first base is useless), and all-equal scores."""
    ops = _ops()
    q = rnd(S, 64)
    k = rnd(S, 64, seed=1)
    v = rnd(S, 64, seed=2)
    for j, row in enumerate(range(70, S, 197)):      # spikes spread over many tiles, each aimed at a different query row
        k[row] = q[(37 * j + 5) % S] * gain
    k[:64] = -q[300] * gain                           # query row 300: its first tile sits far below its later scores
    k[S - 1] = q[300] * gain
    q[11] = 0                                         # all scores equal: uniform softmax
    vt = v.t().contiguous().view(1, 64, S)
    o = ops.attn_spatial(q, k, vt, 1, 1, S)
    assert torch.isfinite(o.float()).all()
    close(o, _sdpa(q[None], k[None], v[None])[0], f"attn max-free S={S} gain={gain}", rtol=2e-2, arel=3e-2)


@pytest.mark.parametrize("B,T,S,heads", [(2, 25, 40, 5), (1, 25, 144, 2), (2, 7, 9, 1), (1, 32, 16, 3)])
def test_attn_temporal(B, T, S, heads):
    ops = _ops()
    Cc = heads * 64
    qkv = rnd(B * T * S, 3 * Cc)
    o = ops.attn_temporal(qkv, B, T, S, heads)
    t5 = qkv.float().view(B, T, S, 3, heads, 64)
    q, k, v = (t5[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))  # b s h t d
    ref = _sdpa(q, k, v).permute(0, 3, 1, 2, 4).reshape(B * T * S, Cc)
    close(o, ref, "attn_temporal", rtol=2e-2, arel=3e-2)


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("n,S,C,fpg,silu,eps", [(4, 144, 320, 1, True, 1e-5), (6, 100, 64, 1, False, 1e-6), (6, 64, 192, 3, True, 1e-5),
                                                 (2, 576, 960, 1, True, 1e-5), (50, 16, 2560, 25, True, 1e-5), (2, 300, 1920, 2, False, 1e-5),
                                                 # > 256 partials per group: the two-level finalize (VAE decoder's 576x1024 levels)
                                                 (2, 40000, 64, 1, True, 1e-6), (10, 9216, 64, 5, True, 1e-5), (3, 33000, 128, 3, False, 1e-5)])
def test_groupnorm(n, S, C, fpg, silu, eps):
    ops = _ops()
    x = (rnd(n, S, C).float() * 1.5 + 0.7).to(BF16)
    gamma = rnd(C, seed=1).float() + 1.0
    beta = rnd(C, seed=2).float()
    y = ops.groupnorm(x, gamma, beta, eps, silu, frames_per_group=fpg)
    xg = x.float().view(n // fpg, fpg * S, C).permute(0, 2, 1)  # (groups of images, C, fpg*S)
    ref = F.group_norm(xg, 32, gamma, beta, eps)
    if silu:
        ref = F.silu(ref)
    close(y, ref.permute(0, 2, 1).reshape(n, S, C), f"groupnorm C={C} fpg={fpg}")


@pytest.mark.parametrize("rows,C", [(1000, 320), (300, 64), (257, 640), (129, 1280)])
def test_layernorm(rows, C):
    ops = _ops()
    x = (rnd(rows, C).float() * 2 + 0.3).to(BF16)
    gamma = rnd(C, seed=1).float() + 1.0
    beta = rnd(C, seed=2).float()
    y = ops.layernorm(x, gamma, beta)
    close(y, F.layer_norm(x.float(), (C,), gamma, beta, 1e-5), f"layernorm C={C}")
    rpv = 50
    av = rnd((rows + rpv - 1) // rpv, C, seed=3).float()
    y2, s2 = ops.layernorm(x, gamma, beta, addvec=av, rows_per_vec=rpv, want_sum=True)
    u = (x.float() + av.repeat_interleave(rpv, 0)[:rows]).to(BF16)
    assert torch.equal(s2, u), "layernorm sum_out must be the bf16-rounded x + addvec"
    close(y2, F.layer_norm(u.float(), (C,), gamma, beta, 1e-5), f"layernorm+add C={C}")


# ------------------------------------------------------------------------------------------------ elementwise
def test_concat_and_layout():
    ops = _ops()
    a, b = rnd(7, 33, 64), rnd(7, 33, 128, seed=1)
    assert torch.equal(ops.concat_channels(a, b), torch.cat([a, b], -1))
    x = rnd(3, 8, 9, 16).float()
    t = ops.nchw_to_tokens(x, 64)
    assert torch.equal(t[..., :8], x.permute(0, 2, 3, 1).reshape(3, 144, 8).to(BF16))
    assert t[..., 8:].abs().max().item() == 0
    y = torch.randn(3, 144, 4, device="cuda")
    assert torch.equal(ops.tokens_to_nchw(y, 3, 4, 9, 16), y.view(3, 9, 16, 4).permute(0, 3, 1, 2).contiguous())


def test_timestep_embedding_and_emb_combine():
    ops = _ops()
    t = torch.tensor([0.25 * math.log(700.0), 0.0, -1.553652, 3.0, 24.0], device="cuda")
    dim = 320
    e = ops.timestep_embedding(t, dim)
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=F32, device="cuda") / half)
    args = t[:, None] * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    assert (e.float() - ref).abs().max().item() < 1e-2
    # KAT from the reference: timestep_embedding(0.25*ln 700, 320)[:3] and [160:163]
    kat = torch.tensor([-0.06692367, 0.02463922, 0.11090361, 0.99775809, 0.99969643, 0.99383116])
    got = torch.cat([e[0, :3], e[0, 160:163]]).float().cpu()
    assert (got - kat).abs().max().item() < 6e-3
    a, b, c = (torch.randn(5, 64, device="cuda") for _ in range(3))
    m = torch.tensor([1.0, 0, 0, 1, 0], device="cuda")
    emb, se = ops.emb_combine(a, b, c, m)
    ref = a * m[:, None] + b * (1 - m[:, None]) + c
    assert torch.allclose(emb, ref, atol=1e-6)
    close(se, F.silu(ref), "silu(emb)")
    emb2, _ = ops.emb_combine(None, b, c, None)
    assert torch.allclose(emb2, b + c, atol=1e-6)


def test_sampler_elementwise():
    ops = _ops()
    T, H, W = 5, 9, 16
    x = torch.randn(T, 4, H, W, device="cuda") * 10
    cf = torch.randn(T, 4, H, W, device="cuda")
    mask = torch.tensor([1.0, 0, 0, 1, 0], device="cuda")
    cc = torch.randn(T, 4, H, W, device="cuda")
    c_in = 0.37
    x0 = x.clone()
    net_in = ops.sampler_prepare(x, cf, mask, None, cc, 64, c_in, True)
    xr = x0 * (1 - mask.view(-1, 1, 1, 1)) + cf * mask.view(-1, 1, 1, 1)
    assert torch.allclose(x, xr, atol=1e-6)
    tok = lambda z: z.permute(0, 2, 3, 1).reshape(T, H * W, 4)
    assert torch.equal(net_in[:T, :, :4], tok(xr * c_in).to(BF16))
    assert torch.equal(net_in[T:, :, :4], tok(xr * c_in).to(BF16))
    assert net_in[:T, :, 4:].abs().max().item() == 0
    assert torch.equal(net_in[T:, :, 4:8], tok(cc).to(BF16))
    assert net_in[T:, :, 8:].abs().max().item() == 0
    net = torch.randn(2 * T, H * W, 4, device="cuda")
    scale = torch.linspace(1, 2.5, T, device="cuda")
    c_out, c_skip, sig, sign = -0.9, 0.2, 3.0, 2.0
    xb = x.clone()
    ops.sampler_update(x, net, scale, c_out, c_skip, sig, sign)
    untok = lambda z: z.view(T, H, W, 4).permute(0, 3, 1, 2)
    du = untok(net[:T]) * c_out + xb * c_skip
    dc = untok(net[T:]) * c_out + xb * c_skip
    g = du + scale.view(-1, 1, 1, 1) * (dc - du)
    ref = xb + (xb - g) / sig * (sign - sig)
    assert torch.allclose(x, ref, rtol=1e-5, atol=1e-5)
    # generic pieces
    n2 = torch.randn(2 * T, 4, H, W, device="cuda")
    assert torch.allclose(ops.cfg_combine(n2, scale), n2[:T] + scale.view(-1, 1, 1, 1) * (n2[T:] - n2[:T]), atol=1e-6)
    sg = torch.full((T,), 3.0, device="cuda"); sn = torch.full((T,), 2.0, device="cuda")
    assert torch.allclose(ops.euler_step(xb, g, sg, sn), ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(ops.mask_replace(x0, cf, mask), xr, atol=1e-6)
    co = torch.randn(T, device="cuda"); cs = torch.randn(T, device="cuda")
    assert torch.allclose(ops.denoiser_combine(xb, cf, co, cs), xb * co.view(-1, 1, 1, 1) + cf * cs.view(-1, 1, 1, 1), atol=1e-5)
    assert torch.allclose(ops.scale_rows(xb, co), xb * co.view(-1, 1, 1, 1), atol=1e-6)


# ------------------------------------------------------------------------------------------------ block-tile variants
@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 7])
def test_gemm_family_all_tile_configs(cfg):
    """Every block-tile variant (128x128, 256x128, 256x256, 256x320 sixteen-wave, 128x160, 256x320 eight-wave pipelined) of every loader /
    epilogue, on shapes with ragged M and N edges (a forced variant that does not take a problem falls back to the launcher's choice)."""
    ops = _ops()
    ops.TILE_CFG = cfg
    try:
        M, N, K = 1100, 960, 640
        x = rnd(M, K)
        w = rnd(N, K, scale=K ** -0.5, seed=1)
        b = rnd(N, seed=2).float()
        r1 = rnd(M, N, seed=4)
        out = ops.linear(x, ops.pack_linear(w, b), res1=r1, alpha=0.5)
        close(out, 0.5 * (x.float() @ w.float().t() + b + r1.float()), f"linear cfg{cfg}")
        C = 320
        xg = rnd(700, C)
        wg = rnd(8 * C, C, scale=C ** -0.5, seed=1)
        bg = rnd(8 * C, seed=2).float()
        h = xg.float() @ wg.float().t() + bg
        a, g = h.chunk(2, dim=-1)
        close(ops.linear(xg, ops.pack_geglu(wg, bg)), a * F.gelu(g), f"geglu cfg{cfg}")
        n_img, S = 3, 400
        xv = rnd(n_img * S, C)
        wv = rnd(C, C, scale=C ** -0.5, seed=3)
        close(ops.linear_vt(xv, ops.pack_linear(wv, None), S), (xv.float() @ wv.float().t()).view(n_img, S, C).transpose(1, 2), f"vt cfg{cfg}")
        n, H, W, Cin, Cout = 3, 18, 32, 128, 320
        xc = rnd(n, H * W, Cin)
        wc = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=1)
        bc = rnd(Cout, seed=2).float()
        oc, _, _ = ops.conv3x3(xc, ops.pack_conv3x3(wc, bc), n, H, W)
        close(oc, _nchw2tok(F.conv2d(_tok2nchw(xc, n, H, W), wc.float(), bc, padding=1)), f"conv cfg{cfg}")
        B, T, S2, Ct = 2, 5, 130, 128
        xt = rnd(B * T, S2, Ct)
        wt = rnd(Ct, Ct, 3, 1, 1, scale=(3 * Ct) ** -0.5, seed=1)
        bt = rnd(Ct, seed=2).float()
        ot = ops.conv_t3(xt, ops.pack_conv_t3(wt, bt), T, S2)
        x5 = xt.float().view(B, T, S2, 1, Ct).permute(0, 4, 1, 2, 3)
        close(ot, F.conv3d(x5, wt.float(), bt, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(B * T, S2, Ct), f"conv_t3 cfg{cfg}")
    finally:
        ops.TILE_CFG = 0


def test_conv_t3_halo_frames_equal_a_longer_clip():
    """Frame-sharded temporal conv: a 4-frame middle slice with halo frames from its neighbours == the slice of the full-clip conv."""
    ops = _ops()
    B, T, S, C = 2, 8, 40, 128
    x = rnd(B * T, S, C)
    w = rnd(C, C, 3, 1, 1, scale=(3 * C) ** -0.5, seed=1)
    b = rnd(C, seed=2).float()
    pw = ops.pack_conv_t3(w, b)
    full = ops.conv_t3(x, pw, T, S).view(B, T, S, C)
    x4 = x.view(B, T, S, C)
    for t0, t1 in ((0, 3), (3, 7), (7, 8)):
        loc = x4[:, t0:t1].reshape(B * (t1 - t0), S, C).contiguous()
        prev = x4[:, t0 - 1].contiguous() if t0 > 0 else None
        nxt = x4[:, t1].contiguous() if t1 < T else None
        out = ops.conv_t3(loc, pw, t1 - t0, S, halo_prev=prev, halo_next=nxt).view(B, t1 - t0, S, C)
        assert torch.equal(out, full[:, t0:t1]), (t0, t1)


# ------------------------------------------------------------------------------------------------ split-K
def _with_and_without_splitk(fn):
    """Runs fn() with the split-K workspace enabled and disabled; returns (split, plain)."""
    ops = _ops()
    a = fn()
    saved, ops.SPLITK_WS_BYTES = ops.SPLITK_WS_BYTES, 0
    try:
        b = fn()
    finally:
        ops.SPLITK_WS_BYTES = saved
    return a, b


def test_splitk_dense_matches_reference_and_plain_kernel():
    """M = 4032 rows (a level-2 activation of one 8-GPU rank), N = 1280, K = 5120: 64 tiles -> 4 K slices."""
    ops = _ops()
    M, N, K = 4032, 1280, 5120
    x = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    b = rnd(N, seed=2).float()
    r1 = rnd(M, N, seed=3)
    r2 = rnd(M, N, seed=4)
    pw = ops.pack_linear(w, b)
    split, plain = _with_and_without_splitk(lambda: ops.linear(x, pw, res1=r1, res2=r2, alpha=0.6, beta=0.4))
    ref = 0.6 * (x.float() @ w.float().t().cuda() + b.cuda() + r1.float()) + 0.4 * r2.float()
    close(split, ref, "split-K dense")
    close(plain, ref, "plain dense")
    # a different fp32 summation order is allowed: the two results are at most one bf16 ulp apart
    assert (split.float() - plain.float()).abs().max().item() <= 2 ** -6 * ref.abs().max().item()
    s32, p32 = _with_and_without_splitk(lambda: ops.linear(x, pw, out_f32=True))
    assert torch.allclose(s32, p32, rtol=1e-4, atol=1e-4) and torch.equal(s32, ops.linear(x, pw, out_f32=True)), "split-K must be repeatable"
    assert not torch.equal(s32, p32), "the split-K path was not taken (its fp32 summation order differs from the plain kernel's)"


def test_splitk_conv3x3_and_temporal_conv():
    """Level-3 convs of the UNet (N = 50 images, 9x16 pixels, 1280 channels): 116 tiles of 256x320 -> 2 K slices."""
    ops = _ops()
    n, H, W, C = 50, 9, 16, 1280
    x = rnd(n, H * W, C)
    w = rnd(C, C, 3, 3, scale=(9 * C) ** -0.5, seed=1)
    b = rnd(C, seed=2).float()
    rv = rnd(n, C, seed=3).float()
    pw = ops.pack_conv3x3(w, b)
    split, plain = _with_and_without_splitk(lambda: ops.conv3x3(x, pw, n, H, W, rowvec=rv)[0])
    ref = _nchw2tok(F.conv2d(_tok2nchw(x, n, H, W), w.float().cuda(), b.cuda(), padding=1)) + rv.cuda()[:, None, :]
    close(split, ref, "split-K conv3x3")
    close(plain, ref, "plain conv3x3")
    wt = rnd(C, C, 3, 1, 1, scale=(3 * C) ** -0.5, seed=5)
    pwt = ops.pack_conv_t3(wt, b)
    T, S = 25, H * W
    st, pt = _with_and_without_splitk(lambda: ops.conv_t3(x, pwt, T, S, res2=x, alpha=0.3, beta=1.0))
    x5 = x.float().view(2, T, S, C).permute(0, 3, 1, 2)[..., None]
    reft = F.conv3d(x5, wt.float().cuda(), b.cuda(), padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(n, S, C)
    reft = 0.3 * reft + x.float()
    close(st, reft, "split-K temporal conv")
    close(pt, reft, "plain temporal conv")


# ------------------------------------------------------------------------------------------------ round 4: the pipelined 256x320 kernel
@pytest.mark.parametrize("kind", ["dense+res+stats", "dense_strided_A", "qkv_lnfold", "ff_out+blend", "geglu_lnfold", "conv3x3+emb+res", "conv3x3_stride2",
                                  "conv3x3_asym", "conv3x3_ups2", "conv_t3", "conv_t3+blend"])
@pytest.mark.parametrize("n,H,W,C", [(3, 20, 24, 320), (5, 9, 13, 640), (9, 36, 64, 320)])   # ragged last tile, tiles spanning 3-4 images, > 256 tiles
def test_gemm_pipe_is_bitwise_the_sixteen_wave_kernel(kind, n, H, W, C):
    """gemm_pipe.hip (VkGemmDesc.tile_cfg = 7: eight waves, 64x160 wave tiles, fragments double-buffered across the K-step barrier, the next
    K-step's LDS-DMA pieces issued between the MFMAs as buffer loads whose out-of-range offsets are the conv padding / the rows past M) against the
    sixteen-wave 256x320 kernel (tile_cfg 4), bit for bit, for every loader x epilogue it takes -- and against torch fp32. Reference call sites:
    openaimodel.py:198,232 (ResBlock convs), :136 (Downsample), model.py:77-81 (asymmetric pad), video_model.py:38-52 (time_stack),
    attention.py:85-110 (GEGLU / FeedForward), :344-346,421 (projections)."""
    ops = _ops()
    S = H * W
    M = n * S
    x = rnd(M, C)
    x3 = x.view(n, S, C)
    res = rnd(M, C, seed=3)
    rv = rnd(n, C, seed=5).float()
    ref = None
    if kind == "dense+res+stats":
        w, b = rnd(C, C, scale=C ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_linear(w, b)
        fn = lambda: ops.linear(x, pw, res1=res, rowvec=rv, rows_per_vec=S, emit_stats=True)  # noqa: E731
        ref = x.float() @ w.float().t() + b + res.float() + rv.repeat_interleave(S, 0)
    elif kind == "dense_strided_A":   # A is a column block of a wider tensor (lda = 3C): the last tile's buffer range ends inside the last row
        wide = rnd(M, 3 * C, seed=11)
        xs = wide[:, C:2 * C]
        w, b = rnd(C, C, scale=C ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_linear(w, b)
        fn = lambda: ops.linear(xs, pw, res1=res)  # noqa: E731
        ref = xs.float() @ w.float().t() + b + res.float()
    elif kind == "qkv_lnfold":
        nrm = _Norm(C, 7)
        w, b = rnd(3 * C, C, scale=C ** -0.5, seed=1), rnd(3 * C, seed=2).float()
        pw = ops.pack_linear(w, b, ln=nrm)
        st = ops.rowstats(x)
        fn = lambda: ops.linear(x, pw, ln=st)  # noqa: E731
        ref = _ln_ref(x, nrm.weight, nrm.bias) @ w.float().t() + b
    elif kind == "ff_out+blend":
        h4 = rnd(M, 4 * C, seed=9)
        w, b = rnd(C, 4 * C, scale=(4 * C) ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_linear(w, b)
        fn = lambda: ops.linear(h4, pw, res1=res, alpha=0.4, res2=x, rowvec2=rv, beta=0.6, rows_per_vec=S)  # noqa: E731
        ref = 0.4 * (h4.float() @ w.float().t() + b + res.float()) + 0.6 * (x.float() + rv.repeat_interleave(S, 0))
    elif kind == "geglu_lnfold":
        nrm = _Norm(C, 7)
        w, b = rnd(8 * C, C, scale=C ** -0.5, seed=1), rnd(8 * C, seed=2).float()
        pw = ops.pack_geglu(w, b, ln=nrm)
        st = ops.rowstats(x)
        fn = lambda: ops.linear(x, pw, ln=st)  # noqa: E731
        a, g = (_ln_ref(x, nrm.weight, nrm.bias) @ w.float().t() + b).chunk(2, dim=-1)
        ref = a * F.gelu(g)
    elif kind.startswith("conv3x3"):
        w, b = rnd(C, C, 3, 3, scale=(9 * C) ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_conv3x3(w, b)
        xn = _tok2nchw(x3, n, H, W)
        if kind == "conv3x3+emb+res":
            fn = lambda: ops.conv3x3(x3, pw, n, H, W, rowvec=rv, res1=x3)[0]  # noqa: E731
            ref = _nchw2tok(F.conv2d(xn, w.float(), b, padding=1)) + rv[:, None, :] + x3.float()
        elif kind == "conv3x3_ups2":   # Upsample.forward: nearest x2, then conv (openaimodel.py:100-102), the upsample fused into the loader
            fn = lambda: ops.conv3x3(x3, pw, n, H, W, ups=2, rowvec=rv)[0]  # noqa: E731
            ref = _nchw2tok(F.conv2d(F.interpolate(xn, scale_factor=2, mode="nearest"), w.float(), b, padding=1)) + rv[:, None, :]
        elif kind == "conv3x3_stride2":
            if H % 2 or W % 2:
                pytest.skip("stride 2 needs even H, W")
            fn = lambda: ops.conv3x3(x3, pw, n, H, W, stride=2)[0]  # noqa: E731
            ref = _nchw2tok(F.conv2d(xn, w.float(), b, stride=2, padding=1))
        else:
            if H % 2 or W % 2:
                pytest.skip("the asymmetric-pad Downsample needs even H, W")
            fn = lambda: ops.conv3x3(x3, pw, n, H, W, stride=2, asym_pad=True)[0]  # noqa: E731
            ref = _nchw2tok(F.conv2d(F.pad(xn, (0, 1, 0, 1)), w.float(), b, stride=2))
    else:
        T = n   # one clip of n frames
        w, b = rnd(C, C, 3, 1, 1, scale=(3 * C) ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_conv_t3(w, b)
        x5 = x3.float().view(1, T, S, 1, C).permute(0, 4, 1, 2, 3)
        ref = F.conv3d(x5, w.float(), b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(n, S, C)
        if kind == "conv_t3":
            fn = lambda: ops.conv_t3(x3, pw, T, S)  # noqa: E731
        else:
            fn = lambda: ops.conv_t3(x3, pw, T, S, res2=x3, alpha=0.3, beta=1.0)  # noqa: E731
            ref = 0.3 * ref + x3.float()
    outs = {}
    for cfg in (7, 4):
        ops.TILE_CFG = cfg
        try:
            outs[cfg] = fn()
        finally:
            ops.TILE_CFG = 0
    o7, o4 = outs[7], outs[4]
    if isinstance(o7, tuple):
        (o7, s7), (o4, s4) = o7, o4
        assert s7.parts == s4.parts and torch.equal(s7.t, s4.t), "row-sum slabs differ"
        _check_stats(s7, o7)
    close(o7.reshape(ref.shape), ref, f"gemm_pipe {kind}")
    assert torch.equal(o7, o4), "pipelined and sixteen-wave kernels must agree bit for bit"


@pytest.mark.parametrize("kind", ["dense+res+stats", "dense_strided_A", "qkv_lnfold", "ff_out+blend", "geglu_lnfold", "dense_K_32_mod_64"])
@pytest.mark.parametrize("n,H,W,C", [(3, 20, 24, 320), (5, 9, 13, 640), (9, 36, 64, 320)])   # ragged last tile, tiles spanning images, > 512 tiles
def test_gemm_pipe2_is_bitwise_the_pipelined_kernel(kind, n, H, W, C):
    """gemm_pipe2.hip (VkGemmDesc.tile_cfg bit 4: four waves, 128x320 tiles, TWO workgroups per CU, 32-deep K-steps, activation ring of three and
    weight ring of two stages, one counted vmcnt wait per K-step) against the eight-wave 256x320 pipelined kernel (tile_cfg 7), bit for bit incl. the
    row-sum slabs, for every DENSE epilogue it takes, and against torch fp32. An A/B option of round 6 (off by default: profiles/r06_gemm_pipe2.txt).
    Reference call sites: attention.py:85-110 (GEGLU / FeedForward), :344-346,421 (projections)."""
    ops = _ops()
    S = H * W
    M = n * S
    x = rnd(M, C)
    res = rnd(M, C, seed=3)
    rv = rnd(n, C, seed=5).float()
    if kind == "dense+res+stats":
        w, b = rnd(C, C, scale=C ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_linear(w, b)
        fn = lambda: ops.linear(x, pw, res1=res, rowvec=rv, rows_per_vec=S, emit_stats=True)  # noqa: E731
        ref = x.float() @ w.float().t() + b + res.float() + rv.repeat_interleave(S, 0)
    elif kind == "dense_strided_A":
        wide = rnd(M, 3 * C, seed=11)
        xs = wide[:, C:2 * C]
        w, b = rnd(C, C, scale=C ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_linear(w, b)
        fn = lambda: ops.linear(xs, pw, res1=res)  # noqa: E731
        ref = xs.float() @ w.float().t() + b + res.float()
    elif kind == "dense_K_32_mod_64":   # K = C + 64 is a multiple of 64 (the ABI's rule); an odd count of 64-deep steps = 2 (mod 4) 32-deep ones: both ring parities end the loop
        xk = rnd(M, C + 64, seed=13)
        w, b = rnd(C, C + 64, scale=C ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_linear(w, b)
        fn = lambda: ops.linear(xk, pw)  # noqa: E731
        ref = xk.float() @ w.float().t() + b
    elif kind == "qkv_lnfold":
        nrm = _Norm(C, 7)
        w, b = rnd(3 * C, C, scale=C ** -0.5, seed=1), rnd(3 * C, seed=2).float()
        pw = ops.pack_linear(w, b, ln=nrm)
        st = ops.rowstats(x)
        fn = lambda: ops.linear(x, pw, ln=st)  # noqa: E731
        ref = _ln_ref(x, nrm.weight, nrm.bias) @ w.float().t() + b
    elif kind == "ff_out+blend":
        h4 = rnd(M, 4 * C, seed=9)
        w, b = rnd(C, 4 * C, scale=(4 * C) ** -0.5, seed=1), rnd(C, seed=2).float()
        pw = ops.pack_linear(w, b)
        fn = lambda: ops.linear(h4, pw, res1=res, alpha=0.4, res2=x, rowvec2=rv, beta=0.6, rows_per_vec=S)  # noqa: E731
        ref = 0.4 * (h4.float() @ w.float().t() + b + res.float()) + 0.6 * (x.float() + rv.repeat_interleave(S, 0))
    else:
        nrm = _Norm(C, 7)
        w, b = rnd(8 * C, C, scale=C ** -0.5, seed=1), rnd(8 * C, seed=2).float()
        pw = ops.pack_geglu(w, b, ln=nrm)
        st = ops.rowstats(x)
        fn = lambda: ops.linear(x, pw, ln=st)  # noqa: E731
        a, g = (_ln_ref(x, nrm.weight, nrm.bias) @ w.float().t() + b).chunk(2, dim=-1)
        ref = a * F.gelu(g)
    outs = {}
    for cfg in (16, 7):
        ops.TILE_CFG = cfg
        try:
            outs[cfg] = fn()
        finally:
            ops.TILE_CFG = 0
    o2, o7 = outs[16], outs[7]
    if isinstance(o2, tuple):
        (o2, s2), (o7, s7) = o2, o7
        assert s2.parts == s7.parts and torch.equal(s2.t, s7.t), "row-sum slabs differ"
        _check_stats(s2, o2)
    close(o2.reshape(ref.shape), ref, f"gemm_pipe2 {kind}")
    assert torch.equal(o2, o7), "the two-per-CU and the eight-wave pipelined kernels must agree bit for bit"


@pytest.mark.parametrize("kind", ["qkv_lnfold", "dense_K4N+res+stats", "conv3x3+emb+res", "conv_t3+blend"])
@pytest.mark.parametrize("n,H,W", [(29, 36, 64), (30, 35, 64)])   # 261 row tiles of 256 (5 in the last round); 262.5 (ragged last tile)
def test_gemm_tail_split_and_row_ranges_are_bitwise(kind, n, H, W):
    """Round 5: a one-tile-per-workgroup launch of the pipelined 256x320 kernel whose last round would be nearly empty is split by vk_gemm_bf16
    into whole rounds on that kernel + the remaining rows as 128x160 tiles (VkGemmDesc.m_begin / m_end; vk_gemm_tail_split names the row).
    Every output element sees the same MFMA sequence and the row-sum slabs are the same 160-column blocks, so (a) the launcher's own (split)
    launch, (b) the forced single launch (tile_cfg 7 disables the split) and (c) three explicit row-range calls that together cover the rows
    must agree BIT FOR BIT -- outputs and row sums -- and match torch fp32. Reference call sites as test_gemm_pipe_is_bitwise_..."""
    ops = _ops()
    Cc, S = 320, H * W
    M = n * S
    x = rnd(M, Cc)
    x3 = x.view(n, S, Cc)
    res = rnd(M, Cc, seed=3)
    rv = rnd(n, Cc, seed=5).float()
    if kind == "qkv_lnfold":
        nrm = _Norm(Cc, 7)
        w, b = rnd(3 * Cc, Cc, scale=Cc ** -0.5, seed=1), rnd(3 * Cc, seed=2).float()
        pw = ops.pack_linear(w, b, ln=nrm)
        st = ops.rowstats(x)
        fn = lambda **kw: ops.linear(x, pw, ln=st, **kw)  # noqa: E731
        ref = _ln_ref(x, nrm.weight, nrm.bias) @ w.float().t() + b
        N = 3 * Cc
    elif kind == "dense_K4N+res+stats":
        h4 = rnd(M, 4 * Cc, seed=9)
        w, b = rnd(Cc, 4 * Cc, scale=(4 * Cc) ** -0.5, seed=1), rnd(Cc, seed=2).float()
        pw = ops.pack_linear(w, b)
        fn = lambda **kw: ops.linear(h4, pw, res1=res, rowvec=rv, rows_per_vec=S, **kw)  # noqa: E731
        ref = h4.float() @ w.float().t() + b + res.float() + rv.repeat_interleave(S, 0)
        N = Cc
    elif kind == "conv3x3+emb+res":
        w, b = rnd(Cc, Cc, 3, 3, scale=(9 * Cc) ** -0.5, seed=1), rnd(Cc, seed=2).float()
        pw = ops.pack_conv3x3(w, b)
        fn = lambda **kw: ops.conv3x3(x3, pw, n, H, W, rowvec=rv, res1=x3, **kw)[0]  # noqa: E731
        ref = (_nchw2tok(F.conv2d(_tok2nchw(x3, n, H, W), w.float(), b, padding=1)) + rv[:, None, :] + x3.float()).reshape(M, Cc)
        N = Cc
    else:
        w, b = rnd(Cc, Cc, 3, 1, 1, scale=(3 * Cc) ** -0.5, seed=1), rnd(Cc, seed=2).float()
        pw = ops.pack_conv_t3(w, b)
        x5 = x3.float().view(1, n, S, 1, Cc).permute(0, 4, 1, 2, 3)
        ref = (0.3 * F.conv3d(x5, w.float(), b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(n, S, Cc) + x3.float()).reshape(M, Cc)
        fn = lambda **kw: ops.conv_t3(x3, pw, n, S, res2=x3, alpha=0.3, beta=1.0, **kw)  # noqa: E731
        N = Cc
    stats = kind == "dense_K4N+res+stats"
    ops.TILE_CFG = 64   # the launcher's own choice + the tail-split rule (an A/B option, measured without gain and off by default)
    try:
        auto = fn(emit_stats=True) if stats else fn()
    finally:
        ops.TILE_CFG = 0
    ops.TILE_CFG = 7
    try:
        single = fn(emit_stats=True) if stats else fn()
    finally:
        ops.TILE_CFG = 0
    if stats:
        (auto, sa), (single, ss) = auto, single
        assert sa.parts == ss.parts and torch.equal(sa.t, ss.t), "row-sum slabs of the split launch differ from the single launch"
        _check_stats(sa, auto)
    close(auto.reshape(M, N), ref, f"tail split {kind}")
    assert torch.equal(auto, single), "split and single launches must agree bit for bit"
    # the shape is one the rule splits: at least one full round and a last round filled to <= 40 %
    tiles_n, tiles_m = N // 320, (M + 255) // 256
    assert tiles_m * tiles_n // 256 >= 1 and 0 < tiles_m * tiles_n % 256 <= 0.4 * 256
    # (c) explicit row ranges: [0, a) on whatever the launcher picks, [a, b) and [b, M) likewise, written into one output
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    a, b_ = 256 * 100 + 64, M - 300   # deliberately not tile aligned
    for lo, hi in ((0, a), (a, b_), (b_, M)):
        ops.ROW_RANGE = (lo, hi)
        try:
            fn(out=out)
        finally:
            ops.ROW_RANGE = None
    assert torch.equal(out.reshape(-1), single.reshape(-1)), "row-range calls must reproduce the single launch bit for bit"
    # an invalid range is refused
    ops.ROW_RANGE = (M, M)
    try:
        with pytest.raises(Exception):
            fn()
    finally:
        ops.ROW_RANGE = None


def test_gemm_pipe_is_what_the_launcher_runs_and_refusals():
    """The launcher's own choice (vk_gemm_tile_choice) is the pipelined kernel wherever the 256x320 tile is, its split-K form and the fused
    nearest-x2 upsample included; what it does not take (halo frames of a frame-sharded run, fp32 output, two-source A) stays on the sixteen-wave
    kernels and a forced tile_cfg 7 falls back without an error."""
    import ctypes as C
    from vista_amd import _lib
    ops = _ops()
    lib = _lib.load()
    n, H, W, Cc = 4, 18, 32, 320
    x3 = rnd(n, H * W, Cc)
    pw = ops.pack_conv3x3(rnd(Cc, Cc, 3, 3, scale=(9 * Cc) ** -0.5, seed=1), rnd(Cc, seed=2).float())
    ref = _nchw2tok(F.conv2d(F.interpolate(_tok2nchw(x3, n, H, W), scale_factor=2, mode="nearest"), rnd(Cc, Cc, 3, 3, scale=(9 * Cc) ** -0.5, seed=1).float().cuda(),
                             rnd(Cc, seed=2).float().cuda(), padding=1))
    ops.TILE_CFG = 7
    try:
        up = ops.conv3x3(x3, pw, n, H, W, ups=2, out_f32=True)[0]      # fp32 output: not taken, falls back
    finally:
        ops.TILE_CFG = 0
    close(up, ref, "forced pipelined variant on a launch it does not take")
    one = C.c_void_p(4096)
    d = _lib.VkGemmDesc()
    d.A = d.Wt = d.out = one
    d.M, d.N, d.K, d.lda, d.ldc, d.alpha = 50 * 9216, 320, 2880, 2880, 320, 1.0
    d.amode, d.epi = ops.AMODE_CONV3X3, ops.EPI_LINEAR
    d.Cin, d.H, d.Wd, d.Hout, d.Wout, d.stride, d.ups = 320, 72, 128, 72, 128, 1, 1
    assert lib.vk_gemm_tile_choice(C.byref(d)) == 7 * 16 + 1
    d.ups = 2
    d.Hout, d.Wout, d.M = 144, 256, 50 * 144 * 256
    assert lib.vk_gemm_tile_choice(C.byref(d)) == 7 * 16 + 1
    d.halo_prev = one   # (a TEMPORAL3 field: any loader the kernel does not take)
    d.amode, d.K, d.T, d.S = ops.AMODE_TEMPORAL3, 960, 25, 144 * 256
    assert lib.vk_gemm_tile_choice(C.byref(d)) == 4 * 16 + 1


# ------------------------------------------------------------------------------------------------ round 2: folded LayerNorm, row sums,
# two-source (concat) loaders, rowvec2
def _ln_ref(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps)


class _Norm:  # stands in for the LayerNorm parameter container
    def __init__(self, C, seed):
        self.weight = (1.0 + 0.2 * torch.randn(C, generator=torch.Generator().manual_seed(seed))).cuda()
        self.bias = (0.3 * torch.randn(C, generator=torch.Generator().manual_seed(seed + 1))).cuda()
        self.eps = 1e-5


def _check_stats(st, out):
    """RowStats slabs summed over parts == (sum, sum of squares) of the bf16 output rows."""
    o = out.float()
    got = st.t.sum(0)
    ref = torch.stack([o.sum(1), o.pow(2).sum(1)], 1)
    tol = 2e-5 * torch.stack([o.abs().sum(1), o.pow(2).sum(1)], 1) + 1e-6
    assert (got - ref).abs().le(tol).all(), f"row sums off by {(got - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("M,N,K", [(777, 320, 320), (300, 640, 1280), (513, 1280, 640), (100, 64, 128)])
def test_linear_emit_rowstats(cfg, M, N, K):
    ops = _ops()
    x = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    b = rnd(N, seed=2).float()
    r1 = rnd(M, N, seed=4)
    rv = rnd(3, N, seed=5).float()
    ops.TILE_CFG = cfg
    try:
        out, st = ops.linear(x, ops.pack_linear(w, b), res1=r1, rowvec=rv, rows_per_vec=(M + 2) // 3, emit_stats=True)
    finally:
        ops.TILE_CFG = 0
    ref = x.float() @ w.float().t() + b + r1.float() + rv.repeat_interleave((M + 2) // 3, 0)[:M]
    close(out, ref, f"linear+stats cfg{cfg} {M}x{N}x{K}")
    assert st.M == M and st.t.shape == (st.parts, M, 2)
    _check_stats(st, out)


@pytest.mark.parametrize("rows,C", [(1000, 320), (257, 640), (129, 1280), (64, 64)])
def test_rowstats(rows, C):
    ops = _ops()
    x = rnd(rows, C) + 2.0
    st = ops.rowstats(x)
    assert st.parts == 1
    _check_stats(st, x)
    big = rnd(rows, 2 * C, seed=3)
    _check_stats(ops.rowstats(big[:, C:]), big[:, C:])  # strided rows


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("M,C,N", [(600, 320, 640), (300, 1280, 1280), (130, 64, 192)])
def test_linear_layernorm_fold(cfg, M, C, N):
    """Linear(LayerNorm(x)) with the norm folded into the GEMM: x has a LARGE row mean (the fold subtracts mean * colsum in the epilogue)."""
    ops = _ops()
    x = (rnd(M, C, scale=1.5) + rnd(M, 1, scale=4.0, seed=7)).to(BF16)
    w = rnd(N, C, scale=C ** -0.5, seed=1)
    b = rnd(N, seed=2).float()
    norm = _Norm(C, 11)
    pw = ops.pack_linear(w, b, ln=norm)
    ref = _ln_ref(x, norm.weight, norm.bias) @ w.float().t() + b
    ops.TILE_CFG = cfg
    try:
        out = ops.linear(x, pw, ln=ops.rowstats(x))
    finally:
        ops.TILE_CFG = 0
    close(out, ref, f"ln-fold linear cfg{cfg} {M}x{C}->{N}")
    with pytest.raises(ValueError):
        ops.linear(x, pw)  # a folded weight without the row sums
    with pytest.raises(ValueError):
        ops.linear(x, ops.pack_linear(w, b), ln=ops.rowstats(x))


@pytest.mark.parametrize("ratio", [8.0, 60.0])
@pytest.mark.parametrize("C", [320, 1280])
def test_linear_layernorm_fold_large_mean_over_std(ratio, C):
    """the fold takes the variance as E[x^2] - mean^2 from fp32 row sums; rows with |mean| >> std lose precision to
    cancellation. Stated range: |mean| <= 60 std stays inside the kernel tolerance (fp32 sums over C <= 1280 columns resolve the
    variance to ~1 %); the row sums come from BOTH producers of the product path (the read-only pass and a GEMM epilogue)."""
    ops = _ops()
    M, N = 700, 640
    g = torch.Generator(device="cpu").manual_seed(5)
    sign = (torch.randint(0, 2, (M, 1), generator=g) * 2 - 1).float()
    x = (torch.randn(M, C, generator=g) + ratio * sign).to(BF16).cuda()   # per-row mean = +-ratio, std ~1 (bf16 grid: 0.25-0.5 at 60)
    w = rnd(N, C, scale=C ** -0.5, seed=1)
    b = rnd(N, seed=2).float()
    norm = _Norm(C, 11)
    pw = ops.pack_linear(w, b, ln=norm)
    ref = _ln_ref(x, norm.weight, norm.bias) @ w.float().t() + b
    out = ops.linear(x, pw, ln=ops.rowstats(x))
    close(out, ref, f"ln-fold, mean/std {ratio}, C {C}, rowstats pass")
    # the same rows produced by a GEMM epilogue (identity weight): its (sum, sum of squares) slabs feed the consumer
    eye = ops.pack_linear(torch.eye(C), None)
    x2, st = ops.linear(x, eye, emit_stats=True)
    assert torch.equal(x2, x)
    close(ops.linear(x2, pw, ln=st), ref, f"ln-fold, mean/std {ratio}, C {C}, producer-epilogue stats")


def test_layernorm_fold_chain_producer_stats():
    """producer GEMM (emits row sums of its output) -> consumer GEMMs with the folded norm: LINEAR (q|k), TRANS (v^T), GEGLU."""
    ops = _ops()
    n_img, S, C = 3, 176, 320
    M = n_img * S
    x0 = rnd(M, C)
    w0 = rnd(C, C, scale=C ** -0.5, seed=1)
    res = rnd(M, C, seed=3) + 1.5
    x, st = ops.linear(x0, ops.pack_linear(w0, None), res1=res, emit_stats=True)
    _check_stats(st, x)
    norm = _Norm(C, 21)
    y = _ln_ref(x, norm.weight, norm.bias)
    wq, wk, wv = (rnd(C, C, scale=C ** -0.5, seed=s) for s in (4, 5, 6))
    qk = ops.linear(x, ops.pack_linear_cat([wq, wk], ln=norm), ln=st)
    close(qk, torch.cat([y @ wq.float().t(), y @ wk.float().t()], 1), "folded q|k")
    vt = ops.linear_vt(x, ops.pack_linear(wv, None, ln=norm), S, ln=st)
    close(vt, (y @ wv.float().t()).view(n_img, S, C).transpose(1, 2), "folded v^T")
    wg = rnd(8 * C, C, scale=C ** -0.5, seed=8)
    bg = rnd(8 * C, seed=9).float()
    h = ops.linear(x, ops.pack_geglu(wg, bg, ln=norm), ln=st)
    a, g = (y @ wg.float().t() + bg).chunk(2, dim=-1)
    close(h, a * F.gelu(g), "folded GEGLU")
    # the same through the one-slab rowstats kernel is bitwise the same function of (mean, rstd) up to the slab summation order
    h2 = ops.linear(x, ops.pack_geglu(wg, bg, ln=norm), ln=ops.rowstats(x))
    close(h2, a * F.gelu(g), "folded GEGLU (rowstats kernel)")


def test_linear_two_source_concat():
    ops = _ops()
    M, C1, C2, N = 700, 640, 320, 320
    a, b = rnd(M, C1), rnd(M, C2, seed=1)
    w = rnd(N, C1 + C2, scale=(C1 + C2) ** -0.5, seed=2)
    bias = rnd(N, seed=3).float()
    pw = ops.pack_linear(w, bias)
    for cfg in (0, 1, 2, 3, 4):
        ops.TILE_CFG = cfg
        try:
            out = ops.linear(a, pw, x2=b)
            cat = ops.linear(torch.cat([a, b], 1).contiguous(), pw)
        finally:
            ops.TILE_CFG = 0
        assert torch.equal(out, cat), f"two-source loader must equal the GEMM over the materialised concat (cfg {cfg})"
    close(out, torch.cat([a, b], 1).float() @ w.float().t() + bias, "two-source linear")
    with pytest.raises(ValueError):
        ops.linear(a, pw, x2=b[:10])


@pytest.mark.parametrize("n,S,C1,C2,silu", [(3, 144, 640, 320, True), (2, 576, 320, 320, True), (2, 100, 64, 128, False), (4, 144, 1280, 1280, True)])
def test_groupnorm_cat_equals_groupnorm_of_concat(n, S, C1, C2, silu):
    """640+320 -> 30 channels per group: group 21 straddles the two tensors."""
    ops = _ops()
    a, b = rnd(n, S, C1) + 0.5, rnd(n, S, C2, seed=1, scale=2.0)
    C = C1 + C2
    gamma, beta = rnd(C, seed=2).float() + 1.0, rnd(C, seed=3).float()
    got = ops.groupnorm_cat(a, b, gamma, beta, 1e-5, silu)
    want = ops.groupnorm(torch.cat([a, b], 2).contiguous(), gamma, beta, 1e-5, silu)
    assert torch.equal(got, want), "same arithmetic in the same order: bitwise equal"
    ref = F.group_norm(torch.cat([a, b], 2).float().transpose(1, 2), 32, gamma, beta, 1e-5).transpose(1, 2)
    close(got, F.silu(ref) if silu else ref, "groupnorm_cat")


def test_groupnorm_large_mean():
    """|mean| ~ 30 std per group: the single-pass fp32 (sum, sum of squares) statistics must still resolve the variance."""
    ops = _ops()
    n, S, C = 2, 2304, 320
    g = torch.Generator().manual_seed(5)
    mean = 30.0 * (torch.rand(n, 1, 32, 1, generator=g) - 0.5).sign() * (0.5 + torch.rand(n, 1, 32, 1, generator=g))
    x = (torch.randn(n, S, 32, C // 32, generator=g) + mean).reshape(n, S, C).to(BF16).cuda()
    gamma, beta = rnd(C, seed=2).float() + 1.0, rnd(C, seed=3).float()
    got = ops.groupnorm(x, gamma, beta, 1e-5, silu=False)
    ref = F.group_norm(x.float().transpose(1, 2), 32, gamma, beta, 1e-5).transpose(1, 2)
    close(got, ref, "groupnorm |mean| = 15..45 std")
    # and over 25 frames (the temporal ResBlock's 5-D statistics), 2.3e5 elements per group
    x5 = x[:1].repeat(25, 1, 1).contiguous()
    got5 = ops.groupnorm(x5, gamma, beta, 1e-5, silu=False, frames_per_group=25)
    close(got5[:1], ref[:1], "groupnorm 5-D |mean| >> std")


def test_rowvec2_blend_epilogue():
    """out = alpha*(acc + bias + res1) + beta*(res2 + rowvec2): the AlphaBlender epilogue that recovers x_spatial = x_mix - emb."""
    ops = _ops()
    M, N, K, rpv = 600, 320, 1280, 200
    x = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5, seed=1)
    b = rnd(N, seed=2).float()
    r1, r2 = rnd(M, N, seed=4), rnd(M, N, seed=5)
    rv2 = rnd(M // rpv, N, seed=6).float()
    out = ops.linear(x, ops.pack_linear(w, b), res1=r1, res2=r2, rowvec2=rv2, rows_per_vec=rpv, alpha=0.4, beta=0.6)
    ref = 0.4 * (x.float() @ w.float().t() + b + r1.float()) + 0.6 * (r2.float() + rv2.repeat_interleave(rpv, 0))
    close(out, ref, "rowvec2 epilogue")
    with pytest.raises(ValueError):
        ops.linear(x, ops.pack_linear(w, b), rowvec2=rv2, rows_per_vec=rpv)


def test_packed_weights_follow_parameter_versions():
    """packs must be rebuilt after a load_state_dict issued on a PARENT container and after in-place updates."""
    import torch.nn as nn
    from vista_amd.modules.attention import FeedForward
    ff = FeedForward(64, glu=True).cuda()
    holder = nn.ModuleDict({"inner": nn.ModuleList([ff])})
    x = rnd(100, 64)
    y0 = ff(x)
    pk0 = ff.packed()
    assert ff.packed() is pk0, "unchanged parameters: the cached pack is reused"
    sd = {k: torch.randn_like(v) * 0.2 for k, v in holder.state_dict().items()}
    holder.load_state_dict(sd)                      # never calls ff.load_state_dict
    assert ff.packed() is not pk0
    y1 = ff(x)
    ref = (lambda h: (h[:, :256] * F.gelu(h[:, 256:])) @ sd["inner.0.net.2.weight"].float().t() + sd["inner.0.net.2.bias"])(
        x.float() @ sd["inner.0.net.0.proj.weight"].float().t() + sd["inner.0.net.0.proj.bias"])
    close(y1, ref.to(BF16).float(), "FeedForward after parent load_state_dict", rtol=3e-2, arel=3e-2)
    assert not torch.equal(y0, y1)
    pk1 = ff.packed()
    with torch.no_grad():
        ff.net[2].weight.mul_(0.5)                  # in-place update on the Parameter: bumps its version counter
    assert ff.packed() is not pk1
    pk2 = ff.packed()
    ff.net[2].weight.data.mul_(2.0)                 # `.data` has its own version counter BY DESIGN: invisible, needs the explicit call
    assert ff.packed() is pk2
    from vista_amd.modules.attention import invalidate_packed
    invalidate_packed(holder)
    assert ff.packed() is not pk2
    # a REPLACED Parameter object (load_state_dict(assign=True), re-assignment): the old object's version would never change
    pk3 = ff.packed()
    ff.net[2].weight = nn.Parameter(ff.net[2].weight.detach() * 3.0)
    assert ff.packed() is not pk3
    pk4 = ff.packed()
    holder.load_state_dict({k: v.clone() for k, v in holder.state_dict().items()}, assign=True)
    assert ff.packed() is not pk4
    y2 = ff(x)
    ref2 = (lambda h: (h[:, :256] * F.gelu(h[:, 256:])) @ (3.0 * sd["inner.0.net.2.weight"].float()).t() + sd["inner.0.net.2.bias"])(
        x.float() @ sd["inner.0.net.0.proj.weight"].float().t() + sd["inner.0.net.0.proj.bias"])
    close(y2, ref2.to(BF16).float(), "FeedForward after Parameter replacement", rtol=3e-2, arel=3e-2)


# ------------------------------------------------------------------------------------------------ fused FeedForward (level 0)
class _Norm:
    def __init__(self, C, seed=7):
        g = torch.Generator().manual_seed(seed)
        self.weight = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
        self.bias = (0.1 * torch.randn(C, generator=g)).cuda()
        self.eps = 1e-5


def _ff_reference(x, w1, b1, w2, b2, norm):
    """FeedForward.net = [GEGLU, Dropout, Linear] (vwm/modules/attention.py:85-128) in fp32, with the hidden activation rounded to bf16
    where both kernel forms round it."""
    xf = x.float()
    if norm is not None:
        xf = F.layer_norm(xf, (xf.shape[1],), norm.weight, norm.bias, norm.eps)
    a, g = (xf @ w1.float().t() + b1).chunk(2, dim=1)
    return (a * F.gelu(g)).to(BF16).float() @ w2.float().t() + b2


@pytest.mark.parametrize("M", [128, 1000, 4173, 33000])   # one tile, ragged, several tiles, more tiles than CUs
@pytest.mark.parametrize("ln", [False, True])
@pytest.mark.parametrize("mode", ["plain", "res", "blend", "rowvec"])
def test_ff_fused_matches_reference_and_two_kernel_form(M, ln, mode):
    """vk_ff_fused_bf16 (GEGLU in-projection -> gelu -> out-projection in one launch) against the fp32 reference of the same FeedForward and
    against the two-GEMM form it replaces, for every epilogue the UNet uses: attention.py:524 (+x, + frame embedding, row sums for the next
    LayerNorm), video_attention.py:119-121 (ff_in) and :137-141 (AlphaBlender mix: alpha, res2, rowvec2, beta)."""
    ops = _ops()
    C, H = 320, 1280
    x = rnd(M, C)
    w1 = rnd(2 * H, C, scale=C ** -0.5, seed=1)
    b1 = rnd(2 * H, seed=2).float() * 0.5
    w2 = rnd(C, H, scale=H ** -0.5, seed=3)
    b2 = rnd(C, seed=4).float()
    norm = _Norm(C) if ln else None
    pin, pout, pfo = ops.pack_geglu(w1, b1, ln=norm), ops.pack_linear(w2, b2), ops.pack_ff_out(w2, b2)
    st = ops.rowstats(x) if ln else None
    ref = _ff_reference(x, w1, b1, w2, b2, norm)
    kw = {}
    if mode == "res":
        kw = dict(res1=x, emit_stats=True)
        ref = ref + x.float()
    elif mode == "blend":
        S = 100
        xm, rv2 = rnd(M, C, seed=5), rnd((M + S - 1) // S, C, seed=6).float()
        kw = dict(res1=x, alpha=0.4, res2=xm, rowvec2=rv2, beta=0.6, rows_per_vec=S)
        ref = 0.4 * (ref + x.float()) + 0.6 * (xm.float() + rv2.repeat_interleave(S, 0)[:M])
    elif mode == "rowvec":
        S = 64
        rv = rnd((M + S - 1) // S, C, seed=8).float()
        kw = dict(res1=x, rowvec=rv, rows_per_vec=S, emit_stats=True)
        ref = ref + x.float() + rv.repeat_interleave(S, 0)[:M]
    two = ops.linear(ops.linear(x, pin, ln=st), pout, **kw)
    fus = ops.ff_fused(x, pin, pfo, ln=st, **kw)
    if isinstance(fus, tuple):
        (two, st2), (fus, stf) = two, fus
        assert stf.parts == 2 and stf.t.shape == (2, M, 2)
        s2, sf = st2.t.sum(0), stf.t.sum(0)     # the row sums of the bf16-rounded outputs feed the next LayerNorm fold
        assert ((s2 - sf).abs().max() / s2.abs().max()).item() < 1e-3
        o = fus.float()
        assert torch.allclose(sf[:, 0], o.sum(1), rtol=1e-3, atol=2e-2) and torch.allclose(sf[:, 1], (o * o).sum(1), rtol=1e-3, atol=2e-2)
    close(fus, ref, f"ff_fused {M} ln={ln} {mode}")
    # same products, same bf16 rounding of the hidden activation; only the fp32 summation order of the out-projection differs
    assert ((fus.float() - two.float()).norm() / two.float().norm()).item() < 2e-3


def test_ff_fused_other_hidden_width_strided_input_and_refusals():
    ops = _ops()
    M, C, H = 777, 320, 640
    xw = rnd(M, 2 * C)
    x = xw[:, :C]                                # a strided view (row stride 640): lda != K
    w1, b1 = rnd(2 * H, C, scale=C ** -0.5, seed=1), rnd(2 * H, seed=2).float()
    w2, b2 = rnd(C, H, scale=H ** -0.5, seed=3), rnd(C, seed=4).float()
    out = ops.ff_fused(x, ops.pack_geglu(w1, b1), ops.pack_ff_out(w2, b2), res1=x)
    close(out, _ff_reference(x, w1, b1, w2, b2, None) + x.float(), "ff_fused H=640, strided x")
    with pytest.raises(ValueError):              # a vk_gemm_bf16 weight is not a fused-kernel operand and vice versa
        ops.ff_fused(x, ops.pack_geglu(w1, b1), ops.pack_linear(w2, b2))
    with pytest.raises(ValueError):
        ops.linear(rnd(M, H), ops.pack_ff_out(w2, b2))
    with pytest.raises(ValueError):              # width 640: not covered
        ops.ff_fused(rnd(M, 640), ops.pack_geglu(rnd(5120, 640), rnd(5120).float()), ops.pack_ff_out(rnd(640, 2560), None))


def test_ff_fused_is_bitwise_repeatable_and_batch_independent():
    ops = _ops()
    C, H = 320, 1280
    x = rnd(3000, C)
    pin, pfo = ops.pack_geglu(rnd(2 * H, C, scale=C ** -0.5, seed=1), rnd(2 * H, seed=2).float()), ops.pack_ff_out(rnd(C, H, scale=H ** -0.5, seed=3), None)
    a = ops.ff_fused(x, pin, pfo, res1=x)
    b = ops.ff_fused(x, pin, pfo, res1=x)
    assert torch.equal(a, b)
    c = ops.ff_fused(x[1280:2560], pin, pfo, res1=x[1280:2560])   # rows are independent: a sub-range computes the same bits
    assert torch.equal(a[1280:2560], c)


# ------------------------------------------------------------------------------------------------ weight-stationary streaming GEMM (level-0 K = 320 projections)
@pytest.mark.parametrize("M,S", [(32 * 300 + 7, 288), (70016, 9216 // 4)])   # ragged last tile; more row tiles than workgroups
@pytest.mark.parametrize("kind", ["plain", "res", "stats", "res+rowvec+stats", "qkv_lnfold", "n640"])
def test_gemm_stream_forced_matches_reference_and_tiled_kernel(M, S, kind):
    """gemm_stream.hip (VkGemmDesc.tile_cfg = 6: weights in registers, activations / residual / LayerNorm statistics through LDS-DMA rings
    with counted vmcnt waits) for every epilogue it supports, against torch fp32 and against the tiled kernel (bitwise: same accumulation
    order, same epilogue arithmetic). Reference call sites: attention.py:344-346 (q|k|v), :421 (to_out), :579,602 (proj_in / proj_out)."""
    ops = _ops()
    C = 320
    x = rnd(M, C)
    res = rnd(M, C, seed=3)
    kw, ln, N = {}, None, C
    if kind == "qkv_lnfold":
        N = 3 * C
    elif kind == "n640":
        N = 2 * C
    w, b = rnd(N, C, scale=C ** -0.5, seed=1), rnd(N, seed=2).float()
    ref = x.float() @ w.float().t() + b
    if kind == "qkv_lnfold":
        nrm = _Norm(C)
        ln = ops.rowstats(x)
        pw = ops.pack_linear(w, b, ln=nrm)
        ref = F.layer_norm(x.float(), (C,), nrm.weight, nrm.bias, nrm.eps) @ w.float().t() + b
    else:
        pw = ops.pack_linear(w, b)
    if kind in ("res", "res+rowvec+stats"):
        kw["res1"] = res
        ref = ref + res.float()
    if kind == "res+rowvec+stats":
        rv = rnd((M + S - 1) // S, C, seed=5).float()
        kw.update(rowvec=rv, rows_per_vec=S)
        ref = ref + rv.repeat_interleave(S, 0)[:M]
    if "stats" in kind:
        kw["emit_stats"] = True
    outs = {}
    for cfg in (6, 4):
        ops.TILE_CFG = cfg
        try:
            outs[cfg] = ops.linear(x, pw, ln=ln, **kw)
        finally:
            ops.TILE_CFG = 0
    o6, o4 = outs[6], outs[4]
    if "stats" in kind:
        (o6, s6), (o4, s4) = o6, o4
        assert s6.parts == 1, "the streaming kernel combines its waves' row sums into one slab"
        of = o6.float()
        assert torch.allclose(s6.t[0, :, 0], of.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(s6.t[0, :, 1], (of * of).sum(1), rtol=1e-4, atol=1e-2)
    close(o6, ref, f"gemm_stream {kind} M={M}")
    assert torch.equal(o6, o4), "streaming and tiled kernels must agree bit for bit"


def test_gemm_stream_is_chosen_where_it_wins_and_only_there():
    """The launcher's rule (vk_gemm_tile_choice: variant * 16 + K slices): N = 320 + residual at BASELINE size -> variant 6; anything else stays tiled."""
    import ctypes as C
    ops = _ops()
    lib = ops._lib.load()
    x = rnd(65536 * 2, 320)
    pw = ops.pack_linear(rnd(320, 320, seed=1), rnd(320, seed=2).float())
    out = torch.empty_like(x)

    def choice(**kw):
        d = ops.VkGemmDesc()
        d.A, d.lda, d.amode, d.epi = ops._p(x), 320, 0, 0
        ops._fill_epilogue(d, pw, out, x.shape[0], kw.get("rowvec"), kw.get("rows_per_vec", 0), kw.get("res1"), None, 1.0, 0.0)
        if kw.get("stats"):
            d.rowstat_out = ops._p(torch.empty(4, device="cuda"))
        return lib.vk_gemm_tile_choice(C.byref(d)) // 16
    if os.environ.get("VISTA_GEMM_STREAM", "1") != "0":
        assert choice(res1=x) == 6
    assert choice() != 6 and choice(res1=x, stats=True) != 6
