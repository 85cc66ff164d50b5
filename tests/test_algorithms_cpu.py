"""CPU restatements of two device-side algorithms whose correctness argument is not "the same arithmetic in another order":

* the MAX-FREE fast path of the spatial attention kernel (vista_amd/csrc/attention.hip): exponentials taken against the existing
  softmax base, the tile validated afterwards through its row sums, redone with a re-base only when they exceed the limit;
* the per-lane (tap, slab) stepping of the fp8 implicit-GEMM loaders (vista_amd/csrc/gemm_fp8.hip): two (slab, tap) units per
  128-byte K-step, each lane advancing its own unit by two per step.

* the block-scale rule of the GEGLU epilogue's MX fp8 output (gemm_fp8.hip): frexp of max|h| / 448 gives the smallest power of two
  that keeps the block inside e4m3's range.

All are emulated here in plain torch on small cases and compared with the closed forms (exact softmax attention; the
[Cin/64][tap][64] K order of the packed weights). The GPU tests check the kernels themselves."""
import math

import pytest
import torch


def _attention_max_free(q, k, v, tile=64, thr=6.0):
    """fp32 emulation of the kernel's tile loop for ONE query block (rows of q), keys streamed in tiles of `tile`.
    Returns (out, number of tiles that took the slow path)."""
    scale_log2 = q.shape[-1] ** -0.5 * math.log2(math.e)
    n = q.shape[0]
    m_run = torch.full((n,), -1e30)
    l_run = torch.zeros(n)
    o = torch.zeros(n, v.shape[-1])
    limit = 2.0 ** (min(thr, 100.0) + 5.0)
    slow_tiles = 0
    for t0 in range(0, k.shape[0], tile):
        s = q @ k[t0:t0 + tile].t()                                # scores of the tile

        def rebase():
            nonlocal m_run, l_run, o
            mx = torch.maximum((s.amax(1) * scale_log2), torch.tensor(-1e30))
            m_new = torch.maximum(m_run, mx)
            alpha = torch.exp2(m_run - m_new)
            m_run, l_run, o = m_new, l_run * alpha, o * alpha[:, None]
        if t0 == 0:
            rebase()
        p = torch.exp2(s * scale_log2 - m_run[:, None]).to(torch.bfloat16).float()   # rounded like the MFMA operand
        # the kernel tests every LANE's partial sum (32 of the 64 keys); testing per row halves is the same predicate here
        halves = torch.stack([p[:, :tile // 2].sum(1), p[:, tile // 2:].sum(1)], 1)
        if not bool((halves <= limit).all()):                       # wave-uniform in the kernel: any failing lane redoes the tile
            slow_tiles += 1
            rebase()
            p = torch.exp2(s * scale_log2 - m_run[:, None]).to(torch.bfloat16).float()
        l_run = l_run + p.sum(1)
        o = o + p @ v[t0:t0 + tile].to(torch.bfloat16).float()
    return o / l_run[:, None], slow_tiles


@pytest.mark.parametrize("gain", [0.0, 3.0, 12.0, 60.0, 400.0])
def test_attention_max_free_fast_path_equals_softmax(gain):
    g = torch.Generator().manual_seed(int(gain) + 1)
    S, d = 640, 64
    q = torch.randn(32, d, generator=g)
    k = torch.randn(S, d, generator=g)
    v = torch.randn(S, d, generator=g)
    if gain:
        for j, row in enumerate(range(70, S, 97)):                 # late keys strongly aligned with single queries
            k[row] = q[(5 * j + 3) % 32] * gain
        k[:64] = -q[9] * gain                                       # query 9: first tile far below its later scores
        k[S - 1] = q[9] * gain
    q[11] = 0                                                       # uniform softmax row
    out, slow = _attention_max_free(q, k, v)
    ref = torch.softmax((q @ k.t()) * d ** -0.5, -1) @ v.to(torch.bfloat16).float()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max() <= 2e-2 * ref.abs().max() + 1e-3
    if gain == 0.0:
        assert slow == 0, "Gaussian scores must never leave the fast path after tile 0"
    if gain >= 60.0:
        assert slow >= 1, "scores that overflow exp2 against the old base must be caught by the row-sum test"


@pytest.mark.parametrize("taps,cin", [(9, 320), (9, 64), (9, 128), (3, 320), (3, 64), (3, 1280), (9, 960)])
def test_fp8_conv_loader_unit_stepping_matches_k_order(taps, cin):
    """Lane state of gemm_fp8.hip's implicit-GEMM loaders: chunk j (0..7) of a 128-byte K-step belongs to unit 2*kt + (j >> 2) of the
    [Cin/64][tap][64] K order; the lane keeps (tap, channel base) and adds two units per step. Emulates the stepping for every
    chunk and checks each staged 16-byte chunk against the closed form, including the empty second half of an odd last step."""
    n_units = taps * (cin // 64)
    ksteps = (n_units + 1) // 2
    for j in range(8):
        tap, cbase = j >> 2, 0                                     # initial lane state (tap_l = lsrc >> 2, cbase_l = 0)
        for kt in range(ksteps):
            inside = cbase < cin
            unit = 2 * kt + (j >> 2)
            assert inside == (unit < n_units)
            if inside:
                # K byte offset this chunk lands at vs the packed weight's K index of (slab, tap, channel)
                k_index = kt * 128 + j * 16
                slab, t = divmod(unit, taps)
                assert (tap, cbase) == (t, slab * 64)
                assert k_index == (slab * taps + t) * 64 + (j & 3) * 16
            tap += 2
            if tap >= taps:
                tap -= taps
                cbase += 64


def test_mx_block_scale_rule_is_the_tightest_power_of_two():
    """GEGLU epilogue, MX fp8 output: per 32-column block `frexpf(amax / 448, &ex)` (amax / 448 = f * 2^ex, f in [0.5, 1)) and the
    E8M0 byte ex + 127. For every positive amax: amax / 2^ex <= 448 (nothing saturates), amax / 2^(ex-1) > 448 (no smaller power of
    two would do), the scaled values round to finite e4m3 codes, and the clamps keep the byte inside [0, 254] (255 is NaN in E8M0)."""
    g = torch.Generator().manual_seed(0)
    amax = torch.cat([torch.exp2(torch.rand(4000, generator=g) * 80 - 40), torch.tensor([448.0, 224.0, 1.0, 447.99, 448.01, 1e-38, 3e38])])
    f, ex = torch.frexp(amax / 448.0)
    assert ((f >= 0.5) & (f < 1.0)).all()
    ex = ex.clamp(-127, 127)
    two_e = torch.exp2(ex.float())
    ok = ex > -127                                                   # (the clamp only bites below 2^-127 * 448)
    assert (amax[ok] / two_e[ok] <= 448.0).all() and (amax[ok] / (two_e[ok] / 2) > 448.0 * (1 - 1e-6)).all()
    codes = (amax[ok] / two_e[ok]).to(torch.float8_e4m3fn).float()
    assert torch.isfinite(codes).all() and (codes >= 208).all() and (codes <= 448).all()
    byte = ex + 127
    assert (byte >= 0).all() and (byte <= 254).all()


def test_epilogue_groupnorm_statistics_lane_algorithm():
    """Lane-level emulation of csrc/gemm_common.h::gn_accumulate_quad / gn_reduce_store (GroupNorm statistics emitted by a convolution's
    epilogue, ABI v6): one 64-row x 160-column wave tile, lane (l31, lh) holds columns 32 fi + 8 g + 4 lh + e of rows 32 fj + l31. Pins the
    index arithmetic the GPU tests can only observe from outside: which pairs straddle two channel groups between the half-waves (16 of 40 with
    10-channel groups, 8 with 20, none with 40), that the transposing butterfly (v_permlane32_swap, then xor 16 / 8 / 4 / 2 / 1 with complementary halves exchanged) leaves value
    32 (L >> 5) + ((L & 31) >> SH) in lane L, and that every slot entry of the wave's groups is written exactly once."""
    import numpy as np
    rng = np.random.default_rng(0)
    lanes = np.arange(64)
    for CPG in (10, 20, 40):
        NG = 160 // CPG
        X = rng.standard_normal((64, 160))
        gs, gq = np.zeros((64, NG)), np.zeros((64, NG))
        straddles = 0
        for lane in range(64):
            l31, lh = lane & 31, lane >> 5
            for fj in range(2):
                for fi in range(5):
                    for g in range(4):
                        for h in range(2):
                            c0 = 32 * fi + 8 * g + 2 * h
                            G0, G1 = c0 // CPG, (c0 + 4) // CPG
                            pair = X[32 * fj + l31, c0 + 4 * lh: c0 + 4 * lh + 2]
                            if G0 == G1:
                                gs[lane, G0] += pair.sum(); gq[lane, G0] += (pair ** 2).sum()
                            else:   # masked by half-wave: the lane's pair goes to its own group, zeros to the other
                                straddles += lane == 0 and fj == 0
                                own = G0 if lh == 0 else G1
                                gs[lane, own] += pair.sum(); gq[lane, own] += (pair ** 2).sum()
        assert straddles == {10: 16, 20: 8, 40: 0}[CPG]   # of a lane's 40 pairs per 32-row block

        def swap32(a, b):   # v_permlane32_swap(vdst = a, src = b): a's upper half <-> b's lower half
            a2, b2 = a.copy(), b.copy()
            a2[32:], b2[:32] = b[:32], a[32:]
            return a2, b2
        r = []
        for i in range(NG):
            s0, s1 = swap32(gs[:, i], gq[:, i])
            r.append(s0 + s1)
        k, n = 16, NG
        while n > 1:
            up = (lanes & k) != 0
            for i in range(n // 2):
                keep, give = np.where(up, r[i + n // 2], r[i]), np.where(up, r[i], r[i + n // 2])
                r[i] = keep + give[lanes ^ k]
            n //= 2
            k //= 2
        while k >= 1:
            r[0] = r[0] + r[0][lanes ^ k]
            k //= 2
        SH = {16: 1, 8: 2, 4: 3}[NG]
        slot = np.full(64, np.nan)
        for lane in range(64):
            if lane & ((1 << SH) - 1) == 0:
                e = 32 * (lane >> 5) + ((lane & 31) >> SH)
                assert np.isnan(slot[e])
                slot[e] = r[0][lane]
        ref_s = X.reshape(64, NG, CPG).sum(axis=(0, 2))
        ref_q = (X ** 2).reshape(64, NG, CPG).sum(axis=(0, 2))
        assert np.allclose(slot[:NG], ref_s, rtol=1e-10, atol=1e-9) and np.allclose(slot[32:32 + NG], ref_q, rtol=1e-10)
        assert np.isnan(slot[NG:32]).all() and np.isnan(slot[32 + NG:]).all()
