"""Config 5 (fp8 e4m3) piece by piece against the round-4 bf16 kernels at the BASELINE shapes: does each fp8 piece still pay, and what does it
cost in accuracy? Times the FeedForward (LayerNorm folded / quantised), the ResBlock GroupNorm+SiLU -> conv3x3 pair and the attention-out
projection in both forms; error of each against fp32 torch.   usage: python tools/fp8_vs_bf16_probe.py [images]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402
from tools.gemm_sweep2 import Norm, timeit  # noqa: E402

BF16 = torch.bfloat16


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm()).item()


def best(fn, n=3):
    return min(timeit(fn) for _ in range(n))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    torch.manual_seed(0)
    rn = lambda *s: torch.randn(*s, device="cuda")  # noqa: E731
    for C, H, W in ((320, 72, 128), (640, 36, 64), (1280, 18, 32)):
        S = H * W
        M = N * S
        x = rn(M, C).to(BF16)
        res = rn(M, C).to(BF16)
        nrm = Norm(C)
        nrm.weight, nrm.bias = 1 + 0.2 * rn(C), 0.3 * rn(C)
        w1, b1 = rn(8 * C, C) * C ** -0.5, rn(8 * C)
        w2, b2 = rn(C, 4 * C) * (4 * C) ** -0.5, rn(C)
        # ---- FeedForward: LayerNorm -> GEGLU -> Linear + residual
        idx = torch.randperm(M, device="cuda")[:4096]
        a, g = (F.layer_norm(x[idx].float(), (C,), nrm.weight, nrm.bias, nrm.eps) @ w1.t() + b1).chunk(2, -1)
        ref = (a * F.gelu(g)) @ w2.t() + b2 + res[idx].float()
        pin = ops.pack_geglu(w1, b1, ln=nrm)
        st = ops.rowstats(x)
        fused = C == 320
        pout = ops.pack_ff_out(w2, b2, "cuda") if fused else ops.pack_linear(w2, b2)
        f16 = (lambda: ops.ff_fused(x, pin, pout, ln=st, res1=res)) if fused else (lambda: ops.linear(ops.linear(x, pin, ln=st), pout, res1=res))
        in8, out8 = ops.pack_geglu_fp8(w1, b1, "cuda"), ops.pack_linear_fp8(w2, b2, "cuda")

        def f8():
            yq, ys = ops.layernorm_quant_fp8(x, nrm)
            h8, hs = ops.linear_fp8(yq, ys, in8, mx_out=True)
            return ops.linear_fp8(h8, None, out8, a_mx=hs, res1=res)
        o16, o8 = f16(), f8()
        print(f"C {C:4d} FeedForward            bf16 {best(f16):.4f} ms (err {rel(o16[idx], ref):.2e})   fp8 {best(f8):.4f} ms (err {rel(o8[idx], ref):.2e})", flush=True)
        # ---- attention-out projection (K = N = C) + residual, input = attention output
        wo, bo = rn(C, C) * C ** -0.5, rn(C)
        refp = x[idx].float() @ wo.t() + bo + res[idx].float()
        po = ops.pack_linear(wo, bo)
        p16 = lambda: ops.linear(x, po, res1=res, emit_stats=True)  # noqa: E731
        po8 = ops.pack_linear_fp8(wo, bo, "cuda")

        def p8():
            xq, xs = ops.quantize_rows_fp8(x)
            return ops.linear_fp8(xq, xs, po8, res1=res)
        q16, q8 = p16()[0], p8()
        print(f"C {C:4d} attention out-proj     bf16 {best(p16):.4f} ms (err {rel(q16[idx], refp):.2e})   fp8 incl. row quantisation {best(p8):.4f} ms (err {rel(q8[idx], refp):.2e})", flush=True)
        # ---- ResBlock half: GroupNorm + SiLU -> conv3x3 (+ emb row vector)
        x3 = x.view(N, S, C)
        gam, bet = 1 + 0.2 * rn(C), 0.3 * rn(C)
        wc, bc = rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)
        rv = rn(N, C)
        pc, pc8 = ops.pack_conv3x3(wc, bc), ops.pack_conv3x3_fp8(wc, bc, "cuda")
        c16 = lambda: ops.conv3x3(ops.groupnorm(x3, gam, bet, 1e-5, True), pc, N, H, W, rowvec=rv)[0]  # noqa: E731

        def c8():
            y8, sc = ops.groupnorm_fp8(x3, gam, bet, 1e-5, True)
            return ops.conv3x3_fp8(y8, sc, pc8, N, H, W, rowvec=rv)
        nimg = min(N, 2)
        xr = x3[:nimg].float().view(nimg, H, W, C).permute(0, 3, 1, 2)
        refc = F.conv2d(F.silu(F.group_norm(xr, 32, gam, bet, 1e-5)), wc, bc, padding=1).permute(0, 2, 3, 1).reshape(nimg, S, C) + rv[:nimg, None, :]
        r16, r8 = c16(), c8()
        print(f"C {C:4d} GroupNorm+SiLU+conv3x3 bf16 {best(c16):.4f} ms (err {rel(r16[:nimg], refc):.2e})   fp8 {best(c8):.4f} ms (err {rel(r8[:nimg], refc):.2e})", flush=True)
        del x, res
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
