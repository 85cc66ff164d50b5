"""Per-kernel micro-benchmarks at the BASELINE shapes (N=50 images, 72x128 latent). Prints one JSON line per kernel.
Usage (GPU box): python tools/kbench.py [--quick]"""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vista_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, ms, flop=None, bytes_=None):
    r = {"kernel": name, "ms": round(ms, 4)}
    if flop:
        r["TFLOPs"] = round(flop / ms / 1e9, 1)
    if bytes_:
        r["GBs"] = round(bytes_ / ms / 1e6, 1)
    print(json.dumps(r), flush=True)


def main():
    quick = "--quick" in sys.argv
    dev = "cuda"
    N = 50
    levels = [(320, 72, 128, 5), (640, 36, 64, 10), (1280, 18, 32, 20)]
    if quick:
        levels = levels[:1]
    for C, H, W, heads in levels:
        S = H * W
        M = N * S
        x = torch.randn(M, C, device=dev).to(BF16)
        # --- linear C->C
        pw = ops.pack_linear(torch.randn(C, C) * C ** -0.5, torch.randn(C))
        report(f"linear {M}x{C}x{C}", timeit(lambda: ops.linear(x, pw)), 2.0 * M * C * C, 4.0 * M * C)
        # --- fused qk
        pqk = ops.pack_linear_cat([torch.randn(C, C) * C ** -0.5, torch.randn(C, C) * C ** -0.5])
        report(f"linear_qk {M}x{2*C}x{C}", timeit(lambda: ops.linear(x, pqk)), 4.0 * M * C * C, 6.0 * M * C)
        pv = ops.pack_linear(torch.randn(C, C) * C ** -0.5, None)
        report(f"linear_vt {M}x{C}x{C}", timeit(lambda: ops.linear_vt(x, pv, S)), 2.0 * M * C * C, 4.0 * M * C)
        # --- GEGLU + out
        pg = ops.pack_geglu(torch.randn(8 * C, C) * C ** -0.5, torch.randn(8 * C))
        hbuf = ops.linear(x, pg)
        report(f"geglu {M}x{8*C}x{C}", timeit(lambda: ops.linear(x, pg)), 2.0 * M * 8 * C * C, 2.0 * M * C + 2.0 * M * 4 * C)
        po = ops.pack_linear(torch.randn(C, 4 * C) * (4 * C) ** -0.5, torch.randn(C))
        report(f"ff_out {M}x{C}x{4*C}", timeit(lambda: ops.linear(hbuf, po, res1=x)), 2.0 * M * 4 * C * C, 2.0 * M * 4 * C + 4.0 * M * C)
        del hbuf
        # --- conv3x3 C->C
        x3 = x.view(N, S, C)
        pc = ops.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5, torch.randn(C))
        report(f"conv3x3 {N}x{H}x{W} {C}->{C}", timeit(lambda: ops.conv3x3(x3, pc, N, H, W)), 2.0 * M * 9 * C * C, 4.0 * M * C)
        # --- temporal conv
        pt = ops.pack_conv_t3(torch.randn(C, C, 3, 1, 1) * (3 * C) ** -0.5, torch.randn(C))
        report(f"conv_t3 {C}", timeit(lambda: ops.conv_t3(x3, pt, 25, S)), 2.0 * M * 3 * C * C, 4.0 * M * C)
        # --- attention
        qk = ops.linear(x, pqk)
        vt = ops.linear_vt(x, pv, S)
        it = 3 if S > 4000 else 5
        report(f"attn_spatial B*h={N*heads} S={S}", timeit(lambda: ops.attn_spatial(qk[:, :C], qk[:, C:], vt, N, heads, S), iters=it, warmup=1),
               4.0 * N * heads * S * S * 64, 8.0 * M * C)
        qkv = torch.randn(M, 3 * C, device=dev).to(BF16)
        report(f"attn_temporal S={S} heads={heads}", timeit(lambda: ops.attn_temporal(qkv, 2, 25, S, heads)), 4.0 * 2 * S * heads * 25 * 25 * 64, 8.0 * M * C)
        del qkv, qk, vt
        # --- norms
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        report(f"groupnorm+silu C={C}", timeit(lambda: ops.groupnorm(x3, g, b, 1e-5, True)), None, 6.0 * M * C)
        report(f"groupnorm5d+silu C={C}", timeit(lambda: ops.groupnorm(x3, g, b, 1e-5, True, frames_per_group=25)), None, 6.0 * M * C)
        report(f"layernorm C={C}", timeit(lambda: ops.layernorm(x, g, b)), None, 4.0 * M * C)
        del x, x3
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
