"""Run ONE hot kernel at a BASELINE shape a few times (for rocprofv3 --pmc passes).
usage: python tools/one_kernel.py {attn|conv|convemb|convgn|geglu|ffout|linear|ffused} [level 0|1|2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402

BF16 = torch.bfloat16
kind = sys.argv[1]
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
C, H, W, heads = [(320, 72, 128, 5), (640, 36, 64, 10), (1280, 18, 32, 20)][lvl]
N = 50
S = H * W
M = N * S
x = torch.randn(M, C, device="cuda").to(BF16)
if kind == "attn":  # the product form since round 3: q | k | v column blocks of ONE GEMM, V read row-major (ds_read_b64_tr_b16)
    pqkv = ops.pack_linear_cat([torch.randn(C, C) * C ** -0.5 for _ in range(3)])
    qkv = ops.linear(x, pqkv)
    pre = os.environ.get("ATTN_LOG2", "1") == "1"  # the product form: pre-scaled query, zero-base softmax in the 512-row kernel
    if pre:
        qkv[:, :C] *= 0.18033688
    fn = lambda: ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], N, heads, S, v_rows=True, q_log2=pre)  # noqa: E731
elif kind == "conv":
    pc = ops.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5, torch.randn(C))
    x3 = x.view(N, S, C)
    fn = lambda: ops.conv3x3(x3, pc, N, H, W)  # noqa: E731
elif kind in ("convemb", "convgn"):  # round 5: conv3x3 + per-image row vector, without / with the epilogue that also emits the GroupNorm statistics of the output
    pc = ops.pack_conv3x3(torch.randn(C, C, 3, 3) * (9 * C) ** -0.5, torch.randn(C))
    x3 = x.view(N, S, C)
    rv = torch.randn(N, C, device="cuda")
    fn = (lambda: ops.conv3x3(x3, pc, N, H, W, rowvec=rv, gn=ops.GnPartials())) if kind == "convgn" else (lambda: ops.conv3x3(x3, pc, N, H, W, rowvec=rv))  # noqa: E731
elif kind == "geglu":
    pg = ops.pack_geglu(torch.randn(8 * C, C) * C ** -0.5, torch.randn(8 * C))
    fn = lambda: ops.linear(x, pg)  # noqa: E731
elif kind == "ffout":
    h = torch.randn(M, 4 * C, device="cuda").to(BF16)
    po = ops.pack_linear(torch.randn(C, 4 * C) * (4 * C) ** -0.5, torch.randn(C))
    fn = lambda: ops.linear(h, po, res1=x)  # noqa: E731
elif kind == "ffused":  # level-0 FeedForward as ONE kernel (round 4): GEGLU in-projection -> gelu -> out-projection, + residual
    pg = ops.pack_geglu(torch.randn(8 * C, C) * C ** -0.5, torch.randn(8 * C))
    po = ops.pack_ff_out(torch.randn(C, 4 * C) * (4 * C) ** -0.5, torch.randn(C))
    fn = lambda: ops.ff_fused(x, pg, po, res1=x)  # noqa: E731
else:  # K = C projection with a residual (attention out / proj_out): gemm_stream.hip at level 0 since round 4, 128x160 two-per-CU tiles at level 1
    pw = ops.pack_linear(torch.randn(C, C) * C ** -0.5, torch.randn(C))
    res = torch.randn(M, C, device="cuda").to(BF16)
    fn = lambda: ops.linear(x, pw, res1=res)  # noqa: E731
for _ in range(3):
    fn()
torch.cuda.synchronize()
