"""Diagnostic for the pipelined attention kernel: which KEYS carry wrong weight? V rows are one-hot (a) of the key's position inside its 64-key
tile and (b) of its tile index group, so the output IS the (normalised) probability mass per position / per tile group; compared with fp32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
torch.manual_seed(0)
q = (torch.randn(S, 64, device="cuda") * 0.18033688 * 1.0).to(torch.bfloat16)
k = torch.randn(S, 64, device="cuda").to(torch.bfloat16)
keys = torch.arange(S, device="cuda")
for name, idx in (("position in tile (key % 64)", keys % 64), ("tile group (key // (S/64))", keys // (S // 64))):
    v = torch.nn.functional.one_hot(idx, 64).to(torch.bfloat16)
    buf = torch.cat([q, k, v], 1).contiguous()
    o = ops.attn_spatial(buf[:, :64], buf[:, 64:128], buf[:, 128:], 1, 1, S, v_rows=True, q_log2=True).float()
    p = torch.softmax((q.float() @ k.float().t()) * 0.6931471805599453, -1)
    ref = p @ v.float()
    err = (o - ref)
    rel = err.pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()
    print(f"== {name}: rel-L2 {rel:.3e}; row sums of o: min {o.sum(1).min():.4f} max {o.sum(1).max():.4f}")
    col = (err.abs().mean(0) / ref.abs().mean(0))
    print("   mean |err| / mean |ref| per column:", " ".join(f"{c:.3f}" for c in col.tolist()))
    rowerr = err.abs().sum(1) / ref.abs().sum(1)
    print("   per query-row block of 32 (mean rel err):", " ".join(f"{rowerr[i:i + 32].mean():.3f}" for i in range(0, min(S, 512), 32)))
