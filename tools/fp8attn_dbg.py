import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops
BF16 = torch.bfloat16
def mxdeq(q, s):
    return q.view(torch.float8_e4m3fn).float() * torch.exp2(s.float() - 127).repeat_interleave(32, 1)
for n_img, heads, S in [(3, 5, 144), (2, 10, 576), (2, 5, 2304), (1, 5, 9216), (2, 5, 4104), (3, 20, 200)]:
    M, C = n_img * S, heads * 64
    g = torch.Generator().manual_seed(S + heads)
    qk8 = (torch.randn(M, 2 * C, generator=g) * 120).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    sc = torch.randint(124, 131, (M, 2 * C // 32), generator=g, dtype=torch.uint8).cuda()
    v = torch.randn(M, C, generator=g).to(BF16).cuda()
    nb = C // 32
    d = mxdeq(qk8, sc)
    q = d[:, :C].view(n_img, S, heads, 64).transpose(1, 2)
    k = d[:, C:].view(n_img, S, heads, 64).transpose(1, 2)
    vv = v.float().view(n_img, S, heads, 64).transpose(1, 2)
    std = (q[0, 0] @ k[0, 0].T).std().item()
    for mult in (1.0, 4.0, 365.0):
        scale = mult / std
        got = ops.attn_spatial_fp8qk(qk8[:, :C], qk8[:, C:], sc[:, :nb], sc[:, nb:], v, n_img, heads, S, scale=scale).float()
        lg = (q.double() @ k.double().transpose(-1, -2)) * scale
        ref = (torch.softmax(lg, -1) @ vv.double()).transpose(1, 2).reshape(M, C).float()
        e = (got - ref).abs()
        rms = ref.pow(2).mean().sqrt()
        bad = (e > 2e-2 * rms + 1.6e-2 * ref.abs())
        rel = ((got - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()
        # per-head error
        ph = (got - ref).view(n_img, S, heads, 64).pow(2).sum((1, 3)).sqrt() / ref.view(n_img, S, heads, 64).pow(2).sum((1, 3)).sqrt()
        print(f"S={S} heads={heads} mult={mult}: rel {rel:.3e} max {e.max().item():.3e} rms {rms.item():.3f} bad {bad.sum().item()} per-head max {ph.max().item():.3e}", flush=True)
        if bad.any():
            idx = bad.nonzero()[:5].tolist()
            print("   first bad (row, col):", idx)
