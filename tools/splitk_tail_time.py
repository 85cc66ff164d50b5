"""Times the K-deep level-0 GEMMs whose launch is R full rounds + a few straggler tiles (1800 tiles of 256x320 = 7.03 rounds) with the in-kernel
split-K fix-up of the stragglers on / off (VISTA_SPLITK_FIXUP is read once per process: run twice)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops
BF16 = torch.bfloat16
rn = lambda *s: torch.randn(*s, device="cuda")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return round(best, 4)
N = 50
out = {"fixup": os.environ.get("VISTA_SPLITK_FIXUP", "1")}
for C, H, W in ((320, 72, 128), (640, 36, 64), (1280, 18, 32)):
    M = N * H * W
    x = rn(M, C).to(BF16); res = rn(M, C).to(BF16); h4 = rn(M, 4 * C).to(BF16); x3 = x.view(N, H * W, C)
    out[f"C{C}_ff_out"] = timeit(lambda pw=ops.pack_linear(rn(C, 4 * C) * (4 * C) ** -0.5, rn(C)): ops.linear(h4, pw, res1=res, emit_stats=True))
    out[f"C{C}_conv3x3"] = timeit(lambda pw=ops.pack_conv3x3(rn(C, C, 3, 3) * (9 * C) ** -0.5, rn(C)): ops.conv3x3(x3, pw, N, H, W))
    out[f"C{C}_conv_t3"] = timeit(lambda pw=ops.pack_conv_t3(rn(C, C, 3, 1, 1) * (3 * C) ** -0.5, rn(C)): ops.conv_t3(x3, pw, 25, H * W))
    del x, res, h4, x3
print(json.dumps(out))
