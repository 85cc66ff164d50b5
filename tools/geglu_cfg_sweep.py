import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vista_amd import ops
BF16 = torch.bfloat16
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n)
    return best
for C, M in ((320, 460800), (640, 115200), (1280, 28800)):
    x = torch.randn(M, C, device="cuda").to(BF16)
    pg = ops.pack_geglu(torch.randn(8 * C, C) * C ** -0.5, torch.randn(8 * C))
    r = {}
    for cfg in (0, 1, 2, 3):
        ops.TILE_CFG = cfg
        r[cfg] = timeit(lambda: ops.linear(x, pg))
    ops.TILE_CFG = 0
    print(C, {k: round(v, 4) for k, v in r.items()})
