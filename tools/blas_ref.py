"""How fast does the vendor library (torch.matmul -> hipBLASLt/rocBLAS) run the BASELINE GEMM shapes on this box?
Measurement only -- a yardstick for vk_gemm_bf16, never part of the product path."""
import json
import torch

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best

for M, N, K in ((460800, 320, 320), (460800, 640, 320), (460800, 320, 1280), (460800, 2560, 320), (115200, 640, 640), (115200, 640, 2560),
                (115200, 5120, 640), (28800, 1280, 1280), (28800, 1280, 5120), (28800, 10240, 1280), (460800, 320, 2880), (8192, 8192, 8192)):
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: torch.matmul(x, w.t()))
    print(json.dumps({"M": M, "N": N, "K": K, "blas_TFLOPs": round(2.0 * M * N * K / ms / 1e9)}), flush=True)
