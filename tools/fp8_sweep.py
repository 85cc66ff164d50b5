"""fp8 vs bf16 dense GEMM on the BASELINE shapes (TFLOP/s, best of 3x10). The fp8 number EXCLUDES the quantisation pass;
the second fp8 column includes one vk_quantize_rows_fp8 of the activation per GEMM (the un-fused worst case)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import ops

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best

for kind, M, N, K in (("linear", 460800, 320, 320), ("linear", 460800, 960, 320), ("ff_out", 460800, 320, 1280), ("geglu", 460800, 2560, 320),
                      ("linear", 115200, 640, 640), ("ff_out", 115200, 640, 2560), ("geglu", 115200, 5120, 640),
                      ("linear", 28800, 1280, 1280), ("ff_out", 28800, 1280, 5120), ("geglu", 28800, 10240, 1280)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w, b = torch.randn(N, K) * K ** -0.5, torch.randn(N)
    if kind == "geglu":
        pb, p8 = ops.pack_geglu(w, b), ops.pack_geglu_fp8(w, b)
    else:
        pb, p8 = ops.pack_linear(w, b), ops.pack_linear_fp8(w, b)
    xq, xs = ops.quantize_rows_fp8(x)
    flop = 2.0 * M * N * K
    t_b = timeit(lambda: ops.linear(x, pb))
    t_8 = timeit(lambda: ops.linear_fp8(xq, xs, p8))
    t_q = timeit(lambda: ops.quantize_rows_fp8(x))
    print(json.dumps({"kind": kind, "M": M, "N": N, "K": K, "bf16_TF": round(flop / t_b / 1e9), "fp8_TF": round(flop / t_8 / 1e9),
                      "fp8_incl_quant_TF": round(flop / (t_8 + t_q) / 1e9), "quant_us": round(t_q * 1e3, 1), "bf16_us": round(t_b * 1e3, 1),
                      "fp8_us": round(t_8 * 1e3, 1)}), flush=True)
    del x, xq, pb, p8
    torch.cuda.empty_cache()
