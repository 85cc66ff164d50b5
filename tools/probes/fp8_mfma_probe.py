import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "fp8_probe.hsaco")
# load via hipModule (kernels are extern "C" __global__): use torch's current context through ctypes on libamdhip64
hip = ctypes.CDLL("libamdhip64.so")
mod = ctypes.c_void_p()
assert hip.hipModuleLoad(ctypes.byref(mod), so.encode()) == 0
def fn(name):
    f = ctypes.c_void_p()
    assert hip.hipModuleGetFunction(ctypes.byref(f), mod, name.encode()) == 0
    return f
def launch(f, grid, block, *args):
    argv = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    rc = hip.hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, 0, None, argv, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
torch.cuda.init(); torch.zeros(1, device="cuda")
g = torch.Generator().manual_seed(0)
A = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
B = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
ref = A.float() @ B.float().t()
Ad, Bd = A.view(torch.uint8).cuda(), B.view(torch.uint8).cuda()
D = torch.zeros(32, 32, device="cuda")
launch(fn("probe_layout"), 1, 64, ctypes.c_void_p(Ad.data_ptr()), ctypes.c_void_p(Bd.data_ptr()), ctypes.c_void_p(D.data_ptr()))
print("layout hypothesis max err:", (D.cpu() - ref).abs().max().item(), " ref rms", ref.pow(2).mean().sqrt().item())
# rate
iters = 20000
out = torch.zeros(1024 * 256, device="cuda")
f = fn("probe_rate")
launch(f, 1024, 256, ctypes.c_void_p(out.data_ptr()), ctypes.c_int(10))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
argv = (ctypes.c_void_p * 2)(ctypes.cast(ctypes.pointer(ctypes.c_void_p(out.data_ptr())), ctypes.c_void_p), ctypes.cast(ctypes.pointer(ctypes.c_int(iters)), ctypes.c_void_p))
hip.hipModuleLaunchKernel(f, 1024, 1, 1, 256, 1, 1, 0, None, argv, None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
flops = 1024 * 4 * iters * 4 * 32 * 32 * 64 * 2
print(f"raw fp8 32x32x64 rate: {flops / ms / 1e9:.0f} TFLOP/s ({ms:.2f} ms)")
# cvt
x = torch.tensor([0.0, 1.0, -1.5, 447.0, 448.0, 500.0, 1e6, -1e6, 0.001953125, 0.0009765625, 3.3, -0.07], device="cuda")
q = torch.zeros(12, dtype=torch.uint8, device="cuda")
launch(fn("probe_cvt"), 1, 64, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(q.data_ptr()), ctypes.c_int(12))
print("cvt:", q.cpu().view(torch.float8_e4m3fn).float().tolist())
print("torch:", x.cpu().to(torch.float8_e4m3fn).float().tolist())
