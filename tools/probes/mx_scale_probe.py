import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "mx_scale_probe.hsaco")
hip = ctypes.CDLL("libamdhip64.so")
mod = ctypes.c_void_p()
assert hip.hipModuleLoad(ctypes.byref(mod), so.encode()) == 0
def fn(name):
    f = ctypes.c_void_p()
    assert hip.hipModuleGetFunction(ctypes.byref(f), mod, name.encode()) == 0
    return f
def launch(f, *args):
    argv = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
    assert hip.hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, None, argv, None) == 0
    torch.cuda.synchronize()
torch.cuda.init(); torch.zeros(1, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
ones = torch.full((32, 64), 0x38, dtype=torch.uint8, device="cuda")  # fp8 e4m3 1.0
def run(SA, SB, name="probe_scale0", A=ones, B=ones):
    D = torch.zeros(32, 32, device="cuda")
    SAd, SBd = SA.cuda(), SB.cuda()   # keep both alive: temporaries would be freed and the allocator hands the SAME block to both
    launch(fn(name), p(A), p(B), p(SAd), p(SBd), p(D))
    return D.cpu()
unit = torch.full((64,), 127, dtype=torch.uint8)
print("all scales 1.0: D unique", run(unit, unit).unique().tolist())
for L in (0, 1, 5, 31, 32, 33, 37, 63):
    s = unit.clone(); s[L] = 128
    D = run(s, unit)
    rows = [(i, sorted(set(D[i].tolist()))) for i in range(32) if (D[i] != 64).any()]
    cols = [j for j in range(32) if (D[:, j] != 64).any()]
    print(f"A-scale x2 on lane {L}: rows {rows[:4]}{'...' if len(rows) > 4 else ''} | n_rows {len(rows)} n_cols {len(cols)}")
for L in (0, 5, 32, 37):
    s = unit.clone(); s[L] = 128
    D = run(unit, s)
    cols = [(j, sorted(set(D[:, j].tolist()))) for j in range(32) if (D[:, j] != 64).any()]
    rows = [i for i in range(32) if (D[i] != 64).any()]
    print(f"B-scale x2 on lane {L}: cols {cols[:4]}{'...' if len(cols) > 4 else ''} | n_cols {len(cols)} n_rows {len(rows)}")
A0 = ones.clone(); A0[:, 32:] = 0
for L in (3, 35):
    s = unit.clone(); s[L] = 128
    D = run(s, unit, A=A0)
    print(f"A nonzero only in K-block 0, A-scale x2 on lane {L}: D unique {D.unique().tolist()}")
s = unit.clone(); s[7] = 129
print("opsel 2, A-scale x4 on lane 7: rows", [(i, Di.unique().tolist()) for i, Di in enumerate(run(s, unit, name='probe_scale2')) if (Di != 64).any()][:3])

print("---- hand-placed instruction (no register overlap, wait states)")
for L in (0, 5, 37):
    s = unit.clone(); s[L] = 128
    D = run(s, unit, name="probe_scale_asm")
    rows = [(i, sorted(set(D[i].tolist()))) for i in range(32) if (D[i] != 64).any()]
    print(f"asm A-scale x2 on lane {L}: rows {rows[:4]} n_rows {len(rows)}")
    D = run(unit, s, name="probe_scale_asm")
    cols = [(j, sorted(set(D[:, j].tolist()))) for j in range(32) if (D[:, j] != 64).any()]
    rows = [i for i in range(32) if (D[i] != 64).any()]
    print(f"asm B-scale x2 on lane {L}: cols {cols[:4]} n_cols {len(cols)} n_rows {len(rows)}")

print("---- one packed scale VGPR (A scale byte 0 / op_sel 0, B scale byte 1 / op_sel 1)")
for L in (0, 5, 37):
    s = unit.clone(); s[L] = 128
    D = run(s, unit, name="probe_scale_packed")
    rows = [(i, sorted(set(D[i].tolist()))) for i in range(32) if (D[i] != 64).any()]
    cols = [j for j in range(32) if (D[:, j] != 64).any()]
    print(f"packed: A-scale x2 on lane {L}: rows {rows[:3]} n_rows {len(rows)} n_cols {len(cols)}")
    D = run(unit, s, name="probe_scale_packed")
    cols = [(j, sorted(set(D[:, j].tolist()))) for j in range(32) if (D[:, j] != 64).any()]
    rows = [i for i in range(32) if (D[i] != 64).any()]
    print(f"packed: B-scale x2 on lane {L}: cols {cols[:3]} n_cols {len(cols)} n_rows {len(rows)}")
print("---- global checks")
allx2 = torch.full((64,), 128, dtype=torch.uint8)
print("SA all x2, SB unit  (asm):", run(allx2, unit, name="probe_scale_asm").unique().tolist())
print("SA unit, SB all x2  (asm):", run(unit, allx2, name="probe_scale_asm").unique().tolist())
print("SA all x2, SB all x2 (asm):", run(allx2, allx2, name="probe_scale_asm").unique().tolist())
half = unit.clone(); half[:32] = 128
print("SA lanes 0-31 x2 (asm):", run(half, unit, name="probe_scale_asm").unique().tolist(), " SB lanes 0-31 x2:", run(unit, half, name="probe_scale_asm").unique().tolist())
B0 = ones.clone(); B0[:, 32:] = 0     # B nonzero only in K-block 0
s = unit.clone(); s[32] = 128          # lane 32 = (row 0, K-block 1)
print("B only in K-block 0, SB x2 on lane 32:", run(unit, s, name="probe_scale_asm", B=B0).unique().tolist(), "(A still has block 1 -> if the scale reaches A(row0, blk1) nothing changes because B blk1 = 0)")
s = unit.clone(); s[0] = 128
D = run(unit, s, name="probe_scale_asm", B=B0)
print("B only in K-block 0, SB x2 on lane 0: row0", D[0].unique().tolist(), "col0", D[:, 0].unique().tolist(), "rest", D[1:, 1:].unique().tolist())

print("---- first scale operand, all four bytes = x2, every op_sel:")
for n in ("probe_a0", "probe_a1", "probe_a2", "probe_a3"):
    print(n, run(allx2, unit, name=n).unique().tolist())
s5 = unit.clone(); s5[5] = 128
D = run(s5, unit, name="probe_swapped")
print("swapped slots, SA x2 on lane 5: n_rows", sum(bool((D[i] != 64).any()) for i in range(32)), "n_cols", sum(bool((D[:, j] != 64).any()) for j in range(32)), D.unique().tolist())

print("---- random data + random scales against the two layout hypotheses")
g = torch.Generator().manual_seed(0)
A = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
B = (torch.randn(32, 64, generator=g) * 2).to(torch.float8_e4m3fn)
SA = torch.randint(120, 134, (64,), generator=g, dtype=torch.uint8)
SB = torch.randint(120, 134, (64,), generator=g, dtype=torch.uint8)
D = run(SA, SB, name="probe_scale_asm", A=A.view(torch.uint8).cuda(), B=B.view(torch.uint8).cuda())
def ref(layout):
    fa, fb = A.float().clone(), B.float().clone()
    for l in range(64):
        r, h = l & 31, l >> 5
        ks = list(range(32 * h, 32 * h + 32)) if layout == "H1" else list(range(16 * h, 16 * h + 16)) + list(range(32 + 16 * h, 48 + 16 * h))
        # bytes the lane LOADS are [32h, 32h+32) of the row; under H2 its first 16 bytes act as K-block 0 and the last 16 as K-block 1
        if layout == "H1":
            fa[r, 32 * h:32 * h + 32] *= 2.0 ** (float(SA[l]) - 127); fb[r, 32 * h:32 * h + 32] *= 2.0 ** (float(SB[l]) - 127)
        else:  # H2: scale of lane r+32b applies to the first (b=0) / second (b=1) 16 bytes of BOTH lanes r and r+32
            for hh in (0, 1):
                fa[r, 32 * hh + 16 * h:32 * hh + 16 * h + 16] *= 2.0 ** (float(SA[r + 32 * h]) - 127)
                fb[r, 32 * hh + 16 * h:32 * hh + 16 * h + 16] *= 2.0 ** (float(SB[r + 32 * h]) - 127)
    return fa @ fb.t()
for lay in ("H1", "H2"):
    R = ref(lay)
    print(lay, "max err", (D - R).abs().max().item(), "ref rms", R.pow(2).mean().sqrt().item())
