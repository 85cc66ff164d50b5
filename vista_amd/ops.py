"""Torch-tensor front end of the C-ABI kernels (include/vista_hip.h).

PyTorch is used for device memory (caching allocator), streams and views only; every arithmetic op below is a
hand-written gfx950 kernel in libvista_hip.so. Activations are token-major bf16: (n_img, S=H*W, C), C contiguous.
"""
import ctypes as C
import math
import os
import threading

import torch

from . import _lib
from ._lib import VkFp8Args, VkGemmDesc, check

AMODE_DENSE, AMODE_CONV3X3, AMODE_TEMPORAL3, AMODE_CONV3D = 0, 1, 2, 3
EPI_LINEAR, EPI_GEGLU, EPI_TRANS = 0, 1, 2
# The 16-bit storage type of activations and packed weights: bf16 unless the process runs the fp16 build (VISTA_ACT_DTYPE=fp16, _lib.ACT_DTYPE).
# `BF16` is the historical name used throughout this file for "the storage type"; ACT is the same object under an honest name.
ACT = torch.float16 if _lib.ACT_DTYPE == "fp16" else torch.bfloat16
BF16 = ACT
F32 = torch.float32


class storage:
    """Context manager: run the enclosed ops with another 16-bit storage type -- `with ops.storage(torch.bfloat16):` switches ops.ACT / ops.BF16 (every
    function of this file reads them at call time) and the library _lib.load() hands out, and restores both on exit. A no-op when the type is already
    current (any bf16 process). Used by the first-stage VAE and the conditioner, which store bf16 in every process (the reference runs them without
    autocast; fp16's exponent range is not safe there), so that an fp16 process (VISTA_ACT_DTYPE=fp16) can run the whole pipeline: denoiser in fp16,
    decode / encode / CLIP tower in bf16. Weight packs are built under the context that uses them (a module is only ever run under one type).
    Process-global, not thread-local: do not interleave two storage types from concurrent host threads."""

    def __init__(self, dtype):
        if dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("ops.storage: torch.bfloat16 or torch.float16")
        self.dtype, self.name = dtype, ("fp16" if dtype is torch.float16 else "bf16")

    def __enter__(self):
        global ACT, BF16
        self.prev = (ACT, _lib.CURRENT)
        ACT = BF16 = self.dtype
        _lib.CURRENT = self.name
        return self

    def __exit__(self, *exc):
        global ACT, BF16
        ACT = BF16 = self.prev[0]
        _lib.CURRENT = self.prev[1]
        return False


def bf16_storage(fn):
    """Decorator for forwards that always store bf16 (first-stage VAE, conditioner): ops.storage(torch.bfloat16) around the call."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        if ACT is torch.bfloat16:
            return fn(*a, **k)
        with storage(torch.bfloat16):
            return fn(*a, **k)
    return wrapped


def _stream():
    # raw hipStream_t of torch's current stream on the current device; the C-level getters cost ~1 us against ~9 us for
    # torch.cuda.current_stream() (1500 launches per step go through here)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _need(t, dtype, name):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise _lib.VistaHipError(f"{name}: tensor is on {t.device}; vista_amd kernels run on the MI355X only (no CPU fallback)")


def _rows2d(t, name):
    """(…, C) tensor whose rows are uniformly strided -> (rows, ld)."""
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: last dim must be contiguous")
    t2 = t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t
    if t2.dim() != 2:
        raise ValueError(f"{name}: pass a 2-D strided view")
    return t2, t2.stride(0)


def ceil_to(x, m):
    return (x + m - 1) // m * m


# ---------------------------------------------------------------------------------------------- weight packing
class PackedWeight:
    """bf16 [ceil256(N)][K] weight (K contiguous) + f32 bias, laid out for vk_gemm_bf16. `colsum` (f32 [Np]) marks a weight with a
    LayerNorm folded in (pack_*_ln): wt = gamma (.) W, bias = W beta + b, colsum = row sums of the bf16-rounded wt."""

    __slots__ = ("wt", "bias", "N", "K", "geglu", "colsum", "ln_eps", "ffout")

    def __init__(self, wt, bias, N, K, geglu=False, colsum=None, ln_eps=0.0, ffout=False):
        self.wt, self.bias, self.N, self.K, self.geglu, self.colsum, self.ln_eps = wt, bias, N, K, geglu, colsum, ln_eps
        self.ffout = ffout  # laid out for vk_ff_fused_bf16's out-projection (pack_ff_out): not a vk_gemm_bf16 operand


def _finish_pack(w2d, bias, device, geglu=False, ln=None):
    """ln = (gamma, beta, eps): fold `LayerNorm(x) @ W^T + b` into the GEMM (include/vista_hip.h, VkGemmDesc.ln_*):
    LN(x) W^T + b = rstd * (x W'^T - mean * s) + t with W' = gamma (.) W, s_n = sum_k bf16(W')[n][k], t = W beta + b."""
    N, K = w2d.shape
    Kp = ceil_to(K, 64)
    Np = max(ceil_to(N, 256), ceil_to(N, 320))  # readable by every block-tile variant (128/256/320-wide) without bounds checks
    colsum, eps = None, 0.0
    if ln is not None:
        gamma, beta, eps = ln
        if Kp != K:
            raise ValueError("LayerNorm fold needs K % 64 == 0")
        gamma, beta = gamma.detach().float().to(w2d.device), beta.detach().float().to(w2d.device)
        t = w2d @ beta
        bias = t if bias is None else bias.to(w2d.device) + t
        w2d = w2d * gamma[None, :]
    wt = torch.zeros((Np, Kp), dtype=BF16, device=device)
    wt[:N, :K] = w2d.to(device=device, dtype=BF16)
    if ln is not None:
        colsum = wt.float().sum(dim=1).contiguous()  # of the ROUNDED weights: the mean term then cancels exactly in the epilogue
    b = None
    if bias is not None:
        b = torch.zeros((Np,), dtype=F32, device=device)
        b[:N] = bias.to(device=device, dtype=F32)
    return PackedWeight(wt, b, ceil_to(N, 4), Kp, geglu, colsum, float(eps))  # N % 4 == 0 for the 4-column epilogue quads; the extra rows/bias are zero


def _ln_tuple(norm):
    """(gamma, beta, eps) of a LayerNorm parameter container."""
    return (norm.weight, norm.bias, norm.eps)


KPERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


def pack_linear(weight, bias=None, device="cuda", ln=None):
    """nn.Linear.weight [N][K] (or a 1x1 conv weight [N][K][1][1]). ln: LayerNorm container applied to the input (folded)."""
    return _finish_pack(weight.detach().reshape(weight.shape[0], -1).float(), None if bias is None else bias.detach().float(), device,
                        ln=None if ln is None else _ln_tuple(ln))


def ff_out_layout(w):
    """[N][K] -> the out-projection operand of vk_ff_fused_bf16 (include/vista_hip.h): K permuted inside every 16-group to KPERM16 (the order in
    which a lane of the in-projection's 32x32 MFMA accumulator holds the hidden units), then chunk-major [K / 32][N][32] so that a 32-wide
    hidden chunk of all N rows is one contiguous 20 KB block (whole cache lines per LDS-DMA piece). Pure index shuffle (CPU-testable)."""
    N, K = w.shape
    if K % 32:
        raise ValueError("ff_out_layout needs K % 32 == 0")
    w = w.reshape(N, K // 16, 16)[:, :, list(KPERM16)].reshape(N, K // 32, 32)
    return w.permute(1, 0, 2).contiguous().reshape(K // 32 * N, 32)


def pack_ff_out(weight, bias=None, device="cuda"):
    """FeedForward.net[2] weight [N][K] for ops.ff_fused (bf16, ff_out_layout; bias f32)."""
    w = weight.detach().reshape(weight.shape[0], -1).float()
    N, K = w.shape
    b = None if bias is None else bias.detach().to(device=device, dtype=F32).contiguous()
    return PackedWeight(ff_out_layout(w).to(device=device, dtype=BF16), b, N, K, ffout=True)


def pack_rows_as_weight(t, N, K):
    """Treat an activation buffer whose first N rows are [N][K] bf16 (K contiguous) as the weight operand of a GEMM
    (q.k^T and P.v of the VAE decoder's AttnBlock). The caller guarantees the rows up to the next tile boundary are readable."""
    _need(t, BF16, "t")
    return PackedWeight(t, None, N, K)


def pack_linear_cat(weights, device="cuda", ln=None):
    """Several bias-free Linear weights sharing the input, concatenated along N (fused q|k or q|k|v projection)."""
    return _finish_pack(torch.cat([w.detach().float() for w in weights], 0), None, device, ln=None if ln is None else _ln_tuple(ln))


def geglu_perm(nout):
    """Row order of a packed GEGLU weight: every 32-row MFMA fragment holds [16 value rows | the 16 gate rows of the same output
    columns], so the GEMM epilogue forms value*gelu(gate) inside a lane for any block / wave tile shape."""
    if nout % 16:
        raise ValueError("GEGLU width must be a multiple of 16")
    idx = torch.arange(nout).reshape(-1, 16)
    return torch.stack([idx, idx + nout], 1).reshape(-1)  # [v-block0, g-block0, v-block1, g-block1, ...]


def pack_geglu(weight, bias, device="cuda", ln=None):
    """GEGLU.proj weight [2*Nout][K]: value rows then gate rows (attention.py:85-92), packed in `geglu_perm` order."""
    w = weight.detach().float()
    b = bias.detach().float()
    perm = geglu_perm(w.shape[0] // 2).to(w.device)
    return _finish_pack(w[perm], b[perm], device, geglu=True, ln=None if ln is None else _ln_tuple(ln))


def _slab_major(w):
    """[Cout][taps][Cin] (Cin % 64 == 0) -> [Cout][Cin/64][taps][64] flattened along K: the implicit-GEMM K-loop visits all taps of one
    64-channel slab on consecutive K-steps (the input rows a tile re-reads then sit in L2; csrc/gemm.hip header)."""
    cout, taps, cin = w.shape
    return w.reshape(cout, taps, cin // 64, 64).permute(0, 2, 1, 3).reshape(cout, taps * cin)


def pack_conv3x3(weight, bias=None, cin_pad=None, device="cuda"):
    """nn.Conv2d weight [Cout][Cin][3][3] -> [Cout][Cin(_pad)/64][ky][kx][64]."""
    w = weight.detach().float().permute(0, 2, 3, 1).contiguous()  # Cout, ky, kx, Cin
    cout, _, _, cin = w.shape
    cp = cin_pad or ceil_to(cin, 64)
    if cp != cin:
        w = torch.nn.functional.pad(w, (0, cp - cin))
    return _finish_pack(_slab_major(w.reshape(cout, 9, cp)), None if bias is None else bias.detach().float(), device)


def pack_conv_t3(weight, bias=None, device="cuda", cin_pad=None):
    """nn.Conv3d weight [Cout][Cin][3][1][1] -> [Cout][Cin(_pad)/64][kt][64]."""
    w = weight.detach().float()[:, :, :, 0, 0].permute(0, 2, 1).contiguous()
    cout, _, cin = w.shape
    if cin_pad and cin_pad != cin:
        w = torch.nn.functional.pad(w, (0, cin_pad - cin))
        cin = cin_pad
    if cin % 64:
        raise ValueError("temporal conv needs Cin % 64 == 0")
    return _finish_pack(_slab_major(w), None if bias is None else bias.detach().float(), device)


# ---------------------------------------------------------------------------------------------- GEMM family
TILE_CFG = 0  # 0 = auto; tests force 1..7 to cover every block-tile variant (include/vista_hip.h, VkGemmDesc.tile_cfg)


SPLITK_WS_BYTES = int(os.environ.get("VISTA_SPLITK_WS_MB", "160")) << 20  # fp32 split-K workspace per (device, stream) (0 disables split-K)
class _SplitKTLS(threading.local):
    def __init__(self):
        self.ws = {}


_SPLITK_WS = _SplitKTLS()


class _GraphTLS(threading.local):
    """Per host thread: `depth` > 0 inside graph_workspace() (then _splitk_workspace hands out the graph workspace instead of a per-stream buffer);
    `ws` = {device: the ONE workspace of every hipGraph warm-up / capture of THIS thread on that device}. Thread-local like _SplitKTLS, so a worker
    thread's buffer goes with the thread (a module-level dict keyed by thread ident leaked 160 MiB per thread and let a new thread inherit
    a dead thread's buffer) -- unless a captured graph still needs it: sampling.FusedLoop keeps graph_workspace_tensor() in its graph-cache entry,
    because the graph's split-K launches have the pointer baked in. A graph may be replayed from any thread, one replay at a time."""

    def __init__(self):
        self.depth = 0
        self.slot = 0    # the workspace slot of the innermost graph_workspace()
        self.ws = {}     # device (slot 0) or (device, slot) -> workspace


_GRAPH_TLS = _GraphTLS()


class graph_workspace:
    """Context manager for hipGraph warm-up + capture (sampling.FusedLoop): every split-K GEMM enqueued inside uses ONE workspace per (thread,
    device, slot), allocated here -- OUTSIDE any capture, so it belongs to the ordinary caching allocator and not to a graph's private pool --
    instead of a fresh 160 MB buffer per (thread, stream). Graph replays run on the launch stream one after the other, so they may share it; eager
    work on other streams keeps its per-stream buffers. `slot` > 0: a workspace of its own for a graph that is replayed CONCURRENTLY with the
    slot-0 graphs on another stream (the two guidance halves of a step, FusedLoop(cfg_streams=True))."""

    def __init__(self, slot=0):
        self.slot = slot

    def __enter__(self):
        dev = torch._C._cuda_getDevice()
        key = dev if self.slot == 0 else (dev, self.slot)
        if SPLITK_WS_BYTES and key not in _GRAPH_TLS.ws:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ops.graph_workspace must be entered before the capture starts")
            _GRAPH_TLS.ws[key] = torch.empty(SPLITK_WS_BYTES // 4, dtype=F32, device=f"cuda:{dev}")
        _GRAPH_TLS.depth += 1
        self._outer, _GRAPH_TLS.slot = _GRAPH_TLS.slot, self.slot
        return self

    def __exit__(self, *exc):
        _GRAPH_TLS.depth -= 1
        _GRAPH_TLS.slot = self._outer
        return False


def _graph_ws_key():
    dev, slot = torch._C._cuda_getDevice(), _GRAPH_TLS.slot
    return dev if slot == 0 else (dev, slot)


def graph_workspace_tensor():
    """The calling thread's graph workspace on the current device (inside graph_workspace(slot): that slot's; None before the first
    graph_workspace()): a graph-cache entry holds a reference to it for as long as its graph lives."""
    return _GRAPH_TLS.ws.get(_graph_ws_key())


def _splitk_workspace(stream):
    """One workspace per (device, stream) of each host thread: launches on one stream are ordered, so they may share it; GEMMs in flight on
    different streams must not (include/vista_hip.h, VkGemmDesc.splitk_ws), and thread ranks (tests) share a stream but enqueue
    concurrently. Held in thread-local storage, so a worker thread's buffers are released when the thread exits. Inside
    graph_workspace() (hipGraph warm-up / capture) the thread's single graph workspace of the device is used instead."""
    if _GRAPH_TLS.depth > 0:
        return _GRAPH_TLS.ws[_graph_ws_key()]
    key = (torch._C._cuda_getDevice(), stream.value)
    ws = _SPLITK_WS.ws.get(key)
    if ws is None:
        ws = _SPLITK_WS.ws[key] = torch.empty(SPLITK_WS_BYTES // 4, dtype=F32, device=f"cuda:{key[0]}")
    return ws


class RowStats:
    """Per-row (sum, sum of squares) partials of an activation tensor, f32 [parts][M][2] (include/vista_hip.h, VkGemmDesc.ln_stats):
    written by the producing GEMM's epilogue (linear(..., emit_stats=True)) or by rowstats(); consumed by a GEMM with a folded LayerNorm."""

    __slots__ = ("t", "parts", "M")

    def __init__(self, t, parts, M):
        self.t, self.parts, self.M = t, parts, M


ROW_RANGE = None  # tests: (m_begin, m_end) -> the next GEMM calls compute only those output rows (VkGemmDesc.m_begin / m_end, ABI v5)

# GroupNorm statistics from the producing convolution's epilogue (VkGemmDesc.gnstat_out, ABI v6): 1 = a conv3x3 / conv_t3 call that is handed a
# GnPartials and whose launch can emit them (vk_gemm_gnstat_fit) does, and groupnorm(..., gn=...) then skips its statistics pass; 0 = the
# three-launch GroupNorm everywhere (A/B hook, VISTA_GN_EPI). A launch-affecting switch: part of the hipGraph cache key (sampling.py).
GN_EPI = int(os.environ.get("VISTA_GN_EPI", "1"))


class GnPartials:
    """Stage-1 GroupNorm partial sums of a tensor, written by the epilogue of the convolution that produced it: f32 [n_img * nchunks][64] =
    [32 group sums | 32 group sums of squares] per 64 consecutive rows (include/vista_hip.h, VkGemmDesc.gnstat_out). `t` stays None when the
    producer's launch could not emit them (split-K, a sixteen-wave tile, halo frames, rows per image not a multiple of 64 ...): the consumer
    then runs its own statistics pass -- same result up to the summation order. Consumed by ONE groupnorm call (the fold works in place)."""

    __slots__ = ("t", "nchunks", "rows", "C")

    def __init__(self):
        self.t, self.nchunks, self.rows, self.C = None, 0, 0, 0


def _ask_gn(d, gn, rows_per_image, device):
    """Before the launch: if this GEMM can emit the GroupNorm statistics of its output, give it the buffer."""
    if gn is None or not GN_EPI:
        return
    d.gn_rows = int(rows_per_image)
    d.tile_cfg = TILE_CFG
    if ROW_RANGE is not None:
        return
    if SPLITK_WS_BYTES:   # (the launcher's split-K decision looks at the workspace: ask with what _gemm will set)
        ws = _splitk_workspace(_stream())
        d.splitk_ws, d.splitk_ws_bytes = _p(ws), ws.numel() * 4
    slots = _lib.load().vk_gemm_gnstat_fit(C.byref(d))
    if slots < 0:
        raise _lib.VistaHipError(f"vk_gemm_gnstat_fit failed with code {slots}")
    if slots > 0:
        gn.t = torch.empty(slots * 64, dtype=F32, device=device)
        gn.nchunks, gn.rows, gn.C = int(rows_per_image) // 64, d.M, d.N
        d.gnstat_out = _p(gn.t)



def _gemm(desc, emit_stats=False, device=None):
    lib = _lib.load()
    desc.tile_cfg = TILE_CFG
    if ROW_RANGE is not None:
        desc.m_begin, desc.m_end = ROW_RANGE
    stream = _stream()
    if SPLITK_WS_BYTES:
        ws = _splitk_workspace(stream)
        desc.splitk_ws, desc.splitk_ws_bytes = _p(ws), ws.numel() * 4
    stats = None
    if emit_stats:
        parts = lib.vk_gemm_rowstat_parts(C.byref(desc))
        if parts <= 0:
            raise _lib.VistaHipError(f"vk_gemm_rowstat_parts failed with code {parts}")
        stats = RowStats(torch.empty((parts, desc.M, 2), dtype=F32, device=device), parts, desc.M)
        desc.rowstat_out = _p(stats.t)
    check(lib.vk_gemm_bf16(C.byref(desc), stream), "vk_gemm_bf16")
    return stats


def _fill_ln(d, pw, ln, M):
    """ln: RowStats of the GEMM's input rows; pw must carry a folded LayerNorm (and vice versa)."""
    if (ln is None) != (pw.colsum is None):
        raise ValueError("a weight packed with a folded LayerNorm needs the input's RowStats (ln=...), and only such a weight takes them")
    if ln is not None:
        if ln.M != M:
            raise ValueError(f"RowStats of {ln.M} rows passed to a GEMM over {M} rows")
        d.ln_stats, d.ln_parts, d.ln_colsum, d.ln_eps = _p(ln.t), ln.parts, _p(pw.colsum), pw.ln_eps


def _fill_epilogue(d, pw, out, M, rowvec, rows_per_vec, res1, res2, alpha, beta, rowvec2=None):
    d.Wt = _p(pw.wt)
    d.bias = _p(pw.bias)
    d.M, d.N, d.K = M, pw.N, pw.K
    d.out = _p(out)
    d.ldc = out.stride(0)
    d.out_f32 = 1 if out.dtype == F32 else 0
    d.alpha, d.beta = float(alpha), float(beta)
    if rowvec is not None:
        _need(rowvec, F32, "rowvec")
        d.rowvec, d.ldv, d.rows_per_vec = _p(rowvec), rowvec.stride(0), int(rows_per_vec)
    if res1 is not None:
        _need(res1, BF16, "res1")
        r, ld = _rows2d(res1, "res1")
        d.res1, d.ld_res1 = _p(r), ld
    if res2 is not None:
        _need(res2, BF16, "res2")
        r, ld = _rows2d(res2, "res2")
        d.res2, d.ld_res2 = _p(r), ld
    if rowvec2 is not None:
        _need(rowvec2, F32, "rowvec2")
        if res2 is None:
            raise ValueError("rowvec2 is added to res2 (out = alpha*(...) + beta*(res2 + rowvec2))")
        if rowvec is not None and rowvec.stride(0) != rowvec2.stride(0):
            raise ValueError("rowvec and rowvec2 share one row stride")
        d.rowvec2, d.ldv, d.rows_per_vec = _p(rowvec2), rowvec2.stride(0), int(rows_per_vec)


def linear(x, pw, *, out=None, out_f32=False, rowvec=None, rows_per_vec=0, res1=None, res2=None, alpha=1.0, beta=0.0, rowvec2=None,
           x2=None, ln=None, emit_stats=False, act=None, mx8_cols=0, alt_cols_from=0):
    """out = alpha*(act(X @ W^T + bias + rowvec[row // rows_per_vec]) + res1) + beta*(res2 + rowvec2[row // rows_per_vec]); act: None | "gelu".
    x: (..., K) bf16. x2: second source of a channel concat, X = [x | x2] (never materialised). ln: RowStats of x's rows when pw has
    a LayerNorm folded in (X = LayerNorm(x)). emit_stats: also return the RowStats of the (bf16) output -> (out, stats)."""
    _need(x, BF16, "x")
    x2d, lda = _rows2d(x, "x")
    M = x2d.shape[0]
    k_in = x2d.shape[1]
    d = VkGemmDesc()
    if x2 is not None:
        _need(x2, BF16, "x2")
        b2d, ldb = _rows2d(x2, "x2")
        if b2d.shape[0] != M:
            raise ValueError("linear: x and x2 must have the same rows")
        d.A2, d.lda2, d.k_split = _p(b2d), ldb, k_in
        k_in += b2d.shape[1]
    if k_in != pw.K:
        raise ValueError(f"linear: K mismatch {k_in} vs {pw.K}")
    if pw.ffout:
        raise ValueError("linear: this weight is laid out for ff_fused (pack_ff_out), not for vk_gemm_bf16")
    if pw.geglu:
        nout = pw.N // 2
        if out is None:
            out = torch.empty((M, nout), dtype=BF16, device=x.device)
    elif out is None:
        out = torch.empty((M, pw.N - mx8_cols), dtype=F32 if out_f32 else BF16, device=x.device)
    d.A, d.lda = _p(x2d), lda
    d.amode = AMODE_DENSE
    d.epi = EPI_GEGLU if pw.geglu else EPI_LINEAR
    _fill_epilogue(d, pw, out, M, rowvec, rows_per_vec, res1, res2, alpha, beta, rowvec2)
    _fill_ln(d, pw, ln, M)
    mx = None
    if mx8_cols:
        # BASELINE config 5: the leading mx8_cols output columns leave as MX fp8 (e4m3 bytes + one E8M0 scale per row and 32 columns), the rest as
        # bf16 into `out` (M, N - mx8_cols): the q | k blocks of the fused q|k|v projection for the fp8 QK^T (include/vista_hip.h, mx8_*)
        if pw.geglu or emit_stats or mx8_cols % 320 or pw.N % 320 or out.shape[1] != pw.N - mx8_cols:
            raise ValueError("linear: mx8_cols needs a LINEAR weight, N and mx8_cols multiples of 320, no emit_stats")
        mx = (torch.empty((M, mx8_cols), dtype=torch.uint8, device=x.device), torch.empty((M, mx8_cols // 32), dtype=torch.uint8, device=x.device))
        d.mx8_out, d.mx8_scales, d.mx8_cols, d.ld_mx8, d.ld_mx8s = _p(mx[0]), _p(mx[1]), mx8_cols, mx8_cols, mx8_cols // 32
    if act is not None:
        if act != "gelu" or pw.geglu:
            raise ValueError("linear: act must be None or 'gelu' (exact-erf GELU in the LINEAR epilogue)")
        d.act = 1
    # fp16 build: output columns from alt_cols_from on leave as bf16 (the V block of a fused q|k|v projection: the attention kernels keep P.V in
    # bf16 in both builds, include/vista_hip.h). Ignored by the bf16 build, where every column is bf16 anyway.
    d.alt_cols_from = int(alt_cols_from)
    stats = _gemm(d, emit_stats, x.device)
    if mx is not None:
        return out, mx[0], mx[1]
    return (out, stats) if emit_stats else out


def linear_vt(x, pw, S, out=None, ln=None):
    """V^T projection for spatial attention: x (n_img*S, K) -> out (n_img, N, S) bf16 = (x @ W^T + bias)^T per image (EPI_TRANS)."""
    _need(x, BF16, "x")
    x2, lda = _rows2d(x, "x")
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M // S, pw.N, S), dtype=BF16, device=x.device)
    elif out.dtype != BF16 or not out.is_contiguous() or out.shape != (M // S, pw.N, S):
        raise ValueError("linear_vt: out must be contiguous bf16 (n_img, N, S)")
    d = VkGemmDesc()
    d.A, d.lda = _p(x2), lda
    d.amode, d.epi = AMODE_DENSE, EPI_TRANS
    d.bias = _p(pw.bias)
    d.Wt, d.M, d.N, d.K, d.out, d.ldc, d.S = _p(pw.wt), M, pw.N, pw.K, _p(out), S, S
    d.alpha = 1.0
    _fill_ln(d, pw, ln, M)
    _gemm(d)
    return out


def rowstats(x):
    """RowStats (one slab) of x (..., C) bf16 by a read-only pass: for tensors whose producer is not a GEMM epilogue on this rank."""
    _need(x, BF16, "x")
    x2, ldx = _rows2d(x, "x")
    M, Cc = x2.shape
    st = torch.empty((1, M, 2), dtype=F32, device=x.device)
    check(_lib.load().vk_rowstats_bf16(_p(x2), _p(st), M, Cc, ldx, _stream()), "vk_rowstats_bf16")
    return RowStats(st, 1, M)


FF_FUSED_WIDTH, FF_FUSED_MAX_HIDDEN = 320, 1280
FF_FUSED_DBG_BUF = None
FF_FUSED_DBG = 0  # timing experiments only (tools/ff_fused_probe.py), honoured by a library built with -DFF_TIMING; the product build ignores it


def ff_fused_ok(pw_in, pw_out):
    """Shapes vk_ff_fused_bf16 covers: the level-0 FeedForward (width 320, hidden a multiple of 64 up to 1280)."""
    return (pw_in.geglu and pw_in.K == FF_FUSED_WIDTH and pw_out.N == FF_FUSED_WIDTH and pw_in.N == 2 * pw_out.K
            and pw_out.K % 64 == 0 and 128 <= pw_out.K <= FF_FUSED_MAX_HIDDEN)


def ff_fused(x, pw_in, pw_out, *, out=None, rowvec=None, rows_per_vec=0, res1=None, res2=None, alpha=1.0, beta=0.0, rowvec2=None, ln=None,
             emit_stats=False):
    """linear(linear(x, pw_in, ln=ln), pw_out', ...epilogue) in ONE kernel (vk_ff_fused_bf16): pw_in = pack_geglu(...), pw_out' =
    pack_ff_out(...). The hidden activation never leaves the CU. Returns out, or (out, RowStats) with emit_stats."""
    _need(x, BF16, "x")
    x2d, lda = _rows2d(x, "x")
    M = x2d.shape[0]
    if not ff_fused_ok(pw_in, pw_out) or not pw_out.ffout or x2d.shape[1] != pw_in.K:
        raise ValueError("ff_fused: needs a packed GEGLU weight [2H][320] and a pack_ff_out weight [320][H], H % 64 == 0, 128 <= H <= 1280")
    if out is None:
        out = torch.empty((M, pw_out.N), dtype=BF16, device=x.device)
    g = VkGemmDesc()
    g.A, g.lda = _p(x2d), lda
    g.amode, g.epi = AMODE_DENSE, EPI_GEGLU
    g.Wt, g.bias, g.M, g.N, g.K = _p(pw_in.wt), _p(pw_in.bias), M, pw_in.N, pw_in.K
    g.out, g.ldc = _p(out), out.stride(0)   # (ignored by the kernel; validate() wants a non-NULL pointer)
    g.alpha = 1.0
    g.tile_cfg = FF_FUSED_DBG
    if FF_FUSED_DBG == 8 and FF_FUSED_DBG_BUF is not None:
        g.splitk_ws = _p(FF_FUSED_DBG_BUF)
    _fill_ln(g, pw_in, ln, M)
    d = VkGemmDesc()
    d.A, d.lda = _p(x2d), lda               # (ignored)
    d.amode, d.epi = AMODE_DENSE, EPI_LINEAR
    _fill_epilogue(d, pw_out, out, M, rowvec, rows_per_vec, res1, res2, alpha, beta, rowvec2)
    lib = _lib.load()
    stats = None
    if emit_stats:
        parts = lib.vk_ff_fused_rowstat_parts()
        stats = RowStats(torch.empty((parts, M, 2), dtype=F32, device=x.device), parts, M)
        d.rowstat_out = _p(stats.t)
    check(lib.vk_ff_fused_bf16(C.byref(g), C.byref(d), _stream()), "vk_ff_fused_bf16")
    return (out, stats) if emit_stats else out


def conv3x3(x, pw, n_img, H, W, *, stride=1, ups=1, out=None, out_f32=False, rowvec=None, res1=None, res2=None, alpha=1.0,
            beta=0.0, asym_pad=False, gn=None):
    """3x3 conv, pad 1, over token-major x (n_img, H*W, Cin); `ups`=2 applies a nearest x2 upsample to the source
    on the fly (Upsample.forward, openaimodel.py:100-102); stride 2 = Downsample (openaimodel.py:136)."""
    _need(x, BF16, "x")
    if not x.is_contiguous():
        raise ValueError("conv3x3: x must be contiguous")
    cin = x.shape[-1]
    if pw.K != 9 * cin:
        raise ValueError(f"conv3x3: weight K {pw.K} != 9*{cin}")
    He, We = H * ups, W * ups
    pad = 1 if asym_pad else 2  # asym_pad: F.pad(x, (0,1,0,1)) + conv pad 0 (VAE encoder Downsample, model.py:77-81)
    Hout = (He + pad - 3) // stride + 1
    Wout = (We + pad - 3) // stride + 1
    M = n_img * Hout * Wout
    if out is None:
        out = torch.empty((M, pw.N), dtype=F32 if out_f32 else BF16, device=x.device)
    d = VkGemmDesc()
    d.A, d.lda = _p(x), cin
    d.amode, d.epi = AMODE_CONV3X3, EPI_LINEAR
    d.H, d.Wd, d.Cin, d.Hout, d.Wout, d.stride, d.ups = H, W, cin, Hout, Wout, stride, ups
    d.asym_pad = 1 if asym_pad else 0
    _fill_epilogue(d, pw, out, M, rowvec, Hout * Wout, res1, res2, alpha, beta)
    _ask_gn(d, gn, Hout * Wout, x.device)   # gn: a GnPartials to fill with the GroupNorm statistics of `out` (the norm that follows this conv)
    _gemm(d)
    return (out.view(n_img, Hout * Wout, pw.N) if out.is_contiguous() else out), Hout, Wout


def conv_t3(x, pw, T, S, *, out=None, out_f32=False, rowvec=None, res1=None, res2=None, alpha=1.0, beta=0.0, halo_prev=None,
            halo_next=None, gn=None):
    """3x1x1 temporal conv, pad (1,0,0) (video_model.py:38-52) over x ((b t), S, C). halo_prev / halo_next: (clips, S, C) frames
    adjacent to the local frame range (frame-sharded multi-GPU); None = zero padding."""
    _need(x, BF16, "x")
    if not x.is_contiguous():
        raise ValueError("conv_t3: x must be contiguous")
    cin = x.shape[-1]
    if pw.K != 3 * cin:
        raise ValueError("conv_t3: weight K mismatch")
    M = x.shape[0] * x.shape[1]
    if out is None:
        out = torch.empty((M, pw.N), dtype=F32 if out_f32 else BF16, device=x.device)
    d = VkGemmDesc()
    d.A, d.lda = _p(x), cin
    d.amode, d.epi = AMODE_TEMPORAL3, EPI_LINEAR
    d.Cin, d.T, d.S = cin, T, S
    for name, h in (("halo_prev", halo_prev), ("halo_next", halo_next)):
        if h is not None:
            _need(h, BF16, name)
            if not h.is_contiguous() or h.shape != (x.shape[0] // T, S, cin):
                raise ValueError(f"{name}: expected contiguous (clips, S, C) = {(x.shape[0] // T, S, cin)}, got {tuple(h.shape)}")
            setattr(d, name, _p(h))
    _fill_epilogue(d, pw, out, M, rowvec, S, res1, res2, alpha, beta)
    _ask_gn(d, gn, S, x.device)
    _gemm(d)
    return out.view(x.shape[0], S, pw.N)


def pack_conv3d(weight, bias=None, device="cuda", cin_pad=None):
    """nn.Conv3d weight [Cout][Cin][3][3][3] -> [Cout][Cin(_pad)/64][kt][ky][kx][64] (temporal VAE decoder)."""
    w = weight.detach().float().permute(0, 2, 3, 4, 1).contiguous()  # Cout, kt, ky, kx, Cin
    cout, _, _, _, cin = w.shape
    cp = cin_pad or ceil_to(cin, 64)
    if cp != cin:
        w = torch.nn.functional.pad(w, (0, cp - cin))
    return _finish_pack(_slab_major(w.reshape(cout, 27, cp)), None if bias is None else bias.detach().float(), device)


def conv3d(x, pw, T, H, W, *, out=None, out_f32=False, res1=None, res2=None, alpha=1.0, beta=0.0):
    """3x3x3 conv, pad 1, over x ((b t), H*W, Cin) with clips of T frames (AE3DConv.time_mix_conv and the decoder's time_stack)."""
    _need(x, BF16, "x")
    if not x.is_contiguous():
        raise ValueError("conv3d: x must be contiguous")
    cin = x.shape[-1]
    if pw.K != 27 * cin:
        raise ValueError(f"conv3d: weight K {pw.K} != 27*{cin}")
    M = x.shape[0] * H * W
    if out is None:
        out = torch.empty((M, pw.N), dtype=F32 if out_f32 else BF16, device=x.device)
    d = VkGemmDesc()
    d.A, d.lda = _p(x), cin
    d.amode, d.epi = AMODE_CONV3D, EPI_LINEAR
    d.H, d.Wd, d.Cin, d.T = H, W, cin, T
    _fill_epilogue(d, pw, out, M, None, 0, res1, res2, alpha, beta)
    _gemm(d)
    return out.view(x.shape[0], H * W, -1) if out.dim() == 2 and out.is_contiguous() else out


# ---------------------------------------------------------------------------------------------- fp8 GEMM (BASELINE config 5)
FP8_MAX = 448.0  # OCP e4m3fn


class PackedWeightF8:
    """fp8 e4m3 [ceil-tile(N)][Kp] weight (Kp = ceil128(K), zero-filled) + f32 per-output-channel scale + f32 bias."""

    __slots__ = ("wt", "scale", "bias", "N", "K", "Kp", "geglu")

    def __init__(self, wt, scale, bias, N, K, Kp, geglu=False):
        self.wt, self.scale, self.bias, self.N, self.K, self.Kp, self.geglu = wt, scale, bias, N, K, Kp, geglu


def _finish_pack_fp8(w2d, bias, device, geglu=False):
    N, K = w2d.shape
    if K % 16:
        raise ValueError("fp8 GEMM needs K % 16 == 0")
    Kp = ceil_to(K, 128)
    Np = max(ceil_to(N, 256), ceil_to(N, 320))
    amax = w2d.abs().amax(dim=1).clamp_min(1e-12)
    sc = (amax / FP8_MAX).float()
    q = (w2d / sc[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    wt = torch.zeros((Np, Kp), dtype=torch.uint8, device=device)
    wt[:N, :K] = q.view(torch.uint8).to(device)
    scale = torch.ones((Np,), dtype=F32, device=device)
    scale[:N] = sc.to(device)
    b = None
    if bias is not None:
        b = torch.zeros((Np,), dtype=F32, device=device)
        b[:N] = bias.to(device=device, dtype=F32)
    return PackedWeightF8(wt, scale, b, ceil_to(N, 4), K, Kp, geglu)


def pack_linear_fp8(weight, bias=None, device="cuda"):
    """nn.Linear weight [N][K] -> per-output-channel-scaled fp8 (host-side, once per weight load)."""
    return _finish_pack_fp8(weight.detach().reshape(weight.shape[0], -1).float().cpu(), None if bias is None else bias.detach().float().cpu(), device)


def pack_conv3x3_fp8(weight, bias=None, device="cuda"):
    """nn.Conv2d weight [Cout][Cin][3][3] (Cin % 64 == 0) -> e4m3 [Cout][Cin/64][ky][kx][64], one scale per output channel."""
    w = weight.detach().float().cpu().permute(0, 2, 3, 1).contiguous()
    cout, _, _, cin = w.shape
    if cin % 64:
        raise ValueError("fp8 conv3x3 needs Cin % 64 == 0")
    return _finish_pack_fp8(_slab_major(w.reshape(cout, 9, cin)), None if bias is None else bias.detach().float().cpu(), device)


def pack_conv_t3_fp8(weight, bias=None, device="cuda"):
    """nn.Conv3d weight [Cout][Cin][3][1][1] (Cin % 64 == 0) -> e4m3 [Cout][Cin/64][kt][64], one scale per output channel."""
    w = weight.detach().float().cpu()[:, :, :, 0, 0].permute(0, 2, 1).contiguous()
    if w.shape[2] % 64:
        raise ValueError("fp8 temporal conv needs Cin % 64 == 0")
    return _finish_pack_fp8(_slab_major(w), None if bias is None else bias.detach().float().cpu(), device)


def pack_geglu_fp8(weight, bias, device="cuda"):
    """GEGLU.proj in the value/gate-interleaved row order of pack_geglu, quantised per packed row."""
    w = weight.detach().float().cpu()
    b = bias.detach().float().cpu()
    perm = geglu_perm(w.shape[0] // 2)
    return _finish_pack_fp8(w[perm], b[perm], device, geglu=True)


def quantize_rows_fp8(x):
    """x (..., K) bf16 (rows may be strided) -> (q uint8 (M, K) holding e4m3 bytes, scale f32 (M,)) with x ~= q * scale[:, None]."""
    _need(x, BF16, "x")
    x2, ldx = _rows2d(x, "x")
    M, K = x2.shape
    q = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    scale = torch.empty((M,), dtype=F32, device=x.device)
    check(_lib.load().vk_quantize_rows_fp8(_p(x2), _p(q), _p(scale), M, K, ldx, K, _stream()), "vk_quantize_rows_fp8")
    return q, scale


def groupnorm_fp8(x, gamma, beta, eps, silu, frames_per_group=1, x2=None):
    """GroupNorm(32)[+SiLU] with e4m3 output and one scale per image group: x (n_img, S, C) bf16 [x2: second tensor of a channel concat]
    -> (y8 uint8 (n_img, S, C[+C2]), scale f32 (n_img / frames_per_group,)), GN(x) ~= y8 * scale[image group]. The scale is an upper bound
    from the statistics pass (include/vista_hip.h: vk_groupnorm_silu_fp8), not a measured maximum."""
    _need(x, BF16, "x")
    if not x.is_contiguous() or (x2 is not None and (not x2.is_contiguous() or x2.shape[:2] != x.shape[:2])):
        raise ValueError("groupnorm_fp8: contiguous (n_img, S, C) inputs with equal n_img, S required")
    n_img, S, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[2]
    ng = n_img // frames_per_group
    y = torch.empty((n_img, S, c1 + c2), dtype=torch.uint8, device=x.device)
    scale = torch.empty((ng,), dtype=F32, device=x.device)
    ws = torch.empty(ng * 64 + n_img * ((S + 31) // 32) * 65, dtype=F32, device=x.device)
    check(_lib.load().vk_groupnorm_silu_fp8(_p(x), _p(x2), _p(y), _p(scale), _p(gamma), _p(beta), _p(ws), n_img, S, c1, c2, frames_per_group,
                                            float(eps), 1 if silu else 0, _stream()), "vk_groupnorm_silu_fp8")
    return y, scale


def _conv_fp8(d, x8, scale, rows_per_scale, pw, M, rowvec, rows_per_vec, res1, res2, alpha, beta):
    out = torch.empty((M, pw.N), dtype=BF16, device=x8.device)
    _fill_epilogue(d, pw, out, M, rowvec, rows_per_vec, res1, res2, alpha, beta)
    d.K = pw.Kp
    d.tile_cfg = 0
    a = VkFp8Args()
    a.a_scale, a.w_scale, a.k_real, a.a_scale_rows = _p(scale), _p(pw.scale), pw.K, rows_per_scale
    check(_lib.load().vk_gemm_fp8_mx(C.byref(d), C.byref(a), _stream()), "vk_gemm_fp8_mx")
    return out


def conv3x3_fp8(x8, scale, pw, n_img, H, W, *, frames_per_scale=1, rowvec=None, res1=None, res2=None, alpha=1.0, beta=0.0):
    """3x3 conv, stride 1, pad 1, over the e4m3 output of groupnorm_fp8 (x8 (n_img, H*W, Cin) uint8, scale one per `frames_per_scale`
    images) with fp8 weights (pack_conv3x3_fp8) -> (n_img, H*W, Cout) bf16; epilogue options as conv3x3."""
    if x8.dtype != torch.uint8 or not x8.is_contiguous() or x8.shape[:2] != (n_img, H * W):
        raise TypeError("conv3x3_fp8: x8 must be a contiguous (n_img, H*W, Cin) uint8 tensor")
    cin = x8.shape[-1]
    if pw.K != 9 * cin:
        raise ValueError(f"conv3x3_fp8: weight K {pw.K} != 9*{cin}")
    M = n_img * H * W
    d = VkGemmDesc()
    d.A, d.lda = _p(x8), cin
    d.amode, d.epi = AMODE_CONV3X3, EPI_LINEAR
    d.H, d.Wd, d.Cin, d.Hout, d.Wout, d.stride, d.ups = H, W, cin, H, W, 1, 1
    return _conv_fp8(d, x8, scale, frames_per_scale * H * W, pw, M, rowvec, H * W, res1, res2, alpha, beta).view(n_img, H * W, pw.N)


def conv_t3_fp8(x8, scale, pw, T, S, *, rowvec=None, res1=None, res2=None, alpha=1.0, beta=0.0):
    """3x1x1 temporal conv, pad (1,0,0), over the e4m3 output of groupnorm_fp8(frames_per_group=T): x8 ((b t), S, Cin) uint8, scale (b,)."""
    if x8.dtype != torch.uint8 or not x8.is_contiguous() or x8.shape[1] != S or x8.shape[0] % T:
        raise TypeError("conv_t3_fp8: x8 must be a contiguous ((b t), S, Cin) uint8 tensor")
    cin = x8.shape[-1]
    if pw.K != 3 * cin:
        raise ValueError("conv_t3_fp8: weight K mismatch")
    M = x8.shape[0] * S
    d = VkGemmDesc()
    d.A, d.lda = _p(x8), cin
    d.amode, d.epi = AMODE_TEMPORAL3, EPI_LINEAR
    d.Cin, d.T, d.S = cin, T, S
    return _conv_fp8(d, x8, scale, T * S, pw, M, rowvec, S, res1, res2, alpha, beta).view(x8.shape[0], S, pw.N)


def layernorm_quant_fp8(x, norm):
    """LayerNorm fused with per-row e4m3 quantisation (one pass): x (..., C) bf16 -> (q uint8 (M, C), scale f32 (M,)) with LN(x) ~= q * scale[:, None]."""
    _need(x, BF16, "x")
    if not x.is_contiguous():
        raise ValueError("layernorm_quant_fp8: x must be contiguous")
    Cc = x.shape[-1]
    M = x.numel() // Cc
    q = torch.empty((M, Cc), dtype=torch.uint8, device=x.device)
    scale = torch.empty((M,), dtype=F32, device=x.device)
    check(_lib.load().vk_layernorm_quant_fp8(_p(x), _p(q), _p(scale), _p(norm.weight), _p(norm.bias), M, Cc, float(norm.eps), _stream()),
          "vk_layernorm_quant_fp8")
    return q, scale


def linear_fp8(xq, a_scale, pw, *, out=None, out_f32=False, rowvec=None, rows_per_vec=0, res1=None, res2=None, alpha=1.0, beta=0.0,
               rowvec2=None, a_mx=None, mx_out=False, emit_stats=False):
    """out = alpha*((xq*scales) @ (Wq*w_scale)^T + bias + rowvec + res1) + beta*(res2 + rowvec2), fp8 x fp8 -> f32 accumulate.
    Activation scales: `a_scale` f32 (M,) per row, OR `a_mx` uint8 (M, K/32) E8M0 block scales (MX; a_scale None).
    mx_out (GEGLU weights only): return (h8 uint8 (M, nout), hs uint8 (M, nout/32)) -- the gated output quantised to MX fp8 in the epilogue.
    emit_stats: also return the RowStats of the (bf16) output."""
    if xq.dtype != torch.uint8 or xq.dim() != 2 or xq.stride(1) != 1:
        raise TypeError("linear_fp8: xq must be a (M, K) uint8 tensor of e4m3 bytes")
    M, K = xq.shape
    if K != pw.K:
        raise ValueError(f"linear_fp8: K mismatch {K} vs {pw.K}")
    if (a_scale is None) == (a_mx is None):
        raise ValueError("linear_fp8: exactly one of a_scale (per row) / a_mx (MX block scales)")
    nout = pw.N // 2 if pw.geglu else pw.N
    hs = None
    if mx_out:
        if not pw.geglu or nout % 32:
            raise ValueError("mx_out: GEGLU weights with an output width that is a multiple of 32")
        out = torch.empty((M, nout), dtype=torch.uint8, device=xq.device)
        hs = torch.empty((M, nout // 32), dtype=torch.uint8, device=xq.device)
    elif out is None:
        out = torch.empty((M, nout), dtype=F32 if (out_f32 and not pw.geglu) else BF16, device=xq.device)
    d = VkGemmDesc()
    d.A, d.lda = _p(xq), xq.stride(0)
    d.amode = AMODE_DENSE
    d.epi = EPI_GEGLU if pw.geglu else EPI_LINEAR
    _fill_epilogue(d, pw, out, M, rowvec, rows_per_vec, res1, res2, alpha, beta, rowvec2)
    d.K = pw.Kp
    d.tile_cfg = TILE_CFG & 7
    lib = _lib.load()
    stats = None
    if emit_stats:
        parts = lib.vk_gemm_fp8_rowstat_parts(C.byref(d))
        if parts <= 0:
            raise _lib.VistaHipError(f"vk_gemm_fp8_rowstat_parts failed with code {parts}")
        stats = RowStats(torch.empty((parts, M, 2), dtype=F32, device=xq.device), parts, M)
        d.rowstat_out = _p(stats.t)
    a = VkFp8Args()
    a.a_scale, a.w_scale, a.k_real = _p(a_scale), _p(pw.scale), K
    if a_mx is not None:
        if a_mx.dtype != torch.uint8 or a_mx.dim() != 2 or a_mx.shape[0] != M or a_mx.shape[1] < K // 32 or a_mx.stride(1) != 1 or a_mx.stride(0) % 4:
            raise ValueError("a_mx: uint8 (M, >= K/32) E8M0 block scales, row stride a multiple of 4 (the kernel reads one dword per 128 K-bytes)")
        a.a_mx, a.ld_mx = _p(a_mx), a_mx.stride(0)
    if mx_out:
        a.mx_out, a.ld_mx_out = _p(hs), hs.stride(0)
    check(lib.vk_gemm_fp8_mx(C.byref(d), C.byref(a), _stream()), "vk_gemm_fp8_mx")
    if mx_out:
        return out, hs
    return (out, stats) if emit_stats else out


# ---------------------------------------------------------------------------------------------- attention
PROFILE_ATTN = None  # bench.py sets this to a list: (S, n_img*heads, start_event, end_event) per launch, on the launch stream


LOG2E = 1.4426950408889634


def attn_spatial(q, k, vt, n_img, heads, S, scale=None, v_rows=False, q_log2=False):
    """q, k: 2-D strided views (n_img*S, heads*64) bf16. vt: (n_img, heads*64, S) = V transposed (linear_vt), or with v_rows=True a
    2-D strided view (n_img*S, heads*64) like q and k -- the v column block of ONE fused q|k|v GEMM. Returns (n_img*S, heads*64).
    q_log2 (with v_rows): q already carries softmax_scale * log2(e) (folded into the query weights at pack time), so q.k is the base-2
    exponent itself and the kernel's zero-base path needs no scale / base fma (vk_attn_spatial_qkv_log2_bf16); `scale` is not used."""
    _need(q, BF16, "q"); _need(k, BF16, "k"); _need(vt, BF16, "v" if v_rows else "vt")
    o = torch.empty((n_img * S, heads * 64), dtype=BF16, device=q.device)
    lib = _lib.load()
    ev = None
    if PROFILE_ATTN is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    sc = float(scale if scale is not None else 1.0 / math.sqrt(64))
    if v_rows:
        if vt.dim() != 2 or vt.stride(1) != 1 or vt.shape != (n_img * S, heads * 64):
            raise ValueError("attn_spatial: v must be a (n_img*S, heads*64) view with contiguous rows")
        if q_log2:
            check(lib.vk_attn_spatial_qkv_log2_bf16(_p(q), _p(k), _p(vt), _p(o), n_img, heads, S, q.stride(0), k.stride(0), vt.stride(0), o.stride(0),
                                                    _stream()), "vk_attn_spatial_qkv_log2_bf16")
        else:
            check(lib.vk_attn_spatial_qkv_bf16(_p(q), _p(k), _p(vt), _p(o), n_img, heads, S, q.stride(0), k.stride(0), vt.stride(0), o.stride(0), sc,
                                               _stream()), "vk_attn_spatial_qkv_bf16")
    else:
        if ACT is not torch.bfloat16:
            raise _lib.VistaHipError("attn_spatial with a V^T tensor: bf16 build only (in the fp16 build V must come from a q|k|v projection launched "
                                     "with alt_cols_from, i.e. as bf16 rows)")
        check(lib.vk_attn_spatial_bf16(_p(q), _p(k), _p(vt), _p(o), n_img, heads, S, q.stride(0), k.stride(0), o.stride(0), sc, _stream()),
              "vk_attn_spatial_bf16")
    if ev is not None:
        ev[1].record()
        PROFILE_ATTN.append((S, n_img * heads, ev[0], ev[1]))
    return o


def attn_spatial_fp8qk(q8, k8, qs, ks, v, n_img, heads, S, scale=None, mx_out=False):
    """BASELINE config 5: spatial self-attention with the score product in fp8. q8 / k8: (n_img*S, heads*64) uint8 views of e4m3 bytes (rows
    may be strided: the q | k blocks of linear(..., mx8_cols=2C)); qs / ks: (n_img*S, 2*heads) uint8 views of their E8M0 block scales; v:
    (n_img*S, heads*64) bf16 rows. Returns bf16 (n_img*S, heads*64), or with mx_out=True (o8 uint8 (M, C), o_scales uint8 (M, C/32)).
    scale=0.0: q already carries softmax_scale * log2(e) (the pre-scaled query of attn_spatial(..., q_log2=True))."""
    _need(v, BF16, "v")
    for name, t in (("q8", q8), ("k8", k8), ("qs", qs), ("ks", ks)):
        if t.dtype != torch.uint8 or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
            raise TypeError(f"attn_spatial_fp8qk: {name} must be a 2-D uint8 device view with contiguous rows")
    M, c = n_img * S, heads * 64
    if q8.shape != (M, c) or k8.shape != (M, c) or qs.shape != (M, 2 * heads) or ks.shape != (M, 2 * heads) or v.shape != (M, c) or v.stride(1) != 1:
        raise ValueError("attn_spatial_fp8qk: shape mismatch")
    o = o8 = osc = None
    if mx_out:
        o8 = torch.empty((M, c), dtype=torch.uint8, device=v.device)
        osc = torch.empty((M, ceil_to(2 * heads, 4)), dtype=torch.uint8, device=v.device)  # row stride % 4 for the consumer GEMM; the kernel sets the pad bytes to 2^0
    else:
        o = torch.empty((M, c), dtype=BF16, device=v.device)
    ev = None
    if PROFILE_ATTN is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    check(_lib.load().vk_attn_spatial_fp8qk(_p(q8), _p(k8), _p(qs), _p(ks), _p(v), _p(o), _p(o8), _p(osc), n_img, heads, S, q8.stride(0), k8.stride(0),
                                            qs.stride(0), ks.stride(0), v.stride(0), c, c, ceil_to(2 * heads, 4),
                                            float(scale if scale is not None else 1.0 / math.sqrt(64)), _stream()), "vk_attn_spatial_fp8qk")
    if ev is not None:
        ev[1].record()
        PROFILE_ATTN.append((S, n_img * heads, ev[0], ev[1]))
    return (o8, osc) if mx_out else o


def attn_small(qkv, n_img, heads, S, D, scale=None):
    """qkv: (n_img*S, 3*heads*D) bf16 row-major [q | k | v]; softmax(q k^T * scale) v per (image, head), D in {64, 80, 128}
    (the OpenCLIP image tower's nn.MultiheadAttention). Returns (n_img*S, heads*D)."""
    _need(qkv, BF16, "qkv")
    c = heads * D
    if qkv.dim() != 2 or qkv.stride(1) != 1 or qkv.shape != (n_img * S, 3 * c):
        raise ValueError(f"attn_small: qkv must be (n_img*S, 3*heads*D) = {(n_img * S, 3 * c)} with contiguous rows, got {tuple(qkv.shape)}")
    o = torch.empty((n_img * S, c), dtype=BF16, device=qkv.device)
    check(_lib.load().vk_attn_small_bf16(_p(qkv), _p(o), n_img, heads, S, D, qkv.stride(0), c, 2 * c, o.stride(0),
                                         float(scale if scale is not None else 1.0 / math.sqrt(D)), _stream()), "vk_attn_small_bf16")
    return o


def antialias_blur_params(in_size, out_size):
    """(sigma, kernel size) of kornia-0.6.9 `resize(..., antialias=True)` along one axis (kornia/geometry/transform/affwarp.py: the
    blur only exists when downscaling; sigma = max((factor - 1) / 2, 0.001), ks = int(max(4 sigma, 3)) made odd). No blur: (1.0, 1)."""
    factor = in_size / out_size
    sigma = max((factor - 1.0) / 2.0, 0.001)
    ks = int(max(2.0 * 2 * sigma, 3))
    if ks % 2 == 0:
        ks += 1
    return sigma, ks


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess_patches(img, out_hw=224, patch=14, kpad=None, antialias=True, mean=CLIP_MEAN, std=CLIP_STD):
    """img (n, 3, H, W) f32 in [-1, 1] -> (n * (1 + g*g), Kp) bf16, g = out_hw / patch: row 0 of every image zero (class-token slot), row
    1 + py*g + px = patch (py, px) of the resized / normalised image flattened [c][ky][kx], zero-padded to Kp = ceil64(3*patch^2).
    FrozenOpenCLIPImageEmbedder.preprocess (modules.py:304-315) fused with the patch convolution's im2col."""
    _need(img, F32, "img")
    img = img.contiguous()
    n, c, H, W = img.shape
    if c != 3:
        raise ValueError("clip_preprocess_patches: 3-channel images")
    g = out_hw // patch
    kp = kpad or ceil_to(3 * patch * patch, 64)
    out = torch.zeros((n * (1 + g * g), kp), dtype=BF16, device=img.device)
    down = antialias and max(H / out_hw, W / out_hw) > 1  # kornia blurs BOTH axes as soon as either one is downscaled
    (sy, ky), (sx, kx) = (antialias_blur_params(H, out_hw), antialias_blur_params(W, out_hw)) if down else ((1.0, 1), (1.0, 1))
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(_lib.load().vk_clip_preprocess_patches(_p(img), _p(out), n, H, W, out_hw, patch, kp, sy, sx, ky, kx, m3, s3, _stream()),
          "vk_clip_preprocess_patches")
    return out


def attn_temporal(qkv, B, T, S, heads, scale=None):
    """qkv: ((b t) s, 3*heads*64) bf16 row-major [q | k | v]. Returns ((b t) s, heads*64)."""
    _need(qkv, BF16, "qkv")
    c = heads * 64
    o = torch.empty((B * T * S, c), dtype=BF16, device=qkv.device)
    lib = _lib.load()
    check(lib.vk_attn_temporal_bf16(_p(qkv), _p(o), B, T, S, heads, qkv.stride(0), c, 2 * c, o.stride(0),
                                    float(scale if scale is not None else 1.0 / math.sqrt(64)), _stream()), "vk_attn_temporal_bf16")
    return o


# ---------------------------------------------------------------------------------------------- norms
def softmax_rows(x, out=None):
    """x (rows, cols) fp32 (row stride may exceed cols) -> bf16 softmax over the last dim (VAE decoder AttnBlock)."""
    _need(x, F32, "x")
    rows, cols = x.shape
    if x.stride(1) != 1:
        raise ValueError("softmax_rows: rows must be contiguous")
    if out is None:
        out = torch.empty((rows, cols), dtype=BF16, device=x.device)
    check(_lib.load().vk_softmax_rows_f32_bf16(_p(x), _p(out), rows, cols, x.stride(0), out.stride(0), _stream()), "vk_softmax_rows_f32_bf16")
    return out


def _gn_partials_ok(gn, n_img, S, Cc):
    """The producer's partials describe exactly this tensor (and have not been consumed yet)."""
    if gn is None or gn.t is None:
        return False
    if gn.rows != n_img * S or gn.C != Cc or gn.nchunks * 64 != S:
        raise ValueError(f"GnPartials of a ({gn.rows} rows, {gn.C} channels, {gn.nchunks * 64} rows per image) tensor passed to a GroupNorm over ({n_img}, {S}, {Cc})")
    return True


def groupnorm(x, gamma, beta, eps, silu, frames_per_group=1, out=None, gn=None):
    """x (n_img, S, C) bf16 contiguous. gn: the GnPartials the producer of x filled (conv3x3 / conv_t3 (..., gn=...)): fold + apply, no
    statistics pass over x."""
    _need(x, BF16, "x")
    if not x.is_contiguous():
        raise ValueError("groupnorm: x must be contiguous")
    n_img, S, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    if _gn_partials_ok(gn, n_img, S, Cc):
        lib = _lib.load()
        count = float(Cc // 32) * float(S) * float(frames_per_group)
        if frames_per_group * gn.nchunks <= lib.vk_groupnorm_fold_max():   # ABI v7: the apply workgroups fold the slots themselves (no finalize launch)
            check(lib.vk_groupnorm_apply_partials_bf16(_p(x), _p(out), _p(gamma), _p(beta), _p(gn.t), n_img, S, Cc, gn.nchunks, frames_per_group, count,
                                                       float(eps), 1 if silu else 0, _stream()), "vk_groupnorm_apply_partials_bf16")
            gn.t = None   # consumed
            return out
        sums = torch.empty((n_img // frames_per_group) * 64, dtype=F32, device=x.device)
        check(lib.vk_groupnorm_finalize_partials(_p(gn.t), _p(sums), n_img, gn.nchunks, frames_per_group, _stream()), "vk_groupnorm_finalize_partials")
        gn.t = None   # consumed
        check(lib.vk_groupnorm_apply_bf16(_p(x), _p(out), _p(gamma), _p(beta), _p(sums), n_img, S, Cc, frames_per_group, count, float(eps),
                                          1 if silu else 0, _stream()), "vk_groupnorm_apply_bf16")
        return out
    ws = torch.empty(((n_img // frames_per_group) + n_img * ((S + 31) // 32)) * 64, dtype=F32, device=x.device)
    lib = _lib.load()
    check(lib.vk_groupnorm_silu_bf16(_p(x), _p(out), _p(gamma), _p(beta), _p(ws), n_img, S, Cc, frames_per_group, float(eps),
                                     1 if silu else 0, _stream()), "vk_groupnorm_silu_bf16")
    return out


def groupnorm_cat(a, b, gamma, beta, eps, silu, frames_per_group=1):
    """GroupNorm(32)[+SiLU] of the channel concat [a | b] ((n_img, S, C1) and (n_img, S, C2) bf16) -> (n_img, S, C1+C2), without
    materialising the concat (the skip `torch.cat` of the UNet's output blocks, video_model.py:493)."""
    _need(a, BF16, "a"); _need(b, BF16, "b")
    if not (a.is_contiguous() and b.is_contiguous()) or a.shape[:2] != b.shape[:2]:
        raise ValueError("groupnorm_cat: contiguous (n_img, S, C) inputs with equal n_img, S required")
    n_img, S, c1 = a.shape
    c2 = b.shape[2]
    out = torch.empty((n_img, S, c1 + c2), dtype=BF16, device=a.device)
    ws = torch.empty(((n_img // frames_per_group) + n_img * ((S + 31) // 32)) * 64, dtype=F32, device=a.device)
    check(_lib.load().vk_groupnorm_silu_cat_bf16(_p(a), _p(b), _p(out), _p(gamma), _p(beta), _p(ws), n_img, S, c1, c2, frames_per_group,
                                                 float(eps), 1 if silu else 0, _stream()), "vk_groupnorm_silu_cat_bf16")
    return out


def groupnorm_sharded(x, gamma, beta, eps, silu, frames_per_group, allreduce, global_count, gn=None):
    """GroupNorm whose statistics also span other ranks' shards of the same images (pixel-sharded temporal ResBlock):
    local fixed-order sums -> `allreduce(sums)` (in place, SUM over ranks) -> apply with the GLOBAL element count.
    gn: the producer's GnPartials of x (as groupnorm)."""
    _need(x, BF16, "x")
    n_img, S, Cc = x.shape
    out = torch.empty_like(x)
    ng = n_img // frames_per_group
    sums = torch.empty(ng * 64, dtype=F32, device=x.device)
    lib = _lib.load()
    if _gn_partials_ok(gn, n_img, S, Cc):
        check(lib.vk_groupnorm_finalize_partials(_p(gn.t), _p(sums), n_img, gn.nchunks, frames_per_group, _stream()), "vk_groupnorm_finalize_partials")
        gn.t = None
    else:
        part = torch.empty(n_img * ((S + 31) // 32) * 64, dtype=F32, device=x.device)
        check(lib.vk_groupnorm_stats_bf16(_p(x), _p(sums), _p(part), n_img, S, Cc, frames_per_group, _stream()), "vk_groupnorm_stats_bf16")
    allreduce(sums)
    check(lib.vk_groupnorm_apply_bf16(_p(x), _p(out), _p(gamma), _p(beta), _p(sums), n_img, S, Cc, frames_per_group, float(global_count),
                                      float(eps), 1 if silu else 0, _stream()), "vk_groupnorm_apply_bf16")
    return out


def layernorm(x, gamma, beta, eps=1e-5, addvec=None, rows_per_vec=0, want_sum=False):
    """x (..., C) bf16 contiguous -> LN(x + addvec[row // rows_per_vec]); optionally also returns the sum."""
    _need(x, BF16, "x")
    if not x.is_contiguous():
        raise ValueError("layernorm: x must be contiguous")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty_like(x)
    s = torch.empty_like(x) if (want_sum and addvec is not None) else None
    lib = _lib.load()
    check(lib.vk_layernorm_bf16(_p(x), _p(y), _p(s), _p(gamma), _p(beta), _p(addvec), rows, Cc, int(rows_per_vec),
                                addvec.stride(0) if addvec is not None else 0, float(eps), _stream()), "vk_layernorm_bf16")
    return (y, s) if want_sum else y


# ---------------------------------------------------------------------------------------------- elementwise
def concat_channels(a, b):
    _need(a, BF16, "a"); _need(b, BF16, "b")
    if not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("concat_channels: contiguous inputs required")
    c1, c2 = a.shape[-1], b.shape[-1]
    rows = a.numel() // c1
    out = torch.empty(a.shape[:-1] + (c1 + c2,), dtype=BF16, device=a.device)
    check(_lib.load().vk_concat_channels_bf16(_p(a), _p(b), _p(out), rows, c1, c2, _stream()), "vk_concat_channels_bf16")
    return out


def nchw_to_tokens(x, cpad):
    _need(x, F32, "x")
    n, c, h, w = x.shape
    x = x.contiguous()
    out = torch.empty((n, h * w, cpad), dtype=BF16, device=x.device)
    check(_lib.load().vk_nchw_to_tokens_bf16(_p(x), _p(out), n, c, h * w, cpad, _stream()), "vk_nchw_to_tokens_bf16")
    return out


def tokens_to_nchw(x, n_img, Cc, H, W):
    _need(x, F32, "x")
    out = torch.empty((n_img, Cc, H, W), dtype=F32, device=x.device)
    check(_lib.load().vk_tokens_to_nchw_f32(_p(x), _p(out), n_img, Cc, H * W, x.stride(-2), _stream()), "vk_tokens_to_nchw_f32")
    return out


def timestep_embedding(t, dim, max_period=10000.0, out_f32=False):
    _need(t, F32, "t")
    t = t.contiguous()
    out = torch.empty((t.shape[0], dim), dtype=F32 if out_f32 else BF16, device=t.device)
    lib = _lib.load()
    fn = lib.vk_timestep_embedding_f32 if out_f32 else lib.vk_timestep_embedding_bf16
    check(fn(_p(t), _p(out), t.shape[0], dim, float(max_period), _stream()), "vk_timestep_embedding")
    return out


def emb_combine(a, b, c, mask):
    """emb = a*mask + b*(1-mask) + c ; returns (emb f32, silu(emb) bf16)."""
    n, dim = b.shape
    emb = torch.empty((n, dim), dtype=F32, device=b.device)
    se = torch.empty((n, dim), dtype=BF16, device=b.device)
    check(_lib.load().vk_emb_combine(_p(a), _p(b), _p(c), _p(mask), _p(emb), _p(se), n, dim, _stream()), "vk_emb_combine")
    return emb, se


def silu_to_bf16(x):
    _need(x, F32, "x")
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(_lib.load().vk_silu_f32_to_bf16(_p(x), _p(y), x.numel(), _stream()), "vk_silu_f32_to_bf16")
    return y


def cast_to_bf16(x):
    _need(x, F32, "x")
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(_lib.load().vk_cast_f32_to_bf16(_p(x), _p(y), x.numel(), _stream()), "vk_cast_f32_to_bf16")
    return y


# ---- sampler-side ----
def sampler_prepare(x, cond_frame, mask, concat_uc, concat_c, cpad, c_in, replace):
    T, _, H, W = x.shape
    net_in = torch.empty((2 * T, H * W, cpad), dtype=BF16, device=x.device)
    check(_lib.load().vk_sampler_prepare(_p(x), _p(cond_frame), _p(mask), _p(concat_uc), _p(concat_c), _p(net_in), T, H * W, cpad,
                                         float(c_in), 1 if replace else 0, _stream()), "vk_sampler_prepare")
    return net_in


def sampler_update(x, net_out, scale, c_out, c_skip, sigma, sigma_next):
    T, _, H, W = x.shape
    check(_lib.load().vk_sampler_update(_p(x), _p(net_out), _p(scale), T, H * W, net_out.stride(-2), float(c_out), float(c_skip),
                                        float(sigma), float(sigma_next), _stream()), "vk_sampler_update")
    return x


def _c(*ts):
    return [t.contiguous() for t in ts]


def denoiser_combine(net, x, c_out, c_skip):
    net, x, c_out, c_skip = _c(net, x, c_out, c_skip)
    out = torch.empty_like(x)
    n = x.shape[0]
    check(_lib.load().vk_denoiser_combine(_p(net), _p(x), _p(c_out), _p(c_skip), _p(out), n, x.numel() // n, _stream()),
          "vk_denoiser_combine")
    return out


def cfg_combine(x2, scale):
    x2, scale = _c(x2, scale)
    T = x2.shape[0] // 2
    out = torch.empty((T,) + tuple(x2.shape[1:]), dtype=F32, device=x2.device)
    check(_lib.load().vk_cfg_combine(_p(x2), _p(scale), _p(out), T, out.numel() // T, _stream()), "vk_cfg_combine")
    return out


def euler_step(x, den, sigma, sigma_next):
    x, den, sigma, sigma_next = _c(x, den, sigma, sigma_next)
    out = torch.empty_like(x)
    n = x.shape[0]
    check(_lib.load().vk_euler_step(_p(x), _p(den), _p(sigma), _p(sigma_next), _p(out), n, x.numel() // n, _stream()), "vk_euler_step")
    return out


def mask_replace(x, cond, mask):
    x, cond, mask = _c(x, cond, mask)
    out = torch.empty_like(x)
    n = x.shape[0]
    check(_lib.load().vk_mask_replace(_p(x), _p(cond), _p(mask), _p(out), n, x.numel() // n, _stream()), "vk_mask_replace")
    return out


def gaussian_sample(moments, noise=None, scale=1.0):
    """moments (n, 2C, h, w) f32 -> (n, C, h, w): (mean + exp(0.5*clamp(logvar)) * noise) * scale; noise None = mode."""
    _need(moments, F32, "moments")
    moments = moments.contiguous()
    n, c2, h, w = moments.shape
    if noise is not None:
        _need(noise, F32, "noise")
        noise = noise.contiguous()
        if noise.shape != (n, c2 // 2, h, w):
            raise ValueError("gaussian_sample: noise must be (n, C, h, w)")
    out = torch.empty((n, c2 // 2, h, w), dtype=F32, device=moments.device)
    check(_lib.load().vk_gaussian_sample(_p(moments), _p(noise), _p(out), n, c2 // 2, h * w, float(scale), _stream()), "vk_gaussian_sample")
    return out


def ensemble_variance_sum(x):
    """x (E, ...) f32 -> python float: sum over elements of the unbiased variance across the E ensemble members."""
    _need(x, F32, "x")
    x = x.contiguous()
    E = x.shape[0]
    n = x.numel() // E
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    ws = torch.empty(512, dtype=torch.float64, device=x.device)
    check(_lib.load().vk_ensemble_variance_sum(_p(x), _p(out), _p(ws), E, n, _stream()), "vk_ensemble_variance_sum")
    return float(out.item())


def scale_rows(x, s):
    x, s = _c(x, s)
    out = torch.empty_like(x)
    n = x.shape[0]
    check(_lib.load().vk_scale_rows(_p(x), _p(s), _p(out), n, x.numel() // n, _stream()), "vk_scale_rows")
    return out
