"""ctypes binding of libvista_hip.so (C ABI declared in include/vista_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the HIP library is missing or fails to load, every op
raises. `load()` is the single place the shared object is opened (in-tree, vista_amd/lib/libvista_hip.so).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VISTA_HIP_LIB: A/B tooling only (tools/build_rev.sh builds the library of another git revision next to the in-tree one so that two
# kernel versions can be timed on the SAME box in one call); the product always loads the in-tree build.
# VISTA_ACT_DTYPE (round 6): the 16-bit storage type of activations and weights of THIS process's denoiser -- "bf16" (default: libvista_hip.so, the
# dtype BASELINE config 2 names) or "fp16" (libvista_hip_f16.so, the same kernels built with -DVK_F16=1: the reference's own autocast width,
# sample_utils.py:301-303; 7.6x closer to the fp32 reference at +4 % step time). ops.ACT follows it. The first-stage VAE and the
# conditioner always store bf16 (the reference runs them without autocast; fp16's 5-bit exponent is not safe there): inside an fp16 process their
# forwards run under ops.storage(torch.bfloat16), which switches CURRENT -- the library load() hands out -- for the duration of the call.
ACT_DTYPE = os.environ.get("VISTA_ACT_DTYPE", "bf16").lower()
if ACT_DTYPE not in ("bf16", "fp16"):
    raise ValueError(f"VISTA_ACT_DTYPE must be 'bf16' or 'fp16', not {ACT_DTYPE!r}")
CURRENT = ACT_DTYPE   # the storage type whose library load() returns: ops.storage() switches it
_LIB_FILES = {"bf16": "libvista_hip.so", "fp16": "libvista_hip_f16.so"}
LIB_PATH = os.environ.get("VISTA_HIP_LIB") or os.path.join(_HERE, "lib", _LIB_FILES[ACT_DTYPE])


def lib_path(dtype_name):
    """VISTA_HIP_LIB (A/B tooling) replaces the library of the process's own storage type only."""
    return LIB_PATH if dtype_name == ACT_DTYPE else os.path.join(_HERE, "lib", _LIB_FILES[dtype_name])


_vp = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float


class VkFp8Args(C.Structure):
    _fields_ = [("a_scale", _vp), ("w_scale", _vp), ("k_real", _i32), ("a_mx", _vp), ("ld_mx", _i32), ("mx_out", _vp), ("ld_mx_out", _i32),
                ("a_scale_rows", _i32)]


class VkGemmDesc(C.Structure):
    _fields_ = [
        ("A", _vp), ("Wt", _vp), ("out", _vp), ("bias", _vp), ("rowvec", _vp), ("res1", _vp), ("res2", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("lda", _i32), ("ldc", _i32), ("ld_res1", _i32), ("ld_res2", _i32), ("ldv", _i32), ("rows_per_vec", _i32),
        ("alpha", _f32), ("beta", _f32),
        ("amode", _i32), ("epi", _i32), ("out_f32", _i32),
        ("H", _i32), ("Wd", _i32), ("Cin", _i32), ("Hout", _i32), ("Wout", _i32), ("stride", _i32), ("ups", _i32),
        ("T", _i32), ("S", _i32), ("tile_cfg", _i32), ("halo_prev", _vp), ("halo_next", _vp), ("splitk_ws", _vp), ("splitk_ws_bytes", _i64), ("asym_pad", _i32),
        ("k_split", _i32), ("A2", _vp), ("lda2", _i32),
        ("ln_parts", _i32), ("ln_stats", _vp), ("ln_colsum", _vp), ("ln_eps", _f32),
        ("rowstat_out", _vp), ("rowvec2", _vp), ("act", _i32),
        ("mx8_out", _vp), ("mx8_scales", _vp), ("mx8_cols", _i32), ("ld_mx8", _i32), ("ld_mx8s", _i32),
        ("m_begin", _i32), ("m_end", _i32),
        ("gnstat_out", _vp), ("gn_rows", _i32),
        ("alt_cols_from", _i32),
    ]


ABI_VERSION = 7  # vk_abi_version() of the library this table mirrors

# name -> argtypes; every entry returns int. Must list every symbol include/vista_hip.h declares
# (tests/test_abi.py checks the header against this table and against the built library).
SIGNATURES = {
    "vk_gemm_bf16": [C.POINTER(VkGemmDesc), _vp],
    "vk_gemm_rowstat_parts": [C.POINTER(VkGemmDesc)],
    "vk_gemm_tile_choice": [C.POINTER(VkGemmDesc)],
    "vk_gemm_tail_split": [C.POINTER(VkGemmDesc)],
    "vk_gemm_gnstat_fit": [C.POINTER(VkGemmDesc)],
    "vk_gemm_fp8": [C.POINTER(VkGemmDesc), _vp, _vp, _i32, _vp],
    "vk_gemm_fp8_mx": [C.POINTER(VkGemmDesc), C.POINTER(VkFp8Args), _vp],
    "vk_gemm_fp8_rowstat_parts": [C.POINTER(VkGemmDesc)],
    "vk_ff_fused_bf16": [C.POINTER(VkGemmDesc), C.POINTER(VkGemmDesc), _vp],
    "vk_ff_fused_rowstat_parts": [],
    "vk_layernorm_quant_fp8": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp],
    "vk_groupnorm_silu_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "vk_quantize_rows_fp8": [_vp, _vp, _vp, _i32, _i32, _i64, _i64, _vp],
    "vk_attn_spatial_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "vk_attn_spatial_qkv_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "vk_attn_spatial_qkv_log2_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "vk_attn_spatial_fp8qk": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "vk_attn_small_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "vk_clip_preprocess_patches": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _vp],
    "vk_attn_temporal_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "vk_softmax_rows_f32_bf16": [_vp, _vp, _i64, _i32, _i64, _i64, _vp],
    "vk_groupnorm_silu_bf16": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "vk_groupnorm_stats_bf16": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "vk_groupnorm_apply_bf16": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp],
    "vk_groupnorm_finalize_partials": [_vp, _vp, _i32, _i32, _i32, _vp],
    "vk_groupnorm_fold_max": [],
    "vk_groupnorm_apply_partials_bf16": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp],
    "vk_layernorm_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "vk_rowstats_bf16": [_vp, _vp, _i32, _i32, _i64, _vp],
    "vk_groupnorm_silu_cat_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "vk_concat_channels_bf16": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "vk_nchw_to_tokens_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "vk_tokens_to_nchw_f32": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "vk_timestep_embedding_bf16": [_vp, _vp, _i32, _i32, _f32, _vp],
    "vk_timestep_embedding_f32": [_vp, _vp, _i32, _i32, _f32, _vp],
    "vk_emb_combine": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp],
    "vk_silu_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "vk_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "vk_sampler_prepare": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp],
    "vk_sampler_update": [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp],
    "vk_denoiser_combine": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp],
    "vk_cfg_combine": [_vp, _vp, _vp, _i32, _i32, _vp],
    "vk_euler_step": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp],
    "vk_mask_replace": [_vp, _vp, _vp, _vp, _i32, _i32, _vp],
    "vk_scale_rows": [_vp, _vp, _vp, _i32, _i32, _vp],
    "vk_gaussian_sample": [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp],
    "vk_ensemble_variance_sum": [_vp, _vp, _vp, _i32, _i64, _vp],
    "vk_abi_version": [],
    "vk_act_dtype": [],
}

_libs = {}


class VistaHipError(RuntimeError):
    pass


def load(dtype_name=None):
    """Open the library of a storage type (default: the current one, CURRENT) and declare prototypes. Raises VistaHipError when it is absent."""
    name = CURRENT if dtype_name is None else dtype_name
    lib = _libs.get(name)
    if lib is not None:
        return lib
    path = lib_path(name)
    if not os.path.exists(path):
        raise VistaHipError(
            f"{path} not found: build it with `python -m vista_amd.build` (hipcc --offload-arch=gfx950). "
            "vista_amd has no CPU / eager fallback.")
    lib = C.CDLL(path)
    for fname, argtypes in SIGNATURES.items():
        fn = getattr(lib, fname)  # AttributeError here = ABI mismatch, surfaced loudly
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.vk_abi_version() != ABI_VERSION:
        raise VistaHipError(f"{path} has ABI version {lib.vk_abi_version()}, this package expects {ABI_VERSION}: rebuild it "
                            "(python -m vista_amd.build --force)")
    if lib.vk_act_dtype() != (1 if name == "fp16" else 0):
        raise VistaHipError(f"{path} stores {'fp16' if lib.vk_act_dtype() else 'bf16'} but was opened as the {name} library: "
                            "the library and the host side must agree on the storage type")
    _libs[name] = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise VistaHipError(f"{what} failed with code {rc} (see include/vista_hip.h: -22 bad argument, -5 launch failure)")
