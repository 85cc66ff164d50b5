"""Build libvista_hip.so (gfx950) in-tree with hipcc. No torch extension machinery: the product boundary is a
plain C-ABI shared library (include/vista_hip.h) that the Python host side binds with ctypes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIBDIR, "libvista_hip.so")
# per-file flags. ff_fused.hip: the SLP vectorizer packs the GEGLU arithmetic into v_pk_fma_f32 / v_pk_mul_f32 and pays for the operand
# pairing with a v_mov per packed instruction -- a third more VALU instructions in a kernel whose in-projection waves are VALU-issue bound
# attention.hip: the pipelined spatial kernel's row-sum adds must stay single v_add_f32 (packed they are gathered at the end of a unit and keep
# its sixteen exponentials live: spills in a loop that is 256 registers wide)
EXTRA_FLAGS = {"ff_fused.hip": ["-fno-slp-vectorize"], "attention.hip": ["-fno-slp-vectorize"]}
SOURCES = ["gemm.hip", "gemm_pipe.hip", "gemm_pipe2.hip", "gemm_stream.hip", "gemm_fp8.hip", "ff_fused.hip", "attention.hip", "norm.hip", "elementwise.hip"]


# A/B builds: VISTA_BUILD_PIPE4=1 adds the four-wave build of the pipelined GEMM (gemm_pipe4.hip = gemm_pipe.hip with -DPIPE_W4; VISTA_GEMM_PIPE4=1|2|3 then routes
# launches to it; =2 in VISTA_BUILD_PIPE4 also sets -DPIPE_W4_SPREAD=1). Measured 3-20 % behind the eight-wave kernel (profiles/r06_gemm_pipe4.txt): not in the default library.
if os.environ.get("VISTA_BUILD_PIPE4", "0") != "0":
    SOURCES.insert(2, "gemm_pipe4.hip")
    if os.environ["VISTA_BUILD_PIPE4"] == "2":
        EXTRA_FLAGS["gemm_pipe4.hip"] = ["-DPIPE_W4_SPREAD=1"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


# the two builds of the same sources: bf16 storage (the default; BASELINE config 2 names bf16) and IEEE fp16 storage (-DVK_F16=1: the reference's
# autocast width, csrc/common.h). _lib.load() opens the one VISTA_ACT_DTYPE names.
VARIANTS = {"bf16": ("libvista_hip.so", "", []), "fp16": ("libvista_hip_f16.so", "_f16", ["-DVK_F16=1"])}
LIB_F16 = os.path.join(LIBDIR, VARIANTS["fp16"][0])


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "vista_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, variants=("bf16", "fp16")):
    """Compile every HIP source for gfx950 and link the requested variants (default: both). Returns the bf16 library's path."""
    os.makedirs(LIBDIR, exist_ok=True)
    procs, links = [], []
    for v in variants:
        libname, suffix, defs = VARIANTS[v]
        lib = os.path.join(LIBDIR, libname)
        if not force and not needs_build(lib):
            continue
        objs = []
        for s in SOURCES:
            o = os.path.join(LIBDIR, s.replace(".hip", suffix + ".o"))
            cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"),
                   "-I" + CSRC, "-c", os.path.join(CSRC, s), "-o", o] + EXTRA_FLAGS.get(s, []) + defs
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, s + suffix))
            objs.append(o)
        links.append((lib, objs))
    # at most eight compilers at a time (this container has 8 CPUs; gemm.hip alone takes ~2 minutes)
    running = []
    for cmd, name in procs:
        while len(running) >= 8:
            p, n = running.pop(0)
            if p.wait() != 0:
                raise RuntimeError(f"hipcc failed on {n}")
        running.append((subprocess.Popen(cmd), name))
    for p, n in running:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {n}")
    for lib, objs in links:
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=LIBDIR)   # (the offload bundler drops per-target intermediates into the cwd: keep them in lib/)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
