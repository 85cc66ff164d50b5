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
SOURCES = ["gemm.hip", "gemm_pipe.hip", "gemm_stream.hip", "gemm_fp8.hip", "ff_fused.hip", "attention.hip", "norm.hip", "elementwise.hip"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(REPO, "include", "vista_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"),
               "-I" + CSRC, "-c", os.path.join(CSRC, s), "-o", o] + EXTRA_FLAGS.get(s, [])
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), s))
        objs.append(o)
    for p, s in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
