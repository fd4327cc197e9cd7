"""Build libsteppingstone.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the GPU-less build container; the resulting .so travels to the
GPU box with the repository snapshot.  `python -m steppingstone_amd.build [--force]`.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libsteppingstone.so")
SOURCES = ["ss_api.hip"]
HEADERS = ["ss_math.hpp", "ss_dynamics.hpp", "ss_kernels.hpp", "ss_model_tables.hpp",
           os.path.join("..", "..", "include", "steppingstone.h")]
# IEEE -O3 without the SLP vectorizer.  Measured in round 1 on gfx950 / ROCm 7.2 (tools/gpu_debug2.py):
#   * with SLP vectorisation (packed v_pk_fma_f32 / v_pk_mul_f32) the 28k-instruction step kernel is MISCOMPILED at
#     -O2/-O3 (wrong and run-to-run varying results); -O1 and -O3 -fno-slp-vectorize match the oracle to 2e-7;
#   * -fno-signed-zeros triggers the same failure even at plain -O3;
#   * packed f32 VALU is not a throughput win on gfx950 anyway (MI355X_MICROARCH.md, per-instruction constants).
# No fast-math family flags: -ffinite-math-only would delete the non-finite guard of PHYSICS.md 4.8.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize"]

def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
