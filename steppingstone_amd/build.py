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
SOURCES = ["ss_api.hip", "ss_rollout3.hip"]
# ss_rollout3.hip (the three-helper rollout kernel) is compiled without the max-ILP scheduling strategy: see its header
NO_MAX_ILP = {"ss_rollout3.hip"}
HEADERS = ["ss_math.hpp", "ss_pair.hpp", "ss_dynamics.hpp", "ss_kernels.hpp", "ss_model_tables.hpp",
           os.path.join("..", "..", "include", "steppingstone.h")]
# -O3 without the SLP vectorizer, signed zeros not honoured.  Measured on gfx950 / ROCm 7.2:
#   * SLP vectorisation (packed v_pk_fma_f32 / v_pk_mul_f32) miscompiled the 1-env-per-lane kernel of v1-v3 (wrong,
#     run-to-run varying results next to KBs of scratch); the current kernel passes parity with it but is 30 %
#     slower (aligned register pairs -> 536 B/lane of scratch): -fno-slp-vectorize stays;
#   * -fno-signed-zeros lets the compiler drop the "+ 0" of zero-initialised accumulators and fold x*y+0 into a
#     multiply: 0.1068 -> 0.1026 ms/step, all GPU parity tests green (it had triggered the v1-v3 miscompile too, so
#     it is re-validated by the full GPU suite on every build change);
#   * -fassociative-math would give another 1.5 % but re-orders the sums of the spec: not used.
# No other fast-math flags: -ffinite-math-only would delete the non-finite guard of PHYSICS.md 4.8.
#   * -ffp-contract=on (fuse a*b+c only inside one source expression) instead of HIP's default "fast": the kernel
#     variants (with / without helper wavefronts) then round identically, so results do not depend on batch size or on
#     the number of GPUs a batch is sharded over (bitwise; tested).  Costs 3 % on the plain variant, nothing on the
#     helper variant.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-fno-signed-zeros",
         "-ffp-contract=on"]
# scheduling only (no effect on values): the max-ILP machine scheduler is worth 1.4 % on the step kernel.  It is an
# -mllvm option, so build() falls back to the plain flags if a compiler does not know it.
OPTIONAL_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]

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


FUZZ_LIB = os.path.join(LIBDIR, "libsteppingstone_fuzz.so")


def _compile_lib(out, extra_flags, tag, verbose):
    """Both translation units in parallel, then the link.  A compiler that does not know the optional -mllvm flag gets the plain
    flags on a second pass; that pass keeps its diagnostics (a genuine compile error must be readable)."""
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src, with_optional, quiet):
        obj = os.path.join(objdir, src.replace(".hip", tag + ".o"))
        cmd = [hipcc()] + FLAGS + extra_flags + (OPTIONAL_FLAGS if with_optional else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        return obj, subprocess.Popen(cmd, stderr=subprocess.DEVNULL if quiet else None)

    for with_optional in (True, False):
        jobs = [compile_one(src, with_optional and src not in NO_MAX_ILP, quiet=with_optional and not verbose) for src in SOURCES]
        rcs = [p.wait() for _, p in jobs]
        if all(rc == 0 for rc in rcs):
            break
        if not with_optional:
            raise subprocess.CalledProcessError(max(rcs), "hipcc -c (diagnostics above)")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in jobs] + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def _stale(lib):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_fuzz(force=False, verbose=False):
    """The -DSS_FUZZ_SCHED build of the same sources (lib/libsteppingstone_fuzz.so): every wavefront sleeps a pseudo-random time at the
    start of each barrier window.  TEST BUILD -- tests/test_gpu_sched_fuzz.py requires its results to be bit-identical to the
    product library's; nothing in the package loads it."""
    if not force and not _stale(FUZZ_LIB):
        return FUZZ_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    return _compile_lib(FUZZ_LIB, ["-DSS_FUZZ_SCHED"], "_fuzz", verbose)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    return _compile_lib(LIB, [], "", verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    if "--fuzz" in sys.argv:
        build_fuzz(force="--force" in sys.argv, verbose=True)
    print(LIB)
