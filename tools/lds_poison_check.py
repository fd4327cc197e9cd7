"""Does any env kernel read an LDS word before writing it?  (run on the GPU box; needs var/liblds_poison.so, see
tools/probes/lds_poison.hip; the register-file poisoner is generated and built by this tool: `--build-only` here first)  For both robots and an env count per kernel variant, the same injected state is stepped after the
LDS of every CU has been filled with different patterns (zeros, NaN bits, hashed garbage): one launch per step, and 3 steps in one
launch of the rollout kernel.  Every result must be bitwise equal to the first."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as ol
from steppingstone_amd.envs import SteppingStoneVecEnv
P = C.CDLL(os.path.join(ROOT, "var", "liblds_poison.so"))
P.lds_poison.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")


def reg_poison_source():
    """the register-file poisoner: one v_mov_b32 per VGPR and one v_accvgpr_write_b32 per AGPR (256 + 256), every register in the
    clobber list -- generated here at run time (not a committed 500-line file)"""
    movs = "".join('    "v_mov_b32 v%d, %%0\\n"\n' % i for i in range(256)) + "".join('    "v_accvgpr_write_b32 a%d, %%0\\n"\n' % i for i in range(256))
    clob = ",".join('"v%d"' % i for i in range(256)) + "," + ",".join('"a%d"' % i for i in range(256))
    return ('#include <hip/hip_runtime.h>\n#include <cstdint>\n'
            '__global__ __launch_bounds__(64) void reg_poison_kernel(uint32_t pat) {\n'
            '  uint32_t p = pat ^ (threadIdx.x * 2654435761u * (pat & 1u));\n  asm volatile(\n' + movs +
            '    :: "s"(__builtin_amdgcn_readfirstlane(p)) : ' + clob + ');\n}\n'
            'extern "C" int reg_poison(uint32_t pat, void* stream) {\n'
            '  hipLaunchKernelGGL(reg_poison_kernel, dim3(8192), dim3(64), 0, (hipStream_t)stream, pat);\n'
            '  return (int)hipGetLastError();\n}\n')


def build_reg_poison():
    import subprocess
    so, src = os.path.join(ROOT, "var", "libreg_poison.so"), os.path.join(ROOT, "var", "reg_poison.hip")
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        open(src, "w").write(reg_poison_source())
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-shared", "-fPIC", src, "-o", so])
    return so


if "--build-only" in sys.argv:      # in the build container (hipcc cross-compiles); the .so travels with the snapshot
    print(build_reg_poison())
    sys.exit(0)
RG = C.CDLL(build_reg_poison())
if RG:
    RG.reg_poison.argtypes = [C.c_uint32, C.c_void_p]
PATTERNS = [(0, 0), (0x7fc00000, 0), (0xffffffff, 0), (0x3f800000, 0), (0x12345678, 2654435761), (0x7f800000, 0), (0xdeadbeef, 40503)]
bad = 0
for env_id, kind in (("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike")):
    for n in (1, 3, 33, 700, 4096, 12000, 40000):
        o = ol.OracleEnv(kind, min(n, 700), seed=2); o.set_curriculum(5); o.reset()
        for t in range(12):
            o.step(o.random_actions(t))
        st = np.resize(o.get_state().astype(np.float32), (n, ol.STATE_DIM)) if n > 700 else o.get_state().astype(np.float32)
        g = SteppingStoneVecEnv(env_id, n, seed=2, device="cuda:0", return_numpy=False)
        g.update_curriculum(5); g.reset()
        std = torch.as_tensor(st).cuda()
        act = g.random_actions(50)
        for mode in ("step", "rollout3"):
            ref = None
            for pat, mix in PATTERNS:
                g.set_state(std)
                P.lds_poison(pat, mix, sink.data_ptr(), None)
                if RG:
                    RG.reg_poison(pat, None)      # ... and the register files (reg_poison_source above)
                if mode == "step":
                    ob, rw, dn, _ = g.step(act)
                else:
                    ob, rw, dn = g.rollout_random(3, t0=7, steps_per_launch=3)
                out = torch.cat([ob.flatten(), rw.flatten(), dn.flatten().float(), g.get_state().flatten(), g._info.flatten().float()]).clone()
                if ref is None:
                    ref = out
                elif not torch.equal(out.view(torch.int32), ref.view(torch.int32)):
                    bad += 1
                    idx = torch.nonzero(out.view(torch.int32) != ref.view(torch.int32)).flatten()
                    print("DIFF", env_id, n, mode, "pattern %08x/%d" % (pat, mix), "words differing:", idx.numel(), "first:", idx[:6].tolist(), flush=True)
        g.close()
        print(env_id, n, "checked", flush=True)
print("pattern-dependent results:", bad)
