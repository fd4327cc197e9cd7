"""Static instruction counts of the three-helper rollout kernel between its barriers (hipcc -S of that one instantiation; no
GPU needed): which wavefront issues how many VALU instructions in which window of a substep -- the numbers behind DESIGN.md
section 4.1's window table and section 7's "what is left".  The compiler lays the code out as: prologue | helper wavefronts
(#0->#0b cos/sin, #0b->#1 kinematics / detection / rows / bias / extra, #1->#2 operators part A, #2->#3 part B) | main wavefront
(#0->#0b joint torques, #0b->#1 pass 1 + leg/arm half of pass 2, #1->#2 spine + base factorisation, #2->#3 pass 3 + foot twist,
#3->end: rows y = Lambda w, the PGS (its sweep loop appears ONCE here and runs 7 + 1 times), response of the tree, integration and
the control step's epilogue) | tail.  usage: python tools/isa_windows.py [extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-signed-zeros", "-ffp-contract=on"]   # as build.py compiles ss_rollout3.hip
KERNEL = "_ZN2ss21rollout_kernel_helpedINS_13ModelWalker3DELi3EEEvNS_6ParamsENS_6StepIOE"


def cls(op):
    if op.startswith("v_pk_"):
        return "packed f32"
    if re.match(r"v_(fma|fmac|mul|add|sub|subrev|fmamk|fmaak|mac)_f32", op):
        return "scalar f32"
    if op.startswith("v_accvgpr"):
        return "agpr moves"
    if op.startswith("v_mov"):
        return "v_mov"
    if op.startswith("v_"):
        return "other VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "scalar ALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "global memory"
    return "other"


def main():
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, "one.hip"), os.path.join(d, "one.s")
        open(src, "w").write('#include <hip/hip_runtime.h>\n#include "%s"\ntemplate __global__ void ss::rollout_kernel_helped<ss::ModelWalker3D, 3>'
                             '(ss::Params, ss::StepIO);\n' % os.path.join(ROOT, "steppingstone_amd", "csrc", "ss_kernels.hpp"))
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + ["-S", "--cuda-device-only", src, "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    a = next(i for i, l in enumerate(text) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(text)) if "s_endpgm" in text[i])
    body = text[a:b + 1]
    res = {}
    for l in text[b:]:
        m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|codeLenInByte): (\d+)", l)
        if m and m.group(1) not in res:
            res[m.group(1)] = int(m.group(2))
        if len(res) == 4:
            break
    print("kernel rollout_kernel_helped<Walker3D,3>: %s" % res)
    bars = [i for i, l in enumerate(body) if re.match(r"\s+s_barrier", l)]
    marks = [0] + bars + [len(body)]
    cols = ["packed f32", "scalar f32", "agpr moves", "v_mov", "other VALU", "LDS", "scalar ALU", "global memory", "scratch", "s_waitcnt"]
    print("%-14s %6s | %s" % ("ISA lines", "VALU", " ".join("%13s" % c for c in cols)))
    for lo, hi in zip(marks[:-1], marks[1:]):
        c = collections.Counter()
        for l in body[lo:hi]:
            m = re.match(r"\s+([a-z_0-9]+)", l)
            if m and not l.strip().startswith((".", ";")):
                c[cls(m.group(1))] += 1
        valu = sum(c[k] for k in cols[:5])
        if valu or c["LDS"]:
            print("%6d-%-7d %6d | %s" % (lo, hi, valu, " ".join("%13d" % c[k] for k in cols)))


if __name__ == "__main__":
    main()
