#!/usr/bin/env python3
"""Builds extras/fused_learner/lib/libsslearner.so (hipcc, gfx950).  NOT part of the product build: __graft_entry__.build() and
`python -m steppingstone_amd.build` never come here."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from steppingstone_amd.build import hipcc  # noqa: E402  (the same compiler resolution as the product build)

LIB = os.path.join(HERE, "lib", "libsslearner.so")


def build(force=False, verbose=True):
    deps = [os.path.join(HERE, "ss_learner.hip"), os.path.join(HERE, "steppingstone_learner.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", HERE, deps[0], "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
