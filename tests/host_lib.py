"""Builds and binds tests/host/host_harness.cpp: the product's kernel source compiled for the CPU (hipcc
--cuda-host-only).  TEST INFRASTRUCTURE for the GPU-less container; never imported by steppingstone_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "tests", "host")
LIB = os.path.join(HOST_DIR, "libss_host.so")
SRC = os.path.join(HOST_DIR, "host_harness.cpp")
CSRC = os.path.join(ROOT, "steppingstone_amd", "csrc")

INFO_DTYPE = np.dtype([("ep_ret", "f4"), ("ep_len", "f4"), ("bad_transition", "i4"), ("steps_reached", "i4"),
                       ("update_terrain", "i4"), ("ep_ret_lo", "f4")])


def build():
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    subprocess.check_call(["hipcc", "--cuda-host-only", "-x", "hip", "-O1", "-std=c++17", "-fPIC", "-shared",
                           "-fno-signed-zeros", "-fno-math-errno", "-DSS_HOST_HARNESS", "-pthread", SRC, "-o", LIB], stderr=subprocess.DEVNULL)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        vp = C.c_void_p
        _lib.hh_step.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def step(kind, packed, act, seed=0, curriculum=0, prob=None):
    lib = load()
    packed = np.ascontiguousarray(packed, np.float32)
    n = packed.shape[0]
    act = np.ascontiguousarray(act, np.float32).reshape(n, 21)
    out = np.zeros_like(packed)
    obs = np.zeros((n, 60), np.float32)
    rew = np.zeros(n, np.float32)
    done = np.zeros(n, np.uint8)
    info = np.zeros(n, INFO_DTYPE)
    pp = None if prob is None else _p(np.ascontiguousarray(prob, np.float64))
    lib.hh_step(int(kind), n, int(seed), int(curriculum), pp, _p(packed), _p(act), _p(out), _p(obs), _p(rew),
                _p(done), _p(info))
    return out, obs, rew, done, info
