"""The C-ABI library loads without a GPU, exports every symbol include/steppingstone.h declares, and the product
path fails loudly (no CPU fallback) when no GPU is visible.  CPU only."""
import ctypes as C
import os
import re

import pytest

from steppingstone_amd import SteppingStoneError, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "steppingstone.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libsteppingstone.so does not export %s" % n
    assert set(_lib.SYMBOLS) == set(names)


def test_constants_match_header():
    src = open(os.path.join(ROOT, "include", "steppingstone.h")).read()
    for name, val in (("SS_OBS_DIM", _lib.OBS_DIM), ("SS_ACT_DIM", _lib.ACT_DIM), ("SS_NCELL", _lib.NCELL),
                      ("SS_STATE_DIM", _lib.STATE_DIM), ("SS_NUM_STONES", _lib.NUM_STONES),
                      ("SS_MAX_EPISODE_STEPS", _lib.MAX_EPISODE_STEPS)):
        assert re.search(r"#define\s+%s\s+%d\b" % (name, val), src), name


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.ss_create(C.byref(h), 0, 64, 0, 0, 0)
    assert rc == -3 and not h.value
    assert b"no CPU fallback" in lib.ss_last_error()
    from steppingstone_amd.envs import SteppingStoneVecEnv, make_env
    with pytest.raises(SteppingStoneError):
        SteppingStoneVecEnv("Walker3DStepperEnv-v0", 8)
    with pytest.raises(SteppingStoneError):
        make_env("MikeStepperEnv-v0")


def test_invalid_arguments_are_rejected():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.ss_create(C.byref(h), 7, 64, 0, 0, 0) == -1          # unknown robot kind
    assert lib.ss_create(C.byref(h), 0, 0, 0, 0, 0) == -1           # no envs
    assert lib.ss_step(None, None, None, None, None, None, None) == -1
    assert lib.ss_set_curriculum(None, 3) == -1


def test_mirror_indices_are_a_permutation_free_partition():
    idx = _lib.mirror_indices()
    neg_o, r_o, l_o, neg_a, r_a, l_a = idx
    assert len(r_o) == len(l_o) and len(r_a) == len(l_a) == 9
    assert not (set(r_o) & set(l_o)) and not (set(r_a) & set(l_a))
    assert max(neg_o) < 60 and max(neg_a) < 21
    # swapped joints are the same in the action and in both joint blocks of the observation
    assert [i - 6 for i in r_o[:9]] == list(r_a) and [i - 27 for i in r_o[9:18]] == list(r_a)


def test_header_is_plain_c_and_usable_without_python(tmp_path):
    """include/steppingstone.h compiles as C99 and a dlopen client (tests/host/abi_c_client.c) reaches the library;
    without a GPU ss_create answers SS_ERR_NO_DEVICE and an error message (no CPU fallback)."""
    import shutil, subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from steppingstone_amd import build
    lib = build.build()
    exe = str(tmp_path / "abi_c_client")
    src = os.path.join(ROOT, "tests", "host", "abi_c_client.c")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", src, "-o", exe, "-ldl"])
    out = subprocess.run([exe, lib], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "version 4" in out.stdout and "obs_dim 60 act_dim 21" in out.stdout
    import torch
    if not torch.cuda.is_available():
        assert "create rc -3" in out.stdout and "no CPU fallback" in out.stdout
