"""Builds tests/host/first_launch_check (plain C over the C ABI, links libamdhip64 only for hipMalloc / hipMemcpy of its buffers).
TEST INFRASTRUCTURE: used by tests/test_gpu_first_launch.py; __graft_entry__.build() tries it and carries on without it.
The ROCm root comes from ROCM_PATH / the resolved hipcc, not from a hard-coded /opt/rocm."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "first_launch_check.c")
EXE = os.path.join(ROOT, "tests", "host", "first_launch_check")


def rocm_root():
    if os.environ.get("ROCM_PATH"):
        return os.environ["ROCM_PATH"]
    sys.path.insert(0, ROOT)
    from steppingstone_amd.build import hipcc
    exe = hipcc()
    exe = exe if os.path.isabs(exe) else (shutil.which(exe) or "/opt/rocm/bin/hipcc")
    return os.path.dirname(os.path.dirname(os.path.realpath(exe)))


def build_client():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        root = rocm_root()
        subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-Wall", SRC, "-o", EXE, "-I" + os.path.join(root, "include"),
                               "-L" + os.path.join(root, "lib"), "-lamdhip64", "-ldl", "-Wl,-rpath," + os.path.join(root, "lib")])
    return EXE


if __name__ == "__main__":
    print(build_client())
