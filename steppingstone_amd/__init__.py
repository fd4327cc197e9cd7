"""steppingstone_amd -- MI355X-native vectorised stepping-stone locomotion environments.

Drop-in for the `env.step()` hot path of belinghy/SteppingStone (Walker3DStepperEnv-v0 / MikeStepperEnv-v0):
hand-written HIP kernels for gfx950 behind a C ABI (include/steppingstone.h), with a thin Python mirror of the
reference's make_env / make_vec_envs / VecEnv protocol (steppingstone_amd.envs).  No CPU fallback.
"""
from ._lib import ACT_DIM, OBS_DIM, SteppingStoneError  # noqa: F401


def __getattr__(name):
    # envs needs torch; keep `import steppingstone_amd` light for the build / model tooling
    if name in ("make_env", "make_vec_envs", "SteppingStoneVecEnv", "SteppingStoneEnv", "HipBackend", "Box"):
        from . import envs
        return getattr(envs, name)
    raise AttributeError(name)
