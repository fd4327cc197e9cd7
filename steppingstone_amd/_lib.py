"""ctypes binding of libsteppingstone.so (include/steppingstone.h).  No torch types cross this boundary: every
buffer is a raw device pointer (int).  Loading fails loudly when the HIP library is missing -- there is no CPU
fallback anywhere in this package."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# STEPPINGSTONE_LIB overrides the path (A/B builds of the same HIP source during tuning); it is still a HIP build.
LIB_PATH = os.environ.get("STEPPINGSTONE_LIB") or os.path.join(PKG, "lib", "libsteppingstone.so")

OBS_DIM, ACT_DIM, GRID, NCELL, NUM_STONES, STATE_DIM, MAX_EPISODE_STEPS = 60, 21, 11, 121, 20, 186, 1000
INFO_WORDS = 6            # ss_info: ep_ret, ep_len, bad_transition, steps_reached, update_terrain, ep_ret_lo
ABI_VERSION = 4           # include/steppingstone.h SS_ABI_VERSION
WALKER3D, MIKE = 0, 1

SYMBOLS = [
    "ss_create", "ss_destroy", "ss_last_error", "ss_reset", "ss_step", "ss_step_packed", "ss_rollout_random", "ss_random_actions",
    "ss_set_curriculum", "ss_set_specialist", "ss_set_sample_prob", "ss_set_mirror", "ss_set_power", "ss_set_auto_reset",
    "ss_create_temp_states", "ss_get_mirror_indices", "ss_get_state", "ss_set_state", "ss_get_obs", "ss_num_envs",
    "ss_version", "ss_set_sample_prob_device", "ss_debug_calib_copy", "ss_debug_phase_cycles",
    "ss_peer_alloc", "ss_peer_free", "ss_peer_ipc_handle", "ss_peer_ipc_open", "ss_peer_ipc_close", "ss_peer_connect",
    "ss_step_packed_peers", "ss_peer_wait", "ss_peer_error", "ss_rollout_random_packed", "ss_debug_set_id_mask",
]


class SteppingStoneError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SteppingStoneError(
            "libsteppingstone.so is missing (%s): build it with `python -m steppingstone_amd.build`; "
            "this package has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, u64, i64, f32 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64, C.c_float
    lib.ss_create.argtypes = [C.POINTER(vp), C.c_int, i32, C.c_int, u64, i64]
    lib.ss_destroy.argtypes = [vp]
    lib.ss_destroy.restype = None
    lib.ss_last_error.restype = C.c_char_p
    lib.ss_reset.argtypes = [vp, vp, vp]
    lib.ss_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.ss_rollout_random.argtypes = [vp, i32, i32, u64, vp, vp, vp, vp, vp]
    lib.ss_step_packed.argtypes = [vp, vp, C.c_int, u64, vp, vp, vp]
    lib.ss_rollout_random_packed.argtypes = [vp, i32, u64, vp, vp, vp]
    lib.ss_random_actions.argtypes = [vp, u64, vp, vp]
    lib.ss_set_curriculum.argtypes = [vp, i32]
    lib.ss_set_specialist.argtypes = [vp, i32]
    lib.ss_set_sample_prob.argtypes = [vp, vp, C.c_int]
    lib.ss_set_sample_prob_device.argtypes = [vp, vp, C.c_int, vp]
    lib.ss_peer_alloc.argtypes = [C.POINTER(vp), u64]
    lib.ss_peer_free.argtypes = [vp]
    lib.ss_peer_ipc_handle.argtypes = [vp, vp]
    lib.ss_peer_ipc_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.ss_peer_ipc_close.argtypes = [vp]
    lib.ss_peer_connect.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.ss_step_packed_peers.argtypes = [vp, vp, C.c_int, u64, i32, C.c_uint32, vp, vp, vp]
    lib.ss_peer_wait.argtypes = [vp, i32, C.c_uint32, vp]
    lib.ss_peer_error.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.ss_debug_calib_copy.argtypes = [vp, vp, u64, vp]
    lib.ss_debug_phase_cycles.argtypes = [vp, vp, C.c_int]
    lib.ss_debug_set_id_mask.argtypes = [vp, C.c_uint32]
    lib.ss_set_mirror.argtypes = [vp, i32]
    lib.ss_set_power.argtypes = [vp, f32]
    lib.ss_set_auto_reset.argtypes = [vp, i32]
    lib.ss_create_temp_states.argtypes = [vp, vp, vp]
    lib.ss_get_mirror_indices.argtypes = [C.c_int, vp, vp]
    lib.ss_get_state.argtypes = [vp, vp, vp]
    lib.ss_set_state.argtypes = [vp, vp, vp]
    lib.ss_get_obs.argtypes = [vp, vp, vp]
    lib.ss_num_envs.argtypes = [vp]
    lib.ss_num_envs.restype = i32
    lib.ss_version.restype = C.c_int
    if lib.ss_version() != ABI_VERSION:
        # the argument lists and struct sizes changed between versions (2: steps_per_launch in ss_rollout_random; 3: ss_info has 6
        # words, the packed state 186): a stale library would be called with the wrong layout and no error
        raise SteppingStoneError("%s has ABI version %d, this binding needs %d: rebuild it with `python -m steppingstone_amd.build --force`"
                                 % (LIB_PATH, lib.ss_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().ss_last_error()
        raise SteppingStoneError("libsteppingstone error %d: %s" % (rc, msg.decode() if msg else "?"))


def mirror_indices(kind=WALKER3D):
    """The six index lists of env.unwrapped.get_mirror_indices() (playground/train.py:160)."""
    import numpy as np
    lib = load()
    buf = np.zeros(2 * (OBS_DIM + ACT_DIM), np.int32)
    lens = np.zeros(6, np.int32)
    check(lib.ss_get_mirror_indices(kind, buf.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p)))
    out, o = [], 0
    for n in lens:
        out.append(buf[o:o + n].astype(np.int64))
        o += n
    return out
