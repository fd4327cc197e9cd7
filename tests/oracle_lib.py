"""ctypes binding of the CPU oracle (oracle/ss_oracle.c).  TEST INFRASTRUCTURE: only tests/, smoke() and the
cpu_baseline leg of bench.py may import this module; nothing under steppingstone_amd/ does."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
OBS_DIM, ACT_DIM, STATE_DIM, NCELL = 60, 21, 186, 121
KIND = {"walker3d": 0, "mike": 1}


NEAR_CAP = 6
MAX_DECISIONS = 400      # sso_max_decisions(): 4 substeps x (42 limit switches + 48 contact predicates) + 26 reward / done tests


def tap_dtype(real):
    return np.dtype([("Li", real, (12, 12)), ("V0", real, (12,)), ("W", real, (8, 3, 6)), ("bn", real, (8,)), ("lam", real, (8, 3)),
                     ("nrm", real, (8, 3)), ("pen", real, (8,)), ("qdf", real, (21,)), ("v0f", real, (6,)), ("dqd", real, (21,)),
                     ("dv0", real, (6,)), ("active", np.int32, (8,)), ("stone", np.int32, (8,))])


class Info(C.Structure):
    _fields_ = [("ep_ret", C.c_float), ("ep_len", C.c_float), ("bad_transition", C.c_int32),
                ("steps_reached", C.c_int32), ("update_terrain", C.c_int32), ("ep_ret_lo", C.c_float)]


INFO_DTYPE = np.dtype([("ep_ret", "f4"), ("ep_len", "f4"), ("bad_transition", "i4"), ("steps_reached", "i4"),
                       ("update_terrain", "i4"), ("ep_ret_lo", "f4")])


def build():
    """(Re)build the oracle libraries with gcc if missing or stale."""
    src = os.path.join(ORACLE_DIR, "ss_oracle.c")
    for prec in ("f32", "f64"):
        lib = os.path.join(ORACLE_DIR, "lib", "libss_oracle_%s.so" % prec)
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
            break


_libs = {}


def load(prec="f32"):
    if prec in _libs:
        return _libs[prec]
    build()
    # SS_ORACLE_LIB_F32 / _F64: another build of the same source (bench.py's cpu_baseline leg points its worker processes
    # at the -O3 -march=native build it made on the box)
    lib = C.CDLL(os.environ.get("SS_ORACLE_LIB_%s" % prec.upper()) or os.path.join(ORACLE_DIR, "lib", "libss_oracle_%s.so" % prec))
    vp, i32, u64, i64, dbl = C.c_void_p, C.c_int, C.c_uint64, C.c_int64, C.c_double
    lib.sso_create.restype = vp
    lib.sso_create.argtypes = [i32, i32, u64, i64]
    lib.sso_destroy.argtypes = [vp]
    lib.sso_reset.argtypes = [vp, vp]
    lib.sso_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.sso_step_margins.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.sso_step_near.argtypes = [vp, vp, vp, vp, vp, vp, vp, dbl, vp, vp, i32]
    lib.sso_step_forced.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32]
    lib.sso_step_ex.argtypes = [vp, vp, vp, vp, vp, vp, vp, dbl, vp, vp, vp, vp, i32, vp, vp, i32]
    lib.sso_debug_contact.argtypes = [vp, i32, vp, vp]
    lib.sso_debug_contact_after.argtypes = [vp, i32, vp, i32, vp]
    lib.sso_set_curriculum.argtypes = [vp, i32]
    lib.sso_set_specialist.argtypes = [vp, i32]
    lib.sso_set_sample_prob.argtypes = [vp, vp, i32]
    lib.sso_set_power.argtypes = [vp, dbl]
    lib.sso_set_auto_reset.argtypes = [vp, i32]
    lib.sso_create_temp_states.argtypes = [vp, vp]
    lib.sso_get_state.argtypes = [vp, vp]
    lib.sso_set_state.argtypes = [vp, vp]
    lib.sso_get_obs.argtypes = [vp, vp]
    lib.sso_random_actions.argtypes = [vp, u64, vp]
    lib.sso_philox.argtypes = [vp, vp, vp]
    lib.sso_debug_aba.argtypes = [i32, vp, vp, vp, vp]
    lib.sso_debug_substeps.argtypes = [vp, i32, vp, i32, vp]
    lib.sso_debug_fk.argtypes = [i32, vp, vp, vp]
    _libs[prec] = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """Batched oracle env with the same call surface as the C-ABI handle (numpy in/out)."""

    def __init__(self, kind="walker3d", num_envs=1, seed=0, env_offset=0, prec="f32"):
        self.lib = load(prec)
        self.real = np.float32 if prec == "f32" else np.float64
        self.kind = KIND[kind] if isinstance(kind, str) else int(kind)
        self.n = int(num_envs)
        self.h = self.lib.sso_create(self.kind, self.n, int(seed), int(env_offset))

    def close(self):
        if self.h:
            self.lib.sso_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        obs = np.zeros((self.n, OBS_DIM), np.float32)
        self.lib.sso_reset(self.h, _p(obs))
        return obs

    def step(self, act):
        act = np.ascontiguousarray(act, np.float32).reshape(self.n, ACT_DIM)
        obs = np.zeros((self.n, OBS_DIM), np.float32)
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        info = np.zeros(self.n, INFO_DTYPE)
        self.lib.sso_step(self.h, _p(act), _p(obs), _p(rew), _p(done), _p(info))
        return obs, rew, done, info

    def step_margins(self, act):
        """step() that also returns margins [N,2]: the distance of the closest discrete decision of this control step
        to its threshold -- column 0: decisions that change the state (contact set, stone choice, joint-limit
        switches; metres / radians), column 1: decisions that only enter reward / done."""
        act = np.ascontiguousarray(act, np.float32).reshape(self.n, ACT_DIM)
        obs = np.zeros((self.n, OBS_DIM), np.float32)
        rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, np.uint8)
        info = np.zeros(self.n, INFO_DTYPE)
        margins = np.zeros((self.n, 2), self.real)
        self.lib.sso_step_margins(self.h, _p(act), _p(obs), _p(rew), _p(done), _p(info), _p(margins))
        return obs, rew, done, info, margins

    def _out(self):
        return (np.zeros((self.n, OBS_DIM), np.float32), np.zeros(self.n, np.float32), np.zeros(self.n, np.uint8),
                np.zeros(self.n, INFO_DTYPE))

    def step_near(self, act, tol=1e-5, cap=NEAR_CAP):
        """step_margins() that also lists the decisions within `tol` of their threshold: near [N,cap] (indices of the
        decision sites in visiting order, -1 padded) and nnear [N] (their true number; > cap means the list is cut)."""
        act = np.ascontiguousarray(act, np.float32).reshape(self.n, ACT_DIM)
        obs, rew, done, info = self._out()
        margins = np.zeros((self.n, 2), self.real)
        near = np.full((self.n, cap), -1, np.int32)
        nnear = np.zeros(self.n, np.int32)
        self.lib.sso_step_near(self.h, _p(act), _p(obs), _p(rew), _p(done), _p(info), _p(margins), float(tol), _p(near),
                               _p(nnear), int(cap))
        return obs, rew, done, info, margins, near, nnear

    def step_forced(self, act, force, nforce):
        """step() with, per env, the outcome of the decisions force[e, :nforce[e]] inverted (the other branch of a
        near-threshold decision)."""
        act = np.ascontiguousarray(act, np.float32).reshape(self.n, ACT_DIM)
        force = np.ascontiguousarray(force, np.int32).reshape(self.n, -1)
        nforce = np.ascontiguousarray(nforce, np.int32).reshape(self.n)
        obs, rew, done, info = self._out()
        self.lib.sso_step_forced(self.h, _p(act), _p(obs), _p(rew), _p(done), _p(info), _p(force), _p(nforce), force.shape[1])
        return obs, rew, done, info

    def step_ex(self, act, tol=None, cap=NEAR_CAP, force=None, nforce=None, record=False, replay=None):
        """The general step: returns a dict with obs / rew / done / info and, on request, `margins` [N,2] + `near`
        [N,cap] + `nnear` [N] (tol given: decisions within tol of their threshold), `trace` [N,MAX_DECISIONS] uint8
        (record=True: the outcome of every decision).  force / nforce invert the listed decisions; replay (a trace)
        freezes every decision to the recorded outcome."""
        act = np.ascontiguousarray(act, np.float32).reshape(self.n, ACT_DIM)
        obs, rew, done, info = self._out()
        out = dict(obs=obs, rew=rew, done=done, info=info)
        margins = near = nnear = trace = None
        if tol is not None:
            margins = out["margins"] = np.zeros((self.n, 2), self.real)
            near = out["near"] = np.full((self.n, cap), -1, np.int32)
            nnear = out["nnear"] = np.zeros(self.n, np.int32)
        if force is not None:
            force = np.ascontiguousarray(force, np.int32).reshape(self.n, cap)
            nforce = np.ascontiguousarray(nforce, np.int32).reshape(self.n)
        if record:
            trace = out["trace"] = np.zeros((self.n, MAX_DECISIONS), np.uint8)
        if replay is not None:
            replay = np.ascontiguousarray(replay, np.uint8).reshape(self.n, MAX_DECISIONS)
        nul = lambda a: _p(a) if a is not None else None
        self.lib.sso_step_ex(self.h, _p(act), _p(obs), _p(rew), _p(done), _p(info), nul(margins), float(tol or 0.0), nul(near),
                             nul(nnear), nul(force), nul(nforce), int(cap), nul(trace), nul(replay), MAX_DECISIONS)
        return out

    def debug_contact(self, e, tau, prior=0):
        """ONE substep of env e under fixed motor torques (the env's state advances); returns the contact stage's
        intermediate quantities as a dict of arrays (oracle/ss_oracle.c: contact_tap).  prior: that many substeps of the same
        control step run first, so that the tapped one is warm-started from them (PHYSICS.md 3.4)."""
        tau = np.ascontiguousarray(tau, self.real)
        tap = np.zeros(1, tap_dtype(self.real))
        assert tap.nbytes == self.lib.sso_tap_size()
        self.lib.sso_debug_contact_after(self.h, int(e), _p(tau), int(prior), _p(tap))
        return {k: tap[k][0].copy() for k in tap.dtype.names}

    def set_curriculum(self, c):
        self.lib.sso_set_curriculum(self.h, int(c))

    def set_specialist(self, c):
        self.lib.sso_set_specialist(self.h, int(c))

    def set_sample_prob(self, p):
        p = np.ascontiguousarray(p, np.float64)
        per_env = 1 if p.size == self.n * NCELL and p.ndim == 3 else 0
        assert p.size == (self.n * NCELL if per_env else NCELL)
        self.lib.sso_set_sample_prob(self.h, _p(p), per_env)

    def set_power(self, power):
        self.lib.sso_set_power(self.h, float(power))

    def set_auto_reset(self, on):
        self.lib.sso_set_auto_reset(self.h, 1 if on else 0)

    def create_temp_states(self):
        out = np.zeros((self.n, NCELL, OBS_DIM), np.float32)
        self.lib.sso_create_temp_states(self.h, _p(out))
        return out

    def get_state(self):
        st = np.zeros((self.n, STATE_DIM), self.real)
        self.lib.sso_get_state(self.h, _p(st))
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, self.real).reshape(self.n, STATE_DIM)
        self.lib.sso_set_state(self.h, _p(st))

    def get_obs(self):
        obs = np.zeros((self.n, OBS_DIM), np.float32)
        self.lib.sso_get_obs(self.h, _p(obs))
        return obs

    def random_actions(self, t):
        act = np.zeros((self.n, ACT_DIM), np.float32)
        self.lib.sso_random_actions(self.h, int(t), _p(act))
        return act

    def substeps(self, e, tau, n):
        tau = np.ascontiguousarray(tau, self.real)
        flags = np.zeros(4, np.int32)
        self.lib.sso_debug_substeps(self.h, int(e), _p(tau), int(n), _p(flags))
        return flags


def philox(ctr, key):
    lib = load("f32")
    ctr = np.asarray(ctr, np.uint32)
    key = np.asarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib.sso_philox(_p(ctr), _p(key), _p(out))
    return out


def debug_aba(kind, packed, tau, prec="f64"):
    lib = load(prec)
    real = np.float32 if prec == "f32" else np.float64
    packed = np.ascontiguousarray(packed, real)
    tau = np.ascontiguousarray(tau, real)
    qdd = np.zeros(21, real)
    a0 = np.zeros(6, real)
    lib.sso_debug_aba(KIND[kind], _p(packed), _p(tau), _p(qdd), _p(a0))
    return qdd, a0


def debug_fk(kind, packed, prec="f64"):
    lib = load(prec)
    real = np.float32 if prec == "f32" else np.float64
    packed = np.ascontiguousarray(packed, real)
    pos = np.zeros((22, 3), real)
    rot = np.zeros((22, 3, 3), real)
    lib.sso_debug_fk(KIND[kind], _p(packed), _p(pos), _p(rot))
    return pos, rot


# packed-state slices (layout documented in include/steppingstone.h)
S_POS, S_QUAT, S_VEL, S_Q, S_QD = slice(0, 3), slice(3, 7), slice(7, 13), slice(13, 34), slice(34, 55)
S_POT, S_ZINIT, S_EPRET, S_NNDR, S_N, S_COUNT, S_ELAPSED, S_CTRLO, S_CTRHI, S_FLAGS = range(55, 65)
S_TERRAIN = slice(65, 185)
S_EPRET_LO = 185
