"""Independent numpy evaluation of the Philox-driven parts of the environment (docs/PHYSICS.md 6 and 7): Philox4x32-10 written
from the Random123 description, the stone draw (inverse CDF of the 11x11 grid, step length, tilts, heading recursion) and the
reset pose.  TEST INFRASTRUCTURE ONLY: pins the C oracle's sampler / reset (tests/test_oracle_terrain_numpy.py); the device is
compared with the oracle bit-exactly on these integer-driven paths (tests/test_gpu_parity.py)."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF
GRID = 11
DEG = np.pi / 180.0
YAW = np.linspace(-20.0, 20.0, GRID) * DEG
PITCH = np.linspace(-30.0, 30.0, GRID) * DEG


def philox4x32_10(ctr, key):
    c = [int(x) & MASK for x in ctr]
    k = [int(x) & MASK for x in key]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & MASK, (p0 >> 32) ^ c[3] ^ k[1], p0 & MASK]
        k = [(k[0] + W0) & MASK, (k[1] + W1) & MASK]
    return c


def uniforms(seed, ctr, stream, env_id):
    """One block of four uniforms u = (x >> 8) * 2^-24 for the env-level stream layout of PHYSICS.md 6."""
    x = philox4x32_10([ctr, stream, env_id, 0], [seed & MASK, (seed >> 32) & MASK])
    return [np.float32((v >> 8) * (1.0 / 16777216.0)) for v in x]


def window_grid(level, ring=False):
    p = np.zeros((GRID, GRID), np.float32)
    for i in range(GRID):
        for j in range(GRID):
            m = max(abs(i - 5), abs(j - 5))
            p[i, j] = 1.0 if (m == level if ring else m <= level) else 0.0
    return (p / np.float32(p.sum())).astype(np.float32)


def pick_cell(prob, u0):
    """Inverse CDF in index order i*11+j with an fp32 running sum; first cell with u0 < cdf, fallback last cell with p > 0."""
    p = np.asarray(prob, np.float32).reshape(-1)
    cdf = np.float32(0.0)
    last = 0
    for k in range(p.size):
        if p[k] > 0:
            last = k
        cdf = np.float32(cdf + p[k])
        if u0 < cdf:
            return k // GRID, k % GRID
    return last // GRID, last % GRID


def draw_stone(prev, prob, level, u):
    """Stone k from stone k-1 = (x, y, z, phi, ...), the sampling grid, the curriculum level c and one block of uniforms."""
    i, j = pick_cell(prob, u[0])
    yaw, pitch = YAW[i], PITCH[j]
    dr = 0.65 + float(u[1]) * 0.6 * level / 5.0
    xt = (2.0 * float(u[2]) - 1.0) * 15.0 * DEG * level / 5.0
    yt = (2.0 * float(u[3]) - 1.0) * 15.0 * DEG * level / 5.0
    phi = prev[3] + yaw
    return np.array([prev[0] + dr * np.cos(pitch) * np.cos(phi), prev[1] + dr * np.cos(pitch) * np.sin(phi),
                     prev[2] + dr * np.sin(pitch), phi, xt, yt]), (i, j)


def reset_joint_angles(m, seed, ctr, env_id):
    """q = clip(q0 + 0.05 (2u - 1), lo + 0.02, hi - 0.02) from 6 consecutive blocks (joint j uses uniform j of the 24)."""
    u = []
    for b in range(6):
        u += uniforms(seed, ctr + b, 0, env_id)
    lo, hi = m["range"][:, 0], m["range"][:, 1]
    return np.clip(m["q0"] + 0.05 * (2.0 * np.array(u[:21], np.float64) - 1.0), lo + 0.02, hi - 0.02)
