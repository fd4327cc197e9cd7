// ss_kernels.hpp -- environment logic around the physics substep: control step, reward, termination,
// auto-reset, terrain sampler, observation assembly (docs/PHYSICS.md sections 4-8).
//
// HBM layout (structure of arrays, env index fastest so every field access of a wavefront is one coalesced
// 256-byte transaction):
//   fstate [NF][Npad] float : pos3 quat4 twist6 q21 qd21 pot z_init ep_ret nn_dr | 3 active stones x 8 | 3 x (cos, sin) of their headings | ep_ret_lo
//   istate [NI][Npad] int   : next_step_index, target_reached_count, elapsed, rng_ctr, flags, prov_from
//   terrain [20*6][Npad] float : terrain_info rows of the stones k < prov_from; the stones from prov_from on are the provisional
//                                straight flat path (x = 0.75 k, everything else 0; PHYSICS.md 6) BY DEFINITION and their rows are
//                                not written: a reset sets prov_from = 0 instead of storing 120 words that are scattered over 120
//                                rows -- at the benchmark's reset rate that was 1.0 MB of sector writes plus 1.2 MB of fill reads
//                                per 4096-env step, a third of the launch's HBM traffic (round 3).  Touched on stone advance,
//                                get_state / set_state and by create_temp_states only.
//   prob   [121] float shared grid, or [121][Npad] per-env grids
#pragma once
#include "ss_dynamics.hpp"
// Two tuning options measured in round 6 and NOT adopted (profiles/r06_ab_disc_vs_plank_variants.txt): the benchmark actions' six Philox
// blocks split between the lanes of a pair (-0.4 % / -0.6 % at 4096 envs, +0.9 % in the plain K-step kernel), and the one-launch-per-
// step kernel's output stage on a helper wavefront (SS_STEP_EMIT_OFFLOAD below: no gain, as in round 3).
#ifndef SS_SPLIT_ACTION_PHILOX
#define SS_SPLIT_ACTION_PHILOX 0
#endif
#ifndef SS_EMIT_ON_LAST_HELPER
#define SS_EMIT_ON_LAST_HELPER 1
#endif
#ifndef SS_NUM_SUBSTEPS
#define SS_NUM_SUBSTEPS 4
#endif
#include "../../include/steppingstone.h"

namespace ss {

enum { F_POS = 0, F_QUAT = 3, F_VEL = 7, F_Q = 13, F_QD = 34, F_POT = 55, F_ZINIT = 56, F_EPRET = 57, F_NNDR = 58,
       F_STONE = 59, F_HEAD = 59 + 24, F_EPRET_LO = 59 + 30, NF = 59 + 30 + 1 };
enum { I_N = 0, I_COUNT = 1, I_ELAPSED = 2, I_RNG = 3, I_FLAGS = 4, I_PROV = 5, NI = 6 };
constexpr int kNumStones = 20;
constexpr float kDeg = 0.017453292519943295f;

// What the hooks (update_curriculum / update_specialist / update_sample_prob / set_robot_params / auto-reset switch)
// change between steps.  It lives in HBM and the kernels read it through Params::knobs, so a hipGraph that captured
// step launches sees every later update (a by-value kernel argument would be frozen at capture time).
struct Knobs {
  const float* prob;     // shared [121] or per-env [121][Npad]
  int per_env_prob;
  int curriculum;
  float power;
  int auto_reset;
};

struct Params {
  float* fstate;
  int* istate;
  float* terrain;
  const Knobs* knobs;    // device-resident (host harness: host memory)
  int n;                 // number of envs
  int npad;              // padded to a multiple of 64
  uint32_t seed_lo, seed_hi;
  uint32_t env_offset;
  uint32_t id_mask;      // global env id = env_offset + (e & id_mask): all ones, except under ss_debug_set_id_mask (duplicate-env self-check)
  unsigned long long* prof;   // 16 phase counters, tuning builds (-DSS_PROFILE_PHASES) only
};

using Cache = Stones;    // active stones n-1, n, n+1: centre, normal, tilts (ss_dynamics.hpp)

struct Dyn {             // full-robot dynamic state, true world (reset / obs / temp-state kernels)
  float pos[3];
  float quat[4];
  SV v0;                 // base twist, body coordinates
  float q[NJ];
  float qd[NJ];
};

SSD float yaw_sample(int i) { return (-20.0f + 4.0f * (float)i) * kDeg; }
SSD float pitch_sample(int j) { return (-30.0f + 6.0f * (float)j) * kDeg; }

SSD void stone_normal(float phi, float xt, float yt, float n[3]) {
  float sx, cx, sy, cy, sp, cp;
  sincosf(xt, &sx, &cx);
  sincosf(yt, &sy, &cy);
  sincosf(phi, &sp, &cp);
  float x1 = sy * cx, y1 = -sx, z1 = cy * cx;
  n[0] = cp * x1 - sp * y1;
  n[1] = sp * x1 + cp * y1;
  n[2] = z1;
}

SSD void env_block(const Params& P, int e, uint32_t& ctr, uint32_t out[4]) {
  philox4x32_10(ctr, 0u, P.env_offset + ((uint32_t)e & P.id_mask), 0u, P.seed_lo, P.seed_hi, out);
  ctr += 1u;
}

SSD int sample_cell(const Params& P, const Knobs& K, int e, float u) {
  float cdf = 0.f;
  int last = 0, pick = -1;
  const float* pr = K.per_env_prob ? K.prob + e : K.prob;
  const int stride = K.per_env_prob ? P.npad : 1;
#pragma unroll 1
  for (int k = 0; k < SS_NCELL; ++k) {
    float pk = pr[(size_t)k * stride];
    if (pk > 0.f) last = k;
    cdf += pk;
    if (pick < 0 && u < cdf) pick = k;
  }
  return pick < 0 ? last : pick;
}

// draw stone k from stone k-1 (terrain table), write it to the table; returns dr and the new stone's data
SSD float draw_stone(const Params& P, const Knobs& K, int e, uint32_t& ctr, int k, float out_p[3], float out_n[3],
                     float out_t[2], float out_h[2], bool store = true) {
  uint32_t r[4];
  env_block(P, e, ctr, r);
  int cell = sample_cell(P, K, e, u01(r[0]));
  float ratio = (float)K.curriculum / 5.0f;
  float dr = 0.65f + u01(r[1]) * (0.6f * ratio);
  float tilt = 15.0f * kDeg * ratio;
  float xt = (2.f * u01(r[2]) - 1.f) * tilt, yt = (2.f * u01(r[3]) - 1.f) * tilt;
  float yaw = yaw_sample(cell / SS_GRID), pitch = pitch_sample(cell % SS_GRID);
  const size_t np = (size_t)P.npad;
  float* T = P.terrain + e;
  // stones from prov_from on are provisional (not stored): the first draw of an episode (k = 3) starts from the provisional
  // stone 2 and puts the provisional stones 0..2 into the table
  int prov = P.istate[e + I_PROV * np];
  const bool prev_stored = k - 1 < prov;
  float px = prev_stored ? T[((k - 1) * 6 + 0) * np] : 0.75f * (float)(k - 1);
  float py = prev_stored ? T[((k - 1) * 6 + 1) * np] : 0.f, pz = prev_stored ? T[((k - 1) * 6 + 2) * np] : 0.f;
  float phi = (prev_stored ? T[((k - 1) * 6 + 3) * np] : 0.f) + yaw;
  float sp, cp, sph, cph;
  sincosf(pitch, &sp, &cp);
  sincosf(phi, &sph, &cph);
  float planar = dr * cp;
  out_p[0] = px + planar * cph;
  out_p[1] = py + planar * sph;
  out_p[2] = pz + dr * sp;
  out_t[0] = xt; out_t[1] = yt;
  out_h[0] = cph; out_h[1] = sph;
  stone_normal(phi, xt, yt, out_n);
  if (store) {
#pragma unroll 1
    for (int j = prov; j < k; ++j) {            // (rare: once per episode that gets past its second stone)
      T[(j * 6 + 0) * np] = 0.75f * (float)j;
#pragma unroll
      for (int i = 1; i < 6; ++i) T[(j * 6 + i) * np] = 0.f;
    }
    T[(k * 6 + 0) * np] = out_p[0]; T[(k * 6 + 1) * np] = out_p[1]; T[(k * 6 + 2) * np] = out_p[2];
    T[(k * 6 + 3) * np] = phi; T[(k * 6 + 4) * np] = xt; T[(k * 6 + 5) * np] = yt;
    if (prov < k + 1) P.istate[e + I_PROV * np] = k + 1;
  }
  return dr;
}

SSD void quat_rpy(const float q[4], float& roll, float& pitch, float& yaw) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  roll = atan2f(2.f * (w * x + y * z), 1.f - 2.f * (x * x + y * y));
  pitch = asinf(fminf(fmaxf(2.f * (w * y - z * x), -1.f), 1.f));
  yaw = atan2f(2.f * (w * z + x * y), 1.f - 2.f * (y * y + z * z));
}

SSD float planar_dist(const float a[3], const float b[3]) {
  float dx = a[0] - b[0], dy = a[1] - b[1];
  return sqrtf(dx * dx + dy * dy);
}

// roll and pitch of the base (PHYSICS.md 5), and cos / sin of its yaw WITHOUT the angle: yaw = atan2(B, A) with
// A = 1 - 2(y^2 + z^2), B = 2(wz + xy), so (cos yaw, sin yaw) = (A, B) / |(A, B)| (degenerate |(A,B)| = 0: yaw = 0).
SSD void quat_roll_pitch_cs(const float q[4], float& roll, float& pitch, float& cy, float& sy) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  roll = atan2f(2.f * (w * x + y * z), 1.f - 2.f * (x * x + y * y));
  pitch = asinf(fminf(fmaxf(2.f * (w * y - z * x), -1.f), 1.f));
  float A = 1.f - 2.f * (y * y + z * z), B = 2.f * (w * z + x * y);
  float n2 = A * A + B * B;
  float inv = SS_RSQRT(fmaxf(n2, 1e-30f));
  cy = n2 > 1e-30f ? A * inv : 1.f;
  sy = n2 > 1e-30f ? B * inv : 0.f;
}

// target block of the observation (PHYSICS.md 5): [sin(dtheta) d, cos(dtheta) d, dz, x_tilt, y_tilt] with dtheta the
// bearing of the stone relative to the body yaw and d the planar distance -- i.e. the planar offset rotated by -yaw:
// sin(a - yaw) d = dy cos(yaw) - dx sin(yaw), cos(a - yaw) d = dx cos(yaw) + dy sin(yaw).  No atan2 / sincos needed.
SSD void target_features(const float pos[3], float cy, float sy, const float sp[3], const float tilt[2], float o[5]) {
  float dx = sp[0] - pos[0], dy = sp[1] - pos[1], dz = sp[2] - pos[2];
  o[0] = dy * cy - dx * sy; o[1] = dx * cy + dy * sy; o[2] = dz; o[3] = tilt[0]; o[4] = tilt[1];
}

SSD float clip5(float x) { return fminf(fmaxf(x, -5.f), 5.f); }

// observation (PHYSICS.md section 5) written row-major to obs[60]
template <class Model>
SSD void write_obs(const Dyn& s, float z_init, int flags, const Cache& c, float* obs) {
  float roll, pitch, sy, cy;
  quat_roll_pitch_cs(s.quat, roll, pitch, cy, sy);
  float R[3][3];
  quat_rot(s.quat, R);
  float vw[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) vw[r] = R[r][0] * s.v0.v[0] + R[r][1] * s.v0.v[1] + R[r][2] * s.v0.v[2];
  obs[0] = clip5(s.pos[2] - z_init);
  obs[1] = clip5(cy * vw[0] + sy * vw[1]);
  obs[2] = clip5(-sy * vw[0] + cy * vw[1]);
  obs[3] = clip5(vw[2]);
  obs[4] = clip5(roll);
  obs[5] = clip5(pitch);
  static_for<0, NJ>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value;
    constexpr float mid = 0.5f * (Model::lo[j] + Model::hi[j]);
    constexpr float span = Model::hi[j] - Model::lo[j];
    constexpr float ps = (float)kPolicySign[j];       // policy coordinates, PHYSICS.md 2 (negations are exact: same bits as emit_outputs)
    obs[6 + j] = clip5(2.f * (ps * s.q[j] - ps * mid) / span);
    obs[27 + j] = clip5(0.1f * (ps * s.qd[j]));
  });
  obs[48] = (flags & 1) ? 1.f : 0.f;
  obs[49] = (flags & 2) ? 1.f : 0.f;
  float t[5];
  target_features(s.pos, cy, sy, c.p[1], c.tilt[1], t);
#pragma unroll
  for (int i = 0; i < 5; ++i) obs[50 + i] = t[i];
  target_features(s.pos, cy, sy, c.p[2], c.tilt[2], t);
#pragma unroll
  for (int i = 0; i < 5; ++i) obs[55 + i] = t[i];
}

// ---------------------------------------------------------------------------------------------------------------
SSD void load_dyn(const Params& P, int e, Dyn& s) {
  const float* F = P.fstate + e;
  const size_t np = (size_t)P.npad;
#pragma unroll
  for (int i = 0; i < 3; ++i) s.pos[i] = F[(F_POS + i) * np];
#pragma unroll
  for (int i = 0; i < 4; ++i) s.quat[i] = F[(F_QUAT + i) * np];
#pragma unroll
  for (int i = 0; i < 3; ++i) { s.v0.w[i] = F[(F_VEL + i) * np]; s.v0.v[i] = F[(F_VEL + 3 + i) * np]; }
#pragma unroll
  for (int j = 0; j < NJ; ++j) { s.q[j] = F[(F_Q + j) * np]; s.qd[j] = F[(F_QD + j) * np]; }
}
SSD void store_dyn(const Params& P, int e, const Dyn& s) {
  float* F = P.fstate + e;
  const size_t np = (size_t)P.npad;
#pragma unroll
  for (int i = 0; i < 3; ++i) F[(F_POS + i) * np] = s.pos[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) F[(F_QUAT + i) * np] = s.quat[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) { F[(F_VEL + i) * np] = s.v0.w[i]; F[(F_VEL + 3 + i) * np] = s.v0.v[i]; }
#pragma unroll
  for (int j = 0; j < NJ; ++j) { F[(F_Q + j) * np] = s.q[j]; F[(F_QD + j) * np] = s.qd[j]; }
}
SSD void load_cache(const Params& P, int e, Cache& c) {
  const float* F = P.fstate + e;
  const size_t np = (size_t)P.npad;
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { c.p[sl][i] = F[(F_STONE + sl * 8 + i) * np]; c.nrm[sl][i] = F[(F_STONE + sl * 8 + 3 + i) * np]; }
    c.tilt[sl][0] = F[(F_STONE + sl * 8 + 6) * np];
    c.tilt[sl][1] = F[(F_STONE + sl * 8 + 7) * np];
  }
}
SSD void store_cache(const Params& P, int e, const Cache& c) {
  float* F = P.fstate + e;
  const size_t np = (size_t)P.npad;
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { F[(F_STONE + sl * 8 + i) * np] = c.p[sl][i]; F[(F_STONE + sl * 8 + 3 + i) * np] = c.nrm[sl][i]; }
    F[(F_STONE + sl * 8 + 6) * np] = c.tilt[sl][0];
    F[(F_STONE + sl * 8 + 7) * np] = c.tilt[sl][1];
  }
}
// the active stones' headings (cos, sin of phi): what the plank footprint is aligned with (PHYSICS.md 3.3).  Their own 6 rows of fstate,
// written on reset / advance / set_state only; inside a step they live in the lane's LDS (kLdsHead), never in the Cache registers
SSD void store_headings(const Params& P, int e, const float (&h)[3][2]) {
  float* F = P.fstate + e;
  const size_t np = (size_t)P.npad;
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) { F[(F_HEAD + sl * 2) * np] = h[sl][0]; F[(F_HEAD + sl * 2 + 1) * np] = h[sl][1]; }
}
// rebuild the active-stone cache of env e from the terrain table (after set_state)
SSD void cache_from_terrain(const Params& P, int e, int n, Cache& c, float (&hd)[3][2]) {
  const size_t np = (size_t)P.npad;
  const float* T = P.terrain + e;
  int idx[3] = {n - 1 < 0 ? 0 : n - 1, n, n + 1 > kNumStones - 1 ? kNumStones - 1 : n + 1};
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    int k = idx[sl];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.p[sl][i] = T[(k * 6 + i) * np];
    float phi = T[(k * 6 + 3) * np], xt = T[(k * 6 + 4) * np], yt = T[(k * 6 + 5) * np];
    c.tilt[sl][0] = xt; c.tilt[sl][1] = yt;
    stone_normal(phi, xt, yt, c.nrm[sl]);
    sincosf(phi, &hd[sl][1], &hd[sl][0]);
  }
}

// PHYSICS.md section 7
template <class Model>
SSD void env_reset(const Params& P, int e, Dyn& s, Cache& c, uint32_t& ctr, float& pot, float& z_init, float& nn_dr) {
  const size_t np = (size_t)P.npad;
  P.istate[e + I_PROV * np] = 0;                // the whole path is provisional again: nothing is written to the terrain table
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    c.p[sl][0] = 0.75f * (float)sl; c.p[sl][1] = 0.f; c.p[sl][2] = 0.f;
    c.nrm[sl][0] = 0.f; c.nrm[sl][1] = 0.f; c.nrm[sl][2] = 1.f;
    c.tilt[sl][0] = 0.f; c.tilt[sl][1] = 0.f;
  }
  s.pos[0] = 0.f; s.pos[1] = 0.f; s.pos[2] = Model::stand_height + 0.01f;
  s.quat[0] = 1.f; s.quat[1] = s.quat[2] = s.quat[3] = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { s.v0.w[i] = 0.f; s.v0.v[i] = 0.f; }
  uint32_t r[6][4];
#pragma unroll
  for (int b = 0; b < 6; ++b) env_block(P, e, ctr, r[b]);
  static_for<0, NJ>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value;
    constexpr float q0 = Model::q0[j], lo = Model::lo[j] + 0.02f, hi = Model::hi[j] - 0.02f;
    float q = q0 + 0.05f * (2.f * u01(r[j / 4][j % 4]) - 1.f);
    s.q[j] = fminf(fmaxf(q, lo), hi);
    s.qd[j] = 0.f;
  });
  z_init = s.pos[2];
  nn_dr = 0.75f;
  pot = -planar_dist(c.p[1], s.pos) / kDt;
}

// Peer-to-peer all-gather fused into the step kernel (multi-GPU, ss_step_packed_peers): every workgroup stores its rows of
// the packed block straight into each peer's gather buffer (xGMI stores into fine-grained memory) instead of a local
// buffer that a collective then ships.  Completion: every workgroup fences (system scope) and counts itself; the last
// one of a launch publishes flag_value in each peer's flag word of this rank.
constexpr int kMaxPeers = 8;
struct PeerTable {
  float* dst[kMaxPeers];        // peer p's gather buffer [G * n_local, 62] (p == my rank: my own)
  uint32_t* flag[kMaxPeers];    // peer p's flag array [G]
  uint32_t* done_counter;       // local, device scope: workgroups finished (monotonic)
  int count;                    // G
  int rank;                     // my rank: rows [rank * n_local, (rank+1) * n_local) and flag[p][rank] are mine
  int n_local;
};

struct StepIO {
  const float* act;   // [N,21] or null when actions are generated on device
  float* obs;         // [N,60]
  float* rew;         // [N]
  uint8_t* done;      // [N]
  ss_info* info;      // [N] or null
  uint64_t t;         // action-stream index for RANDOM_ACT
  float* packed;      // optional [N,62] = obs | rew | done(0/1): the block the multi-GPU all-gather ships; when set,
                      // obs / rew / done above may be null
  int nsteps;         // control steps per launch (rollout kernels; 1 otherwise)
  const PeerTable* peers;   // optional (device memory): also store the packed block into every peer's gather buffer
  uint32_t flag_value;      // what the last workgroup publishes in the peers' flag words (the step number)
  long long packed_step_stride;   // rollout kernels: step k of the launch writes its packed block at packed + k * stride
                                  // (0: every step overwrites the same block)
};

// draw of the reset joint noise for global joint gj (PHYSICS.md section 7); r = the 6 Philox blocks of the reset
template <class Model, int GJ>
SSD float reset_angle(const uint32_t (&r)[6][4]) {
  constexpr float q0 = Model::q0[GJ], lo = Model::lo[GJ] + 0.02f, hi = Model::hi[GJ] - 0.02f;
  float q = q0 + 0.05f * (2.f * u01(r[GJ / 4][GJ % 4]) - 1.f);
  return fminf(fmaxf(q, lo), hi);
}

// Global stores of a step's outputs and state rows.  -DSS_NT_STORES makes them non-temporal (`nt`) in the one-launch-per-step
// kernels: nothing reads the rows again before the kernel ends, and a kernel boundary on this chip writes back whatever is dirty in
// eight private L2s -- measured 0.0648 -> 0.0640 ms/step at 4096 envs (system-scope write-through stores: the same).  Off by
// default: 1.2 % of a secondary figure.  (Round 3 left them off because a non-reproducible one-step mismatch had shown up next to
// them; round 4 root-caused that to copy_block's padding workgroup, DESIGN.md 5.1b -- unrelated to the stores.)  The committed
// profiles are of the plain stores.
#if defined(__HIP_DEVICE_COMPILE__) && defined(SS_NT_STORES)
typedef float ss_v4f __attribute__((ext_vector_type(4)));
template <bool NT, class T>
SSD void gst(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool NT>
SSD void gst4(float4* p, float4 v) {
  if constexpr (NT) __builtin_nontemporal_store(ss_v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<ss_v4f*>(p));
  else *p = v;
}
#else
template <bool NT, class T>
SSD void gst(T* p, T v) { *p = v; }
template <bool NT>
SSD void gst4(float4* p, float4 v) { *p = v; }
#endif
// ---- output stage of a control step: observation / reward / done / info rows and the bulk of the state write-back ----
struct StepOut {        // what it needs, true world, after the optional reset
  float pos[3], quat[4];
  SV v0;
  float qt[NH], qdt[NH];                      // this lane's joints
  float r, z_init, roll, pitch, cyaw, syaw;   // roll .. syaw: of the orientation the step ended in (not used after a reset)
  int d, flags, do_reset;
  float tgt[2][5];                            // stones n and n+1: centre (3), tilts (2)
  ss_info inf;
};
// Observation / info rows are staged in LDS (region A is free at that point) and written out by the whole wavefront as
// contiguous 256-byte stores: a lane writing its own [60]-float row directly would touch 32 partial cache lines
// per store instruction (measured 3.1x the algorithmic HBM traffic before this).  One body for the device and for the
// host pre-flight (tests/host/host_harness.cpp runs the 64 lanes of a wavefront as threads around the same LDS block).
template <class Model, bool ROLLOUT>
SSD void emit_outputs(const Params& P, const StepIO& io, const StepOut& o, int e, int side, bool valid, int lane, int lane_global,
                      int kstep, float* lds) {
  const size_t np = (size_t)P.npad;
  // Rows are staged back to back in the layout of the output block -- [32][60] for obs, [32][62] = obs | rew | done for the
  // packed block -- so that the wavefront copies the block with float4 loads / stores (8 iterations instead of 31 scalar
  // ones with an index division each; the 4-way bank conflicts of the stride-60 staging writes are fire-and-forget).
  constexpr int kPackW = SS_OBS_DIM + 2;
  const bool packed_layout = io.packed != nullptr || io.peers != nullptr;   // wavefront-uniform
  const int stride = packed_layout ? kPackW : SS_OBS_DIM;
  constexpr int kInfoBase = kEnvsPerWave * 64;         // behind the largest staged block
  float* stage = lds + (lane >> 1) * stride;
  uint32_t* istage = reinterpret_cast<uint32_t*>(lds) + kInfoBase + (lane >> 1) * SS_INFO_WORDS;
#define SS_OBS(i) stage[i]
  SS_WAVE_SYNC();          // the staging area is region A: every lane must be through with its contact operators
  if (valid) {
    float* Fo = P.fstate + e;
    // per-joint state + observation entries: own limbs by each lane, spine by the right lane
    static_for<0, NH>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value, jr = kHalf[k];
      constexpr int jl = jr < 3 ? jr : (jr < 8 ? jr + 5 : jr + 4);
      const int gj = side ? jl : jr;
      if (jr >= 3 || side == 0) {
        // The observation carries POLICY coordinates (PHYSICS.md 2): ps = kPolicySign of THIS joint (left x / z joints: about the
        // mirrored axis; knees: negative in flexion), applied to the true-world angle and to the middle of its true range (a left
        // x / z joint's true range is (-hi, -lo) of its right twin's).  Same expression, same bits as write_obs.  The state arrays
        // keep angles about the +axis.
        constexpr float midr = 0.5f * (Model::lo[jr] + Model::hi[jr]);
        constexpr float span = Model::hi[jr] - Model::lo[jr];
        const float ps = policy_true_sign(jr, side);
        const float mid = (side && mirror_flips(jr)) ? -midr : midr;
        gst<!ROLLOUT>(&Fo[(F_Q + gj) * np], o.qt[k]);
        gst<!ROLLOUT>(&Fo[(F_QD + gj) * np], o.qdt[k]);
        SS_OBS(6 + gj) = clip5(2.f * (ps * o.qt[k] - ps * mid) / span);
        SS_OBS(27 + gj) = clip5(0.1f * (ps * o.qdt[k]));
      }
    });
    if (side == 0) {
      float R[3][3];
      quat_rot(o.quat, R);
      float vw[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) vw[i] = R[i][0] * o.v0.v[0] + R[i][1] * o.v0.v[1] + R[i][2] * o.v0.v[2];
      // a reset leaves the identity orientation: roll = pitch = yaw = 0 exactly; otherwise the values of the final orientation
      const float r2 = o.do_reset ? 0.f : o.roll, p2 = o.do_reset ? 0.f : o.pitch;
      const float cy = o.do_reset ? 1.f : o.cyaw, sy = o.do_reset ? 0.f : o.syaw;
      SS_OBS(0) = clip5(o.pos[2] - o.z_init);
      SS_OBS(1) = clip5(cy * vw[0] + sy * vw[1]);
      SS_OBS(2) = clip5(-sy * vw[0] + cy * vw[1]);
      SS_OBS(3) = clip5(vw[2]);
      SS_OBS(4) = clip5(r2);
      SS_OBS(5) = clip5(p2);
      SS_OBS(48) = (o.flags & 1) ? 1.f : 0.f;
      SS_OBS(49) = (o.flags & 2) ? 1.f : 0.f;
      float t[5];
      target_features(o.pos, cy, sy, &o.tgt[0][0], &o.tgt[0][3], t);
#pragma unroll
      for (int i = 0; i < 5; ++i) SS_OBS(50 + i) = t[i];
      target_features(o.pos, cy, sy, &o.tgt[1][0], &o.tgt[1][3], t);
#pragma unroll
      for (int i = 0; i < 5; ++i) SS_OBS(55 + i) = t[i];
      if (io.rew) io.rew[e] = o.r;
      if (io.done) io.done[e] = o.d ? 1 : 0;
      if (packed_layout) {
        stage[SS_OBS_DIM] = o.r;
        stage[SS_OBS_DIM + 1] = o.d ? 1.f : 0.f;
      }
      istage[0] = SS_F2U(o.inf.ep_ret); istage[1] = SS_F2U(o.inf.ep_len);
      istage[2] = (uint32_t)o.inf.bad_transition; istage[3] = (uint32_t)o.inf.steps_reached; istage[4] = (uint32_t)o.inf.update_terrain;
      istage[5] = SS_F2U(o.inf.ep_ret_lo);
#pragma unroll
      for (int i = 0; i < 3; ++i) gst<!ROLLOUT>(&Fo[(F_POS + i) * np], o.pos[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) gst<!ROLLOUT>(&Fo[(F_QUAT + i) * np], o.quat[i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) { gst<!ROLLOUT>(&Fo[(F_VEL + i) * np], o.v0.w[i]); gst<!ROLLOUT>(&Fo[(F_VEL + 3 + i) * np], o.v0.v[i]); }
    }
  }
  {
    SS_WAVE_SYNC();
    const int env0 = (lane_global - lane) >> 1;                                   // first env of this wavefront
    // (never negative: a wavefront without a valid env -- not launched since round 4, ss_api.hip: grid = ceil(n / 32) -- would
    // otherwise reach copy_block with nfl < 0, where `n4 << 2` rounds below nfl and lanes 0 and 1 stored two out-of-range LDS words
    // into the reward / done slots of env n - 1's PACKED row: the schedule-dependent step of DESIGN.md 5.1b)
    const int nleft = P.n - env0;
    const int nvalid = nleft <= 0 ? 0 : (kEnvsPerWave < nleft ? kEnvsPerWave : nleft);
    // copy the staged block (nfl floats from the start of the LDS staging area) to dst: float4 when dst is 16-byte aligned
    auto copy_block = [&](float* dst, int nfl) {
      if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const int n4 = nfl >> 2;
        float4* d4 = reinterpret_cast<float4*>(dst);
        const float4* s4 = reinterpret_cast<const float4*>(lds);
#pragma unroll 1
        for (int g = lane; g < n4; g += kWave) gst4<!ROLLOUT>(&d4[g], s4[g]);
        for (int g = (n4 << 2) + lane; g < nfl; g += kWave) gst<!ROLLOUT>(&dst[g], lds[g]);
      } else {
#pragma unroll 1
        for (int g = lane; g < nfl; g += kWave) gst<!ROLLOUT>(&dst[g], lds[g]);
      }
    };
    if (io.obs) {
      float* og = io.obs + (size_t)env0 * SS_OBS_DIM;
      if (!packed_layout) {
        copy_block(og, nvalid * SS_OBS_DIM);
      } else {                                   // both outputs requested: the staging has the packed layout
#pragma unroll 1
        for (int g = lane; g < nvalid * SS_OBS_DIM; g += kWave) {
          const int el = g / SS_OBS_DIM, idx = g - el * SS_OBS_DIM;
          gst<!ROLLOUT>(&og[g], lds[el * kPackW + idx]);
        }
      }
    }
    if (io.packed) {
      float* pk = io.packed + (size_t)env0 * kPackW;
      if constexpr (ROLLOUT) pk += (long long)kstep * io.packed_step_stride;
      copy_block(pk, nvalid * kPackW);
    }
    if (io.info) {
      uint32_t* ig = reinterpret_cast<uint32_t*>(io.info + env0);
      const uint32_t* is = reinterpret_cast<const uint32_t*>(lds) + kInfoBase;
      for (int g = lane; g < nvalid * SS_INFO_WORDS; g += kWave) gst<!ROLLOUT>(&ig[g], is[g]);
    }
#if defined(__HIP_DEVICE_COMPILE__)                   // the peer-store exchange exists on the device only (xGMI stores, system-scope atomics)
    if constexpr (!ROLLOUT) {
      if (io.peers) {
        constexpr int kPack = SS_OBS_DIM + 2;
        const PeerTable* T = io.peers;
        const int G = T->count;
        const size_t row0 = (size_t)T->rank * (size_t)T->n_local + (size_t)env0;
#pragma unroll 1
        for (int p = 0; p < G; ++p) copy_block(T->dst[p] + row0 * kPack, nvalid * kPack);
        __threadfence_system();                                   // my rows are visible to every agent ...
        if (lane == 0) {
          const uint32_t prev = atomicAdd(T->done_counter, 1u);   // ... before I count myself
          if ((prev + 1u) % gridDim.x == 0u) {                    // last workgroup of this launch (launches are stream-ordered)
            __threadfence_system();
            for (int p = 0; p < G; ++p)
              __hip_atomic_store(T->flag[p] + T->rank, io.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    }
#endif
  }
#undef SS_OBS
}

// The three-helper rollout kernel hands the output stage to helper 1: the main wavefront leaves 14 words in the hand-off region (the
// place of the detection's Rf / pen, which that variant does not use, and the two spare words), everything else is the state in
// LDS region B (refreshed after a reset).  Helper 1 works while the main wavefront is already in the next
// control step (between barriers #0b and #1 of its first substep; the state in LDS changes at that substep's end only).
#ifndef SS_STEP_EMIT_OFFLOAD
#define SS_STEP_EMIT_OFFLOAD 0
#endif
constexpr bool out_offload(int helpers, bool rollout) { return (rollout || SS_STEP_EMIT_OFFLOAD) && helpers >= 3; }   // (one launch per step: 0.0697 vs 0.0687 ms inline, round 3)
constexpr int kHandOut = kHandDet, kHandOut2 = kHandFloats;     // 13 + 2 words
static_assert(kHandOut2 + 2 <= kHandSlots * 4, "hand-off region");
#if !defined(SS_HOST_HARNESS)
template <class Model, bool ROLLOUT>
__device__ __forceinline__ void emit_from_handoff(const Params& P, const StepIO& io, const Lds& L, int lane, int lane_global, int kstep,
                                                  float* lds) {
  const int e_raw = lane_global >> 1, side = lane_global & 1;
  const bool valid = e_raw < P.n;
  const int e = valid ? e_raw : P.n - 1;
  const float m = side ? -1.f : 1.f;
  StepOut o;
  o.pos[0] = L.s(S_POS); o.pos[1] = m * L.s(S_POS + 1); o.pos[2] = L.s(S_POS + 2);
  o.quat[0] = L.s(S_QUAT); o.quat[1] = m * L.s(S_QUAT + 1); o.quat[2] = L.s(S_QUAT + 2); o.quat[3] = m * L.s(S_QUAT + 3);
  o.v0 = SV{{m * L.s(S_VW), L.s(S_VW + 1), m * L.s(S_VW + 2)}, {L.s(S_VV), m * L.s(S_VV + 1), L.s(S_VV + 2)}};
  static_for<0, NH>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value, jr = kHalf[k];
    const float sg = mirror_flips(jr) ? m : 1.f;
    o.qt[k] = sg * L.s(S_Q + k);
    o.qdt[k] = sg * L.s(S_QD + k);
  });
  o.r = L.hs(kHandOut + 0);
  o.z_init = L.hs(kHandOut + 1);
  {   // the lane pair shares the word: the right lane left the leading part of the episode return, the left lane the trailing part
    const float w = L.hs(kHandOut + 2), w2 = xchg(w);
    o.inf.ep_ret = side ? w2 : w;
    o.inf.ep_ret_lo = side ? w : w2;
  }
  const int bits = __builtin_bit_cast(int, L.hs(kHandOut2 + 0));
  o.d = bits & 1;
  o.inf.bad_transition = (bits >> 1) & 1;
  o.inf.update_terrain = (bits >> 2) & 1;
  o.do_reset = (bits >> 3) & 1;
  o.flags = (bits >> 4) & 3;
  o.inf.steps_reached = (bits >> 8) & 31;
  o.inf.ep_len = (float)((bits >> 16) & 0xffff);
#pragma unroll
  for (int i = 0; i < 5; ++i) { o.tgt[0][i] = L.hs(kHandOut + 3 + i); o.tgt[1][i] = L.hs(kHandOut + 8 + i); }
  quat_roll_pitch_cs(o.quat, o.roll, o.pitch, o.cyaw, o.syaw);      // same function, same bits as the main wavefront's
  emit_outputs<Model, ROLLOUT>(P, io, o, e, side, valid, lane, lane_global, kstep, lds);
}
#endif

// Benchmark actions of control step tt for this lane's half of env e (PHYSICS.md 5: six Philox blocks per env and step, 21 of the
// 24 words -> U(-1,1)), in the lane's own (mirrored) world.
template <bool SPLIT = false, class Write>
SSD void random_actions_half(const Params& P, int e, int side, float m, uint32_t tt, Write&& write) {
  uint32_t ra[6][4];
  if constexpr (SPLIT && SS_SPLIT_ACTION_PHILOX) {   // three of the six blocks per lane of the pair, exchanged (integer-exact; like the reset noise): 300 instructions less per step
      // wherever the main wavefront draws the actions itself
    uint32_t mine[3][4];
#pragma unroll
    for (int b = 0; b < 3; ++b)
      philox4x32_10(6u * tt + (side ? 3u : 0u) + b, 1u, P.env_offset + ((uint32_t)e & P.id_mask), 0u, P.seed_lo, P.seed_hi, mine[b]);
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t other = xchg_u32(mine[b][i]);
        ra[b][i] = side ? other : mine[b][i];
        ra[3 + b][i] = side ? mine[b][i] : other;
      }
  } else {
#pragma unroll
    for (int b = 0; b < 6; ++b) philox4x32_10(6u * tt + b, 1u, P.env_offset + ((uint32_t)e & P.id_mask), 0u, P.seed_lo, P.seed_hi, ra[b]);
  }
  static_for<0, NH>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value, jr = kHalf[k];
    constexpr int jl = jr < 3 ? jr : (jr < 8 ? jr + 5 : jr + 4);
    const float sg = action_lane_sign(jr, m);
    const uint32_t bits = side ? ra[jl / 4][jl % 4] : ra[jr / 4][jr % 4];
    write(k, sg * (2.f * u01(bits) - 1.f));
  });
}

// One control step, lane `lane_global` = 2*env + side (side 0: right half, true world; side 1: left half, mirrored
// world).  PHYSICS.md section 4.  Env-level logic runs redundantly (and identically) in both lanes in the true world.
// ROLLOUT: io.nsteps control steps in ONE launch (actions from the benchmark Philox stream at io.t, io.t+1, ...): the
// state is loaded into LDS once and stays there between steps (the epilogue refreshes the LDS copy when an env is reset
// or its target advances); every step still writes its outputs and the HBM copy of the state, so the result after K
// steps is bit-identical to K single-step launches (tested).
template <class Model, bool RANDOM_ACT, int HELPERS = 0, bool ROLLOUT = false>
SSD void step_env(const Params& P, const StepIO& io, int lane_global, int lane, float* lds) {
  static_assert(!ROLLOUT || RANDOM_ACT, "a multi-step launch draws its actions on the device");
  constexpr bool kOffload =            // output stage on helper 1
#if defined(__HIP_DEVICE_COMPILE__)
      out_offload(HELPERS, ROLLOUT);
#else
      false;
#endif
  const int e_raw = lane_global >> 1, side = lane_global & 1;
  const bool valid = e_raw < P.n;
  int e = valid ? e_raw : P.n - 1;
  const size_t np = (size_t)P.npad;
  const float m = side ? -1.f : 1.f;            // y-mirror factor of this lane's world
  const Lds L{lds, lane};
  const float* F = P.fstate + e;

  // Only what the substeps need is loaded before them; everything the step's epilogue needs (stone tilts,
  // counters, episode statistics) is (re)loaded afterwards from the L2-hot arrays, so that nothing sits in scratch
  // across the four substeps (those parked values were 2.2 MB of scratch write-back per launch).
  {   // measured: merging these loads with the state loads below is slower (0.0867 vs 0.0857 ms/step)
    Cache c0;
    load_cache(P, e, c0);
    float h0[3][2];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) { h0[sl][0] = F[(F_HEAD + sl * 2) * np]; h0[sl][1] = F[(F_HEAD + sl * 2 + 1) * np]; }
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      L.s(S_STP + sl * 3 + 0) = c0.p[sl][0]; L.s(S_STP + sl * 3 + 1) = m * c0.p[sl][1]; L.s(S_STP + sl * 3 + 2) = c0.p[sl][2];
      L.s(S_STN + sl * 3 + 0) = c0.nrm[sl][0]; L.s(S_STN + sl * 3 + 1) = m * c0.nrm[sl][1]; L.s(S_STN + sl * 3 + 2) = c0.nrm[sl][2];
      L.q2(kLdsHead + sl) = make_float2(h0[sl][0], m * h0[sl][1]);       // the heading in this lane's (y-mirrored) world
    }
  }

  // 1. this lane's half of the state, mirrored for the left lane, into LDS.  All global loads are issued before the
  //    first LDS store: written load-store-load-store the compiler waited for every load in turn (~25 exposed L2
  //    round trips per step).
  float gin[13], qin[NH], qdin[NH];
#pragma unroll
  for (int i = 0; i < 3; ++i) gin[i] = F[(F_POS + i) * np];
#pragma unroll
  for (int i = 0; i < 4; ++i) gin[3 + i] = F[(F_QUAT + i) * np];
#pragma unroll
  for (int i = 0; i < 6; ++i) gin[7 + i] = F[(F_VEL + i) * np];
  static_for<0, NH>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value, jr = kHalf[k];
    constexpr int jl = jr < 3 ? jr : (jr < 8 ? jr + 5 : jr + 4);         // the left twin of a right-side joint
    const int gj = side ? jl : jr;
    qin[k] = F[(F_Q + gj) * np];
    qdin[k] = F[(F_QD + gj) * np];
  });
  float ain[NH];
  if constexpr (!RANDOM_ACT) {
    static_for<0, NH>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value, jr = kHalf[k];
      constexpr int jl = jr < 3 ? jr : (jr < 8 ? jr + 5 : jr + 4);
      ain[k] = io.act[(size_t)e * NJ + (side ? jl : jr)];
    });
  }
  L.s(S_POS + 0) = gin[0]; L.s(S_POS + 1) = m * gin[1]; L.s(S_POS + 2) = gin[2];
  L.s(S_QUAT + 0) = gin[3]; L.s(S_QUAT + 1) = m * gin[4]; L.s(S_QUAT + 2) = gin[5]; L.s(S_QUAT + 3) = m * gin[6];
  L.s(S_VW + 0) = m * gin[7]; L.s(S_VW + 1) = gin[8]; L.s(S_VW + 2) = m * gin[9];
  L.s(S_VV + 0) = gin[10]; L.s(S_VV + 1) = m * gin[11]; L.s(S_VV + 2) = gin[12];
  static_for<0, NH>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value, jr = kHalf[k];
    const float sg = mirror_flips(jr) ? m : 1.f;
    L.s(S_Q + k) = sg * qin[k];
    L.s(S_QD + k) = sg * qdin[k];
  });

  const int nsteps = ROLLOUT ? io.nsteps : 1;
#pragma unroll 1
  for (int kstep = 0; kstep < nsteps; ++kstep) {
  SS_FUZZ(0x60u);
  // clipped actions of this lane's joints (its own world) into LDS
  if constexpr (RANDOM_ACT) {
    bool drawn = false;
    if constexpr (ROLLOUT && HELPERS > 1) {   // (a single helper is the critical path already: 0.0612 vs 0.0638 ms/step at 16384 envs)
      if (kstep > 0) {               // the last helper wavefront drew them during the previous step (rollout_kernel_helped)
        float a[NH];
#pragma unroll
        for (int k = 0; k < NH; ++k) a[k] = L.hs(kHandAct + k);
#pragma unroll
        for (int k = 0; k < NH; ++k) L.s(S_ACT + k) = a[k];
        drawn = true;
      }
    }
    if (!drawn) random_actions_half<!(HELPERS == 0 && ROLLOUT)>(P, e, side, m, (uint32_t)io.t + (uint32_t)kstep, [&](int k, float a) { L.s(S_ACT + k) = a; });
  } else {
    static_for<0, NH>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value, jr = kHalf[k];
      const float sg = action_lane_sign(jr, m);
      const float x = ain[k];
      float a = fminf(fmaxf(x, -1.f), 1.f);
      a = (x != x) ? x : a;      // a NaN action is not clipped away (fmaxf would): it ends the episode, PHYSICS.md 4.8
      L.s(S_ACT + k) = sg * a;
    });
  }

  // 2. four substeps on the LDS-resident state
  const float power = P.knobs->power;
  FootReport fr;
#if defined(SS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
  Prof prof;
#pragma unroll
  for (int i = 0; i < 16; ++i) prof.t[i] = 0;
  prof.last = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
  Warm wm;                                     // contact impulses carried from substep to substep; every control step starts cold
  warm_clear(wm);
  if constexpr (kPgsWarm && warm_in_lds(HELPERS)) L.s(S_WKEY) = 0.f;
#pragma unroll 1
  for (int k = 0; k < SS_NUM_SUBSTEPS; ++k) substep<Model, HELPERS>(SS_PROF_ARG power, fr, L, wm);
  SS_FUZZ(0x61u);
  SS_PROF(12);
  SS_MEMBAR();
  SS_OPAQUE(e);                                 // recompute every global address below instead of spilling 27 pointers
  F = P.fstate + e;
  const Knobs K = *P.knobs;                     // wavefront-uniform scalar loads
  Cache c;                                      // true world
  load_cache(P, e, c);                          // (keeping these 32 words resident in LDS between the steps of the rollout kernel was
  // measured slower, 0.0556 vs 0.0539 ms/step: the loads' latency is covered by the arithmetic below already)
  float pot_prev = F[F_POT * np], z_init = F[F_ZINIT * np];
  float ep_ret = F[F_EPRET * np], ep_lo = F[F_EPRET_LO * np], nn_dr = F[F_NNDR * np];
  int n = P.istate[e + I_N * np], count = P.istate[e + I_COUNT * np], elapsed = P.istate[e + I_ELAPSED * np];
  uint32_t ctr = (uint32_t)P.istate[e + I_RNG * np];
  SS_PROFE(1);       // address arithmetic + issue of the epilogue's global loads

  // 3-4. back to the true world; the pair shares its feet
  float pos[3] = {L.s(S_POS), m * L.s(S_POS + 1), L.s(S_POS + 2)};
  float quat[4] = {L.s(S_QUAT), m * L.s(S_QUAT + 1), L.s(S_QUAT + 2), m * L.s(S_QUAT + 3)};
  SV v0 = {{m * L.s(S_VW), L.s(S_VW + 1), m * L.s(S_VW + 2)}, {L.s(S_VV), m * L.s(S_VV + 1), L.s(S_VV + 2)}};
  elapsed += 1;
  const float my_sole[3] = {fr.sole[0], m * fr.sole[1], fr.sole[2]};
  float ot_sole[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) ot_sole[i] = xchg(my_sole[i]);
  const int ot_contact = xchg_i(fr.contact), ot_target = xchg_i(fr.on_target);
  float sole[2][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { sole[0][i] = side ? ot_sole[i] : my_sole[i]; sole[1][i] = side ? my_sole[i] : ot_sole[i]; }
  int flags = side ? (ot_contact | (fr.contact << 1)) : (fr.contact | (ot_contact << 1));
  const int on_target = fr.on_target | ot_target;
  SS_PROFE(2);       // LDS state read-back, pair exchange of the foot reports
  // partial sums over this lane's joints (the spine is counted by the right lane only)
  float accv = 0.f, e_sum = 0.f, a2 = 0.f;
  int at_limit = 0;
  static_for<0, NH>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value, jr = kHalf[k];
    constexpr float mid = 0.5f * (Model::lo[jr] + Model::hi[jr]);
    constexpr float span = Model::hi[jr] - Model::lo[jr];
    const float q = L.s(S_Q + k), qd = L.s(S_QD + k), a = L.s(S_ACT + k);
    const bool mine = (jr >= 3) || (side == 0);
    accv += q + qd;
    if (mine) {
      e_sum += fabsf(a * (0.1f * qd));
      a2 += a * a;
      if (fabsf(2.f * (q - mid) / span) > 0.99f) at_limit += 1;
    }
  });
  accv += pos[0] + pos[1] + pos[2] + quat[0] + quat[1] + quat[2] + quat[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) accv += v0.w[i] + v0.v[i];
  e_sum += xchg(e_sum);
  a2 += xchg(a2);
  at_limit += xchg_i(at_limit);
  const bool finite = finite_bits(accv) && (xchg_i(finite_bits(accv) ? 1 : 0) != 0);
  SS_PROFE(3);       // joint sums (energy, limits, finiteness)

  // 5. target logic
  float target_old[3] = {c.p[1][0], c.p[1][1], c.p[1][2]};
  float step_bonus = 0.f;
  int advanced = 0;
  float hd[3][2] = {{1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}};      // headings of the active stones, true world: meaningful after an advance / a reset only
  if (on_target != 0) {
    count += 1;
    if (count == 1) {
      float d0 = planar_dist(sole[0], target_old), d1 = planar_dist(sole[1], target_old);
      step_bonus = 50.f * expf(-fminf(d0, d1) / 0.25f);
    }
    if (count >= 2 && n < kNumStones - 1) {
      n += 1;
      count = 0;
      advanced = 1;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        c.p[0][i] = c.p[1][i]; c.nrm[0][i] = c.nrm[1][i];
        c.p[1][i] = c.p[2][i]; c.nrm[1][i] = c.nrm[2][i];
      }
      c.tilt[0][0] = c.tilt[1][0]; c.tilt[0][1] = c.tilt[1][1];
      c.tilt[1][0] = c.tilt[2][0]; c.tilt[1][1] = c.tilt[2][1];
      // the headings move with the stones: slot 2 from the draw (at n = 19 it keeps the last stone's).  hd[] is the TRUE-world copy that
      // the write-back below stores; the lane's LDS holds its own (y-mirrored) world.  LDS is read here only: a target advance is rare,
      // a reset is not, and the write-back path must not wait for LDS
      const float2 h1 = L.q2(kLdsHead + 1), h2 = L.q2(kLdsHead + 2);
      hd[0][0] = h1.x; hd[0][1] = m * h1.y;
      hd[1][0] = h2.x; hd[1][1] = m * h2.y;
      hd[2][0] = h2.x; hd[2][1] = m * h2.y;
      if (n + 1 <= kNumStones - 1) nn_dr = draw_stone(P, K, e, ctr, n + 1, c.p[2], c.nrm[2], c.tilt[2], hd[2], valid && side == 0);
      L.q2(kLdsHead + 0) = h1;
      L.q2(kLdsHead + 1) = h2;
      L.q2(kLdsHead + 2) = make_float2(hd[2][0], m * hd[2][1]);
    }
  }
  // 6. progress
  float pot = -planar_dist(target_old, pos) / kDt;
  SS_PROFE(4);       // first use of the loaded scalars (wait for L2) + target logic (+ draw on advance)
  float progress = pot - pot_prev;
  pot_prev = advanced ? -planar_dist(c.p[1], pos) / kDt : pot;
  // 7-8
  float target_bonus = (n == kNumStones - 1 && planar_dist(c.p[1], pos) < 0.15f) ? 2.f : 0.f;
  float zs = fminf(sole[0][2], sole[1][2]);
  float tall_bonus = (pos[2] - zs > 0.7f) ? 2.f : -1.f;
  float zlow = fminf(fminf(c.p[0][2], c.p[1][2]), c.p[2][2]);
  bool d = (tall_bonus < 0.f) || (pos[2] < zlow + 0.3f) || !finite;
  bool timeout = elapsed >= SS_MAX_EPISODE_STEPS;
  int bad = timeout ? 1 : 0;                     // TimeLimitMask (common/envs_utils.py:59-65): done at the step limit, whatever else ended it
  d = d || timeout;
  // 9. reward
  float roll, pitch, cyaw, syaw;                 // of the state the step ended in (before a possible reset)
  quat_roll_pitch_cs(quat, roll, pitch, cyaw, syaw);
  float posture = 0.f;
  if (!(pitch > -0.2f && pitch < 0.4f)) posture += fabsf(pitch);
  if (!(roll > -0.4f && roll < 0.4f)) posture += fabsf(roll);
  float energy = (4.5f / NJ) * (e_sum / NJ) + (0.225f / NJ) * (a2 / NJ);
  float r = progress + step_bonus + target_bonus + tall_bonus - energy - posture - 0.1f * (float)at_limit;
  if (!finite || !finite_bits(r)) r = 0.f;
  SS_PROFE(5);       // progress, termination, roll / pitch, reward
  {   // episode return as an unevaluated float pair (ep_ret, ep_lo): error-free two-sum of the step reward, then renormalised, so
      // that the pair carries the fp64 sum of the fp32 step rewards (Monitor.update sums Python floats, common/envs_utils.py:134)
    const float s = ep_ret + r, bb = s - ep_ret;
    const float err = (ep_ret - (s - bb)) + (r - bb);
    const float lo = ep_lo + err;
    ep_ret = s + lo;
    ep_lo = lo - (ep_ret - s);
  }
  // 10. outputs, auto-reset
  ss_info inf;
  inf.ep_ret = ep_ret;
  inf.ep_ret_lo = ep_lo;
  inf.ep_len = (float)elapsed;
  inf.bad_transition = bad;
  inf.steps_reached = n;
  inf.update_terrain = advanced;
  const bool do_reset = d && K.auto_reset;
  SS_PROFE(6);
  uint32_t rr[6][4];
  if (do_reset) {
    // PHYSICS.md section 7: provisional terrain (prov_from = 0: no table writes), standing pose, joint noise from 6 Philox blocks
    if (valid && side == 0) P.istate[e + I_PROV * np] = 0;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      c.p[sl][0] = 0.75f * (float)sl; c.p[sl][1] = 0.f; c.p[sl][2] = 0.f;
      c.nrm[sl][0] = 0.f; c.nrm[sl][1] = 0.f; c.nrm[sl][2] = 1.f;
      c.tilt[sl][0] = 0.f; c.tilt[sl][1] = 0.f;
      hd[sl][0] = 1.f; hd[sl][1] = 0.f;
      L.q2(kLdsHead + sl) = make_float2(1.f, 0.f);
    }
    pos[0] = 0.f; pos[1] = 0.f; pos[2] = Model::stand_height + 0.01f;
    quat[0] = 1.f; quat[1] = quat[2] = quat[3] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { v0.w[i] = 0.f; v0.v[i] = 0.f; }
    // the six Philox blocks of the reset noise: three per lane of the pair, exchanged (round 6: both lanes used to draw all six --
    // 300 more instructions on every control step in which ANY of the wavefront's 32 envs resets, i.e. on 3 steps of 4 under random
    // actions).  Integer-exact: the words are the ones env_block() would have produced.
    {
      uint32_t c3 = ctr + (side ? 3u : 0u), mine[3][4];
#pragma unroll
      for (int b = 0; b < 3; ++b) env_block(P, e, c3, mine[b]);
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t other = xchg_u32(mine[b][i]);
          rr[b][i] = side ? other : mine[b][i];
          rr[3 + b][i] = side ? mine[b][i] : other;
        }
      ctr += 6u;
    }
    z_init = pos[2];
    nn_dr = 0.75f;
    pot_prev = -planar_dist(c.p[1], pos) / kDt;
    n = 1; count = 0; elapsed = 0; flags = 0; ep_ret = 0.f; ep_lo = 0.f;
  }
  SS_PROFE(7);       // reset branch (Philox blocks)
  // joint values of this lane in the TRUE world (after the optional reset)
  float qt[NH], qdt[NH];
  static_for<0, NH>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value, jr = kHalf[k];
    constexpr int jl = jr < 3 ? jr : (jr < 8 ? jr + 5 : jr + 4);
    const float sg = mirror_flips(jr) ? m : 1.f;
    if (do_reset) {
      qt[k] = side ? reset_angle<Model, jl>(rr) : reset_angle<Model, jr>(rr);
      qdt[k] = 0.f;
    } else {
      qt[k] = sg * L.s(S_Q + k);
      qdt[k] = sg * L.s(S_QD + k);
    }
  });
  // 11. output stage (emit_outputs): inline, or on helper 1 in the three-helper rollout kernel
  SS_PROFE(8);       // joint values of the next state
  SS_FUZZ(0x62u);
  if constexpr (ROLLOUT || kOffload) {
    // the LDS copy of the state is what comes next (the next step, helper 1's output stage): refresh what the env logic changed
    if (do_reset) {
      L.s(S_POS + 0) = pos[0]; L.s(S_POS + 1) = m * pos[1]; L.s(S_POS + 2) = pos[2];
      L.s(S_QUAT + 0) = quat[0]; L.s(S_QUAT + 1) = m * quat[1]; L.s(S_QUAT + 2) = quat[2]; L.s(S_QUAT + 3) = m * quat[3];
      L.s(S_VW + 0) = m * v0.w[0]; L.s(S_VW + 1) = v0.w[1]; L.s(S_VW + 2) = m * v0.w[2];
      L.s(S_VV + 0) = v0.v[0]; L.s(S_VV + 1) = m * v0.v[1]; L.s(S_VV + 2) = v0.v[2];
      static_for<0, NH>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value, jr = kHalf[k];
        const float sg = mirror_flips(jr) ? m : 1.f;
        L.s(S_Q + k) = sg * qt[k];
        L.s(S_QD + k) = sg * qdt[k];
      });
    }
    if (advanced || do_reset) {
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        L.s(S_STP + sl * 3 + 0) = c.p[sl][0]; L.s(S_STP + sl * 3 + 1) = m * c.p[sl][1]; L.s(S_STP + sl * 3 + 2) = c.p[sl][2];
        L.s(S_STN + sl * 3 + 0) = c.nrm[sl][0]; L.s(S_STN + sl * 3 + 1) = m * c.nrm[sl][1]; L.s(S_STN + sl * 3 + 2) = c.nrm[sl][2];
      }
    }
  }
  if constexpr (kOffload) {
    L.hs(kHandOut + 0) = r;
    L.hs(kHandOut + 1) = z_init;
    L.hs(kHandOut + 2) = side ? inf.ep_ret_lo : inf.ep_ret;
    const int bits = (d ? 1 : 0) | (inf.bad_transition << 1) | (inf.update_terrain << 2) | ((do_reset ? 1 : 0) << 3) | (flags << 4) |
                     (inf.steps_reached << 8) | ((int)inf.ep_len << 16);
    L.hs(kHandOut2 + 0) = __builtin_bit_cast(float, bits);
#pragma unroll
    for (int i = 0; i < 3; ++i) { L.hs(kHandOut + 3 + i) = c.p[1][i]; L.hs(kHandOut + 8 + i) = c.p[2][i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { L.hs(kHandOut + 6 + i) = c.tilt[1][i]; L.hs(kHandOut + 11 + i) = c.tilt[2][i]; }
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (!ROLLOUT) __syncthreads();      // one launch per step: the helper emits NOW, beside this wavefront's env-level stores
#endif
  } else {
    StepOut o;
#pragma unroll
    for (int i = 0; i < 3; ++i) o.pos[i] = pos[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) o.quat[i] = quat[i];
    o.v0 = v0;
#pragma unroll
    for (int k = 0; k < NH; ++k) { o.qt[k] = qt[k]; o.qdt[k] = qdt[k]; }
    o.r = r; o.z_init = z_init; o.roll = roll; o.pitch = pitch; o.cyaw = cyaw; o.syaw = syaw;
    o.d = d ? 1 : 0; o.flags = flags; o.do_reset = do_reset ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { o.tgt[0][i] = c.p[1][i]; o.tgt[1][i] = c.p[2][i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { o.tgt[0][3 + i] = c.tilt[1][i]; o.tgt[1][3 + i] = c.tilt[2][i]; }
    o.inf = inf;
    emit_outputs<Model, ROLLOUT>(P, io, o, e, side, valid, lane, lane_global, kstep, lds);
  }
  if (valid && side == 0) {            // the env-level scalars
  SS_PROFE(9);       // LDS refresh + hand-off to the emitting helper (or the inline output stage)
    float* Fo = P.fstate + e;
    if (advanced || do_reset) {
      store_cache(P, e, c);
      store_headings(P, e, hd);
    }
    gst<!ROLLOUT>(&Fo[F_POT * np], pot_prev);
    gst<!ROLLOUT>(&Fo[F_ZINIT * np], z_init);
    gst<!ROLLOUT>(&Fo[F_EPRET * np], ep_ret);
    gst<!ROLLOUT>(&Fo[F_EPRET_LO * np], ep_lo);
    gst<!ROLLOUT>(&Fo[F_NNDR * np], nn_dr);
    gst<!ROLLOUT>(&P.istate[e + I_N * np], n);
    gst<!ROLLOUT>(&P.istate[e + I_COUNT * np], count);
    gst<!ROLLOUT>(&P.istate[e + I_ELAPSED * np], elapsed);
    gst<!ROLLOUT>(&P.istate[e + I_RNG * np], (int)ctr);
    gst<!ROLLOUT>(&P.istate[e + I_FLAGS * np], flags);
  }
  SS_MEMBAR();
#if defined(SS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
  SS_PROFE(10);      // env-level stores
  SS_PROF(13);
  if (lane == 0 && P.prof)
    for (int i = 0; i < 16; ++i) atomicAdd(P.prof + i, (unsigned long long)prof.t[i]);
#endif
  }   // control steps of this launch
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (out_offload(HELPERS, ROLLOUT) && ROLLOUT) __syncthreads();   // a helper emits the last step's outputs
#endif
}

#ifndef SS_HOST_HARNESS
// Experiment (-DSS_CODE_PREFETCH=bytes, off by default): the helper wavefronts, idle until the main wavefront has loaded
// the state, read the kernel's own code (from the entry point on) as data, so that the main wavefront's instruction fetches of a
// launch's first control step find the lines in L2 instead of HBM / MALL (a kernel boundary invalidates the L2s of all eight XCDs).
#if defined(SS_CODE_PREFETCH)
__device__ __forceinline__ void prefetch_code(unsigned long long entry_pc, int helper, int nhelpers, int lane, uint32_t* sink) {
  const uint4* p = reinterpret_cast<const uint4*>(entry_pc & ~63ull);
  uint32_t acc = 0;
#pragma unroll 4
  for (int i = helper * kWave + lane; i < (SS_CODE_PREFETCH) / 16; i += nhelpers * kWave) { const uint4 v = p[i]; acc ^= v.x ^ v.w; }
  if (acc == 0x9E3779B9u && sink) *sink = acc;       // (keeps the loads alive; practically never taken)
}
#define SS_PREFETCH_ENTRY() const unsigned long long entry_pc_ = __builtin_amdgcn_s_getpc()
#define SS_PREFETCH(helper, n, lane) prefetch_code(entry_pc_, helper, n, lane, reinterpret_cast<uint32_t*>(P.prof))
#else
#define SS_PREFETCH_ENTRY() ((void)0)
#define SS_PREFETCH(helper, n, lane) ((void)0)
#endif
// Layout experiment (-DSS_ENTRY_PAD=n / -DSS_HELPER_PAD=n: n `s_nop`s, 4 bytes each, executed once per launch at the kernel's entry /
// at the head of the helper wavefronts' branch): the helped kernels are 77-85 KB of code against a 64 KB instruction cache shared by
// two CUs, so where the main and the helper wavefronts' hot code falls in the cache is worth a few per cent.
#define SS_STR2(x) #x
#define SS_STR(x) SS_STR2(x)
#if defined(SS_ENTRY_PAD) && defined(__HIP_DEVICE_COMPILE__)
#define SS_PAD_ENTRY() asm volatile(".rept " SS_STR(SS_ENTRY_PAD) "\n s_nop 0\n .endr")
#else
#define SS_PAD_ENTRY() ((void)0)
#endif
#if defined(SS_HELPER_PAD) && defined(__HIP_DEVICE_COMPILE__)
#define SS_PAD_HELPER() asm volatile(".rept " SS_STR(SS_HELPER_PAD) "\n s_nop 0\n .endr")
#else
#define SS_PAD_HELPER() ((void)0)
#endif
template <class Model, bool RANDOM_ACT>
__global__ __launch_bounds__(kWave, 1) void step_kernel(Params P, StepIO io) {
  __shared__ float4 lds4[kLdsSlots * kWave];
  step_env<Model, RANDOM_ACT>(P, io, blockIdx.x * kWave + threadIdx.x, threadIdx.x, reinterpret_cast<float*>(lds4));   // lane = 2*env + side
}
// Small-batch variant: 1 + HELPERS wavefronts per 32 envs.  Wavefront 0 runs the step as above; the helper wavefront(s)
// compute the contact-space operators of every substep concurrently on the CU's other SIMDs (ss_dynamics.hpp:
// helper_substep).  Worth it only while the batch leaves SIMDs idle (4096 envs occupy 128 of 1024).
template <class Model, bool RANDOM_ACT, int HELPERS>
__global__ __launch_bounds__(kWave * (1 + HELPERS), 1) void step_kernel_helped(Params P, StepIO io) {
  SS_PREFETCH_ENTRY();
  SS_PAD_ENTRY();
  __shared__ float4 lds4[(kLdsSlots + hand_slots(HELPERS)) * kWave];      // (three helpers: + the rows' own 40 words per lane)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & (kWave - 1);
  float* lds = reinterpret_cast<float*>(lds4);
  SS_FUZZ(0x70u + wave);
  if (wave == 0) {
    step_env<Model, RANDOM_ACT, HELPERS>(P, io, blockIdx.x * kWave + lane, lane, lds);
  } else {
    SS_PREFETCH(wave - 1, HELPERS, lane);
    SS_PAD_HELPER();
    const Lds L{lds, lane};
#pragma unroll 1
    for (int k = 0; k < SS_NUM_SUBSTEPS; ++k) helper_substep<Model, HELPERS>(wave - 1, L, [](int) {});
    if constexpr (out_offload(HELPERS, false)) {     // (tuning option) the output stage on a helper, beside the main wavefront's own stores
      __syncthreads();
      if (wave == HELPERS) emit_from_handoff<Model, false>(P, io, L, lane, blockIdx.x * kWave + lane, 0, lds);
    }
  }
}
// K control steps per launch (ss_rollout_random): same code, state resident in LDS between the steps
template <class Model>
__global__ __launch_bounds__(kWave, 1) void rollout_kernel(Params P, StepIO io) {
  __shared__ float4 lds4[kLdsSlots * kWave];
  step_env<Model, true, 0, true>(P, io, blockIdx.x * kWave + threadIdx.x, threadIdx.x, reinterpret_cast<float*>(lds4));
}
template <class Model, int HELPERS>
__global__ __launch_bounds__(kWave * (1 + HELPERS), 1) void rollout_kernel_helped(Params P, StepIO io) {
  SS_PREFETCH_ENTRY();
  SS_PAD_ENTRY();
  __shared__ float4 lds4[(kLdsSlots + hand_slots(HELPERS)) * kWave];      // (three helpers: + the rows' own 40 words per lane)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & (kWave - 1);
  float* lds = reinterpret_cast<float*>(lds4);
  SS_FUZZ(0x70u + wave);
  if (wave == 0) {
    step_env<Model, true, HELPERS, true>(P, io, blockIdx.x * kWave + lane, lane, lds);
  } else {
    SS_PREFETCH(wave - 1, HELPERS, lane);
    SS_PAD_HELPER();
    const Lds L{lds, lane};
    const int lane_global = blockIdx.x * kWave + lane, side = lane_global & 1;
    const int e = min(lane_global >> 1, P.n - 1);
    const float m = side ? -1.f : 1.f;
#pragma unroll 1
    for (int kstep = 0; kstep < io.nsteps; ++kstep)
#pragma unroll 1
      for (int k = 0; k < SS_NUM_SUBSTEPS; ++k)
        helper_substep<Model, HELPERS>(wave - 1, L, [&](int helper) {
          if (k != 0) return;
          SS_FUZZ(0x80u + helper);
          // the next control step's actions, while the main wavefront is in pass 1 / 2 of this step's first substep
          // (three helpers: helper 1 -- which also has the spine's bias forces in this window -- draws the actions, helper 2 has the
          // longer job, the previous step's outputs, to itself; rounds 3-5 had them the other way round)
          constexpr int kActHelper = SS_EMIT_ON_LAST_HELPER && out_offload(HELPERS, true) ? 1 : HELPERS - 1;
          constexpr int kEmitHelper = SS_EMIT_ON_LAST_HELPER && out_offload(HELPERS, true) ? HELPERS - 1 : 1;
          if (HELPERS > 1 && helper == kActHelper && kstep + 1 < io.nsteps)
            random_actions_half<true>(P, e, side, m, (uint32_t)io.t + (uint32_t)kstep + 1u, [&](int j, float a) { L.hs(kHandAct + j) = a; });
          // the previous control step's outputs
          if (out_offload(HELPERS, true) && helper == kEmitHelper && kstep > 0) emit_from_handoff<Model, true>(P, io, L, lane, lane_global, kstep - 1, lds);
        });
    if constexpr (out_offload(HELPERS, true)) {
      __syncthreads();                 // the last step's results are in the hand-off region
      SS_FUZZ(0x90u + wave);
      if (wave == 2) emit_from_handoff<Model, true>(P, io, L, lane, lane_global, io.nsteps - 1, lds);
    }
  }
}
#endif  // SS_HOST_HARNESS

// (The non-template kernels of this header are `static`: the header is included by two translation units, ss_api.hip and
// ss_rollout3.hip, and only ss_api.hip launches them.)
// Consumer side of the peer-store all-gather: lane r waits until peer r has published `value` (or a later step) in this
// rank's flag array.  Bounded spin: on time-out it raises *error instead of hanging the GPU.
#ifndef SS_HOST_HARNESS
static __global__ void peer_wait_kernel(const uint32_t* flags, int count, uint32_t value, uint32_t* error) {
  const int r = threadIdx.x;
  if (r >= count) return;
  for (long long it = 0; it < (1ll << 23); ++it) {     // ~1 s
    const uint32_t v = __hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int32_t)(v - value) >= 0) return;
    __builtin_amdgcn_s_sleep(8);
  }
  *error = 1u + (uint32_t)r;
}
#endif

// hook updates, stream-ordered (ss_api.hip)
#ifndef SS_HOST_HARNESS
static __global__ void set_knobs_kernel(Knobs* dst, Knobs v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}
// per-env sampling grids: [N][121] row-major (the caller's layout, playground/train.py:267-271) -> [121][Npad]
static __global__ void transpose_prob_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int npad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * SS_NCELL) return;
  const int e = i / SS_NCELL, k = i - e * SS_NCELL;
  dst[(size_t)k * npad + e] = src[i];
}
static __global__ void copy_prob_kernel(const float* __restrict__ src, float* __restrict__ dst) {
  if (threadIdx.x < SS_NCELL) dst[threadIdx.x] = src[threadIdx.x];
}
#endif

#ifndef SS_HOST_HARNESS
template <class Model>
__global__ __launch_bounds__(kWave) void reset_kernel(Params P, float* obs) {
  const int e = blockIdx.x * kWave + threadIdx.x;
  if (e >= P.n) return;
  const size_t np = (size_t)P.npad;
  Dyn s;
  Cache c;
  uint32_t ctr = (uint32_t)P.istate[e + I_RNG * np];
  float pot, z_init, nn_dr;
  env_reset<Model>(P, e, s, c, ctr, pot, z_init, nn_dr);
  store_dyn(P, e, s);
  store_cache(P, e, c);
  const float straight[3][2] = {{1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}};      // the provisional path runs along +x
  store_headings(P, e, straight);
  P.fstate[e + F_POT * np] = pot;
  P.fstate[e + F_ZINIT * np] = z_init;
  P.fstate[e + F_EPRET * np] = 0.f;
  P.fstate[e + F_EPRET_LO * np] = 0.f;
  P.fstate[e + F_NNDR * np] = nn_dr;
  P.istate[e + I_N * np] = 1;
  P.istate[e + I_COUNT * np] = 0;
  P.istate[e + I_ELAPSED * np] = 0;
  P.istate[e + I_RNG * np] = (int)ctr;
  P.istate[e + I_FLAGS * np] = 0;
  if (obs) {
    float o[SS_OBS_DIM];
    write_obs<Model>(s, z_init, 0, c, o);
#pragma unroll
    for (int i = 0; i < SS_OBS_DIM; ++i) obs[(size_t)e * SS_OBS_DIM + i] = o[i];
  }
}
#endif  // SS_HOST_HARNESS

#ifndef SS_HOST_HARNESS
template <class Model>
__global__ __launch_bounds__(kWave) void obs_kernel(Params P, float* obs) {
  const int e = blockIdx.x * kWave + threadIdx.x;
  if (e >= P.n) return;
  const size_t np = (size_t)P.npad;
  Dyn s;
  Cache c;
  load_dyn(P, e, s);
  load_cache(P, e, c);
  float o[SS_OBS_DIM];
  write_obs<Model>(s, P.fstate[e + F_ZINIT * np], P.istate[e + I_FLAGS * np], c, o);
#pragma unroll
  for (int i = 0; i < SS_OBS_DIM; ++i) obs[(size_t)e * SS_OBS_DIM + i] = o[i];
}
#endif  // SS_HOST_HARNESS

// one thread per (env, grid cell): PHYSICS.md section 8
#ifndef SS_HOST_HARNESS
// create_temp_states (common/envs_utils.py:573-578, playground/train.py:247-257): per env the 121 variants of the
// current observation with the look-ahead stone moved to each (yaw, pitch) grid cell.  Only obs[55..59] differ:
// obs_kernel first writes the current observation rows (lane per env, coalesced state loads) to a scratch [N,60];
// then one 256-thread workgroup per env computes the 121 x 5 target features (one lane per cell) and streams the
// [121,60] block out as 1815 coalesced float4 -- the one HBM-bound kernel of the path (29 KB written per env).
#ifndef SS_TEMP_THREADS
#define SS_TEMP_THREADS 240            // a multiple of 15: every thread keeps ONE float4 column of the row
#endif
constexpr int kTempThreads = SS_TEMP_THREADS;
static __global__ __launch_bounds__(kTempThreads) void temp_states_kernel(Params P, const float* __restrict__ obs_rows, float* out) {
  __shared__ __attribute__((aligned(16))) float base[SS_OBS_DIM];
  __shared__ float feat[SS_NCELL * 5];
  const int e = blockIdx.x, t = threadIdx.x;
  const size_t np = (size_t)P.npad;
  if (t < SS_OBS_DIM) base[t] = obs_rows[(size_t)e * SS_OBS_DIM + t];
  if (t < SS_NCELL) {
    const int cell = t;
    float pos[3], quat[4], p1[3], p2[3], tilt2[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) pos[i] = P.fstate[e + (F_POS + i) * np];
#pragma unroll
    for (int i = 0; i < 4; ++i) quat[i] = P.fstate[e + (F_QUAT + i) * np];
#pragma unroll
    for (int i = 0; i < 3; ++i) { p1[i] = P.fstate[e + (F_STONE + 8 + i) * np]; p2[i] = P.fstate[e + (F_STONE + 16 + i) * np]; }
    tilt2[0] = P.fstate[e + (F_STONE + 16 + 6) * np];
    tilt2[1] = P.fstate[e + (F_STONE + 16 + 7) * np];
    const int n = P.istate[e + I_N * np];
    if (n + 1 <= kNumStones - 1) {
      const float* T = P.terrain + e;
      const float phi_n = n < P.istate[e + I_PROV * np] ? T[(n * 6 + 3) * np] : 0.f;      // a provisional stone is not stored
      float phi = phi_n + yaw_sample(cell / SS_GRID), pitch = pitch_sample(cell % SS_GRID);
      float dr = P.fstate[e + F_NNDR * np];
      float sp, cp, sph, cph;
      sincosf(pitch, &sp, &cp);
      sincosf(phi, &sph, &cph);
      float planar = dr * cp;
      p2[0] = p1[0] + planar * cph;
      p2[1] = p1[1] + planar * sph;
      p2[2] = p1[2] + dr * sp;
    }
    float cyaw, syaw;                      // yaw only: (cos, sin) = (A, B) / |(A, B)| as in quat_roll_pitch_cs
    {
      const float A = 1.f - 2.f * (quat[2] * quat[2] + quat[3] * quat[3]), B = 2.f * (quat[0] * quat[3] + quat[1] * quat[2]);
      const float n2 = A * A + B * B, inv = rsqrtf(fmaxf(n2, 1e-30f));
      cyaw = n2 > 1e-30f ? A * inv : 1.f;
      syaw = n2 > 1e-30f ? B * inv : 0.f;
    }
    float f[5];
    target_features(pos, cyaw, syaw, p2, tilt2, f);
#pragma unroll
    for (int i = 0; i < 5; ++i) feat[cell * 5 + i] = f[i];
  }
  __syncthreads();
  // Measured in round 2 (profiles/r02_*_temp_states.txt): this one-workgroup-per-env shape writes 5.0-5.3 TB/s at 32768
  // envs (a torch fill of the same buffer: 6.9 TB/s) -- and it stays there with the feature computation removed, with 60-
  // or 120-thread workgroups, with persistent workgroups (4.1-5.0 TB/s) and with one workgroup per 16 rows (1.8 TB/s,
  // latency-bound): the limit is the write pattern of 29,040-byte blocks, not the prologue.
  constexpr int kRow4 = SS_OBS_DIM / 4;                      // 15 float4 per row
  static_assert(kTempThreads % kRow4 == 0, "a thread must stay in its column");
  constexpr int kRowsPerPass = kTempThreads / kRow4;
  float4* o4 = reinterpret_cast<float4*>(out) + (size_t)e * (SS_NCELL * kRow4);
  const float4* b4 = reinterpret_cast<const float4*>(base);
  const int c4 = t % kRow4;
  const float4 bv = b4[c4 < kRow4 - 1 ? c4 : kRow4 - 2];     // this thread's column of the common part, in registers
#pragma unroll 1
  for (int row = t / kRow4; row < SS_NCELL; row += kRowsPerPass) {
    float4 v = bv;
    const float* f = feat + row * 5;
    if (c4 == kRow4 - 2) v.w = f[0];                         // obs[52..54], obs[55]
    if (c4 == kRow4 - 1) v = make_float4(f[1], f[2], f[3], f[4]);
    o4[row * kRow4 + c4] = v;          // plain stores: nontemporal ones measured 20 % slower here
  }
}
#endif  // SS_HOST_HARNESS

#ifndef SS_HOST_HARNESS
static __global__ void random_actions_kernel(Params P, uint64_t t, float* act) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.n) return;
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    uint32_t r[4];
    philox4x32_10((uint32_t)(6u * (uint32_t)t + b), 1u, P.env_offset + ((uint32_t)e & P.id_mask), 0u, P.seed_lo, P.seed_hi, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int j = b * 4 + i;
      if (j < NJ) act[(size_t)e * NJ + j] = 2.f * u01(r[i]) - 1.f;
    }
  }
}
#endif  // SS_HOST_HARNESS

#ifndef SS_HOST_HARNESS
// PMC calibration: a dword-per-lane coalesced copy with the step kernel's access shape (tools/hbm_traffic.py)
static __global__ void calib_copy_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1.0f;
}
#endif

// packed [N,186] <-> structure of arrays (PHYSICS / include/steppingstone.h layout)
SSD void pack_env(const Params& P, int e, float* packed) {
  const size_t np = (size_t)P.npad;
  float* o = packed + (size_t)e * SS_STATE_DIM;
  for (int i = 0; i < 59; ++i) o[i] = P.fstate[e + (size_t)i * np];
  o[59] = (float)P.istate[e + I_N * np];
  o[60] = (float)P.istate[e + I_COUNT * np];
  o[61] = (float)P.istate[e + I_ELAPSED * np];
  uint32_t ctr = (uint32_t)P.istate[e + I_RNG * np];
  o[62] = (float)(ctr & 0xFFFFu);
  o[63] = (float)(ctr >> 16);
  o[64] = (float)P.istate[e + I_FLAGS * np];
  const int prov = P.istate[e + I_PROV * np];
  for (int i = 0; i < 120; ++i) {
    const int k = i / 6;
    o[65 + i] = k < prov ? P.terrain[e + (size_t)i * np] : (i % 6 == 0 ? 0.75f * (float)k : 0.f);
  }
  o[185] = P.fstate[e + (size_t)F_EPRET_LO * np];
}
SSD void unpack_env(const Params& P, int e, const float* packed) {
  const size_t np = (size_t)P.npad;
  const float* o = packed + (size_t)e * SS_STATE_DIM;
  for (int i = 0; i < 59; ++i) P.fstate[e + (size_t)i * np] = o[i];
  int n = (int)o[59];
  P.istate[e + I_N * np] = n;
  P.istate[e + I_COUNT * np] = (int)o[60];
  P.istate[e + I_ELAPSED * np] = (int)o[61];
  P.istate[e + I_RNG * np] = (int)((uint32_t)o[62] | ((uint32_t)o[63] << 16));
  P.istate[e + I_FLAGS * np] = (int)o[64];
  for (int i = 0; i < 120; ++i) P.terrain[e + (size_t)i * np] = o[65 + i];
  P.istate[e + I_PROV * np] = kNumStones;        // an injected terrain is stored in full
  P.fstate[e + (size_t)F_EPRET_LO * np] = o[185];
  Cache c;
  float hd[3][2];
  cache_from_terrain(P, e, n, c, hd);
  store_cache(P, e, c);
  store_headings(P, e, hd);
}
#ifndef SS_HOST_HARNESS
static __global__ void pack_state_kernel(Params P, float* packed) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < P.n) pack_env(P, e, packed);
}
#endif  // SS_HOST_HARNESS
#ifndef SS_HOST_HARNESS
static __global__ void unpack_state_kernel(Params P, const float* packed) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < P.n) unpack_env(P, e, packed);
}
#endif  // SS_HOST_HARNESS

}  // namespace ss
