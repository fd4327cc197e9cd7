// ss_dynamics.hpp -- one physics substep (docs/PHYSICS.md section 3), TWO LANES PER ENVIRONMENT.
//
// Lane layout ("half-body lanes").  Lane 2e simulates the right half of env e, lane 2e+1 the left half.  Each lane
// owns the 12 joints {spine 0,1,2 | leg 3..7 | arm 13..16} of ITS half and the left lane works in the y-mirrored
// world: base pose, stones, joint angles and actions are reflected on load, so that both lanes execute the very
// same right-side code with the same compile-time constants (the robots are mirror symmetric; the generator
// asserts it).  Spine, torso and the 6x6 base solve are computed redundantly by both lanes; what crosses the lane
// pair (one DPP/shuffle exchange each, mirrored on receipt) is
//     pass 2: the leg's articulated inertia + bias at the pelvis, the arm's at the torso      (2 x 27 floats)
//     detect: "any contact" flag
//     operators: the pelvis-twist columns G of the own foot's unit impulses, once per substep  (36 floats)
//     PGS:    per sweep the other foot's wrench increments                                     (6 floats)
//     final:  the pelvis bias impulse of the other leg                                         (6 floats)
// The 4096-env workload therefore runs on 128 wavefronts of 32 envs and every serial chain (ABA sweeps, Lambda^-1
// columns, contact rows, PGS) is half as long as with one env per lane.
//
// Per lane: up to 512 VGPR+AGPR and a private 640-byte share of LDS (40 KiB per wavefront, one wavefront per
// workgroup, four workgroups = one per SIMD on a CU; no barriers: lanes only read their own columns).  Region A
// (20 float4-slots): the contact operators C (own-foot twist per unit impulse on the partner's foot) and T, 6 columns x 3
// float2 items each, 8-byte lane stride (ds_*_b64); afterwards the staging area of the step's output rows.
// Region B (scalars, element-major, ds_*_b32): actions, q, qd, free qd, base pose + twist, stones.  The link twists,
// the 12 joint records, the 12 PGS rows and Lambda_own live in registers (the compiler parks them in AGPRs).
#pragma once
#include "ss_math.hpp"
#include "ss_pair.hpp"

namespace ss {

constexpr float kH = 1.0f / 240.0f;
constexpr float kDt = 1.0f / 60.0f;
constexpr float kGrav = 9.8f;
// the stepping surface of a stone (PHYSICS.md 3.3, round 6): a plank whose footprint, seen from above, is 2 kPlankA x 2 kPlankB, aligned
// with the stone's heading (ss_model_tables.hpp: kStonePlankHalfLength / kStonePlankHalfWidth)
constexpr float kPlankA = kStonePlankHalfLength, kPlankB = kStonePlankHalfWidth;
// PHYSICS.md 3.4: 5 sweeps, warm-started from the previous substep of the same control step (Bullet's
// numSolverIterations = 5 with warm starting; rounds 1-4 ran 8 cold sweeps -- DESIGN.md section 5.1 has the measured trade)
#ifndef SS_PGS_ITERS
#define SS_PGS_ITERS 5
#endif
#ifndef SS_PGS_WARM
#define SS_PGS_WARM 1
#endif
constexpr int kPgsIters = SS_PGS_ITERS;
constexpr bool kPgsWarm = SS_PGS_WARM != 0;
constexpr float kErp = 0.2f;
constexpr float kSlop = 0.001f;
constexpr float kVcorrMax = 2.0f;

// The PGS sweep, the row set-up and the contact operators run in packed f32 (v_pk_fma_f32); the scalar / Omega-recursion
// variants of rounds 1 (v7) were removed in round 2 -- they are in the git history with their measurements.

constexpr int kWave = 64;
constexpr int kEnvsPerWave = 32;
constexpr int kLdsSlots = 40;          // 40 float4 = 640 B per lane = 40,960 B per wavefront: four wavefronts (one per SIMD) per CU
constexpr int kSlotsA = 20;            // region A: the contact operators C and T (72 floats per lane), output staging
constexpr int kLdsC = 0;               // 6 columns x 3 float2: own-foot twist per unit impulse on the PARTNER's foot (T . mirror(G_partner))
constexpr int kLdsT = 18;              // 6 columns x 3 float2: own-foot twist per unit pelvis twist
constexpr int kLdsHead = 36;           // 3 float2: (cos, sin) of the heading of the active stones n-1, n, n+1 in this lane's world; items 36..38 of
                                       // region A's 40 are beyond the operators (0..35) and beyond the output staging (the first 2240 floats)
constexpr int kScalarBase = kSlotsA * kWave * 4;   // region B, in floats
// helper-wavefront variant (small batches): a hand-off region behind the main wavefront's 40 slots -- the joint records
// of the spine+leg chain (8 x 9 floats), the base Cholesky factor (21), and back: the six Lambda_own columns (36)
constexpr int kHandJc = 0, kHandL0 = 72, kHandLc = 93, kHandDet = 129, kHandAct = 146, kHandFloats = 158, kHandCs = kHandLc;   // Det: Rf 9, pen 4 (single-helper variant only), flags (active | cslot << 4 | contact << 12 | on_target << 13), sole 3; Act: the next step's 12 actions (rollout kernel); Cs: cos[12], sin[12] of the joint angles, aliasing Lc (dead between barriers #0 and #2)
// The directions of the 12 contact rows (36 floats) and the 4 Baumgarte terms, computed by helper 0: they take the place of the
// leg joint records once every helper has loaded those (written after barrier #2, read by the main wavefront after #3).
// The velocity-product bias forces of the massive spine bodies (3, 2, 1) and of the base, computed by helper 1 between barriers
// #0b and #1 (three-helper variant): 4 x 6 floats in the place of the spine joint records, which the main wavefront writes only
// at the end of the #1 -> #2 window (after it has read the biases).
constexpr bool bias_offload(int helpers) { return helpers >= 3; }
constexpr int kHandBias = kHandJc;
static_assert(24 <= 3 * 9, "biases fit the spine records' place");
constexpr int kHandRows = kHandJc + 3 * 9;
// Three-helper variants (at most 8192 envs: one workgroup per CU): the rows get a place of their own behind the region, so that
// helper 0 can write them BEFORE barrier #2 (as soon as it has them, round 6) and the main wavefront can fetch them and form their
// moment parts while it waits at barrier #3 for the operators instead of after it.
constexpr int kHandRows3 = 160;
constexpr int kHandFloats3 = kHandRows3 + 40;
constexpr int kHandSlots3 = (kHandFloats3 + 3) / 4;
constexpr int hand_slots(int helpers) { return helpers >= 3 ? kHandSlots3 : (kHandFloats + 3) / 4; }
// ... with three helpers; a single helper is the critical path in its windows already (16384 envs: 0.0676 vs 0.0609 ms/step)
constexpr bool rows_offload(int helpers) { return helpers >= 3; }
constexpr int kHandSlots = (kHandFloats + 3) / 4;  // float4-slots per lane
// (a larger hand-off region is not free: 81 KiB per workgroup cost 6 us per launch, 99 KiB 14 us -- measured)
static_assert(kLdsSlots + kHandSlots <= 80, "two helper-variant workgroups must fit the 160 KiB of a CU (16384 envs: 512 workgroups)");
static_assert(kHandRows + 40 <= kHandJc + 8 * 9, "rows fit the leg records' place");
constexpr int kHandBase = kLdsSlots * kWave * 4;   // in floats
constexpr int NH = 12;                 // joints per half
enum { S_ACT = 0, S_Q = 12, S_QD = 24, S_WLAM = 36, S_POS = 48, S_QUAT = 51, S_VW = 55, S_VV = 58, S_STP = 61, S_STN = 70,
       S_WKEY = 79, S_END = 80 };
// S_WLAM / S_WKEY: the warm-start impulses (4 corners x 3) and their key, for the variants whose registers are full (below);
// rounds 1-2 kept the free joint velocities there, they have lived in registers since
static_assert(S_END <= (kLdsSlots - kSlotsA) * 4, "LDS scalar region overflow");
// Where the warm-start impulses live between the substeps of a control step: in registers where there is room (three helper
// wavefronts: 192 - 205 of 256 AGPRs), in the lane's LDS scalars otherwise (the plain and the one-helper rollout kernels sit at
// 252 - 255 AGPRs and would spill 20 - 40 B per lane; 13 ds_write + 13 ds_read per substep instead).  Values are the same either way.
#ifndef SS_CS_PER_HELPER
#define SS_CS_PER_HELPER 6      // three-helper variants: 6 = helpers 1 and 2 evaluate six cos / sin pairs each, helper 0 idles; 4 = all three
                                // take four each (ss_rollout3.hip sets it for the rollout kernel: measured per kernel, see there)
#endif
#ifndef SS_WARM_LDS_BELOW
#define SS_WARM_LDS_BELOW 3
#endif
constexpr bool warm_in_lds(int helpers) { return helpers < SS_WARM_LDS_BELOW; }

// the half-tree: global (right-side) joint ids, spine first
constexpr int kHalf[NH] = {0, 1, 2, 3, 4, 5, 6, 7, 13, 14, 15, 16};
// joints whose angle changes sign under the y-mirror (rotation about x or z)
constexpr bool mirror_flips(int j) { return kAxis[j] != 1; }
// Policy coordinates (PHYSICS.md 2, kPolicySign): actions and observations carry sigma_j x (value about the +axis).  For the limbs
// sigma is one physical convention on both sides (the generator asserts sigma[left] = sigma[right] for y joints and -sigma[right] for
// x / z joints: the left one is measured about the mirrored axis), and the left lane's mirrored world holds exactly "about the
// mirrored axis" -- so BOTH lanes convert between their own world and policy coordinates with the RIGHT twin's sigma; the spine
// (sigma = +1) changes sign in the left lane for its z / x joints only.  The same factor serves actions in and observations out.
__host__ __device__ constexpr float action_lane_sign(int jr, float m) {
  return jr < 3 ? (mirror_flips(jr) ? m : 1.f) : (float)kPolicySign[jr];
}
// TRUE-world value (about the +axis, as the state arrays hold it) -> policy coordinates: sigma of the joint itself
__host__ __device__ constexpr float policy_true_sign(int jr, int side) {
  return jr < 3 ? 1.f : (float)kPolicySign[jr] * ((mirror_flips(jr) && side) ? -1.f : 1.f);
}
// highest-index child of body b inside the half-tree; -1 for leaves
constexpr int first_child_half(int b) {
  int r = -1;
  for (int i = 0; i < NH; ++i)
    if (kParent[kHalf[i]] == b) r = kHalf[i] + 1;
  return r;
}

#define SS_MEMBAR() asm volatile("" ::: "memory")
// Between "every lane has written its LDS staging rows" and "the wavefront copies them out together".  On the device a
// wavefront runs in lockstep and its LDS operations complete in order, so this is a compiler barrier; the host harness runs
// the 64 lanes as threads and needs a real one.
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_WAVE_SYNC() asm volatile("" ::: "memory")
#else
#define SS_WAVE_SYNC() ss_host_wave_sync()
#endif

// optional per-phase cycle accounting (-DSS_PROFILE_PHASES; tuning builds only)
#if defined(SS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
struct Prof { uint32_t t[16]; uint32_t last; };
#define SS_PROF_DECL Prof& prof,
#define SS_PROF_ARG prof,
// wave-uniform accounting (readfirstlane keeps the counters in SGPRs, so divergent branches do not skew them)
#define SS_PROF_RAW(i) do { uint32_t _n = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_amdgcn_s_memtime()); \
    prof.t[i] = __builtin_amdgcn_readfirstlane(prof.t[i] + (_n - __builtin_amdgcn_readfirstlane(prof.last))); prof.last = _n; } while (0)
#ifdef SS_PROFILE_EPILOGUE      // the substeps as one figure (slot 0), the control step's epilogue in detail (SS_PROFE, slots 1..15)
#define SS_PROF(i) SS_PROF_RAW(0)
#define SS_PROFE(i) SS_PROF_RAW(i)
#else
#define SS_PROF(i) SS_PROF_RAW(i)
#define SS_PROFE(i) ((void)0)
#endif
#else
#define SS_PROFE(i) ((void)0)
struct Prof { int unused; };
#define SS_PROF_DECL
#define SS_PROF_ARG
#define SS_PROF(i) ((void)0)
#endif

// Schedule fuzzing (-DSS_FUZZ_SCHED; test builds only, tools/sched_fuzz.py): every wavefront sleeps a pseudo-random time at the
// start of each barrier window (and at the other hand-over points of a control step), so that the relative timing of the main
// and the helper wavefronts differs from launch to launch and from window to window.  Nothing but the barriers may order the
// hand-off region: the fuzzed build must produce the bits of the plain build.
#if defined(SS_FUZZ_SCHED) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void ss_fuzz(uint32_t site) {
  uint32_t h = (uint32_t)__builtin_amdgcn_s_memtime() ^ (site * 0x9E3779B9u);     // differs per wavefront, launch and visit
  h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  h = __builtin_amdgcn_readfirstlane(h);
  // one visit in four: up to 255 x 64 clocks (longer than the longest window); the others up to 15 x 64 clocks
  const uint32_t n = ((h >> 8) & 3u) == 0u ? (h & 255u) : (h & 15u);
#pragma unroll 1
  for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
#define SS_FUZZ(site) ss_fuzz(site)
#else
#define SS_FUZZ(site) ((void)0)
#endif

// cos / sin of the joint angles on the helper wavefronts (and the joint torques ahead of them on the main one) from this many
// helpers on.  Measured: with three helpers 0.0598 -> 0.0581 ms/step at 4096 envs; a single helper that also evaluates the 12
// cos / sin is slower than leaving them on the main wavefront (16384 envs, rollout kernel: 0.0631 vs 0.0608 ms/step).
#ifndef SS_CS_OFFLOAD_MIN
#define SS_CS_OFFLOAD_MIN 3
#endif
constexpr bool cs_offload(int helpers) { return helpers >= SS_CS_OFFLOAD_MIN; }

struct Lds {       // lane-private view of the workgroup's LDS
  float* base;
  int lane;
  SSD float2& q2(int item) const { return reinterpret_cast<float2*>(base)[item * kWave + lane]; }   // region A, 8-B items
  SSD float& s(int idx) const { return base[kScalarBase + idx * kWave + lane]; }   // region B scalar
  SSD float& hs(int idx) const { return base[kHandBase + idx * kWave + lane]; }    // hand-off scalar (helper variant)
};

struct Stones {    // the three active stones n-1, n, n+1: centre, unit normal, tilts (x, y)
  float p[3][3], nrm[3][3], tilt[3][2];
};

struct FootReport {   // this lane's foot
  int contact;        // has a contact
  int on_target;      // touches stone n (slot 1)
  float sole[3];      // sole centre, this lane's world
};

struct Warm {      // this lane's foot: the contact impulses at the end of the previous substep of this control step
  float lam[4][3];
  int key;          // bit k: corner k was in contact (key = 0: nothing to start from)
};
SSD void warm_clear(Warm& w) {
#pragma unroll
  for (int k = 0; k < 4; ++k) w.lam[k][0] = w.lam[k][1] = w.lam[k][2] = 0.f;
  w.key = 0;
}

struct JRec {      // what the ABA leaves behind per joint
  float cs, sn, Uw[3], Uv[3], Dinv, u;
};
struct JointCache {
  JRec r[NH];      // indexed by position in kHalf
  Chol6 L0;
};
constexpr int half_pos(int j) { return j <= 7 ? j : j - 5; }   // 13..16 -> 8..11

// ---- lane-pair exchange.  Partner data lives in the mirrored world: reflect on receipt.
// On the device this is a DPP quad_perm [1,0,3,2] move (full VALU rate, no LDS round trip like ds_bpermute).
SSD float xchg(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
#else
  return ss_host_xchg(x);
#endif
}
SSD uint32_t xchg_u32(uint32_t x) {      // all 32 bits, unchanged (random words)
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
#else
  return __builtin_bit_cast(uint32_t, ss_host_xchg(__builtin_bit_cast(float, x)));
#endif
}
SSD int xchg_i(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
#else
  return (int)ss_host_xchg((float)x);
#endif
}
// y-mirror of a spatial vector: polar part (x,-y,z), axial part (-x,y,-z)
SSD SV mirror_sv(const SV& a) {
  SV o = {{-a.w[0], a.w[1], -a.w[2]}, {a.v[0], -a.v[1], a.v[2]}};
  return o;
}
SSD SV xchg_sv(const SV& a) {
  SV t;
#pragma unroll
  for (int i = 0; i < 3; ++i) { t.w[i] = xchg(a.w[i]); t.v[i] = xchg(a.v[i]); }
  return mirror_sv(t);
}
SSD ABI xchg_abi(const ABI& a) {
  ABI o;
  // A: sign s_w[i] s_w[j], s_w = (-,+,-): xy and yz flip.  C: s_v = (+,-,+): xy and yz flip.  B: s_w[i] s_v[j].
  constexpr float sa[6] = {1.f, 1.f, 1.f, -1.f, 1.f, -1.f};
#pragma unroll
  for (int i = 0; i < 6; ++i) { o.A.m[i] = sa[i] * xchg(a.A.m[i]); o.C.m[i] = sa[i] * xchg(a.C.m[i]); }
  constexpr float sw[3] = {-1.f, 1.f, -1.f}, sv[3] = {1.f, -1.f, 1.f};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o.B[i][j] = (sw[i] * sv[j]) * xchg(a.B[i][j]);
  return o;
}

// cos/sin on the reduced range with Cody-Waite reduction; |error| ~1e-7 for the |x| < 1e3 the joints can reach.
// (libm's sincosf inlines a Payne-Hanek slow path per call: 21 copies of it were 3000 instructions of the kernel)
SSD void ss_sincos(float x, float& s, float& c) {
  float k = rintf(x * 0.6366197723675814f);
  float r = fmaf(k, -1.5707962512969971f, x);
  r = fmaf(k, -7.5497894158615964e-08f, r);
  float r2 = r * r;
  float sp = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * -1.9515295891e-4f));
  float cp = 1.0f - 0.5f * r2 + r2 * r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
  int n = (int)k;
  float ss_ = (n & 1) ? cp : sp, cc = (n & 1) ? sp : cp;
  s = (n & 2) ? -ss_ : ss_;
  c = ((n + 1) & 2) ? -cc : cc;
}

SSD SV base_twist(const Lds& L) {
  SV v = {{L.s(S_VW), L.s(S_VW + 1), L.s(S_VW + 2)}, {L.s(S_VV), L.s(S_VV + 1), L.s(S_VV + 2)}};
  return v;
}

// one step of the impulse recursion towards the root through joint J: returns the bias impulse at the parent
template <class Model, int J>
SSD SV imp_up(const JointCache& jc, float* ul, const SV& p) {
  constexpr int ax = kAxis[J], k = half_pos(J);
  const JRec& r = jc.r[k];
  float u = -p.w[ax];
  ul[k] = u;
  float du = r.Dinv * u;
  SV pa;
#pragma unroll
  for (int i = 0; i < 3; ++i) { pa.w[i] = p.w[i] + r.Uw[i] * du; pa.v[i] = p.v[i] + r.Uv[i] * du; }
  return xforce<Model, J>(r.cs, r.sn, pa);
}
// one step away from the root through joint J; LOADED says whether ul[] holds a bias for this joint
template <class Model, int J, bool LOADED>
SSD SV imp_down(const JointCache& jc, const float* ul, const SV& dpar, float* dq_out) {
  constexpr int ax = kAxis[J], k = half_pos(J);
  const JRec& r = jc.r[k];
  SV d = xmotion<Model, J>(r.cs, r.sn, dpar);
  float dotv = r.Uw[0] * d.w[0] + r.Uw[1] * d.w[1] + r.Uw[2] * d.w[2] + r.Uv[0] * d.v[0] + r.Uv[1] * d.v[1] +
               r.Uv[2] * d.v[2];
  float dq;
  if constexpr (LOADED) dq = r.Dinv * (ul[k] - dotv);
  else dq = -r.Dinv * dotv;
  d.w[ax] += dq;
  if (dq_out) *dq_out = dq;
  return d;
}

SSD JRec opaque_rec(const JRec& r) {
  // through local scalars, field by field: with the asm operands inside a struct copy one record of the helpers' part A stayed in
  // scratch (32 B per lane stored and re-loaded every substep, and written back at the end of every launch: 0.8 MB of the 4096-env
  // step's HBM traffic, the "1.65 x" of round 2's five-barrier schedule)
  float cs = r.cs, sn = r.sn, Dinv = r.Dinv, u = r.u;
  float w0 = r.Uw[0], w1 = r.Uw[1], w2 = r.Uw[2], v0 = r.Uv[0], v1 = r.Uv[1], v2 = r.Uv[2];
  SS_REG(cs); SS_REG(sn); SS_REG(Dinv); SS_REG(u);
  SS_REG(w0); SS_REG(w1); SS_REG(w2); SS_REG(v0); SS_REG(v1); SS_REG(v2);
  JRec o;
  o.cs = cs; o.sn = sn; o.Dinv = Dinv; o.u = u;
  o.Uw[0] = w0; o.Uw[1] = w1; o.Uw[2] = w2; o.Uv[0] = v0; o.Uv[1] = v1; o.Uv[2] = v2;
  return o;
}
#ifndef SS_PIN
#define SS_PIN(J) ((J) == 7)
#endif
// two columns at once in packed f32: the unloaded down step applies the same joint operator to every column of T, so
// a pair of columns shares each instruction (v_pk_*), coefficients broadcast
template <class Model, int J>
SSD SV2 imp_down_pair(const JointCache& jc, const SV2& p) {
  constexpr int ax = kAxis[J], k = half_pos(J), ai = (ax + 1) % 3, aj = (ax + 2) % 3;
  constexpr float rx = Model::r[J][0], ry = Model::r[J][1], rz = Model::r[J][2];
  const JRec& r = jc.r[k];
  float c = r.cs, s = r.sn;
  if constexpr (SS_PIN(J)) { SS_REG(c); SS_REG(s); }
  SV2 d;
  d.w[ax] = p.w[ax];
  d.w[ai] = p.w[ai] * c + p.w[aj] * s;
  d.w[aj] = p.w[aj] * c - p.w[ai] * s;
  ssf2 t[3] = {p.v[0], p.v[1], p.v[2]};
  if constexpr (has_offset<Model, J>()) {         // v_p + w_p x r = v_p - r x w_p
    if constexpr (ry != 0.f) t[0] -= p.w[2] * ry;
    if constexpr (rz != 0.f) t[0] += p.w[1] * rz;
    if constexpr (rz != 0.f) t[1] -= p.w[0] * rz;
    if constexpr (rx != 0.f) t[1] += p.w[2] * rx;
    if constexpr (rx != 0.f) t[2] -= p.w[1] * rx;
    if constexpr (ry != 0.f) t[2] += p.w[0] * ry;
  }
  d.v[ax] = t[ax];
  d.v[ai] = t[ai] * c + t[aj] * s;
  d.v[aj] = t[aj] * c - t[ai] * s;
  // the record's fields as opaque register values: a splat of a field that is still a load when instcombine runs becomes an
  // overlapping <2 x float> load, and the record then stays in scratch (the ankle record did: 32 B per helper lane stored and
  // re-loaded every substep and written back at the end of every launch)
  float uw0 = r.Uw[0], uw1 = r.Uw[1], uw2 = r.Uw[2], uv0 = r.Uv[0], uv1 = r.Uv[1], uv2 = r.Uv[2], di = r.Dinv;
  if constexpr (SS_PIN(J)) { SS_REG(uw0); SS_REG(uw1); SS_REG(uw2); SS_REG(uv0); SS_REG(uv1); SS_REG(uv2); SS_REG(di); }
  ssf2 dotv = d.w[0] * uw0 + d.w[1] * uw1 + d.w[2] * uw2 + d.v[0] * uv0 + d.v[1] * uv1 + d.v[2] * uv2;
  d.w[ax] -= dotv * di;
  return d;
}

template <class Model, int J>
SSD SV2 imp_up_pair(const JointCache& jc, ssf2* ul2, const SV2& p) {
  constexpr int ax = kAxis[J], k = half_pos(J), ai = (ax + 1) % 3, aj = (ax + 2) % 3;
  constexpr float rx = Model::r[J][0], ry = Model::r[J][1], rz = Model::r[J][2];
  const JRec& r = jc.r[k];
  float c = r.cs, s = r.sn, di = r.Dinv;
  float uw[3] = {r.Uw[0], r.Uw[1], r.Uw[2]}, uv[3] = {r.Uv[0], r.Uv[1], r.Uv[2]};
  if constexpr (SS_PIN(J)) {
    SS_REG(c); SS_REG(s); SS_REG(di);
#pragma unroll
    for (int i = 0; i < 3; ++i) { SS_REG(uw[i]); SS_REG(uv[i]); }
  }
  const ssf2 u = -p.w[ax];
  ul2[k] = u;
  const ssf2 du = u * di;
  ssf2 fw[3], fv[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { fw[i] = p.w[i] + du * uw[i]; fv[i] = p.v[i] + du * uv[i]; }
  SV2 o;                                             // f_p = R f_c,  n_p = R n_c + r x f_p
  o.v[ax] = fv[ax]; o.v[ai] = fv[ai] * c - fv[aj] * s; o.v[aj] = fv[ai] * s + fv[aj] * c;
  o.w[ax] = fw[ax]; o.w[ai] = fw[ai] * c - fw[aj] * s; o.w[aj] = fw[ai] * s + fw[aj] * c;
  if constexpr (has_offset<Model, J>()) {
    if constexpr (ry != 0.f) o.w[0] += o.v[2] * ry;
    if constexpr (rz != 0.f) o.w[0] -= o.v[1] * rz;
    if constexpr (rz != 0.f) o.w[1] += o.v[0] * rz;
    if constexpr (rx != 0.f) o.w[1] -= o.v[2] * rx;
    if constexpr (rx != 0.f) o.w[2] += o.v[1] * rx;
    if constexpr (ry != 0.f) o.w[2] -= o.v[0] * ry;
  }
  return o;
}
template <class Model, int J>
SSD SV2 imp_down_pair_loaded(const JointCache& jc, const ssf2* ul2, const SV2& p) {
  constexpr int ax = kAxis[J], k = half_pos(J);
  const JRec& r = jc.r[k];
  SV2 d = imp_down_pair<Model, J>(jc, p);            // includes  - Dinv * (U . d)
  float di = r.Dinv;
  if constexpr (SS_PIN(J)) SS_REG(di);
  d.w[ax] += ul2[k] * di;
  return d;
}
SSD SV2 chol6_solve_neg_pair(const Chol6& L, const SV2& b) {
  ssf2 y[6] = {-b.w[0], -b.w[1], -b.w[2], -b.v[0], -b.v[1], -b.v[2]};
  static_for<0, 6>([&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    ssf2 s = y[i];
    static_for<0, i>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      s -= y[k] * L.template get<i, k>();
    });
    y[i] = s * L.di[i];
  });
  static_rfor<5, 0>([&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    ssf2 s = y[i];
    static_for<i + 1, 6>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      s -= y[k] * L.template get<k, i>();
    });
    y[i] = s * L.di[i];
  });
  SV2 x;
  x.w[0] = y[0]; x.w[1] = y[1]; x.w[2] = y[2]; x.v[0] = y[3]; x.v[1] = y[4]; x.v[2] = y[5];
  return x;
}


// Contact-space operators for the column pair (2c, 2c+1): T = K = P_7 ... P_3 (own-foot twist per unit pelvis twist through
// the unloaded leg; LDS), and by unit impulses on the own foot through the whole tree G (pelvis twist; LDS) and
// Lambda_own (own-foot twist; returned as columns of three pairs).  Two columns share every instruction (packed f32).
// Opaque register copies of what the packed recursions read as scalars: the callee is optimised on its own before it
// is inlined, and instcombine then widens 'splat (load float)' of neighbouring record fields into overlapping
// <2 x float> loads, which pins the record in scratch after inlining.
struct LamPair { ssf2 a[3], b[3]; };
// Part A needs the leg records (joints 3..7) only: the T columns and the unit impulses carried from the foot up to the pelvis.
// Part B needs the spine records and the base factor as well: up the spine, base solve, down to the pelvis (G) and the foot
// (Lambda_own).  The helper wavefronts run A while the main wavefront is still in the spine and the base solve.
struct OpCarry { SV2 p; ssf2 ul2[NH]; };
// opaque register copies of the records a part reads (see above)
SSD void operator_records_leg(const JointCache& jc_in, JointCache& jc) {
#pragma unroll
  for (int k = 3; k < 8; ++k) jc.r[k] = opaque_rec(jc_in.r[k]);
}
SSD void operator_records_spine(const JointCache& jc_in, JointCache& jc) {
#pragma unroll
  for (int k = 0; k < 3; ++k) jc.r[k] = opaque_rec(jc_in.r[k]);
#pragma unroll
  for (int i = 0; i < 15; ++i) { jc.L0.l[i] = jc_in.L0.l[i]; SS_REG(jc.L0.l[i]); }
#pragma unroll
  for (int i = 0; i < 6; ++i) { jc.L0.di[i] = jc_in.L0.di[i]; SS_REG(jc.L0.di[i]); }
}
// part A, first half: T columns 2c, 2c+1 -> LDS (every T column must be there before any part B: C needs all six)
template <class Model, int CPAIR>
SSD void operator_T(const JointCache& jc, const Lds& L) {
  SV2 d;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    d.w[m] = ssf2{2 * CPAIR == m ? 1.f : 0.f, 2 * CPAIR + 1 == m ? 1.f : 0.f};
    d.v[m] = ssf2{2 * CPAIR == m + 3 ? 1.f : 0.f, 2 * CPAIR + 1 == m + 3 ? 1.f : 0.f};
  }
  static_for<3, 8>([&](auto Jc) { d = imp_down_pair<Model, decltype(Jc)::value>(jc, d); });
  L.q2(kLdsT + (2 * CPAIR) * 3 + 0) = make_float2(d.w[0].x, d.w[1].x);
  L.q2(kLdsT + (2 * CPAIR) * 3 + 1) = make_float2(d.w[2].x, d.v[0].x);
  L.q2(kLdsT + (2 * CPAIR) * 3 + 2) = make_float2(d.v[1].x, d.v[2].x);
  L.q2(kLdsT + (2 * CPAIR + 1) * 3 + 0) = make_float2(d.w[0].y, d.w[1].y);
  L.q2(kLdsT + (2 * CPAIR + 1) * 3 + 1) = make_float2(d.w[2].y, d.v[0].y);
  L.q2(kLdsT + (2 * CPAIR + 1) * 3 + 2) = make_float2(d.v[1].y, d.v[2].y);
}
// part A, second half: unit impulses on the own foot (columns 2c, 2c+1) carried up to the pelvis
template <class Model, int CPAIR>
SSD void operator_up(const JointCache& jc, OpCarry& oc) {
  SV2 p;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    p.w[m] = ssf2{2 * CPAIR == m ? -1.f : 0.f, 2 * CPAIR + 1 == m ? -1.f : 0.f};
    p.v[m] = ssf2{2 * CPAIR == m + 3 ? -1.f : 0.f, 2 * CPAIR + 1 == m + 3 ? -1.f : 0.f};
  }
  static_rfor<7, 3>([&](auto Jc) { p = imp_up_pair<Model, decltype(Jc)::value>(jc, oc.ul2, p); });
  oc.p = p;
}
template <class Model, int CPAIR>
SSD void operator_pair_a(const JointCache& jc_in, const Lds& L, JointCache& jc, OpCarry& oc) {
  operator_records_leg(jc_in, jc);
  operator_T<Model, CPAIR>(jc, L);
  operator_up<Model, CPAIR>(jc, oc);
}
// part B: up the spine, base solve, down to the pelvis: G columns (pelvis twist per unit impulse on the own foot).  What the
// PGS needs of G is its effect on the OTHER foot, so the lane pair swaps the columns (mirrored on receipt) and each lane
// stores C = T_own . mirror(G_partner): own-foot twist per unit impulse on the partner's foot -- one 6x6 operator per sweep
// in the PGS instead of two (18 instead of 36 packed FMAs and LDS reads per sweep).  Then down the own leg: Lambda_own.
template <class Model, int CPAIR>
SSD LamPair operator_pair_b(const Lds& L, const JointCache& jc, OpCarry& oc) {
  SV2 p = oc.p;
  static_rfor<2, 0>([&](auto Jc) { p = imp_up_pair<Model, decltype(Jc)::value>(jc, oc.ul2, p); });
  SV2 d = chol6_solve_neg_pair(jc.L0, p);
  static_for<0, 3>([&](auto Jc) { d = imp_down_pair_loaded<Model, decltype(Jc)::value>(jc, oc.ul2, d); });
  {
    // partner's columns 2c, 2c+1 in this lane's world (y-mirror: axial part (-,+,-), polar part (+,-,+))
    ssf2 go[6];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const float sw = (m == 1) ? 1.f : -1.f, sv = (m == 1) ? -1.f : 1.f;
      go[m] = ssf2{sw * xchg(d.w[m].x), sw * xchg(d.w[m].y)};
      go[3 + m] = ssf2{sv * xchg(d.v[m].x), sv * xchg(d.v[m].y)};
    }
    ssf2 ca[3], cb[3];
#pragma unroll
    for (int l = 0; l < 6; ++l) {
      const ssf2 ga = {go[l].x, go[l].x}, gb = {go[l].y, go[l].y};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float2 t = L.q2(kLdsT + l * 3 + i);
        const ssf2 t2 = {t.x, t.y};
        if (l == 0) { ca[i] = t2 * ga; cb[i] = t2 * gb; }
        else { ca[i] = t2 * ga + ca[i]; cb[i] = t2 * gb + cb[i]; }
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      L.q2(kLdsC + (2 * CPAIR) * 3 + i) = make_float2(ca[i].x, ca[i].y);
      L.q2(kLdsC + (2 * CPAIR + 1) * 3 + i) = make_float2(cb[i].x, cb[i].y);
    }
  }
  static_for<3, 8>([&](auto Jc) { d = imp_down_pair_loaded<Model, decltype(Jc)::value>(jc, oc.ul2, d); });
  LamPair o;
  o.a[0] = ssf2{d.w[0].x, d.w[1].x}; o.a[1] = ssf2{d.w[2].x, d.v[0].x}; o.a[2] = ssf2{d.v[1].x, d.v[2].x};
  o.b[0] = ssf2{d.w[0].y, d.w[1].y}; o.b[1] = ssf2{d.w[2].y, d.v[0].y}; o.b[2] = ssf2{d.v[1].y, d.v[2].y};
  return o;
}

// Forward kinematics of spine + own leg and contact detection of the own sole's four corners against the three active
// stones (PHYSICS.md 3.3).  Inputs: cos / sin of joints 0..7, base rotation, base position and stones from LDS.
struct DetectOut {
  float Rf[3][3];          // foot orientation (world)
  float pen[4];            // penetration depth per corner
  int active;              // bit k: corner k touches a stone
  int cslot;               // 2 bits per corner: which stone slot
};
template <class Model, bool BRANCHFREE = true>
SSD void fk_detect(const float* cs8, const float* sn8, const float (&Rb)[3][3], const Lds& L, DetectOut& o, FootReport& fr) {
  float Rf[3][3], pf[3];
  {
    float Rw[9][3][3], pw[9][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      pw[0][a] = L.s(S_POS + a);
#pragma unroll
      for (int c = 0; c < 3; ++c) Rw[0][a][c] = Rb[a][c];
    }
    static_for<0, 8>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
      constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
      constexpr float rx = Model::r[j][0], ry = Model::r[j][1], rz = Model::r[j][2];
      float c = cs8[j], sn = sn8[j];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float o_ = pw[p][r];
        SS_ACC(o_, rx, Rw[p][r][0]); SS_ACC(o_, ry, Rw[p][r][1]); SS_ACC(o_, rz, Rw[p][r][2]);
        pw[b][r] = o_;
        Rw[b][r][ai] = c * Rw[p][r][ai] + sn * Rw[p][r][aj];
        Rw[b][r][aj] = c * Rw[p][r][aj] - sn * Rw[p][r][ai];
        Rw[b][r][ax] = Rw[p][r][ax];
      }
    });
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      pf[a] = pw[RFOOT][a];
#pragma unroll
      for (int c = 0; c < 3; ++c) { Rf[a][c] = Rw[RFOOT][a][c]; o.Rf[a][c] = Rf[a][c]; }
    }
  }
  int active = 0, cslot = 0;
  fr.contact = 0;
  fr.on_target = 0;
  fr.sole[0] = fr.sole[1] = fr.sole[2] = 0.f;
  {
    float sp[3][3], sn_[3][3], hc[3], hs_[3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { sp[sl][i] = L.s(S_STP + sl * 3 + i); sn_[sl][i] = L.s(S_STN + sl * 3 + i); }
      const float2 hd = L.q2(kLdsHead + sl);
      hc[sl] = hd.x; hs_[sl] = hd.y;
    }
    static_for<0, 4>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      constexpr float cx = Model::corners[k][0], cy = Model::corners[k][1], cz = Model::corners[k][2];
      float P[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        P[r] = pf[r] + Rf[r][0] * cx + Rf[r][1] * cy + Rf[r][2] * cz;
        fr.sole[r] += 0.25f * P[r];
      }
      float best = 0.f;
      int slot = -1;
#pragma unroll
      for (int si = 0; si < 3; ++si) {
        const int sl = si == 0 ? 1 : (si == 1 ? 0 : 2);      // the target stone n first: it wins an exact tie (then n-1, then n+1)
        float dx = P[0] - sp[sl][0], dy = P[1] - sp[sl][1], dz = P[2] - sp[sl][2];
        float d = dx * sn_[sl][0] + dy * sn_[sl][1] + dz * sn_[sl][2];
        float lx = dx - d * sn_[sl][0], ly = dy - d * sn_[sl][1], lz = dz - d * sn_[sl][2];
        (void)lz;
        // the plank's footprint seen from above: the horizontal components of the in-plane offset along / across the stone's heading
        const float u = lx * hc[sl] + ly * hs_[sl], v = ly * hc[sl] - lx * hs_[sl];
        // the deeper stone wins, an exact tie goes to the stone visited first.  Same predicate, two codings (measured per variant,
        // profiles/r06_ab_disc_vs_plank_variants.txt): with && the compiler nests exec-mask branches around u and v -- the faster form
        // where a dedicated helper wavefront runs the detection beside the main one (three helpers), 3 % slower where the detection
        // sits on the longest path (plain kernel: the main wavefront; one helper: the helper that does everything)
        if constexpr (BRANCHFREE) {
          const bool touch = (d < 0.f) & (d > -0.10f) & (fabsf(u) < kPlankA) & (fabsf(v) < kPlankB) & (d < best);
          best = touch ? d : best;
          slot = touch ? sl : slot;
        } else {
          const bool touch = (d < 0.f) && (d > -0.10f) && (fabsf(u) < kPlankA) && (fabsf(v) < kPlankB);
          if (touch && d < best) { best = d; slot = sl; }
        }
      }
      o.pen[k] = -best;
      // on the target = a corner CARRIED by stone n (round 6; rounds 1-5: within stone n's disc, whichever stone carried it)
      if (slot == 1) fr.on_target = 1;
      if (slot >= 0) {
        active |= 1 << k;
        cslot |= slot << (2 * k);
        fr.contact = 1;
      }
    });
  }
  o.active = active;
  o.cslot = cslot;
}

// Jacobian rows of the own foot's four sole corners: per (corner k, direction d = normal, t1, t2) the row w = (c x dir, dir) in
// foot coordinates as three float pairs, and the Baumgarte term of the normal row (PHYSICS.md 3.4).  Inactive corners get finite
// rows (normal +z).
// moment part of a row: c x dir for sole corner K (w[3..5] = dir in foot coordinates)
template <class Model, int K>
SSD void row_moment(float (&w)[6]) {
  constexpr float cx = Model::corners[K][0], cy = Model::corners[K][1], cz = Model::corners[K][2];
  w[0] = cy * w[5] - cz * w[4];
  w[1] = cz * w[3] - cx * w[5];
  w[2] = cx * w[4] - cy * w[3];
}
template <class Model>
SSD void jacobian_rows(const DetectOut& det, const Lds& L, ssf2 (&rWp)[12][3], float (&rB)[4]) {
  const int active = det.active, cslot = det.cslot;
  const float (&Rf)[3][3] = det.Rf;
  const float (&pen)[4] = det.pen;
  static_for<0, 4>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value;
    const bool on = (active >> k) & 1;
    const int sl = (cslot >> (2 * k)) & 3;
    float n[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) n[i] = on ? L.s(S_STN + sl * 3 + i) : (i == 2 ? 1.f : 0.f);
    float t1[3] = {1.f - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
    float inv = SS_RSQRT(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
    t1[0] *= inv; t1[1] *= inv; t1[2] *= inv;
    float t2[3];
    cross(n, t1, t2);
    float corr = fmaxf(pen[k] - kSlop, 0.f);
    rB[k] = on ? fminf(kErp * corr * (1.0f / kH), kVcorrMax) : 0.f;
    static_for<0, 3>([&](auto Dc) {
      constexpr int d = decltype(Dc)::value, row = k * 3 + d;
      const float* dir = d == 0 ? n : (d == 1 ? t1 : t2);
      float w[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) w[3 + c] = Rf[0][c] * dir[0] + Rf[1][c] * dir[1] + Rf[2][c] * dir[2];
      row_moment<Model, k>(w);
#pragma unroll
      for (int i = 0; i < 3; ++i) rWp[row][i] = ssf2{w[2 * i], w[2 * i + 1]};
    });
  });
}

#ifndef SS_HOST_HARNESS
// Helper wavefronts of the small-batch variant (one workgroup = main wavefront + HELPERS helpers, five barriers per substep):
//   #0 state of the substep is in LDS         helpers 1, 2: cos / sin of the joint angles -> LDS (main: joint torques)
//   #0b                                        helper 0: forward kinematics, contact detection -> LDS
//   #1 leg joint records are handed over      helper h: operators of column pair h, part A (T, impulses up to the pelvis)
//   #2 spine records + base factor as well    part B (G, Lambda_own)
//   #3 operators are in LDS
// while the main wavefront runs cos / sin, pass 1 and the leg half of pass 2 | the spine and the base solve | pass 3, the foot
// twist and the Jacobian rows | the PGS and the rest.
// `extra(helper)` runs between #0b and #1, where helpers 1 and 2 are idle: the rollout kernel draws the next step's actions there
// (last helper) and emits the previous step's outputs (helper 1, three-helper variant).
template <class Model, int HELPERS, class Extra>
__device__ __forceinline__ void helper_substep(int helper, const Lds& L, Extra&& extra) {
  __syncthreads();                                   // #0
  SS_FUZZ(0x10u + helper);
  if constexpr (cs_offload(HELPERS)) {   // cos / sin of the 12 joint angles for everybody: helpers 1 and 2 six each, or (SS_CS_PER_HELPER
                                         // = 4, the rollout kernel's unit) all three helpers four each
    constexpr int kPer = HELPERS >= 3 ? SS_CS_PER_HELPER : NH;
    const int first = HELPERS >= 3 ? (SS_CS_PER_HELPER == 4 ? helper * 4 : (helper - 1) * 6) : 0;
    if (HELPERS < 3 || SS_CS_PER_HELPER == 4 || helper >= 1) {
      float qh[kPer], c_[kPer], s_[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) qh[k] = L.s(S_Q + first + k);
      SS_MEMBAR();
#pragma unroll
      for (int k = 0; k < kPer; ++k) ss_sincos(qh[k], s_[k], c_[k]);
#pragma unroll
      for (int k = 0; k < kPer; ++k) { L.hs(kHandCs + first + k) = c_[k]; L.hs(kHandCs + NH + first + k) = s_[k]; }
    }
    __syncthreads();                                 // #0b
    SS_FUZZ(0x20u + helper);
  }
  float rowdir[12][3], rowB[4];       // helper 0: directions of the contact rows, Baumgarte terms
  DetectOut det0;                     // helper 0: what the detection found (kept in registers across barrier #1)
  if (helper == 0) {
    float cs8[8], sn8[8];
    float quat[4] = {L.s(S_QUAT), L.s(S_QUAT + 1), L.s(S_QUAT + 2), L.s(S_QUAT + 3)};
    if constexpr (cs_offload(HELPERS)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { cs8[k] = L.hs(kHandCs + k); sn8[k] = L.hs(kHandCs + NH + k); }
      SS_MEMBAR();
    } else {
      float q8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) q8[k] = L.s(S_Q + k);
      SS_MEMBAR();
#pragma unroll
      for (int k = 0; k < 8; ++k) ss_sincos(q8[k], sn8[k], cs8[k]);
    }
    float Rb[3][3];
    quat_rot(quat, Rb);
    DetectOut det;
    FootReport fr;
    fk_detect<Model, (HELPERS < 3)>(cs8, sn8, Rb, L, det, fr);
    const int flags = det.active | (det.cslot << 4) | (fr.contact << 12) | (fr.on_target << 13);
    L.hs(kHandDet + 13) = __builtin_bit_cast(float, flags);
#pragma unroll
    for (int i = 0; i < 3; ++i) L.hs(kHandDet + 14 + i) = fr.sole[i];
    if constexpr (!rows_offload(HELPERS)) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) L.hs(kHandDet + a * 3 + c) = det.Rf[a][c];
#pragma unroll
      for (int k = 0; k < 4; ++k) L.hs(kHandDet + 9 + k) = det.pen[k];
    }
    det0 = det;
  }
  if constexpr (bias_offload(HELPERS)) {
    if (helper == 1) {                 // spine velocities from the base twist, then the bias forces of bodies 1..3 and 0
      const SV v0 = base_twist(L);
      float qd3[3], cs3[3], sn3[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { qd3[k] = L.s(S_QD + k); cs3[k] = L.hs(kHandCs + k); sn3[k] = L.hs(kHandCs + NH + k); }
      SS_MEMBAR();
      SV prev = v0;
      static_for<0, 3>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value, bd = j + 1;
        SV v = xmotion<Model, j>(cs3[j], sn3[j], prev);
        v.w[kAxis[j]] += qd3[j];
        prev = v;
        if constexpr (Model::mass[bd] != 0.f) {
          const SV pb = body_bias<Model, bd>(v);
#pragma unroll
          for (int m = 0; m < 3; ++m) { L.hs(kHandBias + bd * 6 + m) = pb.w[m]; L.hs(kHandBias + bd * 6 + 3 + m) = pb.v[m]; }
        }
      });
      const SV pb0 = body_bias<Model, 0>(v0);
#pragma unroll
      for (int m = 0; m < 3; ++m) { L.hs(kHandBias + m) = pb0.w[m]; L.hs(kHandBias + 3 + m) = pb0.v[m]; }
    }
  }
  extra(helper);
  __syncthreads();                                   // #1: leg joint records are in the hand-off region
  SS_FUZZ(0x30u + helper);
  if constexpr (rows_offload(HELPERS)) {
    // the rows' directions (round 6: moved here from the window before #1, where helper 0's kinematics + detection + rows were the
    // longest path of the workgroup; between #1 and #2 the helpers' part A is a third of the main wavefront's spine window).  They stay
    // in registers until the leg records' place is free (after #2)
    if (helper == 0) {
      ssf2 rWp[12][3];
      jacobian_rows<Model>(det0, L, rWp, rowB);
#pragma unroll
      for (int row = 0; row < 12; ++row) { rowdir[row][0] = rWp[row][1].y; rowdir[row][1] = rWp[row][2].x; rowdir[row][2] = rWp[row][2].y; }
#pragma unroll
      for (int row = 0; row < 12; ++row)
#pragma unroll
        for (int i = 0; i < 3; ++i) L.hs(kHandRows3 + row * 3 + i) = rowdir[row][i];
#pragma unroll
      for (int k = 0; k < 4; ++k) L.hs(kHandRows3 + 36 + k) = rowB[k];
    }
  }
  JointCache jin, jc;
  static_for<3, 8>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value;
    JRec& r = jin.r[k];
    r.cs = L.hs(kHandJc + k * 9 + 0); r.sn = L.hs(kHandJc + k * 9 + 1); r.Dinv = L.hs(kHandJc + k * 9 + 2);
#pragma unroll
    for (int m = 0; m < 3; ++m) { r.Uw[m] = L.hs(kHandJc + k * 9 + 3 + m); r.Uv[m] = L.hs(kHandJc + k * 9 + 6 + m); }
  });
  OpCarry oc[HELPERS == 1 ? 3 : 1];
  static_for<0, 3>([&](auto Cc) {
    constexpr int c = decltype(Cc)::value;
    if (HELPERS == 1 || helper == c) operator_pair_a<Model, c>(jin, L, jc, oc[HELPERS == 1 ? c : 0]);
  });
  __syncthreads();                                   // #2: spine records and the base factor
  SS_FUZZ(0x40u + helper);
  static_for<0, 3>([&](auto Kc) {
    constexpr int k = decltype(Kc)::value;
    JRec& r = jin.r[k];
    r.cs = L.hs(kHandJc + k * 9 + 0); r.sn = L.hs(kHandJc + k * 9 + 1); r.Dinv = L.hs(kHandJc + k * 9 + 2);
#pragma unroll
    for (int m = 0; m < 3; ++m) { r.Uw[m] = L.hs(kHandJc + k * 9 + 3 + m); r.Uv[m] = L.hs(kHandJc + k * 9 + 6 + m); }
  });
#pragma unroll
  for (int i = 0; i < 15; ++i) jin.L0.l[i] = L.hs(kHandL0 + i);
#pragma unroll
  for (int i = 0; i < 6; ++i) jin.L0.di[i] = L.hs(kHandL0 + 15 + i);
  operator_records_spine(jin, jc);
  static_for<0, 3>([&](auto Cc) {
    constexpr int c = decltype(Cc)::value;
    if (HELPERS == 1 || helper == c) {
      const LamPair lp = operator_pair_b<Model, c>(L, jc, oc[HELPERS == 1 ? c : 0]);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        L.hs(kHandLc + (2 * c) * 6 + 2 * i) = lp.a[i].x; L.hs(kHandLc + (2 * c) * 6 + 2 * i + 1) = lp.a[i].y;
        L.hs(kHandLc + (2 * c + 1) * 6 + 2 * i) = lp.b[i].x; L.hs(kHandLc + (2 * c + 1) * 6 + 2 * i + 1) = lp.b[i].y;
      }
    }
  });
  __syncthreads();                                   // #3: C, T and Lambda_own are ready
  SS_FUZZ(0x50u + helper);
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// State (q, qd, base pose/twist), stones and clipped actions of THIS lane's world live in LDS (region B).
template <class Model, int HELPERS = 0>
SSD void substep(SS_PROF_DECL float power, FootReport& fr, const Lds& L, Warm& wm) {
  constexpr float h = kH;
  JointCache jc;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (HELPERS > 0) { __syncthreads(); SS_FUZZ(0x1Fu); }   // #0: the state of this substep is in LDS (helper 0: kinematics + detection)
#endif
  SS_PROF(0);
  // LDS round trips (~100 cycles) are fully exposed with one wavefront per SIMD, and the compiler issues each
  // ds_read right before its use: batch the loads of a phase up front / prefetch one joint ahead instead.
  float qdfr[NH];              // free joint velocities
#define SS_QDF(k) qdfr[k]
  float qd_all[NH], q_all[NH], act_all[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) { q_all[k] = L.s(S_Q + k); qd_all[k] = L.s(S_QD + k); act_all[k] = L.s(S_ACT + k); }
  SS_MEMBAR();
  // explicit joint torque and implicit diagonal of joint j (PHYSICS.md 3.1)
  auto joint_tau = [&](auto Jc, float q, float qd, float act, float& tau, float& Dadd) {
    constexpr int j = decltype(Jc)::value;
    constexpr float lo = Model::lo[j], hi = Model::hi[j], kd = Model::damping[j], ks = Model::stiffness[j];
    constexpr float klim = Model::klim[j], dlim = Model::dlim[j], arm = Model::armature[j], tq = Model::torque[j];
    float viol = q > hi ? q - hi : (q < lo ? q - lo : 0.f);
    bool lim = viol != 0.f;
    float kl = lim ? klim : 0.f, dl = lim ? dlim : 0.f;
    tau = power * tq * act - kd * qd - ks * (q + h * qd) - kl * (viol + h * qd) - dl * qd;
    Dadd = arm + h * (kd + dl) + (h * h) * (ks + kl);
  };
  float tau_pre[NH], dadd_pre[NH];     // helper variant: the joint torques here, while the helpers evaluate cos / sin
  if constexpr (cs_offload(HELPERS) && HELPERS > 0) {
    static_for<0, NH>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      joint_tau(std::integral_constant<int, kHalf[k]>{}, q_all[k], qd_all[k], act_all[k], tau_pre[k], dadd_pre[k]);
    });
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    SS_PROF(1);
    __syncthreads();                   // #0b: cos / sin are in the hand-off region
    SS_PROF(12);                       // (tuning builds: the wait at barrier #0b)
    SS_FUZZ(0x2Fu);
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int k = 0; k < NH; ++k) { jc.r[k].cs = L.hs(kHandCs + k); jc.r[k].sn = L.hs(kHandCs + NH + k); }
    SS_MEMBAR();
  } else {
    static_for<0, NH>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      ss_sincos(q_all[k], jc.r[k].sn, jc.r[k].cs);
    });
  }
  auto tau_of = [&](auto Jc, auto Kc, float& tau, float& Dadd) {
    constexpr int k = decltype(Kc)::value;
    if constexpr (cs_offload(HELPERS) && HELPERS > 0) { tau = tau_pre[k]; Dadd = dadd_pre[k]; }
    else joint_tau(Jc, q_all[k], qd_all[k], act_all[k], tau, Dadd);
  };
  SS_PROF(1);

  // ================= leg joints 3..6 and arm joints 13..16 as float pairs {leg, arm} (ss_pair.hpp) =================
  // Same axes, same massless/massive pattern: one v_pk instruction serves both chains.  Spine joints 0..2 and the
  // ankle (joint 7) stay scalar.  Topology relied on (asserted): arm on the torso, leg on the pelvis, pelvis on the spine.
  static_assert(kParent[13] == 0 && kParent[3] == 3 && kParent[2] == 2 && kParent[1] == 1 && kParent[0] == 0 &&
                kParent[7] == 7 && kParent[6] == 6 && kParent[16] == 16, "half-tree topology");
  SV a0;
  {
    const SV v0 = base_twist(L);
    ssf2 c2[4], s2[4], qd2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c2[i] = pkv(jc.r[3 + i].cs, jc.r[8 + i].cs); s2[i] = pkv(jc.r[3 + i].sn, jc.r[8 + i].sn);
      qd2[i] = pkv(qd_all[3 + i], qd_all[8 + i]);
    }
    // ---- pass 1: velocities
    SV vs[3], vfoot;          // bodies 1, 2, 3 and 8
    SV2 vp[4];                // bodies (4,14) (5,15) (6,16) (7,17)
    {
      SV prev = v0;
      static_for<0, 3>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        SV v = xmotion<Model, j>(jc.r[j].cs, jc.r[j].sn, prev);
        v.w[kAxis[j]] += qd_all[j];
        vs[j] = v;
        prev = v;
      });
      SV2 pp = sv_pack(prev, v0);
      static_for<0, 4>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        SV2 v = xmotionP<Model, 3 + i, 13 + i>(c2[i], s2[i], pp);
        v.w[kAxis[3 + i]] += qd2[i];
        vp[i] = v;
        pp = v;
      });
      vfoot = xmotion<Model, 7>(jc.r[7].cs, jc.r[7].sn, sv_half(pp, 0));
      vfoot.w[kAxis[7]] += qd_all[7];
    }
    SS_PROF(2);
    // ---- pass 2: articulated inertias, leaves -> root
    // one scalar joint: consumes the articulated inertia / bias of its child body, leaves the joint record, returns the
    // contribution to the parent (parent coordinates)
    auto joint_scalar = [&](auto Jc, ABI I, const SV& pA, const SV& vb, ABI& Ip, SV& pp) {
      constexpr int j = decltype(Jc)::value, k = half_pos(j), ax = kAxis[j];
      constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
      float tau, Dadd;
      tau_of(Jc, std::integral_constant<int, k>{}, tau, Dadd);
      const float qd = qd_all[k];
      JRec& r = jc.r[k];
      r.Uw[0] = I.A.template get<0, ax>(); r.Uw[1] = I.A.template get<1, ax>(); r.Uw[2] = I.A.template get<2, ax>();
      r.Uv[0] = I.B[ax][0]; r.Uv[1] = I.B[ax][1]; r.Uv[2] = I.B[ax][2];
      r.Dinv = SS_RCP(r.Uw[ax] + Dadd);
      r.u = tau - pA.w[ax];
      const float* Uw = r.Uw;
      const float* Uv = r.Uv;
      float sw[3] = {r.Dinv * Uw[0], r.Dinv * Uw[1], r.Dinv * Uw[2]};
      float sv[3] = {r.Dinv * Uv[0], r.Dinv * Uv[1], r.Dinv * Uv[2]};
      I.A.m[0] -= sw[0] * Uw[0]; I.A.m[1] -= sw[1] * Uw[1]; I.A.m[2] -= sw[2] * Uw[2];
      I.A.m[3] -= sw[0] * Uw[1]; I.A.m[4] -= sw[0] * Uw[2]; I.A.m[5] -= sw[1] * Uw[2];
      I.C.m[0] -= sv[0] * Uv[0]; I.C.m[1] -= sv[1] * Uv[1]; I.C.m[2] -= sv[2] * Uv[2];
      I.C.m[3] -= sv[0] * Uv[1]; I.C.m[4] -= sv[0] * Uv[2]; I.C.m[5] -= sv[1] * Uv[2];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) I.B[a][c] -= sw[a] * Uv[c];
      float cwi = qd * vb.w[aj], cwj = -qd * vb.w[ai];
      float cvi = qd * vb.v[aj], cvj = -qd * vb.v[ai];
      float du = r.Dinv * r.u;
      SV pa;
      {
        const Sym3 &A = I.A, &C = I.C;
        float Af[3][3] = {{A.m[0], A.m[3], A.m[4]}, {A.m[3], A.m[1], A.m[5]}, {A.m[4], A.m[5], A.m[2]}};
        float Cf[3][3] = {{C.m[0], C.m[3], C.m[4]}, {C.m[3], C.m[1], C.m[5]}, {C.m[4], C.m[5], C.m[2]}};
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          pa.w[rr] = pA.w[rr] + Af[rr][ai] * cwi + Af[rr][aj] * cwj + I.B[rr][ai] * cvi + I.B[rr][aj] * cvj + Uw[rr] * du;
          pa.v[rr] = pA.v[rr] + I.B[ai][rr] * cwi + I.B[aj][rr] * cwj + Cf[rr][ai] * cvi + Cf[rr][aj] * cvj + Uv[rr] * du;
        }
      }
      Ip = xinertia<Model, j>(r.cs, r.sn, I);
      pp = xforce<Model, j>(r.cs, r.sn, pa);
    };
    struct JRec2 { ssf2 Uw[3], Uv[3], Dinv, u; };
    JRec2 jr2[4];
    auto joint_pair = [&](auto Ic, ABIP I, const SV2& pA, ABIP& Ip, SV2& pp) {
      constexpr int i = decltype(Ic)::value, jl = 3 + i, ja = 13 + i, kl = 3 + i, ka = 8 + i, ax = kAxis[jl];
      constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
      float taul, Daddl, taua, Dadda;
      tau_of(std::integral_constant<int, jl>{}, std::integral_constant<int, kl>{}, taul, Daddl);
      tau_of(std::integral_constant<int, ja>{}, std::integral_constant<int, ka>{}, taua, Dadda);
      const ssf2 qd = qd2[i];
      const SV2& vb = vp[i];
      JRec2& r = jr2[i];
      r.Uw[0] = I.A.template get<0, ax>(); r.Uw[1] = I.A.template get<1, ax>(); r.Uw[2] = I.A.template get<2, ax>();
      r.Uv[0] = I.B[ax][0]; r.Uv[1] = I.B[ax][1]; r.Uv[2] = I.B[ax][2];
      const ssf2 D = r.Uw[ax] + pkv(Daddl, Dadda);
      r.Dinv = pkv(SS_RCP(D.x), SS_RCP(D.y));
      r.u = pkv(taul, taua) - pA.w[ax];
      const ssf2* Uw = r.Uw;
      const ssf2* Uv = r.Uv;
      ssf2 sw[3] = {r.Dinv * Uw[0], r.Dinv * Uw[1], r.Dinv * Uw[2]};
      ssf2 sv[3] = {r.Dinv * Uv[0], r.Dinv * Uv[1], r.Dinv * Uv[2]};
      I.A.m[0] -= sw[0] * Uw[0]; I.A.m[1] -= sw[1] * Uw[1]; I.A.m[2] -= sw[2] * Uw[2];
      I.A.m[3] -= sw[0] * Uw[1]; I.A.m[4] -= sw[0] * Uw[2]; I.A.m[5] -= sw[1] * Uw[2];
      I.C.m[0] -= sv[0] * Uv[0]; I.C.m[1] -= sv[1] * Uv[1]; I.C.m[2] -= sv[2] * Uv[2];
      I.C.m[3] -= sv[0] * Uv[1]; I.C.m[4] -= sv[0] * Uv[2]; I.C.m[5] -= sv[1] * Uv[2];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) I.B[a][c] -= sw[a] * Uv[c];
      ssf2 cwi = qd * vb.w[aj], cwj = -qd * vb.w[ai];
      ssf2 cvi = qd * vb.v[aj], cvj = -qd * vb.v[ai];
      ssf2 du = r.Dinv * r.u;
      SV2 pa;
      {
        const Sym3P &A = I.A, &C = I.C;
        ssf2 Af[3][3] = {{A.m[0], A.m[3], A.m[4]}, {A.m[3], A.m[1], A.m[5]}, {A.m[4], A.m[5], A.m[2]}};
        ssf2 Cf[3][3] = {{C.m[0], C.m[3], C.m[4]}, {C.m[3], C.m[1], C.m[5]}, {C.m[4], C.m[5], C.m[2]}};
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          pa.w[rr] = pA.w[rr] + Af[rr][ai] * cwi + Af[rr][aj] * cwj + I.B[rr][ai] * cvi + I.B[rr][aj] * cvj + Uw[rr] * du;
          pa.v[rr] = pA.v[rr] + I.B[ai][rr] * cwi + I.B[aj][rr] * cwj + Cf[rr][ai] * cvi + Cf[rr][aj] * cvj + Uv[rr] * du;
        }
      }
      Ip = xinertiaP<Model, jl, ja>(c2[i], s2[i], I);
      pp = xforceP<Model, jl, ja>(c2[i], s2[i], pa);
      // scalar joint records for the contact stage (sub-register views of the pairs)
      JRec& rl = jc.r[kl];
      JRec& ra = jc.r[ka];
#pragma unroll
      for (int m = 0; m < 3; ++m) { rl.Uw[m] = r.Uw[m].x; ra.Uw[m] = r.Uw[m].y; rl.Uv[m] = r.Uv[m].x; ra.Uv[m] = r.Uv[m].y; }
      rl.Dinv = r.Dinv.x; ra.Dinv = r.Dinv.y; rl.u = r.u.x; ra.u = r.u.y;
    };
    ABI acc0;                  // what reaches the torso: arm (pair of this and the partner lane) + spine
    SV pacc0;
    {
      ABI If;                  // ankle: foot body 8 is a leaf
      SV pf_;
      joint_scalar(std::integral_constant<int, 7>{}, abi_body<Model, 8>(), body_bias<Model, 8>(vfoot), vfoot, If, pf_);
      // knee / elbow (bodies 7, 17): the leg half carries the foot
      ABIP I2 = abi_zeroP();
#pragma unroll
      for (int m = 0; m < 6; ++m) { I2.A.m[m] = pkv(If.A.m[m], 0.f); I2.C.m[m] = pkv(If.C.m[m], 0.f); }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) I2.B[a][c] = pkv(If.B[a][c], 0.f);
      SV2 p2;
#pragma unroll
      for (int m = 0; m < 3; ++m) { p2.w[m] = pkv(pf_.w[m], 0.f); p2.v[m] = pkv(pf_.v[m], 0.f); }
      static_rfor<3, 0>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value, bl = 4 + i, ba = 14 + i;     // child bodies of joints 3+i / 13+i
        if constexpr (massiveP<Model, bl, ba>()) {
          abi_add_bodyP<Model, bl, ba>(I2);
          SV2 pb = body_biasP<Model, bl, ba>(vp[i]);
#pragma unroll
          for (int m = 0; m < 3; ++m) { p2.w[m] += pb.w[m]; p2.v[m] += pb.v[m]; }
        }
        ABIP Ipn;
        SV2 ppn;
        joint_pair(Ic, I2, p2, Ipn, ppn);
        I2 = Ipn;
        p2 = ppn;
      });
#if defined(__HIP_DEVICE_COMPILE__)
      if constexpr (HELPERS > 0) {     // leg joint records to the helper wavefront(s): operators, part A
        static_for<3, 8>([&](auto Kc) {
          constexpr int k = decltype(Kc)::value;
          const JRec& r = jc.r[k];
          L.hs(kHandJc + k * 9 + 0) = r.cs; L.hs(kHandJc + k * 9 + 1) = r.sn; L.hs(kHandJc + k * 9 + 2) = r.Dinv;
#pragma unroll
          for (int m = 0; m < 3; ++m) { L.hs(kHandJc + k * 9 + 3 + m) = r.Uw[m]; L.hs(kHandJc + k * 9 + 6 + m) = r.Uv[m]; }
        });
        // the machine scheduler would otherwise pull the spine and the base solve in front of the barrier (it orders memory
        // operations only), and the helpers' part A would overlap nothing
        __builtin_amdgcn_sched_barrier(0);
        SS_PROF(3);
        __syncthreads();               // #1
        SS_PROF(14);                   // (tuning builds: the main wavefront's wait at barrier #1)
        SS_FUZZ(0x3Fu);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      // I2 / p2: leg half in pelvis coordinates, arm half in torso coordinates.  Add the partner lane's limbs
      // (mirrored), commutative (mine + partner) so that both lanes get bit-identical totals.
      ABI Il = abi_half(I2, 0), Ia = abi_half(I2, 1);
      SV pl = sv_half(p2, 0), pa_ = sv_half(p2, 1);
      {
        ABI Io = xchg_abi(Il);
        SV po = xchg_sv(pl);
        abi_add(Il, Io);
#pragma unroll
        for (int m = 0; m < 3; ++m) { pl.w[m] += po.w[m]; pl.v[m] += po.v[m]; }
        Io = xchg_abi(Ia);
        po = xchg_sv(pa_);
        abi_add(Ia, Io);
#pragma unroll
        for (int m = 0; m < 3; ++m) { pa_.w[m] += po.w[m]; pa_.v[m] += po.v[m]; }
      }
      // spine: bodies 3 (pelvis, carries both legs), 2, 1
      ABI Is = Il;
      SV ps = pl;
      static_rfor<2, 0>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value, bd = j + 1;
        if constexpr (Model::mass[bd] != 0.f) {
          abi_add_body<Model, bd>(Is);
          SV pb;
          if constexpr (bias_offload(HELPERS)) {
#pragma unroll
            for (int m = 0; m < 3; ++m) { pb.w[m] = L.hs(kHandBias + bd * 6 + m); pb.v[m] = L.hs(kHandBias + bd * 6 + 3 + m); }
          } else {
            pb = body_bias<Model, bd>(vs[j]);
          }
#pragma unroll
          for (int m = 0; m < 3; ++m) { ps.w[m] += pb.w[m]; ps.v[m] += pb.v[m]; }
        }
        ABI Ipn;
        SV ppn;
        joint_scalar(Jc, Is, ps, vs[j], Ipn, ppn);
        Is = Ipn;
        ps = ppn;
      });
      acc0 = Is;
      abi_add(acc0, Ia);
#pragma unroll
      for (int m = 0; m < 3; ++m) { pacc0.w[m] = ps.w[m] + pa_.w[m]; pacc0.v[m] = ps.v[m] + pa_.v[m]; }
    }
    SS_PROF(3);
    // ---- base (redundant in both lanes)
    {
      ABI I0 = acc0;
      abi_add_body<Model, 0>(I0);
      SV pb;
      if constexpr (bias_offload(HELPERS)) {
#pragma unroll
        for (int m = 0; m < 3; ++m) { pb.w[m] = L.hs(kHandBias + m); pb.v[m] = L.hs(kHandBias + 3 + m); }
      } else {
        pb = body_bias<Model, 0>(v0);
      }
      SV p0;
#pragma unroll
      for (int i = 0; i < 3; ++i) { p0.w[i] = pacc0.w[i] + pb.w[i]; p0.v[i] = pacc0.v[i] + pb.v[i]; }
      float M[6][6];
      abi_dense(I0, M);
      jc.L0 = chol6(M);
      a0 = chol6_solve_neg(jc.L0, p0);
    }
    SS_PROF(4);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (HELPERS > 0) {       // spine joint records and the base factor: operators, part B
      static_for<0, 3>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value;
        const JRec& r = jc.r[k];
        L.hs(kHandJc + k * 9 + 0) = r.cs; L.hs(kHandJc + k * 9 + 1) = r.sn; L.hs(kHandJc + k * 9 + 2) = r.Dinv;
#pragma unroll
        for (int m = 0; m < 3; ++m) { L.hs(kHandJc + k * 9 + 3 + m) = r.Uw[m]; L.hs(kHandJc + k * 9 + 6 + m) = r.Uv[m]; }
      });
#pragma unroll
      for (int m = 0; m < 15; ++m) L.hs(kHandL0 + m) = jc.L0.l[m];
#pragma unroll
      for (int m = 0; m < 6; ++m) L.hs(kHandL0 + 15 + m) = jc.L0.di[m];
      __builtin_amdgcn_sched_barrier(0);
      SS_PROF(4);
      __syncthreads();                 // #2
      SS_PROF(15);                     // (tuning builds: the wait at barrier #2)
      SS_FUZZ(0x4Fu);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    // ---- pass 3: accelerations -> free velocities
    {
      auto acc_scalar = [&](auto Jc, const SV& aprev, const SV& vb) {
        constexpr int j = decltype(Jc)::value, k = half_pos(j), ax = kAxis[j];
        constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
        const JRec& r = jc.r[k];
        SV a = xmotion<Model, j>(r.cs, r.sn, aprev);
        float qd = qd_all[k];
        a.w[ai] += qd * vb.w[aj]; a.w[aj] -= qd * vb.w[ai];
        a.v[ai] += qd * vb.v[aj]; a.v[aj] -= qd * vb.v[ai];
        float dotv = r.Uw[0] * a.w[0] + r.Uw[1] * a.w[1] + r.Uw[2] * a.w[2] + r.Uv[0] * a.v[0] + r.Uv[1] * a.v[1] +
                     r.Uv[2] * a.v[2];
        float qdd = r.Dinv * (r.u - dotv);
        a.w[ax] += qdd;
        SS_QDF(k) = qd + h * qdd;
        return a;
      };
      SV prev = a0;
      static_for<0, 3>([&](auto Jc) { prev = acc_scalar(Jc, prev, vs[decltype(Jc)::value]); });
      SV2 pp = sv_pack(prev, a0);
      static_for<0, 4>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value, jl = 3 + i, ja = 13 + i, ax = kAxis[jl];
        constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
        const JRec2& r = jr2[i];
        const SV2& vb = vp[i];
        SV2 a = xmotionP<Model, jl, ja>(c2[i], s2[i], pp);
        const ssf2 qd = qd2[i];
        a.w[ai] += qd * vb.w[aj]; a.w[aj] -= qd * vb.w[ai];
        a.v[ai] += qd * vb.v[aj]; a.v[aj] -= qd * vb.v[ai];
        ssf2 dotv = r.Uw[0] * a.w[0] + r.Uw[1] * a.w[1] + r.Uw[2] * a.w[2] + r.Uv[0] * a.v[0] + r.Uv[1] * a.v[1] +
                    r.Uv[2] * a.v[2];
        ssf2 qdd = r.Dinv * (r.u - dotv);
        a.w[ax] += qdd;
        const ssf2 qf = qd + qdd * h;
        SS_QDF(3 + i) = qf.x;
        SS_QDF(8 + i) = qf.y;
        pp = a;
      });
      acc_scalar(std::integral_constant<int, 7>{}, sv_half(pp, 0), vfoot);
    }
  }
  float quat[4] = {L.s(S_QUAT), L.s(S_QUAT + 1), L.s(S_QUAT + 2), L.s(S_QUAT + 3)};
  float Rb[3][3];
  quat_rot(quat, Rb);
  SV v0f;
  {
    const SV v0 = base_twist(L);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v0f.w[i] = v0.w[i] + h * a0.w[i];
      v0f.v[i] = v0.v[i] + h * (a0.v[i] - kGrav * Rb[2][i]);   // R^T g = -9.8 * (third row of R)
    }
  }
  SS_MEMBAR();
  SS_PROF(5);

  // ---- detect: FK of spine + own leg, own sole corners vs stones.  (Round 2 measured this on helper 0, overlapped with
  // pass 2: the main wavefront then waits for the operators at barrier #2 instead -- 0.0685 vs 0.0654 ms/step; rejected.)
  DetectOut det;
  if constexpr (HELPERS > 0) {         // helper 0 did it between barriers #0 and #1
    if constexpr (!rows_offload(HELPERS)) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) det.Rf[a][c] = L.hs(kHandDet + a * 3 + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) det.pen[k] = L.hs(kHandDet + 9 + k);
    }
    const int flags = __builtin_bit_cast(int, L.hs(kHandDet + 13));
    det.active = flags & 15;
    det.cslot = (flags >> 4) & 255;
    fr.contact = (flags >> 12) & 1;
    fr.on_target = (flags >> 13) & 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) fr.sole[i] = L.hs(kHandDet + 14 + i);
  } else {
    float cs8[8], sn8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { cs8[k] = jc.r[k].cs; sn8[k] = jc.r[k].sn; }
    fk_detect<Model>(cs8, sn8, Rb, L, det, fr);
  }
  SS_PROF(6);

  // ---- contact solve
  float dqd[NH];
  SV dv0;
#pragma unroll
  for (int k = 0; k < NH; ++k) dqd[k] = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { dv0.w[i] = 0.f; dv0.v[i] = 0.f; }
  const int active = det.active;
  const int pair_active = active | xchg_i(active);   // both lanes must take the contact branch together
#ifdef SS_ABLATE_CONTACT
  const bool in_contact = false;
#else
  const bool in_contact = pair_active != 0;
#endif
  ssf2 Lc[6][3];                       // column b of Lambda_own as three pairs
  SS_PROF(7);
  // What does not need the contact operators: with helper wavefronts it overlaps part B of their work.
  float V[6];                          // own-foot twist under the free velocities
  ssf2 rWp[12][3];                     // Jacobian row w = (c x dir, dir) per (corner, direction), as three float pairs
  float rB[4];
  auto rows_free = [&]() {          // V and the Jacobian rows: nothing here reads the operators
      {
        SV a = v0f;
        static_for<0, 8>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          a = xmotion<Model, j>(jc.r[j].cs, jc.r[j].sn, a);
          a.w[kAxis[j]] += SS_QDF(j);
        });
#pragma unroll
        for (int i = 0; i < 3; ++i) { V[i] = a.w[i]; V[3 + i] = a.v[i]; }
      }
      if constexpr (!rows_offload(HELPERS)) jacobian_rows<Model>(det, L, rWp, rB);
      if constexpr (rows_offload(HELPERS)) {
        // the rows' directions and Baumgarte terms from helper 0 (written before barrier #2); the moment part here -- all of it while
        // the helpers are still forming the operators (the main wavefront waited ~ 600 clocks per substep at barrier #3)
        static_for<0, 12>([&](auto Rc) {
          constexpr int row = decltype(Rc)::value;
          float w[6];
#pragma unroll
          for (int i = 0; i < 3; ++i) w[3 + i] = L.hs(kHandRows3 + row * 3 + i);
          row_moment<Model, row / 3>(w);
#pragma unroll
          for (int i = 0; i < 3; ++i) rWp[row][i] = ssf2{w[2 * i], w[2 * i + 1]};
        });
#pragma unroll
        for (int k = 0; k < 4; ++k) rB[k] = L.hs(kHandRows3 + 36 + k);
      }
  };
  auto solve = [&]() {              // y = Lambda w, PGS, response of the whole tree
      float ul[NH];
      float wlam_prev[4][3];           // the previous substep's impulses
      int wkey_prev = 0;
      auto load_warm = [&]() {
        if constexpr (kPgsWarm) {
          if constexpr (warm_in_lds(HELPERS)) {
            wkey_prev = __builtin_bit_cast(int, L.s(S_WKEY));
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
              for (int d = 0; d < 3; ++d) wlam_prev[k][d] = L.s(S_WLAM + 3 * k + d);
          } else {
            wkey_prev = wm.key;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
              for (int d = 0; d < 3; ++d) wlam_prev[k][d] = wm.lam[k][d];
          }
        }
      };
      // with helper wavefronts the loads are issued here and their LDS latency hides behind the rows; the plain variant has no
      // register to hold them that long (20 B per lane of scratch otherwise) and loads them where they are used
      if constexpr (HELPERS > 0) load_warm();
      if constexpr (HELPERS > 0) {
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
          for (int i = 0; i < 3; ++i) Lc[b][i] = pkv(L.hs(kHandLc + b * 6 + 2 * i), L.hs(kHandLc + b * 6 + 2 * i + 1));
      }
      // rows (registers, float pairs): per (corner, direction) y = Lambda_own w, 1/A.  Inactive corners keep finite rows
      // (normal +z) and get 1/A = 0, b = 0, which freezes their lambda at 0.
      ssf2 rYp[12][3];
      float rIA[12];
      static_for<0, 12>([&](auto Rc) {
        constexpr int row = decltype(Rc)::value, k = row / 3;
        const bool on = (active >> k) & 1;
        const float w[6] = {rWp[row][0].x, rWp[row][0].y, rWp[row][1].x, rWp[row][1].y, rWp[row][2].x, rWp[row][2].y};
        ssf2 y[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) y[i] = Lc[0][i] * ssf2{w[0], w[0]};
#pragma unroll
        for (int b = 1; b < 6; ++b)
#pragma unroll
          for (int i = 0; i < 3; ++i) y[i] = Lc[b][i] * ssf2{w[b], w[b]} + y[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) rYp[row][i] = y[i];
        ssf2 acc = rWp[row][0] * y[0];
        acc = rWp[row][1] * y[1] + acc;
        acc = rWp[row][2] * y[2] + acc;
        rIA[row] = on ? SS_RCP(acc.x + acc.y) : 0.f;
      });
      SS_PROF(8);
      // projected Gauss-Seidel in packed f32 (v_pk_fma_f32: two lanes of the 6-vectors per instruction): the foot twist,
      // the rows y = Lambda w and w, the sweep's wrench and the G / T columns are held as three float pairs each
      // warm start (PHYSICS.md 3.4): a corner that was in contact in the previous substep of this control step starts from that
      // substep's impulses; every other corner from zero
      if constexpr (HELPERS == 0) load_warm();
      float lam[4][3];
      const int key = active;
      int warm_any = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool w = kPgsWarm && (((active & wkey_prev) >> k) & 1);
        warm_any |= w ? 1 : 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) lam[k][d] = w ? wlam_prev[k][d] : 0.f;
      }
      constexpr float mu = Model::friction;
      ssf2 Vp[3] = {{V[0], V[1]}, {V[2], V[3]}, {V[4], V[5]}};
      ssf2 Wp[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      ssf2 dWp[3];
      auto sweep = [&]() {           // Gauss-Seidel over the own foot's 12 rows
#pragma unroll
        for (int i = 0; i < 3; ++i) dWp[i] = ssf2{0.f, 0.f};
        static_for<0, 12>([&](auto Rc) {
          constexpr int row = decltype(Rc)::value, k = row / 3, d = row % 3;
          ssf2 acc = rWp[row][0] * Vp[0];
          acc = rWp[row][1] * Vp[1] + acc;
          acc = rWp[row][2] * Vp[2] + acc;
          float vrel = acc.x + acc.y;
          float ln = lam[k][d] + ((d == 0 ? rB[k] : 0.f) - vrel) * rIA[row];
          if constexpr (d == 0) {
            ln = fmaxf(ln, 0.f);
          } else {
            float lim = mu * lam[k][0];
            ln = fminf(fmaxf(ln, -lim), lim);
          }
          const float dl = ln - lam[k][d];
          lam[k][d] = ln;
          const ssf2 dl2 = {dl, dl};
#pragma unroll
          for (int i = 0; i < 3; ++i) { Vp[i] = rYp[row][i] * dl2 + Vp[i]; dWp[i] = rWp[row][i] * dl2 + dWp[i]; }
        });
#pragma unroll
        for (int i = 0; i < 3; ++i) Wp[i] += dWp[i];
      };
#ifdef SS_PGS_NO_PEEL
      constexpr int kCoupled = kPgsIters;
#else
      constexpr int kCoupled = kPgsIters - 1;   // after the last sweep nothing reads the foot twist any more: its coupling is dead
#endif
      ssf2 Cc[6][3];                   // read once (the compiler parks what does not fit in AGPRs: 0.0514 -> 0.0511 ms/step)
#pragma unroll
      for (int l = 0; l < 6; ++l)
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float2 c = L.q2(kLdsC + l * 3 + i); Cc[l][i] = ssf2{c.x, c.y}; }
#if defined(__HIP_DEVICE_COMPILE__)
      // all LDS reads land before the loop: otherwise its body carries eleven `s_waitcnt lgkmcnt(n)` for the first iteration's sake
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), leave vmcnt / expcnt alone
#endif
      if constexpr (kPgsWarm) {
        // the starting impulses move both feet before the first sweep: the own foot through y = Lambda_own w, the partner's foot
        // through C (like a sweep's increments).  A lane pair without a warm corner skips it (the products would add zeros).
        if ((warm_any | xchg_i(warm_any)) != 0) {
          static_for<0, 12>([&](auto Rc) {
            constexpr int row = decltype(Rc)::value, k = row / 3, d = row % 3;
            const ssf2 l2 = {lam[k][d], lam[k][d]};
#pragma unroll
            for (int i = 0; i < 3; ++i) { Vp[i] = rYp[row][i] * l2 + Vp[i]; Wp[i] = rWp[row][i] * l2 + Wp[i]; }
          });
          ssf2 Wo[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) Wo[i] = ssf2{xchg(Wp[i].x), xchg(Wp[i].y)};
#pragma unroll
          for (int l = 0; l < 6; ++l) {
            const float sc = (l & 1) ? Wo[l >> 1].y : Wo[l >> 1].x;
            const ssf2 s2 = {sc, sc};
#pragma unroll
            for (int i = 0; i < 3; ++i) Vp[i] = Cc[l][i] * s2 + Vp[i];
          }
        }
      }
#pragma unroll 1                       // (unrolled by 2 or fully: 0.0522 vs 0.0504 ms/step; C formed by the helpers after barrier #3
                                       // behind a sixth barrier instead of inside part B: no gain either)
      for (int it = 0; it < kCoupled; ++it) {
        sweep();
        // the partner's sweep impulses (its own world, as they are) move this foot through C = T_own . mirror(G_partner)
        ssf2 dWo[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) dWo[i] = ssf2{xchg(dWp[i].x), xchg(dWp[i].y)};
#pragma unroll
        for (int l = 0; l < 6; ++l) {
          const float sc = (l & 1) ? dWo[l >> 1].y : dWo[l >> 1].x;
          const ssf2 s2 = {sc, sc};
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            Vp[i] = Cc[l][i] * s2 + Vp[i];
          }
        }
      }
#ifndef SS_PGS_NO_PEEL
      sweep();
#endif
      SV W = {{Wp[0].x, Wp[0].y, Wp[1].x}, {Wp[1].y, Wp[2].x, Wp[2].y}};
      if constexpr (kPgsWarm) {        // what the next substep of this control step starts from
        if constexpr (warm_in_lds(HELPERS)) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int d = 0; d < 3; ++d) L.s(S_WLAM + 3 * k + d) = lam[k][d];
          L.s(S_WKEY) = __builtin_bit_cast(float, key);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int d = 0; d < 3; ++d) wm.lam[k][d] = lam[k][d];
          wm.key = key;
        }
      }
      SS_PROF(9);
      // accumulated foot wrenches -> whole tree: own leg up, pelvis biases summed over the pair, spine, base, down
#ifndef SS_ABLATE_FINAL
      {
        SV p = {{-W.w[0], -W.w[1], -W.w[2]}, {-W.v[0], -W.v[1], -W.v[2]}};
        static_rfor<7, 3>([&](auto Jc) { p = imp_up<Model, decltype(Jc)::value>(jc, ul, p); });
        const SV po = xchg_sv(p);
#pragma unroll
        for (int i = 0; i < 3; ++i) { p.w[i] += po.w[i]; p.v[i] += po.v[i]; }
        static_rfor<2, 0>([&](auto Jc) { p = imp_up<Model, decltype(Jc)::value>(jc, ul, p); });
        dv0 = chol6_solve_neg(jc.L0, p);
        SV d = dv0;
        static_for<0, 8>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          d = imp_down<Model, j, true>(jc, ul, d, &dqd[half_pos(j)]);
        });
        d = dv0;
        static_for<13, 17>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          d = imp_down<Model, j, false>(jc, ul, d, &dqd[half_pos(j)]);
        });
      }
#endif
  };
  if (kPgsWarm && !in_contact) {                 // no contact in this substep: nothing to start the next one from
    if constexpr (warm_in_lds(HELPERS)) L.s(S_WKEY) = 0.f;
    else wm.key = 0;
  }
  if constexpr (HELPERS == 0) {     // one block, the operators first (before the rows occupy the registers)
    if (in_contact) {
      JointCache jo;                   // opaque copies of the spine + leg records and the base factor
      operator_records_leg(jc, jo);
      operator_records_spine(jc, jo);
      static_for<0, 3>([&](auto Cc) { operator_T<Model, decltype(Cc)::value>(jo, L); });    // all six T columns first: C needs them
      static_for<0, 3>([&](auto Cc) {
        constexpr int c = decltype(Cc)::value;
        OpCarry oc;
        operator_up<Model, c>(jo, oc);
        const LamPair lp = operator_pair_b<Model, c>(L, jo, oc);
#pragma unroll
        for (int i = 0; i < 3; ++i) { Lc[2 * c][i] = lp.a[i]; Lc[2 * c + 1][i] = lp.b[i]; }
      });
      rows_free();
      solve();
    }
  } else {
    if (in_contact) rows_free();
#if defined(__HIP_DEVICE_COMPILE__)
    SS_PROF(8);
    __syncthreads();                 // #3: the helper wavefront(s) have written C, T and Lambda_own
    SS_PROF(7);                      // (tuning builds, helper variants: the wait at barrier #3)
    SS_FUZZ(0x5Fu);
#endif
    if (in_contact) solve();
  }
  SS_MEMBAR();
  SS_PROF(10);

  // ---- integrate (semi-implicit Euler), state back to LDS
  {
    float qf[NH], qq[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { qf[k] = SS_QDF(k); qq[k] = L.s(S_Q + k); }   // (q kept in registers to here: slower)
    SS_MEMBAR();
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      float qd = qf[k] + dqd[k];
      L.s(S_QD + k) = qd;
      L.s(S_Q + k) = qq[k] + h * qd;
    }
  }
  SV v0n;
#pragma unroll
  for (int i = 0; i < 3; ++i) { v0n.w[i] = v0f.w[i] + dv0.w[i]; v0n.v[i] = v0f.v[i] + dv0.v[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { L.s(S_VW + i) = v0n.w[i]; L.s(S_VV + i) = v0n.v[i]; }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    L.s(S_POS + r) += h * (Rb[r][0] * v0n.v[0] + Rb[r][1] * v0n.v[1] + Rb[r][2] * v0n.v[2]);
  {
    float qw = quat[0], qx = quat[1], qy = quat[2], qz = quat[3];
    float ox = v0n.w[0], oy = v0n.w[1], oz = v0n.w[2], hh = 0.5f * h;
    float nw = qw + hh * (-qx * ox - qy * oy - qz * oz);
    float nx = qx + hh * (qw * ox + qy * oz - qz * oy);
    float ny = qy + hh * (qw * oy - qx * oz + qz * ox);
    float nz = qz + hh * (qw * oz + qx * oy - qy * ox);
    float inv = SS_RSQRT(nw * nw + nx * nx + ny * ny + nz * nz);
    L.s(S_QUAT) = nw * inv; L.s(S_QUAT + 1) = nx * inv; L.s(S_QUAT + 2) = ny * inv; L.s(S_QUAT + 3) = nz * inv;
  }
  SS_MEMBAR();
  SS_PROF(11);
}

}  // namespace ss
