// ss_dynamics.hpp -- one physics substep (docs/PHYSICS.md section 3) for one environment per lane.
//
// Structure (all loops over the kinematic tree are unrolled at compile time):
//   pass 1  link velocities                     root -> leaves
//   pass 2  articulated inertias / bias forces  leaves -> root    (U, 1/D, u kept per joint)
//   base    6x6 Cholesky solve
//   pass 3  accelerations                        root -> leaves   -> free velocities
//   detect  sole corners vs the three active stones
//   solve   Lambda^-1 blocks by 12 unit impulse responses, contact rows precomputed, 8 projected Gauss-Seidel
//           sweeps in the 12-dim foot-twist space, one whole-tree impulse response to apply the foot wrenches
//   integrate (semi-implicit Euler)
//
// Register / LDS budget.  One wavefront per workgroup, one workgroup per CU: each lane owns 512 VGPR+AGPR and a
// private 2560-byte share of the CU's 160 KiB LDS; no barrier is ever needed (a workgroup is one wavefront and
// lanes never read each other's data).  The environment's dynamic state is staged through LDS for the whole
// control step; the register file holds only the sweep in flight plus the 13 leg/spine joint records.
//   region A (108 float4 slots, lane stride 16 B -> conflict-free ds_*_b128)
//       contact phase: slots 0..35 Lambda^-1 columns, slots 36..107 24 contact rows x 3 float4
//       ABA phase:     the same bytes hold the 21 link twists as scalars (dead before the contact phase)
//   region B (52 slots = 208 scalars, element-major [idx][lane] -> conflict-free ds_*_b32)
//       arm-joint records 80 | actions 21 | q 21 | qd 21 | free qd 21 | base pose+twist 13 | stones 18
#pragma once
#include "ss_math.hpp"

namespace ss {

constexpr float kH = 1.0f / 240.0f;
constexpr float kDt = 1.0f / 60.0f;
constexpr float kGrav = 9.8f;
constexpr float kStoneR2 = 0.25f * 0.25f;
constexpr int kPgsIters = 8;
constexpr float kErp = 0.2f;
constexpr float kSlop = 0.001f;
constexpr float kVcorrMax = 2.0f;

constexpr int kWave = 64;
constexpr int kLdsSlots = 160;         // 160 float4 = 2560 B per lane = 163,840 B per wavefront (all of the CU's LDS)
constexpr int kSlotsA = 108;
constexpr int kLdsLinv = 0;            // region A
constexpr int kLdsRows = 36;
constexpr int kScalarBase = kSlotsA * kWave * 4;   // region B, in floats
enum { S_ARMS = 0, S_ACT = 80, S_Q = 101, S_QD = 122, S_QDF = 143, S_POS = 164, S_QUAT = 167, S_VW = 171, S_VV = 174,
       S_STP = 177, S_STN = 186, S_END = 195 };
static_assert(S_END <= (kLdsSlots - kSlotsA) * 4, "LDS scalar region overflow");
constexpr int kNumLegJoints = 13;      // joints 0..12 (spine + legs) keep their records in registers

#if defined(__HIP_DEVICE_COMPILE__) || !defined(__HIP__)
#define SS_MEMBAR() asm volatile("" ::: "memory")
#else
#define SS_MEMBAR() asm volatile("" ::: "memory")
#endif

// optional per-phase cycle accounting (-DSS_PROFILE_PHASES; tuning builds only)
#if defined(SS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
struct Prof { uint32_t t[16]; uint32_t last; };
#define SS_PROF_DECL Prof& prof,
#define SS_PROF_ARG prof,
#define SS_PROF(i) do { uint32_t _n = (uint32_t)__builtin_amdgcn_s_memtime(); prof.t[i] += _n - prof.last; prof.last = _n; } while (0)
#else
struct Prof { int unused; };
#define SS_PROF_DECL
#define SS_PROF_ARG
#define SS_PROF(i) ((void)0)
#endif

struct Lds {       // lane-private view of the workgroup's LDS
  float* base;
  int lane;
  SSD float4& q4(int slot) const { return reinterpret_cast<float4*>(base)[slot * kWave + lane]; }
  SSD float& s(int idx) const { return base[kScalarBase + idx * kWave + lane]; }   // region B scalar
  SSD float& av(int idx) const { return base[idx * kWave + lane]; }                // ABA-phase scalar over region A
};

struct Dyn {       // dynamic state of one env (registers, only at the edges of the control step)
  float pos[3];
  float quat[4];
  SV v0;           // base twist, body coordinates
  float q[NJ];
  float qd[NJ];
};

struct Stones {    // the three active stones n-1, n, n+1: centre, unit normal, tilts (x, y)
  float p[3][3], nrm[3][3], tilt[3][2];
};

struct FootReport {
  int contact;     // bit f: foot f has a contact
  int on_target;   // bit f: foot f touches stone n (slot 1)
  float sole[2][3];
};

struct JRec {      // what the ABA leaves behind per joint
  float cs, sn, Uw[3], Uv[3], Dinv, u;
};
struct JointCache {
  JRec r[kNumLegJoints];
  Chol6 L0;
};

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

template <int J>
SSD JRec jrec_get(const JointCache& jc, const Lds& L) {
  if constexpr (J < kNumLegJoints) {
    return jc.r[J];
  } else {
    constexpr int o = S_ARMS + (J - kNumLegJoints) * 10;
    JRec r;
    r.cs = L.s(o + 0); r.sn = L.s(o + 1);
    r.Uw[0] = L.s(o + 2); r.Uw[1] = L.s(o + 3); r.Uw[2] = L.s(o + 4);
    r.Uv[0] = L.s(o + 5); r.Uv[1] = L.s(o + 6); r.Uv[2] = L.s(o + 7);
    r.Dinv = L.s(o + 8); r.u = L.s(o + 9);
    return r;
  }
}
template <int J>
SSD void jrec_put(JointCache& jc, const Lds& L, const JRec& r) {
  if constexpr (J < kNumLegJoints) {
    jc.r[J] = r;
  } else {
    constexpr int o = S_ARMS + (J - kNumLegJoints) * 10;
    L.s(o + 2) = r.Uw[0]; L.s(o + 3) = r.Uw[1]; L.s(o + 4) = r.Uw[2];
    L.s(o + 5) = r.Uv[0]; L.s(o + 6) = r.Uv[1]; L.s(o + 7) = r.Uv[2];
    L.s(o + 8) = r.Dinv; L.s(o + 9) = r.u;
  }
}
template <int J>
SSD void jcs_get(const JointCache& jc, const Lds& L, float& c, float& s) {
  if constexpr (J < kNumLegJoints) { c = jc.r[J].cs; s = jc.r[J].sn; }
  else { c = L.s(S_ARMS + (J - kNumLegJoints) * 10 + 0); s = L.s(S_ARMS + (J - kNumLegJoints) * 10 + 1); }
}

template <int B>
SSD SV vel_get(const Lds& L) {
  SV v = {{L.av(6 * B + 0), L.av(6 * B + 1), L.av(6 * B + 2)}, {L.av(6 * B + 3), L.av(6 * B + 4), L.av(6 * B + 5)}};
  return v;
}
template <int B>
SSD void vel_put(const Lds& L, const SV& v) {
#pragma unroll
  for (int i = 0; i < 3; ++i) { L.av(6 * B + i) = v.w[i]; L.av(6 * B + 3 + i) = v.v[i]; }
}
SSD SV base_twist(const Lds& L) {
  SV v = {{L.s(S_VW), L.s(S_VW + 1), L.s(S_VW + 2)}, {L.s(S_VV), L.s(S_VV + 1), L.s(S_VV + 2)}};
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// ABA impulse response restricted to what the contact stage needs.
//   fR / fL : spatial impulses on the right / left foot (foot frame); LOAD_* says which are non-zero
//   outputs : foot twists VR (if WANT_R), VL; if FULL also dv0 and dqd[21] (whole tree, arms included)
template <class Model, bool LOAD_R, bool LOAD_L, bool WANT_R, bool FULL>
SSD void impulse_response(const JointCache& jc, const Lds& L, const SV& fR, const SV& fL, SV& VR, SV& VL, SV* dv0,
                          float* dqd) {
  float ul[kNumLegJoints];   // only loaded leg + spine entries are used
  auto up = [&](auto Jc, const SV& p) {
    constexpr int j = decltype(Jc)::value, ax = kAxis[j];
    const JRec& r = jc.r[j];
    float u = -p.w[ax];
    ul[j] = u;
    float du = r.Dinv * u;
    SV pa;
#pragma unroll
    for (int i = 0; i < 3; ++i) { pa.w[i] = p.w[i] + r.Uw[i] * du; pa.v[i] = p.v[i] + r.Uv[i] * du; }
    SV o = xforce<Model, j>(r.cs, r.sn, pa);
    SS_FENCE();
    return o;
  };
  auto leg_up = [&](auto J0c, const SV& f) {
    constexpr int j0 = decltype(J0c)::value;
    SV p = {{-f.w[0], -f.w[1], -f.w[2]}, {-f.v[0], -f.v[1], -f.v[2]}};
    static_rfor<j0 + 4, j0>([&](auto Jc) { p = up(Jc, p); });
    return p;
  };
  SV pPel;
  if constexpr (LOAD_R) pPel = leg_up(std::integral_constant<int, 3>{}, fR);
  if constexpr (LOAD_L) {
    SV t = leg_up(std::integral_constant<int, 8>{}, fL);
    if constexpr (LOAD_R) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { pPel.w[i] += t.w[i]; pPel.v[i] += t.v[i]; }
    } else {
      pPel = t;
    }
  }
  SV p = pPel;
  static_rfor<2, 0>([&](auto Jc) { p = up(Jc, p); });
  SV d0 = chol6_solve_neg(jc.L0, p);
  if constexpr (FULL) *dv0 = d0;
  auto down = [&](auto Jc, const SV& dpar) {
    constexpr int j = decltype(Jc)::value, ax = kAxis[j];
    constexpr bool loaded = (j <= 2) || (LOAD_R && j >= 3 && j <= 7) || (LOAD_L && j >= 8 && j <= 12);
    const JRec r = jrec_get<j>(jc, L);
    SV d = xmotion<Model, j>(r.cs, r.sn, dpar);
    float dotv = r.Uw[0] * d.w[0] + r.Uw[1] * d.w[1] + r.Uw[2] * d.w[2] + r.Uv[0] * d.v[0] + r.Uv[1] * d.v[1] +
                 r.Uv[2] * d.v[2];
    float dq;
    if constexpr (loaded) dq = r.Dinv * (ul[j] - dotv);
    else dq = -r.Dinv * dotv;
    d.w[ax] += dq;
    if constexpr (FULL) dqd[j] = dq;
    SS_FENCE();
    return d;
  };
  SV d3 = down(std::integral_constant<int, 2>{}, down(std::integral_constant<int, 1>{}, down(std::integral_constant<int, 0>{}, d0)));
  if constexpr (WANT_R || FULL) {
    SV a = d3;
    static_for<3, 8>([&](auto Jc) { a = down(Jc, a); });
    VR = a;
  }
  {
    SV a = d3;
    static_for<8, 13>([&](auto Jc) { a = down(Jc, a); });
    VL = a;
  }
  if constexpr (FULL) {
    SV a = d0;
    static_for<13, 17>([&](auto Jc) { a = down(Jc, a); });
    a = d0;
    static_for<17, 21>([&](auto Jc) { a = down(Jc, a); });
  }
}

// ---------------------------------------------------------------------------------------------------------------
// State (q, qd, base pose/twist), stones and clipped actions live in LDS (region B); power: torque scale.
template <class Model>
SSD void substep(SS_PROF_DECL float power, FootReport& fr, const Lds& L) {
  constexpr float h = kH;
  JointCache jc;
  SS_PROF(0);
  static_for<0, NJ>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value;
    float sn, cs;
    ss_sincos(L.s(S_Q + j), sn, cs);
    if constexpr (j < kNumLegJoints) { jc.r[j].cs = cs; jc.r[j].sn = sn; }
    else { L.s(S_ARMS + (j - kNumLegJoints) * 10 + 0) = cs; L.s(S_ARMS + (j - kNumLegJoints) * 10 + 1) = sn; }
  });
  SS_MEMBAR();
  SS_PROF(1);

  // ---- pass 1: velocities (kept in LDS; the chain predecessor stays in registers)
  {
    const SV v0 = base_twist(L);
    SV prev = v0;
    static_for<0, NJ>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
      float c, sn;
      jcs_get<j>(jc, L, c, sn);
      SV vp;
      if constexpr (p == j) vp = prev;               // parent is the body processed just before
      else if constexpr (p == 0) vp = v0;
      else vp = vel_get<p>(L);
      SV v = xmotion<Model, j>(c, sn, vp);
      v.w[ax] += L.s(S_QD + j);
      vel_put<b>(L, v);
      prev = v;
      SS_FENCE();
    });
  }
  SS_MEMBAR();
  SS_PROF(2);

  // ---- pass 2: articulated inertias
  ABI acc[NB];
  SV pacc[NB];
  static_rfor<NJ - 1, 0>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
    constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
    constexpr bool leaf = first_child(b) < 0;
    constexpr bool massive = Model::mass[b] != 0.f;
    const SV vb = vel_get<b>(L);
    ABI I;
    SV pA;
    if constexpr (leaf) {
      I = abi_body<Model, b>();
      pA = body_bias<Model, b>(vb);
    } else {
      I = acc[b];
      pA = pacc[b];
      if constexpr (massive) {
        abi_add_body<Model, b>(I);
        SV pb = body_bias<Model, b>(vb);
#pragma unroll
        for (int i = 0; i < 3; ++i) { pA.w[i] += pb.w[i]; pA.v[i] += pb.v[i]; }
      }
    }
    // joint torque (explicit part) and implicit diagonal, PHYSICS.md 3.1
    constexpr float lo = Model::lo[j], hi = Model::hi[j], kd = Model::damping[j], ks = Model::stiffness[j];
    constexpr float klim = Model::klim[j], dlim = Model::dlim[j], arm = Model::armature[j];
    constexpr float tq = Model::torque[j];
    float q = L.s(S_Q + j), qd = L.s(S_QD + j);
    float viol = q > hi ? q - hi : (q < lo ? q - lo : 0.f);
    bool lim = viol != 0.f;
    float kl = lim ? klim : 0.f, dl = lim ? dlim : 0.f;
    float tau_m = power * tq * L.s(S_ACT + j);
    float tau = tau_m - kd * qd - ks * (q + h * qd) - kl * (viol + h * qd) - dl * qd;
    float Dadd = arm + h * (kd + dl) + (h * h) * (ks + kl);
    // U = I S
    JRec r;
    jcs_get<j>(jc, L, r.cs, r.sn);
    r.Uw[0] = I.A.template get<0, ax>(); r.Uw[1] = I.A.template get<1, ax>(); r.Uw[2] = I.A.template get<2, ax>();
    r.Uv[0] = I.B[ax][0]; r.Uv[1] = I.B[ax][1]; r.Uv[2] = I.B[ax][2];
    r.Dinv = 1.0f / (r.Uw[ax] + Dadd);
    r.u = tau - pA.w[ax];
    jrec_put<j>(jc, L, r);
    const float* Uw = r.Uw;
    const float* Uv = r.Uv;
    // Ia = I - U Dinv U^T
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
    // c = v x S qd : only components ai, aj are non-zero
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
    ABI Ip = xinertia<Model, j>(r.cs, r.sn, I);
    SV pp = xforce<Model, j>(r.cs, r.sn, pa);
    if constexpr (b == first_child(p)) {
      acc[p] = Ip;
      pacc[p] = pp;
    } else {
      abi_add(acc[p], Ip);
#pragma unroll
      for (int i = 0; i < 3; ++i) { pacc[p].w[i] += pp.w[i]; pacc[p].v[i] += pp.v[i]; }
    }
    SS_FENCE();
  });

  SS_PROF(3);
  // ---- base
  SV a0;
  {
    const SV v0 = base_twist(L);
    ABI I0 = acc[0];
    abi_add_body<Model, 0>(I0);
    SV pb = body_bias<Model, 0>(v0);
    SV p0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { p0.w[i] = pacc[0].w[i] + pb.w[i]; p0.v[i] = pacc[0].v[i] + pb.v[i]; }
    float M[6][6];
    abi_dense(I0, M);
    jc.L0 = chol6(M);
    a0 = chol6_solve_neg(jc.L0, p0);
  }
  SS_MEMBAR();
  SS_PROF(4);

  // ---- pass 3: accelerations -> free velocities (to LDS)
  {
    SV prev = a0;
    SV a3 = a0;   // acceleration of the pelvis (body 3), branch point of the legs
    static_for<0, NJ>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
      constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
      const JRec r = jrec_get<j>(jc, L);
      const SV vb = vel_get<b>(L);
      SV ap;
      if constexpr (p == j) ap = prev;
      else if constexpr (p == 0) ap = a0;
      else ap = a3;                                   // p == 3 is the only other branch point
      static_assert(p == j || p == 0 || p == 3, "tree shape");
      SV a = xmotion<Model, j>(r.cs, r.sn, ap);
      float qd = L.s(S_QD + j);
      a.w[ai] += qd * vb.w[aj]; a.w[aj] -= qd * vb.w[ai];
      a.v[ai] += qd * vb.v[aj]; a.v[aj] -= qd * vb.v[ai];
      float dotv = r.Uw[0] * a.w[0] + r.Uw[1] * a.w[1] + r.Uw[2] * a.w[2] + r.Uv[0] * a.v[0] + r.Uv[1] * a.v[1] +
                   r.Uv[2] * a.v[2];
      float qdd = r.Dinv * (r.u - dotv);
      a.w[ax] += qdd;
      if constexpr (b == 3) a3 = a;
      prev = a;
      L.s(S_QDF + j) = qd + h * qdd;
      SS_FENCE();
    });
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

  // ---- detect: FK of spine + legs, sole corners vs stones
  float Rf[2][3][3], pf[2][3];
  {
    float Rw[14][3][3], pw[14][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      pw[0][a] = L.s(S_POS + a);
#pragma unroll
      for (int c = 0; c < 3; ++c) Rw[0][a][c] = Rb[a][c];
    }
    static_for<0, 13>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
      constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
      constexpr float rx = Model::r[j][0], ry = Model::r[j][1], rz = Model::r[j][2];
      float c = jc.r[j].cs, sn = jc.r[j].sn;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float o = pw[p][r];
        SS_ACC(o, rx, Rw[p][r][0]); SS_ACC(o, ry, Rw[p][r][1]); SS_ACC(o, rz, Rw[p][r][2]);
        pw[b][r] = o;
        Rw[b][r][ai] = c * Rw[p][r][ai] + sn * Rw[p][r][aj];
        Rw[b][r][aj] = c * Rw[p][r][aj] - sn * Rw[p][r][ai];
        Rw[b][r][ax] = Rw[p][r][ax];
      }
    });
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      pf[0][a] = pw[RFOOT][a]; pf[1][a] = pw[LFOOT][a];
#pragma unroll
      for (int c = 0; c < 3; ++c) { Rf[0][a][c] = Rw[RFOOT][a][c]; Rf[1][a][c] = Rw[LFOOT][a][c]; }
    }
  }
  int active = 0;          // bit k
  int cslot = 0;           // 2 bits per contact: stone slot
  float pen[8];
  fr.contact = 0;
  fr.on_target = 0;
  {
    float sp[3][3], sn_[3][3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
      for (int i = 0; i < 3; ++i) { sp[sl][i] = L.s(S_STP + sl * 3 + i); sn_[sl][i] = L.s(S_STN + sl * 3 + i); }
    static_for<0, 2>([&](auto Fc) {
      constexpr int f = decltype(Fc)::value;
      fr.sole[f][0] = fr.sole[f][1] = fr.sole[f][2] = 0.f;
      static_for<0, 4>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value, ck = f * 4 + k;
        constexpr float cx = Model::corners[k][0], cy = Model::corners[k][1], cz = Model::corners[k][2];
        float P[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          P[r] = pf[f][r] + Rf[f][r][0] * cx + Rf[f][r][1] * cy + Rf[f][r][2] * cz;
          fr.sole[f][r] += 0.25f * P[r];
        }
        float best = 0.f;
        int slot = -1;
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
          float dx = P[0] - sp[sl][0], dy = P[1] - sp[sl][1], dz = P[2] - sp[sl][2];
          float d = dx * sn_[sl][0] + dy * sn_[sl][1] + dz * sn_[sl][2];
          float lx = dx - d * sn_[sl][0], ly = dy - d * sn_[sl][1], lz = dz - d * sn_[sl][2];
          float rho2 = lx * lx + ly * ly + lz * lz;
          bool hit = (d < 0.f) && (d > -0.10f) && (rho2 < kStoneR2) && (d < best);
          if (hit) { best = d; slot = sl; }
        }
        pen[ck] = -best;
        if (slot >= 0) {
          active |= 1 << ck;
          cslot |= slot << (2 * ck);
          fr.contact |= 1 << f;
          if (slot == 1) fr.on_target |= 1 << f;
        }
      });
    });
  }

  SS_PROF(6);
  // ---- contact solve
  float dqd[NJ];
  SV dv0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) dqd[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { dv0.w[i] = 0.f; dv0.v[i] = 0.f; }
  if (active != 0) {
    // Lambda^-1 blocks -> LDS.  Column i of an R impulse holds [RR(:,i) ; LR(:,i)], of an L impulse [ - ; LL(:,i)].
    SV zero = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
// measured round 1 (4096 envs, ms/step): chains 1/2/4 with branchy PGS 0.283/0.295/-, branch-free PGS 0.318/0.325/0.315
#ifndef SS_LINV_CHAINS
#define SS_LINV_CHAINS 1
#endif
    auto put_r = [&](int col, const SV& VR, const SV& VL) {
      L.q4(kLdsLinv + col * 3 + 0) = make_float4(VR.w[0], VR.w[1], VR.w[2], VR.v[0]);
      L.q4(kLdsLinv + col * 3 + 1) = make_float4(VR.v[1], VR.v[2], VL.w[0], VL.w[1]);
      L.q4(kLdsLinv + col * 3 + 2) = make_float4(VL.w[2], VL.v[0], VL.v[1], VL.v[2]);
    };
    auto put_l = [&](int col, const SV& VL) {
      L.q4(kLdsLinv + (6 + col) * 3 + 1) = make_float4(0.f, 0.f, VL.w[0], VL.w[1]);
      L.q4(kLdsLinv + (6 + col) * 3 + 2) = make_float4(VL.w[2], VL.v[0], VL.v[1], VL.v[2]);
    };
#if SS_LINV_CHAINS == 4
    // four independent chains per iteration (R/L impulse x angular/linear component) so that the dependent
    // up/solve/down recursions of one chain hide the latency of the others
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
      SV ea, eb;
#pragma unroll
      for (int m = 0; m < 3; ++m) { ea.w[m] = (i == m) ? 1.f : 0.f; ea.v[m] = 0.f; eb.w[m] = 0.f; eb.v[m] = (i == m) ? 1.f : 0.f; }
      SV VRa, VLa, VRb, VLb, VRc, VLc, VRd, VLd;
      impulse_response<Model, true, false, true, false>(jc, L, ea, zero, VRa, VLa, nullptr, nullptr);
      impulse_response<Model, true, false, true, false>(jc, L, eb, zero, VRb, VLb, nullptr, nullptr);
      impulse_response<Model, false, true, false, false>(jc, L, zero, ea, VRc, VLc, nullptr, nullptr);
      impulse_response<Model, false, true, false, false>(jc, L, zero, eb, VRd, VLd, nullptr, nullptr);
      put_r(i, VRa, VLa); put_r(3 + i, VRb, VLb); put_l(i, VLc); put_l(3 + i, VLd);
    }
#elif SS_LINV_CHAINS == 2
    // two independent chains per iteration (right-foot and left-foot unit impulse)
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
      SV e;
#pragma unroll
      for (int m = 0; m < 3; ++m) { e.w[m] = (i == m) ? 1.f : 0.f; e.v[m] = (i == m + 3) ? 1.f : 0.f; }
      SV VRa, VLa, VRc, VLc;
      impulse_response<Model, true, false, true, false>(jc, L, e, zero, VRa, VLa, nullptr, nullptr);
      impulse_response<Model, false, true, false, false>(jc, L, zero, e, VRc, VLc, nullptr, nullptr);
      put_r(i, VRa, VLa); put_l(i, VLc);
    }
#else
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
      SV e;
#pragma unroll
      for (int m = 0; m < 3; ++m) { e.w[m] = (i == m) ? 1.f : 0.f; e.v[m] = (i == m + 3) ? 1.f : 0.f; }
      SV VR, VL;
      impulse_response<Model, true, false, true, false>(jc, L, e, zero, VR, VL, nullptr, nullptr);
      put_r(i, VR, VL);
    }
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
      SV e;
#pragma unroll
      for (int m = 0; m < 3; ++m) { e.w[m] = (i == m) ? 1.f : 0.f; e.v[m] = (i == m + 3) ? 1.f : 0.f; }
      SV VR, VL;
      impulse_response<Model, false, true, false, false>(jc, L, zero, e, VR, VL, nullptr, nullptr);
      put_l(i, VL);
    }
#endif
    SS_PROF(7);
    // foot twists under the free velocities
    float V[12];
    {
      SV a = v0f;
      static_for<0, 3>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        a = xmotion<Model, j>(jc.r[j].cs, jc.r[j].sn, a);
        a.w[kAxis[j]] += L.s(S_QDF + j);
      });
      SV b = a;
      static_for<3, 8>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        a = xmotion<Model, j>(jc.r[j].cs, jc.r[j].sn, a);
        a.w[kAxis[j]] += L.s(S_QDF + j);
      });
      static_for<8, 13>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        b = xmotion<Model, j>(jc.r[j].cs, jc.r[j].sn, b);
        b.w[kAxis[j]] += L.s(S_QDF + j);
      });
#pragma unroll
      for (int i = 0; i < 3; ++i) { V[i] = a.w[i]; V[3 + i] = a.v[i]; V[6 + i] = b.w[i]; V[9 + i] = b.v[i]; }
    }
    // rows -> LDS: per (contact, direction) 3 float4: y_own[6], dir[3], 1/A, b_n
    static_for<0, 8>([&](auto Kc) {
      constexpr int ck = decltype(Kc)::value, f = ck / 4, k = ck % 4;
      if (active & (1 << ck)) {
        constexpr float cx = Model::corners[k][0], cy = Model::corners[k][1], cz = Model::corners[k][2];
        const int sl = (cslot >> (2 * ck)) & 3;
        float n[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) n[i] = L.s(S_STN + sl * 3 + i);
        float t1[3] = {1.f - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
        float inv = SS_RSQRT(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
        t1[0] *= inv; t1[1] *= inv; t1[2] *= inv;
        float t2[3];
        cross(n, t1, t2);
        float corr = fmaxf(pen[ck] - kSlop, 0.f);
        const float bnv = fminf(kErp * corr * (1.0f / kH), kVcorrMax);
        static_for<0, 3>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          const float* dir = d == 0 ? n : (d == 1 ? t1 : t2);
          float w[6];
          // direction in the foot frame: R_f^T dir
#pragma unroll
          for (int c = 0; c < 3; ++c) w[3 + c] = Rf[f][0][c] * dir[0] + Rf[f][1][c] * dir[1] + Rf[f][2][c] * dir[2];
          // r x d
          w[0] = cy * w[5] - cz * w[4];
          w[1] = cz * w[3] - cx * w[5];
          w[2] = cx * w[4] - cy * w[3];
          float y[6];
#pragma unroll
          for (int o = 0; o < 6; ++o) y[o] = 0.f;
#pragma unroll
          for (int l = 0; l < 6; ++l) {
            if constexpr (f == 0) {
              float4 c0 = L.q4(kLdsLinv + l * 3 + 0), c1 = L.q4(kLdsLinv + l * 3 + 1);
              y[0] += c0.x * w[l]; y[1] += c0.y * w[l]; y[2] += c0.z * w[l]; y[3] += c0.w * w[l];
              y[4] += c1.x * w[l]; y[5] += c1.y * w[l];
            } else {
              float4 c1 = L.q4(kLdsLinv + (6 + l) * 3 + 1), c2 = L.q4(kLdsLinv + (6 + l) * 3 + 2);
              y[0] += c1.z * w[l]; y[1] += c1.w * w[l];
              y[2] += c2.x * w[l]; y[3] += c2.y * w[l]; y[4] += c2.z * w[l]; y[5] += c2.w * w[l];
            }
          }
          float A = 0.f;
#pragma unroll
          for (int l = 0; l < 6; ++l) A += w[l] * y[l];
          constexpr int row = kLdsRows + (ck * 3 + d) * 3;
          L.q4(row + 0) = make_float4(y[0], y[1], y[2], y[3]);
          L.q4(row + 1) = make_float4(y[4], y[5], w[3], w[4]);
          L.q4(row + 2) = make_float4(w[5], 1.0f / A, d == 0 ? bnv : 0.f, 0.f);
        });
      } else {   // inactive contact: all-zero rows make every PGS update a no-op, so the sweeps stay branch-free
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 9; ++i) L.q4(kLdsRows + ck * 9 + i) = z;
      }
    });
    SS_PROF(8);
    // projected Gauss-Seidel.  Rows of foot f only read V_f, so the effect of foot f's impulses on the other
    // foot's twist is applied once per sweep of foot f (exactly equivalent to updating it row by row).
    float lam[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) lam[k][0] = lam[k][1] = lam[k][2] = 0.f;
    SV W[2] = {zero, zero};          // accumulated foot wrenches
    constexpr float mu = Model::friction;
#pragma unroll 1
    for (int it = 0; it < kPgsIters; ++it) {
      static_for<0, 2>([&](auto Fc) {
        constexpr int f = decltype(Fc)::value;
#ifndef SS_PGS_BRANCHFREE
        if ((active >> (4 * f)) & 15)
#endif
        {
          SV dW = zero;
          float* Vw = V + f * 6;
          float* Vv = V + f * 6 + 3;
          static_for<0, 4>([&](auto Kc) {
            constexpr int k = decltype(Kc)::value, ck = f * 4 + k;
            constexpr float cx = Model::corners[k][0], cy = Model::corners[k][1], cz = Model::corners[k][2];
#ifndef SS_PGS_BRANCHFREE
            if (active & (1 << ck))
#endif
            {
              float fc[3] = {0.f, 0.f, 0.f};
              static_for<0, 3>([&](auto Dc) {
                constexpr int d = decltype(Dc)::value;
                constexpr int row = kLdsRows + (ck * 3 + d) * 3;
                float4 r0 = L.q4(row + 0), r1 = L.q4(row + 1), r2 = L.q4(row + 2);
                // velocity of the corner: v + w x r, projected on the row direction
                float px = Vv[0] + Vw[1] * cz - Vw[2] * cy;
                float py = Vv[1] + Vw[2] * cx - Vw[0] * cz;
                float pz = Vv[2] + Vw[0] * cy - Vw[1] * cx;
                float vrel = r1.z * px + r1.w * py + r2.x * pz;
                float ln = lam[ck][d] + (r2.z - vrel) * r2.y;
                if constexpr (d == 0) {
                  ln = fmaxf(ln, 0.f);
                } else {
                  float lim = mu * lam[ck][0];
                  ln = fminf(fmaxf(ln, -lim), lim);
                }
                float dl = ln - lam[ck][d];
                lam[ck][d] = ln;
                Vw[0] += r0.x * dl; Vw[1] += r0.y * dl; Vw[2] += r0.z * dl;
                Vv[0] += r0.w * dl; Vv[1] += r1.x * dl; Vv[2] += r1.y * dl;
                fc[0] += r1.z * dl; fc[1] += r1.w * dl; fc[2] += r2.x * dl;
              });
              dW.v[0] += fc[0]; dW.v[1] += fc[1]; dW.v[2] += fc[2];
              dW.w[0] += cy * fc[2] - cz * fc[1];
              dW.w[1] += cz * fc[0] - cx * fc[2];
              dW.w[2] += cx * fc[1] - cy * fc[0];
            }
          });
          // cross-foot coupling through LR = d V_L / d W_R (and its transpose)
          const float dw[6] = {dW.w[0], dW.w[1], dW.w[2], dW.v[0], dW.v[1], dW.v[2]};
#pragma unroll
          for (int l = 0; l < 6; ++l) {
            float4 c1 = L.q4(kLdsLinv + l * 3 + 1), c2 = L.q4(kLdsLinv + l * 3 + 2);
            if constexpr (f == 0) {       // V_L += LR[:, l] * dW_R[l]
              V[6] += c1.z * dw[l]; V[7] += c1.w * dw[l]; V[8] += c2.x * dw[l];
              V[9] += c2.y * dw[l]; V[10] += c2.z * dw[l]; V[11] += c2.w * dw[l];
            } else {                      // V_R[l] += LR[:, l] . dW_L
              V[l] += c1.z * dw[0] + c1.w * dw[1] + c2.x * dw[2] + c2.y * dw[3] + c2.z * dw[4] + c2.w * dw[5];
            }
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) { W[f].w[i] += dW.w[i]; W[f].v[i] += dW.v[i]; }
        }
      });
    }
    SS_PROF(9);
    // accumulated foot wrenches -> whole tree
    SV VR, VL;
    impulse_response<Model, true, true, true, true>(jc, L, W[0], W[1], VR, VL, &dv0, dqd);
  }
  SS_MEMBAR();
  SS_PROF(10);

  // ---- integrate (semi-implicit Euler), state back to LDS
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float qd = L.s(S_QDF + j) + dqd[j];
    L.s(S_QD + j) = qd;
    L.s(S_Q + j) += h * qd;
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

// move the control-step state between registers and its LDS home
SSD void dyn_to_lds(const Dyn& s, const Stones& st, const Lds& L) {
#pragma unroll
  for (int i = 0; i < 3; ++i) { L.s(S_POS + i) = s.pos[i]; L.s(S_VW + i) = s.v0.w[i]; L.s(S_VV + i) = s.v0.v[i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) L.s(S_QUAT + i) = s.quat[i];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { L.s(S_Q + j) = s.q[j]; L.s(S_QD + j) = s.qd[j]; }
#pragma unroll
  for (int sl = 0; sl < 3; ++sl)
#pragma unroll
    for (int i = 0; i < 3; ++i) { L.s(S_STP + sl * 3 + i) = st.p[sl][i]; L.s(S_STN + sl * 3 + i) = st.nrm[sl][i]; }
}
SSD void dyn_from_lds(Dyn& s, const Lds& L) {
#pragma unroll
  for (int i = 0; i < 3; ++i) { s.pos[i] = L.s(S_POS + i); s.v0.w[i] = L.s(S_VW + i); s.v0.v[i] = L.s(S_VV + i); }
#pragma unroll
  for (int i = 0; i < 4; ++i) s.quat[i] = L.s(S_QUAT + i);
#pragma unroll
  for (int j = 0; j < NJ; ++j) { s.q[j] = L.s(S_Q + j); s.qd[j] = L.s(S_QD + j); }
}

}  // namespace ss
