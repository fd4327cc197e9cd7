// ss_dynamics.hpp -- one physics substep (docs/PHYSICS.md section 3) for one environment per lane.
//
// Structure (all loops over the kinematic tree are unrolled at compile time):
//   pass 1  link velocities                     root -> leaves
//   pass 2  articulated inertias / bias forces  leaves -> root    (U, 1/D, u kept per joint)
//   base    6x6 Cholesky solve
//   pass 3  accelerations                        root -> leaves   -> free velocities
//   detect  sole corners vs the three active stones
//   solve   Lambda^-1 by 12 unit impulse responses (columns parked in LDS), contact rows precomputed into
//           LDS, 8 projected Gauss-Seidel sweeps in the 12-dim foot-twist space, one whole-tree impulse
//           response to apply the foot wrenches
//   integrate (semi-implicit Euler)
//
// Register / LDS budget.  One wavefront per workgroup, one workgroup per CU: each lane owns 512 VGPR+AGPR and a
// private 2560-byte column of the CU's 160 KiB LDS (slot-major float4, lane stride 16 B: every ds_read/write_b128
// is conflict-free and no barrier is ever needed).  What does not fit the register file lives in that column:
//   slots   0..43   link velocities (ABA phase)        aliased with
//   slots   0..35   Lambda^-1 columns (contact phase)
//   slots  36..131  24 contact rows x 4 float4: y[12], dir[3], 1/A
//   slots 132..133  normal-row targets b_n[8]
//   slots 134..153  per-joint cache of the 8 arm joints (cs, sn, U[6], 1/D, u)
//   slots 154..159  the 21 clipped actions of this control step
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
constexpr int kLdsVel = 0;
constexpr int kLdsLinv = 0;
constexpr int kLdsRows = 36;
constexpr int kLdsBn = 132;
constexpr int kLdsArms = 134;
constexpr int kLdsAct = 154;
constexpr int kLdsSlots = 160;         // 160 float4 = 2560 B per lane = 163,840 B per wavefront (all of the CU's LDS)
constexpr int kNumLegJoints = 13;      // joints 0..12 (spine + legs) stay in registers, 13..20 (arms) live in LDS

struct Lds {       // lane-private view of the workgroup's LDS
  float4* base;
  int lane;
  SSD float4& q(int slot) const { return base[slot * kWave + lane]; }
  SSD float& f(int slot, int comp) const { return reinterpret_cast<float*>(base + slot * kWave + lane)[comp]; }
  SSD float& flat(int slot0, int idx) const { return f(slot0 + idx / 4, idx % 4); }
};

struct Dyn {       // dynamic state of one env
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

template <int J>
SSD JRec jrec_get(const JointCache& jc, const Lds& L) {
  if constexpr (J < kNumLegJoints) {
    return jc.r[J];
  } else {
    constexpr int o = (J - kNumLegJoints) * 10;
    JRec r;
    r.cs = L.flat(kLdsArms, o + 0); r.sn = L.flat(kLdsArms, o + 1);
    r.Uw[0] = L.flat(kLdsArms, o + 2); r.Uw[1] = L.flat(kLdsArms, o + 3); r.Uw[2] = L.flat(kLdsArms, o + 4);
    r.Uv[0] = L.flat(kLdsArms, o + 5); r.Uv[1] = L.flat(kLdsArms, o + 6); r.Uv[2] = L.flat(kLdsArms, o + 7);
    r.Dinv = L.flat(kLdsArms, o + 8); r.u = L.flat(kLdsArms, o + 9);
    return r;
  }
}
template <int J>
SSD void jrec_put(JointCache& jc, const Lds& L, const JRec& r) {
  if constexpr (J < kNumLegJoints) {
    jc.r[J] = r;
  } else {
    constexpr int o = (J - kNumLegJoints) * 10;
    L.flat(kLdsArms, o + 0) = r.cs; L.flat(kLdsArms, o + 1) = r.sn;
    L.flat(kLdsArms, o + 2) = r.Uw[0]; L.flat(kLdsArms, o + 3) = r.Uw[1]; L.flat(kLdsArms, o + 4) = r.Uw[2];
    L.flat(kLdsArms, o + 5) = r.Uv[0]; L.flat(kLdsArms, o + 6) = r.Uv[1]; L.flat(kLdsArms, o + 7) = r.Uv[2];
    L.flat(kLdsArms, o + 8) = r.Dinv; L.flat(kLdsArms, o + 9) = r.u;
  }
}
// cos / sin only (pass 1 runs before U, 1/D, u exist)
template <int J>
SSD void jcs_get(const JointCache& jc, const Lds& L, float& c, float& s) {
  if constexpr (J < kNumLegJoints) { c = jc.r[J].cs; s = jc.r[J].sn; }
  else { c = L.flat(kLdsArms, (J - kNumLegJoints) * 10 + 0); s = L.flat(kLdsArms, (J - kNumLegJoints) * 10 + 1); }
}

template <int B>
SSD SV vel_get(const Lds& L) {
  float4 a = L.q(kLdsVel + 2 * B), b = L.q(kLdsVel + 2 * B + 1);
  SV v = {{a.x, a.y, a.z}, {b.x, b.y, b.z}};
  return v;
}
template <int B>
SSD void vel_put(const Lds& L, const SV& v) {
  L.q(kLdsVel + 2 * B) = make_float4(v.w[0], v.w[1], v.w[2], 0.f);
  L.q(kLdsVel + 2 * B + 1) = make_float4(v.v[0], v.v[1], v.v[2], 0.f);
}

// ---------------------------------------------------------------------------------------------------------------
// ABA impulse response restricted to what the contact stage needs.
//   fR / fL : spatial impulses on the right / left foot (foot frame); LOAD_* says which are non-zero
//   outputs : foot twists VR, VL; if FULL also dv0 and dqd[21] (whole tree, arms included)
template <class Model, bool LOAD_R, bool LOAD_L, bool FULL>
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
  {
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
// power: torque scale; the clipped actions of this control step sit in LDS (kLdsAct)
template <class Model>
SSD void substep(Dyn& s, float power, const Stones& st, FootReport& fr, const Lds& L) {
  constexpr float h = kH;
  JointCache jc;
  static_for<0, NJ>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value;
    float sn, cs;
    sincosf(s.q[j], &sn, &cs);
    if constexpr (j < kNumLegJoints) { jc.r[j].cs = cs; jc.r[j].sn = sn; }
    else { L.flat(kLdsArms, (j - kNumLegJoints) * 10 + 0) = cs; L.flat(kLdsArms, (j - kNumLegJoints) * 10 + 1) = sn; }
  });

  // ---- pass 1: velocities (kept in LDS; the chain predecessor stays in registers)
  vel_put<0>(L, s.v0);
  {
    SV prev = s.v0;
    static_for<0, NJ>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
      float c, sn;
      jcs_get<j>(jc, L, c, sn);
      SV vp;
      if constexpr (p == j) vp = prev;               // parent is the body processed just before
      else if constexpr (p == 0) vp = s.v0;
      else vp = vel_get<p>(L);
      SV v = xmotion<Model, j>(c, sn, vp);
      v.w[ax] += s.qd[j];
      vel_put<b>(L, v);
      prev = v;
      SS_FENCE();
    });
  }

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
    float q = s.q[j], qd = s.qd[j];
    float viol = q > hi ? q - hi : (q < lo ? q - lo : 0.f);
    bool lim = viol != 0.f;
    float kl = lim ? klim : 0.f, dl = lim ? dlim : 0.f;
    float tau_m = power * tq * L.flat(kLdsAct, j);
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

  // ---- base
  SV a0;
  {
    ABI I0 = acc[0];
    abi_add_body<Model, 0>(I0);
    SV pb = body_bias<Model, 0>(s.v0);
    SV p0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { p0.w[i] = pacc[0].w[i] + pb.w[i]; p0.v[i] = pacc[0].v[i] + pb.v[i]; }
    float M[6][6];
    abi_dense(I0, M);
    jc.L0 = chol6(M);
    a0 = chol6_solve_neg(jc.L0, p0);
  }
  SS_FENCE();

  // ---- pass 3: accelerations -> free velocities
  float qdf[NJ];
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
      float qd = s.qd[j];
      a.w[ai] += qd * vb.w[aj]; a.w[aj] -= qd * vb.w[ai];
      a.v[ai] += qd * vb.v[aj]; a.v[aj] -= qd * vb.v[ai];
      float dotv = r.Uw[0] * a.w[0] + r.Uw[1] * a.w[1] + r.Uw[2] * a.w[2] + r.Uv[0] * a.v[0] + r.Uv[1] * a.v[1] +
                   r.Uv[2] * a.v[2];
      float qdd = r.Dinv * (r.u - dotv);
      a.w[ax] += qdd;
      if constexpr (b == 3) a3 = a;
      prev = a;
      qdf[j] = qd + h * qdd;
      SS_FENCE();
    });
  }
  float Rb[3][3];
  quat_rot(s.quat, Rb);
  SV v0f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v0f.w[i] = s.v0.w[i] + h * a0.w[i];
    v0f.v[i] = s.v0.v[i] + h * (a0.v[i] - kGrav * Rb[2][i]);   // R^T g = -9.8 * (third row of R)
  }

  // ---- detect: FK of spine + legs, sole corners vs stones
  float Rf[2][3][3], pf[2][3];
  {
    float Rw[14][3][3], pw[14][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      pw[0][a] = s.pos[a];
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
  SS_FENCE();
  int active = 0;          // bit k
  int cslot = 0;           // 2 bits per contact: stone slot
  float pen[8];
  fr.contact = 0;
  fr.on_target = 0;
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
        float dx = P[0] - st.p[sl][0], dy = P[1] - st.p[sl][1], dz = P[2] - st.p[sl][2];
        float d = dx * st.nrm[sl][0] + dy * st.nrm[sl][1] + dz * st.nrm[sl][2];
        float lx = dx - d * st.nrm[sl][0], ly = dy - d * st.nrm[sl][1], lz = dz - d * st.nrm[sl][2];
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

  // ---- contact solve
  float dqd[NJ];
  SV dv0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) dqd[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { dv0.w[i] = 0.f; dv0.v[i] = 0.f; }
  if (active != 0) {
    // Lambda^-1 columns -> LDS
    SV zero = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
      SV e;
#pragma unroll
      for (int m = 0; m < 3; ++m) { e.w[m] = (i == m) ? 1.f : 0.f; e.v[m] = (i == m + 3) ? 1.f : 0.f; }
      SV VR, VL;
      impulse_response<Model, true, false, false>(jc, L, e, zero, VR, VL, nullptr, nullptr);
      L.q(kLdsLinv + i * 3 + 0) = make_float4(VR.w[0], VR.w[1], VR.w[2], VR.v[0]);
      L.q(kLdsLinv + i * 3 + 1) = make_float4(VR.v[1], VR.v[2], VL.w[0], VL.w[1]);
      L.q(kLdsLinv + i * 3 + 2) = make_float4(VL.w[2], VL.v[0], VL.v[1], VL.v[2]);
    }
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
      SV e;
#pragma unroll
      for (int m = 0; m < 3; ++m) { e.w[m] = (i == m) ? 1.f : 0.f; e.v[m] = (i == m + 3) ? 1.f : 0.f; }
      SV VR, VL;
      impulse_response<Model, false, true, false>(jc, L, zero, e, VR, VL, nullptr, nullptr);
      L.q(kLdsLinv + (6 + i) * 3 + 0) = make_float4(VR.w[0], VR.w[1], VR.w[2], VR.v[0]);
      L.q(kLdsLinv + (6 + i) * 3 + 1) = make_float4(VR.v[1], VR.v[2], VL.w[0], VL.w[1]);
      L.q(kLdsLinv + (6 + i) * 3 + 2) = make_float4(VL.w[2], VL.v[0], VL.v[1], VL.v[2]);
    }
    // foot twists under the free velocities
    float V[12];
    {
      SV a = v0f;
      static_for<0, 3>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        a = xmotion<Model, j>(jc.r[j].cs, jc.r[j].sn, a);
        a.w[kAxis[j]] += qdf[j];
      });
      SV b = a;
      static_for<3, 8>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        a = xmotion<Model, j>(jc.r[j].cs, jc.r[j].sn, a);
        a.w[kAxis[j]] += qdf[j];
      });
      static_for<8, 13>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        b = xmotion<Model, j>(jc.r[j].cs, jc.r[j].sn, b);
        b.w[kAxis[j]] += qdf[j];
      });
#pragma unroll
      for (int i = 0; i < 3; ++i) { V[i] = a.w[i]; V[3 + i] = a.v[i]; V[6 + i] = b.w[i]; V[9 + i] = b.v[i]; }
    }
    SS_FENCE();
    // rows -> LDS
    static_for<0, 8>([&](auto Kc) {
      constexpr int ck = decltype(Kc)::value, f = ck / 4, k = ck % 4;
      if (active & (1 << ck)) {
        constexpr float cx = Model::corners[k][0], cy = Model::corners[k][1], cz = Model::corners[k][2];
        const int sl = (cslot >> (2 * ck)) & 3;
        float n[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) n[i] = sl == 0 ? st.nrm[0][i] : (sl == 1 ? st.nrm[1][i] : st.nrm[2][i]);
        float t1[3] = {1.f - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
        float inv = SS_RSQRT(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
        t1[0] *= inv; t1[1] *= inv; t1[2] *= inv;
        float t2[3];
        cross(n, t1, t2);
        float corr = fmaxf(pen[ck] - kSlop, 0.f);
        L.flat(kLdsBn, ck) = fminf(kErp * corr * (1.0f / kH), kVcorrMax);
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
          float y[12];
#pragma unroll
          for (int o = 0; o < 12; ++o) y[o] = 0.f;
#pragma unroll
          for (int l = 0; l < 6; ++l) {
            float4 c0 = L.q(kLdsLinv + (f * 6 + l) * 3 + 0);
            float4 c1 = L.q(kLdsLinv + (f * 6 + l) * 3 + 1);
            float4 c2 = L.q(kLdsLinv + (f * 6 + l) * 3 + 2);
            y[0] += c0.x * w[l]; y[1] += c0.y * w[l]; y[2] += c0.z * w[l]; y[3] += c0.w * w[l];
            y[4] += c1.x * w[l]; y[5] += c1.y * w[l]; y[6] += c1.z * w[l]; y[7] += c1.w * w[l];
            y[8] += c2.x * w[l]; y[9] += c2.y * w[l]; y[10] += c2.z * w[l]; y[11] += c2.w * w[l];
          }
          float A = 0.f;
#pragma unroll
          for (int l = 0; l < 6; ++l) A += w[l] * y[f * 6 + l];
          constexpr int row = kLdsRows + (ck * 3 + d) * 4;
          L.q(row + 0) = make_float4(y[0], y[1], y[2], y[3]);
          L.q(row + 1) = make_float4(y[4], y[5], y[6], y[7]);
          L.q(row + 2) = make_float4(y[8], y[9], y[10], y[11]);
          L.q(row + 3) = make_float4(w[3], w[4], w[5], 1.0f / A);
        });
      }
      SS_FENCE();
    });
    // projected Gauss-Seidel
    float lam[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) lam[k][0] = lam[k][1] = lam[k][2] = 0.f;
    constexpr float mu = Model::friction;
#pragma unroll 1
    for (int it = 0; it < kPgsIters; ++it) {
      static_for<0, 8>([&](auto Kc) {
        constexpr int ck = decltype(Kc)::value, f = ck / 4, k = ck % 4;
        constexpr float cx = Model::corners[k][0], cy = Model::corners[k][1], cz = Model::corners[k][2];
        if (active & (1 << ck)) {
          const float bn = L.flat(kLdsBn, ck);
          static_for<0, 3>([&](auto Dc) {
            constexpr int d = decltype(Dc)::value;
            constexpr int row = kLdsRows + (ck * 3 + d) * 4;
            float4 y0 = L.q(row + 0), y1 = L.q(row + 1), y2 = L.q(row + 2), w1 = L.q(row + 3);
            // velocity of the corner: v + w x r, projected on the row direction
            const float* Vw = V + f * 6;
            const float* Vv = V + f * 6 + 3;
            float px = Vv[0] + Vw[1] * cz - Vw[2] * cy;
            float py = Vv[1] + Vw[2] * cx - Vw[0] * cz;
            float pz = Vv[2] + Vw[0] * cy - Vw[1] * cx;
            float vrel = w1.x * px + w1.y * py + w1.z * pz;
            float target = d == 0 ? bn : 0.f;
            float ln = lam[ck][d] + (target - vrel) * w1.w;
            if constexpr (d == 0) {
              ln = fmaxf(ln, 0.f);
            } else {
              float lim = mu * lam[ck][0];
              ln = fminf(fmaxf(ln, -lim), lim);
            }
            float dl = ln - lam[ck][d];
            lam[ck][d] = ln;
            V[0] += y0.x * dl; V[1] += y0.y * dl; V[2] += y0.z * dl; V[3] += y0.w * dl;
            V[4] += y1.x * dl; V[5] += y1.y * dl; V[6] += y1.z * dl; V[7] += y1.w * dl;
            V[8] += y2.x * dl; V[9] += y2.y * dl; V[10] += y2.z * dl; V[11] += y2.w * dl;
          });
        }
      });
    }
    // accumulated foot wrenches -> whole tree
    SV WR = zero, WL = zero;
    static_for<0, 8>([&](auto Kc) {
      constexpr int ck = decltype(Kc)::value, f = ck / 4, k = ck % 4;
      constexpr float cx = Model::corners[k][0], cy = Model::corners[k][1], cz = Model::corners[k][2];
      if (active & (1 << ck)) {
        // total contact force at this corner (foot frame), then its moment about the foot origin
        float fx = 0.f, fy = 0.f, fz = 0.f;
        static_for<0, 3>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          float4 w1 = L.q(kLdsRows + (ck * 3 + d) * 4 + 3);
          fx += w1.x * lam[ck][d]; fy += w1.y * lam[ck][d]; fz += w1.z * lam[ck][d];
        });
        SV& W = f == 0 ? WR : WL;
        W.v[0] += fx; W.v[1] += fy; W.v[2] += fz;
        W.w[0] += cy * fz - cz * fy;
        W.w[1] += cz * fx - cx * fz;
        W.w[2] += cx * fy - cy * fx;
      }
    });
    SV VR, VL;
    impulse_response<Model, true, true, true>(jc, L, WR, WL, VR, VL, &dv0, dqd);
  }

  // ---- integrate (semi-implicit Euler)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s.qd[j] = qdf[j] + dqd[j];
    s.q[j] += h * s.qd[j];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { s.v0.w[i] = v0f.w[i] + dv0.w[i]; s.v0.v[i] = v0f.v[i] + dv0.v[i]; }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    s.pos[r] += h * (Rb[r][0] * s.v0.v[0] + Rb[r][1] * s.v0.v[1] + Rb[r][2] * s.v0.v[2]);
  {
    float qw = s.quat[0], qx = s.quat[1], qy = s.quat[2], qz = s.quat[3];
    float ox = s.v0.w[0], oy = s.v0.w[1], oz = s.v0.w[2], hh = 0.5f * h;
    float nw = qw + hh * (-qx * ox - qy * oy - qz * oz);
    float nx = qx + hh * (qw * ox + qy * oz - qz * oy);
    float ny = qy + hh * (qw * oy - qx * oz + qz * ox);
    float nz = qz + hh * (qw * oz + qx * oy - qy * ox);
    float inv = SS_RSQRT(nw * nw + nx * nx + ny * ny + nz * nz);
    s.quat[0] = nw * inv; s.quat[1] = nx * inv; s.quat[2] = ny * inv; s.quat[3] = nz * inv;
  }
}

}  // namespace ss
