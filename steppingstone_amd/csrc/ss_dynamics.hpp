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
// LDS is used as per-lane private storage: slot-major float4 columns, lane stride 16 B (conflict-free
// ds_read_b128 / ds_write_b128, no barriers because a workgroup is a single wavefront).
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
// LDS map (float4 slots per lane)
constexpr int kLdsLinv = 0;            // 12 columns x 3 float4
constexpr int kLdsRows = 36;           // 24 rows x 5 float4: y[12], w[6], 1/A, b
constexpr int kLdsSlots = 36 + 120;    // 156 float4 = 2496 B per lane = 159,744 B per wavefront

struct Dyn {       // dynamic state of one env
  float pos[3];
  float quat[4];
  SV v0;           // base twist, body coordinates
  float q[NJ];
  float qd[NJ];
};

struct Stones {    // the three active stones n-1, n, n+1
  float p[3][3];   // centre
  float n[3][3];   // unit normal
};

struct FootReport {
  int contact;     // bit f: foot f has a contact
  int on_target;   // bit f: foot f touches stone n (slot 1)
  float sole[2][3];
};

struct JointCache {  // kept from the ABA for the impulse responses
  float cs[NJ], sn[NJ];
  float Uw[NJ][3], Uv[NJ][3], Dinv[NJ];
  Chol6 L0;
};

#define LDS4(slot) lds4[(slot) * kWave + lane]

// ---------------------------------------------------------------------------------------------------------------
// ABA impulse response restricted to what the contact stage needs.
//   fR / fL : spatial impulses on the right / left foot (foot frame); LOAD_* says which are non-zero
//   outputs : foot twists VR, VL; if FULL also dv0 and dqd[21] (whole tree, arms included)
template <class Model, bool LOAD_R, bool LOAD_L, bool FULL>
SSD void impulse_response(const JointCache& jc, const SV& fR, const SV& fL, SV& VR, SV& VL, SV* dv0, float* dqd) {
  float ul[NJ];   // only leg + spine entries are ever non-zero
  SV pPel;        // impulse bias accumulated at the pelvis (body 3)
  bool pel_init = false;
  auto leg_up = [&](auto J0c, const SV& f) {
    constexpr int j0 = decltype(J0c)::value;
    SV p = {{-f.w[0], -f.w[1], -f.w[2]}, {-f.v[0], -f.v[1], -f.v[2]}};
    static_rfor<j0 + 4, j0>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value, ax = kAxis[j];
      float u = -p.w[ax];
      ul[j] = u;
      float du = jc.Dinv[j] * u;
      SV pa;
#pragma unroll
      for (int i = 0; i < 3; ++i) { pa.w[i] = p.w[i] + jc.Uw[j][i] * du; pa.v[i] = p.v[i] + jc.Uv[j][i] * du; }
      p = xforce<Model, j>(jc.cs[j], jc.sn[j], pa);
    });
    return p;
  };
  if constexpr (LOAD_R) { pPel = leg_up(std::integral_constant<int, 3>{}, fR); pel_init = true; }
  if constexpr (LOAD_L) {
    SV t = leg_up(std::integral_constant<int, 8>{}, fL);
    if constexpr (LOAD_R) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { pPel.w[i] += t.w[i]; pPel.v[i] += t.v[i]; }
    } else {
      pPel = t;
    }
  }
  (void)pel_init;
  // spine 2,1,0
  SV p = pPel;
  static_rfor<2, 0>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value, ax = kAxis[j];
    float u = -p.w[ax];
    ul[j] = u;
    float du = jc.Dinv[j] * u;
    SV pa;
#pragma unroll
    for (int i = 0; i < 3; ++i) { pa.w[i] = p.w[i] + jc.Uw[j][i] * du; pa.v[i] = p.v[i] + jc.Uv[j][i] * du; }
    p = xforce<Model, j>(jc.cs[j], jc.sn[j], pa);
  });
  SV d0 = chol6_solve_neg(jc.L0, p);
  if constexpr (FULL) *dv0 = d0;
  // down
  auto down = [&](auto Jc, const SV& dpar, bool) {
    constexpr int j = decltype(Jc)::value, ax = kAxis[j];
    constexpr bool loaded = (j <= 2) || (LOAD_R && j >= 3 && j <= 7) || (LOAD_L && j >= 8 && j <= 12);
    SV d = xmotion<Model, j>(jc.cs[j], jc.sn[j], dpar);
    float dotv = jc.Uw[j][0] * d.w[0] + jc.Uw[j][1] * d.w[1] + jc.Uw[j][2] * d.w[2] + jc.Uv[j][0] * d.v[0] +
                 jc.Uv[j][1] * d.v[1] + jc.Uv[j][2] * d.v[2];
    float dq;
    if constexpr (loaded) dq = jc.Dinv[j] * (ul[j] - dotv);
    else dq = -jc.Dinv[j] * dotv;
    d.w[ax] += dq;
    if constexpr (FULL) dqd[j] = dq;
    return d;
  };
  SV d1 = down(std::integral_constant<int, 0>{}, d0, true);
  SV d2 = down(std::integral_constant<int, 1>{}, d1, true);
  SV d3 = down(std::integral_constant<int, 2>{}, d2, true);
  {
    SV a = down(std::integral_constant<int, 3>{}, d3, true);
    a = down(std::integral_constant<int, 4>{}, a, true);
    a = down(std::integral_constant<int, 5>{}, a, true);
    a = down(std::integral_constant<int, 6>{}, a, true);
    VR = down(std::integral_constant<int, 7>{}, a, true);
  }
  {
    SV a = down(std::integral_constant<int, 8>{}, d3, true);
    a = down(std::integral_constant<int, 9>{}, a, true);
    a = down(std::integral_constant<int, 10>{}, a, true);
    a = down(std::integral_constant<int, 11>{}, a, true);
    VL = down(std::integral_constant<int, 12>{}, a, true);
  }
  if constexpr (FULL) {
    SV a = down(std::integral_constant<int, 13>{}, d0, true);
    a = down(std::integral_constant<int, 14>{}, a, true);
    a = down(std::integral_constant<int, 15>{}, a, true);
    a = down(std::integral_constant<int, 16>{}, a, true);
    a = down(std::integral_constant<int, 17>{}, d0, true);
    a = down(std::integral_constant<int, 18>{}, a, true);
    a = down(std::integral_constant<int, 19>{}, a, true);
    a = down(std::integral_constant<int, 20>{}, a, true);
    (void)a;
  }
}

// ---------------------------------------------------------------------------------------------------------------
template <class Model>
SSD void substep(Dyn& s, const float (&tau_m)[NJ], const Stones& st, FootReport& fr, float4* lds4, int lane) {
  constexpr float h = kH;
  JointCache jc;
#pragma unroll
  for (int j = 0; j < NJ; ++j) sincosf(s.q[j], &jc.sn[j], &jc.cs[j]);

  // ---- pass 1: velocities
  SV vel[NB];
  vel[0] = s.v0;
  static_for<0, NJ>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
    vel[b] = xmotion<Model, j>(jc.cs[j], jc.sn[j], vel[p]);
    vel[b].w[ax] += s.qd[j];
  });

  // ---- pass 2: articulated inertias
  ABI acc[NB];
  SV pacc[NB];
  float uu[NJ];
  static_rfor<NJ - 1, 0>([&](auto Jc) {
    constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
    constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
    constexpr bool leaf = first_child(b) < 0;
    constexpr bool massive = Model::mass[b] != 0.f;
    ABI I;
    SV pA;
    if constexpr (leaf) {
      I = abi_body<Model, b>();
      pA = body_bias<Model, b>(vel[b]);
    } else {
      I = acc[b];
      pA = pacc[b];
      if constexpr (massive) {
        abi_add_body<Model, b>(I);
        SV pb = body_bias<Model, b>(vel[b]);
#pragma unroll
        for (int i = 0; i < 3; ++i) { pA.w[i] += pb.w[i]; pA.v[i] += pb.v[i]; }
      }
    }
    // joint torque (explicit part) and implicit diagonal, PHYSICS.md 3.1
    constexpr float lo = Model::lo[j], hi = Model::hi[j], kd = Model::damping[j], ks = Model::stiffness[j];
    constexpr float klim = Model::klim[j], dlim = Model::dlim[j], arm = Model::armature[j];
    float q = s.q[j], qd = s.qd[j];
    float viol = q > hi ? q - hi : (q < lo ? q - lo : 0.f);
    bool lim = viol != 0.f;
    float kl = lim ? klim : 0.f, dl = lim ? dlim : 0.f;
    float tau = tau_m[j] - kd * qd - ks * (q + h * qd) - kl * (viol + h * qd) - dl * qd;
    float Dadd = arm + h * (kd + dl) + (h * h) * (ks + kl);
    // U = I S
    float Uw[3] = {I.A.template get<0, ax>(), I.A.template get<1, ax>(), I.A.template get<2, ax>()};
    float Uv[3] = {I.B[ax][0], I.B[ax][1], I.B[ax][2]};
    float Dinv = 1.0f / (Uw[ax] + Dadd);
    float u = tau - pA.w[ax];
#pragma unroll
    for (int i = 0; i < 3; ++i) { jc.Uw[j][i] = Uw[i]; jc.Uv[j][i] = Uv[i]; }
    jc.Dinv[j] = Dinv;
    uu[j] = u;
    // Ia = I - U Dinv U^T
    float sw[3] = {Dinv * Uw[0], Dinv * Uw[1], Dinv * Uw[2]};
    float sv[3] = {Dinv * Uv[0], Dinv * Uv[1], Dinv * Uv[2]};
    I.A.m[0] -= sw[0] * Uw[0]; I.A.m[1] -= sw[1] * Uw[1]; I.A.m[2] -= sw[2] * Uw[2];
    I.A.m[3] -= sw[0] * Uw[1]; I.A.m[4] -= sw[0] * Uw[2]; I.A.m[5] -= sw[1] * Uw[2];
    I.C.m[0] -= sv[0] * Uv[0]; I.C.m[1] -= sv[1] * Uv[1]; I.C.m[2] -= sv[2] * Uv[2];
    I.C.m[3] -= sv[0] * Uv[1]; I.C.m[4] -= sv[0] * Uv[2]; I.C.m[5] -= sv[1] * Uv[2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) I.B[a][c] -= sw[a] * Uv[c];
    // c = v x S qd : only components ai, aj are non-zero
    float cwi = qd * vel[b].w[aj], cwj = -qd * vel[b].w[ai];
    float cvi = qd * vel[b].v[aj], cvj = -qd * vel[b].v[ai];
    float du = Dinv * u;
    SV pa;
    {
      const Sym3 &A = I.A, &C = I.C;
      float Af[3][3] = {{A.m[0], A.m[3], A.m[4]}, {A.m[3], A.m[1], A.m[5]}, {A.m[4], A.m[5], A.m[2]}};
      float Cf[3][3] = {{C.m[0], C.m[3], C.m[4]}, {C.m[3], C.m[1], C.m[5]}, {C.m[4], C.m[5], C.m[2]}};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        pa.w[r] = pA.w[r] + Af[r][ai] * cwi + Af[r][aj] * cwj + I.B[r][ai] * cvi + I.B[r][aj] * cvj + Uw[r] * du;
        pa.v[r] = pA.v[r] + I.B[ai][r] * cwi + I.B[aj][r] * cwj + Cf[r][ai] * cvi + Cf[r][aj] * cvj + Uv[r] * du;
      }
    }
    ABI Ip = xinertia<Model, j>(jc.cs[j], jc.sn[j], I);
    SV pp = xforce<Model, j>(jc.cs[j], jc.sn[j], pa);
    if constexpr (b == first_child(p)) {
      acc[p] = Ip;
      pacc[p] = pp;
    } else {
      abi_add(acc[p], Ip);
#pragma unroll
      for (int i = 0; i < 3; ++i) { pacc[p].w[i] += pp.w[i]; pacc[p].v[i] += pp.v[i]; }
    }
  });

  // ---- base
  SV a0;
  {
    ABI I0 = acc[0];
    abi_add_body<Model, 0>(I0);
    SV pb = body_bias<Model, 0>(vel[0]);
    SV p0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { p0.w[i] = pacc[0].w[i] + pb.w[i]; p0.v[i] = pacc[0].v[i] + pb.v[i]; }
    float M[6][6];
    abi_dense(I0, M);
    jc.L0 = chol6(M);
    a0 = chol6_solve_neg(jc.L0, p0);
  }

  // ---- pass 3: accelerations -> free velocities
  float qdf[NJ];
  {
    SV acl[NB];
    acl[0] = a0;
    static_for<0, NJ>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
      constexpr int ai = (ax + 1) % 3, aj = (ax + 2) % 3;
      SV a = xmotion<Model, j>(jc.cs[j], jc.sn[j], acl[p]);
      float qd = s.qd[j];
      a.w[ai] += qd * vel[b].w[aj]; a.w[aj] -= qd * vel[b].w[ai];
      a.v[ai] += qd * vel[b].v[aj]; a.v[aj] -= qd * vel[b].v[ai];
      float dotv = jc.Uw[j][0] * a.w[0] + jc.Uw[j][1] * a.w[1] + jc.Uw[j][2] * a.w[2] + jc.Uv[j][0] * a.v[0] +
                   jc.Uv[j][1] * a.v[1] + jc.Uv[j][2] * a.v[2];
      float qdd = jc.Dinv[j] * (uu[j] - dotv);
      a.w[ax] += qdd;
      acl[b] = a;
      qdf[j] = qd + h * qdd;
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
      float c = jc.cs[j], sn = jc.sn[j];
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
  float pen[8];
  float cn[8][3];          // contact normal (world)
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
      float bn[3] = {0.f, 0.f, 1.f};
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        float dx = P[0] - st.p[sl][0], dy = P[1] - st.p[sl][1], dz = P[2] - st.p[sl][2];
        float d = dx * st.n[sl][0] + dy * st.n[sl][1] + dz * st.n[sl][2];
        float lx = dx - d * st.n[sl][0], ly = dy - d * st.n[sl][1], lz = dz - d * st.n[sl][2];
        float rho2 = lx * lx + ly * ly + lz * lz;
        bool hit = (d < 0.f) && (d > -0.10f) && (rho2 < kStoneR2) && (d < best);
        if (hit) { best = d; slot = sl; bn[0] = st.n[sl][0]; bn[1] = st.n[sl][1]; bn[2] = st.n[sl][2]; }
      }
      pen[ck] = -best;
      cn[ck][0] = bn[0]; cn[ck][1] = bn[1]; cn[ck][2] = bn[2];
      if (slot >= 0) {
        active |= 1 << ck;
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
      impulse_response<Model, true, false, false>(jc, e, zero, VR, VL, nullptr, nullptr);
      LDS4(kLdsLinv + i * 3 + 0) = make_float4(VR.w[0], VR.w[1], VR.w[2], VR.v[0]);
      LDS4(kLdsLinv + i * 3 + 1) = make_float4(VR.v[1], VR.v[2], VL.w[0], VL.w[1]);
      LDS4(kLdsLinv + i * 3 + 2) = make_float4(VL.w[2], VL.v[0], VL.v[1], VL.v[2]);
    }
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
      SV e;
#pragma unroll
      for (int m = 0; m < 3; ++m) { e.w[m] = (i == m) ? 1.f : 0.f; e.v[m] = (i == m + 3) ? 1.f : 0.f; }
      SV VR, VL;
      impulse_response<Model, false, true, false>(jc, zero, e, VR, VL, nullptr, nullptr);
      LDS4(kLdsLinv + (6 + i) * 3 + 0) = make_float4(VR.w[0], VR.w[1], VR.w[2], VR.v[0]);
      LDS4(kLdsLinv + (6 + i) * 3 + 1) = make_float4(VR.v[1], VR.v[2], VL.w[0], VL.w[1]);
      LDS4(kLdsLinv + (6 + i) * 3 + 2) = make_float4(VL.w[2], VL.v[0], VL.v[1], VL.v[2]);
    }
    // foot twists under the free velocities
    float V[12];
    {
      SV vb[14];
      vb[0] = v0f;
      static_for<0, 13>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value, b = j + 1, p = kParent[j], ax = kAxis[j];
        vb[b] = xmotion<Model, j>(jc.cs[j], jc.sn[j], vb[p]);
        vb[b].w[ax] += qdf[j];
      });
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        V[i] = vb[RFOOT].w[i]; V[3 + i] = vb[RFOOT].v[i];
        V[6 + i] = vb[LFOOT].w[i]; V[9 + i] = vb[LFOOT].v[i];
      }
    }
    // rows -> LDS
    static_for<0, 8>([&](auto Kc) {
      constexpr int ck = decltype(Kc)::value, f = ck / 4, k = ck % 4;
      if (active & (1 << ck)) {
        constexpr float cx = Model::corners[k][0], cy = Model::corners[k][1], cz = Model::corners[k][2];
        float n[3] = {cn[ck][0], cn[ck][1], cn[ck][2]};
        float t1[3] = {1.f - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
        float inv = SS_RSQRT(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
        t1[0] *= inv; t1[1] *= inv; t1[2] *= inv;
        float t2[3];
        cross(n, t1, t2);
        float corr = fmaxf(pen[ck] - kSlop, 0.f);
        float bnv = fminf(kErp * corr * (1.0f / kH), kVcorrMax);
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
            float4 c0 = LDS4(kLdsLinv + (f * 6 + l) * 3 + 0);
            float4 c1 = LDS4(kLdsLinv + (f * 6 + l) * 3 + 1);
            float4 c2 = LDS4(kLdsLinv + (f * 6 + l) * 3 + 2);
            y[0] += c0.x * w[l]; y[1] += c0.y * w[l]; y[2] += c0.z * w[l]; y[3] += c0.w * w[l];
            y[4] += c1.x * w[l]; y[5] += c1.y * w[l]; y[6] += c1.z * w[l]; y[7] += c1.w * w[l];
            y[8] += c2.x * w[l]; y[9] += c2.y * w[l]; y[10] += c2.z * w[l]; y[11] += c2.w * w[l];
          }
          float A = 0.f;
#pragma unroll
          for (int l = 0; l < 6; ++l) A += w[l] * y[f * 6 + l];
          constexpr int row = kLdsRows + (ck * 3 + d) * 5;
          LDS4(row + 0) = make_float4(y[0], y[1], y[2], y[3]);
          LDS4(row + 1) = make_float4(y[4], y[5], y[6], y[7]);
          LDS4(row + 2) = make_float4(y[8], y[9], y[10], y[11]);
          LDS4(row + 3) = make_float4(w[0], w[1], w[2], w[3]);
          LDS4(row + 4) = make_float4(w[4], w[5], 1.0f / A, d == 0 ? bnv : 0.f);
        });
      }
    });
    // projected Gauss-Seidel
    float lam[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) lam[k][0] = lam[k][1] = lam[k][2] = 0.f;
    constexpr float mu = Model::friction;
#pragma unroll 1
    for (int it = 0; it < kPgsIters; ++it) {
      static_for<0, 8>([&](auto Kc) {
        constexpr int ck = decltype(Kc)::value, f = ck / 4;
        if (active & (1 << ck)) {
          static_for<0, 3>([&](auto Dc) {
            constexpr int d = decltype(Dc)::value;
            constexpr int row = kLdsRows + (ck * 3 + d) * 5;
            float4 y0 = LDS4(row + 0), y1 = LDS4(row + 1), y2 = LDS4(row + 2), w0 = LDS4(row + 3), w1 = LDS4(row + 4);
            float vrel = w0.x * V[f * 6 + 0] + w0.y * V[f * 6 + 1] + w0.z * V[f * 6 + 2] + w0.w * V[f * 6 + 3] +
                         w1.x * V[f * 6 + 4] + w1.y * V[f * 6 + 5];
            float ln = lam[ck][d] + (w1.w - vrel) * w1.z;
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
      constexpr int ck = decltype(Kc)::value, f = ck / 4;
      if (active & (1 << ck)) {
        static_for<0, 3>([&](auto Dc) {
          constexpr int d = decltype(Dc)::value;
          constexpr int row = kLdsRows + (ck * 3 + d) * 5;
          float4 w0 = LDS4(row + 3), w1 = LDS4(row + 4);
          float l = lam[ck][d];
          SV& W = f == 0 ? WR : WL;
          W.w[0] += w0.x * l; W.w[1] += w0.y * l; W.w[2] += w0.z * l;
          W.v[0] += w0.w * l; W.v[1] += w1.x * l; W.v[2] += w1.y * l;
        });
      }
    });
    SV VR, VL;
    impulse_response<Model, true, true, true>(jc, WR, WL, VR, VL, &dv0, dqd);
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
