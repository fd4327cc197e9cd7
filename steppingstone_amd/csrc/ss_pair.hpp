// ss_pair.hpp -- packed-f32 ("pair") versions of the spatial algebra of ss_math.hpp.
//
// The leg joints 3..6 (hip x, z, y, knee) and the arm joints 13..16 (shoulder x, z, y, elbow) of a half body have the
// same axes and the same massless/massive link pattern (tools/gen_model_tables.py asserts it), so the three ABA sweeps
// apply the SAME operator sequence to both chains with different constants.  Here every quantity is a float pair
// {leg, arm} and every operation one v_pk_*_f32 instruction: half the VALU instructions for those eight joints.
// Constants become literal pairs; a term is dropped only when it is zero for BOTH chains.
#pragma once
#include "ss_math.hpp"

namespace ss {

typedef float ssf2 __attribute__((ext_vector_type(2)));

struct SV2 { ssf2 w[3], v[3]; };

struct Sym3P {  // symmetric 3x3 of pairs: xx yy zz xy xz yz
  ssf2 m[6];
  template <int I, int J>
  SSD ssf2& at() {
    if constexpr (I == J) return m[I];
    else if constexpr (I + J == 1) return m[3];
    else if constexpr (I + J == 2) return m[4];
    else return m[5];
  }
  template <int I, int J>
  SSD ssf2 get() const {
    if constexpr (I == J) return m[I];
    else if constexpr (I + J == 1) return m[3];
    else if constexpr (I + J == 2) return m[4];
    else return m[5];
  }
};
struct ABIP {
  Sym3P A;
  ssf2 B[3][3];
  Sym3P C;
};

SSD ssf2 pk(float a, float b) { return ssf2{a, b}; }          // constants
// run-time scalars: made opaque first, otherwise instcombine turns "insert (load float from an SV still in memory)"
// into overlapping <2 x float> loads, which keeps that SV in scratch (seen as 88 B/lane and +25 % wait cycles)
#if defined(__HIP_DEVICE_COMPILE__)
#define SS_REG(x) asm("" : "+v"(x))
#else
#define SS_REG(x) asm("" : "+x"(x))
#endif
SSD ssf2 pkv(float a, float b) { SS_REG(a); SS_REG(b); return ssf2{a, b}; }
SSD SV sv_half(const SV2& a, int h) {
  SV o;
#pragma unroll
  for (int i = 0; i < 3; ++i) { o.w[i] = h ? a.w[i].y : a.w[i].x; o.v[i] = h ? a.v[i].y : a.v[i].x; }
  return o;
}
SSD SV2 sv_pack(const SV& l, const SV& a) {
  SV2 o;
#pragma unroll
  for (int i = 0; i < 3; ++i) { o.w[i] = pkv(l.w[i], a.w[i]); o.v[i] = pkv(l.v[i], a.v[i]); }
  return o;
}
SSD ABI abi_half(const ABIP& a, int h) {
  ABI o;
#pragma unroll
  for (int i = 0; i < 6; ++i) { o.A.m[i] = h ? a.A.m[i].y : a.A.m[i].x; o.C.m[i] = h ? a.C.m[i].y : a.C.m[i].x; }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o.B[i][j] = h ? a.B[i][j].y : a.B[i][j].x;
  return o;
}

SSD void crossP(const ssf2 a[3], const ssf2 b[3], ssf2 o[3]) {
  ssf2 x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
template <int AX>
SSD void rotP(ssf2 c, ssf2 s, const ssf2 v[3], ssf2 o[3]) {
  constexpr int i = (AX + 1) % 3, j = (AX + 2) % 3;
  ssf2 vi = v[i], vj = v[j];
  o[AX] = v[AX];
  o[i] = c * vi - s * vj;
  o[j] = s * vi + c * vj;
}
template <int AX>
SSD void rotTP(ssf2 c, ssf2 s, const ssf2 v[3], ssf2 o[3]) {
  constexpr int i = (AX + 1) % 3, j = (AX + 2) % 3;
  ssf2 vi = v[i], vj = v[j];
  o[AX] = v[AX];
  o[i] = c * vi + s * vj;
  o[j] = c * vj - s * vi;
}
template <class Model, int JL, int JA>
constexpr bool has_offsetP() { return has_offset<Model, JL>() || has_offset<Model, JA>(); }
// o = r x f with the constexpr offsets of joints JL (leg half) and JA (arm half)
template <class Model, int JL, int JA>
SSD void cross_rP(const ssf2 f[3], ssf2 o[3]) {
  constexpr float rxl = Model::r[JL][0], ryl = Model::r[JL][1], rzl = Model::r[JL][2];
  constexpr float rxa = Model::r[JA][0], rya = Model::r[JA][1], rza = Model::r[JA][2];
  ssf2 o0 = {0.f, 0.f}, o1 = {0.f, 0.f}, o2 = {0.f, 0.f};
  if constexpr (ryl != 0.f || rya != 0.f) { o0 += pk(ryl, rya) * f[2]; o2 -= pk(ryl, rya) * f[0]; }
  if constexpr (rzl != 0.f || rza != 0.f) { o0 -= pk(rzl, rza) * f[1]; o1 += pk(rzl, rza) * f[0]; }
  if constexpr (rxl != 0.f || rxa != 0.f) { o1 -= pk(rxl, rxa) * f[2]; o2 += pk(rxl, rxa) * f[1]; }
  o[0] = o0; o[1] = o1; o[2] = o2;
}
template <class Model, int JL, int JA>
SSD SV2 xmotionP(ssf2 c, ssf2 s, const SV2& p) {
  static_assert(kAxis[JL] == kAxis[JA], "paired joints must share their axis");
  constexpr int AX = kAxis[JL];
  SV2 o;
  rotTP<AX>(c, s, p.w, o.w);
  ssf2 t[3] = {p.v[0], p.v[1], p.v[2]};
  if constexpr (has_offsetP<Model, JL, JA>()) {
    ssf2 rxw[3];
    cross_rP<Model, JL, JA>(p.w, rxw);
    t[0] -= rxw[0]; t[1] -= rxw[1]; t[2] -= rxw[2];
  }
  rotTP<AX>(c, s, t, o.v);
  return o;
}
template <class Model, int JL, int JA>
SSD SV2 xforceP(ssf2 c, ssf2 s, const SV2& f) {
  constexpr int AX = kAxis[JL];
  SV2 o;
  rotP<AX>(c, s, f.v, o.v);
  rotP<AX>(c, s, f.w, o.w);
  if constexpr (has_offsetP<Model, JL, JA>()) {
    ssf2 t[3];
    cross_rP<Model, JL, JA>(o.v, t);
    o.w[0] += t[0]; o.w[1] += t[1]; o.w[2] += t[2];
  }
  return o;
}
template <int AX>
SSD Sym3P rot_symP(ssf2 c, ssf2 s, const Sym3P& S) {
  constexpr int i = (AX + 1) % 3, j = (AX + 2) % 3, k = AX;
  Sym3P o;
  ssf2 Sii = S.get<i, i>(), Sjj = S.get<j, j>(), Sij = S.get<i, j>(), Sik = S.get<i, k>(), Sjk = S.get<j, k>();
  ssf2 cc = c * c, ss_ = s * s, cs = c * s;
  o.at<k, k>() = S.get<k, k>();
  o.at<i, k>() = c * Sik - s * Sjk;
  o.at<j, k>() = s * Sik + c * Sjk;
  ssf2 t = (cs + cs) * Sij;
  o.at<i, i>() = cc * Sii - t + ss_ * Sjj;
  o.at<j, j>() = ss_ * Sii + t + cc * Sjj;
  o.at<i, j>() = cs * (Sii - Sjj) + (cc - ss_) * Sij;
  return o;
}
template <int AX>
SSD void rot_genP(ssf2 c, ssf2 s, const ssf2 M[3][3], ssf2 O[3][3]) {
  constexpr int i = (AX + 1) % 3, j = (AX + 2) % 3, k = AX;
  ssf2 T[3][3];
#pragma unroll
  for (int col = 0; col < 3; ++col) {
    T[i][col] = c * M[i][col] - s * M[j][col];
    T[j][col] = s * M[i][col] + c * M[j][col];
    T[k][col] = M[k][col];
  }
#pragma unroll
  for (int row = 0; row < 3; ++row) {
    O[row][i] = c * T[row][i] - s * T[row][j];
    O[row][j] = s * T[row][i] + c * T[row][j];
    O[row][k] = T[row][k];
  }
}
template <class Model, int JL, int JA>
SSD ABIP xinertiaP(ssf2 c, ssf2 s, const ABIP& I) {
  constexpr int AX = kAxis[JL];
  ABIP o;
  o.A = rot_symP<AX>(c, s, I.A);
  o.C = rot_symP<AX>(c, s, I.C);
  ssf2 Bp[3][3];
  rot_genP<AX>(c, s, I.B, Bp);
  if constexpr (has_offsetP<Model, JL, JA>()) {
    ssf2 rB[3][3];
#pragma unroll
    for (int row = 0; row < 3; ++row) cross_rP<Model, JL, JA>(Bp[row], rB[row]);
    const Sym3P& C = o.C;
    ssf2 Cc[3][3] = {{C.m[0], C.m[3], C.m[4]}, {C.m[3], C.m[1], C.m[5]}, {C.m[4], C.m[5], C.m[2]}};
#pragma unroll
    for (int col = 0; col < 3; ++col) {
      ssf2 t[3];
      cross_rP<Model, JL, JA>(Cc[col], t);
      Bp[0][col] += t[0]; Bp[1][col] += t[1]; Bp[2][col] += t[2];
    }
    ssf2 rBp[3][3];
#pragma unroll
    for (int row = 0; row < 3; ++row) cross_rP<Model, JL, JA>(Bp[row], rBp[row]);
    o.A.m[0] += rBp[0][0] + rB[0][0];
    o.A.m[1] += rBp[1][1] + rB[1][1];
    o.A.m[2] += rBp[2][2] + rB[2][2];
    o.A.m[3] += rBp[0][1] + rB[1][0];
    o.A.m[4] += rBp[0][2] + rB[2][0];
    o.A.m[5] += rBp[1][2] + rB[2][1];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) o.B[a][b] = Bp[a][b];
  return o;
}
template <class Model, int BL, int BA>
constexpr bool massiveP() { return Model::mass[BL] != 0.f || Model::mass[BA] != 0.f; }
// add the constexpr rigid-body inertias of bodies BL (leg half) and BA (arm half)
template <class Model, int BL, int BA>
SSD void abi_add_bodyP(ABIP& I) {
  constexpr float ml = Model::mass[BL], ma = Model::mass[BA];
  if constexpr (ml != 0.f || ma != 0.f) {
    constexpr float cxl = Model::com[BL][0], cyl = Model::com[BL][1], czl = Model::com[BL][2];
    constexpr float cxa = Model::com[BA][0], cya = Model::com[BA][1], cza = Model::com[BA][2];
    static_for<0, 6>([&](auto Ic) {
      constexpr int i = decltype(Ic)::value;
      constexpr float kl = Model::inertia[BL][i], ka = Model::inertia[BA][i];
      if constexpr (kl != 0.f || ka != 0.f) I.A.m[i] += pk(kl, ka);
    });
    I.C.m[0] += pk(ml, ma); I.C.m[1] += pk(ml, ma); I.C.m[2] += pk(ml, ma);
    if constexpr (czl != 0.f || cza != 0.f) { I.B[0][1] += pk(-ml * czl, -ma * cza); I.B[1][0] += pk(ml * czl, ma * cza); }
    if constexpr (cyl != 0.f || cya != 0.f) { I.B[0][2] += pk(ml * cyl, ma * cya); I.B[2][0] += pk(-ml * cyl, -ma * cya); }
    if constexpr (cxl != 0.f || cxa != 0.f) { I.B[1][2] += pk(-ml * cxl, -ma * cxa); I.B[2][1] += pk(ml * cxl, ma * cxa); }
  }
}
SSD ABIP abi_zeroP() {
  ABIP I;
  const ssf2 z = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 6; ++i) { I.A.m[i] = z; I.C.m[i] = z; }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) I.B[i][j] = z;
  return I;
}
// velocity-product bias forces of bodies BL / BA:  p = v x* (I_b v)
template <class Model, int BL, int BA>
SSD SV2 body_biasP(const SV2& v) {
  constexpr float ml = Model::mass[BL], ma = Model::mass[BA];
  constexpr float hxl = ml * Model::com[BL][0], hyl = ml * Model::com[BL][1], hzl = ml * Model::com[BL][2];
  constexpr float hxa = ma * Model::com[BA][0], hya = ma * Model::com[BA][1], hza = ma * Model::com[BA][2];
  const ssf2 z = {0.f, 0.f};
  ssf2 n[3] = {z, z, z}, f[3];
  static_for<0, 3>([&](auto Rc) {
    constexpr int r = decltype(Rc)::value;
    static_for<0, 3>([&](auto Cc) {
      constexpr int c = decltype(Cc)::value;
      constexpr int idx = r == c ? r : (r + c == 1 ? 3 : (r + c == 2 ? 4 : 5));
      constexpr float kl = Model::inertia[BL][idx], ka = Model::inertia[BA][idx];
      if constexpr (kl != 0.f || ka != 0.f) n[r] += pk(kl, ka) * v.w[c];
    });
  });
  if constexpr (hyl != 0.f || hya != 0.f) { n[0] += pk(hyl, hya) * v.v[2]; n[2] -= pk(hyl, hya) * v.v[0]; }
  if constexpr (hzl != 0.f || hza != 0.f) { n[0] -= pk(hzl, hza) * v.v[1]; n[1] += pk(hzl, hza) * v.v[0]; }
  if constexpr (hxl != 0.f || hxa != 0.f) { n[1] -= pk(hxl, hxa) * v.v[2]; n[2] += pk(hxl, hxa) * v.v[1]; }
  f[0] = pk(ml, ma) * v.v[0]; f[1] = pk(ml, ma) * v.v[1]; f[2] = pk(ml, ma) * v.v[2];
  if constexpr (hyl != 0.f || hya != 0.f) { f[0] -= pk(hyl, hya) * v.w[2]; f[2] += pk(hyl, hya) * v.w[0]; }
  if constexpr (hzl != 0.f || hza != 0.f) { f[0] += pk(hzl, hza) * v.w[1]; f[1] -= pk(hzl, hza) * v.w[0]; }
  if constexpr (hxl != 0.f || hxa != 0.f) { f[1] += pk(hxl, hxa) * v.w[2]; f[2] -= pk(hxl, hxa) * v.w[1]; }
  SV2 p;
  ssf2 a[3], b[3];
  crossP(v.w, n, a);
  crossP(v.v, f, b);
  p.w[0] = a[0] + b[0]; p.w[1] = a[1] + b[1]; p.w[2] = a[2] + b[2];
  crossP(v.w, f, p.v);
  return p;
}

}  // namespace ss
