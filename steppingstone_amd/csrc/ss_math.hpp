// ss_math.hpp -- per-lane spatial algebra for the gfx950 stepping-stone kernels.
//
// One environment per lane.  Everything here is written for full compile-time specialisation: joint index,
// rotation axis, link offsets and body inertias are template/constexpr values, so the 21-link tree unrolls into
// straight-line VALU code with the per-link quantities held in VGPR/AGPR (512 per lane at one wave per SIMD).
// Conventions follow docs/PHYSICS.md section 1 (Featherstone spatial vectors, angular part first).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ss_model_tables.hpp"

// __host__ as well: tests/host_harness.cpp compiles the very same source for the CPU (debug / pre-flight only;
// the shipped library contains no host path).
#define SSD __host__ __device__ __forceinline__

#if defined(__HIP_DEVICE_COMPILE__)
// optional scheduling fence between links of the unrolled tree sweeps (-DSS_SCHED_FENCE): measured round 1, it
// lowers spills slightly (1325 -> 1193) but costs 4 % of step time, so it is off by default
#if defined(SS_SCHED_FENCE)
#define SS_FENCE() __builtin_amdgcn_sched_barrier(0)
#elif defined(SS_MEM_FENCE)
#define SS_FENCE() asm volatile("" ::: "memory")
#else
#define SS_FENCE() ((void)0)
#endif
// hides a value from CSE so that address arithmetic is redone after a long region instead of being kept live
#define SS_OPAQUE(x) asm volatile("" : "+v"(x))
#define SS_RSQRT(x) rsqrtf(x)
// reciprocal: v_rcp_f32 (1 ulp) + one Newton step = 3 VALU instructions where the IEEE division expands to ~10
// (24 of them per substep: joint 1/D and contact-row 1/A).  -DSS_IEEE_DIV restores the division.
#ifdef SS_IEEE_DIV
#define SS_RCP(x) (1.0f / (x))
#else
static __device__ __forceinline__ float ss_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
#define SS_RCP(x) ss_rcp(x)
#endif
#define SS_UMULHI(a, b) __umulhi((a), (b))
#define SS_F2U(x) __float_as_uint(x)
#else
#include <cmath>
#include <cstring>
float ss_host_xchg(float x);   // lane-pair exchange, provided by tests/host/host_harness.cpp
void ss_host_wave_sync();      // barrier over the 64 lane threads of a wavefront (ditto)
static inline float ss_host_rsqrt(float x) { return 1.0f / sqrtf(x); }
static inline unsigned ss_host_umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned ss_host_f2u(float x) { unsigned u; std::memcpy(&u, &x, 4); return u; }
#define SS_FENCE() ((void)0)
#define SS_OPAQUE(x) asm volatile("" : "+r"(x))
#define SS_RSQRT(x) ss_host_rsqrt(x)
#define SS_RCP(x) (1.0f / (x))
#define SS_UMULHI(a, b) ss_host_umulhi((a), (b))
#define SS_F2U(x) ss_host_f2u(x)
#endif
// accumulate k*x only when the constexpr coefficient k is non-zero (x*0 is not foldable under IEEE rules)
#define SS_ACC(o, k, x)                       \
  do {                                        \
    if constexpr ((k) != 0.0f) (o) += (k) * (x); \
  } while (0)

namespace ss {

constexpr int NJ = 21;
constexpr int NB = 22;
constexpr int RFOOT = 8;
constexpr int LFOOT = 13;

template <int I, int N, class F>
SSD void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// I = N-1 ... LO
template <int I, int LO, class F>
SSD void static_rfor(F&& f) {
  if constexpr (I >= LO) {
    f(std::integral_constant<int, I>{});
    static_rfor<I - 1, LO>(f);
  }
}

// highest-index child of body b (processed first in the leaves->root sweep); -1 for leaves
constexpr int first_child(int b) {
  int r = -1;
  for (int j = 0; j < NJ; ++j)
    if (kParent[j] == b) r = j + 1;
  return r;
}
constexpr bool on_leg_path(int b) {  // bodies whose joints lie between a foot and the root
  return (b >= 1 && b <= 13);
}

struct SV {  // spatial motion or force vector
  float w[3];
  float v[3];
};

struct Sym3 {  // symmetric 3x3: xx yy zz xy xz yz
  float m[6];
  template <int I, int J>
  SSD float& at() {
    if constexpr (I == J) return m[I];
    else if constexpr (I + J == 1) return m[3];
    else if constexpr (I + J == 2) return m[4];
    else return m[5];
  }
  template <int I, int J>
  SSD float get() const {
    if constexpr (I == J) return m[I];
    else if constexpr (I + J == 1) return m[3];
    else if constexpr (I + J == 2) return m[4];
    else return m[5];
  }
};

// articulated-body inertia [[A, B], [B^T, C]] : n = A w + B v, f = B^T w + C v
struct ABI {
  Sym3 A;
  float B[3][3];
  Sym3 C;
};

SSD void cross(const float a[3], const float b[3], float o[3]) {
  float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
SSD float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
SSD float dot6(const SV& a, const SV& b) { return dot3(a.w, b.w) + dot3(a.v, b.v); }

// active rotation about coordinate axis AX by the angle with cosine c / sine s:  o = R v
template <int AX>
SSD void rot(float c, float s, const float v[3], float o[3]) {
  constexpr int i = (AX + 1) % 3, j = (AX + 2) % 3;
  float vi = v[i], vj = v[j];
  o[AX] = v[AX];
  o[i] = c * vi - s * vj;
  o[j] = s * vi + c * vj;
}
// o = R^T v
template <int AX>
SSD void rotT(float c, float s, const float v[3], float o[3]) {
  constexpr int i = (AX + 1) % 3, j = (AX + 2) % 3;
  float vi = v[i], vj = v[j];
  o[AX] = v[AX];
  o[i] = c * vi + s * vj;
  o[j] = c * vj - s * vi;
}

// o = r x f with constexpr r = Model::r[J]
template <class Model, int J>
SSD void cross_r(const float f[3], float o[3]) {
  constexpr float rx = Model::r[J][0], ry = Model::r[J][1], rz = Model::r[J][2];
  float o0 = 0.f, o1 = 0.f, o2 = 0.f;
  SS_ACC(o0, ry, f[2]); SS_ACC(o0, -rz, f[1]);
  SS_ACC(o1, rz, f[0]); SS_ACC(o1, -rx, f[2]);
  SS_ACC(o2, rx, f[1]); SS_ACC(o2, -ry, f[0]);
  o[0] = o0; o[1] = o1; o[2] = o2;
}
template <class Model, int J>
constexpr bool has_offset() {
  return Model::r[J][0] != 0.f || Model::r[J][1] != 0.f || Model::r[J][2] != 0.f;
}

// motion vector parent frame -> child frame of joint J:  w_c = R^T w_p,  v_c = R^T (v_p + w_p x r)
template <class Model, int J>
SSD SV xmotion(float c, float s, const SV& p) {
  constexpr int AX = kAxis[J];
  SV o;
  rotT<AX>(c, s, p.w, o.w);
  float t[3] = {p.v[0], p.v[1], p.v[2]};
  if constexpr (has_offset<Model, J>()) {
    float rxw[3];
    cross_r<Model, J>(p.w, rxw);  // r x w = -(w x r)
    t[0] -= rxw[0]; t[1] -= rxw[1]; t[2] -= rxw[2];
  }
  rotT<AX>(c, s, t, o.v);
  return o;
}
// force vector child frame -> parent frame:  f_p = R f_c,  n_p = R n_c + r x f_p
template <class Model, int J>
SSD SV xforce(float c, float s, const SV& f) {
  constexpr int AX = kAxis[J];
  SV o;
  rot<AX>(c, s, f.v, o.v);
  rot<AX>(c, s, f.w, o.w);
  if constexpr (has_offset<Model, J>()) {
    float t[3];
    cross_r<Model, J>(o.v, t);
    o.w[0] += t[0]; o.w[1] += t[1]; o.w[2] += t[2];
  }
  return o;
}

// S' = R S R^T for a symmetric block, rotation in the (i,j) plane
template <int AX>
SSD Sym3 rot_sym(float c, float s, const Sym3& S) {
  constexpr int i = (AX + 1) % 3, j = (AX + 2) % 3, k = AX;
  Sym3 o;
  float Sii = S.get<i, i>(), Sjj = S.get<j, j>(), Sij = S.get<i, j>(), Sik = S.get<i, k>(), Sjk = S.get<j, k>();
  float cc = c * c, ss_ = s * s, cs = c * s;
  o.at<k, k>() = S.get<k, k>();
  o.at<i, k>() = c * Sik - s * Sjk;
  o.at<j, k>() = s * Sik + c * Sjk;
  float t = 2.f * cs * Sij;
  o.at<i, i>() = cc * Sii - t + ss_ * Sjj;
  o.at<j, j>() = ss_ * Sii + t + cc * Sjj;
  o.at<i, j>() = cs * (Sii - Sjj) + (cc - ss_) * Sij;
  return o;
}
// M' = R M R^T for a general 3x3
template <int AX>
SSD void rot_gen(float c, float s, const float M[3][3], float O[3][3]) {
  constexpr int i = (AX + 1) % 3, j = (AX + 2) % 3, k = AX;
  float T[3][3];
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

// articulated inertia of the child (in child coords) -> parent coords:  X^T I X
//   rotate every block into the parent orientation, then shift the origin by r:
//   C_p = C', B_p = B' + r x C' (column-wise), A_p[i][j] = A'[i][j] + (r x Bp_row_i)[j] + (r x B'_row_j)[i]
template <class Model, int J>
SSD ABI xinertia(float c, float s, const ABI& I) {
  constexpr int AX = kAxis[J];
  ABI o;
  o.A = rot_sym<AX>(c, s, I.A);
  o.C = rot_sym<AX>(c, s, I.C);
  float Bp[3][3];
  rot_gen<AX>(c, s, I.B, Bp);
  if constexpr (has_offset<Model, J>()) {
    // r x B'_row_j for the three rows of B' (before the shift)
    float rB[3][3];
#pragma unroll
    for (int row = 0; row < 3; ++row) cross_r<Model, J>(Bp[row], rB[row]);
    // B_p = B' + r x C' column-wise (C' symmetric: column col = row col)
    const Sym3& C = o.C;
    float Cc[3][3] = {{C.m[0], C.m[3], C.m[4]}, {C.m[3], C.m[1], C.m[5]}, {C.m[4], C.m[5], C.m[2]}};
#pragma unroll
    for (int col = 0; col < 3; ++col) {
      float t[3];
      cross_r<Model, J>(Cc[col], t);
      Bp[0][col] += t[0]; Bp[1][col] += t[1]; Bp[2][col] += t[2];
    }
    float rBp[3][3];
#pragma unroll
    for (int row = 0; row < 3; ++row) cross_r<Model, J>(Bp[row], rBp[row]);
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

SSD void abi_add(ABI& a, const ABI& b) {
#pragma unroll
  for (int i = 0; i < 6; ++i) { a.A.m[i] += b.A.m[i]; a.C.m[i] += b.C.m[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) a.B[i][j] += b.B[i][j];
}

// add the constexpr rigid-body inertia of body Bd:  A += I_O, B += m [c]x, C += m 1
template <class Model, int Bd>
SSD void abi_add_body(ABI& I) {
  constexpr float m = Model::mass[Bd];
  if constexpr (m != 0.f) {
    constexpr float cx = Model::com[Bd][0], cy = Model::com[Bd][1], cz = Model::com[Bd][2];
    static_for<0, 6>([&](auto Ic) {
      constexpr int i = decltype(Ic)::value;
      constexpr float k = Model::inertia[Bd][i];
      if constexpr (k != 0.f) I.A.m[i] += k;
    });
    I.C.m[0] += m; I.C.m[1] += m; I.C.m[2] += m;
    // m [c]x = m * [[0,-cz,cy],[cz,0,-cx],[-cy,cx,0]]
    if constexpr (cz != 0.f) { I.B[0][1] += -m * cz; I.B[1][0] += m * cz; }
    if constexpr (cy != 0.f) { I.B[0][2] += m * cy; I.B[2][0] += -m * cy; }
    if constexpr (cx != 0.f) { I.B[1][2] += -m * cx; I.B[2][1] += m * cx; }
  }
}
template <class Model, int Bd>
SSD ABI abi_body() {
  ABI I;
#pragma unroll
  for (int i = 0; i < 6; ++i) { I.A.m[i] = 0.f; I.C.m[i] = 0.f; }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) I.B[i][j] = 0.f;
  abi_add_body<Model, Bd>(I);
  return I;
}

// velocity-product bias force of a rigid body:  p = v x* (I_b v), constexpr inertia
template <class Model, int Bd>
SSD SV body_bias(const SV& v) {
  constexpr float m = Model::mass[Bd];
  constexpr float cx = Model::com[Bd][0], cy = Model::com[Bd][1], cz = Model::com[Bd][2];
  constexpr float ixx = Model::inertia[Bd][0], iyy = Model::inertia[Bd][1], izz = Model::inertia[Bd][2];
  constexpr float ixy = Model::inertia[Bd][3], ixz = Model::inertia[Bd][4], iyz = Model::inertia[Bd][5];
  // h = m c (first moment)
  constexpr float hx = m * cx, hy = m * cy, hz = m * cz;
  // n = I_O w + h x v ;  f = m v - h x w
  float n[3] = {0.f, 0.f, 0.f}, f[3];
  SS_ACC(n[0], ixx, v.w[0]); SS_ACC(n[0], ixy, v.w[1]); SS_ACC(n[0], ixz, v.w[2]);
  SS_ACC(n[1], ixy, v.w[0]); SS_ACC(n[1], iyy, v.w[1]); SS_ACC(n[1], iyz, v.w[2]);
  SS_ACC(n[2], ixz, v.w[0]); SS_ACC(n[2], iyz, v.w[1]); SS_ACC(n[2], izz, v.w[2]);
  SS_ACC(n[0], hy, v.v[2]); SS_ACC(n[0], -hz, v.v[1]);
  SS_ACC(n[1], hz, v.v[0]); SS_ACC(n[1], -hx, v.v[2]);
  SS_ACC(n[2], hx, v.v[1]); SS_ACC(n[2], -hy, v.v[0]);
  f[0] = m * v.v[0]; f[1] = m * v.v[1]; f[2] = m * v.v[2];
  SS_ACC(f[0], -hy, v.w[2]); SS_ACC(f[0], hz, v.w[1]);
  SS_ACC(f[1], -hz, v.w[0]); SS_ACC(f[1], hx, v.w[2]);
  SS_ACC(f[2], -hx, v.w[1]); SS_ACC(f[2], hy, v.w[0]);
  // v x* [n; f] = [w x n + v x f ; w x f]
  SV p;
  float a[3], b[3];
  cross(v.w, n, a);
  cross(v.v, f, b);
  p.w[0] = a[0] + b[0]; p.w[1] = a[1] + b[1]; p.w[2] = a[2] + b[2];
  cross(v.w, f, p.v);
  return p;
}

// dense symmetric 6x6 from the block form, rows/cols ordered (w, v)
SSD void abi_dense(const ABI& I, float M[6][6]) {
  const Sym3 &A = I.A, &C = I.C;
  float Af[3][3] = {{A.m[0], A.m[3], A.m[4]}, {A.m[3], A.m[1], A.m[5]}, {A.m[4], A.m[5], A.m[2]}};
  float Cf[3][3] = {{C.m[0], C.m[3], C.m[4]}, {C.m[3], C.m[1], C.m[5]}, {C.m[4], C.m[5], C.m[2]}};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      M[i][j] = Af[i][j];
      M[i][j + 3] = I.B[i][j];
      M[i + 3][j] = I.B[j][i];
      M[i + 3][j + 3] = Cf[i][j];
    }
}

// Cholesky factor of a 6x6 SPD matrix, lower triangle packed; diagonal stored as reciprocals
struct Chol6 {
  float l[15];   // strictly-lower entries, row-major: (1,0) (2,0) (2,1) (3,0) ...
  float di[6];   // 1 / L_ii
  template <int I, int J>
  SSD float& at() { return l[I * (I - 1) / 2 + J]; }
  template <int I, int J>
  SSD float get() const { return l[I * (I - 1) / 2 + J]; }
};

SSD Chol6 chol6(const float M[6][6]) {
  Chol6 L;
  static_for<0, 6>([&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    static_for<0, i + 1>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      float s = M[i][j];
      static_for<0, j>([&](auto Kc) {
        constexpr int k = decltype(Kc)::value;
        if constexpr (i == j) s -= L.template get<i, k>() * L.template get<i, k>();
        else s -= L.template get<i, k>() * L.template get<j, k>();
      });
      if constexpr (i == j) L.di[i] = SS_RSQRT(s);
      else L.template at<i, j>() = s * L.di[j];
    });
  });
  return L;
}
// x = -(L L^T)^-1 b   (the sign is what every caller needs)
SSD SV chol6_solve_neg(const Chol6& L, const SV& b) {
  float y[6] = {-b.w[0], -b.w[1], -b.w[2], -b.v[0], -b.v[1], -b.v[2]};
  static_for<0, 6>([&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    float s = y[i];
    static_for<0, i>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      s -= L.template get<i, k>() * y[k];
    });
    y[i] = s * L.di[i];
  });
  static_rfor<5, 0>([&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    float s = y[i];
    static_for<i + 1, 6>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      s -= L.template get<k, i>() * y[k];
    });
    y[i] = s * L.di[i];
  });
  SV x;
  x.w[0] = y[0]; x.w[1] = y[1]; x.w[2] = y[2]; x.v[0] = y[3]; x.v[1] = y[4]; x.v[2] = y[5];
  return x;
}

// quaternion (w,x,y,z) -> rotation matrix (body -> world)
SSD void quat_rot(const float q[4], float R[3][3]) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - w * z); R[0][2] = 2.f * (x * z + w * y);
  R[1][0] = 2.f * (x * y + w * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - w * x);
  R[2][0] = 2.f * (x * z - w * y); R[2][1] = 2.f * (y * z + w * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

SSD bool finite_bits(float x) { return (SS_F2U(x) & 0x7f800000u) != 0x7f800000u; }

// Philox4x32-10 (PHYSICS.md section 6)
SSD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = SS_UMULHI(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = SS_UMULHI(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
SSD float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

}  // namespace ss
