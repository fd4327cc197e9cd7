// aba4_probe.hip -- does spreading the 6x6 algebra of ONE pass-2 joint step of the ABA over a DPP lane pair pay?  (VERDICT r3 item 3;
// DESIGN.md 9 / docs/HISTORY.md 7 "four lanes per env": until now estimated by instruction count only.)
//
// The env kernels run two lanes per env (one per half body); "four lanes per env" means two lanes per half-body chain.  Lanes of a
// wavefront execute ONE instruction stream, so the two lanes of a chain can only share a joint step if they run the SAME operation
// sequence on different operands.  The split measured here is the natural one with that property: the articulated inertia
// M = [[A, B], [B^T, C]] by block ROWS -- lane P holds the moment rows [A | B] and the moment half of the bias force, lane Q the force
// rows [B^T | C] and the force half -- both as plain 3x6 row blocks:
//     U = M S            : column `ax` of the moment part: 3 values per lane, swapped by DPP          (each lane then has all 6)
//     M -= U U^T / D     : 18 FMAs on the own rows                                                   (21 for the symmetric whole)
//     p^a = p + M^a c + U u / D : own 3 rows                                                         (15 + 3 instead of 30 + 6)
//     Y = M^a X          : own rows times the motion transform (rotation of the column pairs, origin shift): row-local
//     M' = X^T Y         : rotate the rows (own 3), then the moment rows take  r x (force rows): Q's rotated rows travel to P by DPP
//     p' = X^T p^a       : same pattern on the vector
// against the kernels' own single-lane joint step (symmetric storage: 21 entries; ss_math.hpp xinertia / xforce; the body of
// ss_dynamics.hpp: joint_scalar).  Both variants chain `iters` dependent joint steps (the output inertia, halved, plus a constant body
// inertia is the next input) on one wavefront per SIMD (40 KiB of LDS per 64-thread workgroup, 1024 workgroups) and report shader
// clocks per joint step per wavefront (s_memtime); `empty` is the loop without the joint step.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-signed-zeros -ffp-contract=on -I steppingstone_amd/csrc \
//         tools/probes/aba4_probe.hip -o var/aba4_probe && var/aba4_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "ss_math.hpp"

using namespace ss;
using Model = ModelWalker3D;
constexpr int J = 6;                       // the knee: axis y, offset (0, 0, -0.383): the commonest shape of a limb joint

__device__ __forceinline__ float dpp_swap(float x) {          // lane 2k <-> 2k+1
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}

// ---------------------------------------------------------------- single lane: the kernels' joint step (joint_scalar)
__device__ __forceinline__ void joint_single(ABI I, const SV& pA, const SV& vb, float qd, float tau, float Dadd, float cs, float sn,
                                             ABI& Ip, SV& pp, float& Dinv_out) {
  constexpr int ax = kAxis[J], ai = (ax + 1) % 3, aj = (ax + 2) % 3;
  float Uw[3] = {I.A.template get<0, ax>(), I.A.template get<1, ax>(), I.A.template get<2, ax>()};
  float Uv[3] = {I.B[ax][0], I.B[ax][1], I.B[ax][2]};
  const float Dinv = SS_RCP(Uw[ax] + Dadd), u = tau - pA.w[ax];
  float sw[3] = {Dinv * Uw[0], Dinv * Uw[1], Dinv * Uw[2]}, sv[3] = {Dinv * Uv[0], Dinv * Uv[1], Dinv * Uv[2]};
  I.A.m[0] -= sw[0] * Uw[0]; I.A.m[1] -= sw[1] * Uw[1]; I.A.m[2] -= sw[2] * Uw[2];
  I.A.m[3] -= sw[0] * Uw[1]; I.A.m[4] -= sw[0] * Uw[2]; I.A.m[5] -= sw[1] * Uw[2];
  I.C.m[0] -= sv[0] * Uv[0]; I.C.m[1] -= sv[1] * Uv[1]; I.C.m[2] -= sv[2] * Uv[2];
  I.C.m[3] -= sv[0] * Uv[1]; I.C.m[4] -= sv[0] * Uv[2]; I.C.m[5] -= sv[1] * Uv[2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) I.B[a][c] -= sw[a] * Uv[c];
  const float cwi = qd * vb.w[aj], cwj = -qd * vb.w[ai], cvi = qd * vb.v[aj], cvj = -qd * vb.v[ai], du = Dinv * u;
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
  Ip = xinertia<Model, J>(cs, sn, I);
  pp = xforce<Model, J>(cs, sn, pa);
  Dinv_out = Dinv;
}

// ---------------------------------------------------------------- lane pair: block rows
// Row block of this lane: M[3][6] = lane P: [A | B], lane Q: [B^T | C]; vector half h[3] = P: moment part, Q: force part.
struct Rows { float m[3][6]; };
__device__ __forceinline__ void joint_rows(Rows M, const float (&pA)[3], const float (&cw)[2], const float (&cv)[2], float tau, float Dadd,
                                           float cs, float sn, bool lane_q, Rows& Mp, float (&pp)[3], float& Dinv_out) {
  constexpr int ax = kAxis[J], ai = (ax + 1) % 3, aj = (ax + 2) % 3;
  constexpr float rx = Model::r[J][0], ry = Model::r[J][1], rz = Model::r[J][2];
  // U = M S: own three entries are column ax of the moment part; the other three come from the partner
  float Uown[3] = {M.m[0][ax], M.m[1][ax], M.m[2][ax]};
  float Uoth[3] = {dpp_swap(Uown[0]), dpp_swap(Uown[1]), dpp_swap(Uown[2])};
  float Uw[3], Uv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { Uw[k] = lane_q ? Uoth[k] : Uown[k]; Uv[k] = lane_q ? Uown[k] : Uoth[k]; }
  const float pax_other = dpp_swap(pA[ax]);                      // (outside the select: a DPP move under a divergent branch reads 0)
  const float pax = lane_q ? pax_other : pA[ax];                 // the moment half holds p.w[ax]
  const float Dinv = SS_RCP(Uw[ax] + Dadd), u = tau - pax, du = Dinv * u;
  float s[3] = {Dinv * Uown[0], Dinv * Uown[1], Dinv * Uown[2]};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) { M.m[r][c] -= s[r] * Uw[c]; M.m[r][3 + c] -= s[r] * Uv[c]; }
  float pa[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    pa[r] = pA[r] + M.m[r][ai] * cw[0] + M.m[r][aj] * cw[1] + M.m[r][3 + ai] * cv[0] + M.m[r][3 + aj] * cv[1] + Uown[r] * du;
  // Y = M X (row-local): motion transform child -> parent on the columns: rotate the (ai, aj) column pairs of both halves, and the
  // moment columns take  -(force columns) x r  (the transpose of  n_p = ... + r x f_p)
  float Y[3][6];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a = M.m[r][3 * h + ai], b = M.m[r][3 * h + aj];
      Y[r][3 * h + ax] = M.m[r][3 * h + ax];
      Y[r][3 * h + ai] = cs * a - sn * b;
      Y[r][3 * h + aj] = sn * a + cs * b;
    }
    // columns of the moment part += (r x)^T applied to the force columns:  y_w += y_v x r  ... = -(r x y_v)
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    SS_ACC(t0, ry, Y[r][5]); SS_ACC(t0, -rz, Y[r][4]);
    SS_ACC(t1, rz, Y[r][3]); SS_ACC(t1, -rx, Y[r][5]);
    SS_ACC(t2, rx, Y[r][4]); SS_ACC(t2, -ry, Y[r][3]);
    Y[r][0] += t0; Y[r][1] += t1; Y[r][2] += t2;
  }
  // M' = X^T Y: rotate the own rows ...
  float Z[3][6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    Z[ax][c] = Y[ax][c];
    Z[ai][c] = cs * Y[ai][c] - sn * Y[aj][c];
    Z[aj][c] = sn * Y[ai][c] + cs * Y[aj][c];
  }
  float zp[3];
  zp[ax] = pa[ax]; zp[ai] = cs * pa[ai] - sn * pa[aj]; zp[aj] = sn * pa[ai] + cs * pa[aj];
  // ... and the moment rows take  r x (force rows): the partner's rotated rows and vector arrive by DPP.  Only lane P adds them: its
  // copy of r is the joint's offset, lane Q's is zero (run-time per-lane constants: no select in the chain)
  const float kx = lane_q ? 0.f : rx, ky = lane_q ? 0.f : ry, kz = lane_q ? 0.f : rz;
  auto add_cross = [&](float f0, float f1, float f2, float& o0, float& o1, float& o2) {
    if constexpr (ry != 0.f) o0 += ky * f2;
    if constexpr (rz != 0.f) o0 -= kz * f1;
    if constexpr (rz != 0.f) o1 += kz * f0;
    if constexpr (rx != 0.f) o1 -= kx * f2;
    if constexpr (rx != 0.f) o2 += kx * f1;
    if constexpr (ry != 0.f) o2 -= ky * f0;
  };
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    Mp.m[0][c] = Z[0][c]; Mp.m[1][c] = Z[1][c]; Mp.m[2][c] = Z[2][c];
    add_cross(dpp_swap(Z[0][c]), dpp_swap(Z[1][c]), dpp_swap(Z[2][c]), Mp.m[0][c], Mp.m[1][c], Mp.m[2][c]);
  }
  pp[0] = zp[0]; pp[1] = zp[1]; pp[2] = zp[2];
  add_cross(dpp_swap(zp[0]), dpp_swap(zp[1]), dpp_swap(zp[2]), pp[0], pp[1], pp[2]);
  Dinv_out = Dinv;
}

// ---------------------------------------------------------------- kernels
template <int MODE>    // 0 empty loop, 1 single lane, 2 lane pair
__global__ __launch_bounds__(64, 1) void probe(int iters, float* sink, unsigned long long* clocks, float* check) {
  __shared__ float4 occupy[40 * 64];        // 40 KiB: one workgroup per SIMD, like the env kernels
  if (iters < 0) occupy[threadIdx.x] = make_float4(0, 0, 0, 0);
  const int lane = threadIdx.x;
  const bool lane_q = lane & 1;
  const float seed = 1.0f + 1e-3f * (float)((blockIdx.x * 64 + lane) >> 1 & 7);     // a few distinct values; equal within a lane pair
  // a plausible start: the foot's rigid-body inertia, a small bias force, a link velocity
  ABI I = abi_body<Model, 8>();
  SV pA = {{0.1f * seed, -0.2f, 0.05f}, {1.0f, 0.3f * seed, 9.0f}}, vb = {{0.3f, -0.7f * seed, 0.2f}, {0.5f, 0.1f, -0.4f}};
  const float qd = 0.8f * seed, tau = 20.f, Dadd = 0.01f;
  float cs = 0.9f, sn = 0.43588989f;
  SS_OPAQUE(cs); SS_OPAQUE(sn);
  Rows R;
  float ph[3];
  if (MODE == 2) {
    float M6[6][6];
    abi_dense(I, M6);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) R.m[r][c] = lane_q ? M6[3 + r][c] : M6[r][c];
#pragma unroll
    for (int r = 0; r < 3; ++r) ph[r] = lane_q ? pA.v[r] : pA.w[r];
  }
  constexpr int ax = kAxis[J], ai = (ax + 1) % 3, aj = (ax + 2) % 3;
  const float cw[2] = {qd * vb.w[aj], -qd * vb.w[ai]}, cv[2] = {qd * vb.v[aj], -qd * vb.v[ai]};
  float acc = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    float Dinv = 0.f;
    if (MODE == 1) {
      ABI Ip;
      SV pp;
      joint_single(I, pA, vb, qd, tau, Dadd, cs, sn, Ip, pp, Dinv);
#pragma unroll
      for (int k = 0; k < 6; ++k) { I.A.m[k] = 0.5f * Ip.A.m[k]; I.C.m[k] = 0.5f * Ip.C.m[k]; }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) I.B[a][c] = 0.5f * Ip.B[a][c];
      abi_add_body<Model, 7>(I);
#pragma unroll
      for (int k = 0; k < 3; ++k) { pA.w[k] = 0.5f * pp.w[k]; pA.v[k] = 0.5f * pp.v[k]; }
    } else if (MODE == 2) {
      Rows Rp;
      float pq[3];
      joint_rows(R, ph, cw, cv, tau, Dadd, cs, sn, lane_q, Rp, pq, Dinv);
      // halve, add the shin's rigid-body inertia in row form (constants per lane: P rows [I_O | m c x], Q rows [(m c x)^T | m 1])
      ABI Bd = abi_body<Model, 7>();
      float B6[6][6];
      abi_dense(Bd, B6);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) R.m[r][c] = 0.5f * Rp.m[r][c] + (lane_q ? B6[3 + r][c] : B6[r][c]);
#pragma unroll
      for (int k = 0; k < 3; ++k) ph[k] = 0.5f * pq[k];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) { I.A.m[k] = 0.5f * I.A.m[k]; I.C.m[k] = 0.5f * I.C.m[k]; }
      abi_add_body<Model, 7>(I);
      Dinv = I.A.m[0];
    }
    acc += Dinv;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) clocks[blockIdx.x] = t1 - t0;
  float out = acc;
  if (MODE == 1) out += I.A.m[0] + I.A.m[5] + I.B[1][2] + I.C.m[3] + pA.w[0] + pA.v[2];
  if (MODE == 2) out += R.m[0][0] + R.m[1][2] + R.m[1][5] + R.m[2][4] + ph[0] + ph[2];
  sink[blockIdx.x * 64 + lane] = out;
  // cross-check of the two formulations: entries both hold after the chain (block 0 only)
  if (blockIdx.x == 0 && lane < 2 && check) {
    if (MODE == 1 && lane == 0) { check[0] = I.A.m[0]; check[1] = I.A.m[4]; check[2] = I.B[0][1]; check[3] = I.C.m[2]; check[4] = I.B[2][0]; check[5] = pA.w[1]; check[6] = pA.v[0]; }
    if (MODE == 2) {
      if (!lane_q) { check[8] = R.m[0][0]; check[9] = R.m[0][2]; check[10] = R.m[0][4]; check[13] = ph[1]; }
      else { check[11] = R.m[2][5]; check[12] = R.m[0][2]; check[14] = ph[0]; }
    }
  }
}

template <int MODE>
static double run(const char* name, int iters, float* sink, unsigned long long* clk, float* check) {
  const int groups = 1024;
  hipLaunchKernelGGL(probe<MODE>, dim3(groups), dim3(64), 0, 0, 64, sink, clk, check);      // warm-up (code fetch)
  hipLaunchKernelGGL(probe<MODE>, dim3(groups), dim3(64), 0, 0, iters, sink, clk, check);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(groups);
  hipMemcpy(h.data(), clk, sizeof(unsigned long long) * groups, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[groups / 2] / iters, lo = (double)h[groups / 20] / iters, hi = (double)h[groups - groups / 20] / iters;
  printf("%-12s %8.1f shader clocks per iteration (5 %%: %.1f, 95 %%: %.1f)\n", name, med, lo, hi);
  return med;
}

int main() {
  float *sink, *check;
  unsigned long long* clk;
  hipMalloc(&sink, sizeof(float) * 1024 * 64);
  hipMalloc(&check, sizeof(float) * 16);
  hipMemset(check, 0, sizeof(float) * 16);
  hipMalloc(&clk, sizeof(unsigned long long) * 1024);
  const int iters = 4000;
  const double e = run<0>("empty loop", iters, sink, clk, nullptr);
  const double s = run<1>("single lane", 12, sink, clk, check), s2 = run<1>("single lane", iters, sink, clk, nullptr);
  const double p = run<2>("lane pair", 12, sink, clk, check), p2 = run<2>("lane pair", iters, sink, clk, nullptr);
  (void)s; (void)p;
  float h[16];
  hipMemcpy(h, check, sizeof h, hipMemcpyDeviceToHost);
  printf("cross-check after 12 chained joint steps (single lane | lane pair): A00 %.6g | %.6g, A02 %.6g | %.6g, B01 %.6g | %.6g, C22 %.6g | %.6g, "
         "B20 %.6g | %.6g, p.w1 %.6g | %.6g, p.v0 %.6g | %.6g\n", h[0], h[8], h[1], h[9], h[2], h[10], h[3], h[11], h[4], h[12], h[5], h[13], h[6], h[14]);
  printf("joint step: single lane %.1f clocks, lane pair %.1f clocks (loop overhead %.1f subtracted from neither): ratio %.3f\n", s2, p2, e, p2 / s2);
  return 0;
}
