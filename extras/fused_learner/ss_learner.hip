// ss_learner.hip -- fused PPO minibatch step for the reference's actor / critic-ensemble networks on gfx950.
//
// once the env step runs at 65 M env-steps/s the learner is 98 % of the PPO wall clock: one minibatch
// step of algorithms/ppo.py:55-100 (forward of SoftsignActor + critic ensemble, clipped-surrogate and value losses,
// backward, clip_grad_norm_, Adam) is ~90 tiny launches of a generic framework, ~0.87 ms at batch 1024 even inside a
// hipGraph.  Here it is 15 launches of hand-written kernels:
//   6 x forward layer   (actor layer l and every critic's layer l in ONE launch): Y = act(X W^T + b)
//   1 x loss            per sample: log-prob, ratio, clipped surrogate, value error -> output deltas, partial sums
//   6 x backward layer  dX = (delta W) * act'(X)  and  dW = delta^T X (batch split in S slices), db = sum(delta)
//   1 x reduce          G = sum of the S slices, partial sums of squares (deterministic order)
//   1 x adam            total norm -> clip coefficient -> Adam update of the flat parameter vector
// All GEMMs run on the matrix cores in EXACT f32 (v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, f32 accumulate), one
// 32x32 output tile per WORKGROUP (its four wavefronts split the reduction range and meet in LDS), operands streamed from
// L2 with a k-permutation that turns the per-lane operand into
// float4 loads (lane l covers k = 8c + 4 (l >> 5) + t, t = 0..3 of every 8-block: A and B use the same map, the sum is
// unchanged).  No LDS tiling: at these sizes (M = 1024, N = K = 256) the whole working set is L2-resident.
//
// Network (common/controller.py:217-261 and :55-145): actor 60 -> 256 x5 -> 21 with softsign x3, relu x2, tanh; critic e:
// 60 -> 256 x4 -> 1 with relu; state-independent log-std.  Parameters live in ONE flat f32 vector in torch's state_dict
// order per module (weights [out][in] row-major, then bias); the Python side keeps the nn.Parameters as views into it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

#include "steppingstone_learner.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define SSL_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t _e = (call);                                                                    \
    if (_e != hipSuccess) return fail(-2, std::string(#call) + ": " + hipGetErrorString(_e));  \
  } while (0)

constexpr int kObs = 60, kHid = 256, kAct = 21, kOutPad = 32;
constexpr int kActorLayers = 6, kCriticLayers = 5, kMaxEns = 4, kMaxSplit = 8;
constexpr int kLossRows = 32;      // samples per loss-kernel workgroup (and per row of its partial sums)
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SOFTSIGN = 2, ACT_TANH = 3 };

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Layer {           // one Linear: y[n_out] = W[n_out][n_in] x + b
  long long w_off, b_off;
  int n_in, n_out, act;
};

struct Net {
  Layer actor[kActorLayers];
  Layer critic[kMaxEns][kCriticLayers];
  long long logstd_off;
  long long n_params;
  int n_ens;
};

__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(x, 0.f);
    case ACT_SOFTSIGN: return x / (1.f + fabsf(x));
    case ACT_TANH: return tanhf(x);
    default: return x;
  }
}
// derivative expressed through the OUTPUT y of the activation
__device__ __forceinline__ float act_bwd(float y, int act) {
  switch (act) {
    case ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ACT_SOFTSIGN: { float t = 1.f - fabsf(y); return t * t; }     // y = x/(1+|x|)  ->  dy/dx = (1-|y|)^2
    case ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// One 32x32 tile: acc += sum_k A(i, k) B(k, j) over k in [0, K) (K a multiple of 8 after masking), on v_mfma_f32_32x32x2.
// Loaders return the four operand values of this lane for the 8-block c: k = 8c + 4 kk + t.
template <class LA, class LB>
__device__ __forceinline__ void tile_gemm(f32x16& acc, int c0, int c1, LA la, LB lb) {
  if (c0 >= c1) return;
  float4 a = la(c0), b = lb(c0);
#pragma unroll 1
  for (int c = c0; c < c1; ++c) {
    float4 an = a, bn = b;
    if (c + 1 < c1) { an = la(c + 1); bn = lb(c + 1); }          // prefetch the next block under the four MFMAs
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    a = an; b = bn;
  }
}
// C/D element (reg r of lane l): row = (r & 3) + 8 (r >> 2) + 4 (l >> 5), col = l & 31
__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// The four wavefronts of a workgroup share ONE 32x32 output tile and split its reduction (K) range four ways: at the
// reference's minibatch of 1024 a layer has only 256 output tiles, so one wavefront per tile would leave three SIMDs of
// every CU idle behind a 128-deep dependent MFMA chain (64 cycles each).  Partial accumulators meet in LDS (16 KB);
// wavefront w then finalises accumulator registers 4w .. 4w+3 (rows c_row(4w + q, lane)) in a fixed summation order.
__device__ __forceinline__ void split_range(int nblocks, int wave, int& c0, int& c1) {
  const int per = (nblocks + 3) / 4;
  c0 = wave * per < nblocks ? wave * per : nblocks;
  c1 = c0 + per < nblocks ? c0 + per : nblocks;
}
__device__ __forceinline__ void combine4(const f32x16& acc, float (*red)[16][64], int wave, int lane, float out[4]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 4 * wave + q;
    out[q] = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
  }
}

// Left-right mirror augmentation (common/envs_utils.py:687-740, PPO.update's mirror_function): the minibatch is doubled, row
// B + i being the mirror image of row i -- column c of the mirrored observation / action is sgn[src] * x[src], src = perm[c]
// (negate the lateral quantities, then swap the right and left limbs).  Only the kernels that read the rollout arrays
// through idx know about it: first-layer forward, first-layer weight gradient, loss.
struct Mirror {
  const int* obs_perm; const float* obs_sgn;     // [60]
  const int* act_perm; const float* act_sgn;     // [21]
  int half;                                      // B: rows >= half are mirrored copies of row - half; 0 = no augmentation
};

struct FwdJob {            // Y[M][ldy] = act(X[M][ldx] W^T + b); X rows optionally gathered through idx
  const float* X; int ldx;
  const long long* idx;    // null or [M]: row i of X is X[idx[i]]
  const float* W; const float* b; int K, N, act;
  float* Y; int ldy;       // ldy >= N rounded up to 32; columns >= N are written as 0
};
struct FwdArgs { FwdJob job[1 + kMaxEns]; int njobs; int M; Mirror mir; };

__global__ __launch_bounds__(256) void fwd_layer_kernel(FwdArgs A) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_m = A.M / 32;
  int t = blockIdx.x;
  for (int jb = 0; jb < A.njobs; ++jb) {
    const FwdJob& J = A.job[jb];
    const int tiles_n = (J.N + 31) / 32, ntiles = tiles_m * tiles_n;
    if (t >= ntiles) { t -= ntiles; continue; }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int i = tm * 32 + (lane & 31), j = tn * 32 + (lane & 31), kk = lane >> 5;
    const bool mirrored = J.idx && A.mir.half > 0 && i >= A.mir.half;
    const long long xi = J.idx ? J.idx[mirrored ? i - A.mir.half : i] : (long long)i;
    const float* xrow = J.X + xi * J.ldx;
    const float* wrow = J.W + (long long)(j < J.N ? j : 0) * J.K;
    const bool jok = j < J.N;
    const int K = J.K;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto la = [&](int c) {
      const int k0 = 8 * c + 4 * kk;
      if (k0 + 3 >= K) return make_float4(0, 0, 0, 0);
      if (!mirrored) return *reinterpret_cast<const float4*>(xrow + k0);
      const int* pm = A.mir.obs_perm + k0;
      const int s0 = pm[0], s1 = pm[1], s2 = pm[2], s3 = pm[3];
      return make_float4(A.mir.obs_sgn[s0] * xrow[s0], A.mir.obs_sgn[s1] * xrow[s1], A.mir.obs_sgn[s2] * xrow[s2], A.mir.obs_sgn[s3] * xrow[s3]);
    };
    auto lb = [&](int c) { const int k0 = 8 * c + 4 * kk; return (jok && k0 + 3 < K) ? *reinterpret_cast<const float4*>(wrow + k0) : make_float4(0, 0, 0, 0); };
    int c0, c1;
    split_range((K + 7) / 8, wave, c0, c1);
    tile_gemm(acc, c0, c1, la, lb);
    float o[4];
    combine4(acc, red, wave, lane, o);
    const float bias = jok ? J.b[j] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = tm * 32 + c_row(4 * wave + q, lane);
      J.Y[(long long)row * J.ldy + j] = jok ? act_fwd(o[q] + bias, J.act) : 0.f;
    }
    return;
  }
}

struct BwdJob {            // layer l of one net: delta [M][ldd] (gradient w.r.t. the layer's pre-activation), input X [M][ldx]
  const float* D; int ldd;
  const float* X; int ldx; const long long* idx;
  const float* W; int K, N;
  float* DX; int lddx; int act_prev;   // delta of the previous layer = (D W) * act'(X) -> [M][lddx]; null for the first layer
  float* GW; float* GB;                // slice-0 pointers of dW [N][K] and db [N] inside the partial-gradient buffer
};
struct BwdArgs { BwdJob job[1 + kMaxEns]; int njobs; int M; int S; long long slice_stride; Mirror mir; };

__global__ __launch_bounds__(256) void bwd_layer_kernel(BwdArgs A) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_m = A.M / 32;
  int t = blockIdx.x;
  const int kk = lane >> 5, l31 = lane & 31;
  for (int jb = 0; jb < A.njobs; ++jb) {
    const BwdJob& J = A.job[jb];
    const int K = J.K, N = J.N;
    // ---- (a) dX tiles: [M x K], reduction over n
    if (J.DX) {
      const int tiles_k = (K + 31) / 32, ntiles = tiles_m * tiles_k;
      if (t < ntiles) {
        const int tm = t / tiles_k, tk = t - tm * tiles_k;
        const int i = tm * 32 + l31, kcol = tk * 32 + l31;
        const float* drow = J.D + (long long)i * J.ldd;
        const bool kok = kcol < K;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto la = [&](int c) { const int n0 = 8 * c + 4 * kk; return (n0 + 3 < J.ldd) ? *reinterpret_cast<const float4*>(drow + n0) : make_float4(0, 0, 0, 0); };
        auto lb = [&](int c) {
          const int n0 = 8 * c + 4 * kk;
          float4 v;
          v.x = (kok && n0 + 0 < N) ? J.W[(long long)(n0 + 0) * K + kcol] : 0.f;
          v.y = (kok && n0 + 1 < N) ? J.W[(long long)(n0 + 1) * K + kcol] : 0.f;
          v.z = (kok && n0 + 2 < N) ? J.W[(long long)(n0 + 2) * K + kcol] : 0.f;
          v.w = (kok && n0 + 3 < N) ? J.W[(long long)(n0 + 3) * K + kcol] : 0.f;
          return v;
        };
        int c0, c1;
        split_range((N + 7) / 8, wave, c0, c1);
        tile_gemm(acc, c0, c1, la, lb);
        float o[4];
        combine4(acc, red, wave, lane, o);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = tm * 32 + c_row(4 * wave + q, lane);
          const float y = kok ? J.X[(long long)row * J.ldx + kcol] : 0.f;     // hidden layers are never gathered
          J.DX[(long long)row * J.lddx + kcol] = kok ? o[q] * act_bwd(y, J.act_prev) : 0.f;
        }
        return;
      }
      t -= ntiles;
    }
    // ---- (b) dW tiles: [N x K] per batch slice, reduction over the slice's rows
    {
      const int tiles_n = (N + 31) / 32, tiles_k = (K + 31) / 32, ntiles = tiles_n * tiles_k * A.S;
      if (t < ntiles) {
        const int s = t / (tiles_n * tiles_k), r2 = t - s * (tiles_n * tiles_k);
        const int tn = r2 / tiles_k, tk = r2 - tn * tiles_k;
        const int n = tn * 32 + l31, kcol = tk * 32 + l31;
        const bool nok = n < N, kok = kcol < K;
        const int rows = A.M / A.S, i0 = s * rows;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto la = [&](int c) {
          const int i = i0 + 8 * c + 4 * kk;
          float4 v;
          v.x = nok ? J.D[(long long)(i + 0) * J.ldd + n] : 0.f;
          v.y = nok ? J.D[(long long)(i + 1) * J.ldd + n] : 0.f;
          v.z = nok ? J.D[(long long)(i + 2) * J.ldd + n] : 0.f;
          v.w = nok ? J.D[(long long)(i + 3) * J.ldd + n] : 0.f;
          return v;
        };
        auto lb = [&](int c) {
          const int i = i0 + 8 * c + 4 * kk;
          float4 v;
          if (J.idx) {
            const int half = A.mir.half;
            auto gx = [&](int row) {
              if (!kok) return 0.f;
              if (half > 0 && row >= half) {
                const int src = A.mir.obs_perm[kcol];
                return A.mir.obs_sgn[src] * J.X[J.idx[row - half] * J.ldx + src];
              }
              return J.X[J.idx[row] * J.ldx + kcol];
            };
            v.x = gx(i + 0); v.y = gx(i + 1); v.z = gx(i + 2); v.w = gx(i + 3);
          } else {
            v.x = kok ? J.X[(long long)(i + 0) * J.ldx + kcol] : 0.f;
            v.y = kok ? J.X[(long long)(i + 1) * J.ldx + kcol] : 0.f;
            v.z = kok ? J.X[(long long)(i + 2) * J.ldx + kcol] : 0.f;
            v.w = kok ? J.X[(long long)(i + 3) * J.ldx + kcol] : 0.f;
          }
          return v;
        };
        int c0, c1;
        split_range(rows / 8, wave, c0, c1);
        tile_gemm(acc, c0, c1, la, lb);
        float o[4];
        combine4(acc, red, wave, lane, o);
        float* gw = J.GW + (long long)s * A.slice_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = tn * 32 + c_row(4 * wave + q, lane);
          if (row < N && kok) gw[(long long)row * K + kcol] = o[q];
        }
        return;
      }
      t -= ntiles;
    }
    // ---- (c) db: column sums of the slice
    {
      const int tiles_n = (N + 31) / 32, ntiles = tiles_n * A.S;
      if (t < ntiles) {
        const int s = t / tiles_n, tn = t - s * tiles_n;
        const int n = tn * 32 + l31;
        const int rows = A.M / A.S, i0 = s * rows;
        float sum = 0.f;
        if (n < N)
          for (int i = i0 + 2 * wave + kk; i < i0 + rows; i += 8) sum += J.D[(long long)i * J.ldd + n];
        sum += __shfl_xor(sum, 32);
        red[wave][0][lane] = sum;
        __syncthreads();
        if (wave == 0 && kk == 0 && n < N)
          (J.GB + (long long)s * A.slice_stride)[n] = ((red[0][0][lane] + red[1][0][lane]) + red[2][0][lane]) + red[3][0][lane];
        return;
      }
      t -= ntiles;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct LossArgs {
  const float* mean; int ldm;            // [M][32] actor output (tanh applied)
  const float* val[kMaxEns]; int ldv;    // [M][32], column 0
  const float* logstd;                   // [21]
  const float* act; const float* old_logp; const float* adv; const float* ret; const float* vpred;   // full rollout arrays
  const long long* idx;                  // [M] rows of the rollout arrays used by this minibatch
  float* dmean; float* dval[kMaxEns];    // output deltas (pre-activation), same shapes as mean / val
  float* part;                           // [M / kLossRows][32]: 0..20 dlogstd, 21 action-loss sum, 22 value-loss sum
  int M, n_ens; float clip; int clipped_value_loss;
  Mirror mir;
};

// Half a wavefront (32 lanes) per sample, lane j = column j of the padded 32-wide output rows: every row of the actor's
// mean / its delta and of the critics' value / delta is read and written as ONE coalesced 128-byte access (a thread per
// sample wrote 64 scattered lines per store instruction: 18 us of a 150 us step).  The 21-term log-prob sum is a
// 5-step xor-shuffle inside the half; the log-std gradient is accumulated per lane over the samples of the half and
// combined across the eight halves of the block in LDS, fixed order.
__global__ __launch_bounds__(256) void loss_kernel(LossArgs A) {
  __shared__ float red[8][24];
  const int lane = threadIdx.x & 63, j = lane & 31, half = threadIdx.x >> 5;     // 8 halves per block
  const bool jact = j < kAct;
  const float ls = jact ? A.logstd[j] : 0.f;
  const float isd = jact ? expf(-2.f * ls) : 0.f;                                  // 1 / sigma^2
  float g_ls = 0.f, g_al = 0.f, g_vl = 0.f;
  const int per_block = kLossRows;                                                 // samples per block: four per half-wavefront
  const int i_end = min(A.M, (int)(blockIdx.x + 1) * per_block);
  for (int i = blockIdx.x * per_block + half; i < i_end; i += 8) {
    const bool mirrored = A.mir.half > 0 && i >= A.mir.half;
    const long long r = A.idx ? A.idx[mirrored ? i - A.mir.half : i] : (long long)i;
    const float mu = A.mean[(long long)i * A.ldm + j];
    float aj = 0.f;
    if (jact) {
      const int src = mirrored ? A.mir.act_perm[j] : j;
      aj = A.act[r * kAct + src] * (mirrored ? A.mir.act_sgn[src] : 1.f);
    }
    const float z = jact ? aj - mu : 0.f;
    float term = jact ? -(z * z) * (0.5f * isd) - ls - 0.9189385332046727f : 0.f;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) term += __shfl_xor(term, off);        // stays inside the 32-lane half
    const float logp = term;
    const float adv = A.adv[r];
    const float ratio = expf(logp - A.old_logp[r]);
    const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, 1.f - A.clip), 1.f + A.clip) * adv;
    // d(-min(s1,s2))/dlogp: the unclipped branch carries the gradient whenever it is the smaller one or the ratio is
    // inside the clip range (then s1 == s2 and the two halves of torch.min's tie-split add up to the same thing)
    const bool inside = ratio >= 1.f - A.clip && ratio <= 1.f + A.clip;
    const float glogp = (inside || s1 < s2) ? -adv * ratio / (float)A.M : 0.f;
    A.dmean[(long long)i * A.ldm + j] = jact ? glogp * z * isd * (1.f - mu * mu) : 0.f;   // through the output tanh
    g_ls += jact ? glogp * (z * z * isd - 1.f) : 0.f;                                      // d logp / d logstd
    if (j == 0) g_al += -fminf(s1, s2);
    const float ret = A.ret[r];
    const float inv = 1.f / ((float)A.M * (float)A.n_ens);
    for (int e = 0; e < A.n_ens; ++e) {
      const float v = A.val[e][(long long)i * A.ldv];                                // column 0, broadcast read
      float err = v - ret, gv = err * inv, lv = 0.5f * err * err;
      if (A.clipped_value_loss) {
        const float vp = A.vpred[r];
        const float vc = vp + fminf(fmaxf(v - vp, -A.clip), A.clip);
        const float e2 = vc - ret;
        if (e2 * e2 > err * err) { lv = 0.5f * e2 * e2; gv = (fabsf(v - vp) < A.clip) ? e2 * inv : 0.f; }
      }
      if (j == 0) g_vl += lv;
      A.dval[e][(long long)i * A.ldv + j] = j == 0 ? gv : 0.f;
    }
  }
  if (jact) red[half][j] = g_ls;
  if (j == 0) { red[half][21] = g_al; red[half][22] = g_vl; }
  __syncthreads();
  if (threadIdx.x < 23) {
    float v = 0.f;
    for (int h = 0; h < 8; ++h) v += red[h][threadIdx.x];
    A.part[blockIdx.x * 32 + threadIdx.x] = v;
  }
}

// G[p] = sum over slices; logstd gradient and the loss sums from the loss kernel's block partials; per-block sum of squares
struct ReduceArgs {
  const float* gpart; long long slice_stride; int S;
  const float* lpart; int nlossblocks; long long logstd_off;
  float* G; long long n; float* sq;      // sq[gridDim]
  float* stats;                          // [3]: value loss, action loss, entropy
  const float* params; int M, n_ens;
};
__global__ __launch_bounds__(256) void reduce_kernel(ReduceArgs A) {
  __shared__ float red[256];
  const int nblk = (int)((A.n + 255) / 256);
  if ((int)blockIdx.x == nblk) {
    // the extra workgroup: column sums of the loss kernel's partial rows (log-std gradient 0..20, action / value loss sums
    // 21, 22), each by a strided per-thread sum and a fixed tree
    float gsq = 0.f;
    for (int c = 0; c < 23; ++c) {
      float part = 0.f;
      for (int b2 = threadIdx.x; b2 < A.nlossblocks; b2 += 256) part += A.lpart[b2 * 32 + c];
      red[threadIdx.x] = part;
      __syncthreads();
      for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
      }
      const float tot = red[0];
      __syncthreads();
      if (threadIdx.x == 0) {
        if (c < kAct) { A.G[A.logstd_off + c] = tot; gsq += tot * tot; }
        else if (c == 21) A.stats[1] = tot / (float)A.M;
        else A.stats[0] = tot / ((float)A.M * (float)A.n_ens);
      }
    }
    if (threadIdx.x == 0) {
      A.sq[nblk] = gsq;
      float ent = 0.f;
      for (int j = 0; j < kAct; ++j) ent += 0.5f + 0.9189385332046727f + A.params[A.logstd_off + j];
      A.stats[2] = ent;
    }
    return;
  }
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  float g = 0.f;
  const bool is_logstd = p >= A.logstd_off && p < A.logstd_off + kAct;
  if (p < A.n && !is_logstd) {
    for (int s = 0; s < A.S; ++s) g += A.gpart[(long long)s * A.slice_stride + p];
    A.G[p] = g;
  }
  float v = g * g;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) A.sq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Data-parallel path: the all-reduced gradient comes back from the collective; scale it (1 / world size) and rebuild the
// per-block sums of squares the Adam kernel's clip coefficient is computed from (same 256-element blocks, fixed order).
__global__ __launch_bounds__(256) void scale_sq_kernel(float* G, long long n, float scale, float* sq) {
  __shared__ float red[4];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  float g = 0.f;
  if (p < n) { g = G[p] * scale; G[p] = g; }
  float v = g * g;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) sq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
  float* P; const float* G; float* m; float* v; long long n;
  const float* sq; int nsq;
  const float* lr; const float* step;     // device scalars: learning rate, step count AFTER this update (float)
  float beta1, beta2, eps, max_norm;
};
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs A) {
  __shared__ float red[256];
  __shared__ float coef_s, bc1_s, bc2s_s;
  {   // total squared norm: fixed summation order (thread t: entries t, t+256, ...; then a fixed tree), identical in every block
    float part = 0.f;
    for (int b = threadIdx.x; b < A.nsq; b += 256) part += A.sq[b];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float c = A.max_norm / (sqrtf(red[0]) + 1e-6f);
      coef_s = c < 1.f ? c : 1.f;
      const float t = *A.step;
      bc1_s = 1.f - powf(A.beta1, t);
      bc2s_s = sqrtf(1.f - powf(A.beta2, t));
    }
    __syncthreads();
  }
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= A.n) return;
  const float g = A.G[p] * coef_s;
  const float lr = *A.lr;
  const float m = A.beta1 * A.m[p] + (1.f - A.beta1) * g;
  const float v = A.beta2 * A.v[p] + (1.f - A.beta2) * g * g;
  A.m[p] = m;
  A.v[p] = v;
  const float denom = sqrtf(v) / bc2s_s + A.eps;
  A.P[p] -= (lr / bc1_s) * (m / denom);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
struct ssl_learner {
  Net net;
  int device, max_batch;
  float* act_buf[kActorLayers];            // outputs of actor layers 1..6 ([M][256] x5, [M][32])
  float* cri_buf[kMaxEns][kCriticLayers];
  float* dA[2];                            // ping-pong deltas of the actor ([M][256])
  float* dC[kMaxEns][2];
  float* dout;                             // [M][32] actor output delta
  float* dvout[kMaxEns];                   // [M][32]
  float* gpart;                            // [S][n_params]
  float* G;                                // [n_params]
  float* lpart;                            // [max_batch/256][32]
  float* sq;                               // [ceil(n_params/256)]
  float* stats;                            // [3]
};

namespace {
Net make_net(int n_ens) {
  Net n;
  std::memset(&n, 0, sizeof n);
  long long o = 0;
  auto al4 = [](long long x) { return (x + 3) / 4 * 4; };          // every tensor starts 16-byte aligned (float4 operand loads)
  const int ain[kActorLayers] = {kObs, kHid, kHid, kHid, kHid, kHid}, aout[kActorLayers] = {kHid, kHid, kHid, kHid, kHid, kAct};
  const int aact[kActorLayers] = {ACT_SOFTSIGN, ACT_SOFTSIGN, ACT_SOFTSIGN, ACT_RELU, ACT_RELU, ACT_TANH};
  // torch order of ActorCritic.parameters(): logstd, actor.fc1..out (weight, bias), critics.e.(0,2,4,6,8) (weight, bias)
  n.logstd_off = o; o = al4(o + kAct);
  for (int l = 0; l < kActorLayers; ++l) {
    const long long w = o, b = al4(w + (long long)ain[l] * aout[l]);
    n.actor[l] = Layer{w, b, ain[l], aout[l], aact[l]};
    o = al4(b + aout[l]);
  }
  const int cin[kCriticLayers] = {kObs, kHid, kHid, kHid, kHid}, cout[kCriticLayers] = {kHid, kHid, kHid, kHid, 1};
  for (int e = 0; e < n_ens; ++e)
    for (int l = 0; l < kCriticLayers; ++l) {
      const long long w = o, b = al4(w + (long long)cin[l] * cout[l]);
      n.critic[e][l] = Layer{w, b, cin[l], cout[l], l + 1 < kCriticLayers ? ACT_RELU : ACT_NONE};
      o = al4(b + cout[l]);
    }
  n.n_params = o;
  n.n_ens = n_ens;
  return n;
}
}  // namespace

extern "C" {

const char* ssl_last_error(void) { return g_err.c_str(); }

int64_t ssl_num_params(int32_t n_ens) { return (n_ens < 1 || n_ens > kMaxEns) ? -1 : make_net(n_ens).n_params; }

int ssl_create(ssl_learner** out, int device, int32_t n_ens, int32_t max_batch) {
  if (!out) return fail(-1, "out is null");
  *out = nullptr;
  if (n_ens < 1 || n_ens > kMaxEns) return fail(-1, "1 <= num_ensembles <= 4");
  if (max_batch < 32 || max_batch % 32) return fail(-1, "max_batch must be a positive multiple of 32");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(-3, "no HIP device visible: the fused learner has no CPU path");
  SSL_HIP(hipSetDevice(device));
  ssl_learner* L = new ssl_learner();
  std::memset(L, 0, sizeof *L);
  L->net = make_net(n_ens);
  L->device = device;
  L->max_batch = max_batch;
  const size_t M = (size_t)max_batch;
  auto alloc = [&](float** p, size_t n) { return hipMalloc(p, n * sizeof(float)) == hipSuccess && hipMemset(*p, 0, n * sizeof(float)) == hipSuccess; };
  bool ok = true;
  for (int l = 0; l < kActorLayers; ++l) ok = ok && alloc(&L->act_buf[l], M * (l + 1 < kActorLayers ? kHid : kOutPad));
  for (int e = 0; e < n_ens; ++e)
    for (int l = 0; l < kCriticLayers; ++l) ok = ok && alloc(&L->cri_buf[e][l], M * (l + 1 < kCriticLayers ? kHid : kOutPad));
  for (int k = 0; k < 2; ++k) ok = ok && alloc(&L->dA[k], M * kHid);
  for (int e = 0; e < n_ens; ++e)
    for (int k = 0; k < 2; ++k) ok = ok && alloc(&L->dC[e][k], M * kHid);
  ok = ok && alloc(&L->dout, M * kOutPad);
  for (int e = 0; e < n_ens; ++e) ok = ok && alloc(&L->dvout[e], M * kOutPad);
  ok = ok && alloc(&L->gpart, (size_t)kMaxSplit * L->net.n_params);
  ok = ok && alloc(&L->G, (size_t)L->net.n_params);
  ok = ok && alloc(&L->lpart, (M / kLossRows + 1) * 32);
  ok = ok && alloc(&L->sq, (size_t)(L->net.n_params + 255) / 256 + 1);
  ok = ok && alloc(&L->stats, 4);
  if (!ok) { ssl_destroy(L); return fail(-4, "hipMalloc failed for the learner workspace"); }
  SSL_HIP(hipDeviceSynchronize());
  *out = L;
  return 0;
}

void ssl_destroy(ssl_learner* L) {
  if (!L) return;
  (void)hipSetDevice(L->device);
  for (int l = 0; l < kActorLayers; ++l) if (L->act_buf[l]) (void)hipFree(L->act_buf[l]);
  for (int e = 0; e < kMaxEns; ++e) {
    for (int l = 0; l < kCriticLayers; ++l) if (L->cri_buf[e][l]) (void)hipFree(L->cri_buf[e][l]);
    for (int k = 0; k < 2; ++k) if (L->dC[e][k]) (void)hipFree(L->dC[e][k]);
    if (L->dvout[e]) (void)hipFree(L->dvout[e]);
  }
  for (int k = 0; k < 2; ++k) if (L->dA[k]) (void)hipFree(L->dA[k]);
  float* rest[] = {L->dout, L->gpart, L->G, L->lpart, L->sq, L->stats};
  for (float* p : rest) if (p) (void)hipFree(p);
  delete L;
}

// forward, loss, backward, slice reduction: the gradient of the minibatch loss lands in `grad` ([n_params], device), the
// per-block sums of squares in L->sq, the three loss values in stats_out
static int grad_impl(ssl_learner* L, const float* params, const float* obs, const float* act, const float* old_logp, const float* adv,
                     const float* ret, const float* vpred, const int64_t* idx, int32_t batch, float clip_param,
                     int32_t use_clipped_value_loss, float* stats_out, float* grad, void* stream, const Mirror& mir) {
  if (!L || !params || !obs || !act || !old_logp || !adv || !ret || !idx || !grad) return fail(-1, "null argument");
  const int rows_total = mir.half > 0 ? 2 * batch : batch;
  if (batch < 32 || batch % 32 || rows_total > L->max_batch)
    return fail(-1, "batch must be a multiple of 32 and (doubled, with the mirror augmentation) <= max_batch");
  SSL_HIP(hipSetDevice(L->device));
  hipStream_t st = (hipStream_t)stream;
  const Net& N = L->net;
  const int M = rows_total, E = N.n_ens;
  // batch slices of the weight-gradient GEMMs (each slice's four wavefronts split it again); every slice must be a multiple
  // of the 8-row MFMA block
  int S = kMaxSplit;                 // measured at M = 1024 .. 4096: 8 slices beat 4 and 2 (150 vs 189 vs 230 us per step at 1024)
  if (const char* es = std::getenv("SSL_SPLIT")) S = std::atoi(es) > 0 ? std::atoi(es) : S;   // tuning override
  while (S > 1 && (M % (8 * S) != 0)) --S;
  const long long* idx64 = reinterpret_cast<const long long*>(idx);

  // ---- forward: launch l runs actor layer l and every critic's layer l
  for (int l = 0; l < kActorLayers; ++l) {
    FwdArgs A;
    std::memset(&A, 0, sizeof A);
    A.M = M;
    A.mir = mir;
    int total = 0;
    auto add = [&](const Layer& Ly, const float* X, int ldx, const long long* ix, float* Y) {
      FwdJob& J = A.job[A.njobs++];
      J.X = X; J.ldx = ldx; J.idx = ix; J.W = params + Ly.w_off; J.b = params + Ly.b_off; J.K = Ly.n_in; J.N = Ly.n_out; J.act = Ly.act;
      J.Y = Y; J.ldy = (Ly.n_out + 31) / 32 * 32;
      total += (M / 32) * ((Ly.n_out + 31) / 32);
    };
    add(N.actor[l], l == 0 ? obs : L->act_buf[l - 1], l == 0 ? kObs : kHid, l == 0 ? idx64 : nullptr, L->act_buf[l]);
    if (l < kCriticLayers)
      for (int e = 0; e < E; ++e)
        add(N.critic[e][l], l == 0 ? obs : L->cri_buf[e][l - 1], l == 0 ? kObs : kHid, l == 0 ? idx64 : nullptr, L->cri_buf[e][l]);
    hipLaunchKernelGGL(fwd_layer_kernel, dim3(total), dim3(256), 0, st, A);
  }
  // ---- loss
  {
    LossArgs A;
    std::memset(&A, 0, sizeof A);
    A.mean = L->act_buf[kActorLayers - 1]; A.ldm = kOutPad; A.ldv = kOutPad;
    for (int e = 0; e < E; ++e) { A.val[e] = L->cri_buf[e][kCriticLayers - 1]; A.dval[e] = L->dvout[e]; }
    A.logstd = params + N.logstd_off;
    A.act = act; A.old_logp = old_logp; A.adv = adv; A.ret = ret; A.vpred = vpred; A.idx = idx64;
    A.dmean = L->dout; A.part = L->lpart; A.M = M; A.n_ens = E; A.clip = clip_param; A.mir = mir;
    A.clipped_value_loss = (use_clipped_value_loss && vpred) ? 1 : 0;
    hipLaunchKernelGGL(loss_kernel, dim3((M + kLossRows - 1) / kLossRows), dim3(256), 0, st, A);
  }
  // ---- backward: launch for layer l = last..first of the actor; critics' layer (l-1) ride along (they have one layer less)
  for (int l = kActorLayers - 1; l >= 0; --l) {
    BwdArgs A;
    std::memset(&A, 0, sizeof A);
    A.M = M; A.S = S; A.slice_stride = N.n_params; A.mir = mir;
    int total = 0;
    auto add = [&](const Layer& Ly, const float* D, int ldd, const float* X, int ldx, const long long* ix, float* DX, int act_prev) {
      BwdJob& J = A.job[A.njobs++];
      J.D = D; J.ldd = ldd; J.X = X; J.ldx = ldx; J.idx = ix; J.W = params + Ly.w_off; J.K = Ly.n_in; J.N = Ly.n_out;
      J.DX = DX; J.lddx = kHid; J.act_prev = act_prev;
      J.GW = L->gpart + Ly.w_off; J.GB = L->gpart + Ly.b_off;
      const int tn = (Ly.n_out + 31) / 32, tk = (Ly.n_in + 31) / 32;
      total += (DX ? (M / 32) * tk : 0) + tn * tk * S + tn * S;
    };
    {
      const float* D = l == kActorLayers - 1 ? L->dout : L->dA[(l + 1) & 1];
      add(N.actor[l], D, l == kActorLayers - 1 ? kOutPad : kHid, l == 0 ? obs : L->act_buf[l - 1], l == 0 ? kObs : kHid,
          l == 0 ? idx64 : nullptr, l == 0 ? nullptr : L->dA[l & 1], l == 0 ? ACT_NONE : N.actor[l - 1].act);
    }
    const int lc = l - 1;                       // critic layer index handled in this launch
    if (lc >= 0)
      for (int e = 0; e < E; ++e) {
        const float* D = lc == kCriticLayers - 1 ? L->dvout[e] : L->dC[e][(lc + 1) & 1];
        add(N.critic[e][lc], D, lc == kCriticLayers - 1 ? kOutPad : kHid, lc == 0 ? obs : L->cri_buf[e][lc - 1], lc == 0 ? kObs : kHid,
            lc == 0 ? idx64 : nullptr, lc == 0 ? nullptr : L->dC[e][lc & 1], lc == 0 ? ACT_NONE : N.critic[e][lc - 1].act);
      }
    hipLaunchKernelGGL(bwd_layer_kernel, dim3(total), dim3(256), 0, st, A);
  }
  // ---- reduce slices, norm partials, losses
  const int nblk = (int)((N.n_params + 255) / 256);
  {
    ReduceArgs A;
    A.gpart = L->gpart; A.slice_stride = N.n_params; A.S = S; A.lpart = L->lpart; A.nlossblocks = (M + kLossRows - 1) / kLossRows;
    A.logstd_off = N.logstd_off; A.G = grad; A.n = N.n_params; A.sq = L->sq; A.stats = stats_out ? stats_out : L->stats;
    A.params = params; A.M = M; A.n_ens = E;
    hipLaunchKernelGGL(reduce_kernel, dim3(nblk + 1), dim3(256), 0, st, A);
  }
  SSL_HIP(hipGetLastError());
  return 0;
}

// clip by the global norm (from nsq block sums in L->sq) and one Adam step
static int apply_impl(ssl_learner* L, float* params, float* adam_m, float* adam_v, const float* lr, const float* step, const float* grad,
                      int nsq, float max_grad_norm, float adam_eps, void* stream) {
  if (!L || !params || !adam_m || !adam_v || !lr || !step || !grad) return fail(-1, "null argument");
  const Net& N = L->net;
  const int nblk = (int)((N.n_params + 255) / 256);
  AdamArgs A;
  A.P = params; A.G = grad; A.m = adam_m; A.v = adam_v; A.n = N.n_params; A.sq = L->sq; A.nsq = nsq; A.lr = lr; A.step = step;
  A.beta1 = 0.9f; A.beta2 = 0.999f; A.eps = adam_eps; A.max_norm = max_grad_norm;
  hipLaunchKernelGGL(adam_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, A);
  SSL_HIP(hipGetLastError());
  return 0;
}

static int step_impl(ssl_learner* L, float* params, float* adam_m, float* adam_v, const float* lr, const float* step, const float* obs,
                     const float* act, const float* old_logp, const float* adv, const float* ret, const float* vpred, const int64_t* idx,
                     int32_t batch, float clip_param, float max_grad_norm, float adam_eps, int32_t use_clipped_value_loss,
                     float* stats_out, void* stream, const Mirror& mir) {
  if (!L || !adam_m || !adam_v || !lr || !step) return fail(-1, "null argument");
  if (int rc = grad_impl(L, params, obs, act, old_logp, adv, ret, vpred, idx, batch, clip_param, use_clipped_value_loss, stats_out,
                         L->G, stream, mir)) return rc;
  return apply_impl(L, params, adam_m, adam_v, lr, step, L->G, (int)((L->net.n_params + 255) / 256) + 1, max_grad_norm, adam_eps, stream);
}

int ssl_step(ssl_learner* L, float* params, float* adam_m, float* adam_v, const float* lr, const float* step, const float* obs,
             const float* act, const float* old_logp, const float* adv, const float* ret, const float* vpred, const int64_t* idx,
             int32_t batch, float clip_param, float max_grad_norm, float adam_eps, int32_t use_clipped_value_loss, float* stats_out,
             void* stream) {
  Mirror none;
  std::memset(&none, 0, sizeof none);
  return step_impl(L, params, adam_m, adam_v, lr, step, obs, act, old_logp, adv, ret, vpred, idx, batch, clip_param, max_grad_norm,
                   adam_eps, use_clipped_value_loss, stats_out, stream, none);
}

int ssl_step_mirror(ssl_learner* L, float* params, float* adam_m, float* adam_v, const float* lr, const float* step, const float* obs,
                    const float* act, const float* old_logp, const float* adv, const float* ret, const float* vpred,
                    const int64_t* idx, int32_t batch, float clip_param, float max_grad_norm, float adam_eps,
                    int32_t use_clipped_value_loss, float* stats_out, void* stream, const int32_t* obs_perm, const float* obs_sgn,
                    const int32_t* act_perm, const float* act_sgn) {
  if (!obs_perm || !obs_sgn || !act_perm || !act_sgn) return fail(-1, "null mirror table");
  Mirror m{obs_perm, obs_sgn, act_perm, act_sgn, batch};
  return step_impl(L, params, adam_m, adam_v, lr, step, obs, act, old_logp, adv, ret, vpred, idx, batch, clip_param, max_grad_norm,
                   adam_eps, use_clipped_value_loss, stats_out, stream, m);
}

int ssl_grad(ssl_learner* L, const float* params, const float* obs, const float* act, const float* old_logp, const float* adv,
             const float* ret, const float* vpred, const int64_t* idx, int32_t batch, float clip_param, int32_t use_clipped_value_loss,
             float* stats_out, float* grad_out, void* stream, const int32_t* obs_perm, const float* obs_sgn, const int32_t* act_perm,
             const float* act_sgn) {
  Mirror m;
  std::memset(&m, 0, sizeof m);
  if (obs_perm || obs_sgn || act_perm || act_sgn) {
    if (!obs_perm || !obs_sgn || !act_perm || !act_sgn) return fail(-1, "the four mirror tables come together (or all null)");
    m = Mirror{obs_perm, obs_sgn, act_perm, act_sgn, batch};
  }
  return grad_impl(L, params, obs, act, old_logp, adv, ret, vpred, idx, batch, clip_param, use_clipped_value_loss, stats_out, grad_out,
                   stream, m);
}

int ssl_apply(ssl_learner* L, float* params, float* adam_m, float* adam_v, const float* lr, const float* step, float* grad,
              float grad_scale, float max_grad_norm, float adam_eps, void* stream) {
  if (!L || !grad) return fail(-1, "null argument");
  SSL_HIP(hipSetDevice(L->device));
  const int nblk = (int)((L->net.n_params + 255) / 256);
  hipLaunchKernelGGL(scale_sq_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, grad, (long long)L->net.n_params, grad_scale, L->sq);
  return apply_impl(L, params, adam_m, adam_v, lr, step, grad, nblk, max_grad_norm, adam_eps, stream);
}

/* gradient of the last ssl_step (after the slice reduction, before clipping): [n_params] device pointer (tests) */
const float* ssl_debug_grad(ssl_learner* L) { return L ? L->G : nullptr; }

}  // extern "C"
