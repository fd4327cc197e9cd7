// host_harness.cpp -- compiles the SAME kernel source (steppingstone_amd/csrc/*.hpp) for the CPU so that the
// per-lane device code can be checked against the oracle in the GPU-less build container (pre-flight / debugging).
// TEST INFRASTRUCTURE ONLY: it is built by tests/host_lib.py into tests/host/, never shipped, and the product
// library contains no host path.  Build: hipcc --cuda-host-only -x hip ... (see tests/host_lib.py).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../steppingstone_amd/csrc/ss_kernels.hpp"

// ---- a wavefront on the host: its 64 lanes run as 64 threads around one LDS block.  The two lanes of an env meet at
// every lane-pair exchange; all 64 meet where the device code relies on the wavefront's lockstep (SS_WAVE_SYNC: between
// the staging of the output rows and their block copy in emit_outputs).
namespace {
struct Barrier {
  const int parties;
  std::atomic<int> arrived{0};
  std::atomic<int> generation{0};
  explicit Barrier(int n) : parties(n) {}
  void wait() {
    int g = generation.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) == parties - 1) {
      arrived.store(0, std::memory_order_relaxed);
      generation.store(g + 1, std::memory_order_release);
    } else {
      while (generation.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
  }
};
struct PairSync {
  Barrier b{2};
  float slot[2];
};
thread_local PairSync* t_sync = nullptr;
thread_local Barrier* t_wave = nullptr;
thread_local int t_side = 0;
}  // namespace

float ss_host_xchg(float x) {
  PairSync* s = t_sync;
  s->slot[t_side] = x;
  s->b.wait();
  float r = s->slot[1 - t_side];
  s->b.wait();
  return r;
}
void ss_host_wave_sync() { t_wave->wait(); }

namespace {
void window_prob(float* p, int c) {
  int cnt = 0;
  for (int i = 0; i < 11; ++i)
    for (int j = 0; j < 11; ++j) {
      int di = std::abs(i - 5), dj = std::abs(j - 5), m = di > dj ? di : dj;
      p[i * 11 + j] = (m <= c) ? 1.f : 0.f;
      cnt += (m <= c);
    }
  for (int k = 0; k < 121; ++k) p[k] /= (float)cnt;
}
}  // namespace

extern "C" {

// one control step of n envs: packed state in/out [n,186], act [n,21]; outputs obs [n,60], rew, done, info
int hh_step(int kind, int n, unsigned long long seed, int curriculum, const double* prob /*121 or null*/,
            const float* packed_in, const float* act, float* packed_out, float* obs, float* rew,
            unsigned char* done, ss_info* info) {
  ss::Params P;
  std::memset(&P, 0, sizeof P);
  P.n = n;
  P.npad = (n + 63) / 64 * 64;
  std::vector<float> f((size_t)ss::NF * P.npad, 0.f), terr((size_t)120 * P.npad, 0.f), pr(121);
  std::vector<int> is((size_t)ss::NI * P.npad, 0);
  window_prob(pr.data(), curriculum);
  if (prob) for (int k = 0; k < 121; ++k) pr[k] = (float)prob[k];
  ss::Knobs K;
  K.prob = pr.data(); K.per_env_prob = 0; K.curriculum = curriculum; K.power = 1.f; K.auto_reset = 1;
  P.fstate = f.data(); P.istate = is.data(); P.terrain = terr.data(); P.knobs = &K;
  P.seed_lo = (uint32_t)seed; P.seed_hi = (uint32_t)(seed >> 32); P.env_offset = 0; P.id_mask = 0xFFFFFFFFu;
  ss::StepIO io{act, obs, rew, done, info, 0, nullptr, 1, nullptr, 0, 0};
  for (int e = 0; e < n; ++e) ss::unpack_env(P, e, packed_in);
  const int waves = (2 * n + 63) / 64;
  for (int w = 0; w < waves; ++w) {                      // one wavefront = 32 envs = 64 lane threads, as on the device
    std::vector<float4> lds((size_t)ss::kLdsSlots * 64);
    Barrier wave(64);
    std::vector<PairSync> pairs(32);
    auto run = [&](int lane) {
      t_sync = &pairs[lane >> 1];
      t_wave = &wave;
      t_side = lane & 1;
      const int lg = 64 * w + lane;                      // lanes past the last env redo it with valid = false, like the device
      if (kind == 0) ss::step_env<ss::ModelWalker3D, false>(P, io, lg, lane, reinterpret_cast<float*>(lds.data()));
      else ss::step_env<ss::ModelMike, false>(P, io, lg, lane, reinterpret_cast<float*>(lds.data()));
    };
    std::vector<std::thread> th;
    for (int lane = 1; lane < 64; ++lane) th.emplace_back(run, lane);
    run(0);
    for (auto& t : th) t.join();
  }
  for (int e = 0; e < n; ++e) ss::pack_env(P, e, packed_out);
  return 0;
}

}  // extern "C"
