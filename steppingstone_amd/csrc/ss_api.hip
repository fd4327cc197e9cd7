// ss_api.hip -- C ABI of libsteppingstone.so (include/steppingstone.h).  Host side only: owns the HBM-resident
// structure-of-arrays state and launches the gfx950 kernels on the caller's stream.  There is no CPU path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ss_kernels.hpp"

// compiled in ss_rollout3.hip (its own scheduling strategy, see there)
extern template __global__ void ss::rollout_kernel_helped<ss::ModelWalker3D, 3>(ss::Params, ss::StepIO);
extern template __global__ void ss::rollout_kernel_helped<ss::ModelMike, 3>(ss::Params, ss::StepIO);


namespace {

thread_local std::string g_err;
constexpr int kMaxPeerSlots = 4;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define SS_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (call);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return fail(SS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e));                      \
  } while (0)

}  // namespace

struct ss_env {
  ss::Params P;
  ss::Knobs hk;         // host mirror of the device-resident hook state (P.knobs)
  ss::Knobs* dk;
  int kind;
  int device;
  float* prob_shared;   // [121]
  float* prob_env;      // [121][npad] or null
  float* obs_rows;      // [n][60] scratch: current observation rows for create_temp_states
  int helpers;          // -1 auto, else 0 / 1 / 3 helper wavefronts (env SS_HELPERS)
  int helper_max_groups;  // auto: use the helper wavefront up to this many 32-env groups
  ss::PeerTable* peer_table;   // device copy of the peer-store table (ss_peer_connect), or null
  uint32_t* peer_counter;
  uint32_t* peer_error;
  uint32_t* my_flags[4];       // this rank's flag array [G] of every ring slot (fine-grained)
  int peer_count;
  int peer_slots;
};

namespace {

void window_prob(float* p, int c, bool ring) {
  int cnt = 0;
  for (int i = 0; i < SS_GRID; ++i)
    for (int j = 0; j < SS_GRID; ++j) {
      int di = std::abs(i - 5), dj = std::abs(j - 5), m = di > dj ? di : dj;
      bool in = ring ? (m == c) : (m <= c);
      p[i * SS_GRID + j] = in ? 1.f : 0.f;
      cnt += in;
    }
  for (int k = 0; k < SS_NCELL; ++k) p[k] = p[k] / (float)cnt;
}

// Host-synchronous hook update: a <= 500-byte hipMemcpy on the null stream.  When it returns the device copy is
// current, so every step enqueued afterwards (on any stream, or replayed from a hipGraph) sees it.
int push_knobs(ss_env* env) {
  SS_HIP(hipMemcpy(env->dk, &env->hk, sizeof(ss::Knobs), hipMemcpyHostToDevice));
  return SS_OK;
}

int set_window(ss_env* env, int level, bool ring) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  if (level < 0 || level > 5) return fail(SS_ERR_INVALID, "curriculum level must be in 0..5");
  float p[SS_NCELL];
  window_prob(p, level, ring);
  SS_HIP(hipSetDevice(env->device));
  SS_HIP(hipMemcpy(env->prob_shared, p, sizeof p, hipMemcpyHostToDevice));
  env->hk.curriculum = level;
  env->hk.prob = env->prob_shared;
  env->hk.per_env_prob = 0;
  return push_knobs(env);
}

int helpers_for(const ss_env* env, int groups) {
  // helper wavefronts (contact operators on the CU's other SIMDs) while the batch leaves SIMDs idle: three while every
  // 4-wavefront workgroup gets a CU to itself, one while two 2-wavefront workgroups fit a CU
  return env->helpers >= 0 ? env->helpers
                           : (groups <= env->helper_max_groups / 2 ? 3 : (groups <= env->helper_max_groups ? 1 : 0));
}

inline dim3 grid64(const ss_env* env) { return dim3(env->P.npad / ss::kWave); }

template <bool RANDOM>
int launch_step(ss_env* env, const ss::StepIO& io, hipStream_t st) {
  // two lanes per env: 32 envs per 64-lane wavefront.  ceil(n / 32) workgroups, NOT npad / 32: the arrays are padded to 64 envs, and
  // for n mod 64 in 1..32 the padding used to launch one workgroup without a single valid env (see emit_outputs: nvalid).
  const dim3 grid((env->P.n + ss::kEnvsPerWave - 1) / ss::kEnvsPerWave);
  SS_HIP(hipSetDevice(env->device));                   // the stream belongs to this device
  const int helpers = helpers_for(env, (int)grid.x);
  if (helpers == 3) {
    if (env->kind == SS_WALKER3D)
      hipLaunchKernelGGL((ss::step_kernel_helped<ss::ModelWalker3D, RANDOM, 3>), grid, dim3(4 * ss::kWave), 0, st, env->P, io);
    else
      hipLaunchKernelGGL((ss::step_kernel_helped<ss::ModelMike, RANDOM, 3>), grid, dim3(4 * ss::kWave), 0, st, env->P, io);
  } else if (helpers > 0) {
    if (env->kind == SS_WALKER3D)
      hipLaunchKernelGGL((ss::step_kernel_helped<ss::ModelWalker3D, RANDOM, 1>), grid, dim3(2 * ss::kWave), 0, st, env->P, io);
    else
      hipLaunchKernelGGL((ss::step_kernel_helped<ss::ModelMike, RANDOM, 1>), grid, dim3(2 * ss::kWave), 0, st, env->P, io);
  } else {
    if (env->kind == SS_WALKER3D)
      hipLaunchKernelGGL((ss::step_kernel<ss::ModelWalker3D, RANDOM>), grid, dim3(ss::kWave), 0, st, env->P, io);
    else
      hipLaunchKernelGGL((ss::step_kernel<ss::ModelMike, RANDOM>), grid, dim3(ss::kWave), 0, st, env->P, io);
  }
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// io.nsteps control steps in one launch, actions from the benchmark Philox stream
int launch_rollout(ss_env* env, const ss::StepIO& io, hipStream_t st) {
  const dim3 grid((env->P.n + ss::kEnvsPerWave - 1) / ss::kEnvsPerWave);
  SS_HIP(hipSetDevice(env->device));
  const int helpers = helpers_for(env, (int)grid.x);
  if (helpers == 3) {
    if (env->kind == SS_WALKER3D)
      hipLaunchKernelGGL((ss::rollout_kernel_helped<ss::ModelWalker3D, 3>), grid, dim3(4 * ss::kWave), 0, st, env->P, io);
    else
      hipLaunchKernelGGL((ss::rollout_kernel_helped<ss::ModelMike, 3>), grid, dim3(4 * ss::kWave), 0, st, env->P, io);
  } else if (helpers > 0) {
    if (env->kind == SS_WALKER3D)
      hipLaunchKernelGGL((ss::rollout_kernel_helped<ss::ModelWalker3D, 1>), grid, dim3(2 * ss::kWave), 0, st, env->P, io);
    else
      hipLaunchKernelGGL((ss::rollout_kernel_helped<ss::ModelMike, 1>), grid, dim3(2 * ss::kWave), 0, st, env->P, io);
  } else {
    if (env->kind == SS_WALKER3D)
      hipLaunchKernelGGL((ss::rollout_kernel<ss::ModelWalker3D>), grid, dim3(ss::kWave), 0, st, env->P, io);
    else
      hipLaunchKernelGGL((ss::rollout_kernel<ss::ModelMike>), grid, dim3(ss::kWave), 0, st, env->P, io);
  }
  SS_HIP(hipGetLastError());
  return SS_OK;
}

}  // namespace

extern "C" {

const char* ss_last_error(void) { return g_err.c_str(); }
int ss_version(void) { return SS_ABI_VERSION; }
int32_t ss_num_envs(const ss_env* env) { return env ? env->P.n : 0; }

int ss_create(ss_env** out, int kind, int32_t num_envs, int device, uint64_t seed, int64_t env_id_offset) {
  if (!out) return fail(SS_ERR_INVALID, "out is null");
  *out = nullptr;
  if (kind != SS_WALKER3D && kind != SS_MIKE) return fail(SS_ERR_INVALID, "unknown robot kind");
  if (num_envs <= 0) return fail(SS_ERR_INVALID, "num_envs must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SS_ERR_NO_DEVICE, "no HIP device visible: libsteppingstone has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(SS_ERR_INVALID, "device index out of range");
  SS_HIP(hipSetDevice(device));
  ss_env* env = new ss_env();
  std::memset(env, 0, sizeof *env);
  env->kind = kind;
  env->device = device;
  {   // SS_HELPERS=0|1|3 forces the number of helper wavefronts; default: as many as cannot cost throughput
    const char* h = std::getenv("SS_HELPERS");
    env->helpers = h ? std::atoi(h) : -1;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    env->helper_max_groups = 2 * cus;     // two 2-wavefront workgroups per CU = its four SIMDs
  }
  ss::Params& P = env->P;
  P.n = num_envs;
  P.npad = (num_envs + ss::kWave - 1) / ss::kWave * ss::kWave;
  P.seed_lo = (uint32_t)seed;
  P.seed_hi = (uint32_t)(seed >> 32);
  P.env_offset = (uint32_t)env_id_offset;
  P.id_mask = 0xFFFFFFFFu;
  env->hk.curriculum = 0;
  env->hk.power = 1.0f;
  env->hk.auto_reset = 1;
  const size_t np = (size_t)P.npad;
  hipError_t e1 = hipMalloc(&P.fstate, sizeof(float) * ss::NF * np);
  hipError_t e2 = hipMalloc(&P.istate, sizeof(int) * ss::NI * np);
  hipError_t e3 = hipMalloc(&P.terrain, sizeof(float) * 120 * np);
  hipError_t e4 = hipMalloc(&env->prob_shared, sizeof(float) * SS_NCELL);
  hipError_t e5 = hipMalloc(&env->dk, sizeof(ss::Knobs));
  P.knobs = env->dk;
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
    ss_destroy(env);
    return fail(SS_ERR_ALLOC, "hipMalloc failed for the environment state");
  }
  SS_HIP(hipMemset(P.fstate, 0, sizeof(float) * ss::NF * np));
  SS_HIP(hipMemset(P.istate, 0, sizeof(int) * ss::NI * np));
  SS_HIP(hipMemset(P.terrain, 0, sizeof(float) * 120 * np));
  int rc = set_window(env, 0, false);
  if (rc != SS_OK) { ss_destroy(env); return rc; }
  *out = env;
  return SS_OK;
}

void ss_destroy(ss_env* env) {
  if (!env) return;
  (void)hipSetDevice(env->device);
  if (env->P.fstate) (void)hipFree(env->P.fstate);
  if (env->P.istate) (void)hipFree(env->P.istate);
  if (env->P.terrain) (void)hipFree(env->P.terrain);
  if (env->prob_shared) (void)hipFree(env->prob_shared);
  if (env->dk) (void)hipFree(env->dk);
  if (env->peer_table) (void)hipFree(env->peer_table);
  if (env->peer_counter) (void)hipFree(env->peer_counter);
  if (env->peer_error) (void)hipFree(env->peer_error);
  if (env->prob_env) (void)hipFree(env->prob_env);
  if (env->obs_rows) (void)hipFree(env->obs_rows);
  if (env->P.prof) (void)hipFree(env->P.prof);
  delete env;
}

int ss_reset(ss_env* env, float* obs, void* stream) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  SS_HIP(hipSetDevice(env->device));
  hipStream_t st = (hipStream_t)stream;
  if (env->kind == SS_WALKER3D)
    hipLaunchKernelGGL((ss::reset_kernel<ss::ModelWalker3D>), grid64(env), dim3(ss::kWave), 0, st, env->P, obs);
  else
    hipLaunchKernelGGL((ss::reset_kernel<ss::ModelMike>), grid64(env), dim3(ss::kWave), 0, st, env->P, obs);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ss_step(ss_env* env, const float* act, float* obs, float* rew, uint8_t* done, ss_info* info, void* stream) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  if (!act || !obs || !rew || !done) return fail(SS_ERR_INVALID, "act/obs/rew/done must be device pointers");
  ss::StepIO io{act, obs, rew, done, info, 0, nullptr, 1, nullptr, 0, 0};
  return launch_step<false>(env, io, (hipStream_t)stream);
}

int ss_rollout_random(ss_env* env, int32_t num_steps, int32_t steps_per_launch, uint64_t t0, float* obs, float* rew,
                      uint8_t* done, ss_info* info, void* stream) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  if (!obs || !rew || !done) return fail(SS_ERR_INVALID, "obs/rew/done must be device pointers");
  if (num_steps < 0 || steps_per_launch < 0) return fail(SS_ERR_INVALID, "num_steps / steps_per_launch must be >= 0");
  const int32_t chunk = steps_per_launch > 0 ? steps_per_launch : 1000;     // K = 1000 steps per launch
  for (int32_t k = 0; k < num_steps; k += chunk) {
    const int32_t ns = num_steps - k < chunk ? num_steps - k : chunk;
    ss::StepIO io{nullptr, obs, rew, done, info, t0 + (uint64_t)k, nullptr, ns, nullptr, 0, 0};
    int rc = ns == 1 ? launch_step<true>(env, io, (hipStream_t)stream) : launch_rollout(env, io, (hipStream_t)stream);
    if (rc != SS_OK) return rc;
  }
  return SS_OK;
}

int ss_rollout_random_packed(ss_env* env, int32_t num_steps, uint64_t t0, float* packed, ss_info* info, void* stream) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  if (!packed) return fail(SS_ERR_INVALID, "packed must be a device pointer to [num_steps, N, 62] floats");
  if (num_steps < 1) return fail(SS_ERR_INVALID, "num_steps must be >= 1");
  ss::StepIO io{nullptr, nullptr, nullptr, nullptr, info, t0, packed, num_steps, nullptr, 0, (long long)env->P.n * (SS_OBS_DIM + 2)};
  return launch_rollout(env, io, (hipStream_t)stream);
}

int ss_step_packed(ss_env* env, const float* act, int use_random_actions, uint64_t t, float* packed, ss_info* info,
                   void* stream) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  if (!packed || (!act && !use_random_actions)) return fail(SS_ERR_INVALID, "packed (and act, unless random) must be set");
  ss::StepIO io{act, nullptr, nullptr, nullptr, info, t, packed, 1, nullptr, 0, 0};
  return use_random_actions ? launch_step<true>(env, io, (hipStream_t)stream) : launch_step<false>(env, io, (hipStream_t)stream);
}

// ---- peer-store all-gather (multi-GPU without a collective library in the data path)
int ss_peer_alloc(void** out, uint64_t bytes) {
  if (!out || bytes == 0) return fail(SS_ERR_INVALID, "bad argument");
  SS_HIP(hipExtMallocWithFlags(out, (size_t)bytes, hipDeviceMallocFinegrained));
  SS_HIP(hipMemset(*out, 0, (size_t)bytes));
  SS_HIP(hipDeviceSynchronize());
  return SS_OK;
}
int ss_peer_free(void* ptr) {
  if (ptr) SS_HIP(hipFree(ptr));
  return SS_OK;
}
int ss_peer_ipc_handle(void* ptr, void* handle64) {
  if (!ptr || !handle64) return fail(SS_ERR_INVALID, "null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "ipc handle size");
  SS_HIP(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr));
  return SS_OK;
}
int ss_peer_ipc_open(const void* handle64, void** out) {
  if (!handle64 || !out) return fail(SS_ERR_INVALID, "null argument");
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, sizeof h);
  SS_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
  return SS_OK;
}
int ss_peer_ipc_close(void* ptr) {
  if (ptr) SS_HIP(hipIpcCloseMemHandle(ptr));
  return SS_OK;
}

int ss_peer_connect(ss_env* env, int32_t count, int32_t rank, int32_t slots, float* const* gather_bufs, uint32_t* const* flag_bufs) {
  if (!env || !gather_bufs || !flag_bufs) return fail(SS_ERR_INVALID, "null argument");
  if (count < 1 || count > ss::kMaxPeers || rank < 0 || rank >= count) return fail(SS_ERR_INVALID, "need 1 <= count <= 8, 0 <= rank < count");
  if (slots < 1 || slots > kMaxPeerSlots) return fail(SS_ERR_INVALID, "need 1 <= slots <= 4");
  SS_HIP(hipSetDevice(env->device));
  if (!env->peer_counter) {
    SS_HIP(hipMalloc(&env->peer_counter, sizeof(uint32_t)));
    SS_HIP(hipMalloc(&env->peer_error, sizeof(uint32_t)));
    SS_HIP(hipMalloc(&env->peer_table, sizeof(ss::PeerTable) * kMaxPeerSlots));
  }
  SS_HIP(hipDeviceSynchronize());
  SS_HIP(hipMemset(env->peer_counter, 0, sizeof(uint32_t)));
  SS_HIP(hipMemset(env->peer_error, 0, sizeof(uint32_t)));
  ss::PeerTable t[kMaxPeerSlots];
  std::memset(t, 0, sizeof t);
  for (int s = 0; s < slots; ++s) {
    for (int p = 0; p < count; ++p) {
      if (!gather_bufs[s * count + p] || !flag_bufs[s * count + p]) return fail(SS_ERR_INVALID, "null peer buffer");
      t[s].dst[p] = gather_bufs[s * count + p];
      t[s].flag[p] = flag_bufs[s * count + p];
    }
    t[s].done_counter = env->peer_counter;
    t[s].count = count;
    t[s].rank = rank;
    t[s].n_local = env->P.n;
    env->my_flags[s] = flag_bufs[s * count + rank];
  }
  SS_HIP(hipMemcpy(env->peer_table, t, sizeof t, hipMemcpyHostToDevice));
  env->peer_count = count;
  env->peer_slots = slots;
  return SS_OK;
}

int ss_step_packed_peers(ss_env* env, const float* act, int use_random_actions, uint64_t t, int32_t slot, uint32_t step_id,
                         float* packed, ss_info* info, void* stream) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  if (!env->peer_table) return fail(SS_ERR_INVALID, "ss_peer_connect has not been called");
  if (slot < 0 || slot >= env->peer_slots) return fail(SS_ERR_INVALID, "slot out of range");
  if (!act && !use_random_actions) return fail(SS_ERR_INVALID, "act must be set unless random");
  ss::StepIO io{act, nullptr, nullptr, nullptr, info, t, packed, 1, env->peer_table + slot, step_id, 0};
  return use_random_actions ? launch_step<true>(env, io, (hipStream_t)stream) : launch_step<false>(env, io, (hipStream_t)stream);
}

int ss_peer_wait(ss_env* env, int32_t slot, uint32_t step_id, void* stream) {
  if (!env || !env->peer_table) return fail(SS_ERR_INVALID, "ss_peer_connect has not been called");
  if (slot < 0 || slot >= env->peer_slots) return fail(SS_ERR_INVALID, "slot out of range");
  SS_HIP(hipSetDevice(env->device));
  hipLaunchKernelGGL(ss::peer_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint32_t*)env->my_flags[slot],
                     env->peer_count, step_id, env->peer_error);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ss_peer_error(ss_env* env, uint32_t* out) {
  if (!env || !out || !env->peer_error) return fail(SS_ERR_INVALID, "bad argument");
  SS_HIP(hipSetDevice(env->device));
  SS_HIP(hipMemcpy(out, env->peer_error, sizeof(uint32_t), hipMemcpyDeviceToHost));
  return SS_OK;
}

int ss_random_actions(ss_env* env, uint64_t t, float* act, void* stream) {
  if (!env || !act) return fail(SS_ERR_INVALID, "null argument");
  SS_HIP(hipSetDevice(env->device));
  hipLaunchKernelGGL(ss::random_actions_kernel, dim3((env->P.n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     env->P, t, act);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ss_set_curriculum(ss_env* env, int32_t level) { return set_window(env, level, false); }
int ss_set_specialist(ss_env* env, int32_t level) { return set_window(env, level, true); }

int ss_set_sample_prob(ss_env* env, const double* prob, int per_env) {
  if (!env || !prob) return fail(SS_ERR_INVALID, "null argument");
  SS_HIP(hipSetDevice(env->device));
  if (!per_env) {
    float p[SS_NCELL];
    for (int k = 0; k < SS_NCELL; ++k) p[k] = (float)prob[k];
    SS_HIP(hipMemcpy(env->prob_shared, p, sizeof p, hipMemcpyHostToDevice));
    env->hk.prob = env->prob_shared;
    env->hk.per_env_prob = 0;
    return push_knobs(env);
  }
  const size_t np = (size_t)env->P.npad;
  if (!env->prob_env) SS_HIP(hipMalloc(&env->prob_env, sizeof(float) * SS_NCELL * np));
  std::vector<float> t(SS_NCELL * np, 0.f);
  for (int e = 0; e < env->P.n; ++e)
    for (int k = 0; k < SS_NCELL; ++k) t[(size_t)k * np + e] = (float)prob[(size_t)e * SS_NCELL + k];
  SS_HIP(hipMemcpy(env->prob_env, t.data(), sizeof(float) * t.size(), hipMemcpyHostToDevice));
  env->hk.prob = env->prob_env;
  env->hk.per_env_prob = 1;
  return push_knobs(env);
}

int ss_set_sample_prob_device(ss_env* env, const float* prob, int per_env, void* stream) {
  if (!env || !prob) return fail(SS_ERR_INVALID, "null argument");
  SS_HIP(hipSetDevice(env->device));
  hipStream_t st = (hipStream_t)stream;
  if (!per_env) {
    hipLaunchKernelGGL(ss::copy_prob_kernel, dim3(1), dim3(128), 0, st, prob, env->prob_shared);
    env->hk.prob = env->prob_shared;
    env->hk.per_env_prob = 0;
  } else {
    const size_t np = (size_t)env->P.npad;
    if (!env->prob_env) {
      SS_HIP(hipMalloc(&env->prob_env, sizeof(float) * SS_NCELL * np));      // first use only (allocation synchronises)
      SS_HIP(hipMemset(env->prob_env, 0, sizeof(float) * SS_NCELL * np));
    }
    const int total = env->P.n * SS_NCELL;
    hipLaunchKernelGGL(ss::transpose_prob_kernel, dim3((total + 255) / 256), dim3(256), 0, st, prob, env->prob_env, env->P.n,
                       env->P.npad);
    env->hk.prob = env->prob_env;
    env->hk.per_env_prob = 1;
  }
  // the pointer / flag switch travels on the same stream, behind the grid it refers to
  hipLaunchKernelGGL(ss::set_knobs_kernel, dim3(1), dim3(64), 0, st, env->dk, env->hk);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ss_set_mirror(ss_env* env, int32_t on) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  // ACCEPTED FOR PROTOCOL COMPATIBILITY ONLY: no state, no effect.  The reference forwards set_mirror to every env
  // (common/envs_utils.py:588-590; playground/train.py:109-111 under use_phase_mirror).  What an env does with it is in the
  // absent mocca_envs; the only consumer visible in the reference is the gait-phase-clocked Cassie stepper of train.py:37, whose
  // observation carries a phase variable.  The 60-float Walker3D / Mike observation (docs/PHYSICS.md 5) has no phase term, so there
  // is nothing for the flag to shift; the mirror symmetry itself is carried by ss_get_mirror_indices.
  (void)on;
  return SS_OK;
}

int ss_set_power(ss_env* env, float power) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  SS_HIP(hipSetDevice(env->device));
  env->hk.power = power;
  return push_knobs(env);
}

int ss_set_auto_reset(ss_env* env, int32_t on) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  SS_HIP(hipSetDevice(env->device));
  env->hk.auto_reset = on ? 1 : 0;
  return push_knobs(env);
}

int ss_create_temp_states(ss_env* env, float* out, void* stream) {
  if (!env || !out) return fail(SS_ERR_INVALID, "null argument");
  SS_HIP(hipSetDevice(env->device));
  if ((reinterpret_cast<uintptr_t>(out) & 15u) != 0) return fail(SS_ERR_INVALID, "out must be 16-byte aligned");
  if (!env->obs_rows) SS_HIP(hipMalloc(&env->obs_rows, sizeof(float) * SS_OBS_DIM * (size_t)env->P.npad));
  if (env->kind == SS_WALKER3D)
    hipLaunchKernelGGL((ss::obs_kernel<ss::ModelWalker3D>), grid64(env), dim3(ss::kWave), 0, (hipStream_t)stream, env->P, env->obs_rows);
  else
    hipLaunchKernelGGL((ss::obs_kernel<ss::ModelMike>), grid64(env), dim3(ss::kWave), 0, (hipStream_t)stream, env->P, env->obs_rows);
  hipLaunchKernelGGL(ss::temp_states_kernel, dim3(env->P.n), dim3(ss::kTempThreads), 0, (hipStream_t)stream, env->P,
                     (const float*)env->obs_rows, out);      // one workgroup per env
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ss_get_mirror_indices(int kind, int32_t* buf, int32_t* lens) {
  if (!buf || !lens) return fail(SS_ERR_INVALID, "null argument");
  (void)kind;   // both robots share the topology
  // In POLICY coordinates (docs/PHYSICS.md 2, ss::kPolicySign): the left limbs' x / z joints are measured about the mirrored
  // axis, so a mirror swaps the limbs without negating them and only the spine's z / x joints negate in place -- the lists
  // the reference's shipped actors are equivariant under (tools/checkpoint_layout_probe.py).  Generated from model.py.
  const auto& neg_j = ss::kMirrorNegate;
  const auto& right_j = ss::kMirrorRight;
  const auto& left_j = ss::kMirrorLeft;
  std::vector<int32_t> neg_obs = {2, 4}, right_obs, left_obs, neg_act, right_act, left_act;
  for (int j : neg_j) neg_obs.push_back(6 + j);
  for (int j : neg_j) neg_obs.push_back(27 + j);
  for (int i : {50, 53, 55, 58}) neg_obs.push_back(i);
  for (int j : right_j) right_obs.push_back(6 + j);
  for (int j : right_j) right_obs.push_back(27 + j);
  right_obs.push_back(48);
  for (int j : left_j) left_obs.push_back(6 + j);
  for (int j : left_j) left_obs.push_back(27 + j);
  left_obs.push_back(49);
  for (int j : neg_j) neg_act.push_back(j);
  for (int j : right_j) right_act.push_back(j);
  for (int j : left_j) left_act.push_back(j);
  const std::vector<int32_t>* lists[6] = {&neg_obs, &right_obs, &left_obs, &neg_act, &right_act, &left_act};
  int32_t* o = buf;
  for (int i = 0; i < 6; ++i) {
    lens[i] = (int32_t)lists[i]->size();
    for (int32_t v : *lists[i]) *o++ = v;
  }
  return SS_OK;
}

// PMC calibration helper: copies n floats in -> out (dword per lane, coalesced), device pointers
int ss_debug_calib_copy(const float* in, float* out, uint64_t n, void* stream) {
  if (!in || !out) return fail(SS_ERR_INVALID, "null argument");
  hipLaunchKernelGGL(ss::calib_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                     (size_t)n);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

// Self-check aid (tests/test_gpu_first_launch.py): envs e and e' with (e & mask) == (e' & mask) share their global id, i.e. their
// Philox streams (reset noise, stone draws, benchmark actions); given the same state they must produce the same bits in one launch.
int ss_debug_set_id_mask(ss_env* env, uint32_t mask) {
  if (!env) return fail(SS_ERR_INVALID, "null handle");
  env->P.id_mask = mask;
  return SS_OK;
}

// tuning aid: per-phase shader-clock totals (all zeros unless the library was built with -DSS_PROFILE_PHASES)
int ss_debug_phase_cycles(ss_env* env, unsigned long long* out16, int reset) {
  if (!env || !out16) return fail(SS_ERR_INVALID, "null argument");
  SS_HIP(hipSetDevice(env->device));
  if (!env->P.prof) {
    SS_HIP(hipMalloc(&env->P.prof, 16 * sizeof(unsigned long long)));
    SS_HIP(hipMemset(env->P.prof, 0, 16 * sizeof(unsigned long long)));
  }
  SS_HIP(hipDeviceSynchronize());
  SS_HIP(hipMemcpy(out16, env->P.prof, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) SS_HIP(hipMemset(env->P.prof, 0, 16 * sizeof(unsigned long long)));
  return SS_OK;
}

int ss_get_state(ss_env* env, float* packed, void* stream) {
  if (!env || !packed) return fail(SS_ERR_INVALID, "null argument");
  SS_HIP(hipSetDevice(env->device));
  hipLaunchKernelGGL(ss::pack_state_kernel, dim3((env->P.n + 63) / 64), dim3(64), 0, (hipStream_t)stream, env->P, packed);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ss_set_state(ss_env* env, const float* packed, void* stream) {
  if (!env || !packed) return fail(SS_ERR_INVALID, "null argument");
  SS_HIP(hipSetDevice(env->device));
  hipLaunchKernelGGL(ss::unpack_state_kernel, dim3((env->P.n + 63) / 64), dim3(64), 0, (hipStream_t)stream, env->P, packed);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

int ss_get_obs(ss_env* env, float* obs, void* stream) {
  if (!env || !obs) return fail(SS_ERR_INVALID, "null argument");
  SS_HIP(hipSetDevice(env->device));
  if (env->kind == SS_WALKER3D)
    hipLaunchKernelGGL((ss::obs_kernel<ss::ModelWalker3D>), grid64(env), dim3(ss::kWave), 0, (hipStream_t)stream, env->P, obs);
  else
    hipLaunchKernelGGL((ss::obs_kernel<ss::ModelMike>), grid64(env), dim3(ss::kWave), 0, (hipStream_t)stream, env->P, obs);
  SS_HIP(hipGetLastError());
  return SS_OK;
}

}  // extern "C"
