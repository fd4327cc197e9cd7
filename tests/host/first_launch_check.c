/* In-launch self-check of the HIP env kernels, meant to be run as a FRESH PROCESS many times (tests/test_gpu_first_launch.py;
 * DESIGN.md 5.1b): a plain-C client of include/steppingstone.h (dlopen) with the HIP runtime for the buffers -- no Python, no torch.
 *
 *   first_launch_check LIB KIND N INPUT MODE [REPEATS]
 *     LIB    path of libsteppingstone.so          KIND  0 Walker3D, 1 Mike          N  a power of two
 *     INPUT  float32 file: N x 186 packed states (ss_set_state layout) followed by N x 21 actions
 *     MODE   step: ss_step with the action array (the kernel a policy drives);  rollout: ss_rollout_random, 6 steps in one launch
 *
 * 2N environments are created; ss_debug_set_id_mask(N - 1) makes env e and env e + N share their global id (Philox streams), both
 * get the same state and action.  The FIRST launch of the step / rollout kernel in this process must then
 *   (a) produce bit-equal results for e and e + N (observation, reward, done, info words, full state afterwards), and
 *   (b) be reproduced bit for bit by REPEATS further launches from the same injected state.
 * Prints one line; exit code 0 = all equal, 1 = a mismatch (details on stdout), >= 2 = set-up error. */
#define __HIP_PLATFORM_AMD__ 1
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/steppingstone.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_SS(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s: rc %d (%s)\n", #x, r_, p_last_error()); return 3; } } while (0)

static const char* (*p_last_error)(void);

typedef struct {            /* everything one launch produced, host copies */
  float* obs; float* rew; unsigned char* done; uint32_t* info; float* state;
} result;

static int alloc_result(result* r, int n2) {
  r->obs = (float*)malloc(sizeof(float) * n2 * SS_OBS_DIM);
  r->rew = (float*)malloc(sizeof(float) * n2);
  r->done = (unsigned char*)malloc(n2);
  r->info = (uint32_t*)malloc(sizeof(uint32_t) * n2 * SS_INFO_WORDS);
  r->state = (float*)malloc(sizeof(float) * n2 * SS_STATE_DIM);
  return r->obs && r->rew && r->done && r->info && r->state;
}

/* number of differing 32-bit words between a[0..words) and b[0..words); the first few are printed */
static long diff_words(const char* what, const void* a, const void* b, long words, int per_env, long* printed) {
  const uint32_t* x = (const uint32_t*)a;
  const uint32_t* y = (const uint32_t*)b;
  long bad = 0;
  for (long i = 0; i < words; ++i)
    if (x[i] != y[i]) {
      if (*printed < 12) { printf("  %s env %ld word %ld: %08x vs %08x\n", what, i / per_env, i % per_env, x[i], y[i]); ++*printed; }
      ++bad;
    }
  return bad;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s LIB KIND N INPUT step|rollout [REPEATS]\n", argv[0]); return 2; }
  const int kind = atoi(argv[2]), n = atoi(argv[3]), n2 = 2 * n;
  const int rollout = strcmp(argv[5], "rollout") == 0;
  const int repeats = argc > 6 ? atoi(argv[6]) : 3;
  if (n <= 0 || (n & (n - 1)) != 0) { fprintf(stderr, "N must be a power of two\n"); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
  SYM(ss_create) SYM(ss_destroy) SYM(ss_last_error) SYM(ss_set_curriculum) SYM(ss_debug_set_id_mask) SYM(ss_set_state)
  SYM(ss_get_state) SYM(ss_step) SYM(ss_rollout_random)
  p_last_error = p_ss_last_error;

  /* inputs, duplicated: env e + N = env e */
  float* st = (float*)malloc(sizeof(float) * n2 * SS_STATE_DIM);
  float* act = (float*)malloc(sizeof(float) * n2 * SS_ACT_DIM);
  FILE* f = fopen(argv[4], "rb");
  if (!f || !st || !act) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
  if (fread(st, sizeof(float), (size_t)n * SS_STATE_DIM, f) != (size_t)n * SS_STATE_DIM ||
      fread(act, sizeof(float), (size_t)n * SS_ACT_DIM, f) != (size_t)n * SS_ACT_DIM) { fprintf(stderr, "short input file\n"); return 2; }
  fclose(f);
  memcpy(st + (size_t)n * SS_STATE_DIM, st, sizeof(float) * n * SS_STATE_DIM);
  memcpy(act + (size_t)n * SS_ACT_DIM, act, sizeof(float) * n * SS_ACT_DIM);

  ss_env* env = NULL;
  CHECK_SS(p_ss_create(&env, kind, n2, 0, 2u, 0));
  CHECK_SS(p_ss_set_curriculum(env, 5));
  CHECK_SS(p_ss_debug_set_id_mask(env, (uint32_t)(n - 1)));
  float *d_st, *d_act, *d_obs, *d_rew, *d_state;
  unsigned char* d_done;
  ss_info* d_info;
  CHECK_HIP(hipMalloc((void**)&d_st, sizeof(float) * n2 * SS_STATE_DIM));
  CHECK_HIP(hipMalloc((void**)&d_state, sizeof(float) * n2 * SS_STATE_DIM));
  CHECK_HIP(hipMalloc((void**)&d_act, sizeof(float) * n2 * SS_ACT_DIM));
  CHECK_HIP(hipMalloc((void**)&d_obs, sizeof(float) * n2 * SS_OBS_DIM));
  CHECK_HIP(hipMalloc((void**)&d_rew, sizeof(float) * n2));
  CHECK_HIP(hipMalloc((void**)&d_done, n2));
  CHECK_HIP(hipMalloc((void**)&d_info, sizeof(ss_info) * n2));
  CHECK_HIP(hipMemcpy(d_st, st, sizeof(float) * n2 * SS_STATE_DIM, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_act, act, sizeof(float) * n2 * SS_ACT_DIM, hipMemcpyHostToDevice));

  result first, again;
  if (!alloc_result(&first, n2) || !alloc_result(&again, n2)) return 2;
  long dup_bad = 0, rep_bad = 0, printed = 0;
  for (int rep = 0; rep <= repeats; ++rep) {
    result* r = rep == 0 ? &first : &again;
    CHECK_HIP(hipMemset(d_obs, 0xA5, sizeof(float) * n2 * SS_OBS_DIM));      /* a launch that skipped a row would show */
    CHECK_SS(p_ss_set_state(env, d_st, NULL));
    if (rollout) CHECK_SS(p_ss_rollout_random(env, 6, 6, 40u, d_obs, d_rew, d_done, d_info, NULL));
    else CHECK_SS(p_ss_step(env, d_act, d_obs, d_rew, d_done, d_info, NULL));
    CHECK_SS(p_ss_get_state(env, d_state, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(r->obs, d_obs, sizeof(float) * n2 * SS_OBS_DIM, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(r->rew, d_rew, sizeof(float) * n2, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(r->done, d_done, n2, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(r->info, d_info, sizeof(ss_info) * n2, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(r->state, d_state, sizeof(float) * n2 * SS_STATE_DIM, hipMemcpyDeviceToHost));
    if (rep == 0) {         /* (a) the two copies of every env inside the first launch */
      dup_bad += diff_words("first launch, obs: copy 0 vs copy 1,", first.obs, first.obs + (size_t)n * SS_OBS_DIM, (long)n * SS_OBS_DIM, SS_OBS_DIM, &printed);
      dup_bad += diff_words("first launch, rew: copy 0 vs copy 1,", first.rew, first.rew + n, n, 1, &printed);
      dup_bad += memcmp(first.done, first.done + n, n) != 0;
      dup_bad += diff_words("first launch, info: copy 0 vs copy 1,", first.info, first.info + (size_t)n * SS_INFO_WORDS, (long)n * SS_INFO_WORDS, SS_INFO_WORDS, &printed);
      dup_bad += diff_words("first launch, state: copy 0 vs copy 1,", first.state, first.state + (size_t)n * SS_STATE_DIM, (long)n * SS_STATE_DIM, SS_STATE_DIM, &printed);
    } else {                /* (b) later launches against the first */
      rep_bad += diff_words("repeat vs first launch, obs:", again.obs, first.obs, (long)n2 * SS_OBS_DIM, SS_OBS_DIM, &printed);
      rep_bad += diff_words("repeat vs first launch, rew:", again.rew, first.rew, n2, 1, &printed);
      rep_bad += memcmp(again.done, first.done, n2) != 0;
      rep_bad += diff_words("repeat vs first launch, info:", again.info, first.info, (long)n2 * SS_INFO_WORDS, SS_INFO_WORDS, &printed);
      rep_bad += diff_words("repeat vs first launch, state:", again.state, first.state, (long)n2 * SS_STATE_DIM, SS_STATE_DIM, &printed);
    }
  }
  int ndone = 0;
  for (int e = 0; e < n2; ++e) ndone += first.done[e] != 0;
  printf("first_launch_check kind %d envs 2x%d mode %s repeats %d: duplicate words differing %ld, repeat words differing %ld (done %d)\n",
         kind, n, rollout ? "rollout" : "step", repeats, dup_bad, rep_bad, ndone);
  p_ss_destroy(env);
  return (dup_bad || rep_bad) ? 1 : 0;
}
