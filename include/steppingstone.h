/*
 * steppingstone.h -- C ABI of libsteppingstone.so: the MI355X-native vectorised stepping-stone environment.
 *
 * The reference has no FFI for this path: its boundary is the Python object protocol between
 * common/envs_utils.py (ShmemVecEnv / _subproc_worker) and the gym env in the un-vendored mocca_envs submodule.
 * Every entry point below therefore cites the reference call site(s) whose job it takes over.  A reference
 * maintainer binds these with ctypes (INTEGRATION.md shows the stub); no torch types cross this boundary.
 *
 * Conventions: all functions return 0 on success or a negative ss_status; ss_last_error() returns a thread-local
 * message.  The caller owns every I/O buffer (device pointers, e.g. tensor.data_ptr()); the library owns the
 * structure-of-arrays environment state in HBM.  Every entry point that takes a stream (hipStream_t as void*;
 * NULL = the null stream) only enqueues work on it and never synchronises the host.  The hooks WITHOUT a stream
 * argument (ss_set_curriculum / _specialist / _sample_prob / _power / _auto_reset) are host-synchronous: they
 * copy <= 500 bytes (the [N,11,11] per-env grid of ss_set_sample_prob: N x 484 B) to the device-resident hook
 * state before returning, so every step enqueued afterwards -- on any stream, or replayed from a captured
 * hipGraph -- sees the new value; a step still in flight when a hook is called may see either (as in the
 * reference, hooks belong between step_wait() and the next step_async()).  ss_set_sample_prob_device is the
 * stream-ordered, sync-free form for grids that are computed on the GPU.  One handle per (process, device); a
 * handle is not thread-safe.
 */
#ifndef STEPPINGSTONE_H
#define STEPPINGSTONE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_OBS_DIM 60     /* shipped checkpoints: actor.state_dim */
#define SS_ACT_DIM 21     /* common/render_utils.py:47-69 */
#define SS_GRID 11        /* playground/train.py:132-133 (11x11 yaw x pitch grid, centre index 5) */
#define SS_NCELL 121
#define SS_NUM_STONES 20
#define SS_STATE_DIM 186  /* packed per-env state of ss_get_state / ss_set_state */
#define SS_INFO_WORDS 6    /* 32-bit words of ss_info */
#define SS_ABI_VERSION 4   /* ss_version(): 2 -> 3 added ss_info.ep_ret_lo and state word 185 (round 3); 3 -> 4 (round 4) changed a
                            * MEANING, no layout: actions and observations carry POLICY coordinates, see "Joint conventions" */
#define SS_MAX_EPISODE_STEPS 1000

typedef enum { SS_WALKER3D = 0, SS_MIKE = 1 } ss_kind;   /* ids: README.md:27,31 of the reference */

typedef enum {
  SS_OK = 0,
  SS_ERR_INVALID = -1,   /* bad argument */
  SS_ERR_HIP = -2,       /* a HIP runtime call failed (message has hipGetErrorString) */
  SS_ERR_NO_DEVICE = -3, /* no gfx950 device visible: there is NO CPU fallback */
  SS_ERR_ALLOC = -4
} ss_status;

/* per-env step report; replaces the `info` dict built by Monitor.update (common/envs_utils.py:131-153) and
 * TimeLimitMask.step (:59-65) plus env.update_terrain (playground/train.py:245). ep_* valid when done. */
typedef struct {
  float ep_ret;            /* info["episode"]["r"], leading part (the fp32 nearest to the episode return) */
  float ep_len;            /* info["episode"]["l"] */
  int32_t bad_transition;  /* info["bad_transition"] */
  int32_t steps_reached;   /* next_step_index at the end of the step */
  int32_t update_terrain;  /* env.update_terrain */
  /* Monitor.update sums the step rewards as Python floats and reports round(sum, 6) (common/envs_utils.py:131-138).  The
   * kernel keeps the running sum as an unevaluated pair of floats (error-free two-sum per step):
   * (double)ep_ret + (double)ep_ret_lo is the fp64 sum of the fp32 step rewards to ~1e-11; the host rounds it to 6 decimals. */
  float ep_ret_lo;
} ss_info;

typedef struct ss_env ss_env;

/* gym.make(env_id) + env.seed(seed + rank) for num_envs envs at once (common/envs_utils.py:25-40,48-56).
 * env_id_offset: global index of this handle's first env (multi-GPU sharding keeps RNG streams rank-invariant). */
int ss_create(ss_env** out, int kind, int32_t num_envs, int device, uint64_t seed, int64_t env_id_offset);
void ss_destroy(ss_env* env);                                   /* env.close(), envs_utils.py:667-676 */
const char* ss_last_error(void);

/* Joint conventions.  Joint / action order: common/render_utils.py:47-69 (abdomen z,y,x; right hip x,z,y, knee, ankle; left ...;
 * right shoulder x,z,y, elbow; left ...).  POLICY coordinates -- act[j], obs[6+j] (normalised angle), obs[27+j] (0.1 * rate):
 * sigma_j x (value about the +axis of the link frame), docs/PHYSICS.md 2, sigma = model.POLICY_SIGN: -1 for the left limbs' x / z
 * joints (measured about the MIRRORED axis, so that ss_get_mirror_indices swaps the limbs without negating them) and for both knees
 * (negative in flexion), +1 elsewhere: the conventions the reference's shipped actors were trained in (playground/models/ *.pt are
 * mirror-equivariant under exactly these lists and reject the other knee sign, tests/test_shipped_policy_layout.py).
 * ss_get_state / ss_set_state keep every angle about the +axis.
 *
 * ShmemVecEnv.reset (envs_utils.py:542-548): obs [num_envs, 60] f32 row-major, device pointer. */
int ss_reset(ss_env* env, float* obs, void* stream);

/* ShmemVecEnv.step_async + step_wait + worker auto-reset (envs_utils.py:550-558, 646-649).
 * act [N,21] f32; obs [N,60] f32; rew [N] f32; done [N] u8; info [N] ss_info (may be NULL).  Device pointers. */
int ss_step(ss_env* env, const float* act, float* obs, float* rew, uint8_t* done, ss_info* info, void* stream);

/* Benchmark path (BASELINE.json metric "batched random-action rollout"): num_steps control steps with
 * on-device Philox actions U(-1,1); t0 = index of the first step in the action stream.  steps_per_launch control
 * steps run inside ONE kernel launch with the state resident in LDS between them (0 = default 1000; 1 = one launch
 * per step, the ss_step code path).  Every step writes obs / rew / done / info (the buffers hold the last step's
 * values afterwards) and the HBM copy of the state; the result is bit-identical for every steps_per_launch. */
int ss_rollout_random(ss_env* env, int32_t num_steps, int32_t steps_per_launch, uint64_t t0, float* obs, float* rew,
                      uint8_t* done, ss_info* info, void* stream);
/* The same multi-step launch writing every step's packed block (see ss_step_packed) at packed[k] of a [num_steps, N, 62]
 * f32 device buffer: the rollout of a multi-GPU shard whose exchange ships num_steps blocks per collective instead of one
 * (steppingstone_amd.distributed.ShardedVecEnv.rollout_random: same bytes on the wire, K times fewer collectives and
 * launches).  One launch; num_steps >= 1. */
int ss_rollout_random_packed(ss_env* env, int32_t num_steps, uint64_t t0, float* packed, ss_info* info, void* stream);
/* One step whose results land in ONE packed device buffer [N,62] f32 = obs(60) | rew | done(0/1): the block a
 * multi-GPU shard all-gathers per step (replaces the per-env pipe + shared-memory traffic of
 * common/envs_utils.py:550-558,608-620).  use_random_actions != 0: actions from the benchmark Philox stream at index t. */
int ss_step_packed(ss_env* env, const float* act, int use_random_actions, uint64_t t, float* packed, ss_info* info,
                   void* stream);
/* Peer-store all-gather: the multi-GPU exchange of SURVEY.md 8e WITHOUT a collective in the data path.  Every rank owns
 * a gather buffer [G * N_local, 62] f32 and a flag array [G] u32 in fine-grained device memory (ss_peer_alloc; shared
 * with the other ranks' processes through ss_peer_ipc_handle / ss_peer_ipc_open, 64-byte handles), one pair per ring
 * slot.  ss_peer_connect gives the handle the gather-buffer and flag-array pointers of all G ranks as seen from THIS
 * process (index = slot * G + rank; own entries included).
 * ss_step_packed_peers is ss_step_packed whose kernel additionally stores its rows straight into every peer's gather
 * buffer (xGMI peer-to-peer stores) and, from the last workgroup, publishes step_id in flag[rank] of every peer.
 * ss_peer_wait enqueues a one-wavefront kernel that returns once all G peers have published step_id (or later) here:
 * work enqueued behind it may read this rank's gather buffer.  step_id must increase from launch to launch; a buffer
 * may be overwritten only after every peer has consumed it (use >= 2 buffers in turn).  ss_peer_error: non-zero if a
 * wait timed out.  (packed may be NULL here: the gather buffer already holds this rank's own rows.) */
int ss_peer_alloc(void** out, uint64_t bytes);
int ss_peer_free(void* ptr);
int ss_peer_ipc_handle(void* ptr, void* handle64);
int ss_peer_ipc_open(const void* handle64, void** out);
int ss_peer_ipc_close(void* ptr);
int ss_peer_connect(ss_env* env, int32_t count, int32_t rank, int32_t slots, float* const* gather_bufs,
                    uint32_t* const* flag_bufs);          /* pointer tables [slots][count]; slots <= 4 buffers used in turn */
int ss_step_packed_peers(ss_env* env, const float* act, int use_random_actions, uint64_t t, int32_t slot, uint32_t step_id,
                         float* packed, ss_info* info, void* stream);
int ss_peer_wait(ss_env* env, int32_t slot, uint32_t step_id, void* stream);
int ss_peer_error(ss_env* env, uint32_t* out);
/* Same action stream written to act [N,21] (parity tests / external policies). */
int ss_random_actions(ss_env* env, uint64_t t, float* act, void* stream);

/* Curriculum hooks (envs_utils.py:568-590, 650-664; playground/train.py:118,122,271). */
int ss_set_curriculum(ss_env* env, int32_t level);              /* env.update_curriculum */
int ss_set_specialist(ss_env* env, int32_t level);              /* env.update_specialist */
/* env.update_sample_prob: HOST pointer to f64 probabilities, [N,11,11] if per_env else [11,11]. */
int ss_set_sample_prob(ss_env* env, const double* prob, int per_env);
/* Same hook for a grid that already lives on the GPU (the batched threshold / adaptive sampler computes it there,
 * playground/train.py:229-272): DEVICE pointer to f32 probabilities, [N,11,11] if per_env else [11,11]; the copy /
 * transpose and the switch to the new grid are ordered on `stream`, the host is not synchronised (the first per-env
 * call allocates the [121][N] table). */
int ss_set_sample_prob_device(ss_env* env, const float* prob, int per_env, void* stream);
/* env.set_mirror (common/envs_utils.py:588-590, train.py:109-111).  Accepted for protocol compatibility ONLY -- it stores
 * nothing and changes nothing: the Walker3D / Mike observation has no gait-phase term for the flag to act on (the reference
 * uses it with phase-clocked envs); the symmetry a learner needs is ss_get_mirror_indices. */
int ss_set_mirror(ss_env* env, int32_t on);
int ss_set_power(ss_env* env, float power);                     /* env.set_robot_params({"power": p}) */
/* on (default): worker semantics, a finished env is reset inside the step (envs_utils.py:647-648);
 * off: plain gym env semantics for make_env() users -- terminal obs returned, caller resets (train.py:243-244). */
int ss_set_auto_reset(ss_env* env, int32_t on);

/* env.create_temp_states (train.py:247, envs_utils.py:573-578): out [N,121,60] f32, device pointer. */
int ss_create_temp_states(ss_env* env, float* out, void* stream);

/* env.unwrapped.get_mirror_indices() (train.py:160; consumed by get_mirror_function, envs_utils.py:687-694).
 * buf receives the 6 index lists back to back, lens[6] their lengths; buf must hold 2*(60+21) int32.
 * Lists (policy coordinates): negated obs {vy, roll, abdomen z / x angle and rate, sin(dtheta) d and x_tilt of both targets},
 * negated act {abdomen z, x}; right <-> left: the 9 limb joints (angle, rate, action) and the two contact flags. */
int ss_get_mirror_indices(int kind, int32_t* buf, int32_t* lens);

/* Full-state injection / extraction for parity tests and terrain_info (enjoy.py:60-64).  DEVICE pointers,
 * [N, SS_STATE_DIM] f32 row-major:
 *   0:3 pos | 3:7 quat wxyz | 7:13 base twist (body frame, angular first) | 13:34 q | 34:55 qd |
 *   55 pot_prev | 56 z_init | 57 ep_ret | 58 next-next dr | 59 next_step_index | 60 target_reached_count |
 *   61 elapsed | 62 rng_ctr & 0xffff | 63 rng_ctr >> 16 | 64 flags (bit0 right, bit1 left foot contact) |
 *   65:185 terrain_info [20][6] = x,y,z,phi,x_tilt,y_tilt | 185 ep_ret_lo (trailing part of the episode return) */
int ss_get_state(ss_env* env, float* packed, void* stream);
int ss_set_state(ss_env* env, const float* packed, void* stream);
/* Observation of the current state without stepping (used after ss_set_state). */
int ss_get_obs(ss_env* env, float* obs, void* stream);

int32_t ss_num_envs(const ss_env* env);
/* SS_ABI_VERSION of the library.  A binding must check it at load time (steppingstone_amd/_lib.py does): version 2 inserted
 * steps_per_launch into ss_rollout_random's argument list, version 3 grew ss_info to 6 words and the packed state to 186,
 * version 4 changed the sign convention of the left limbs' x / z joints and of the knees in actions and observations (same layouts). */
int ss_version(void);

/* Measurement aids (tools/hbm_traffic.py, tools/phase_profile.py); not part of the env protocol.
 * ss_debug_calib_copy: dword-per-lane copy out[i] = in[i] + 1 used to calibrate the HBM PMC counters.
 * ss_debug_phase_cycles: 16 per-phase shader-clock totals (zeros unless built with -DSS_PROFILE_PHASES). */
int ss_debug_calib_copy(const float* in, float* out, uint64_t n, void* stream);
int ss_debug_phase_cycles(ss_env* env, unsigned long long* out16, int reset);
/* Self-check aid: the global id of env e becomes env_id_offset + (e & mask) (default mask: all ones), so that envs e and
 * e + 2^k share their Philox streams; injected with the same state they must come out bit-equal from one launch
 * (tests/test_gpu_first_launch.py, tests/host/first_launch_check.c). */
int ss_debug_set_id_mask(ss_env* env, uint32_t mask);

#ifdef __cplusplus
}
#endif
#endif
