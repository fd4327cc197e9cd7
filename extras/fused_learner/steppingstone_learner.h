/*
 * steppingstone_learner.h -- C ABI of the fused PPO minibatch step (libsslearner.so, hand-written gfx950 kernels).
 *
 * the reference's learner is algorithms/ppo.py:40-108 (PPO.update) over the networks of
 * common/controller.py:55-145 (Policy: tanh-mean diagonal Gaussian with a state-independent log-std + critic ensemble)
 * and :217-261 (SoftsignActor).  One ssl_step call is one iteration of the minibatch loop body of ppo.py:55-100:
 * evaluate_actions, the clipped surrogate and value losses, backward, clip_grad_norm_ (max_grad_norm) and the Adam step
 * (lr, eps as in playground/train.py:72-82) -- for the fixed architecture 60 -> 256 x5 -> 21 (softsign x3, relu x2,
 * tanh) and n_ens critics 60 -> 256 x4 -> 1 (relu).  No torch types cross this boundary: plain device pointers, the
 * stream as void*; nothing synchronises the host, so the call can be captured in a hipGraph.
 *
 * Parameters live in ONE flat f32 device vector of ssl_num_params(n_ens) floats: log-std [21], then the actor's six
 * Linear layers, then every critic's five -- each as weight [out][in] row-major followed by bias [out], every tensor
 * starting at a multiple of 4 floats (steppingstone_amd/fused_ppo.py computes the same offsets and keeps the
 * nn.Parameters as views into the vector).  adam_m / adam_v: same size, zero-initialised by the caller.
 */
#ifndef STEPPINGSTONE_LEARNER_H
#define STEPPINGSTONE_LEARNER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssl_learner ssl_learner;

const char* ssl_last_error(void);
int64_t ssl_num_params(int32_t n_ens);
/* workspace for minibatches of up to max_batch rows (a multiple of 32) */
int ssl_create(ssl_learner** out, int device, int32_t n_ens, int32_t max_batch);
void ssl_destroy(ssl_learner* L);

/* One minibatch step.  obs [R][60], act [R][21], old_logp / adv / ret / vpred [R] are the flattened rollout arrays of an
 * update (RolloutStorage, algorithms/storage.py); idx [batch] (int64) selects this minibatch's rows (the sampler of
 * storage.py:84-117).  lr and step are DEVICE scalars (f32): learning rate and the Adam step count of THIS update (1, 2,
 * ...), so that a captured graph can be replayed with new values.  stats_out (device, 3 floats or NULL) receives value
 * loss, action loss, entropy as ppo.py:96-100 accumulates them.  vpred may be NULL unless use_clipped_value_loss. */
int ssl_step(ssl_learner* L, float* params, float* adam_m, float* adam_v, const float* lr, const float* step, const float* obs,
             const float* act, const float* old_logp, const float* adv, const float* ret, const float* vpred, const int64_t* idx,
             int32_t batch, float clip_param, float max_grad_norm, float adam_eps, int32_t use_clipped_value_loss, float* stats_out,
             void* stream);

/* The same step with the left-right mirror augmentation of PPO.update (algorithms/ppo.py:57-58 with the mirror_function of
 * common/envs_utils.py:687-740): the minibatch is doubled, row batch + i being the mirror image of row i.  obs_perm [60] /
 * act_perm [21] (int32) and obs_sgn [60] / act_sgn [21] (f32) are DEVICE tables: column c of a mirrored row is
 * sgn[src] * x[src] with src = perm[c] (negate the lateral quantities, then swap right and left limbs).  2 * batch must
 * fit the workspace (max_batch of ssl_create). */
int ssl_step_mirror(ssl_learner* L, float* params, float* adam_m, float* adam_v, const float* lr, const float* step, const float* obs,
                    const float* act, const float* old_logp, const float* adv, const float* ret, const float* vpred,
                    const int64_t* idx, int32_t batch, float clip_param, float max_grad_norm, float adam_eps,
                    int32_t use_clipped_value_loss, float* stats_out, void* stream, const int32_t* obs_perm, const float* obs_sgn,
                    const int32_t* act_perm, const float* act_sgn);

/* Data-parallel learner (configs[4]: one process per GPU, algorithms/ppo.py's update over every rank's rollout): the step in
 * two halves with the caller's gradient all-reduce between them.
 *   ssl_grad : forward, losses, backward and the slice reduction of ONE rank's minibatch; the gradient of its mean loss is
 *              written to grad_out (device, ssl_num_params floats).  Mirror tables as in ssl_step_mirror, or four NULLs.
 *   (caller) : one all-reduce (sum) of grad_out over the ranks -- RCCL through torch.distributed in fused_ppo.py.
 *   ssl_apply: grad *= grad_scale (1 / world size: the mean over ranks = the gradient of the mean loss over the global
 *              minibatch), clip by the global norm, Adam.  Every rank applies the same step to the same weights.
 * ssl_grad + ssl_apply(grad_scale = 1) on one rank is ssl_step (the norm's summation order differs: ~1e-7 relative). */
int ssl_grad(ssl_learner* L, const float* params, const float* obs, const float* act, const float* old_logp, const float* adv,
             const float* ret, const float* vpred, const int64_t* idx, int32_t batch, float clip_param, int32_t use_clipped_value_loss,
             float* stats_out, float* grad_out, void* stream, const int32_t* obs_perm, const float* obs_sgn, const int32_t* act_perm,
             const float* act_sgn);
int ssl_apply(ssl_learner* L, float* params, float* adam_m, float* adam_v, const float* lr, const float* step, float* grad,
              float grad_scale, float max_grad_norm, float adam_eps, void* stream);

/* gradient of the last ssl_step (slices reduced, before clipping): device pointer to ssl_num_params floats (tests) */
const float* ssl_debug_grad(ssl_learner* L);

#ifdef __cplusplus
}
#endif
#endif
