/*
 * ss_oracle.c -- CPU restatement of the stepping-stone `step()` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (steppingstone_amd/) never does.
 *
 * PARITY UNPINNED: the reference's implementation of this path lives in the un-vendored `mocca_envs` submodule
 * (/root/reference/.gitmodules:1-3) on top of `pybullet` (requirements.txt:3, no version pin); neither is present,
 * and the reference ships no test / golden vector for env.step().  This file
 * therefore restates docs/PHYSICS.md (this repository's own specification), constrained by the interface facts the
 * reference does pin: call sites common/envs_utils.py:642-666 (step / auto-reset / hooks), playground/train.py:
 * 231-272 (update_terrain, create_temp_states -> (121,60), update_sample_prob), playground/enjoy.py:52-64
 * (terrain_info columns), common/render_utils.py:47-69 (joint order).  What IS pinned here: Philox4x32-10 against
 * the Random123 known-answer vectors, ABA against an independent CRBA+RNEA numpy solve and analytic cases, the contact
 * stage (detection, Delassus operator, rows, projected Gauss-Seidel, tree response, integration) and the env logic
 * (observation, reward, termination, target logic) and the sampler / reset (stone draw, reset pose) against independent
 * fp64 numpy evaluations written from docs/PHYSICS.md (tests/test_oracle_*.py, tests/np_dynamics.py, np_contact.py,
 * np_env.py, np_terrain.py).  Those pin this file to the
 * written specification, not the specification to PyBullet.
 *
 * Style: deliberately naive dense 6x6 spatial algebra, array-of-struct, one env at a time -- it shares no code
 * with the HIP kernels.  Build twice: -DSSO_REAL=float (parity) and -DSSO_REAL=double (drift characterisation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ss_model_tables.h"

#ifndef SSO_REAL
#define SSO_REAL float
#endif
typedef SSO_REAL real;

#define NJ 21
#define NB 22
#define NSTONE 20
#define NGRID 11
#define NCELL 121
#define OBS_DIM 60
#define STATE_DIM 186
#define RFOOT 8
#define LFOOT 13

#define H_SUB ((real)(1.0 / 240.0))
#define DT_CTRL ((real)(1.0 / 60.0))
#define GRAV ((real)9.8)
#define STEP_RADIUS ((real)0.25)                    /* the reference's step_radius: scale of the step bonus (PHYSICS.md 4.5) */
#define PGS_ITERS 5   /* PHYSICS.md 3.4 (Bullet's numSolverIterations = 5); rounds 1-4: 8 */
#define PGS_WARM 1    /* warm start from the previous substep of the same control step; rounds 1-4: none */
#define ERP ((real)0.2)
#define SLOP ((real)0.001)
#define VCORR_MAX ((real)2.0)
#define PI_D 3.14159265358979323846

static real r_sin(real x) { return (real)sin((double)x); }
static real r_cos(real x) { return (real)cos((double)x); }
static real r_sqrt(real x) { return (real)sqrt((double)x); }
static real r_atan2(real y, real x) { return (real)atan2((double)y, (double)x); }
static real r_asin(real x) { return (real)asin((double)x); }
static real r_exp(real x) { return (real)exp((double)x); }
static real r_abs(real x) { return x < 0 ? -x : x; }
static real r_clamp(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* ------------------------------------------------------------------------------------------------ Philox4x32-10
 * Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11); PHYSICS.md section 6. */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void sso_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }
static real u01(uint32_t x) { return (real)((float)(x >> 8) * 5.9604644775390625e-08f); }

/* ------------------------------------------------------------------------------------------------ small algebra */
typedef struct { real m[3][3]; } mat3;
static void cross3(const real a[3], const real b[3], real o[3]) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static void m3v(const mat3* A, const real v[3], real o[3]) {
  real t[3];
  for (int i = 0; i < 3; ++i) t[i] = A->m[i][0] * v[0] + A->m[i][1] * v[1] + A->m[i][2] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void m3tv(const mat3* A, const real v[3], real o[3]) {
  real t[3];
  for (int i = 0; i < 3; ++i) t[i] = A->m[0][i] * v[0] + A->m[1][i] * v[1] + A->m[2][i] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static mat3 m3mul(const mat3* A, const mat3* B) {
  mat3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A->m[i][0] * B->m[0][j] + A->m[i][1] * B->m[1][j] + A->m[i][2] * B->m[2][j];
  return C;
}
static mat3 rot_axis(int axis, real q) { /* active rotation about a coordinate axis */
  real c = r_cos(q), s = r_sin(q);
  mat3 R = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  if (axis == 0) { R.m[1][1] = c; R.m[1][2] = -s; R.m[2][1] = s; R.m[2][2] = c; }
  else if (axis == 1) { R.m[0][0] = c; R.m[0][2] = s; R.m[2][0] = -s; R.m[2][2] = c; }
  else { R.m[0][0] = c; R.m[0][1] = -s; R.m[1][0] = s; R.m[1][1] = c; }
  return R;
}
static mat3 quat_to_rot(const real q[4]) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  mat3 R = {{{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
             {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
             {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}}};
  return R;
}
static void quat_rpy(const real q[4], real* roll, real* pitch, real* yaw) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  *roll = r_atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y));
  *pitch = r_asin(r_clamp(2 * (w * y - z * x), -1, 1));
  *yaw = r_atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z));
}

/* Pluecker motion transform parent->child: X = [[E,0],[-E rx, E]]  (PHYSICS.md section 1) */
static void make_X(const mat3* E, const real r[3], real X[6][6]) {
  real rx[3][3] = {{0, -r[2], r[1]}, {r[2], 0, -r[0]}, {-r[1], r[0], 0}};
  memset(X, 0, sizeof(real) * 36);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      X[i][j] = E->m[i][j];
      X[i + 3][j + 3] = E->m[i][j];
      real s = 0;
      for (int k = 0; k < 3; ++k) s += E->m[i][k] * rx[k][j];
      X[i + 3][j] = -s;
    }
}
static void mv6(const real A[6][6], const real v[6], real o[6]) {
  real t[6];
  for (int i = 0; i < 6; ++i) { t[i] = 0; for (int j = 0; j < 6; ++j) t[i] += A[i][j] * v[j]; }
  memcpy(o, t, sizeof t);
}
static void mtv6(const real A[6][6], const real v[6], real o[6]) {
  real t[6];
  for (int i = 0; i < 6; ++i) { t[i] = 0; for (int j = 0; j < 6; ++j) t[i] += A[j][i] * v[j]; }
  memcpy(o, t, sizeof t);
}
static void crm(const real v[6], const real m[6], real o[6]) { /* v x m (motion) */
  real a[3], b[3], c[3];
  cross3(v, m, a); cross3(v, m + 3, b); cross3(v + 3, m, c);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf(const real v[6], const real f[6], real o[6]) { /* v x* f (force) */
  real a[3], b[3], c[3];
  cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
static void body_inertia(const sso_model* M, int b, real I[6][6]) {
  real m = M->mass[b], c[3] = {M->com[b][0], M->com[b][1], M->com[b][2]};
  const float* s = M->inertia[b];
  real Io[3][3] = {{s[0], s[3], s[4]}, {s[3], s[1], s[5]}, {s[4], s[5], s[2]}};
  real cx[3][3] = {{0, -c[2], c[1]}, {c[2], 0, -c[0]}, {-c[1], c[0], 0}};
  memset(I, 0, sizeof(real) * 36);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      I[i][j] = Io[i][j];
      I[i][j + 3] = m * cx[i][j];
      I[i + 3][j] = -m * cx[i][j]; /* (m cx)^T = -m cx */
    }
  for (int i = 0; i < 3; ++i) I[i + 3][i + 3] = m;
}
/* 6x6 SPD solve via Cholesky: L L^T x = b */
static void chol6(const real A[6][6], real L[6][6]) {
  memset(L, 0, sizeof(real) * 36);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      real s = A[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      L[i][j] = (i == j) ? r_sqrt(s) : s / L[j][j];
    }
}
static void chol6_solve(const real L[6][6], const real b[6], real x[6]) {
  real y[6];
  for (int i = 0; i < 6; ++i) { real s = b[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
  for (int i = 5; i >= 0; --i) { real s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
}

/* ------------------------------------------------------------------------------------------------ env state */
typedef struct {
  real pos[3], quat[4], vel[6], q[NJ], qd[NJ];
  real pot_prev, z_init, nn_dr;
  double ep_ret;   /* Monitor.update sums the step rewards as Python floats (common/envs_utils.py:131-138) */
  int n, count, elapsed, flags;
  uint32_t rng_ctr;
  real terrain[NSTONE][6];
  float prob[NCELL]; /* per-env sampling grid (f32 as on the device) */
} env_state;

typedef struct {
  float ep_ret, ep_len;
  int32_t bad_transition, steps_reached, update_terrain;
  float ep_ret_lo;   /* (double)ep_ret + (double)ep_ret_lo = the fp64 sum of the fp32 step rewards (include/steppingstone.h) */
} sso_info;

typedef struct {
  int kind, num_envs, curriculum, auto_reset;
  uint64_t seed;
  int64_t env_offset;
  real power;
  const sso_model* M;
  env_state* e;
} sso_env;

/* per-substep workspace (also exported to tests through sso_debug_*) */
typedef struct {
  real X[NB][6][6], v[NB][6], c[NB][6], IA[NB][6][6], pA[NB][6], U[NB][6], Dinv[NB], u[NB], a[NB][6];
  real L0[6][6];
  mat3 Rw[NB];
  real pw[NB][3];
} work;

/* Discrete decisions (tests only).  Every threshold test of a control step whose outcome changes the result
 * discontinuously goes through decide(): class 0 = decisions that change the STATE (contact predicate of a sole
 * corner against a stone, winner among two touching stones, joint-limit switch), class 1 = decisions that only enter
 * reward / done (height, fall, posture bands, joint-at-limit count, target radius).  The sites are visited in a
 * data-independent order, so the running index of a site identifies it within a control step.  While g_dec points at
 * a `decisions` record the step
 *   - keeps, per class, the smallest distance of a decision to its threshold (metres / radians): margin[2];
 *   - lists the indices of the decisions that lie within `tol` of their threshold: near[], nnear;
 *   - inverts the outcome of the decisions listed in force[]: the "other branch" an fp32 implementation with a
 *     different operation order may legitimately take when the margin is within rounding distance;
 *   - records the outcome of every decision (record[]) or takes the outcomes from such a record (replay[]) instead of
 *     evaluating the conditions: a step with frozen decisions is a smooth function of its inputs, which is what the
 *     first-order sensitivity probe of tests/parity_rule.py differentiates.
 * tests/test_gpu_parity.py asserts that EVERY env-step of the HIP path agrees with the oracle either as it ran or with
 * some subset of its near-threshold decisions inverted -- no env-step passes unbounded. */
typedef struct {
  real* margin;
  int seen;
  int nforce; const int32_t* force;
  real tol; int32_t* near; int nnear, cap;
  uint8_t* record; const uint8_t* replay; int ntrace;   /* outcome of decision idx written to / taken from [idx] */
} decisions;
static _Thread_local decisions* g_dec = 0;
#define FAR_MARGIN ((real)1e30)
static int decide(int which, int cond, real m) {
  decisions* D = g_dec;
  if (!D) return cond;
  int idx = D->seen++;
  if (m < 0) m = -m;
  if (D->margin && m < D->margin[which]) D->margin[which] = m;
  if (D->near && m < D->tol) { if (D->nnear < D->cap) D->near[D->nnear] = idx; D->nnear++; }
  if (D->replay && idx < D->ntrace) cond = D->replay[idx];
  for (int k = 0; k < D->nforce; ++k) if (D->force[k] == idx) { cond = !cond; break; }
  if (D->record && idx < D->ntrace) D->record[idx] = (uint8_t)(cond != 0);
  return cond;
}

/* Contact-stage tap (tests only): while g_tap is set, contact_solve() copies its intermediate quantities out so that
 * tests/np_contact.py can check every one of them against an independent fp64 numpy evaluation of PHYSICS.md 3.3-3.4. */
typedef struct {
  real Li[12][12], V0[12], W[8][3][6], bn[8], lam[8][3], nrm[8][3], pen[8];
  real qdf[NJ], v0f[6], dqd[NJ], dv0[6];
  int32_t active[8], stone[8];
} contact_tap;
static _Thread_local contact_tap* g_tap = 0;

/* ------------------------------------------------------------------------------------------------ dynamics */
static void kinematics(const sso_model* M, const env_state* s, work* w) {
  w->Rw[0] = quat_to_rot(s->quat);
  memcpy(w->pw[0], s->pos, sizeof(real) * 3);
  for (int j = 0; j < NJ; ++j) {
    int b = j + 1, par = SSO_PARENT[j];
    mat3 Rj = rot_axis(SSO_AXIS[j], s->q[j]);
    mat3 E;
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) E.m[i][k] = Rj.m[k][i];
    real r[3] = {M->r[j][0], M->r[j][1], M->r[j][2]};
    make_X(&E, r, w->X[b]);
    real t[3];
    m3v(&w->Rw[par], r, t);
    for (int i = 0; i < 3; ++i) w->pw[b][i] = w->pw[par][i] + t[i];
    w->Rw[b] = m3mul(&w->Rw[par], &Rj);
  }
}

/* PHYSICS.md 3.1 + 3.2.  tau_m: motor torques.  Outputs qdd[21], a0[6] (body coords, gravity included). */
static void aba(const sso_model* M, const env_state* s, const real* tau_m, work* w, real* qdd, real* a0) {
  const real h = H_SUB;
  real tau[NJ], Dadd[NJ];
  for (int j = 0; j < NJ; ++j) {
    real q = s->q[j], qd = s->qd[j], viol = 0, kl = 0, dl = 0;
    int over = decide(0, q > M->hi[j], q - M->hi[j]), under = decide(0, q < M->lo[j], q - M->lo[j]);
    if (over) viol = q - M->hi[j]; else if (under) viol = q - M->lo[j];
    if (over || under) { kl = M->klim[j]; dl = M->dlim[j]; }
    tau[j] = tau_m[j] - M->damping[j] * qd - M->stiffness[j] * (q + h * qd) - kl * (viol + h * qd) - dl * qd;
    Dadd[j] = M->armature[j] + h * (M->damping[j] + dl) + h * h * (M->stiffness[j] + kl);
  }
  /* pass 1 */
  memcpy(w->v[0], s->vel, sizeof(real) * 6);
  for (int b = 0; b < NB; ++b) {
    if (b > 0) {
      int j = b - 1, par = SSO_PARENT[j], ax = SSO_AXIS[j];
      mv6(w->X[b], w->v[par], w->v[b]);
      real vj[6] = {0, 0, 0, 0, 0, 0};
      vj[ax] = s->qd[j];
      w->v[b][ax] += s->qd[j];
      crm(w->v[b], vj, w->c[b]);
    }
    body_inertia(M, b, w->IA[b]);
    real Iv[6];
    mv6(w->IA[b], w->v[b], Iv);
    crf(w->v[b], Iv, w->pA[b]);
  }
  /* pass 2 */
  for (int b = NB - 1; b >= 1; --b) {
    int j = b - 1, par = SSO_PARENT[j], ax = SSO_AXIS[j];
    for (int i = 0; i < 6; ++i) w->U[b][i] = w->IA[b][i][ax];
    real D = w->U[b][ax] + Dadd[j];
    w->Dinv[b] = 1 / D;
    w->u[b] = tau[j] - w->pA[b][ax];
    real Ia[6][6], pa[6];
    for (int i = 0; i < 6; ++i)
      for (int k = 0; k < 6; ++k) Ia[i][k] = w->IA[b][i][k] - w->U[b][i] * w->Dinv[b] * w->U[b][k];
    mv6(Ia, w->c[b], pa);
    for (int i = 0; i < 6; ++i) pa[i] += w->pA[b][i] + w->U[b][i] * (w->Dinv[b] * w->u[b]);
    /* parent += X^T Ia X, X^T pa */
    real T[6][6];
    for (int i = 0; i < 6; ++i)
      for (int k = 0; k < 6; ++k) { real t = 0; for (int l = 0; l < 6; ++l) t += Ia[i][l] * w->X[b][l][k]; T[i][k] = t; }
    for (int i = 0; i < 6; ++i)
      for (int k = 0; k < 6; ++k) { real t = 0; for (int l = 0; l < 6; ++l) t += w->X[b][l][i] * T[l][k]; w->IA[par][i][k] += t; }
    real pp[6];
    mtv6(w->X[b], pa, pp);
    for (int i = 0; i < 6; ++i) w->pA[par][i] += pp[i];
  }
  /* base */
  chol6(w->IA[0], w->L0);
  real rhs[6];
  for (int i = 0; i < 6; ++i) rhs[i] = -w->pA[0][i];
  chol6_solve(w->L0, rhs, w->a[0]);
  /* pass 3 */
  for (int b = 1; b < NB; ++b) {
    int j = b - 1, par = SSO_PARENT[j], ax = SSO_AXIS[j];
    real ap[6];
    mv6(w->X[b], w->a[par], ap);
    real dot = 0;
    for (int i = 0; i < 6; ++i) { ap[i] += w->c[b][i]; dot += w->U[b][i] * ap[i]; }
    qdd[j] = w->Dinv[b] * (w->u[b] - dot);
    memcpy(w->a[b], ap, sizeof ap);
    w->a[b][ax] += qdd[j];
  }
  /* uniform gravity field: a0 += [0; R^T g] */
  real g[3] = {0, 0, -GRAV}, gb[3];
  m3tv(&w->Rw[0], g, gb);
  for (int i = 0; i < 6; ++i) a0[i] = w->a[0][i];
  a0[3] += gb[0]; a0[4] += gb[1]; a0[5] += gb[2];
}

/* ABA impulse response (PHYSICS.md 3.4): impulses fimp[b] (body coords) -> dv of every body, dqd, dv0 */
static void impulse_response(const work* w, real fimp[NB][6], real dvb[NB][6], real* dqd) {
  real p[NB][6], uu[NB];
  for (int b = 0; b < NB; ++b) for (int i = 0; i < 6; ++i) p[b][i] = -fimp[b][i];
  for (int b = NB - 1; b >= 1; --b) {
    int j = b - 1, par = SSO_PARENT[j], ax = SSO_AXIS[j];
    uu[b] = -p[b][ax];
    real pa[6], pp[6];
    for (int i = 0; i < 6; ++i) pa[i] = p[b][i] + w->U[b][i] * (w->Dinv[b] * uu[b]);
    mtv6(w->X[b], pa, pp);
    for (int i = 0; i < 6; ++i) p[par][i] += pp[i];
  }
  real rhs[6];
  for (int i = 0; i < 6; ++i) rhs[i] = -p[0][i];
  chol6_solve(w->L0, rhs, dvb[0]);
  for (int b = 1; b < NB; ++b) {
    int j = b - 1, par = SSO_PARENT[j], ax = SSO_AXIS[j];
    mv6(w->X[b], dvb[par], dvb[b]);
    real dot = 0;
    for (int i = 0; i < 6; ++i) dot += w->U[b][i] * dvb[b][i];
    dqd[j] = w->Dinv[b] * (uu[b] - dot);
    dvb[b][ax] += dqd[j];
  }
}

typedef struct {
  int active, stone;
  real n[3], pen, r[3]; /* normal (world), penetration, corner (foot frame) */
  real lam[3];
} contact;

typedef struct {
  int foot_contact[2], foot_on_target[2];
  real sole[2][3];
} foot_report;

static void stone_normal(const real* st, real nrm[3]) {
  mat3 Rz = rot_axis(2, st[3]), Ry = rot_axis(1, st[5]), Rx = rot_axis(0, st[4]);
  mat3 T = m3mul(&Ry, &Rx), Rs = m3mul(&Rz, &T);
  nrm[0] = Rs.m[0][2]; nrm[1] = Rs.m[1][2]; nrm[2] = Rs.m[2][2];
}

/* The stepping surface (PHYSICS.md 3.3, round 6): a PLANK whose footprint seen from above is 2 a x 2 b, aligned with the stone's heading
 * phi: with l the in-plane offset of the corner from the stone's centre, u = l_x cos(phi) + l_y sin(phi) along the heading and
 * v = l_y cos(phi) - l_x sin(phi) across it; inside when |u| < a and |v| < b.  a, b: SSO_PLANK_HALF_LENGTH / _WIDTH of the tables.
 * Study knobs (tools/sysid_policy.py ONLY; never the specification): another plank, a DISC of radius r instead (rounds 1-5: 0.25 m, then
 * 0.45 m), another stone spacing law, the rounds-1-5 on-target rule, an extra target radius (infinite-plane control). */
static real g_plank_a = SSO_PLANK_HALF_LENGTH, g_plank_b = SSO_PLANK_HALF_WIDTH;
void sso_debug_set_plank(double a, double b) { g_plank_a = a > 0 ? (real)a : SSO_PLANK_HALF_LENGTH; g_plank_b = a > 0 ? (real)b : SSO_PLANK_HALF_WIDTH; }
static real g_stone_r = 0;      /* > 0: disc study mode */
void sso_debug_set_stone_radius(double r) { g_stone_r = (real)r; }
static real g_dr_lo = (real)0.65, g_dr_span = (real)0.6;
void sso_debug_set_dr(double lo, double span) { g_dr_lo = (real)lo; g_dr_span = (real)span; }
static int g_target_carried = 1;    /* 1 (specification): a foot is on the target through a corner that stone n CARRIES; 0: rounds 1-5 */
static real g_target_r = 0;     /* > 0, with the rounds-1-5 rule: the corner must also lie within this radius of stone n's axis */
void sso_debug_set_target_rule(int carried_only) { g_target_carried = carried_only; }
void sso_debug_set_target_radius(double r) { g_target_r = (real)r; }
static int plank_inside(const real* st, const real* l, real* margin) {
  real c = r_cos(st[3]), sn = r_sin(st[3]);
  real u = l[0] * c + l[1] * sn, v = l[1] * c - l[0] * sn;
  real mu = g_plank_a - (u < 0 ? -u : u), mv = g_plank_b - (v < 0 ? -v : v);
  *margin = mu < mv ? mu : mv;
  return mu > 0 && mv > 0;
}
static void detect(const sso_model* M, const env_state* s, const work* w, contact ct[8], foot_report* fr) {
  int n = s->n;
  int idx[3] = {n - 1 < 0 ? 0 : n - 1, n, n + 1 > NSTONE - 1 ? NSTONE - 1 : n + 1};
  for (int f = 0; f < 2; ++f) {
    int b = f == 0 ? RFOOT : LFOOT;
    fr->foot_contact[f] = 0; fr->foot_on_target[f] = 0;
    fr->sole[f][0] = fr->sole[f][1] = fr->sole[f][2] = 0;
    for (int k = 0; k < 4; ++k) {
      contact* c = &ct[f * 4 + k];
      memset(c, 0, sizeof *c);
      for (int i = 0; i < 3; ++i) c->r[i] = M->corners[k][i];
      if (f == 1) c->r[1] = -c->r[1];   /* the left sole's corner list is the mirror image of the right one's */
      real P[3];
      m3v(&w->Rw[b], c->r, P);
      for (int i = 0; i < 3; ++i) { P[i] += w->pw[b][i]; fr->sole[f][i] += (real)0.25 * P[i]; }
      real best = 0;
      for (int si = 0; si < 3; ++si) {
        const int sl = si == 0 ? 1 : (si == 1 ? 0 : 2);   /* the target stone n first: it wins an exact tie, then n-1, then n+1 */
        const real* st = s->terrain[idx[sl]];
        real nrm[3], dv[3] = {P[0] - st[0], P[1] - st[1], P[2] - st[2]};
        stone_normal(st, nrm);
        real d = dv[0] * nrm[0] + dv[1] * nrm[1] + dv[2] * nrm[2];
        real lx = dv[0] - d * nrm[0], ly = dv[1] - d * nrm[1], lz = dv[2] - d * nrm[2];
        real rho2 = lx * lx + ly * ly + lz * lz;
        real g1 = -d, g2 = d + (real)0.10, g3 = 0, l3[3] = {lx, ly, lz};
        int inside = plank_inside(st, l3, &g3);
        if (g_stone_r > 0) { inside = rho2 < g_stone_r * g_stone_r; g3 = g_stone_r - r_sqrt(rho2); }       /* disc study only */
        real gm = g1 < g2 ? g1 : g2;
        if (g3 < gm) gm = g3;
        int touch = decide(0, d < 0 && d > (real)-0.10 && inside, gm);
        /* rounds 1-5 (study only): on the target when a corner touches stone n, whichever stone carries that corner */
        if (touch && sl == 1 && !g_target_carried && (g_target_r <= 0 || rho2 < g_target_r * g_target_r)) fr->foot_on_target[f] = 1;
        /* two touching stones: the deeper one wins (a first touching stone always does, also when its predicate was
         * forced against d >= 0); an exact tie between COPLANAR stones goes to the stone visited first (n, n-1, n+1) and is not a decision -- either
         * winner gives the same normal and the same depth */
        int coplanar_tie = c->active && d == best && nrm[0] == c->n[0] && nrm[1] == c->n[1] && nrm[2] == c->n[2];
        int wins = decide(0, !c->active || d < best, (touch && c->active && !coplanar_tie) ? d - best : FAR_MARGIN);
        if (touch && wins) {
          best = d; c->active = 1; c->stone = idx[sl]; c->pen = -d;
          c->n[0] = nrm[0]; c->n[1] = nrm[1]; c->n[2] = nrm[2];
        }
      }
      if (c->active) fr->foot_contact[f] = 1;
      if (g_target_carried && c->active && c->stone == idx[1]) fr->foot_on_target[f] = 1;
    }
  }
}

/* solver knobs: PHYSICS.md 3.4 fixes them; sso_debug_set_solver() exists only for the convergence study of
 * tools/pgs_convergence.py */
static int g_pgs_iters = PGS_ITERS;
static int g_pgs_warm = PGS_WARM;
void sso_debug_set_solver(int iters, int warm) { g_pgs_iters = iters; g_pgs_warm = warm; }
/* further knobs of the same study (tools/spec_deviations.py -> docs/HISTORY.md section 3 table): Baumgarte factor, and
 * Gauss-Seidel instead of Jacobi BETWEEN the feet (a row then sees the other foot's impulses of the same sweep) */
static real g_erp = ERP;
static int g_seq_feet = 0;
void sso_debug_set_variant(double erp, int seq_feet) { g_erp = (real)erp; g_seq_feet = seq_feet; }

typedef struct { real lam[8][3]; int stone[8]; } warm_state;   /* impulses of the previous substep of this step */

/* PHYSICS.md 3.3-3.4: returns dqd/dv0 to add to the free velocities */
static void contact_solve(const sso_model* M, const work* w, const real* qd_free, const real* v0_free,
                          contact ct[8], warm_state* ws, real* dqd, real* dv0) {
  memset(dqd, 0, sizeof(real) * NJ);
  memset(dv0, 0, sizeof(real) * 6);
  int any = 0;
  for (int k = 0; k < 8; ++k) any |= ct[k].active;
  if (g_tap) {
    memset(g_tap, 0, sizeof *g_tap);
    memcpy(g_tap->qdf, qd_free, sizeof(real) * NJ); memcpy(g_tap->v0f, v0_free, sizeof(real) * 6);
    for (int k = 0; k < 8; ++k) {
      g_tap->active[k] = ct[k].active; g_tap->stone[k] = ct[k].active ? ct[k].stone : -1; g_tap->pen[k] = ct[k].pen;
      memcpy(g_tap->nrm[k], ct[k].n, sizeof(real) * 3);
    }
  }
  if (!any) { for (int k = 0; k < 8; ++k) ws->stone[k] = -1; return; }
  const int foot_body[2] = {RFOOT, LFOOT};
  /* Lambda^-1 by 12 unit impulses */
  real Li[12][12];
  real fimp[NB][6], dvb[NB][6];
  real tmp[NJ];
  for (int f = 0; f < 2; ++f)
    for (int i = 0; i < 6; ++i) {
      memset(fimp, 0, sizeof fimp);
      fimp[foot_body[f]][i] = 1;
      impulse_response(w, fimp, dvb, tmp);
      for (int g = 0; g < 2; ++g)
        for (int k = 0; k < 6; ++k) Li[g * 6 + k][f * 6 + i] = dvb[foot_body[g]][k];
    }
  /* foot twists under the free velocities */
  real vb[NB][6], V[12];
  memcpy(vb[0], v0_free, sizeof(real) * 6);
  for (int b = 1; b < NB; ++b) {
    int j = b - 1;
    mv6(w->X[b], vb[SSO_PARENT[j]], vb[b]);
    vb[b][SSO_AXIS[j]] += qd_free[j];
  }
  for (int f = 0; f < 2; ++f) for (int k = 0; k < 6; ++k) V[f * 6 + k] = vb[foot_body[f]][k];
  if (g_tap) memcpy(g_tap->V0, V, sizeof V);
  /* rows */
  real W[8][3][6];
  real bn[8];
  const real mu = M->friction;
  for (int k = 0; k < 8; ++k) {
    contact* c = &ct[k];
    if (!c->active) continue;
    int f = k / 4;
    real* n = c->n;
    real t1[3] = {1 - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
    real nt = r_sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
    for (int i = 0; i < 3; ++i) t1[i] /= nt;
    real t2[3];
    cross3(n, t1, t2);
    const real* dirs[3] = {n, t1, t2};
    for (int d = 0; d < 3; ++d) {
      real df[3], m[3];
      m3tv(&w->Rw[foot_body[f]], dirs[d], df);
      cross3(c->r, df, m);
      for (int i = 0; i < 3; ++i) { W[k][d][i] = m[i]; W[k][d][i + 3] = df[i]; }
    }
    real corr = c->pen - SLOP;
    if (corr < 0) corr = 0;
    bn[k] = g_erp * corr / H_SUB;
    if (bn[k] > VCORR_MAX) bn[k] = VCORR_MAX;
    c->lam[0] = c->lam[1] = c->lam[2] = 0;
  }
  /* warm start: a corner that was in contact in the previous substep of this control step starts from that substep's
   * impulses (applied to the twists before the first sweep) */
  if (g_pgs_warm)
    for (int k = 0; k < 8; ++k) {
      contact* c = &ct[k];
      if (!c->active || ws->stone[k] < 0) continue;
      int f = k / 4;
      for (int d = 0; d < 3; ++d) {
        c->lam[d] = ws->lam[k][d];
        for (int i = 0; i < 12; ++i) { real y = 0; for (int l = 0; l < 6; ++l) y += Li[i][f * 6 + l] * W[k][d][l]; V[i] += y * c->lam[d]; }
      }
    }
  /* Gauss-Seidel inside each foot, Jacobi between the feet: during a sweep every row sees its own foot's twist
   * up to date, and the other foot's impulses of THIS sweep only once both feet have finished it. */
  for (int it = 0; it < g_pgs_iters; ++it) {
    real Vnext[12];
    memcpy(Vnext, V, sizeof Vnext);
    for (int k = 0; k < 8; ++k) {
      contact* c = &ct[k];
      if (!c->active) continue;
      int f = k / 4, g = 1 - f;
      for (int d = 0; d < 3; ++d) {
        real y[12], A = 0, vrel = 0;
        for (int i = 0; i < 12; ++i) { y[i] = 0; for (int l = 0; l < 6; ++l) y[i] += Li[i][f * 6 + l] * W[k][d][l]; }
        for (int l = 0; l < 6; ++l) { A += W[k][d][l] * y[f * 6 + l]; vrel += W[k][d][l] * V[f * 6 + l]; }
        real target = d == 0 ? bn[k] : 0;
        real lam_new = c->lam[d] + (target - vrel) / A;
        if (d == 0) { if (lam_new < 0) lam_new = 0; }
        else { real lim = mu * c->lam[0]; lam_new = r_clamp(lam_new, -lim, lim); }
        real dl = lam_new - c->lam[d];
        c->lam[d] = lam_new;
        for (int i = 0; i < 6; ++i) {
          V[f * 6 + i] += y[f * 6 + i] * dl;          /* own foot: immediately (also tracked in Vnext) */
          Vnext[f * 6 + i] += y[f * 6 + i] * dl;
          Vnext[g * 6 + i] += y[g * 6 + i] * dl;      /* other foot: visible from the next sweep on */
          if (g_seq_feet) V[g * 6 + i] += y[g * 6 + i] * dl;   /* (study variant: visible at once) */
        }
      }
    }
    memcpy(V, Vnext, sizeof Vnext);
  }
  for (int k = 0; k < 8; ++k) {
    ws->stone[k] = ct[k].active ? ct[k].stone : -1;
    for (int d = 0; d < 3; ++d) ws->lam[k][d] = ct[k].active ? ct[k].lam[d] : 0;
  }
  /* apply accumulated foot wrenches to the whole tree */
  memset(fimp, 0, sizeof fimp);
  for (int k = 0; k < 8; ++k) {
    if (!ct[k].active) continue;
    int f = k / 4;
    for (int d = 0; d < 3; ++d) for (int i = 0; i < 6; ++i) fimp[foot_body[f]][i] += W[k][d][i] * ct[k].lam[d];
  }
  impulse_response(w, fimp, dvb, dqd);
  memcpy(dv0, dvb[0], sizeof(real) * 6);
  if (g_tap) {
    memcpy(g_tap->Li, Li, sizeof Li);
    for (int k = 0; k < 8; ++k) {
      if (!ct[k].active) continue;
      memcpy(g_tap->W[k], W[k], sizeof W[k]); g_tap->bn[k] = bn[k]; memcpy(g_tap->lam[k], ct[k].lam, sizeof(real) * 3);
    }
    memcpy(g_tap->dqd, dqd, sizeof(real) * NJ); memcpy(g_tap->dv0, dv0, sizeof(real) * 6);
  }
}

static void substep(const sso_model* M, env_state* s, const real* tau_m, foot_report* fr, warm_state* ws) {
  work w;
  const real h = H_SUB;
  real qdd[NJ], a0[6], qdf[NJ], v0f[6], dqd[NJ], dv0[6];
  contact ct[8];
  kinematics(M, s, &w);
  aba(M, s, tau_m, &w, qdd, a0);
  for (int j = 0; j < NJ; ++j) qdf[j] = s->qd[j] + h * qdd[j];
  for (int i = 0; i < 6; ++i) v0f[i] = s->vel[i] + h * a0[i];
  detect(M, s, &w, ct, fr);
  contact_solve(M, &w, qdf, v0f, ct, ws, dqd, dv0);
  for (int j = 0; j < NJ; ++j) { s->qd[j] = qdf[j] + dqd[j]; s->q[j] += h * s->qd[j]; }
  for (int i = 0; i < 6; ++i) s->vel[i] = v0f[i] + dv0[i];
  real vw[3];
  m3v(&w.Rw[0], s->vel + 3, vw);
  for (int i = 0; i < 3; ++i) s->pos[i] += h * vw[i];
  real qw = s->quat[0], qx = s->quat[1], qy = s->quat[2], qz = s->quat[3];
  real ox = s->vel[0], oy = s->vel[1], oz = s->vel[2], hh = (real)0.5 * h;
  real nw = qw + hh * (-qx * ox - qy * oy - qz * oz);
  real nx = qx + hh * (qw * ox + qy * oz - qz * oy);
  real ny = qy + hh * (qw * oy - qx * oz + qz * ox);
  real nz = qz + hh * (qw * oz + qx * oy - qy * ox);
  real inv = 1 / r_sqrt(nw * nw + nx * nx + ny * ny + nz * nz);
  s->quat[0] = nw * inv; s->quat[1] = nx * inv; s->quat[2] = ny * inv; s->quat[3] = nz * inv;
}

/* ------------------------------------------------------------------------------------------------ terrain */
static const real DEG = (real)(PI_D / 180.0);
static real yaw_sample(int i) { return (real)(-20.0 + 4.0 * i) * DEG; }
static real pitch_sample(int j) { return (real)(-30.0 + 6.0 * j) * DEG; }

static void env_block(const sso_env* E, int e, env_state* s, uint32_t out[4]) {
  uint64_t gid = (uint64_t)(E->env_offset + e);
  uint32_t ctr[4] = {s->rng_ctr, 0u, (uint32_t)gid, 0u};
  uint32_t key[2] = {(uint32_t)E->seed, (uint32_t)(E->seed >> 32)};
  philox4x32_10(ctr, key, out);
  s->rng_ctr += 1;
}

/* inverse-CDF draw from the 11x11 yaw x pitch grid the trainer pushes with envs.update_sample_prob
 * (playground/train.py:263-271, 356-360; common/envs_utils.py:568-571, 654-655); PHYSICS.md section 6 */
static int sample_cell(const float* prob, float u) {
  float cdf = 0.f;
  int last = 0;
  for (int k = 0; k < NCELL; ++k) {
    if (prob[k] > 0.f) last = k;
    cdf += prob[k];
    if (u < cdf) return k;
  }
  return last;
}

static void place_stone(env_state* s, int k, real yaw, real pitch, real dr, real xt, real yt) {
  const real* p = s->terrain[k - 1];
  real phi = p[3] + yaw, planar = dr * r_cos(pitch);
  s->terrain[k][0] = p[0] + planar * r_cos(phi);
  s->terrain[k][1] = p[1] + planar * r_sin(phi);
  s->terrain[k][2] = p[2] + dr * r_sin(pitch);
  s->terrain[k][3] = phi;
  s->terrain[k][4] = xt;
  s->terrain[k][5] = yt;
}

/* draw stone k from stone k-1 (PHYSICS.md section 6); returns dr.  Counterpart of env.sample_next_next_step() /
 * terrain_info[next_next_step, 0:6] = x,y,z,phi,x_tilt,y_tilt (playground/enjoy.py:52-64) */
static real draw_stone(const sso_env* E, int e, env_state* s, int k) {
  uint32_t r[4];
  env_block(E, e, s, r);
  int cell = sample_cell(s->prob, (float)(r[0] >> 8) * 5.9604644775390625e-08f);
  real ratio = (real)E->curriculum / (real)5;
  real dr = g_dr_lo + u01(r[1]) * (g_dr_span * ratio);
  real tilt = (real)15.0 * DEG * ratio;
  real xt = (2 * u01(r[2]) - 1) * tilt, yt = (2 * u01(r[3]) - 1) * tilt;
  place_stone(s, k, yaw_sample(cell / NGRID), pitch_sample(cell % NGRID), dr, xt, yt);
  return dr;
}

/* ------------------------------------------------------------------------------------------------ obs */
static real planar_dist(const real* a, const real* b) {
  real dx = a[0] - b[0], dy = a[1] - b[1];
  return r_sqrt(dx * dx + dy * dy);
}
static void target_features(const env_state* s, const real* stone, real yaw, float* o) {
  real dx = stone[0] - s->pos[0], dy = stone[1] - s->pos[1], dz = stone[2] - s->pos[2];
  real d = r_sqrt(dx * dx + dy * dy), ang = r_atan2(dy, dx) - yaw;
  o[0] = (float)(r_sin(ang) * d); o[1] = (float)(r_cos(ang) * d); o[2] = (float)dz;
  o[3] = (float)stone[4]; o[4] = (float)stone[5];
}
/* 60-float observation (dims pinned by the shipped checkpoints): 50 robot + 2 x 5 target features;
 * PHYSICS.md section 5 */
static void write_obs(const sso_model* M, const env_state* s, float* o) {
  real roll, pitch, yaw;
  quat_rpy(s->quat, &roll, &pitch, &yaw);
  mat3 R = quat_to_rot(s->quat);
  real vw[3];
  m3v(&R, s->vel + 3, vw);
  real cy = r_cos(yaw), sy = r_sin(yaw);
  real f[50];
  f[0] = s->pos[2] - s->z_init;
  f[1] = cy * vw[0] + sy * vw[1];
  f[2] = -sy * vw[0] + cy * vw[1];
  f[3] = vw[2];
  f[4] = roll; f[5] = pitch;
  for (int j = 0; j < NJ; ++j) {
    real mid = (real)0.5 * (M->lo[j] + M->hi[j]);
    /* policy coordinates (PHYSICS.md 2): sigma_j * (angle, rate about the +axis) */
    f[6 + j] = SSO_POLICY_SIGN[j] * (2 * (s->q[j] - mid) / (M->hi[j] - M->lo[j]));
    f[27 + j] = SSO_POLICY_SIGN[j] * ((real)0.1 * s->qd[j]);
  }
  f[48] = (s->flags & 1) ? 1 : 0;
  f[49] = (s->flags & 2) ? 1 : 0;
  for (int i = 0; i < 50; ++i) o[i] = (float)r_clamp(f[i], -5, 5);
  int n1 = s->n + 1 > NSTONE - 1 ? NSTONE - 1 : s->n + 1;
  target_features(s, s->terrain[s->n], yaw, o + 50);
  target_features(s, s->terrain[n1], yaw, o + 55);
}

/* ------------------------------------------------------------------------------------------------ reset/step */
/* env.reset() as called by the worker (common/envs_utils.py:643-644, 647-648); PHYSICS.md section 7 */
static void env_reset(const sso_env* E, int e) {
  const sso_model* M = E->M;
  env_state* s = &E->e[e];
  /* provisional straight, flat path; stone k >= 3 is drawn when it becomes the look-ahead stone (PHYSICS.md 6) */
  memset(s->terrain, 0, sizeof s->terrain);
  for (int k = 1; k < NSTONE; ++k) s->terrain[k][0] = (real)0.75 * (real)k;
  s->n = 1; s->count = 0; s->elapsed = 0; s->flags = 0;
  s->nn_dr = (real)0.75;
  s->pos[0] = 0; s->pos[1] = 0; s->pos[2] = M->stand_height + (real)0.01;
  s->quat[0] = 1; s->quat[1] = s->quat[2] = s->quat[3] = 0;
  memset(s->vel, 0, sizeof s->vel);
  memset(s->qd, 0, sizeof s->qd);
  for (int b = 0; b < 6; ++b) {
    uint32_t r[4];
    env_block(E, e, s, r);
    for (int i = 0; i < 4; ++i) {
      int j = b * 4 + i;
      if (j >= NJ) break;
      real q = M->q0[j] + (real)0.05 * (2 * u01(r[i]) - 1);
      s->q[j] = r_clamp(q, M->lo[j] + (real)0.02, M->hi[j] - (real)0.02);
    }
  }
  s->z_init = s->pos[2];
  s->ep_ret = 0;
  s->pot_prev = -planar_dist(s->terrain[s->n], s->pos) / DT_CTRL;
}

static int state_finite(const env_state* s) {
  real acc = 0;
  for (int i = 0; i < 3; ++i) acc += s->pos[i];
  for (int i = 0; i < 4; ++i) acc += s->quat[i];
  for (int i = 0; i < 6; ++i) acc += s->vel[i];
  for (int j = 0; j < NJ; ++j) acc += s->q[j] + s->qd[j];
  return isfinite((double)acc);
}

/* env.step(action) + the worker's auto-reset (common/envs_utils.py:645-649: terminal reward/done/info, RESET obs),
 * Monitor's episode statistics (:131-153), TimeLimitMask's bad_transition (:59-65), env.update_terrain
 * (playground/train.py:245); PHYSICS.md section 4 */
static void env_step(const sso_env* E, int e, const float* act, float* obs, float* rew, uint8_t* done, sso_info* info) {
  const sso_model* M = E->M;
  env_state* s = &E->e[e];
  real a[NJ], tau[NJ];
  for (int j = 0; j < NJ; ++j) {
    a[j] = r_clamp((real)act[j], -1, 1);
    tau[j] = E->power * M->torque[j] * (SSO_POLICY_SIGN[j] * a[j]);   /* action in policy coordinates (PHYSICS.md 2) */
  }
  foot_report fr;
  warm_state ws;
  for (int k = 0; k < 8; ++k) ws.stone[k] = -1;
  for (int k = 0; k < 4; ++k) substep(M, s, tau, &fr, &ws);
  s->elapsed += 1;
  s->flags = (fr.foot_contact[0] ? 1 : 0) | (fr.foot_contact[1] ? 2 : 0);
  int finite = state_finite(s);
  int n_old = s->n, advanced = 0;
  /* 5. target logic */
  real step_bonus = 0;
  int reached = fr.foot_on_target[0] || fr.foot_on_target[1];
  if (reached) {
    s->count += 1;
    if (s->count == 1) {
      real d0 = planar_dist(fr.sole[0], s->terrain[n_old]), d1 = planar_dist(fr.sole[1], s->terrain[n_old]);
      step_bonus = 50 * r_exp(-(d0 < d1 ? d0 : d1) / STEP_RADIUS);
    }
    if (s->count >= 2 && s->n < NSTONE - 1) {
      s->n += 1; s->count = 0; advanced = 1;
      if (s->n + 1 <= NSTONE - 1) s->nn_dr = draw_stone(E, e, s, s->n + 1);
    }
  }
  /* 6. progress */
  real pot = -planar_dist(s->terrain[n_old], s->pos) / DT_CTRL;
  real progress = pot - s->pot_prev;
  s->pot_prev = advanced ? -planar_dist(s->terrain[s->n], s->pos) / DT_CTRL : pot;
  /* 7-8 */
  real dist_t = planar_dist(s->terrain[s->n], s->pos);
  int inside = decide(1, dist_t < (real)0.15, s->n == NSTONE - 1 ? dist_t - (real)0.15 : FAR_MARGIN);
  real target_bonus = (s->n == NSTONE - 1 && inside) ? 2 : 0;
  real zs = fr.sole[0][2] < fr.sole[1][2] ? fr.sole[0][2] : fr.sole[1][2];
  real tall_bonus = decide(1, s->pos[2] - zs > (real)0.7, s->pos[2] - zs - (real)0.7) ? 2 : -1;
  int i0 = s->n - 1 < 0 ? 0 : s->n - 1, i2 = s->n + 1 > NSTONE - 1 ? NSTONE - 1 : s->n + 1;
  real zlow = s->terrain[i0][2];
  if (s->terrain[s->n][2] < zlow) zlow = s->terrain[s->n][2];
  if (s->terrain[i2][2] < zlow) zlow = s->terrain[i2][2];
  int fell = decide(1, s->pos[2] < zlow + (real)0.3, s->pos[2] - zlow - (real)0.3);
  int d = tall_bonus < 0 || fell || !finite;
  int timeout = s->elapsed >= 1000;
  int bad = timeout; /* TimeLimitMask, common/envs_utils.py:59-65: done at the step limit, whatever else ended it */
  d = d || timeout;
  /* 9 */
  real roll, pitch, yaw;
  quat_rpy(s->quat, &roll, &pitch, &yaw);
  real posture = 0;
  real mp = pitch + (real)0.2 < (real)0.4 - pitch ? pitch + (real)0.2 : (real)0.4 - pitch;
  real mr = roll + (real)0.4 < (real)0.4 - roll ? roll + (real)0.4 : (real)0.4 - roll;
  if (!decide(1, pitch > (real)-0.2 && pitch < (real)0.4, mp)) posture += r_abs(pitch);
  if (!decide(1, roll > (real)-0.4 && roll < (real)0.4, mr)) posture += r_abs(roll);
  real e_sum = 0, a2 = 0;
  int at_limit = 0;
  for (int j = 0; j < NJ; ++j) {
    e_sum += r_abs(a[j] * ((real)0.1 * s->qd[j]));
    a2 += a[j] * a[j];
    real mid = (real)0.5 * (M->lo[j] + M->hi[j]);
    real qn = r_abs(2 * (s->q[j] - mid) / (M->hi[j] - M->lo[j]));
    if (decide(1, qn > (real)0.99, (qn - (real)0.99) * (real)0.5 * (M->hi[j] - M->lo[j]))) at_limit += 1;
  }
  real energy = ((real)4.5 / NJ) * (e_sum / NJ) + ((real)0.225 / NJ) * (a2 / NJ);
  real r = progress + step_bonus + target_bonus + tall_bonus - energy - posture - (real)0.1 * at_limit;
  if (!finite || !isfinite((double)r)) r = 0;
  s->ep_ret += (double)(float)r;
  *rew = (float)r;
  *done = (uint8_t)d;
  info->ep_ret = (float)s->ep_ret;
  info->ep_ret_lo = (float)(s->ep_ret - (double)info->ep_ret);
  info->ep_len = (float)s->elapsed;
  info->bad_transition = bad;
  info->steps_reached = s->n;
  info->update_terrain = advanced;
  if (d && E->auto_reset) env_reset(E, e);
  write_obs(M, s, obs);
}

/* ------------------------------------------------------------------------------------------------ C API */
/* update_curriculum(c): uniform over the (2c+1)^2 window around the centre cell [5,5] of the 11x11 grid
 * (playground/train.py:132-133 prob_filter[5,5], :412 / :555 window [5-c:5+c+1]); update_specialist(s): the ring */
static void fill_window(float* prob, int c, int ring) {
  int cnt = 0;
  for (int i = 0; i < NGRID; ++i)
    for (int j = 0; j < NGRID; ++j) {
      int di = abs(i - 5), dj = abs(j - 5), m = di > dj ? di : dj;
      int in = ring ? (m == c) : (m <= c);
      prob[i * NGRID + j] = in ? 1.f : 0.f;
      cnt += in;
    }
  for (int k = 0; k < NCELL; ++k) prob[k] = prob[k] / (float)cnt;
}

/* Model override (tools/sysid_policy.py ONLY: the informational search over the specification's free numbers against the
 * reference's shipped policies).  Environments created AFTER the call use the given tables instead of the
 * compiled-in ones; no test and nothing in the package calls it, so the specification the parity tests judge is SSO_MODELS. */
static sso_model g_model_override[2];
static int g_model_overridden[2] = {0, 0};
void sso_debug_set_model(int kind, const sso_model* m) {
  if (kind < 0 || kind > 1) return;
  if (m) { g_model_override[kind] = *m; g_model_overridden[kind] = 1; } else g_model_overridden[kind] = 0;
}
int sso_model_size(void) { return (int)sizeof(sso_model); }

sso_env* sso_create(int kind, int num_envs, uint64_t seed, int64_t env_offset) {
  sso_env* E = (sso_env*)calloc(1, sizeof *E);
  E->kind = kind; E->num_envs = num_envs; E->seed = seed; E->env_offset = env_offset;
  E->M = g_model_overridden[kind] ? &g_model_override[kind] : &SSO_MODELS[kind];
  E->power = 1; E->curriculum = 0; E->auto_reset = 1;
  E->e = (env_state*)calloc((size_t)num_envs, sizeof(env_state));
  for (int e = 0; e < num_envs; ++e) { fill_window(E->e[e].prob, 0, 0); E->e[e].quat[0] = 1; }
  return E;
}
void sso_destroy(sso_env* E) { if (E) { free(E->e); free(E); } }
void sso_reset(sso_env* E, float* obs) {
  for (int e = 0; e < E->num_envs; ++e) { env_reset(E, e); write_obs(E->M, &E->e[e], obs + (size_t)e * OBS_DIM); }
}
void sso_step(sso_env* E, const float* act, float* obs, float* rew, uint8_t* done, sso_info* info) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < E->num_envs; ++e)
    env_step(E, e, act + (size_t)e * NJ, obs + (size_t)e * OBS_DIM, rew + e, done + e, info + e);
}
/* sso_step that also returns margins[N][2]: per env the smallest distance of a class-0 / class-1 decision of this
 * control step to its threshold (see decide() above) */
void sso_step_margins(sso_env* E, const float* act, float* obs, float* rew, uint8_t* done, sso_info* info, real* margins) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < E->num_envs; ++e) {
    decisions D;
    memset(&D, 0, sizeof D);
    margins[2 * e] = margins[2 * e + 1] = FAR_MARGIN;
    D.margin = margins + 2 * e;
    g_dec = &D;
    env_step(E, e, act + (size_t)e * NJ, obs + (size_t)e * OBS_DIM, rew + e, done + e, info + e);
    g_dec = 0;
  }
}
/* sso_step_margins that also lists, per env, the indices of the decisions within `tol` of their threshold:
 * near[N][cap] (first cap of them) and nnear[N] (their true number, which may exceed cap) */
void sso_step_near(sso_env* E, const float* act, float* obs, float* rew, uint8_t* done, sso_info* info, real* margins,
                   double tol, int32_t* near, int32_t* nnear, int cap) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < E->num_envs; ++e) {
    decisions D;
    memset(&D, 0, sizeof D);
    margins[2 * e] = margins[2 * e + 1] = FAR_MARGIN;
    D.margin = margins + 2 * e;
    D.tol = (real)tol; D.near = near + (size_t)e * cap; D.cap = cap;
    g_dec = &D;
    env_step(E, e, act + (size_t)e * NJ, obs + (size_t)e * OBS_DIM, rew + e, done + e, info + e);
    g_dec = 0;
    nnear[e] = D.nnear;
  }
}
/* The general form: any of margins [N][2], near [N][cap] + nnear [N], force [N][cap] + nforce [N], record [N][ntrace],
 * replay [N][ntrace] may be null.  SSO_MAX_DECISIONS bounds the number of decision sites of a control step. */
#define SSO_MAX_DECISIONS 400
int sso_max_decisions(void) { return SSO_MAX_DECISIONS; }
void sso_step_ex(sso_env* E, const float* act, float* obs, float* rew, uint8_t* done, sso_info* info, real* margins, double tol,
                 int32_t* near, int32_t* nnear, const int32_t* force, const int32_t* nforce, int cap, uint8_t* record,
                 const uint8_t* replay, int ntrace) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < E->num_envs; ++e) {
    decisions D;
    memset(&D, 0, sizeof D);
    if (margins) { margins[2 * e] = margins[2 * e + 1] = FAR_MARGIN; D.margin = margins + 2 * e; }
    if (near) { D.tol = (real)tol; D.near = near + (size_t)e * cap; D.cap = cap; }
    if (force) { D.force = force + (size_t)e * cap; D.nforce = nforce[e]; }
    D.ntrace = ntrace;
    if (record) D.record = record + (size_t)e * ntrace;
    if (replay) D.replay = replay + (size_t)e * ntrace;
    g_dec = &D;
    env_step(E, e, act + (size_t)e * NJ, obs + (size_t)e * OBS_DIM, rew + e, done + e, info + e);
    g_dec = 0;
    if (nnear) nnear[e] = D.nnear;
  }
}
/* sso_step with, per env, the outcome of the decisions force[e][0 .. nforce[e]) inverted */
void sso_step_forced(sso_env* E, const float* act, float* obs, float* rew, uint8_t* done, sso_info* info,
                     const int32_t* force, const int32_t* nforce, int cap) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < E->num_envs; ++e) {
    decisions D;
    memset(&D, 0, sizeof D);
    D.force = force + (size_t)e * cap; D.nforce = nforce[e];
    g_dec = &D;
    env_step(E, e, act + (size_t)e * NJ, obs + (size_t)e * OBS_DIM, rew + e, done + e, info + e);
    g_dec = 0;
  }
}
void sso_set_curriculum(sso_env* E, int c) {
  E->curriculum = c;
  for (int e = 0; e < E->num_envs; ++e) fill_window(E->e[e].prob, c, 0);
}
void sso_set_specialist(sso_env* E, int c) {
  E->curriculum = c;
  for (int e = 0; e < E->num_envs; ++e) fill_window(E->e[e].prob, c, 1);
}
void sso_set_sample_prob(sso_env* E, const double* p, int per_env) {
  for (int e = 0; e < E->num_envs; ++e)
    for (int k = 0; k < NCELL; ++k) E->e[e].prob[k] = (float)p[per_env ? (size_t)e * NCELL + (size_t)k : (size_t)k];
}
void sso_set_power(sso_env* E, double power) { E->power = (real)power; }
void sso_set_auto_reset(sso_env* E, int on) { E->auto_reset = on ? 1 : 0; }
/* env.create_temp_states() -> (yaw_size*pitch_size, 60) = (121, 60) hypothetical observations
 * (playground/train.py:247-257; stacked per env at common/envs_utils.py:573-578); PHYSICS.md section 8 */
void sso_create_temp_states(sso_env* E, float* out) {
  for (int e = 0; e < E->num_envs; ++e) {
    env_state tmp = E->e[e];
    float base[OBS_DIM];
    write_obs(E->M, &tmp, base);
    real roll, pitch, yaw;
    quat_rpy(tmp.quat, &roll, &pitch, &yaw);
    int n1 = tmp.n + 1 > NSTONE - 1 ? NSTONE - 1 : tmp.n + 1;
    for (int c = 0; c < NCELL; ++c) {
      float* o = out + ((size_t)e * NCELL + c) * OBS_DIM;
      memcpy(o, base, sizeof base);
      if (n1 > tmp.n) {
        place_stone(&tmp, n1, yaw_sample(c / NGRID), pitch_sample(c % NGRID), tmp.nn_dr, E->e[e].terrain[n1][4],
                    E->e[e].terrain[n1][5]);
        target_features(&tmp, tmp.terrain[n1], yaw, o + 55);
      }
    }
  }
}
/* packed state (STATE_DIM reals per env, layout documented in include/steppingstone.h) */
void sso_get_state(const sso_env* E, real* out) {
  for (int e = 0; e < E->num_envs; ++e) {
    const env_state* s = &E->e[e];
    real* o = out + (size_t)e * STATE_DIM;
    memcpy(o, s->pos, 3 * sizeof(real)); memcpy(o + 3, s->quat, 4 * sizeof(real)); memcpy(o + 7, s->vel, 6 * sizeof(real));
    memcpy(o + 13, s->q, NJ * sizeof(real)); memcpy(o + 34, s->qd, NJ * sizeof(real));
    o[55] = s->pot_prev; o[56] = s->z_init; o[57] = (real)(float)s->ep_ret; o[58] = s->nn_dr;
    o[185] = (real)(float)(s->ep_ret - (double)(float)s->ep_ret);
    o[59] = (real)s->n; o[60] = (real)s->count; o[61] = (real)s->elapsed;
    o[62] = (real)(s->rng_ctr & 0xFFFFu); o[63] = (real)(s->rng_ctr >> 16); o[64] = (real)s->flags;
    memcpy(o + 65, s->terrain, NSTONE * 6 * sizeof(real));
  }
}
void sso_set_state(sso_env* E, const real* in) {
  for (int e = 0; e < E->num_envs; ++e) {
    env_state* s = &E->e[e];
    const real* o = in + (size_t)e * STATE_DIM;
    memcpy(s->pos, o, 3 * sizeof(real)); memcpy(s->quat, o + 3, 4 * sizeof(real)); memcpy(s->vel, o + 7, 6 * sizeof(real));
    memcpy(s->q, o + 13, NJ * sizeof(real)); memcpy(s->qd, o + 34, NJ * sizeof(real));
    s->pot_prev = o[55]; s->z_init = o[56]; s->ep_ret = (double)o[57] + (double)o[185]; s->nn_dr = o[58];
    s->n = (int)o[59]; s->count = (int)o[60]; s->elapsed = (int)o[61];
    s->rng_ctr = (uint32_t)o[62] | ((uint32_t)o[63] << 16); s->flags = (int)o[64];
    memcpy(s->terrain, o + 65, NSTONE * 6 * sizeof(real));
  }
}
void sso_get_obs(const sso_env* E, float* obs) {
  for (int e = 0; e < E->num_envs; ++e) write_obs(E->M, &E->e[e], obs + (size_t)e * OBS_DIM);
}
/* benchmark action stream: Philox stream 1, ctr = 6 t + b, uniform in [-1,1) */
void sso_random_actions(const sso_env* E, uint64_t t, float* act) {
  uint32_t key[2] = {(uint32_t)E->seed, (uint32_t)(E->seed >> 32)};
  for (int e = 0; e < E->num_envs; ++e)
    for (int b = 0; b < 6; ++b) {
      uint32_t ctr[4] = {(uint32_t)(6 * t + b), 1u, (uint32_t)(E->env_offset + e), 0u}, r[4];
      philox4x32_10(ctr, key, r);
      for (int i = 0; i < 4; ++i) {
        int j = b * 4 + i;
        if (j < NJ) act[(size_t)e * NJ + j] = 2.f * ((float)(r[i] >> 8) * 5.9604644775390625e-08f) - 1.f;
      }
    }
}
int sso_real_size(void) { return (int)sizeof(real); }
int sso_state_dim(void) { return STATE_DIM; }

/* ---- debug entry points for the oracle's own validation tests (tests/test_oracle_dynamics.py) */
/* forward dynamics only: packed state of ONE env in, qdd[21] and a0[6] out (no contact) */
void sso_debug_aba(int kind, const real* packed, const real* tau_m, real* qdd, real* a0) {
  sso_env* E = sso_create(kind, 1, 0, 0);
  sso_set_state(E, packed);
  static work w;
  kinematics(E->M, &E->e[0], &w);
  aba(E->M, &E->e[0], tau_m, &w, qdd, a0);
  sso_destroy(E);
}
/* n substeps with fixed motor torques; state in/out; last foot report out (2 contact flags, 2 target flags) */
void sso_debug_substeps(sso_env* E, int e, const real* tau_m, int n, int* flags4) {
  foot_report fr;
  warm_state ws;
  memset(&fr, 0, sizeof fr);
  for (int k = 0; k < 8; ++k) ws.stone[k] = -1;
  for (int k = 0; k < n; ++k) substep(E->M, &E->e[e], tau_m, &fr, &ws);
  flags4[0] = fr.foot_contact[0]; flags4[1] = fr.foot_contact[1];
  flags4[2] = fr.foot_on_target[0]; flags4[3] = fr.foot_on_target[1];
}
/* ONE substep of env e with fixed motor torques (state advanced), with the contact stage's intermediate quantities
 * copied to *tap (layout: contact_tap above, mirrored by tests/oracle_lib.py).  `prior` substeps of the same control step run
 * before it (untapped), so that the tapped one is warm-started from them (prior = 0: the cold first substep of a step). */
void sso_debug_contact_after(sso_env* E, int e, const real* tau_m, int prior, contact_tap* tap) {
  foot_report fr;
  warm_state ws;
  memset(&fr, 0, sizeof fr);
  for (int k = 0; k < 8; ++k) ws.stone[k] = -1;
  for (int k = 0; k < prior; ++k) substep(E->M, &E->e[e], tau_m, &fr, &ws);
  g_tap = tap;
  substep(E->M, &E->e[e], tau_m, &fr, &ws);
  g_tap = 0;
}
void sso_debug_contact(sso_env* E, int e, const real* tau_m, contact_tap* tap) { sso_debug_contact_after(E, e, tau_m, 0, tap); }
int sso_tap_size(void) { return (int)sizeof(contact_tap); }
/* world position of every body (22 x 3) and rotation (22 x 9) for FK checks */
void sso_debug_fk(int kind, const real* packed, real* pos, real* rot) {
  sso_env* E = sso_create(kind, 1, 0, 0);
  sso_set_state(E, packed);
  static work w;
  kinematics(E->M, &E->e[0], &w);
  for (int b = 0; b < NB; ++b) {
    for (int i = 0; i < 3; ++i) pos[b * 3 + i] = w.pw[b][i];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) rot[b * 9 + i * 3 + k] = w.Rw[b].m[i][k];
  }
  sso_destroy(E);
}
