/*
 * mfg_hip.h -- C ABI of the MI355X (gfx950) mean-field-game hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI
 * layer: its hot path is the NumPy body of the methods of mfg_ac2.actor_critic /
 * ac_irl.AC_IRL.  Each entry point below replaces the batch-1 NumPy math of one of
 * those methods by one launch over B independent trajectories.  The Python classes
 * in discrete_mean_field_game_amd/ bind these symbols with ctypes and pass
 * tensor.data_ptr() of PyTorch-owned device memory.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is DEVICE memory unless the name ends in _host.
 *   - row-major; pi is [B,d] fp32, P is [B,d,d] fp32 (P[b][i][j] = prob. i -> j);
 *     theta and the critic weights w[F] live on the device as fp64 so that updates
 *     are stream ordered (no host round trip inside a rollout).
 *   - F = d(d+1)/2 + d + 1; w = [quadratic upper triangle row-major | linear | bias]
 *     (the order of mfg_ac2.py:325-344).
 *   - return 0 on success, a negative MFG_E* code otherwise; mfg_last_error() gives
 *     a thread-local message.  No allocation, no ownership transfer, no implicit
 *     synchronisation: all work is enqueued on `stream` (a hipStream_t).
 *   - all accumulations that feed rewards / TD errors / gradients are fp64.
 */
#ifndef MFG_HIP_H
#define MFG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mfg_stream_t; /* hipStream_t */

enum {
  MFG_OK = 0,
  MFG_EINVAL = -1,       /* bad argument (null pointer, d < 1, B < 0 ...) */
  MFG_ELAUNCH = -2,      /* HIP reported a launch/runtime error */
  MFG_EUNSUPPORTED = -3, /* shape outside the supported range (d > MFG_MAX_D) */
  MFG_EWORKSPACE = -4,   /* workspace too small */
  MFG_ERANGE = -5,       /* an earlier launch reported a numeric-range condition (mfg_status); sticky until cleared */
  MFG_ECOMM = -6         /* mfg_train_rollouts_dist: a launch / collective failed inside the episode loop and the RCCL communicator
                            has been ABORTED (ncclCommAbort, so that the peers fail instead of waiting): the handle is dead -- forget
                            it, do not destroy or use it again */
};

#define MFG_MAX_D 512

/* reward_kind for mfg_step_given_P / mfg_rollout */
enum {
  MFG_REWARD_MFG_AC2 = 0,   /* sum_i pi_i sum_j P_ij^2 (pi_j - pi_i)   mfg_ac2.py:257-287 */
  MFG_REWARD_SYNTHETIC = 1, /* -1/2 sum_i pi_i ||P_i||^2               mfg_synthetic.py:249-265 */
  MFG_REWARD_EXTERNAL = 2   /* reward supplied by the caller (IRL reward net, ac_irl.py:683) */
};

/* precision of the per-element policy math (softplus / sigmoid / digamma / log):
 *   F64   every transcendental in fp64 (matches the fp64 oracle to ~1e-12 on identical inputs)
 *   MIXED fp32 hardware transcendentals, fp64 only for the sums (score g within ~1e-6 relative;
 *         rewards, transitions, values and TD errors are unaffected: they never use these functions) */
enum { MFG_PRECISION_F64 = 0, MFG_PRECISION_MIXED = 1 };

/* flags for mfg_rollout */
enum {
  MFG_ROLLOUT_WRITE_P = 1,     /* materialise P[B,T,d,d] (IRL / generate_trajectories, ac_irl.py:762) */
  MFG_ROLLOUT_TD = 2,          /* also compute reward, delta, g per step (train); else env only */
  MFG_ROLLOUT_DISCOUNT_POW = 4, /* bootstrap with running gamma^t (ac_irl.py:691,710) instead of gamma */
  MFG_ROLLOUT_F64 = 8           /* MFG_PRECISION_F64 instead of the default MFG_PRECISION_MIXED */
};

const char* mfg_last_error(void);
int mfg_abi_version(void);

/* Lane mapping of the d = 21 / 15 mixed-precision sampling launches (mfg_rollout, mfg_sample_dirichlet, the mfg_train_rollout*
 * and mfg_train_episode_irl* families; hot loop of mfg_ac2.py:478-526, ac_irl.py:664-712).  Two kernels produce the SAME bits:
 * the packed one (three / four trajectories per wavefront, a lane per matrix row) and -- round 6, ABI 17 -- one trajectory per
 * wavefront with three / four lanes per matrix row, whose serial chain per wave is 1.65x / 2.2x shorter; the library takes the
 * second where the packed kernel leaves the SIMDs under-occupied (d = 21: <= 16 trajectories per CU = 4 096 on an MI355X;
 * d = 15: <= 16 per CU, <= 12 with MFG_ROLLOUT_WRITE_P, <= 4 for single-step launches -- measured).
 * mode: 0 by batch size (default), 1 always packed, 2 one trajectory per wave wherever it supports the launch.  Process-wide;
 * returns the previous mode.  A measurement / test hook (A/B timing, bit-identity tests): results never depend on it. */
int mfg_set_core_mapping(int mode);

/* One-time per-device setup (fits the 12 KB h(z) = psi(softplus z) sigmoid z table used by the mixed-precision
 * score, one tiny launch + one device synchronise).  Optional: the first TD call does it lazily; call it
 * explicitly before capturing launches into a hipGraph. */
int mfg_init(void);

/* Device status word.  The mixed-precision SAMPLING kernels form e^{theta (pi_j - pi_i - shift)} as a product of two fp32
 * factors e^{theta (pi_j - 1/2)} e^{-theta (pi_i + shift - 1/2)}; that is exact business as usual while
 * |theta| (1/2 + |shift|) <= 86 (theta ~ 130 at the reference's shift 0.16; the reference trains at theta ~ 9).  theta lives
 * on the device, so the host cannot check it before a launch: a sampling kernel that finds theta outside that range (or
 * not finite) sets MFG_STATUS_MIXED_RANGE in a host-visible status word and its outputs are NaN.  Every later call that
 * launches a mixed-precision SAMPLING kernel (sample / rollout / train entry points with MFG_PRECISION_MIXED) then fails
 * with MFG_ERANGE until mfg_clear_status() -- a diverged run stops with an error code instead of carrying NaNs.  The word is
 * one per CONTEXT (below; one per device and process for callers that never bind one): launches the condition does not concern (MFG_PRECISION_F64, kernels on given actions) are
 * not refused, so another model instance or thread on the device keeps working.  mfg_status() reads the word without
 * synchronising (synchronise the stream first to be sure a given launch has reported); it returns MFG_OK or MFG_ERANGE and
 * stores the bits in *bits_host (may be NULL). */
enum { MFG_STATUS_MIXED_RANGE = 1 };
int mfg_status(unsigned* bits_host);
int mfg_clear_status(void);

/* Contexts ("no global mutable state except an opaque mfg_ctx*").  The mutable state a launch touches -- the
 * status word above, optionally an RCCL communicator -- belongs to a context; everything else the library keeps is immutable
 * (the h(z) table, per device) or passed in by the caller (workspaces).  Model: the CURRENT context of the calling thread,
 * like the current device of the HIP runtime -- mfg_ctx_bind(ctx) makes `ctx` the context of every entry point this thread
 * calls from then on (sampling launches report into ITS status word and are refused on ITS sticky bits; mfg_status /
 * mfg_clear_status act on it); mfg_ctx_bind(NULL) returns to the device's default context, which is what a thread that never
 * binds uses (the process-wide word of ABI <= 14: existing callers keep their behaviour).  Two model instances with a context
 * each cannot stop each other: a diverged run poisons only its own word.
 *   mfg_ctx_create      on the current device (initialises the device's table like mfg_init); one context per model instance
 *   mfg_ctx_destroy     also unbinds it from the calling thread and destroys an adopted communicator
 *   mfg_ctx_bind        thread-local, no device work; fails with MFG_EINVAL if ctx belongs to another device than the current
 *   mfg_ctx_status / mfg_ctx_clear_status   explicit-context forms of mfg_status / mfg_clear_status
 *   mfg_ctx_adopt_comm  hand a communicator from mfg_dist_init to the context (lifetime); mfg_ctx_comm returns it
 * Captured hipGraphs: the status-word address is baked into the captured kernel arguments (the capturing thread's context);
 * a replay reports into that word but is not refused on it -- check mfg_ctx_status after synchronising a replay. */
typedef struct mfg_ctx mfg_ctx_t;
int mfg_ctx_create(mfg_ctx_t** ctx_out);
int mfg_ctx_destroy(mfg_ctx_t* ctx);
int mfg_ctx_bind(mfg_ctx_t* ctx);
mfg_ctx_t* mfg_ctx_current(void);
int mfg_ctx_status(mfg_ctx_t* ctx, unsigned* bits_host);
int mfg_ctx_clear_status(mfg_ctx_t* ctx);
int mfg_ctx_adopt_comm(mfg_ctx_t* ctx, void* comm);
void* mfg_ctx_comm(mfg_ctx_t* ctx);

/* Host-side query: multiprocessor count and gcnArchName of the current device. */
int mfg_device_info(int* cu_count_host, char* arch_host, int arch_len);

/* Index of the quadratic feature pi_i*pi_j inside phi / w (host helper, integer
 * bookkeeping of itertools.combinations_with_replacement, mfg_ac2.py:333). */
int64_t mfg_feature_index(int i, int j, int d);
int64_t mfg_num_features(int d);

/* Bytes of scratch the gradient reductions need for N transitions of dimension d.  The first 64 bytes are a control
 * block (completion counter of the in-kernel finalisation): zero the workspace ONCE after allocating it
 * (hipMemset); every call leaves the control block zero again, so one workspace sized for the largest N serves
 * calls with any smaller N.  A workspace must not be shared by calls that can
 * run concurrently on different streams. */
size_t mfg_workspace_bytes(int64_t N, int d);

/* a9: pi0[b,:] = mat_pi0[idx[b],:]                       (mfg_ac2.py:466-469) */
int mfg_gather_start(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d,
                     float* pi0, mfg_stream_t stream);

/* a9 on the device: the per-episode start-state draw  idx_row = randint(num_start_samples)  (mfg_ac2.py:466, ac_irl.py:655)
 * for B trajectories, from the counter-based generator of the action sampler instead of the host's np.random:
 *   Philox4x32-10, key = seed, counter = (0xFFFFFFFF, step, low 32 bits of the global trajectory id traj_offset + b,
 *   bits 32..47 of that id), x = first output word;  idx[b] = floor(x * num_start / 2^32)  (multiply-shift).
 * `step` = the Philox step of the episode's first env step, so an episode's draw and its actions share one counter space
 * (element id 0xFFFFFFFF is outside the action draws' i*d + j) and nothing but (seed, step, trajectory id) enters: the
 * draw does not depend on launch geometry or world size, and a resumed run redraws the same states.
 * Writes idx[B] (int32) and / or the gathered rows pi0[B,d] = mat_pi0[idx[b],:]; either may be NULL (mat_pi0 is only
 * needed for pi0).  The training entry points below draw the same rows inside their first kernel. */
int mfg_draw_start(const float* mat_pi0, int64_t num_start, int64_t B, int d, uint64_t seed, uint32_t step,
                   uint64_t traj_offset, int32_t* idx, float* pi0, mfg_stream_t stream);

/* a1: alpha[b,i,j] = softplus(theta (pi_j - pi_i - shift)) and its theta-derivative
 * (mfg_ac2.py:219-234; ac_irl.py:573-588).  Outputs fp64 [B,d,d]; either may be NULL. */
int mfg_alpha(const float* pi, int64_t B, int d, const double* theta, double shift, double* alpha,
              double* alpha_deriv, mfg_stream_t stream);

/* a2 (normalisation half): P = y / rowsum(y) with zeros -> 1e-20   (mfg_ac2.py:244-249).
 * Used when gamma variates are injected by the host (batch-1 reference RNG parity). */
int mfg_dirichlet_from_gamma(const float* y, int64_t B, int d, float* P, mfg_stream_t stream);

/* a1+a2: P[b] ~ prod_i Dirichlet(alpha[b,i,:] * alpha_scale)  (mfg_ac2.py:211-254).
 * Counter-based RNG: Philox4x32-10 keyed by `seed`, counter = (element i*d+j, step,
 * global trajectory id traj_offset + b, draw) so results do not depend on launch
 * geometry or world size. */
int mfg_sample_dirichlet(const float* pi, int64_t B, int d, const double* theta, double shift,
                         double alpha_scale, uint64_t seed, uint32_t step, uint64_t traj_offset, int precision,
                         float* P, mfg_stream_t stream);

/* Raw Philox4x32-10 blocks for counters (c0 = first_ctr + n, c1, c2, c3): out[n,4] u32.  Test hook
 * for bit-exact RNG parity. */
int mfg_philox_raw(uint64_t seed, uint32_t first_ctr, uint32_t c1, uint32_t c2, uint32_t c3, int64_t n,
                   uint32_t* out, mfg_stream_t stream);

/* a3+a4: pi_next = P^T pi, reward[b] (fp32) per reward_kind     (mfg_ac2.py:497-499).
 * reward may be NULL (transition only, ac_irl.py:679).  The HBM-bound kernel of the path: P is read once
 * (16-byte streaming loads when P is 16-byte aligned); with 16-byte aligned pi_next / reward, large batches
 * (d = 21, 15: >= ~10^5 transitions; d = 128, 256: >= 16 384) write their outputs as batched device-scope
 * bursts.  Results do not depend on which form runs.  pi_next must not alias pi. */
int mfg_step_given_P(const float* pi, const float* P, int64_t B, int d, int reward_kind, float* pi_next,
                     float* reward, mfg_stream_t stream);

/* a5: value[b] = phi(pi_b) . w (fp64 out)                     (mfg_ac2.py:290-322) */
int mfg_value(const float* pi, const double* w, int64_t B, int d, double* value, mfg_stream_t stream);
/* a5: phi[b,:] materialised (fp64 [B,F])                       (mfg_ac2.py:325-344) */
int mfg_features(const float* pi, int64_t B, int d, double* phi, mfg_stream_t stream);

/* a7: g[b] = sum_ij (-psi(alpha_ij) + psi(sum_j alpha_ij) + ln P_ij) alpha'_ij  (mfg_ac2.py:347-381).
 * pi_alpha is the state the concentrations are computed from (the reference reads the alpha left
 * behind by the previous sample_action; pass pi itself for the normal case).  Zeros of P count as
 * 1e-100 (mfg_ac2.py:369); P is not modified. */
int mfg_score(const float* pi_alpha, const float* P, int64_t B, int d, const double* theta, double shift,
              int precision, double* g, mfg_stream_t stream);

/* a5-a8 on given transitions: delta[b] = r + gamma_or_discount V(pi') - V(pi), g[b] as mfg_score,
 * and the batch sums G = [ sum_b delta_b phi(pi_b) (F) | sum_b delta_b g_b | sum_b r_b | B ] (fp64,
 * F+3 entries; accumulate != 0 adds onto the existing contents of G).  (mfg_ac2.py:501-522) */
int mfg_td_pg_accumulate(const float* pi, const float* pi_next, const float* P, const float* reward,
                         const double* w, const double* theta, double shift, double gamma_or_discount,
                         int64_t B, int d, int precision, double* delta, double* g, double* G, int accumulate,
                         void* workspace, size_t workspace_bytes, mfg_stream_t stream);

/* a6/a8: w += lr_critic * G_w / count ; theta += lr_actor * G_theta / count  (mfg_ac2.py:511-522).
 * count = G[F+2] (number of transitions summed, after any all-reduce).  If reward_acc is not NULL the mean
 * reward of the update, G[F+1]/count, is added to *reward_acc (total_reward += reward, mfg_ac2.py:526) so the
 * episode return never needs a host round trip. */
int mfg_apply_update(const double* G, int d, double lr_critic, double lr_actor, double* w, double* theta,
                     double* reward_acc, mfg_stream_t stream);

/* Fused T-step rollout with fixed (theta, w): a1-a5, a7 per step, state kept on chip.
 *   pi_traj[B,T+1,d] fp32 (pi_traj[:,0] = pi0; may be NULL for env-only rollouts), pi_last[B,d] = final state
 *   (contiguous, may be NULL), reward[B,T] fp32, delta[B,T], g[B,T] fp64,
 *   P_out[B,T,d,d] fp32 when MFG_ROLLOUT_WRITE_P; G as in mfg_td_pg_accumulate over all B*T
 *   transitions when MFG_ROLLOUT_TD.  first_step offsets the RNG step counter.
 *   reward_kind = MFG_REWARD_EXTERNAL (the IRL step: the reward network needs the sampled action first,
 *   ac_irl.py:679-691): `reward` is not touched, delta = gamma V(pi') - V(pi) WITHOUT the reward, G must be NULL;
 *   finish with mfg_grad_accumulate(add_reward = 1) once the rewards exist. */
int mfg_rollout(const float* pi0, int64_t B, int d, int T, const double* theta, double shift,
                double alpha_scale, const double* w, double gamma, int reward_kind, uint64_t seed,
                uint32_t first_step, uint64_t traj_offset, int flags, float* pi_traj, float* pi_last,
                float* reward, double* delta, double* g, float* P_out, double* G, int accumulate,
                void* workspace, size_t workspace_bytes, mfg_stream_t stream);

/* a9, one training update per episode (update_every = 'rollout'): start-state gather (mfg_ac2.py:466-469: row idx[b] of
 * the table mat_pi0[num_start,d], read inside the rollout kernel -- no separate gather launch), the fused T-step TD
 * rollout, the batch sums over all B*T transitions.  flags: MFG_ROLLOUT_F64 / MFG_ROLLOUT_DISCOUNT_POW as for
 * mfg_rollout, plus MFG_TRAIN_APPLY: also apply w += lr_critic G_w/N, theta += lr_actor G_theta/N and add the mean
 * reward to *reward_acc (if not NULL) inside the launch that finishes the sums -- a single-GPU update is then 2-3
 * launches.  Without MFG_TRAIN_APPLY G is complete on return: multi-GPU jobs all-reduce it and call mfg_apply_update.
 * Outputs as for mfg_rollout (pi_traj, reward, delta, g are required; pi_last may be NULL).
 * idx == NULL: the start rows are DRAWN inside the rollout kernel (mfg_draw_start with step = first_step) -- batched runs
 * need no host RNG, no index upload and, with several ranks, no broadcast in front of an episode. */
#define MFG_TRAIN_APPLY 16
int mfg_train_rollout(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d, int T, double* theta,
                      double shift, double alpha_scale, double* w, double gamma, int reward_kind, uint64_t seed,
                      uint32_t first_step, uint64_t traj_offset, int flags, double lr_critic, double lr_actor,
                      float* pi_traj, float* pi_last, float* reward, double* delta, double* g, double* G,
                      double* reward_acc, void* workspace, size_t workspace_bytes, mfg_stream_t stream);

/* a9, the episode loop itself (mfg_ac2.py:460-526 with one update per episode): `episodes` training updates issued back to
 * back from native code -- per episode k: start states drawn in the rollout kernel (Philox step first_step + k T), fused
 * T-step TD rollout, batch sums, update with the reference's learning-rate schedule evaluated natively:
 *   e = first_episode + k;  constant != 0: lr_critic, lr_actor;  else lr_critic / (e+1), lr_actor / ((e+1) ln ln (e+20))
 * (mfg_ac2.py:511-522; pass the 1-indexed episode number for ac_irl.py:697-708), mean reward of the update added to
 * reward_acc[k] (may be NULL).  flags as for mfg_train_rollout (MFG_TRAIN_APPLY implied).  Single GPU: between two
 * episodes of a multi-GPU job sits an all-reduce, which stays with the caller (mfg_train_rollout per episode).
 * The output buffers hold the LAST episode's values on return. */
int mfg_train_rollouts(const float* mat_pi0, int64_t num_start, int64_t B, int d, int T, int64_t episodes, int64_t first_episode,
                       int constant, double* theta, double shift, double alpha_scale, double* w, double gamma, int reward_kind,
                       uint64_t seed, uint32_t first_step, uint64_t traj_offset, int flags, double lr_critic, double lr_actor,
                       float* pi_traj, float* pi_last, float* reward, double* delta, double* g, double* G, double* reward_acc,
                       void* workspace, size_t workspace_bytes, mfg_stream_t stream);

/* a9 on several GPUs: the update cycle of a rank without an update launch.  A multi-GPU update is
 *   rollout | batch sums | all-reduce(G) | w, theta += lr G / count              (mfg_ac2.py:511-522 for the global batch)
 * and its last step needs nothing but G: this entry point takes the all-reduced sums of the PREVIOUS update as G_pending
 * (NULL: none) and applies them while the rollout kernel stages its weights (d <= 64; a separate out-of-place launch
 * above): the parameters the rollout runs with -- and (theta_out, w_out), written by one block -- are
 *   theta + lr_actor_pending G_pending[F] / count,  w + lr_critic_pending G_pending[:F] / count,  count = G_pending[F+2],
 * bit for bit what mfg_apply_update would have produced in place; *reward_acc_pending += G_pending[F+1] / count.  The
 * outputs must be OTHER buffers than theta / w (blocks that start later still read the old values): callers ping-pong
 * two parameter sets.  Then as mfg_train_rollout without MFG_TRAIN_APPLY: G holds this rank's sums of the new update on
 * return (G may alias G_pending: it is written by a later launch).  A rank's cycle is rollout -> sums -> all-reduce, and
 * mfg_apply_update only once, after the last episode. */
int mfg_train_rollout_deferred(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d, int T,
                               const double* theta, const double* w, const double* G_pending, double lr_critic_pending,
                               double lr_actor_pending, double* reward_acc_pending, double* theta_out, double* w_out, double shift,
                               double alpha_scale, double gamma, int reward_kind, uint64_t seed, uint32_t first_step,
                               uint64_t traj_offset, int flags, float* pi_traj, float* pi_last, float* reward, double* delta,
                               double* g, double* G, void* workspace, size_t workspace_bytes, mfg_stream_t stream);

/* a9 on several GPUs, natively: the episode loop of mfg_train_rollouts with ONE RCCL all-reduce of G per update issued by
 * this library on the caller's stream -- per episode [rollout kernel (applies the previous update, draws the start
 * states) | batch sums | ncclAllReduce(G)], then mfg_apply_update once after the last episode; no interpreter and no
 * framework call between the updates of a multi-GPU job.  RCCL is resolved at run time from the librccl.so the process
 * already holds (PyTorch's); MFG_EUNSUPPORTED where there is none.
 *   mfg_dist_unique_id : ncclGetUniqueId on ONE rank; ship the 128 bytes to the others (e.g. torch.distributed.broadcast)
 *   mfg_dist_init      : ncclCommInitRank on EVERY rank (collective; the current device is the rank's GPU)
 *   mfg_dist_all_reduce: in-place SUM of n doubles (the exchange of a per-step update; test hook)
 *   mfg_dist_abort     : ncclCommAbort -- frees the communicator WITHOUT the collective hand-shake of mfg_dist_destroy (a rank
 *                        whose peers failed or timed out during set-up; ABI 17)
 *   mfg_train_rollouts_dist: arguments as mfg_train_rollouts (B = this rank's shard, traj_offset = its first global
 *     trajectory id, G / count summed over the ranks) plus the second parameter set (theta_alt, w_alt) the updates
 *     ping-pong through; on return the parameters are in (theta, w) and identical on every rank.  MFG_ECOMM: the call
 *     aborted the communicator (see the error codes). */
typedef struct mfg_rccl_id { char bytes[128]; } mfg_rccl_id_t; /* ncclUniqueId */
int mfg_dist_available(void); /* 1 if librccl's entry points resolve in this process, else 0 (no error recorded) */
int mfg_dist_unique_id(mfg_rccl_id_t* id_host);
int mfg_dist_init(const mfg_rccl_id_t* id_host, int nranks, int rank, void** comm_out);
int mfg_dist_destroy(void* comm);
int mfg_dist_abort(void* comm);
int mfg_dist_all_reduce(void* comm, double* G, int64_t n, mfg_stream_t stream);
int mfg_train_rollouts_dist(void* comm, const float* mat_pi0, int64_t num_start, int64_t B, int d, int T, int64_t episodes,
                            int64_t first_episode, int constant, double* theta, double* w, double* theta_alt, double* w_alt,
                            double shift, double alpha_scale, double gamma, int reward_kind, uint64_t seed, uint32_t first_step,
                            uint64_t traj_offset, int flags, double lr_critic, double lr_actor, float* pi_traj, float* pi_last,
                            float* reward, double* delta, double* g, double* G, double* reward_acc, void* workspace,
                            size_t workspace_bytes, mfg_stream_t stream);

/* f1 (IRL): reward[b] = r_net(state_b, action_b), the reward network of networks.py:46-81 evaluated for B
 * transitions in one launch (ac_irl.py:683 evaluates it with batch 1 per env step).  fp32.  Weight layouts are
 * PyTorch's: conv1_w [k1*k1], conv2_w [f2][k2*k2], fc3_w [n3][d*d*f2] with the input index (pixel*f2 + channel)
 * (TF's NHWC flatten, networks.py:67), fc4_w [n4][n3+d] (inputs = [fc3 activations, state], networks.py:72),
 * out_w [n4].  keep_prob < 1 applies inverted dropout after fc3 and fc4 like tf.contrib.layers.dropout in
 * training mode (the reference leaves it on when the net serves as the RL reward); masks come from Philox keyed
 * by (seed, sample_offset + b).  Supported: d <= 32, f2 <= 2, n3, n4 <= 32, odd k1, k2 <= 7. */
int mfg_reward_net_forward(const float* state, const float* action, int64_t B, int d, int k1, int f2, int k2, int n3,
                           int n4, const float* conv1_w, const float* conv1_b, const float* conv2_w,
                           const float* conv2_b, const float* fc3_w, const float* fc3_b, const float* fc4_w,
                           const float* fc4_b, const float* out_w, const float* out_b, float keep_prob, uint64_t seed,
                           uint64_t sample_offset, float* reward, mfg_stream_t stream);

/* f1 (IRL), reward learning on the device: ONE call = one AC_IRL.update_reward (ac_irl.py:804-846) -- batch assembly,
 * forward of networks.py:46-81 over the sampled demonstration and generated transitions, the max-ent loss of
 * ac_irl.py:390-413
 *     L = -(1/demo_divisor) sum r_demo + log( (1/n_gen) sum_traj exp( sum_t r_gen ) ) [+ l1_l2(fc3_w) + l1_l2(fc4_w)],
 * its gradient and one tf.train.AdamOptimizer step (ac_irl.py:417-418;  lr_t = lr sqrt(1-beta2^t)/(1-beta1^t),
 * m = beta1 m + (1-beta1) g, v = beta2 v + (1-beta2) g^2, p -= lr_t m / (sqrt v + eps)), in two launches and no host
 * synchronisation.
 *   params / adam_m / adam_v [NP] fp32: ONE flat buffer per quantity, tensors in the order
 *     conv1_w [k1*k1] | conv1_b [1] | conv2_w [f2][k2*k2] | conv2_b [f2] | fc3_w [n3][f2*d*d] | fc3_b [n3] |
 *     fc4_w [n4][n3+d] | fc4_b [n4] | out_w [n4] | out_b [1]      (layouts of mfg_reward_net_forward;
 *     mfg_reward_net_param_offsets gives the 10 start offsets and NP as offsets_host[0..10]).
 *   The trajectories live in device-resident stores: *_state [rows, steps, d], *_action [rows, steps, d, d] fp32; the
 *   batch is the store rows demo_rows_host[n_demo] and gen_rows_host[n_gen] (HOST arrays, copied into the kernel
 *   arguments: no index upload; each <= MFG_RN_TRAIN_MAX_TRAJ).  Transition n of the batch (demonstrations first, row
 *   major over (trajectory, step)) draws its dropout masks from Philox key `seed`, counter (unit, 3 | 4, n, block 0) when
 *   keep_prob < 1, like mfg_reward_net_forward with sample_offset 0.
 *   flags & MFG_RN_TRAIN_GRAD_ONLY: stop at the gradient (grad [NP], required) -- the multi-GPU form: all-reduce grad,
 *   then mfg_reward_net_adam.  grad may also be given without the flag (gradient written AND update applied).
 *   stats (may be NULL) [4] <- loss, first term, second term, regulariser, evaluated at the weights BEFORE the update
 *   (what sess.run([r_train_op, loss, ...]) returns, ac_irl.py:846).
 *   workspace: mfg_reward_net_train_workspace_bytes(..., (n_demo + n_gen) * steps); contents are scratch.
 * Supported shapes as mfg_reward_net_forward; batch: (n_demo + n_gen) * steps <= 2048 transitions and
 * (n_demo + n_gen) * steps * (1 + n3) * 4 B <= 60 KB (MFG_EUNSUPPORTED beyond; the reference's batch is 150 transitions). */
#define MFG_RN_TRAIN_MAX_TRAJ 64
enum { MFG_RN_TRAIN_GRAD_ONLY = 1 };
int64_t mfg_reward_net_num_params(int d, int k1, int f2, int k2, int n3, int n4);
int mfg_reward_net_param_offsets(int d, int k1, int f2, int k2, int n3, int n4, int64_t* offsets_host);
size_t mfg_reward_net_train_workspace_bytes(int d, int k1, int f2, int k2, int n3, int n4, int64_t n_transitions);
int mfg_reward_net_train_step(float* params, float* adam_m, float* adam_v, int d, int k1, int f2, int k2, int n3, int n4,
                              const float* demo_state, const float* demo_action, const int32_t* demo_rows_host, int n_demo,
                              const float* gen_state, const float* gen_action, const int32_t* gen_rows_host, int n_gen, int steps,
                              int demo_divisor, float keep_prob, int l1l2, uint64_t seed, double lr, double beta1, double beta2,
                              double eps, int64_t adam_step, int flags, float* grad, float* stats, void* workspace,
                              size_t workspace_bytes, mfg_stream_t stream);
int mfg_reward_net_adam(float* params, float* adam_m, float* adam_v, const float* grad, int64_t n, double lr, double beta1,
                        double beta2, double eps, int64_t adam_step, mfg_stream_t stream);

/* f3: backward value recursion of the mfg_synthetic variant: V^n = r^n + P^n V^{n+1}, r^n_i = -1/2 ||P^n_i||^2,
 * V^T = 0 (mfg_synthetic.py:768-774) for P[B,T,d,d] -> V[B,T+1,d] (fp64), plus per (b,n) the consistency
 * metrics of evaluate_synthetic (diff_l1 = sum_ij |P_ij - value_ij|, :776-790) and, if diff_jsd != NULL, of
 * evaluate_synthetic_JSD (sum_i JSD(P_i, implied row i), entries <= 0 -> 1e-100, :858-880). */
int mfg_backward_value(const float* P, int64_t B, int T, int d, double* V, double* diff_l1, double* diff_jsd,
                       mfg_stream_t stream);

/* a6/a8 batch sums on their own: G (+)= [sum delta phi(pi) | sum delta g | sum reward | N] over the B*T samples
 * n = (b, s), pi_n = pi + b*stride_b + s*d (stride_b = (T+1)*d for a pi_traj, d for a plain [B,d] batch).
 * add_reward != 0: first delta_n <- delta_n + reward_n, written back -- the IRL step, where the rollout
 * (reward_kind = MFG_REWARD_EXTERNAL) leaves delta = gamma V(pi') - V(pi) and the reward network (ac_irl.py:683)
 * runs in between (:691).  Workspace as for mfg_td_pg_accumulate(B*T). */
int mfg_grad_accumulate(const float* pi, int64_t stride_b, double* delta, const double* g, const float* reward, int64_t B,
                        int T, int d, int add_reward, double* G, int accumulate, void* workspace, size_t workspace_bytes,
                        mfg_stream_t stream);

/* mfg_grad_accumulate (accumulate = 0) followed by mfg_apply_update, with the update applied inside the kernel that
 * finishes the sums (single GPU; mfg_ac2.py:511-522 / ac_irl.py:697-708 for the batch): the IRL step
 * rollout(EXTERNAL) -> reward net -> this call is three launches. */
int mfg_grad_apply(const float* pi, int64_t stride_b, double* delta, const double* g, const float* reward, int64_t B, int T,
                   int d, int add_reward, double* G, double lr_critic, double lr_actor, double* w, double* theta,
                   double* reward_acc, void* workspace, size_t workspace_bytes, mfg_stream_t stream);

/* a9, native inner loop of train() with the reference's per-step updates (mfg_ac2.py:478-525) on ONE GPU: for
 * s < T: sample P ~ policy(pi), pi' = P^T pi, r, delta = r + gamma V(pi') - V(pi), g (a1-a7, one fused launch);
 * batch sums over the B trajectories (a6/a8); w += lr_critic G_w/B, theta += lr_actor G_theta/B,
 * *reward_acc += mean reward (if not NULL); pi <- pi'.  3T+... launches issued back to back from native code: no
 * host round trip, no interpreter between dependent small kernels.  pi_io [B,d]: start states in, final states out;
 * pi_scratch [B,d]; reward/delta/g [B] hold the last step's values on return; G [F+3]; workspace as for
 * mfg_td_pg_accumulate(B).  Philox steps first_step .. first_step+T-1.  Multi-GPU jobs keep the per-step
 * mfg_rollout(T=1) + all-reduce + mfg_apply_update sequence instead. */
int mfg_train_episode(float* pi_io, float* pi_scratch, int64_t B, int d, int T, double* theta, double shift,
                      double alpha_scale, double* w, double gamma, int reward_kind, uint64_t seed, uint32_t first_step,
                      uint64_t traj_offset, int precision, double lr_critic, double lr_actor, float* reward, double* delta,
                      double* g, double* G, double* reward_acc, void* workspace, size_t workspace_bytes,
                      mfg_stream_t stream);

/* a9, the episode loop with the reference's per-step updates (mfg_ac2.py:460-526): `episodes` x [start states drawn on the
 * device into pi_io (mfg_draw_start, step = first_step + k T) | mfg_train_episode], learning rates per episode as in
 * mfg_train_rollouts, reward_acc[k] += mean reward of every step's update of episode k (may be NULL).  Single GPU. */
int mfg_train_episodes(const float* mat_pi0, int64_t num_start, float* pi_io, float* pi_scratch, int64_t B, int d, int T,
                       int64_t episodes, int64_t first_episode, int constant, double* theta, double shift, double alpha_scale,
                       double* w, double gamma, int reward_kind, uint64_t seed, uint32_t first_step, uint64_t traj_offset,
                       int precision, double lr_critic, double lr_actor, float* reward, double* delta, double* g, double* G,
                       double* reward_acc, void* workspace, size_t workspace_bytes, mfg_stream_t stream);

/* Weights of the reward network of networks.py:46-81 as device pointers (layouts as for mfg_reward_net_forward). */
typedef struct mfg_reward_net {
  int k1, f2, k2, n3, n4;
  const float *conv1_w, *conv1_b, *conv2_w, *conv2_b, *fc3_w, *fc3_b, *fc4_w, *fc4_b, *out_w, *out_b;
  float keep_prob; /* < 1: inverted dropout after fc3 and fc4 (the reference leaves it on, ac_irl.py:683) */
} mfg_reward_net_t;

/* a9 (IRL flavour), native inner loop of AC_IRL.train with the reference's per-step updates (ac_irl.py:664-712) on ONE
 * GPU.  For s < T, all launches issued back to back from native code (no interpreter between the dependent kernels):
 *   sample P ~ policy(pi), pi' = P^T pi, g, delta0 = discount V(pi') - V(pi)     (one fused launch, P materialised, :674-679)
 *   r = r_net(pi, P)                                                              (mfg_reward_net_forward, :683)
 *   delta = r + delta0; batch sums; w += lr_critic G_w/B, theta += lr_actor G_theta/B; *reward_acc += mean r   (:691-708)
 *   discount *= gamma; pi <- pi'                                                   (:710-711)
 * Where the matrix-core reward-network kernel serves (d = 21 / 15, the reference's layer geometry, n_fc3 <= 16) an env step is
 * TWO launches: [sampling + transition + score, theta formed by every wavefront from the previous step's partial sums; the
 * grid's last blocks reduce those sums and publish w, theta, G, the return] | [reward network + TD error from the updated w
 * + this step's partial sums]; one row reduction closes the episode.  Otherwise three (step | network | row reduction).
 * The dropout masks of step s use the Philox key  rn_seed ^ ((rn_call0 + s + 1) * 0x9E3779B97F4A7C15)  (mod 2^64) and the
 * sample counter rn_sample_offset + b -- the keys the host class would have passed to T separate
 * mfg_reward_net_forward calls numbered rn_call0 + 1 ... .  pi_io [B,d]: start states in, final states out; pi_scratch
 * [B,d]; P [B,d,d], reward / delta / g [B] hold the last step's values on return; G [F+3]; workspace as for
 * mfg_td_pg_accumulate(B).  Philox steps first_step .. first_step+T-1 for the actions. */
int mfg_train_episode_irl(float* pi_io, float* pi_scratch, int64_t B, int d, int T, double* theta, double shift,
                          double alpha_scale, double* w, double gamma, uint64_t seed, uint32_t first_step,
                          uint64_t traj_offset, int precision, double lr_critic, double lr_actor,
                          const mfg_reward_net_t* net_host, uint64_t rn_seed, uint64_t rn_call0, uint64_t rn_sample_offset,
                          float* P, float* reward, double* delta, double* g, double* G, double* reward_acc, void* workspace,
                          size_t workspace_bytes, mfg_stream_t stream);

/* The same episode with the start states DRAWN from the table mat_pi0 [num_start,d] (the draw of mfg_draw_start at
 * step = first_step, trajectory ids traj_offset + b; ac_irl.py:655) -- inside the first step kernel where the two-launch flow
 * serves, by a launch of its own otherwise.  pi_out [B,d]: final states (output only); the rest as above. */
int mfg_train_episode_irl_draw(const float* mat_pi0, int64_t num_start, float* pi_out, float* pi_scratch, int64_t B, int d, int T,
                               double* theta, double shift, double alpha_scale, double* w, double gamma, uint64_t seed,
                               uint32_t first_step, uint64_t traj_offset, int precision, double lr_critic, double lr_actor,
                               const mfg_reward_net_t* net_host, uint64_t rn_seed, uint64_t rn_call0, uint64_t rn_sample_offset,
                               float* P, float* reward, double* delta, double* g, double* G, double* reward_acc, void* workspace,
                               size_t workspace_bytes, mfg_stream_t stream);

/* a9 (IRL flavour), one update per episode (the batched form of ac_irl.py:664-712 with theta, w fixed over the episode) on
 * ONE GPU, issued natively: [fused T-step rollout: start states drawn in the kernel (idx == NULL) or gathered, actions
 * materialised, delta0 = discount V(pi') - V(pi), scores] | [reward network over all B*T transitions, states read in place
 * from pi_traj] | [batch sums with delta = delta0 + r folded in (+ row reduction)] (+ update with MFG_TRAIN_APPLY):
 * 3-4 launches, no framework call in between.  flags: MFG_ROLLOUT_DISCOUNT_POW | MFG_ROLLOUT_F64 | MFG_TRAIN_APPLY.
 * Dropout masks: Philox key rn_key, sample counter rn_sample_offset + b T + t (ONE mfg_reward_net_forward call over the
 * [B*T] transitions).  pi_traj [B,T+1,d], P [B,T,d,d], reward [B*T] (the network's output), delta / g [B*T] (delta final),
 * G [F+3]; workspace as for mfg_td_pg_accumulate(B*T).  *reward_acc += mean reward over the B*T transitions. */
int mfg_train_rollout_irl(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d, int T, double* theta,
                          double shift, double alpha_scale, double* w, double gamma, uint64_t seed, uint32_t first_step,
                          uint64_t traj_offset, int flags, double lr_critic, double lr_actor, const mfg_reward_net_t* net_host,
                          uint64_t rn_key, uint64_t rn_sample_offset, float* pi_traj, float* pi_last, float* P, float* reward,
                          double* delta, double* g, double* G, double* reward_acc, void* workspace, size_t workspace_bytes,
                          mfg_stream_t stream);

/* f1 (optional importance weights, ac_irl.py:270-289 calc_pdf_action, :324-379 calc_z): log-density of the
 * product-Dirichlet policy for N (state, action) pairs under K policies theta_k (device array):
 *   out[n*K + k] = sum_i log Dirichlet(P_n[i,:] ; a_i),  a_ij = max(alpha_floor, alpha_scale * softplus(theta_k x_ij)).
 * The reference evaluates the density itself and divides by a normaliser c before multiplying 15*d factors
 * (:365-373); the log-space value is exact where that under/overflows.  alpha_floor = 1+1e-6 and alpha_scale = 1
 * reproduce calc_z; alpha_floor = 0 reproduces calc_pdf_action.  Entries of P below p_floor are raised to it
 * (p_floor = 0: ln 0 = -inf, density 0, like tf.distributions.Dirichlet.prob).  fp64 out. */
int mfg_policy_logpdf(const float* pi, const float* P, int64_t N, int d, const double* thetas, int K, double shift,
                      double alpha_scale, double alpha_floor, double p_floor, double* out, mfg_stream_t stream);

/* a11: out[b] = JSD(p_b, q_b), zeros -> 1e-100, inputs renormalised like scipy.stats.entropy
 * (mfg_ac2.py:546-563).  fp64 out. */
int mfg_jsd(const float* p, const float* q, int64_t B, int d, double* out, mfg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MFG_HIP_H */
