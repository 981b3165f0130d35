// Core actor-critic kernels: sampling (a1+a2), transition/reward (a3+a4), value (a5), TD error (a6),
// score (a7) -- T steps with fixed (theta, w), state kept on chip.  Included by the translation units
// that instantiate them (mfg_core_small.hip, mfg_core_large_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/mfg_hip.h"
#include "mfg_device.h"

namespace mfg {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / WAVE;

struct CoreArgs {
  const float* pi0;         // [B,d]; or, with start_idx != NULL, the start-state table [num_start,d]
  const int32_t* start_idx; // [B] rows of the table (start-state gather folded into the kernel, mfg_ac2.py:466-469)
  int64_t num_start;        // rows of the table (start_idx != NULL): indices are clamped into [0, num_start)
  int start_draw;           // != 0: pi0 is the table and the row of trajectory b is DRAWN in the kernel (start_draw_row,
                            // mfg_device.h: Philox keyed by seed, first_step, global trajectory id); start_idx is ignored
  const float* pi_alpha;    // GIVEN: state the concentrations are computed from (NULL -> pi0)
  union {
    const float* P_in;      // GIVEN: [B,d,d]
    float* pi_start_out;    // STEP variant (sampling): [B,d] the start states are also written here (drawn in the kernel), or NULL
  };
  const float* pi_next_in;  // GIVEN: [B,d] (may be NULL when no delta is wanted)
  const float* reward_in;   // external reward [B*T] or NULL
  const double* theta;
  const double* w;          // NULL -> no value / delta
  double shift, alpha_scale, gamma;
  int64_t B;
  int d, T, reward_kind, discount_pow;
  uint64_t seed;
  uint32_t first_step;
  int step_nrows;      // != 0: STEP variant (see below); < 0: without rows to reduce (an episode's first env step)
  uint64_t traj_offset;
  float* pi_traj;      // [B,T+1,d] or NULL
  float* pi_next_out;  // [B,d] final state or NULL
  float* reward_out;   // [B,T] or NULL
  double* delta;       // [B,T] or NULL
  double* g;           // [B,T] or NULL
  float* P_out;        // [B,T,d,d] or NULL
  const float4* htab;  // h(z) cubic table (mixed precision TD), see mfg_device.h
  unsigned* status;    // device address of the host-visible status word (mfg_status), or NULL
  union {
    double* part_rows;   // SUMS variant (T == 1, one tile per block): partial rows [ntiles][F+3] of the batch sums, or NULL
    double* step_G;      // STEP variant: [F+3] batch sums of the previous env step (output)
  };
  // Deferred update (packed kernel, mfg_train_rollout_deferred): pend_G != NULL = the all-reduced batch sums [F+3] of the
  // PREVIOUS update, not applied yet.  Every block forms the updated parameters while it stages them (theta, w above are
  // the parameters BEFORE that update); block 0 also writes them to theta_out / w_out (other buffers than theta / w: blocks
  // that start later still read the old values) and adds the update's mean reward to *pend_reward_acc.
  union {
    const double* pend_G;
    const double* step_rows;  // STEP variant: [step_nrows][F+3] partial rows of the previous env step; column F once more,
                              // contiguous, in the step_nrows doubles in FRONT of them (step_rows[-step_nrows .. -1])
  };
  double pend_lr_c, pend_lr_a;
  double* w_out;
  double* theta_out;
  double* pend_reward_acc;
  // IRL env step (STEP variant of the packed kernel, mfg_train_episode_irl): the batch sums of the PREVIOUS env step are still
  // `step_nrows` partial rows [F+3] (left by the reward-network launch).  Every sampling wave adds up column F itself
  // (rows_column_sum over the contiguous copy in front of the rows) and forms theta = updated_param(*theta, pend_lr_a, sum, 1 / B)
  // -- the one parameter sampling needs; the grid's last core_step_red_blocks(F + 3) blocks add up ALL columns (a wave per
  // column) and publish the update:
  // step_G[k], w_out[k] (in place: nobody reads the critic weights in this launch -- the TD error is formed in the
  // reward-network launch that follows), *theta_out (another slot than *theta), *pend_reward_acc.  The row reduction is off
  // the critical path: it runs under the sampling blocks instead of between two launches.
  // These arguments SHARE storage with arguments the variant never reads (step_nrows sits in the padding behind first_step,
  // step_rows = pend_G, step_G = part_rows, pi_start_out = P_in, the sample count is B, the number of reducing blocks follows from d):
  // a longer argument block moves the
  // hidden launch arguments and, with them, the register allocation of every other instantiation of the kernel (measured on the
  // headline kernel: two more spilled registers).
#ifdef MFG_TIMING
  unsigned long long* dbg;  // timing variant only (tools/phase_timing.py): s_memtime stamps of block 0, wave 0
#endif
};
#ifdef MFG_TIMING
#define MFG_STAMP(k) if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0 && s < 4) a.dbg[s * 16 + (k)] = __builtin_amdgcn_s_memtime();
#define MFG_STAMP0(k) if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[(k)] = __builtin_amdgcn_s_memtime();  // slots 8..15 of row 0
// every block's entry / exit time (wave 0), from slot 64 on: how the blocks of a launch spread over its duration
#define MFG_STAMPB(k) if (a.dbg && threadIdx.x == 0) a.dbg[64 + 2 * blockIdx.x + (k)] = __builtin_amdgcn_s_memtime();
#else
#define MFG_STAMP(k)
#define MFG_STAMP0(k)
#define MFG_STAMPB(k)
#endif

// blocks behind the sampling blocks of a STEP launch: a wave per column of the FO-entry rows
__host__ __device__ constexpr int core_step_red_blocks(int FO) { return (FO + WAVES - 1) / WAVES; }

int set_error(int code, const char* msg);  // records mfg_last_error() (defined in mfg_kernels.hip)

// IRL env step: reward network + the batch sums of the TD update over the same samples in one launch
// (mfg_reward_net.hip; used by mfg_train_episode_irl)
struct RnSums {
  const double* delta0;  // [B] discount V(pi') - V(pi) from the step kernel
  const double* g;       // [B] scores
  double* delta_out;     // [B] delta = delta0 + r (may alias delta0)
  double* part_rows;     // [max_rows][F+3]
  int64_t max_rows;
  // td_w != NULL (after reward_net_sums_td_ready() said yes): delta0 is NOT read -- the matrix-core kernel forms
  // td_gamma V(state_next) - V(state) itself from the critic weights td_w [F]
  const double* td_w;
  const float* state_next;  // [B,d]
  double td_gamma;
  double* col_f;            // [max_rows] column F of the rows, contiguous (see RewardNetArgs)
};
// true: a reward_net_forward_sums call of this shape runs the matrix-core SUMS kernel with the TD error formed in the kernel
// (d = 21 / 15 at the reference's layer geometry, n_fc3 <= 16, aligned FC3 weights, one row per block fits max_rows)
bool reward_net_sums_td_ready(int64_t B, int d, int k1, int f2, int k2, int n3, int n4, const float* fc3_w, int64_t max_rows);
int reward_net_forward_sums(const float* state, const float* action, int64_t B, int d, int k1, int f2, int k2, int n3, int n4,
                            const float* conv1_w, const float* conv1_b, const float* conv2_w, const float* conv2_b,
                            const float* fc3_w, const float* fc3_b, const float* fc4_w, const float* fc4_b,
                            const float* out_w, const float* out_b, float keep_prob, uint64_t seed, uint64_t sample_offset,
                            float* reward, const RnSums* sums, int* rows_out, mfg_stream_t stream, int state_T = 0);

// launchers defined in mfg_core_small.hip / mfg_core_large_*.hip; return 0 or MFG_EUNSUPPORTED
int launch_core_small(const CoreArgs& a, bool sample, bool td, bool fast, int num_cus, hipStream_t st);
// d = 21, mixed-precision sampling launches of batches that under-fill the machine: one trajectory per wavefront, three lanes
// per matrix row (mfg_core_row3.hip).  core_row3_wanted: whether launch_core_small hands a launch of this shape to it.
bool core_row3_wanted(const CoreArgs& a, bool sample, bool td, bool fast, int num_cus);
int launch_core_row3(const CoreArgs& a, bool td, int num_cus, hipStream_t st);
int core_mapping_set(int mode);  // 0 by batch size, 1 packed, 2 one trajectory per wave; returns the previous mode
int launch_core_large_f64(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st);
int launch_core_large_mixed(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st);

// row of a.pi0 that holds the start state of local trajectory b: drawn in the kernel, gathered through start_idx, or b itself
__device__ __forceinline__ int64_t core_src_row(const CoreArgs& a, int64_t b) {
  if (a.start_draw) return start_draw_row(a.seed, a.first_step, a.traj_offset + (uint64_t)b, a.num_start);
  return a.start_idx ? start_row(a.start_idx[b], a.num_start) : b;
}

__device__ __forceinline__ double reward_term(int kind, double pii, double pj, double p) {
  // contribution of element (i,j) BEFORE the factor pi_i (kind 0) / -0.5 pi_i (kind 1)
  return kind == MFG_REWARD_MFG_AC2 ? (pj - pii) * p * p : p * p;
}

// Per-element policy quantities: concentration alpha, its theta-derivative alpha', and the gamma
// sampler state for shape alpha*alpha_scale.  x = pi_j - pi_i - shift.
template <bool FAST>
struct PolicyElem {
  double al_d, ad_d;  // strict mode
  float al_f, ad_f;   // mixed mode
  float psi_ad;       // mixed mode, TD: psi(alpha) alpha' = x h(theta x), looked up AT SETUP so that the table load's
                      // latency hides behind the sampling arithmetic instead of stalling the score update
  GammaState gs;
};

// pj / pi: state entries the concentration is computed from (x = pj - pi - shift).
template <bool SAMPLE, bool TD, bool FAST>
__device__ __forceinline__ void policy_setup(PolicyElem<FAST>& e, const CoreArgs& a, double theta, const ThetaSplit& ts,
                                             float pj, float pi) {
  if (FAST) {
    float x, zh, zl, sg;
    theta_times_x(ts, pj, pi, x, zh, zl);
    softplus_sigmoid_fast(zh, zl, e.al_f, sg);
    e.ad_f = x * sg;
    if (TD) e.psi_ad = x * htab_eval(a.htab, x, ts.thn);  // fp32 product is ample for a table lookup (|dh/dz| < 1)
    if (SAMPLE) gamma_setup_hot(e.gs, e.al_f, (float)a.alpha_scale, (9.0f / BM_K2) * (float)a.alpha_scale);
  } else {
    const double x = (double)pj - (double)pi - a.shift;
    double sg;
    softplus_sigmoid(theta * x, e.al_d, sg);
    e.ad_d = x * sg;
    if (SAMPLE) gamma_setup_hot(e.gs, (float)(e.al_d * a.alpha_scale));
  }
}

// Mixed-precision sampling kernels: e^z = Ej * Fi (see exp_f64arg in mfg_device.h).
template <bool SAMPLE, bool TD>
//   pis = pi_i + shift of the element's ROW, formed once per row by the caller: x costs one subtraction per element.
__device__ __forceinline__ void policy_setup_sep(PolicyElem<true>& e, const CoreArgs& a, const ThetaSplit& ts, float pj, float Ej,
                                                 float pis, float Fi) {
  const float x = pj - pis;
  float sg;
  softplus_sigmoid_e(Ej * Fi, e.al_f, sg);
  e.ad_f = x * sg;
  // (round 4: the lookup through a raw buffer resource -- 32-bit offset instead of a 64-bit address per element -- measured
  //  0.6 % SLOWER than the plain global load: not kept)
  if (TD) e.psi_ad = x * htab_eval(a.htab, x, ts.thn);
  if (SAMPLE) gamma_setup_hot(e.gs, e.al_f, (float)a.alpha_scale, (9.0f / BM_K2) * (float)a.alpha_scale);
}

// Fold one finished element into the row sums / score.  v = gamma variate (SAMPLE) or stored probability.
template <bool SAMPLE, bool TD, bool FAST>
__device__ __forceinline__ void policy_accumulate(const PolicyElem<FAST>& e, const float4* __restrict__ htab, float th,
                                                  float v, double& A, double& D, double& gacc) {
  if (!TD) return;
  if (FAST) {
    const float lnv = (!SAMPLE && v == 0.0f) ? (float)(LOG_ZERO_P * INV_LN2) : __builtin_amdgcn_logf(v);  // log2 units
    A += (double)e.al_f;
    D += (double)e.ad_f;
    gacc += (double)__builtin_fmaf(lnv, e.ad_f, -e.psi_ad);
  } else {
    const double lnv = (!SAMPLE && v == 0.0f) ? LOG_ZERO_P : log((double)v);
    A += e.al_d;
    D += e.ad_d;
    gacc = fma(-digamma_pos(e.al_d) + lnv, e.ad_d, gacc);
  }
}

// Contributions of one finished element to (A, D, g) without folding them: the sampling loops add the terms of a quad
// (up to four elements) in the working precision -- fp32 in mixed mode -- and fold ONE fp64 add per quantity per quad
// (a cvt + add_f64 pair costs as much issue time as four fp32 adds).
template <bool FAST>
struct PolicyTerms {
  using T = typename std::conditional<FAST, float, double>::type;
  T al, ad, gt;
};
template <bool SAMPLE, bool FAST>
__device__ __forceinline__ PolicyTerms<FAST> policy_terms(const PolicyElem<FAST>& e, const float4* __restrict__ htab, float th,
                                                          float v) {
  PolicyTerms<FAST> o;
  if constexpr (FAST) {
#ifdef MFG_ABL_LNY
    const float lnv = v;
#else
    // log2 units: the h table is stored divided by ln 2 (mfg_device.h), the sums are scaled by ln 2 once per row / step
    const float lnv = (!SAMPLE && v == 0.0f) ? (float)(LOG_ZERO_P * INV_LN2) : __builtin_amdgcn_logf(v);
#endif
    o.al = e.al_f;
    o.ad = e.ad_f;
    o.gt = __builtin_fmaf(lnv, e.ad_f, -e.psi_ad);
  } else {
    const double lnv = (!SAMPLE && v == 0.0f) ? LOG_ZERO_P : log((double)v);
    o.al = e.al_d;
    o.ad = e.ad_d;
    o.gt = (-digamma_pos(e.al_d) + lnv) * e.ad_d;
  }
  return o;
}

// NE (1..4) matrix elements from ONE Philox block (keyed by elem[0]).  Elements 2h, 2h+1 share a Box-Muller pair.
// Written so that the two chains of a pair sit in one basic block and interleave (measured on gfx950: a single
// dependent chain issues one VALU instruction per ~5 cycles per SIMD, two independent chains one per ~2): no branch on
// the hot path -- the rare exact-acceptance / small-shape continuations hide behind one wave-uniform test per pair.
//   Per element e: pj / ej = state entry and E_j of its column, pai / Fi = state entry (SEP: state entry + shift, see
//   policy_setup_sep) and F_i of its ROW (ej, Fi unused
//   unless SEP), elem = its element id (Philox counter of its own continuation draws), valid = whether it exists
//   (lanes past the last column compute on clamped inputs and are masked out).
//   Out: y = gamma variate (0 when !valid); al / ad / gt = its alpha, alpha', score term (0 when !valid; TD only).
//   qstep = the env step that keys the QUAD's block (block 0 of elem[0]); step = the env step that keys the elements' own
//   continuation draws.  They differ only where a lane runs a row's trailing single element through the quad code
//   (k_core_row3: the pair of sample_tail1 is keyed by the even step); sample_elems_g passes the same value twice.
//   ALLVALID: every element is treated as existing whatever `valid` says (no per-element selects / ballots); the caller
//   discards what it does not want -- the continuation of a discarded element may still run (harmless, wave-uniform branch).
template <int NE, bool TD, bool FAST, bool SEP, bool ALLVALID = false>
__device__ __forceinline__ void sample_elems_gq(const CoreArgs& a, double theta, const ThetaSplit& ts, const float* pj,
                                                const float* ej, const float* pai, const float* Fi, const uint32_t* elem,
                                                const bool* valid, uint32_t qstep, uint32_t step, uint64_t traj, float* y,
                                                typename PolicyTerms<FAST>::T* al, typename PolicyTerms<FAST>::T* ad,
                                                typename PolicyTerms<FAST>::T* gt) {
  QuadRand q;
  quad_rand(q, a.seed, elem[0], qstep, traj);
  // the two Box-Muller pairs one after the other (two interleaved chains each; four at once cost 36 spilled VGPRs at
  // the 128-register cap of the small-d kernel and bought nothing at full occupancy)
#ifndef MFG_QUAD_WIDTH
#define MFG_QUAD_WIDTH 2  // elements whose chains are interleaved: 2 (one Box-Muller pair) or 4 (the whole quad)
#endif
  constexpr int PW = MFG_QUAD_WIDTH;
#pragma unroll
  for (int h = 0; PW * h < NE; ++h) {
    const int n2 = (NE - PW * h) >= PW ? PW : (NE - PW * h);
    PolicyElem<FAST> pe[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      if (u < n2) {
        const int e = PW * h + u;
        if constexpr (SEP) policy_setup_sep<true, TD>(pe[u], a, ts, pj[e], ej[e], pai[e], Fi[e]);
        else policy_setup<true, TD, FAST>(pe[u], a, theta, ts, pj[e], pai[e]);
      }
    }
    float xn[PW], v[PW];
    bool sure[PW];
#pragma unroll
    for (int u = 0; u < PW; u += 2) {
      const int hp = (PW * h + u) >> 1;  // Box-Muller pair index inside the quad
      sure[u] = true;
      if (u + 1 < PW) sure[u + 1] = true;
#ifdef MFG_ABL_BM
      xn[u] = (q.radu[hp] - 0.5f) * 2.0f;
      if (u + 1 < PW) xn[u + 1] = (q.radu[hp] - 0.5f) * q.ang[hp];
#else
      if (u < n2) {
        // radius in units of K = sqrt(2 ln 2) (GammaState, mfg_device.h): sqrt(-log2 u), the sign is a source modifier
        const float rad = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(q.radu[hp]));
        xn[u] = rad * __builtin_amdgcn_cosf(q.ang[hp]);
        if (u + 1 < PW) xn[u + 1] = rad * __builtin_amdgcn_sinf(q.ang[hp]);
      }
#endif
    }
    // "some lane of the wave needs the exact path": the compare masks of the pair, OR-ed as scalars (gamma_try_mask)
    uint64_t coldm = 0;
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      if (u < n2) {
        const int e = PW * h + u;
        uint64_t cm;
        if (quad_kbits(e) == 16) v[u] = gamma_try_mask<16>(pe[u].gs, xn[u], q.kf[e], cm);
        else v[u] = gamma_try_mask<12>(pe[u].gs, xn[u], q.kf[e], cm);
        y[e] = pe[u].gs.dd * v[u];
        if constexpr (ALLVALID) coldm |= cm;
        else coldm |= cm & __builtin_amdgcn_ballot_w64(valid[e]);
      }
    }
    const bool any_cold = coldm != 0;
    if (__builtin_expect(any_cold, 0)) {  // wave-uniform, ~2 % of the pairs at the reference policies
#pragma unroll
      for (int u = 0; u < PW; ++u) {
        const int e = PW * h + u;
        if (u < n2 && (ALLVALID || valid[e])) {
          const bool k16 = quad_kbits(e) == 16;
          if (k16) (void)gamma_try<16>(pe[u].gs, xn[u], q.kf[e], sure[u]);   // (the per-lane flags, recomputed off the hot path)
          else (void)gamma_try<12>(pe[u].gs, xn[u], q.kf[e], sure[u]);
          if (!sure[u] || pe[u].gs.small)
            y[e] = gamma_fix(pe[u].gs, xn[u], q.kf[e], k16 ? TryConst<16>::kscale : TryConst<12>::kscale, sure[u], v[u], a.seed,
                             elem[e], step, traj);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < PW; ++u) {
      if (u < n2) {
        const int e = PW * h + u;
        if (TD) {
          const PolicyTerms<FAST> t = policy_terms<true, FAST>(pe[u], a.htab, ts.th, y[e]);
          al[e] = (ALLVALID || valid[e]) ? t.al : 0;
          ad[e] = (ALLVALID || valid[e]) ? t.ad : 0;
          gt[e] = (ALLVALID || valid[e]) ? t.gt : 0;
        }
        if (!ALLVALID && !valid[e]) y[e] = 0.0f;
      }
    }
  }
}

template <int NE, bool TD, bool FAST, bool SEP>
__device__ __forceinline__ void sample_elems_g(const CoreArgs& a, double theta, const ThetaSplit& ts, const float* pj,
                                               const float* ej, const float* pai, const float* Fi, const uint32_t* elem,
                                               const bool* valid, uint32_t step, uint64_t traj, float* y,
                                               typename PolicyTerms<FAST>::T* al, typename PolicyTerms<FAST>::T* ad,
                                               typename PolicyTerms<FAST>::T* gt) {
  sample_elems_gq<NE, TD, FAST, SEP>(a, theta, ts, pj, ej, pai, Fi, elem, valid, step, step, traj, y, al, ad, gt);
}

// Small-d wrapper: NE neighbouring elements of ONE row (all valid); ys / as / ds / gs RETURN the sums of the NE elements in
// the working precision.  (They used to be added onto zero-initialised sums of the caller: `0.0f + x` is not foldable --
// it turns -0 into +0 -- and cost four v_add_f32 per quad.)
template <int NE, bool TD, bool FAST, bool SEP>
__device__ __forceinline__ void sample_elems(const CoreArgs& a, double theta, const ThetaSplit& ts, const float* pj,
                                             const float* ej, float pai, float Fi, uint32_t elem0, uint32_t step,
                                             uint64_t traj, float* y, float& ys, typename PolicyTerms<FAST>::T& as,
                                             typename PolicyTerms<FAST>::T& ds, typename PolicyTerms<FAST>::T& gs) {
  using TT = typename PolicyTerms<FAST>::T;
  float pa[NE], fi[NE];
  uint32_t el[NE];
  bool ok[NE];
  TT al[NE], ad[NE], gt[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    pa[e] = pai;
    fi[e] = Fi;
    el[e] = elem0 + (uint32_t)e;
    ok[e] = true;
  }
  sample_elems_g<NE, TD, FAST, SEP>(a, theta, ts, pj, ej, pa, fi, el, ok, step, traj, y, al, ad, gt);
  ys = y[0];
  if (TD) {
    as = al[0];
    ds = ad[0];
    gs = gt[0];
  }
#pragma unroll
  for (int e = 1; e < NE; ++e) {
    ys += y[e];
    if (TD) {
      as += al[e];
      ds += ad[e];
      gs += gt[e];
    }
  }
}

// The ONE trailing element of a row whose length is 1 mod 4 (d = 21): its Philox block feeds a Box-Muller PAIR and the
// element needs one normal.  The pair is keyed by the EVEN step (step & ~1): the even step takes the cosine normal and the
// first acceptance integer -- exactly what sample_elems<1> draws -- and leaves the sine normal and the second integer in
// (cxn, ckf) for the odd step that follows, which then skips the block, the field extraction and the radius / angle
// (~65 of the ~140 instructions of this element, every second step).  A launch that STARTS on an odd step has nothing
// carried and recomputes the even step's block: the trajectories do not depend on how a rollout is cut into launches.
// The element's own continuation draws (exact acceptance, small shapes) are keyed by its id and the real step as before.
template <bool TD, bool FAST, bool SEP>
__device__ __forceinline__ void sample_tail1(const CoreArgs& a, double theta, const ThetaSplit& ts, float pj, float ej, float pai,
                                             float Fi, uint32_t elem, uint32_t step, uint64_t traj, bool& have, float& cxn,
                                             float& ckf, float& y, typename PolicyTerms<FAST>::T& al,
                                             typename PolicyTerms<FAST>::T& ad, typename PolicyTerms<FAST>::T& gt) {
  PolicyElem<FAST> pe;
  if constexpr (SEP) policy_setup_sep<true, TD>(pe, a, ts, pj, ej, pai, Fi);
  else policy_setup<true, TD, FAST>(pe, a, theta, ts, pj, pai);
  const bool odd = (step & 1u) != 0;  // (wave-uniform)
  float xn, kf;
  if (odd && have) {
    xn = cxn;
    kf = ckf;
    have = false;
  } else {
    QuadRand q;
    quad_rand(q, a.seed, elem, step & ~1u, traj);
    const float rad = __builtin_amdgcn_sqrtf(-__builtin_amdgcn_logf(q.radu[0]));
    const float xc = rad * __builtin_amdgcn_cosf(q.ang[0]), xs = rad * __builtin_amdgcn_sinf(q.ang[0]);
    xn = odd ? xs : xc;
    kf = odd ? q.kf[1] : q.kf[0];
    cxn = xs;
    ckf = q.kf[1];
    have = !odd;
  }
  static_assert(quad_kbits(0) == 16 && quad_kbits(1) == 16, "both integers of the first pair are 16-bit");
  uint64_t cm;
  float v = gamma_try_mask<16>(pe.gs, xn, kf, cm);
  y = pe.gs.dd * v;
  if (__builtin_expect(cm != 0, 0)) {
    bool sure = true;
    (void)gamma_try<16>(pe.gs, xn, kf, sure);
    if (!sure || pe.gs.small) y = gamma_fix(pe.gs, xn, kf, TryConst<16>::kscale, sure, v, a.seed, elem, step, traj);
  }
  if (TD) {
    const PolicyTerms<FAST> t = policy_terms<true, FAST>(pe, a.htab, ts.th, y);
    al = t.al;
    ad = t.ad;
    gt = t.gt;
  }
}

// V(pi) = phi(pi).w, one wavefront per trajectory, lanes own columns c, rows i <= c.  For fixed i the
// weights w[k(i,c)] are contiguous in c, so the loads are coalesced (w is L2 resident).
__device__ __forceinline__ double value_wave(const float* pis, const double* __restrict__ w, int d, int lane) {
  const int Q = d * (d + 1) / 2;
  double acc = 0.0;
  for (int c = lane; c < d; c += WAVE) {
    const double pc = (double)pis[c];
    double col = 0.0;
    for (int i = 0; i <= c; ++i) col = fma(w[feat_idx(i, c, d)], (double)pis[i], col);
    acc = fma(pc, col + w[Q + c], acc);
  }
  acc = wave_sum(acc);
  return acc + w[Q + d];
}

// ---------------------------------------------------------------------------------------------
// small d (d <= 64): G = 64/d trajectories per wavefront, lane = (trajectory t, row i).
// LDS per block: wl[F] (critic weights, fp64), tile[TB][d][dp] (gamma variates, then P), pis / pin / pal [TB][d].
// ---------------------------------------------------------------------------------------------
// (The batch sums sum delta phi are NOT accumulated here: doing it per step in LDS cost 0.49 ms of a 2.5 ms rollout;
// the separate k_grad_* pass over pi_traj / delta costs ~0.05 ms.)
// D > 0: d is a compile-time constant (constant trip counts / strides); D == 0: generic runtime d.
// Registers: the mixed-precision training kernel needs ~165 VGPRs; capped at 128 (4 waves / SIMD) it spills ~46 of them
// and runs 1-8 % SLOWER at every batch size than at 168 (3 waves / SIMD, no spills): round-2 measurement, d = 21,
// B = 4096 .. 65536.  LDS (36 KB / block) would allow 4 blocks per CU.
#ifndef MFG_CORE_SMALL_WAVES
#define MFG_CORE_SMALL_WAVES 3  // waves per SIMD the mixed-precision kernel is register-capped for (168 VGPRs)
#endif
// SUMS (per-step updates, T == 1; compile-time d): the block also leaves the batch sums of ITS trajectories' transitions
// -- [sum delta phi(pi) | sum delta g | sum r | count], the augmented-vector fp64-MFMA form of k_grad_mfma_small -- as one
// partial row in a.part_rows: the separate gradient kernel (a launch, a re-read of pi / delta / g / reward, a
// fence-and-last-block finish: 10.6 us per env step at B = 4 096) shrinks to the row reduction.  Register budget of two
// waves per SIMD (the three-wave cap spills 31 registers here); used while all tiles are resident at that occupancy.
// STEP: 0 plain; 1 IRL env step with the previous step's partial rows (CoreArgs::step_rows); 2 an IRL episode's FIRST env step
// (nothing to reduce; the start states are also written to pi_start_out)
template <bool SAMPLE, bool TD, bool FAST, int D, bool SUMS = false, int STEP = 0>
#ifndef MFG_CORE_SMALL_WAVES_F64
#define MFG_CORE_SMALL_WAVES_F64 3  // strict precision: 227 registers wanted; at 168 the third wave still pays (5.95 -> 5.80 ms at the bench shape)
#endif
__global__ __launch_bounds__(BLOCK, SUMS ? 2 : (FAST ? MFG_CORE_SMALL_WAVES : MFG_CORE_SMALL_WAVES_F64)) void k_core_small(CoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  MFG_STAMP0(8)
  MFG_STAMPB(0)
  const int d = D ? D : a.d;
  const int dd = d * d, dp = d | 1, T = a.T;
  const int G = WAVE / d, TB = WAVES * G;
  const int Q = d * (d + 1) / 2, F = Q + d + 1;
  const bool want_v = TD && a.w != nullptr;
  // Value function: V = sum_{i<=k} U_ik pi_i pi_k + b.pi + c.  Lane (t, i) owns the terms of "its" entry i:
  //   generic: column i of the upper triangle, k <= i  -- 1 .. d terms per lane, the wave waits for the longest;
  //   CIRC (compile-time odd d, the reference's 21 and 15): the unordered pairs {i, i+m mod d}, m = 0 .. (d-1)/2 --
  //   every pair exactly once, (d+1)/2 terms on EVERY lane, fixed trip count (unrolled, all LDS reads in flight).
  //   Its weights sit in LDS as wc[m][i] = w[k(min, max)], the state as a doubled vector pn2[0 .. 2d) so that
  //   pn2[i + m] needs no modulo.  (The triangular loop cost 12 % of the d = 21 training rollout.)
  constexpr bool CIRC = SAMPLE && D > 0 && (D & 1);
  constexpr int H = (D + 1) / 2;
  const int pnw = CIRC ? 2 * d : d;  // floats per trajectory in pin
  double* wl = reinterpret_cast<double*>(smem_raw);
  // SAMPLE: per state entry k a 16-byte line {pi_k as fp64, 1 / S_k of row k as fp32}: the column pass (transition + reward
  // sums) fetches both with one broadcast read per row
  double* pis64 = wl + (want_v ? ((F + 1) & ~1) : 0);  // (16-byte aligned)
  double* red = pis64 + (SAMPLE ? 2 * TB * d : 0);  // [TB][4][d]: per-lane terms of reward / score / V(next) / V(start), summed by lane 0 / 1
  float* tile = reinterpret_cast<float*>(red + 4 * TB * d);
  float* pis = tile + TB * d * dp;
  float* pex = pis + TB * d;       // SAMPLE, mixed: E_j = e^{theta pi_j} (separable exponential, mfg_device.h).  DIRECTLY behind
                                   // pis: the quad loop reads pi_j and E_j of four columns from ONE base address (TB d <= 252
                                   // floats apart: inside the offset field of ds_read2_b32)
  float* pin = pex + TB * d;       // [TB][pnw]
  float* pal = pin + TB * 2 * d;
  double* scal = reinterpret_cast<double*>(pal + TB * d + ((TB * d) & 1));  // SUMS: (delta, g, r) per trajectory of the tile
  double* tot = scal + TB * 3;                           // [TB][8]: even / odd partial sums of the per-trajectory sums
  float* pst = reinterpret_cast<float*>(tot + TB * 8);   // [TB][pnw]: the rollout's START state (doubled for CIRC): its value is
                                                         // evaluated inside step 0, next to V of the next state
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int t = lane / d, i = lane - t * d;
  // blocks that walk the tiles (STEP: the grid's last red_blocks blocks reduce the previous env step's partial rows instead)
  auto nblk_f = [&]() -> unsigned { return STEP == 1 ? gridDim.x - (unsigned)core_step_red_blocks(F + 3) : gridDim.x; };  // (read where it is used)
  if constexpr (STEP == 1) {
    if (blockIdx.x >= nblk_f()) {
      const int64_t k = (int64_t)(blockIdx.x - nblk_f()) * WAVES + wv;
      const int64_t FO = F + 3;
      if (k >= FO) return;
      double old_val = 0.0;  // read first, under the row reads
      if (lane == 0) {
        if (k < F) old_val = a.w_out[k];
        else if (k == F) old_val = *a.theta;
        else if (k == F + 1 && a.pend_reward_acc) old_val = *a.pend_reward_acc;
      }
      const double gk = rows_column_sum(a.step_rows, a.step_nrows, FO, k, lane);
      if (lane == 0) {
        const double inv = 1.0 / (double)a.B;
        a.step_G[k] = gk;
        if (k < F) a.w_out[k] = updated_param(old_val, a.pend_lr_c, gk, inv);
        else if (k == F) *a.theta_out = updated_param(old_val, a.pend_lr_a, gk, inv);
        else if (k == F + 1 && a.pend_reward_acc) *a.pend_reward_acc = old_val + gk * inv;
      }
      return;
    }
  }
  float pi_first = 0.0f;
  if constexpr (STEP == 1) {  // (the first tile's start state, see below: here in flight under the row reads of theta's update)
    const int64_t b0f = (int64_t)blockIdx.x * TB;
    const int tlf = wv * G + t;
    if (b0f < a.B) {
      const int64_t bf = b0f + ((t < G && b0f + tlf < a.B) ? tlf : 0);
      pi_first = a.pi0[core_src_row(a, bf) * d + i];
    }
  }
  // deferred update of the previous episode (CoreArgs::pend_G): applied here, on the fly, with the arithmetic of
  // k_apply_update (updated_param); a sum over no samples (count 0) leaves the parameters alone
  const bool pend = STEP != 1 && a.pend_G != nullptr && a.pend_G[F + 2] > 0.0;
  const double pinv = pend ? 1.0 / a.pend_G[F + 2] : 0.0;
  double theta = pend ? updated_param(*a.theta, a.pend_lr_a, a.pend_G[F], pinv) : *a.theta;
  if constexpr (STEP == 1) {
    theta = updated_param(theta, a.pend_lr_a, rows_column_sum(a.step_rows - a.step_nrows, a.step_nrows, 1, 0, lane), 1.0 / (double)a.B);
  }
  const ThetaSplit ts = theta_split(theta, a.shift);
  const float inv_d = 1.0f / (float)d;
  // mixed sampling kernels: separable e^z = E_j F_i (mfg_device.h), in range while |theta| (1/2 + |shift|) <= 86; beyond
  // that the launch reports MFG_STATUS_MIXED_RANGE and its outputs are NaN (precision 'f64' has no such limit).
  constexpr bool sep = SAMPLE && FAST;
  if (sep) report_sep_range(a.status, theta, a.shift);
  // first tile's start state: the load is issued BEFORE the weight staging so that its latency (L2 / HBM, ~1 us) overlaps
  // the staging instead of stalling the first use (a T = 1 launch spent 4 300 of its 20 900 cycles waiting for it)
  if constexpr (STEP != 1) {
    const int64_t b0f = (int64_t)blockIdx.x * TB;
    const int tlf = wv * G + t;
    if (b0f < a.B) {
      const int64_t bf = b0f + ((t < G && b0f + tlf < a.B) ? tlf : 0);
      pi_first = a.pi0[core_src_row(a, bf) * d + i];
    }
  }
  auto w_now = [&](int k) -> double { return pend ? updated_param(a.w[k], a.pend_lr_c, a.pend_G[k], pinv) : a.w[k]; };
  if (want_v) {
    if (CIRC) {
      for (int k = tid; k < H * d; k += BLOCK) {
        const int m = k / d, ii = k - m * d;
        int kk = ii + m;
        if (kk >= d) kk -= d;
        wl[k] = w_now(feat_idx(ii < kk ? ii : kk, ii < kk ? kk : ii, d));
      }
      for (int k = Q + tid; k < F; k += BLOCK) wl[k] = w_now(k);
    } else {
      for (int k = tid; k < F; k += BLOCK) wl[k] = w_now(k);
    }
  }
  if (STEP != 1 && a.pend_G != nullptr && blockIdx.x == 0) {
    // block 0 publishes the updated parameters (out of place) and books the update's mean reward
    if (a.w_out && a.w)
      for (int k = tid; k < F; k += BLOCK) a.w_out[k] = w_now(k);
    if (tid == 0) {
      if (a.theta_out) *a.theta_out = theta;
      if (pend && a.pend_reward_acc) *a.pend_reward_acc += a.pend_G[F + 1] * pinv;
    }
  }
  // wl is staged block-wide but read by every wave; the per-step barriers below may be wave-local, so order the
  // staging against all later reads once, here (one block barrier per launch)
  __syncthreads();
  MFG_STAMP0(9)
  // A wave only ever touches the tile rows / state slots of its OWN G trajectories, so when nothing is staged
  // block-wide (SAMPLE; the P copy-out is per wave too) the per-step barriers need not span the block: waves of a
  // block then run their serial chains without waiting for the slowest of the four.
#ifdef MFG_NO_WAVE_LOCAL
  const bool wave_local = false;
#else
  const bool wave_local = SAMPLE;
#endif
  auto tile_sync = [&]() {
    if (wave_local) {
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    } else {
      __syncthreads();
    }
  };
  // value term of this lane for the state in `vec` (CIRC: doubled vector) whose entry i is pi_e
  auto value_term = [&](const float* vec, float pi_e) -> double {
    double col = 0.0;
    if (CIRC) {
      double c0 = 0.0, c1 = 0.0;
#pragma unroll
      for (int m = 0; m + 1 < H; m += 2) {
        c0 = fma(wl[m * d + i], (double)vec[i + m], c0);
        c1 = fma(wl[(m + 1) * d + i], (double)vec[i + m + 1], c1);
      }
      if (H & 1) c0 = fma(wl[(H - 1) * d + i], (double)vec[i + H - 1], c0);
      col = c0 + c1;
    } else {
      for (int k = 0, idx = i; k <= i; idx += d - k - 1, ++k) col = fma(wl[idx], (double)vec[k], col);
    }
    return (double)pi_e * (col + wl[Q + i]);
  };
  const int64_t ntiles = (a.B + TB - 1) / TB;
  for (int64_t tileid = blockIdx.x; tileid < ntiles; tileid += nblk_f()) {
    const int64_t b0 = tileid * TB;
    const int nb = (int)((a.B - b0) < TB ? (a.B - b0) : TB);
    const int tl = wv * G + t;
    const bool valid = (t < G) && (tl < nb);
    const int tlc = valid ? tl : 0;
    const int64_t b = b0 + tlc;
    double* redq = red + (size_t)tlc * 4 * d;
    float* pnv = pin + tlc * pnw;
    // (pi_first: this tile's start state -- loaded ahead of the weight staging for the block's first tile, and, for every
    //  further tile, at the start of the tile before it: a T = 1 launch at large batches runs 3-4 tiles per block and paid
    //  the L2 / HBM latency of this load once per tile)
    float pi_i = pi_first;
    {
      const int64_t tn = tileid + nblk_f();
      if (tn < ntiles) {
        const int64_t b0n = tn * TB;
        const int64_t bn = b0n + ((t < G && b0n + tl < a.B) ? tl : 0);
        pi_first = a.pi0[core_src_row(a, bn) * d + i];
      }
    }
    if (valid && a.pi_traj) a.pi_traj[b * (int64_t)(T + 1) * d + i] = pi_i;
    if constexpr (STEP == 2) {
      if (valid && a.pi_start_out) a.pi_start_out[b * d + i] = pi_i;
    }
    double v_cur = 0.0, discount = 1.0;  // meaningful on lane i == 0 only
    // (V of the start state: evaluated inside step 0 next to V of the next state -- as a prologue it cost three barriers,
    //  two LDS round trips and a serial sum BEFORE any sampling could start: 3 500 of the 20 000 cycles of a T = 1 launch.
    //  GIVEN mode evaluates it inside its single step as before.)
    // (the Box-Muller partner of the row's trailing element, carried from an even step to the odd one behind it: sample_tail1)
    bool tail_have = false;
    float tail_xn = 0.0f, tail_kf = 0.0f;
    MFG_STAMP0(10)
    for (int s = 0; s < T; ++s) {
      MFG_STAMP(0)
      tile_sync();
      float Fi = 0.0f;
      if (valid) {
        pis[tlc * d + i] = pi_i;
        if (SAMPLE && want_v && s == 0) {
          pst[tlc * pnw + i] = pi_i;
          if (CIRC) pst[tlc * pnw + d + i] = pi_i;
        }
        if (SAMPLE) pis64[2 * (tlc * d + i)] = (double)pi_i;
        if (sep) {
          pex[tlc * d + i] = exp_f64arg(theta * ((double)pi_i - SEP_CENTRE));
          Fi = exp_f64arg(-theta * ((double)pi_i + (a.shift - SEP_CENTRE)));
        }
      }
      if (!SAMPLE) {
        // stage the given P tile (flat, coalesced) into the padded LDS tile
        const int n = nb * dd;
        const float* src = a.P_in + b0 * dd;
        for (int k = tid; k < n; k += BLOCK) {
          const int row = (int)(((float)k + 0.5f) * inv_d);
          const int colj = k - row * d;
          tile[row * dp + colj] = src[k];
        }
        if (a.pi_next_in)
          for (int k = tid; k < nb * d; k += BLOCK) pin[k] = a.pi_next_in[b0 * d + k];
        if (a.pi_alpha)
          for (int k = tid; k < nb * d; k += BLOCK) pal[k] = a.pi_alpha[b0 * d + k];
      }
      tile_sync();
      MFG_STAMP(1)
      float* trow = tile + (tlc * d + i) * dp;
      const float* pv = pis + tlc * d;
      const float* pav = (!SAMPLE && a.pi_alpha) ? pal + tlc * d : pv;
      const float pai = pav[i];
      const double pid = (double)pi_i;
      double A = 0.0, D_ = 0.0, Ssum = 0.0, gacc = 0.0, racc = 0.0;
      if (valid) {
        const uint32_t step = a.first_step + (uint32_t)s;
        const uint64_t traj = a.traj_offset + (uint64_t)b;
        if (SAMPLE) {
          // FOUR matrix elements per iteration from one Philox block, their chains interleaved (sample_elems); the row
          // sums of the quad are formed in the working precision and folded with one fp64 add each.
          using TT = typename PolicyTerms<FAST>::T;
          const float* ev = pex + tlc * d;
          const uint32_t erow = (uint32_t)(i * d);
          const float pas = sep ? pai + ts.sh : pai;  // row operand of the sampler (policy_setup_sep takes pi_i + shift)
          const int dq = d & ~3;
          if constexpr (D > 0 && sep) {
            // Running LDS pointers (address space 3, one VGPR each), made opaque to the loop optimiser once per iteration:
            // with the scalar loop counter it re-derived SIX vector addresses per quad from it (two shifts-and-adds and four
            // constant adds); now pi_j and E_j of the quad come from one base with immediate offsets (E_j sits TB*D floats
            // behind pi_j) and the tile row from a second one: two v_add_u32 per quad.
            typedef __attribute__((address_space(3))) const float lds_cf;
            typedef __attribute__((address_space(3))) float lds_f;
            constexpr int EOFF = WAVES * (WAVE / D) * D;  // floats from pis to pex
            static_assert(EOFF + 3 <= 255, "E_j must sit inside the ds_read2_b32 offset field of the pi_j base");
            lds_cf* rp = (lds_cf*)pav;
            lds_f* wp = (lds_f*)trow;
            // one full quad at columns j .. j+3 through the running pointers
            auto quad = [&](int j, float& ys, TT& as, TT& ds, TT& gs) __attribute__((always_inline)) {
              float y[4], pjv[4], ejv[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                pjv[e] = rp[e];
                ejv[e] = rp[EOFF + e];
              }
              sample_elems<4, TD, FAST, sep>(a, theta, ts, pjv, ejv, pas, Fi, erow + (uint32_t)j, step, traj, y, ys, as, ds, gs);
#pragma unroll
              for (int e = 0; e < 4; ++e) wp[e] = y[e];
              rp += 4;
              wp += 4;
              asm volatile("" : "+v"(rp), "+v"(wp));
            };
#ifndef MFG_QUAD_UNROLL
#define MFG_QUAD_UNROLL 1  // quads of a row whose chains the scheduler may interleave (developer switch; 2 measured below)
#endif
#pragma unroll MFG_QUAD_UNROLL
            for (int j = 0; j < dq; j += 4) {
              float ys;
              TT as, ds, gs;
              quad(j, ys, as, ds, gs);
              Ssum += (double)ys;
              if (TD) {
                A += (double)as;
                D_ += (double)ds;
                gacc += (double)gs;
              }
            }
            // (folding the quad sums in PAIRS of quads -- half the conversions and fp64 adds -- needs a two-quad loop body:
            //  twice the code, every cold continuation replicated; measured 1.08 -> 1.27 ms.  Not kept.)
          } else
#pragma unroll 1
          for (int j = 0; j < dq; j += 4) {
            float y[4], ys = 0.0f;
            TT as = 0, ds = 0, gs = 0;
            sample_elems<4, TD, FAST, sep>(a, theta, ts, pav + j, ev + j, pas, Fi, erow + (uint32_t)j, step, traj, y, ys,
                                           as, ds, gs);
#pragma unroll
            for (int e = 0; e < 4; ++e) trow[j + e] = y[e];
            Ssum += (double)ys;
            if (TD) {
              A += (double)as;
              D_ += (double)ds;
              gacc += (double)gs;
            }
          }
          if (dq < d) {  // 1..3 trailing elements of the row (their own Philox block, keyed by the first of them)
            float y[4], ys = 0.0f;
            TT as = 0, ds = 0, gs = 0;
            const int rem = d - dq;
            if (rem == 1)
              sample_tail1<TD, FAST, sep>(a, theta, ts, pav[dq], ev[dq], pas, Fi, erow + (uint32_t)dq, step, traj, tail_have, tail_xn,
                                          tail_kf, y[0], as, ds, gs), ys = y[0];
            else if (rem == 2)
              sample_elems<2, TD, FAST, sep>(a, theta, ts, pav + dq, ev + dq, pas, Fi, erow + (uint32_t)dq, step, traj, y,
                                             ys, as, ds, gs);
            else
              sample_elems<3, TD, FAST, sep>(a, theta, ts, pav + dq, ev + dq, pas, Fi, erow + (uint32_t)dq, step, traj, y,
                                             ys, as, ds, gs);
            for (int e = 0; e < rem; ++e) trow[dq + e] = y[e];
            Ssum += (double)ys;
            if (TD) {
              A += (double)as;
              D_ += (double)ds;
              gacc += (double)gs;
            }
          }
        } else {
          PolicyElem<FAST> pe;
#pragma unroll 2
          for (int j = 0; j < d; ++j) {
            policy_setup<SAMPLE, TD, FAST>(pe, a, theta, ts, pav[j], pai);
            const float p = trow[j];
            policy_accumulate<SAMPLE, TD, FAST>(pe, a.htab, ts.th, p, A, D_, gacc);
            racc += reward_term(a.reward_kind, pid, (double)pv[j], (double)p);
          }
        }
        MFG_STAMP(2)
        if (TD && FAST) gacc *= LN2;  // the element terms were summed in log2 units (policy_terms / policy_accumulate)
        if (SAMPLE) {
          // normalise the row.  strict: P_ij = fl32(y_ij / S_i); mixed: P_ij = y_ij * fl32(1 / S_i) (one fp32 multiply per
          // element, within 1.5 ulp of the strict value; rows still sum to 1 within a few 1e-7)
          //   Mixed mode leaves the variates in the tile and hands 1 / S_i to the column pass (one multiply per element THERE
          //   instead of a read-multiply-write pass over the row here: 2 LDS instructions per element less); only when P is
          //   written out is the tile itself normalised (the copy-out reads it), and the column pass multiplies by 1.
          float* inv_slot = reinterpret_cast<float*>(pis64 + 2 * (tlc * d + i) + 1);
          if (FAST) {
            const float inv32 = fast_rcp_f32_of_f64(Ssum);
            if (a.P_out) {
              for (int j = 0; j < d; ++j) trow[j] *= inv32;
              *inv_slot = 1.0f;
            } else {
              *inv_slot = inv32;
            }
#ifndef MFG_ABL_EPI
            if (TD) gacc -= fast_log_f64(Ssum) * D_;
#endif
          } else {
            const double invS = 1.0 / Ssum;
            for (int j = 0; j < d; ++j) trow[j] = (float)((double)trow[j] * invS);
            *inv_slot = 1.0f;
            if (TD) gacc -= log(Ssum) * D_;
          }
        }
#ifndef MFG_ABL_EPI
        if (TD) gacc = fma(FAST ? digamma_pos_mixed(A) : digamma_pos(A), D_, gacc);
#endif
      }
      MFG_STAMP(3)
      tile_sync();
      float pi_n;
      double rcol = 0.0;
      if (SAMPLE) {
        // column pass, lane = (trajectory, column i): pi'_i = sum_k pi_k P_ki and the reward sums of column i,
        //   u = pi_k P_ki (exact in fp64),  pi'_i += u,  s1 += u P_ki,  s2 += u^2   =>   R = sum_i (pi_i s1_i - s2_i)
        // -- the arithmetic of the given-P kernel k_step_small, so fused and unfused rewards agree bit for bit
        // (consecutive lanes -> consecutive banks; pi_k as fp64 from LDS, one broadcast read per row).
        double acc = 0.0, s1 = 0.0, s2 = 0.0;
        const float* tcol = tile + tlc * d * dp + i;
        const double2* q64 = reinterpret_cast<const double2*>(pis64) + tlc * d;
#ifndef MFG_COLWALK
#define MFG_COLWALK 2  // developer switch (A/B timing): 0 the plain row order of rounds 1-5 (other bits), 1 nested unrolled groups, 2 below
#endif
#if !defined(MFG_ABL_COLT) && !defined(MFG_ABL_COLREW) && MFG_COLWALK == 1
        if constexpr (D == 21) {
          constexpr int GR = col_group_rows(D);
#pragma unroll
          for (int g0 = 0; g0 < D; g0 += GR) {
            double pa, p1, p2;
#pragma unroll
            for (int r = 0; r < GR; ++r) {
              const int k = g0 + r;
              const double2 e = q64[k];
              const double p = (double)(tcol[k * dp] * __int_as_float(__double2loint(e.y)));
              col_walk_row(r == 0, p * e.x, p, pa, p1, p2);
            }
            if (g0 == 0) {
              acc = pa;
              s1 = p1;
              s2 = p2;
            } else {
              acc += pa;
              s1 += p1;
              s2 += p2;
            }
          }
        } else
#elif !defined(MFG_ABL_COLT) && !defined(MFG_ABL_COLREW) && MFG_COLWALK == 2
        if constexpr (D == 21 || D == 15) {
          // d = 21: three groups of seven rows; d = 15: four groups of 4, 4, 4, 3 -- each summed from zero, folded in group order
          // (col_group_rows, mfg_device.h): the summation tree the one-trajectory-per-wave kernel k_core_row3 shares.  Written as THE
          // loop of rounds 2-5 (seven / eight rows in flight) with the folds at group ends, static in every unrolled body: 0 + u,
          // fma(u, p, 0) and 0 + P0 are exact, so these are the bits of col_walk_row; the nested fully unrolled form of the first
          // version cost the T = 1 step kernels 3 us per launch.
          double pa = 0.0, p1 = 0.0, p2 = 0.0;
          constexpr int UNR = D == 15 ? 8 : col_group_rows(D);  // a whole number of groups per body
#pragma unroll UNR
          for (int k = 0; k < D; ++k) {
            const double2 e = q64[k];  // {pi_k, 1 / S_k (fp32 bits in the low word of .y)}
            const double p = (double)(tcol[k * dp] * __int_as_float(__double2loint(e.y)));
            const double u = p * e.x;
            pa += u;
            p1 = fma(u, p, p1);
            p2 = fma(u, u, p2);
            if (col_group_end(k, D)) {
              acc += pa;
              s1 += p1;
              s2 += p2;
              pa = p1 = p2 = 0.0;
            }
          }
        } else
#endif
#ifdef MFG_ABL_COLT
        for (int k = 0; k < 0; ++k) {
#else
#pragma unroll 7  // (3 / 5 / 7 / 11 / 21 rows in flight measured: 7 is the best at d = 21, -0.4 % against 3)
        for (int k = 0; k < d; ++k) {
#endif
          const double2 e = q64[k];  // {pi_k, 1 / S_k (fp32 bits in the low word of .y)}
          const double p = (double)(tcol[k * dp] * __int_as_float(__double2loint(e.y)));
          const double u = p * e.x;
          acc += u;
#ifndef MFG_ABL_COLREW
          s1 = fma(u, p, s1);  // both reward sums unconditionally; the kind selects what is used below
          s2 = fma(u, u, s2);
#endif
        }
        if (a.reward_kind == MFG_REWARD_MFG_AC2) rcol = fma(pid, s1, -s2);
        if (a.reward_kind == MFG_REWARD_SYNTHETIC) rcol = s1;
        pi_n = (float)acc;
        if (valid) {
          pnv[i] = pi_n;
          if (CIRC) pnv[d + i] = pi_n;
        }
        if (a.P_out && wv * G < nb) {
          // copy-out of THIS WAVE's trajectories into [B,T,d,d] (coalesced: a trajectory's matrix is contiguous on both
          // sides when the LDS tile is unpadded).  Wave local on purpose: the rows were written by this wave's own lanes, so
          // no block barrier is needed and the per-step barriers stay wave-local also when P is materialised (round 2
          // copied the block's tile with all four waves behind a __syncthreads: +4.4 us per T = 1 launch at B = 4 096).
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_wave_barrier();
          const int ntr = (nb - wv * G) < G ? (nb - wv * G) : G;  // trajectories of this wave in the tile
          const float* src = tile + (size_t)wv * G * d * dp;
          float* dst = a.P_out + ((b0 + wv * G) * (int64_t)T + s) * dd;
          if constexpr (D > 0 && (D & 1)) {
            // odd compile-time d: the tile is unpadded, a trajectory's matrix is [d*d] contiguous in LDS and in P_out;
            // all LDS reads of a matrix are issued before its stores (a rolled loop paid one LDS round trip per 64 floats)
            constexpr int NIT = (D * D + WAVE - 1) / WAVE;
#pragma unroll
            for (int tl2 = 0; tl2 < WAVE / D; ++tl2) {
              if (tl2 < ntr) {
                float v[NIT];
#pragma unroll
                for (int u = 0; u < NIT; ++u) {
                  const int k = lane + u * WAVE;
                  v[u] = src[tl2 * D * D + (k < D * D ? k : 0)];
                }
#pragma unroll
                for (int u = 0; u < NIT; ++u) {
                  const int k = lane + u * WAVE;
                  if (k < D * D) dst[(int64_t)tl2 * T * dd + k] = v[u];
                }
              }
            }
          } else if (dp == d) {
            for (int tl2 = 0; tl2 < ntr; ++tl2)
              for (int k = lane; k < dd; k += WAVE) dst[(int64_t)tl2 * T * dd + k] = src[tl2 * dd + k];
          } else {
            const int n = ntr * dd;
            for (int k = lane; k < n; k += WAVE) {
              const int row = (int)(((float)k + 0.5f) * inv_d);  // tl*d + i
              const int colj = k - row * d;
              const int tl2 = (int)(((float)row + 0.5f) * inv_d);
              const int ii = row - tl2 * d;
              dst[(int64_t)tl2 * T * dd + ii * d + colj] = src[row * dp + colj];
            }
          }
        }
      } else {
        pi_n = a.pi_next_in ? pnv[i] : 0.0f;
        rcol = pid * racc;
      }
      MFG_STAMP(4)
      // Per-trajectory sums of the lane terms (reward, score, value): every lane parks its terms in an LDS line, lane 0
      // of the trajectory adds up reward and value, lane 1 the score -- two LDS round trips per step instead of the six
      // dependent cross-lane exchanges (ds_bpermute) of a shuffle tree per quantity, in a fixed order.
      const bool ext = a.reward_kind == MFG_REWARD_EXTERNAL;
      if (valid) {
        if (!ext) redq[i] = rcol;
        if (TD) redq[d + i] = gacc;
      }
#ifdef MFG_ABL_V
      if (false) {
#else
      if (want_v) {
#endif
        tile_sync();  // pin complete
        if (!SAMPLE) {
          // GIVEN mode: single step, V(pi) of the current state first.  The barriers are block wide in this mode and sit
          // OUTSIDE the validity test: every wave of the block reaches them, also on a partial last tile.
          if (valid) redq[2 * d + i] = value_term(pv, pi_i);
          tile_sync();
          if (valid && i == 0) {
            double v0 = 0.0;
            for (int k = 0; k < d; ++k) v0 += redq[2 * d + k];
            v_cur = v0 + wl[Q + d];
          }
          tile_sync();
        }
        if (valid) redq[2 * d + i] = value_term(pnv, pi_n);
        if (SAMPLE && s == 0 && valid) redq[3 * d + i] = value_term(pst + tlc * pnw, pi_i);  // (pi_i: still the start state)
      }
      tile_sync();
      // d >= 8: EIGHT lanes of the trajectory share the work -- lane 2 q + p adds the terms of parity p of quantity q (0 reward,
      // 1 V(next), 2 score, 3 V(start), the last at step 0 only) and parks its partial in `tot`; lanes 0 / gl then combine
      // (even + odd), exactly the association of the serial form below: bit-identical results, but the wave issues 11
      // dependent fp64 adds per step instead of 63-84 (the serial sums were 6 of the kernel's 92 instructions per element).
      const bool par = d >= 8;
      double* totq = tot + tlc * 8;
      if (par) {
        if (valid && i < 8) {
          const int q = i >> 1, pp = i & 1;
          const bool need = q == 0 ? !ext : (q == 1 ? want_v : (q == 2 ? (TD && a.g != nullptr) : (SAMPLE && want_v && s == 0)));
          double x = 0.0;
          if (need) {
            const double* src = redq + (q == 0 ? 0 : (q == 1 ? 2 * d : (q == 2 ? d : 3 * d)));
#pragma unroll  // (fully unrolled: all loads in flight; unroll 4 / 2 / 1 measured +1 % / +5 % / +4.5 % on the whole rollout)
            for (int k = 0; k < (d + 1) / 2; ++k) {
              const int kk = 2 * k + pp;
              const double v = src[kk < d ? kk : 0];
              x += kk < d ? v : 0.0;
            }
          }
          totq[i] = x;
        }
        tile_sync();
      }
      const int gl = d > 1 ? 1 : 0;  // lane of the trajectory that sums the score
#ifdef MFG_ABL_TSUM
      if (false) {
#else
      if (valid && i == 0) {
#endif
        double r0 = 0.0, r1 = 0.0, v0 = 0.0, v1 = 0.0;
        if (ext) {
          r0 = a.reward_in ? (double)a.reward_in[b * T + s] : 0.0;
        } else if (par) {
          r0 = totq[0];
          r1 = totq[1];
        } else {
          int k = 0;
#pragma unroll
          for (; k + 1 < d; k += 2) {
            r0 += redq[k];
            r1 += redq[k + 1];
          }
          if (k < d) r0 += redq[k];
        }
        double r = r0 + r1;
        if (a.reward_kind == MFG_REWARD_SYNTHETIC) r *= -0.5;
        if (a.reward_out) a.reward_out[b * T + s] = (float)r;
        if (want_v) {
          if (par) {
            v0 = totq[2];
            v1 = totq[3];
          } else {
            int k = 0;
#pragma unroll
            for (; k + 1 < d; k += 2) {
              v0 += redq[2 * d + k];
              v1 += redq[2 * d + k + 1];
            }
            if (k < d) v0 += redq[2 * d + k];
          }
          const double v_next = (v0 + v1) + wl[Q + d];
          if (SAMPLE && s == 0) {  // V of the start state: the same even / odd sums as for every other state
            double u0 = 0.0, u1 = 0.0;
            if (par) {
              u0 = totq[6];
              u1 = totq[7];
            } else {
              int kk = 0;
#pragma unroll
              for (; kk + 1 < d; kk += 2) {
                u0 += redq[3 * d + kk];
                u1 += redq[3 * d + kk + 1];
              }
              if (kk < d) u0 += redq[3 * d + kk];
            }
            v_cur = (u0 + u1) + wl[Q + d];
          }
          const double gd = a.discount_pow ? discount : a.gamma;
          const double del = r + gd * v_next - v_cur;
          if (a.delta) a.delta[b * T + s] = del;
          if (SUMS) {
            scal[tlc * 3] = del;
            scal[tlc * 3 + 2] = (double)(float)r;  // the reward as the caller sees it (fp32 output)
          }
          v_cur = v_next;
          discount *= a.gamma;
        }
      }
#ifdef MFG_ABL_TSUM
      if (false) {
#else
      if (TD && valid && i == gl && a.g) {
#endif
        double g0 = 0.0, g1 = 0.0;
        if (par) {
          g0 = totq[4];
          g1 = totq[5];
        } else {
          int k = 0;
#pragma unroll
          for (; k + 1 < d; k += 2) {
            g0 += redq[d + k];
            g1 += redq[d + k + 1];
          }
          if (k < d) g0 += redq[d + k];
        }
        a.g[b * T + s] = g0 + g1;
        if (SUMS) scal[tlc * 3 + 1] = g0 + g1;
      }
      if (valid && a.pi_traj) a.pi_traj[(b * (int64_t)(T + 1) + s + 1) * d + i] = pi_n;
      pi_i = pi_n;
      MFG_STAMP(5)
    }
    if (valid && a.pi_next_out) a.pi_next_out[b * d + i] = pi_i;
    if constexpr (SUMS && SAMPLE && TD && D > 0) {
      if (a.part_rows) {
        // batch sums of this tile's transitions (T == 1): sample = trajectory, state = pis (the step's start state).
        // Operand layout and augmented vectors exactly as in k_grad_mfma_small (mfg_kernels.hip).
        constexpr int FO = D * (D + 1) / 2 + D + 1 + 3;
        static_assert(WAVES * FO * 8 <= WAVES * (WAVE / D) * D * (D | 1) * 4, "the partial rows reuse the tile region");
        tile_sync();
        const int li = lane & 15, lk = lane >> 4;
        const int trj = wv * G + lk;
        const bool ok = lk < G && trj < nb;
        const int trc = ok ? trj : 0;
        const bool plo_ok = li < D, phi_ok = 16 + li < D;
        const float plo = pis[trc * D + (plo_ok ? li : 0)], phi = pis[trc * D + (phi_ok ? 16 + li : 0)];
        const double de = ok ? scal[trc * 3] : 0.0, dg = ok ? scal[trc * 3 + 1] : 0.0, rr = ok ? scal[trc * 3 + 2] : 0.0;
        const double one = ok ? 1.0 : 0.0;
        auto aug = [&](int idx, float pv, bool is_pi, double& A, double& B) {
          const float ad = idx == D ? 1.0f : 0.0f, b1 = (idx == D || idx == D + 3) ? 1.0f : 0.0f;
          const double a1 = idx == D + 1 ? 1.0 : 0.0, bg = idx == D + 1 ? 1.0 : 0.0, br = idx == D + 2 ? 1.0 : 0.0;
          A = fma(de, (double)(is_pi ? pv : ad), a1 * one);
          B = fma(bg, dg, fma(br, rr, (double)(is_pi ? pv : b1)));
        };
        double A_lo, B_lo, A_hi, B_hi;
        aug(li, plo, plo_ok, A_lo, B_lo);
        aug(16 + li, phi, phi_ok, A_hi, B_hi);
        const v4d_t z = (v4d_t)(0.0);
        const v4d_t c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A_lo, B_lo, z, 0, 0, 0);
        const v4d_t c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A_lo, B_hi, z, 0, 0, 0);
        const v4d_t c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(A_hi, B_hi, z, 0, 0, 0);
        __syncthreads();  // every wave is done with the tile region: it now holds the waves' rows
        double* rows = reinterpret_cast<double*>(tile);
        constexpr int Qc = D * (D + 1) / 2, Fc = Qc + D + 1;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            // D[i][j] of a tile: this lane holds row lk + 4 v, column li (f64 MFMA layout)
            const int gi = (tt == 2 ? 16 : 0) + 4 * v + lk, gj = (tt == 0 ? 0 : 16) + li;
            int k = -1;
            if (gj < D) {
              if (gi <= gj) k = feat_idx(gi, gj, D);
            } else if (gj == D) {
              if (gi <= D) k = Qc + gi;
            } else if (gj == D + 1) {
              if (gi == D) k = Fc;
            } else if (gj == D + 2) {
              if (gi == D + 1) k = Fc + 1;
            } else if (gj == D + 3) {
              if (gi == D + 1) k = Fc + 2;
            }
            const double val = tt == 0 ? c0[v] : (tt == 1 ? c1[v] : c2[v]);
            if (k >= 0) rows[wv * FO + k] = val;
          }
        }
        __syncthreads();
        for (int k = tid; k < FO; k += BLOCK) {
          double tsum = rows[k];
#pragma unroll
          for (int q = 1; q < WAVES; ++q) tsum += rows[q * FO + k];
          a.part_rows[tileid * FO + k] = tsum;
        }
        __syncthreads();  // (a further tile of this block would reuse the region)
      }
    }
    MFG_STAMP0(11)
    MFG_STAMPB(1)
  }
}

inline size_t core_small_lds(int d, bool want_v, bool sample) {
  const int G = WAVE / d, TB = WAVES * G, dp = d | 1;
  const size_t F = (size_t)d * (d + 1) / 2 + d + 1;
  const size_t fl = (size_t)TB * d * dp + 5 * (size_t)TB * d;  // floats: tile, pis, pin (doubled), pal, pex
  return (want_v ? ((F + 1) & ~(size_t)1) * 8 : 0) + (sample ? (size_t)2 * TB * d * 8 : 0) + (size_t)4 * TB * d * 8 + (fl + (fl & 1)) * 4 +
         (size_t)TB * 3 * 8 + (size_t)TB * 8 * 8 + (size_t)TB * 2 * d * 4;  // + SUMS scalars + partial sums + start state
}

// ---------------------------------------------------------------------------------------------
// large d (d > 64): one wavefront per trajectory, lane owns columns c = lane + 64 m (m < R).
// ---------------------------------------------------------------------------------------------
// SAMPLE mode (round 2): rows are processed TWO at a time so that one Philox block feeds a quad
// {(i, c), (i, c+64), (i+1, c), (i+1, c+64)} of the lane's columns (sample_elems_g); e^z is separable (E_c lives in the
// registers of the lane that owns column c, F_i of the row comes from LDS); in mixed mode the per-row sums S, A, D are fp32
// and reduced TRANSPOSED over batches of rows (round 3, see the sampling loop); the transition and reward sums use the
// u = pi_i P form (4 fp64 operations per element).
// registers: R <= 2 fits 168 VGPRs (3 waves / SIMD) without spilling; R >= 3 needs ~220 (2 waves / SIMD; capping it at
// 168 spills 45 registers and measured slower).  Round 3: the mixed-mode sampling kernels at R <= 2 take the 128-register
// budget of FOUR waves per SIMD -- since the row batches their phases are short enough that the 25 spilled registers (all
// outside the sampling loop) cost less than the fourth wave brings: C3 20.47 -> 20.18 ms.
#ifndef MFG_CORE_LARGE_WAVES
#define MFG_CORE_LARGE_WAVES(R) (((R) <= 2 || (R) == 4) ? 3 : 2)
#endif
#ifndef MFG_CORE_LARGE_WAVES_MIXED_SAMPLING
// R = 3: 176 registers wanted, 8 spilled at 168: d = 192 35.4 -> 31.2 ms with the third wave; R = 5, 6 (190-200 wanted): d = 320
// 24.2 -> 22.6 ms, d = 384 32.7 -> 30.6 ms; R = 7 (208-216 wanted): d = 448 35.8 -> 32.9 ms at 6 144 trajectories (25.2 against 24.9 ms
// at 2 048, where a SIMD holds two waves anyway); R = 8 (228 wanted) is better off with two waves (33.7 against 35.0 ms at d = 512)
#define MFG_CORE_LARGE_WAVES_MIXED_SAMPLING(R) ((R) <= 2 ? 4 : ((R) <= 7 ? 3 : 2))
#endif
// Rows per batch of the mixed-mode sampling loop (transposed row sums): sized so that the stash (KB x 64 R floats per
// wave) keeps three blocks per CU at R <= 4 and two above.
// (R = 3, 4 with TD: 8 rows -- 52 KB per block at d = 256, three blocks per CU; measured 78.85 -> 77.2 ms against 4 rows.  Without
//  TD the kernel fits four blocks per CU with a 4-row stash and keeps them.)
__host__ __device__ constexpr int large_row_batch(int R, bool td) { return R <= 2 ? 8 : (R <= 4 ? (td ? 8 : 4) : 2); }
// dynamic LDS of k_core_large: 4 state vectors per wave, the per-row (A, D, S), the stash
inline size_t core_large_lds(int d, bool sample, bool fast, bool td) {
  const int R = (d + WAVE - 1) / WAVE;
  // mixed-mode sampling: state + F per wave, (A, D, S) as fp32 (TD only), the stash (the next-state vector and the unused
  // alpha-state vector live inside it: they are written after a step's row loop, when the stash is dead)
  if (sample && fast) return (size_t)WAVES * ((td ? 5 : 2) * d + large_row_batch(R, td) * WAVE * R) * 4;
  return (size_t)WAVES * 4 * d * 4 + (size_t)WAVES * 3 * d * 8;
}
// a lane's R variates of one row to / from its stash slot (R consecutive floats, 4 R bytes aligned)
template <int R>
__device__ __forceinline__ void stash_store(float* p, const float* y) {
  if constexpr (R % 4 == 0) {
#pragma unroll
    for (int m = 0; m < R; m += 4) *reinterpret_cast<float4*>(p + m) = make_float4(y[m], y[m + 1], y[m + 2], y[m + 3]);
  } else if constexpr (R % 2 == 0) {
#pragma unroll
    for (int m = 0; m < R; m += 2) *reinterpret_cast<float2*>(p + m) = make_float2(y[m], y[m + 1]);
  } else {
#pragma unroll
    for (int m = 0; m < R; ++m) p[m] = y[m];
  }
}
template <int R>
__device__ __forceinline__ void stash_load(float* y, const float* p) {
  if constexpr (R % 4 == 0) {
#pragma unroll
    for (int m = 0; m < R; m += 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + m);
      y[m] = v.x, y[m + 1] = v.y, y[m + 2] = v.z, y[m + 3] = v.w;
    }
  } else if constexpr (R % 2 == 0) {
#pragma unroll
    for (int m = 0; m < R; m += 2) {
      const float2 v = *reinterpret_cast<const float2*>(p + m);
      y[m] = v.x, y[m + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int m = 0; m < R; ++m) y[m] = p[m];
  }
}
// FULL: d == 64 R (every lane owns R live columns, d even): the validity masks of the quads fold away (d = 128, 256, ...).
template <int R, bool SAMPLE, bool TD, bool FAST, bool FULL>
__global__ __launch_bounds__(BLOCK, (SAMPLE && FAST) ? MFG_CORE_LARGE_WAVES_MIXED_SAMPLING(R) : MFG_CORE_LARGE_WAVES(R))
void k_core_large(CoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = a.d, T = a.T;
  const int64_t dd = (int64_t)d * d;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  constexpr bool batched = SAMPLE && FAST;
  constexpr int NSV = batched ? 2 : 4;  // state vectors per wave (batched: next state / alpha state alias the stash)
  float* pis = smem + wv * NSV * d;  // current state
  float* pfs = pis + d;              // SAMPLE, mixed: F_i = e^{-theta (pi_i + shift)}   (others: next state)
  float* ystash = smem + WAVES * (NSV + (batched && TD ? 3 : 0)) * d + wv * (large_row_batch(R, TD) * WAVE * R);
  float* pin = batched ? ystash : pis + d;      // next state
  float* pal = batched ? ystash : pis + 2 * d;  // state for alpha (GIVEN with pi_alpha)
  if (!batched) pfs = pis + 3 * d;
  // per-row (A_i, D_i, S_i) of this wave's trajectory: psi(A_i) D_i and ln(S_i) D_i are evaluated AFTER the row
  // loop, one row per lane, instead of once per row by the whole wave (a fp64 digamma + log per row amortised
  // over only d/64 elements per lane dominated the TD kernels at d = 128)
  // (mixed-mode sampling: fp32 -- the sums are fp32 sums there -- followed by the lane-private stash of a row batch's variates)
  using RQ = typename std::conditional<batched, float, double>::type;
  RQ* rowq = reinterpret_cast<RQ*>(smem + WAVES * NSV * d) + wv * 3 * d;
  const bool want_v = TD && a.w != nullptr;
  const double theta = *a.theta;
  const ThetaSplit ts = theta_split(theta, a.shift);
  constexpr bool sep = SAMPLE && FAST;
  if (sep) report_sep_range(a.status, theta, a.shift);
  using TT = typename PolicyTerms<FAST>::T;
  const int64_t nw = (int64_t)gridDim.x * WAVES;
  for (int64_t b = (int64_t)blockIdx.x * WAVES + wv; b < a.B; b += nw) {
    float pc[R];
    const int64_t row0 = core_src_row(a, b);
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int c = lane + m * WAVE;
      pc[m] = c < d ? a.pi0[row0 * d + c] : 0.0f;
      if (c < d && a.pi_traj) a.pi_traj[b * (int64_t)(T + 1) * d + c] = pc[m];
    }
    double v_cur = 0.0, discount = 1.0;
    bool have_v = false;
    const uint64_t traj = a.traj_offset + (uint64_t)b;
    for (int s = 0; s < T; ++s) {
      __builtin_amdgcn_wave_barrier();
      float Ec[R];
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const int c = lane + m * WAVE;
        Ec[m] = 0.0f;
        if (c < d) {
          pis[c] = pc[m];
          if (!SAMPLE && a.pi_next_in) pin[c] = a.pi_next_in[b * d + c];
          if (!SAMPLE && a.pi_alpha) pal[c] = a.pi_alpha[b * d + c];
          if (sep) {
            Ec[m] = exp_f64arg(theta * ((double)pc[m] - SEP_CENTRE));
            pfs[c] = exp_f64arg(-theta * ((double)pc[m] + (a.shift - SEP_CENTRE)));
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      const float* pav = (!SAMPLE && a.pi_alpha) ? pal : pis;
      double acc[R], s1[R];
      float pad[R];  // state the concentrations are computed from (== pc when sampling)
      bool okc[R];
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const int c = lane + m * WAVE;
        okc[m] = FULL || c < d;
        pad[m] = SAMPLE ? pc[m] : (c < d ? pav[c] : 0.0f);
        acc[m] = 0.0;
        s1[m] = 0.0;
      }
      double racc = 0.0, s2 = 0.0, gacc = 0.0, guni = 0.0;
      const float* Pb = SAMPLE ? nullptr : a.P_in + b * dd;
      float* Po = (SAMPLE && a.P_out) ? a.P_out + (b * (int64_t)T + s) * dd : nullptr;
      const uint32_t step = a.first_step + (uint32_t)s;
      if constexpr (SAMPLE) {
        // rows per iteration: R a multiple of 4 -> a row's elements of this lane fill whole quads; otherwise two rows
        // per iteration and quads {(i, m), (i, m+1), (i+1, m), (i+1, m+1)} (odd R: last column {(i, m), (i+1, m)})
        constexpr int NR = (R % 4 == 0) ? 1 : 2;
        // sample rows i (and i + 1 when NR == 2): gamma variates y, per-lane partial row sums, score terms
        auto sample_rows = [&](int i, bool row1, float (&y)[NR][R], TT (&ysum)[2], TT (&asum)[2], TT (&dsum)[2], TT& gsum)
                               __attribute__((always_inline)) {
          const int i1 = row1 ? i + 1 : i;
          const float pr[2] = {pis[i], pis[i1]};
          const float prs[2] = {sep ? pr[0] + ts.sh : pr[0], sep ? pr[1] + ts.sh : pr[1]};  // the sampler's row operands
          float fr[2] = {0.0f, 0.0f};
          if (sep) {
            fr[0] = pfs[i];
            fr[1] = pfs[i1];
          }
          if constexpr (NR == 1) {
#pragma unroll
            for (int m = 0; m < R; m += 4) {
              const uint32_t e0 = (uint32_t)(i * d + lane + m * WAVE);
              const float pj[4] = {pad[m], pad[m + 1], pad[m + 2], pad[m + 3]};
              const float ej[4] = {Ec[m], Ec[m + 1], Ec[m + 2], Ec[m + 3]};
              const float pa[4] = {prs[0], prs[0], prs[0], prs[0]};
              const float fi[4] = {fr[0], fr[0], fr[0], fr[0]};
              const uint32_t el[4] = {e0, e0 + WAVE, e0 + 2 * WAVE, e0 + 3 * WAVE};
              const bool ok[4] = {okc[m], okc[m + 1], okc[m + 2], okc[m + 3]};
              float yy[4];
              TT al[4], ad[4], gt[4];
              sample_elems_g<4, TD, FAST, sep>(a, theta, ts, pj, ej, pa, fi, el, ok, step, traj, yy, al, ad, gt);
#pragma unroll
              for (int e = 0; e < 4; ++e) y[0][m + e] = yy[e];
              // (m == 0: assign -- "0 + x" is not foldable without fast-math and costs an instruction per sum)
              const TT y4 = ((TT)yy[0] + (TT)yy[1]) + ((TT)yy[2] + (TT)yy[3]);
              ysum[0] = m == 0 ? y4 : ysum[0] + y4;
              if (TD) {
                const TT a4 = (al[0] + al[1]) + (al[2] + al[3]), d4 = (ad[0] + ad[1]) + (ad[2] + ad[3]);
                const TT g4 = (gt[0] + gt[1]) + (gt[2] + gt[3]);
                asum[0] = m == 0 ? a4 : asum[0] + a4;
                dsum[0] = m == 0 ? d4 : dsum[0] + d4;
                gsum = m == 0 ? g4 : gsum + g4;
              }
            }
          } else {
#pragma unroll
            for (int m = 0; m < R; m += 2) {
              const bool two = m + 1 < R;  // compile time
              const int m1 = two ? m + 1 : m;
              const uint32_t e00 = (uint32_t)(i * d + lane + m * WAVE), e10 = (uint32_t)(i1 * d + lane + m * WAVE);
              if (two) {
                const float pj[4] = {pad[m], pad[m1], pad[m], pad[m1]};
                const float ej[4] = {Ec[m], Ec[m1], Ec[m], Ec[m1]};
                const float pa[4] = {prs[0], prs[0], prs[1], prs[1]};
                const float fi[4] = {fr[0], fr[0], fr[1], fr[1]};
                const uint32_t el[4] = {e00, e00 + WAVE, e10, e10 + WAVE};
                const bool ok[4] = {okc[m], okc[m1], okc[m] && row1, okc[m1] && row1};
                float yy[4];
                TT al[4], ad[4], gt[4];
                sample_elems_g<4, TD, FAST, sep>(a, theta, ts, pj, ej, pa, fi, el, ok, step, traj, yy, al, ad, gt);
                y[0][m] = yy[0];
                y[0][m1] = yy[1];
                y[NR - 1][m] = yy[2];
                y[NR - 1][m1] = yy[3];
                const TT ya = (TT)yy[0] + (TT)yy[1], yb = (TT)yy[2] + (TT)yy[3];
                ysum[0] = m == 0 ? ya : ysum[0] + ya;
                ysum[1] = m == 0 ? yb : ysum[1] + yb;
                if (TD) {
                  const TT g4 = (gt[0] + gt[1]) + (gt[2] + gt[3]);
                  asum[0] = m == 0 ? al[0] + al[1] : asum[0] + (al[0] + al[1]);
                  asum[1] = m == 0 ? al[2] + al[3] : asum[1] + (al[2] + al[3]);
                  dsum[0] = m == 0 ? ad[0] + ad[1] : dsum[0] + (ad[0] + ad[1]);
                  dsum[1] = m == 0 ? ad[2] + ad[3] : dsum[1] + (ad[2] + ad[3]);
                  gsum = m == 0 ? g4 : gsum + g4;
                }
              } else {
                // odd R: the lane's last column; its two rows share the Box-Muller pair
                const float pj[2] = {pad[m], pad[m]};
                const float ej[2] = {Ec[m], Ec[m]};
                const float pa[2] = {prs[0], prs[1]};
                const float fi[2] = {fr[0], fr[1]};
                const uint32_t el[2] = {e00, e10};
                const bool ok[2] = {okc[m], okc[m] && row1};
                float yy[2];
                TT al[2], ad[2], gt[2];
                sample_elems_g<2, TD, FAST, sep>(a, theta, ts, pj, ej, pa, fi, el, ok, step, traj, yy, al, ad, gt);
                y[0][m] = yy[0];
                y[NR - 1][m] = yy[1];
                ysum[0] += (TT)yy[0];
                ysum[1] += (TT)yy[1];
                if (TD) {
                  asum[0] += al[0];
                  asum[1] += al[1];
                  dsum[0] += ad[0];
                  dsum[1] += ad[1];
                  gsum += gt[0] + gt[1];
                }
              }
            }
          }
        };
        // one normalised row into the column sums (transition, both reward sums)
        //   (wp = whether P is written out: decided once per batch, not per element -- a wave-uniform branch around every
        //   store is a TAKEN branch per element in the common no-output case)
        auto fold_row = [&](auto wp, int ir, float pi_row, const float* yr, float inv32, double invS) __attribute__((always_inline)) {
          const double pii = (double)pi_row;
#pragma unroll
          for (int m = 0; m < R; ++m) {
            if (okc[m]) {
              const float p32 = FAST ? yr[m] * inv32 : (float)((double)yr[m] * invS);
              const double p = (double)p32;
              if constexpr (decltype(wp)::value) Po[(int64_t)ir * d + lane + m * WAVE] = p32;
              const double u = p * pii;
              acc[m] += u;
              s1[m] = fma(u, p, s1[m]);  // both reward sums unconditionally (2 FMAs): a run-time kind test per element
              s2 = fma(u, u, s2);        // compiles to selects around them (4 v_cndmask per element)
            }
          }
        };
        if constexpr (FAST) {
          // Mixed mode: rows go in batches of KB.  Phase 1 samples the batch pair by pair -- the variates wait in a lane-private
          // LDS stash, the per-lane partials of the row sums (S and, TD, A, D) are packed TRANSPOSED (row_pair_merge ...,
          // mfg_device.h) so that the batch shares its butterfly steps; phase 2 normalises and folds the rows into the column
          // sums, the reciprocal of a row's sum coming from the lane that owns it through one read-lane.
          constexpr int KB = large_row_batch(R, TD), NQ = TD ? 3 : 1;
          float* yst = ystash + (int64_t)lane * R;
          for (int i0 = 0; i0 < d; i0 += KB) {
            float x[NQ];
            float park[NQ];  // NR == 1: the even row of a pair waits here for the odd one
            TT gb = 0;       // score terms of the batch (fp32), folded into the fp64 sum once per batch: the score stays within
                             // 1.9e-7 of the strict one (tools/score_error.py), a cvt + fp64 add per row pair less
#pragma unroll 1
            for (int k = 0; k < KB; k += NR) {  // rolled: the sampling body is large
              const int i = i0 + k;
              if (!FULL && i >= d && (NR == 2 || (k & 1) == 0)) break;  // (an odd last row still merges with zeros)
              float xr[2][NQ] = {};
              if (FULL || i < d) {
                const bool row1 = NR == 2 && (FULL || i + 1 < d);
                float y[NR][R];
                TT ysum[2] = {0, 0}, asum[2] = {0, 0}, dsum[2] = {0, 0}, gsum = 0;
                sample_rows(i, row1, y, ysum, asum, dsum, gsum);
                if (TD) gb += gsum;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                  stash_store<R>(yst + (k + rr) * (WAVE * R), y[rr]);
                  xr[rr][0] = ysum[rr];
                  if (TD) {
                    xr[rr][NQ > 1 ? 1 : 0] = asum[rr];
                    xr[rr][NQ > 2 ? 2 : 0] = dsum[rr];
                  }
                }
              }
              if (NR == 1 && (k & 1) == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) park[q] = xr[0][q];
                continue;
              }
              float m[NQ];
              row_pair_merge<NQ>(NR == 2 ? xr[0] : park, NR == 2 ? xr[1] : xr[0], m);
              if constexpr (KB == 2) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) x[q] = m[q];
              } else {
                row_pair_deposit<NQ>(x, m, row_pair_mask<KB>(k >> 1));
              }
            }
            if (TD) gacc += (double)gb;
            float tot;  // lane l: the batch totals of row i0 + row_batch_row<KB>(l) -- S where the reciprocal is read from
            if constexpr (TD) {
              tot = row_batch_finish3(x);  // S, A and D in one register (mfg_device.h), published by three lanes per row
              const int ir = i0 + row_batch_row<KB>(lane);
              if (row_batch_owner3<KB>(lane) && (FULL || ir < d)) rowq[3 * ir + row_batch_slot3(lane)] = tot;
            } else {
              row_batch_finish<NQ>(x);
              tot = x[0];
            }
            float inv = __builtin_amdgcn_rcpf(tot);
            inv = __builtin_fmaf(__builtin_fmaf(-tot, inv, 1.0f), inv, inv);
            auto fold_batch = [&](auto wp) __attribute__((always_inline)) {
              float yr[KB][R], pb[KB];  // the whole batch is requested up front: one LDS round trip per batch, not per row
#pragma unroll
              for (int k = 0; k < KB; ++k) stash_load<R>(yr[k], yst + k * (WAVE * R));
              if constexpr (FULL && KB % 4 == 0) {  // the batch's state entries: 16-byte broadcast reads (i0 and d are multiples of KB)
#pragma unroll
                for (int k = 0; k < KB; k += 4) {
                  const float4 pq = *reinterpret_cast<const float4*>(pis + i0 + k);
                  pb[k] = pq.x, pb[k + 1] = pq.y, pb[k + 2] = pq.z, pb[k + 3] = pq.w;
                }
              } else {
#pragma unroll
                for (int k = 0; k < KB; ++k) pb[k] = pis[(FULL || i0 + k < d) ? i0 + k : i0];
              }
#pragma unroll
              for (int k = 0; k < KB; ++k) {
                const int ir = i0 + k;
                if (!FULL && ir >= d) break;
                const float inv_k = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv), row_batch_lane(k)));
                fold_row(wp, ir, pb[k], yr[k], inv_k, 0.0);
              }
            };
            if (Po) fold_batch(std::true_type{});
            else fold_batch(std::false_type{});
          }
        } else {
          for (int i = 0; i < d; i += NR) {
            const bool row1 = NR == 2 && (FULL || i + 1 < d);
            const int i1 = row1 ? i + 1 : i;
            float y[NR][R];
            TT ysum[2] = {0, 0}, asum[2] = {0, 0}, dsum[2] = {0, 0}, gsum = 0;
            sample_rows(i, row1, y, ysum, asum, dsum, gsum);
            if (TD) gacc += (double)gsum;
            double Sr[2] = {1.0, 1.0}, Ar[2] = {0.0, 0.0}, Dr[2] = {0.0, 0.0};
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
              Sr[rr] = ysum[rr];
              if (TD) {
                Ar[rr] = asum[rr];
                Dr[rr] = dsum[rr];
                wave_sum3_dpp(Sr[rr], Ar[rr], Dr[rr]);
              } else {
                Sr[rr] = wave_sum_dpp(Sr[rr]);
              }
            }
            if (NR == 2 && !row1) Sr[1] = 1.0;
            if (TD && lane == 0) {
              rowq[3 * i] = Ar[0];
              rowq[3 * i + 1] = Dr[0];
              rowq[3 * i + 2] = Sr[0];
              if (row1) {
                rowq[3 * i1] = Ar[1];
                rowq[3 * i1 + 1] = Dr[1];
                rowq[3 * i1 + 2] = Sr[1];
              }
            }
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
              if (rr == 1 && !row1) break;
              const int ir = rr ? i1 : i;
              if (Po) fold_row(std::true_type{}, ir, pis[ir], y[rr], 0.0f, 1.0 / Sr[rr]);
              else fold_row(std::false_type{}, ir, pis[ir], y[rr], 0.0f, 1.0 / Sr[rr]);
            }
          }
        }
        // R = sum_j (pi_j s1_j - s2_j)  (kind 0)  /  -1/2 sum_j s1_j  (kind 1)
#pragma unroll
        for (int m = 0; m < R; ++m) racc += (a.reward_kind == MFG_REWARD_MFG_AC2) ? (double)pc[m] * s1[m] : s1[m];
        if (a.reward_kind == MFG_REWARD_MFG_AC2) racc -= s2;
      } else {
        for (int i = 0; i < d; ++i) {
          const double pii = (double)pis[i];
          const float pai = pav[i];
          float y[R];
          double A = 0.0, D = 0.0;
          PolicyElem<FAST> pe;
#pragma unroll
          for (int m = 0; m < R; ++m) {
            const int c = lane + m * WAVE;
            y[m] = 0.0f;
            if (c < d) {
              policy_setup<SAMPLE, TD, FAST>(pe, a, theta, ts, pad[m], pai);
              y[m] = Pb[(int64_t)i * d + c];
              policy_accumulate<SAMPLE, TD, FAST>(pe, a.htab, ts.th, y[m], A, D, gacc);
            }
          }
          if (TD) {
            A = wave_sum_dpp(A);
            D = wave_sum_dpp(D);
            if (lane == 0) {
              rowq[3 * i] = A;
              rowq[3 * i + 1] = D;
              rowq[3 * i + 2] = 1.0;
            }
          }
#pragma unroll
          for (int m = 0; m < R; ++m) {
            const int c = lane + m * WAVE;
            if (c < d) {
              const double p = (double)y[m];
              acc[m] = fma(p, pii, acc[m]);
              racc += pii * reward_term(a.reward_kind, pii, (double)pc[m], p);
            }
          }
        }
      }
      float pn[R];
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const int c = lane + m * WAVE;
        if (SAMPLE) {
          pn[m] = (float)acc[m];
          if (c < d) pin[c] = pn[m];
        } else {
          pn[m] = (c < d && a.pi_next_in) ? pin[c] : 0.0f;
        }
      }
      double r;
      if (a.reward_kind == MFG_REWARD_EXTERNAL) {
        r = a.reward_in ? (double)a.reward_in[b * T + s] : 0.0;
      } else {
        r = wave_sum_dpp(racc);
        if (a.reward_kind == MFG_REWARD_SYNTHETIC) r *= -0.5;
      }
      if (lane == 0 && a.reward_out) a.reward_out[b * T + s] = (float)r;
      if (TD) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#ifdef MFG_ABL_EPI
        for (int rw = d; rw < d; rw += WAVE) {
#else
        for (int rw = lane; rw < d; rw += WAVE) {
#endif
          const double Dr = (double)rowq[3 * rw + 1];
          if (SAMPLE && FAST) {
            guni = fma(digamma_pos_mixed((double)rowq[3 * rw]), Dr, guni);
            guni -= fast_log_f64((double)rowq[3 * rw + 2]) * Dr;
          } else {
            guni = fma(digamma_pos(rowq[3 * rw]), Dr, guni);
            if (SAMPLE) guni -= log(rowq[3 * rw + 2]) * Dr;
          }
        }
        const double gsum = wave_sum_dpp((FAST ? gacc * LN2 : gacc) + guni);  // mixed: element terms in log2 units
        if (lane == 0 && a.g) a.g[b * T + s] = gsum;
#ifdef MFG_ABL_V
        if (false) {
#else
        if (want_v) {
#endif
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_wave_barrier();
          if (!have_v) {
            v_cur = value_wave(pis, a.w, d, lane);
            have_v = true;
          }
          const double v_next = value_wave(pin, a.w, d, lane);
          const double gd = a.discount_pow ? discount : a.gamma;
          const double del = r + gd * v_next - v_cur;
          if (lane == 0 && a.delta) a.delta[b * T + s] = del;
          v_cur = v_next;
          discount *= a.gamma;
        }
      }
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const int c = lane + m * WAVE;
        if (c < d && a.pi_traj) a.pi_traj[(b * (int64_t)(T + 1) + s + 1) * d + c] = pn[m];
        pc[m] = pn[m];
      }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int c = lane + m * WAVE;
      if (c < d && a.pi_next_out) a.pi_next_out[b * d + c] = pc[m];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

constexpr int MFG_CORE_OVERSUBSCRIBE = 8;
inline int core_grid(int64_t work_items, int per_block, int blocks_per_cu, int num_cus) {
  int64_t g = (work_items + per_block - 1) / per_block;
  const int64_t cap = (int64_t)num_cus * blocks_per_cu;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// RSEL selects the instantiations a translation unit carries: 0 all R, 1 only R = 2 and 4 (d = 128 / 256: the C3 and C5
// shapes -- built with LLVM's iterative ILP scheduler, which gains 1.8 % at d = 256 and crashes the register allocator on
// another R), 2 all the others.
template <bool FAST, int RSEL = 0>
inline int launch_core_large_impl(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st) {
  const int d = a.d;
  const int R = (d + WAVE - 1) / WAVE;
  const size_t lds = core_large_lds(d, sample, FAST, td);
  const int grid = core_grid(a.B, WAVES, 8 * MFG_CORE_OVERSUBSCRIBE, num_cus);
#define MFG_CORE_LARGE_GO(RR, SS, TT, FF) \
  hipLaunchKernelGGL((k_core_large<RR, SS, TT, FAST, FF>), dim3(grid), dim3(BLOCK), lds, st, a)
#define MFG_CORE_LARGE_MODE(RR)                                          \
  if (sample && td) { if (full) MFG_CORE_LARGE_GO(RR, true, true, true); else MFG_CORE_LARGE_GO(RR, true, true, false); }   \
  else if (sample) { if (full) MFG_CORE_LARGE_GO(RR, true, false, true); else MFG_CORE_LARGE_GO(RR, true, false, false); }  \
  else MFG_CORE_LARGE_GO(RR, false, true, false);
#define MFG_CORE_LARGE_CASE(RR)                                                              \
  if constexpr (RSEL == 0 || (RSEL == 1) == ((RR) == 2 || (RR) == 4)) {                      \
    if (R == (RR)) {                                                                         \
      MFG_CORE_LARGE_MODE(RR)                                                                \
      return MFG_OK;                                                                         \
    }                                                                                        \
  }
  const bool full = (d == R * WAVE);
  MFG_CORE_LARGE_CASE(2)
  MFG_CORE_LARGE_CASE(3)
  MFG_CORE_LARGE_CASE(4)
  MFG_CORE_LARGE_CASE(5)
  MFG_CORE_LARGE_CASE(6)
  MFG_CORE_LARGE_CASE(7)
  MFG_CORE_LARGE_CASE(8)
#undef MFG_CORE_LARGE_CASE
#undef MFG_CORE_LARGE_GO
#undef MFG_CORE_LARGE_MODE
  return MFG_EUNSUPPORTED;
}

}  // namespace mfg
