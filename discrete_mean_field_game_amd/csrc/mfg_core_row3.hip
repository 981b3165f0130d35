// d = 21 / 15, mixed-precision SAMPLING launches of batches that under-fill the machine: ONE trajectory per wavefront, THREE (d = 21) or
// FOUR (d = 15) lanes per matrix row (round 6).  The description below is written for d = 21; d = 15 (the reference's AC_IRL default,
// ac_irl.py:33) is the same with four lanes per row and ONE unit per lane: the three quads and the 3-element remainder of a row.
//
// The packed kernel k_core_small (mfg_core.h) puts G = 3 trajectories into a wave, lane = (trajectory, row): a wave walks the 21
// elements of its rows one quad after the other -- a dependent chain of ~1 700 vector instructions per env step -- and a batch of
// B trajectories is B / 3 waves.  Below ~12 000 trajectories those are fewer than the 3 x 1 024 waves the 1 024 SIMDs of an
// MI355X hold, at 4 096 (BASELINE configs 2 and 4) 1 366 waves: a third of the SIMDs carries two chains, the others one, and the
// launch lasts as long as two chains back to back (profiles/r05_shards.txt: 108.7 us at 4 096, 80 us for every B <= 2 048 -- the
// length of ONE chain; profiles/r05_pmc_sq_irl_step.txt: the vector unit is active ~35 % of the launch).
// Here lane = (row i, part k), k = 0, 1, 2 (63 lanes; lane 63 shadows lane 62 and produces no output):
//   * sampling: the row's six units of work -- the five quads at columns 4 q and the trailing single element -- go two to a lane:
//     lane k draws quads 2 k and 2 k + 1, lane 2 quad 4 and the tail.  Every unit is keyed exactly as in the packed kernel (Philox
//     block 0 of the quad's first element id i d + 4 q; the tail's Box-Muller pair by the EVEN step, cosine / first integer on even
//     steps, sine / second integer on odd ones -- sample_tail1), so the SAME actions are drawn.  The tail runs through the quad
//     code with per-lane element ids, step key and validity (sample_elems_gq): one instruction stream for all three lanes;
//   * row sums S, A, D, g: the packed kernel folds the units' fp32 sums into fp64 in unit order; the same chain runs ACROSS the
//     three lanes here (two DPP wave_shr:1 hand-overs), so the row totals have the same bits; they end up on lane k = 2, which runs
//     the per-row epilogue (1 / S, ln S, digamma(A));
//   * column pass (pi' = P^T pi, reward sums): lane (column j, k) walks rows 7 k .. 7 k + 6, lane k = 0 folds the three group
//     sums in group order -- the tree col_group_rows (mfg_device.h) gives the packed kernels at d = 21;
//   * value terms, per-trajectory sums, TD error: the packed kernel's code on the lanes k = 0 (and k = 1 for V of the start state,
//     in the same pass), same association.
// Every output (pi_traj, rewards, delta, g, P) is therefore bit for bit what k_core_small<SAMPLE, TD, MIXED, 21> writes for the
// same trajectory: which mapping a launch takes is a function of the batch a rank holds and must not show in the results
// (world-size invariance; tests/test_gpu_row3.py compares the two kernels with array_equal).
// Measured (profiles/r06_ab_*): a wave's chain is 1.65x shorter (two quads instead of five and a half -- but the
// epilogue, the cross-lane folds, the per-trajectory sums and the value terms are per wave, not per lane); the wave count is 3x; a
// wave executes 878 vector instructions per env step against 568 per trajectory of the packed kernel.  Updates are 33-36 % shorter at
// <= 1 024 trajectories, 17 % at 2 048, 5-10 % at 4 096; above that the packed kernel wins.  Selected by launch_core_small while the
// batch fits one resident round at four waves per SIMD (core_row3_wanted).
#include <atomic>
#include <stdlib.h>

#include "mfg_core.h"

namespace mfg {

namespace {
// geometry of the mapping for a compile-time d: lanes per matrix row, quad rounds per lane, copies of the last state entry behind
// the state vector (the lanes that draw a row's last unit read "four columns" from its first column on like every other lane)
template <int D> struct R3Geo;
template <> struct R3Geo<21> { static constexpr int LPR = 3, ROUNDS = 2, NCOPY = 3; };   // 5 quads + 1 trailing element = 6 units
template <> struct R3Geo<15> { static constexpr int LPR = 4, ROUNDS = 1, NCOPY = 1; };   // 3 quads + a 3-element remainder = 4 units
// LDS of one wave: nothing but the critic weights is shared by the block, so every barrier of a step is wave local
template <int D>
struct alignas(16) R3Wave {
  double2 q64[D];            // {pi_k as fp64, 1 / S_k of row k (fp32 bits in the low word of .y)}: one broadcast read per row
  double red[4][D];          // per-entry terms of reward / score / V(next) / V(start), summed by eight lanes
  double tot[8];             // even / odd partial sums of the four per-trajectory sums
  float tile[D * D + 3];     // gamma variates (P when it is written out) of the step, row-major, unpadded
  float pis[D + 3];          // state; the entries behind it repeat entry D - 1 (R3Geo::NCOPY of them are written)
  float pex[D + 3];          // E_j = e^{theta (pi_j - 1/2)} (separable exponential, mfg_device.h), same layout
  float pin[2 * D];          // next state, doubled (circulant value form: vec[i + m] needs no modulo)
  float pst[2 * D];          // the rollout's start state, doubled
  float pad[(4 - (D * D + 3 + 2 * (D + 3) + 4 * D) % 4) % 4 + 4];
};
static_assert(sizeof(R3Wave<21>) % 16 == 0 && sizeof(R3Wave<15>) % 16 == 0, "wave regions must keep the 16-byte alignment of q64");
template <int D> constexpr size_t r3_wl_bytes() { return (size_t)((D * (D + 1) / 2 + D + 1 + 1) & ~1) * 8; }   // critic weights, fp64
}  // namespace

template <int D> inline size_t core_row3_lds() { return r3_wl_bytes<D>() + (size_t)WAVES * sizeof(R3Wave<D>); }

// DPP wave shifts by one lane: shr -> lane l takes lane l - 1 (lane 0: zero), shl -> lane l takes lane l + 1 (lane 63: zero)
__device__ __forceinline__ double r3_shr1(double v) { return dpp_mov_f64<0x138, 0xF>(v); }
__device__ __forceinline__ double r3_shl1(double v) { return dpp_mov_f64<0x130, 0xF>(v); }
__device__ __forceinline__ float r3_shr1(float v) { return dpp_mov_f32<0x138, 0xF>(v); }
__device__ __forceinline__ float r3_shl1(float v) { return dpp_mov_f32<0x130, 0xF>(v); }

#ifndef MFG_ROW3_WAVES
#define MFG_ROW3_WAVES 4   // waves per SIMD the kernel is register-capped for (128 VGPRs): 4 096 trajectories = one resident round
#endif
// STEP: 0 plain; 1 IRL env step with the previous step's partial rows (CoreArgs::step_rows: every sampling wave forms theta from
// their column F, the grid's last core_step_red_blocks(F + 3) blocks reduce all columns and publish the update -- the code of
// k_core_small<..., STEP = 1>, the same functions: the same bits); 2 an IRL episode's FIRST env step (nothing to reduce; the start
// states are also written to pi_start_out)
template <int D, bool TD, int STEP = 0>
__global__ __launch_bounds__(BLOCK, MFG_ROW3_WAVES) void k_core_row3(CoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int H = (D + 1) / 2, Q = D * (D + 1) / 2, F = Q + D + 1, DD = D * D;
  constexpr int LPR = R3Geo<D>::LPR, ROUNDS = R3Geo<D>::ROUNDS, NCOPY = R3Geo<D>::NCOPY, NL = D * LPR;
  static_assert(NL <= WAVE && 4 * ROUNDS * (LPR - 1) < D && 4 * ROUNDS * LPR >= D, "the row's units fit the lanes of a row");
  const int T = a.T;
  const bool want_v = TD && a.w != nullptr;
  double* wl = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);
  R3Wave<D>& W = *reinterpret_cast<R3Wave<D>*>(smem_raw + r3_wl_bytes<D>() + (size_t)wv * sizeof(R3Wave<D>));
  // lane = LPR i + k; the lanes behind the last row shadow its last lane (row D - 1, part LPR - 1): they compute what it computes
  // and store nothing but copies of the last state entry (below)
  const int i3 = LPR == 3 ? (lane * 43) >> 7 : lane >> 2;  // lane / LPR for lane < 64
  const bool live = lane < NL;
  const int i = live ? i3 : D - 1;
  const int k = live ? lane - LPR * i3 : LPR - 1;
  const bool k0 = live && k == 0;

  // blocks that run trajectories (STEP = 1: the grid's last blocks reduce the previous env step's partial rows instead)
  const unsigned nblk = STEP == 1 ? gridDim.x - (unsigned)core_step_red_blocks(F + 3) : gridDim.x;
  if constexpr (STEP == 1) {
    if (blockIdx.x >= nblk) {
      const int64_t kc = (int64_t)(blockIdx.x - nblk) * WAVES + wv;  // a wave per column
      const int64_t FO = F + 3;
      if (kc >= FO) return;
      double old_val = 0.0;  // read first, under the row reads
      if (lane == 0) {
        if (kc < F) old_val = a.w_out[kc];
        else if (kc == F) old_val = *a.theta;
        else if (kc == F + 1 && a.pend_reward_acc) old_val = *a.pend_reward_acc;
      }
      const double gk = rows_column_sum(a.step_rows, a.step_nrows, FO, kc, lane);
      if (lane == 0) {
        const double inv = 1.0 / (double)a.B;
        a.step_G[kc] = gk;
        if (kc < F) a.w_out[kc] = updated_param(old_val, a.pend_lr_c, gk, inv);
        else if (kc == F) *a.theta_out = updated_param(old_val, a.pend_lr_a, gk, inv);
        else if (kc == F + 1 && a.pend_reward_acc) *a.pend_reward_acc = old_val + gk * inv;
      }
      return;
    }
  }
  // first trajectory's start state: issued before the weight staging / theta's row reads (its L2 / HBM latency hides behind them)
  const int64_t ntraj = a.B;
  const int64_t wstride = (int64_t)nblk * WAVES;
  int64_t b = (int64_t)blockIdx.x * WAVES + wv;
  float pi_first = 0.0f;
  if (b < ntraj) pi_first = a.pi0[core_src_row(a, b) * D + i];
  // deferred update of the previous episode (CoreArgs::pend_G), applied on the fly exactly as in k_core_small
  const bool pend = STEP != 1 && a.pend_G != nullptr && a.pend_G[F + 2] > 0.0;
  const double pinv = pend ? 1.0 / a.pend_G[F + 2] : 0.0;
  double theta = pend ? updated_param(*a.theta, a.pend_lr_a, a.pend_G[F], pinv) : *a.theta;
  if constexpr (STEP == 1)  // the one parameter sampling needs: every wave adds up column F of the previous step's rows itself
    theta = updated_param(theta, a.pend_lr_a, rows_column_sum(a.step_rows - a.step_nrows, a.step_nrows, 1, 0, lane), 1.0 / (double)a.B);
  const ThetaSplit ts = theta_split(theta, a.shift);
  report_sep_range(a.status, theta, a.shift);
  auto w_now = [&](int kk) -> double { return pend ? updated_param(a.w[kk], a.pend_lr_c, a.pend_G[kk], pinv) : a.w[kk]; };
  if (want_v) {
    // circulant layout of the quadratic weights: wl[m d + i] = w[k(min, max)] of the pair {i, i + m mod d} (see k_core_small)
    for (int kk = tid; kk < H * D; kk += BLOCK) {
      const int m = kk / D, ii = kk - m * D;
      int jj = ii + m;
      if (jj >= D) jj -= D;
      wl[kk] = w_now(feat_idx(ii < jj ? ii : jj, ii < jj ? jj : ii, D));
    }
    for (int kk = Q + tid; kk < F; kk += BLOCK) wl[kk] = w_now(kk);
  }
  if (STEP != 1 && a.pend_G != nullptr && blockIdx.x == 0) {
    // block 0 publishes the updated parameters (out of place) and books the update's mean reward
    if (a.w_out && a.w)
      for (int kk = tid; kk < F; kk += BLOCK) a.w_out[kk] = w_now(kk);
    if (tid == 0) {
      if (a.theta_out) *a.theta_out = theta;
      if (pend && a.pend_reward_acc) *a.pend_reward_acc += a.pend_G[F + 1] * pinv;
    }
  }
  __syncthreads();  // the one block barrier of the launch: wl is staged block-wide and read by every wave
  auto wave_sync = [&]() {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  };
  // value term of state entry i for the state in the doubled vector `vec` whose entry i is pi_e: k_core_small's circulant form
  auto value_term = [&](const float* vec, float pi_e) -> double {
    double c0 = 0.0, c1 = 0.0;
#pragma unroll
    for (int m = 0; m + 1 < H; m += 2) {
      c0 = fma(wl[m * D + i], (double)vec[i + m], c0);
      c1 = fma(wl[(m + 1) * D + i], (double)vec[i + m + 1], c1);
    }
    if (H & 1) c0 = fma(wl[(H - 1) * D + i], (double)vec[i + H - 1], c0);
    const double col = c0 + c1;
    return (double)pi_e * (col + wl[Q + i]);
  };
  const bool ext = a.reward_kind == MFG_REWARD_EXTERNAL;

  for (; b < ntraj; b += wstride) {
    float pi_i = pi_first;
    if (b + wstride < ntraj) pi_first = a.pi0[core_src_row(a, b + wstride) * D + i];  // the next trajectory's start state
    if (k0 && a.pi_traj) a.pi_traj[b * (int64_t)(T + 1) * D + i] = pi_i;
    if constexpr (STEP == 2) {
      if (k0 && a.pi_start_out) a.pi_start_out[b * D + i] = pi_i;
    }
    double v_cur = 0.0, discount = 1.0;  // meaningful on lane 0 only
    const uint64_t traj = a.traj_offset + (uint64_t)b;
    const uint32_t erow = (uint32_t)(i * D);
    for (int s = 0; s < T; ++s) {
      wave_sync();  // everything of the previous step has been read
      const uint32_t step = a.first_step + (uint32_t)s;
      // ---- state of the step: every lane of a row keeps pi_i and forms F_i; the lane k = 0 publishes pi_i, E_i
      const float Ei = exp_f64arg(theta * ((double)pi_i - SEP_CENTRE));
      const float Fi = exp_f64arg(-theta * ((double)pi_i + (a.shift - SEP_CENTRE)));
      if (lane > (D - 1) * LPR && lane <= (D - 1) * LPR + NCOPY) {  // other lanes that hold the last row's state (i == D - 1):
        const int c = D + lane - (D - 1) * LPR - 1;                 // the copies behind the vector (d = 21: lanes 61 .. 63 -> 21 .. 23)
        W.pis[c] = pi_i;
        W.pex[c] = Ei;
      }
      if (k0) {
        W.pis[i] = pi_i;
        W.pex[i] = Ei;
        W.q64[i].x = (double)pi_i;
        if (want_v && s == 0) {
          W.pst[i] = pi_i;
          W.pst[D + i] = pi_i;
        }
      }
      wave_sync();
      // ---- sampling: ROUNDS units per lane through the quad code
      using TT = float;
      const float pas = pi_i + ts.sh;  // row operand of the separable form (policy_setup_sep)
      float pa[4], fi[4];
      float yk[ROUNDS][4];
      float Sq[ROUNDS];
      TT Aq[ROUNDS], Dq[ROUNDS], Gq[ROUNDS];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pa[e] = pas;
        fi[e] = Fi;
      }
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) Sq[r] = 0.0f, Aq[r] = 0, Dq[r] = 0, Gq[r] = 0;
      const bool wp = a.P_out != nullptr;
      const uint32_t odd = step & 1u;
      float* trow = W.tile + i * D;
      // d = 21 -- round 0: the quad at columns 8 k .. 8 k + 3 (quads 0, 2, 4 of the row); round 1: the quad at columns 8 k + 4 ..
      // 8 k + 7 (quads 1, 3) -- or, on the lanes k = 2, the row's trailing element: its pair is keyed by the EVEN step, element 0 of
      // the pair (cosine, first 16-bit integer) is the draw of an even step, element 1 (sine, second integer) of an odd one; the
      // other elements do not exist there (sample_tail1's arithmetic).  d = 15 -- one round: the quad at columns 4 k, on the lanes
      // k = 3 the 3-element remainder (sample_elems<3>'s arithmetic: the fourth draw is discarded).  ONE copy of the quad code (a
      // rolled loop: two copies interleaved by the scheduler cost 200 spilled registers at the 128-register cap).
      if constexpr (ROUNDS == 1) __builtin_amdgcn_sched_barrier(0);  // (one round is no loop: keep the scheduler from interleaving the
                                                                      //  quad code with the staging before / the folds behind it)
#pragma unroll 1
      for (int r = 0; r < ROUNDS; ++r) {
        float pj[4], ej[4], y[4];
        TT al[4], ad[4], gt[4];
        uint32_t el[4];
        bool ok[4];
        const bool last = r == ROUNDS - 1 && k == LPR - 1;   // the lane's unit is the row's LAST one (tail element / remainder)
        const uint32_t qstep = (D == 21 && last) ? (step & ~1u) : step;
        const int c0 = 4 * ROUNDS * k + 4 * r;  // (last unit: the copies behind the state vector stand in for the missing columns)
        const uint32_t elmax = erow + (uint32_t)(D - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pj[e] = W.pis[c0 + e];
          ej[e] = W.pex[c0 + e];
          const uint32_t id = erow + (uint32_t)(c0 + e);
          el[e] = id < elmax ? id : elmax;     // (d = 21: the tail's elements all carry the tail's id -- its continuation draws are keyed by it)
          ok[e] = true;  // (ALLVALID: the last-unit lanes pick their elements below; the shadow lanes mirror one of them)
        }
        sample_elems_gq<4, TD, true, true, true>(a, theta, ts, pj, ej, pa, fi, el, ok, qstep, step, traj, y, al, ad, gt);
        float Sr = y[0];
        TT Ar = 0, Dr = 0, Gr = 0;
        if (TD) Ar = al[0], Dr = ad[0], Gr = gt[0];
        float S3 = 0.0f;
        TT A3 = 0, D3 = 0, G3 = 0;
#pragma unroll
        for (int e = 1; e < 4; ++e) {
          if (e == 3) S3 = Sr, A3 = Ar, D3 = Dr, G3 = Gr;  // the sums of the first three elements (the d = 15 remainder)
          Sr += y[e];
          if (TD) Ar += al[e], Dr += ad[e], Gr += gt[e];
        }
        if (r == ROUNDS - 1) {
          if constexpr (D == 21) {
            // tail lanes: the unit is ONE element -- element 0 of the pair on even steps, element 1 on odd ones (wave-uniform
            // choice); elements 2, 3 (and the other one of the pair) do not exist: their draws are discarded
            const float ys = odd ? y[1] : y[0];
            Sr = last ? ys : Sr;
            if (TD) {
              const TT as = odd ? al[1] : al[0], ds = odd ? ad[1] : ad[0], gs = odd ? gt[1] : gt[0];
              Ar = last ? as : Ar;
              Dr = last ? ds : Dr;
              Gr = last ? gs : Gr;
            }
            y[0] = last ? ys : y[0];  // what the tail lanes store at column 20 (they store no other element)
          } else {
            // remainder lanes: three elements, the fourth does not exist
            Sr = last ? S3 : Sr;
            if (TD) {
              Ar = last ? A3 : Ar;
              Dr = last ? D3 : Dr;
              Gr = last ? G3 : Gr;
            }
          }
        }
        // (r is the counter of a rolled loop: the per-round sums are parked under a wave-uniform branch with constant indices --
        //  through selects they cost the d = 21 kernel 12 %; the variates of a written-out P go to yk[r]: with two rounds that
        //  is scratch, used in that mode only -- as registers they cost 16 more spilled registers in every mode)
        constexpr int NLAST = D == 21 ? 1 : 3;  // elements of the row's last unit
        if (ROUNDS == 1 || r == 0) {
          Sq[0] = Sr;
          if (TD) Aq[0] = Ar, Dq[0] = Dr, Gq[0] = Gr;
        } else {
          Sq[ROUNDS - 1] = Sr;
          if (TD) Aq[ROUNDS - 1] = Ar, Dq[ROUNDS - 1] = Dr, Gq[ROUNDS - 1] = Gr;
        }
        if (wp) {  // P is written out: the variates wait for the row's normaliser
#pragma unroll
          for (int e = 0; e < 4; ++e) yk[ROUNDS == 1 ? 0 : r][e] = y[e];
        }
        if (!wp && live) {
#pragma unroll
          for (int e = 0; e < NLAST; ++e) trow[c0 + e] = y[e];
          if (!last) {
#pragma unroll
            for (int e = NLAST; e < 4; ++e) trow[c0 + e] = y[e];
          }
        }
      }
      if constexpr (ROUNDS == 1) __builtin_amdgcn_sched_barrier(0);
      const bool lastk = k == LPR - 1;
      // ---- row totals: the packed kernel's chain over the row's units in unit order, in fp64 -- ((((0 + u0) + u1) + u2) + ..) --
      // across the lanes of the row (LPR - 1 DPP hand-overs); the totals end up on the lane k = LPR - 1
      auto chain = [&](const float* rr) -> double {
        double x[ROUNDS];
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q) x[q] = (double)rr[q];
        double c = x[0];                          // k = 0: 0 + u0 (exact)
#pragma unroll
        for (int q = 1; q < ROUNDS; ++q) c += x[q];
#pragma unroll
        for (int j = 1; j < LPR; ++j) {           // k = j: the running sum of the lanes before it + its own units
          c = r3_shr1(c) + x[0];
#pragma unroll
          for (int q = 1; q < ROUNDS; ++q) c += x[q];
        }
        return c;
      };
      const double Ssum = chain(Sq);
      double A = 0.0, D_ = 0.0, gacc = 0.0;
      if (TD) {
        A = chain(Aq);
        D_ = chain(Dq);
        gacc = chain(Gq);
        gacc *= LN2;  // the element terms were summed in log2 units (policy_terms)
      }
      // ---- per-row epilogue (meaningful on the lanes k = LPR - 1, which hold the totals)
      const float inv32 = fast_rcp_f32_of_f64(Ssum);
      if (TD) {
        gacc -= fast_log_f64(Ssum) * D_;
        gacc = fma(digamma_pos_mixed(A), D_, gacc);
      }
      if (wp) {  // P is written out: the tile itself is normalised (the copy-out reads it), the column pass multiplies by 1
        float nrm = inv32, t = inv32;
#pragma unroll
        for (int j = 1; j < LPR; ++j) {           // lane k = LPR - 1 - j takes the value of the row's last lane
          t = r3_shl1(t);
          nrm = k == LPR - 1 - j ? t : nrm;
        }
        if (live) {
#pragma unroll
          for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = 4 * ROUNDS * k + 4 * r + e;
              if (c < D) trow[c] = yk[r][e] * nrm;   // (the last unit's lanes: its existing elements only)
            }
        }
      }
      if (live && lastk) {
        *reinterpret_cast<float*>(&W.q64[i].y) = wp ? 1.0f : inv32;
        if (TD) W.red[1][i] = gacc;
      }
      wave_sync();  // tile and row normalisers complete
      // ---- column pass: lane (column i, part k) walks the rows of group k (col_group_rows: 7 k .. 7 k + 6 at d = 21; 4 k .. 4 k + 3 at
      // d = 15, the last group one row short); lane k = 0 folds the group sums in group order
      double pa_ = 0.0, p1 = 0.0, p2 = 0.0;
      {
        const float* tcol = W.tile + i;
        constexpr int GR = col_group_rows(D);
        static_assert(GR * LPR >= D && GR * (LPR - 1) < D, "one group of rows per lane of a column");
        constexpr int LASTN = D - GR * (LPR - 1);  // rows of the last group
        const int r0 = GR * k;
#pragma unroll
        for (int r = 0; r < GR; ++r) {
          if (r < LASTN || k < LPR - 1) {          // (a row beyond the matrix: the last group's lanes skip it)
            const double2 e = W.q64[r0 + r];
            const double p = (double)(tcol[(r0 + r) * D] * __int_as_float(__double2loint(e.y)));
            col_walk_row(r == 0, p * e.x, p, pa_, p1, p2);
          }
        }
      }
      double acc = pa_, s1 = p1, s2 = p2;
      {
        double ta = pa_, tb = p1, tc = p2;
#pragma unroll
        for (int j = 1; j < LPR; ++j) {            // group j arrives from lane + j
          ta = r3_shl1(ta);
          tb = r3_shl1(tb);
          tc = r3_shl1(tc);
          acc += ta;
          s1 += tb;
          s2 += tc;
        }
      }
      const double pid = (double)pi_i;
      double rcol = 0.0;
      if (a.reward_kind == MFG_REWARD_MFG_AC2) rcol = fma(pid, s1, -s2);
      if (a.reward_kind == MFG_REWARD_SYNTHETIC) rcol = s1;
      float pi_n = (float)acc;  // lanes k = 0
      {
        float u = pi_n;
#pragma unroll
        for (int j = 1; j < LPR; ++j) {            // lane k = j takes the value of the row's first lane
          u = r3_shr1(u);
          pi_n = k == j ? u : pi_n;
        }
      }
      if (k0) {
        W.pin[i] = pi_n;
        W.pin[D + i] = pi_n;
        if (!ext) W.red[0][i] = rcol;
      }
      if (wp) {
        // copy-out of the trajectory's matrix (contiguous in LDS and in P_out); all LDS reads issued before the stores
        wave_sync();
        constexpr int NIT = (DD + WAVE - 1) / WAVE;
        float* dst = a.P_out + (b * (int64_t)T + s) * DD;
        float v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
          const int kk = lane + u * WAVE;
          v[u] = W.tile[kk < DD ? kk : 0];
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
          const int kk = lane + u * WAVE;
          if (kk < DD) dst[kk] = v[u];
        }
      }
      if (want_v) {
        wave_sync();  // pin complete
        // V(next) on the lanes k = 0 and -- at the rollout's first step -- V(start) on the lanes k = 1, in the same pass
        const bool vs = k == 1 && s == 0;
        if (live && (k == 0 || vs)) W.red[vs ? 3 : 2][i] = value_term(vs ? W.pst : W.pin, vs ? pi_i : pi_n);
      }
      wave_sync();
      // ---- per-trajectory sums: EIGHT lanes (k = 0, i < 8) add the terms of parity p of quantity q -- k_core_small's association
      if (k0 && i < 8) {
        const int q = i >> 1, pp = i & 1;
        const bool need = q == 0 ? !ext : (q == 1 ? want_v : (q == 2 ? (TD && a.g != nullptr) : (want_v && s == 0)));
        double x = 0.0;
        if (need) {
          const double* src = W.red[q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 1 : 3))];
#pragma unroll
          for (int kk0 = 0; kk0 < (D + 1) / 2; ++kk0) {
            const int kk = 2 * kk0 + pp;
            const double v = src[kk < D ? kk : 0];
            x += kk < D ? v : 0.0;
          }
        }
        W.tot[i] = x;
      }
      wave_sync();
      if (lane == 0) {
        double r0 = 0.0, r1 = 0.0;
        if (ext) {
          r0 = a.reward_in ? (double)a.reward_in[b * T + s] : 0.0;
        } else {
          r0 = W.tot[0];
          r1 = W.tot[1];
        }
        double r = r0 + r1;
        if (a.reward_kind == MFG_REWARD_SYNTHETIC) r *= -0.5;
        if (a.reward_out) a.reward_out[b * T + s] = (float)r;
        if (want_v) {
          const double v0 = W.tot[2], v1 = W.tot[3];
          const double v_next = (v0 + v1) + wl[Q + D];
          if (s == 0) {
            const double u0 = W.tot[6], u1 = W.tot[7];
            v_cur = (u0 + u1) + wl[Q + D];
          }
          const double gd = a.discount_pow ? discount : a.gamma;
          const double del = r + gd * v_next - v_cur;
          if (a.delta) a.delta[b * T + s] = del;
          v_cur = v_next;
          discount *= a.gamma;
        }
      }
      if (TD && lane == LPR && a.g) {  // (row 1, part 0): the lane that sums the score in the packed kernel
        const double g0 = W.tot[4], g1 = W.tot[5];
        a.g[b * T + s] = g0 + g1;
      }
      if (k0 && a.pi_traj) a.pi_traj[(b * (int64_t)(T + 1) + s + 1) * D + i] = pi_n;
      pi_i = pi_n;
    }
    if (k0 && a.pi_next_out) a.pi_next_out[b * D + i] = pi_i;
  }
}

// lane mapping of the d = 21 sampling launches (mfg_set_core_mapping, include/mfg_hip.h): 0 by batch size, 1 always the packed
// kernel, 2 the one-trajectory-per-wave kernel wherever it supports the launch.  A measurement / test hook: both give the same bits.
static std::atomic<int> g_core_mapping{0};
int core_mapping_set(int mode) { return g_core_mapping.exchange(mode < 0 || mode > 2 ? 0 : mode); }

bool core_row3_wanted(const CoreArgs& a, bool sample, bool td, bool fast, int num_cus) {
  // not the per-step SUMS variant (k_core_small<..., SUMS>: batch sums of 12- / 16-trajectory tiles on the matrix cores); the STEP
  // variants (IRL env step) exist here too -- there part_rows aliases step_G
  if ((a.d != 21 && a.d != 15) || !sample || !fast) return false;
  if (a.step_nrows == 0 && a.part_rows != nullptr) return false;
  if (a.step_nrows != 0 && !td) return false;
  const int mode = g_core_mapping.load();
  if (mode) return mode == 2;
  // d = 21: one resident round at MFG_ROW3_WAVES waves per SIMD (4 SIMDs per CU): 4 096 trajectories on 256 CUs -- rollouts, single
  // steps and the IRL step variants alike (profiles/r06_ab_shards_mapping*.txt, r06_t1_mapping_probe.txt).
  // d = 15 (profiles/r06_ab_shards_d15.txt, r06_irl_d15_probe.txt): the packed kernel holds FOUR trajectories per wave and its chain
  // is short where the policy's shapes stay large (the IRL workload), so the new mapping pays later: single-step launches only
  // while a SIMD holds one wave (4 per CU), rollouts that write their actions out up to three waves per SIMD, others up to four.
  const int64_t per_cu = a.d == 21 ? 4 * MFG_ROW3_WAVES : (a.T == 1 ? 4 : (a.P_out ? 12 : 16));
  return a.B <= (int64_t)num_cus * per_cu;
}

template <int D>
static int launch_core_row3_d(const CoreArgs& a, bool td, int num_cus, hipStream_t st) {
  const size_t lds = core_row3_lds<D>();
  const int step = a.step_nrows > 0 ? 1 : (a.step_nrows < 0 ? 2 : 0);
  const int slot = step ? 1 + step : (td ? 1 : 0);   // 0: <D, false>, 1: <D, true>, 2: <D, true, 1>, 3: <D, true, 2>
  static std::atomic<int> cached_bpc[4][64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int bpc = cached_bpc[slot][dev].load();
  if (bpc == 0) {
    int n = 0;
    hipError_t e;
    if (slot == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_core_row3<D, false>, BLOCK, lds);
    else if (slot == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_core_row3<D, true>, BLOCK, lds);
    else if (slot == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_core_row3<D, true, 1>, BLOCK, lds);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_core_row3<D, true, 2>, BLOCK, lds);
    if (e != hipSuccess || n < 1) n = 1;
    bpc = n;
    cached_bpc[slot][dev].store(n);
  }
  const int grid = core_grid(a.B, WAVES, bpc * MFG_CORE_OVERSUBSCRIBE, num_cus);
  if (slot == 0) hipLaunchKernelGGL((k_core_row3<D, false>), dim3(grid), dim3(BLOCK), lds, st, a);
  else if (slot == 1) hipLaunchKernelGGL((k_core_row3<D, true>), dim3(grid), dim3(BLOCK), lds, st, a);
  else if (slot == 2)  // (the blocks that reduce the previous env step's partial rows ride behind the sampling blocks)
    hipLaunchKernelGGL((k_core_row3<D, true, 1>), dim3(grid + core_step_red_blocks(D * (D + 1) / 2 + D + 1 + 3)), dim3(BLOCK), lds, st, a);
  else hipLaunchKernelGGL((k_core_row3<D, true, 2>), dim3(grid), dim3(BLOCK), lds, st, a);
  return MFG_OK;
}

int launch_core_row3(const CoreArgs& a, bool td, int num_cus, hipStream_t st) {
  return a.d == 21 ? launch_core_row3_d<21>(a, td, num_cus, st) : launch_core_row3_d<15>(a, td, num_cus, st);
}

}  // namespace mfg
