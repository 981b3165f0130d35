// HIP kernels + C ABI of the MFG hot path for MI355X (gfx950 / CDNA4, wave64).
// See include/mfg_hip.h for the contract and DESIGN.md for the layout / roofline notes.
//
// Two work decompositions:
//   small d (d <= 64): G = 64/d trajectories packed per wavefront; the d x d action matrix of each
//       trajectory is staged in LDS (per block tile).  HBM-bound step kernel: lane = (trajectory, column j).
//       Compute-bound sampler / TD kernels: lane = (trajectory, row i) so that all per-row Dirichlet
//       quantities (row sum of gamma variates, sum_j alpha_ij, ...) stay lane-local.
//   large d (d > 64): one wavefront per trajectory, lanes own columns, rows are streamed from HBM
//       with coalesced loads; per-row quantities use wavefront reductions.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <dlfcn.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "../../include/mfg_hip.h"
#include "mfg_core.h"

using namespace mfg;

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
namespace mfg {
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }
}  // namespace mfg
static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MFG_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MFG_OK;
}
#define REQUIRE(cond, msg) \
  do {                     \
    if (!(cond)) return fail(MFG_EINVAL, "%s", msg); \
  } while (0)

static inline hipStream_t S(mfg_stream_t s) { return (hipStream_t)s; }

// CU count of the CURRENT device (cached per device: one context may drive several GPUs from one process)
static int num_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int c = cus[dev].load(std::memory_order_relaxed);
  if (!c) {
    hipDeviceProp_t p;
    c = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}


// ---------------------------------------------------------------------------------------------
// trivial kernels: gather, alpha, features, dirichlet_from_gamma, philox_raw, apply_update
// ---------------------------------------------------------------------------------------------
__global__ void k_gather_start(const float* __restrict__ mat, int64_t num_start, const int32_t* __restrict__ idx, int64_t B,
                               int d, float* __restrict__ out) {
  const int64_t n = B * d;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / d;
    const int j = (int)(e - b * d);
    out[e] = mat[start_row(idx[b], num_start) * d + j];
  }
}

// a9: the start-state draw itself (start_draw_row, mfg_device.h) as a launch of its own: idx[b] and / or the gathered rows
__global__ void k_draw_start(const float* __restrict__ mat, int64_t num_start, int64_t B, int d, uint64_t seed, uint32_t step,
                             uint64_t traj_offset, int32_t* __restrict__ idx_out, float* __restrict__ out) {
  const int64_t n = out ? B * d : B;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = out ? e / d : e;
    const int j = out ? (int)(e - b * d) : 0;
    const int64_t row = start_draw_row(seed, step, traj_offset + (uint64_t)b, num_start);
    if (idx_out && j == 0) idx_out[b] = (int32_t)row;
    if (out) out[e] = mat[row * d + j];
  }
}

__global__ void k_alpha(const float* __restrict__ pi, int64_t B, int d, const double* __restrict__ theta_p,
                        double shift, double* __restrict__ alpha, double* __restrict__ deriv) {
  const double theta = *theta_p;
  const int64_t dd = (int64_t)d * d;
  const int64_t n = B * dd;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / dd;
    const int r = (int)(e - b * dd);
    const int i = r / d, j = r - i * d;
    const double x = (double)pi[b * d + j] - (double)pi[b * d + i] - shift;
    double sp, sg;
    softplus_sigmoid(theta * x, sp, sg);
    if (alpha) alpha[e] = sp;
    if (deriv) deriv[e] = x * sg;
  }
}

__global__ void k_features(const float* __restrict__ pi, int64_t B, int d, double* __restrict__ phi) {
  const int64_t dd = (int64_t)d * d;
  const int64_t F = (int64_t)d * (d + 1) / 2 + d + 1;
  const int64_t Q = (int64_t)d * (d + 1) / 2;
  const int64_t n = B * dd;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / dd;
    const int r = (int)(e - b * dd);
    const int i = r / d, j = r - i * d;
    const double pi_i = (double)pi[b * d + i], pi_j = (double)pi[b * d + j];
    if (j >= i) phi[b * F + feat_idx(i, j, d)] = pi_i * pi_j;
    if (i == 0) phi[b * F + Q + j] = pi_j;
    if (r == 0) phi[b * F + Q + d] = 1.0;
  }
}

__global__ void k_dirichlet_from_gamma(const float* __restrict__ y, int64_t rows, int d, float* __restrict__ P) {
  // one wavefront per row of gamma variates
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE;
  const int64_t nw = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nw) {
    double s = 0.0;
    for (int j = lane; j < d; j += WAVE) {
      float v = y[r * d + j];
      if (v == 0.0f) v = ZERO_GAMMA_REPLACEMENT;
      s += (double)v;
    }
    s = wave_sum(s);
    const double inv = 1.0 / s;
    for (int j = lane; j < d; j += WAVE) {
      float v = y[r * d + j];
      if (v == 0.0f) v = ZERO_GAMMA_REPLACEMENT;
      P[r * d + j] = (float)((double)v * inv);
    }
  }
}

__global__ void k_philox_raw(uint64_t seed, uint32_t first, uint32_t c1, uint32_t c2, uint32_t c3, int64_t n,
                             uint32_t* __restrict__ out) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    u32x4 c{first + (uint32_t)e, c1, c2, c3};
    const u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    out[4 * e + 0] = r.x;
    out[4 * e + 1] = r.y;
    out[4 * e + 2] = r.z;
    out[4 * e + 3] = r.w;
  }
}

__global__ void k_apply_update(const double* __restrict__ G, int64_t F, double lr_c, double lr_a, double* __restrict__ w,
                               double* __restrict__ theta, double* __restrict__ reward_acc) {
  const double count = G[F + 2];
  if (!(count > 0.0)) return;
  const double inv = 1.0 / count;
  if (reward_acc && blockIdx.x == 0 && threadIdx.x == 0) *reward_acc += G[F + 1] * inv;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < F; k += (int64_t)gridDim.x * blockDim.x)
    w[k] = updated_param(w[k], lr_c, G[k], inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) *theta = updated_param(*theta, lr_a, G[F], inv);
}

// the same update out of place: (w_out, theta_out) = (w_in, theta_in) + lr G / count -- the large-d form of the deferred
// update of mfg_train_rollout_deferred (at d <= 64 the rollout kernel applies it while staging its weights)
__global__ void k_apply_update_oop(const double* __restrict__ G, int64_t F, double lr_c, double lr_a, const double* __restrict__ w_in,
                                   const double* __restrict__ theta_in, double* __restrict__ w_out, double* __restrict__ theta_out,
                                   double* __restrict__ reward_acc) {
  const double count = G[F + 2];
  const bool on = count > 0.0;
  const double inv = on ? 1.0 / count : 0.0;
  if (on && reward_acc && blockIdx.x == 0 && threadIdx.x == 0) *reward_acc += G[F + 1] * inv;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < F; k += (int64_t)gridDim.x * blockDim.x)
    w_out[k] = on ? updated_param(w_in[k], lr_c, G[k], inv) : w_in[k];
  if (blockIdx.x == 0 && threadIdx.x == 0) *theta_out = on ? updated_param(*theta_in, lr_a, G[F], inv) : *theta_in;
}

// ---------------------------------------------------------------------------------------------
// a11: JSD, one wavefront per pair of rows
// ---------------------------------------------------------------------------------------------
__global__ void k_jsd(const float* __restrict__ p, const float* __restrict__ q, int64_t B, int d, double* __restrict__ out) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE;
  const int64_t nw = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t b = wave; b < B; b += nw) {
    // zeros -> 1e-100, M = (P+Q)/2 from the un-normalised vectors, entropy() renormalises P, Q and M
    double sp = 0, sq = 0;
    for (int j = lane; j < d; j += WAVE) {
      double a = p[b * d + j], c = q[b * d + j];
      if (a == 0.0) a = 1e-100;
      if (c == 0.0) c = 1e-100;
      sp += a;
      sq += c;
    }
    sp = wave_sum(sp);
    sq = wave_sum(sq);
    const double sm = 0.5 * (sp + sq);
    double acc = 0;
    for (int j = lane; j < d; j += WAVE) {
      double a = p[b * d + j], c = q[b * d + j];
      if (a == 0.0) a = 1e-100;
      if (c == 0.0) c = 1e-100;
      const double m = 0.5 * (a + c) / sm;
      const double pn = a / sp, qn = c / sq;
      acc += pn * log(pn / m) + qn * log(qn / m);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[b] = 0.5 * acc;
  }
}

// ---------------------------------------------------------------------------------------------
// f1 (importance weights of the max-ent IRL loss, ac_irl.py:270-289 calc_pdf_action / :324-379 calc_z): log-density of
// the product-Dirichlet policy, one wavefront per (sample n, policy k):
//   log q_k(P_n | pi_n) = sum_i [ lgamma(sum_j a_ij) - sum_j lgamma(a_ij) + sum_j (a_ij - 1) ln P_ij ],
//   a_ij = max(alpha_floor, alpha_scale * ln(1 + exp(theta_k (pi_j - pi_i - shift)))).
// Log space replaces the reference's divide-by-normaliser trick (pdf / c before the products).  fp64 throughout.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_policy_logpdf(const float* __restrict__ pi, const float* __restrict__ P, int64_t N,
                                                         int d, const double* __restrict__ thetas, int K, double shift,
                                                         double alpha_scale, double alpha_floor, double p_floor,
                                                         double* __restrict__ out) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE;
  const int64_t nw = (int64_t)gridDim.x * blockDim.x / WAVE;
  const int dd = d * d;
  for (int64_t u = wave; u < N * K; u += nw) {
    const int64_t n = u / K;
    const int k = (int)(u - n * K);
    const double th = thetas[k];
    const float* pn = pi + n * d;
    const float* Pn = P + n * (int64_t)dd;
    double acc = 0.0;
    for (int e = lane; e < dd; e += WAVE) {
      const int i = e / d, j = e - i * d;
      double sp, sg;
      softplus_sigmoid(th * ((double)pn[j] - (double)pn[i] - shift), sp, sg);
      double al = alpha_scale * sp;
      if (al < alpha_floor) al = alpha_floor;
      double pv = (double)Pn[e];
      if (pv < p_floor) pv = p_floor;
      acc += (al - 1.0) * log(pv) - lgamma(al);
    }
    for (int i = lane; i < d; i += WAVE) {
      double A = 0.0;
      for (int j = 0; j < d; ++j) {
        double sp, sg;
        softplus_sigmoid(th * ((double)pn[j] - (double)pn[i] - shift), sp, sg);
        double al = alpha_scale * sp;
        if (al < alpha_floor) al = alpha_floor;
        A += al;
      }
      acc += lgamma(A);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[u] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// f3: backward value recursion of mfg_synthetic (mfg_synthetic.py:768-774) and the two consistency metrics
// of evaluate_synthetic (:776-790, sum_ij |P_ij - value_ij|) / evaluate_synthetic_JSD (:858-880, sum_i JSD(P_i,
// implied row i) with entries <= 0 -> 1e-100).  One wavefront per trajectory, lane = row i, reverse-time scan
// with V^{n+1} in LDS.  Evaluation-only: rows are read strided (L2 resident), fp64 throughout.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_backward_value(const float* __restrict__ P, int64_t B, int T, int d,
                                                          double* __restrict__ V, double* __restrict__ diff_l1,
                                                          double* __restrict__ diff_jsd) {
  extern __shared__ __attribute__((aligned(16))) double smd[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  double* vn1 = smd + wv * 2 * d;  // V^{n+1}
  double* vn = vn1 + d;            // V^{n}
  const int64_t nw = (int64_t)gridDim.x * WAVES;
  for (int64_t b = (int64_t)blockIdx.x * WAVES + wv; b < B; b += nw) {
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < d; i += WAVE) {
      vn1[i] = 0.0;
      V[(b * (T + 1) + T) * d + i] = 0.0;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int n = T - 1; n >= 0; --n) {
      const float* Pn = P + (b * T + n) * (int64_t)d * d;
      double sumV = 0.0;
      for (int i = lane; i < d; i += WAVE) {
        const float* row = Pn + (int64_t)i * d;
        double r2 = 0.0, acc = 0.0;
        for (int j = 0; j < d; ++j) {
          const double p = (double)row[j];
          r2 = fma(p, p, r2);
          acc = fma(p, vn1[j], acc);
        }
        const double v = fma(-0.5, r2, acc);
        vn[i] = v;
        V[(b * (T + 1) + n) * d + i] = v;
        sumV += v;
      }
      sumV = wave_sum(sumV);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      double l1 = 0.0, js = 0.0;
      for (int i = lane; i < d; i += WAVE) {
        const float* row = Pn + (int64_t)i * d;
        const double vi = vn[i];
        const double diag = 1.0 - (sumV - (double)d * vi);
        double sp = 0.0, sq = 0.0;
        for (int j = 0; j < d; ++j) {
          const double p = (double)row[j];
          const double val = (j == i) ? diag : vn[j] - vi;
          l1 += fabs(p - val);
          sp += (p <= 0.0) ? 1e-100 : p;
          sq += (val <= 0.0) ? 1e-100 : val;
        }
        if (diff_jsd) {
          const double sm = 0.5 * (sp + sq);
          double kl = 0.0;
          for (int j = 0; j < d; ++j) {
            double p = (double)row[j];
            double q = (j == i) ? diag : vn[j] - vi;
            if (p <= 0.0) p = 1e-100;
            if (q <= 0.0) q = 1e-100;
            const double m = 0.5 * (p + q) / sm;
            const double pn = p / sp, qn = q / sq;
            kl += pn * log(pn / m) + qn * log(qn / m);
          }
          js += 0.5 * kl;
        }
      }
      l1 = wave_sum(l1);
      if (diff_jsd) js = wave_sum(js);
      if (lane == 0) {
        diff_l1[b * T + n] = l1;
        if (diff_jsd) diff_jsd[b * T + n] = js;
      }
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < d; i += WAVE) vn1[i] = vn[i];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// a5: V(pi) = phi(pi).w, one wavefront per trajectory, lanes own columns c, rows i <= c.
// w rows are contiguous in k for fixed i, so the loads are coalesced (L2 resident).
// ---------------------------------------------------------------------------------------------
__global__ void k_value(const float* __restrict__ pi, const double* __restrict__ w, int64_t B, int d,
                        double* __restrict__ out) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE;
  const int64_t nw = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t b = wave; b < B; b += nw) {
    const double v = value_wave(pi + b * d, w, d, lane);
    if (lane == 0) out[b] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// a3+a4, small d: HBM-bound.  Block = 4 waves, tile = 4*G consecutive trajectories whose P slab
// (contiguous in HBM) is copied flat with 16-byte loads into LDS; lane = (trajectory, column j).
// Per element the lane does: cvt, p^2, and three fp64 FMAs
//     pi'_j += p pi_i,   s1_j += pi_i p^2,   s2_j += pi_i^2 p^2        (reward_j = pi_j s1_j - s2_j)
// with (pi_i, pi_i^2) staged once per tile as fp64 pairs in LDS (one broadcast 16-byte read per row).
// D > 0: compile-time d (rows unrolled, immediate LDS offsets); D == 0: runtime d.
// ---------------------------------------------------------------------------------------------
// The P slab of the NEXT tile is prefetched into registers (PER 16-byte loads per thread, all in flight
// together) while the current tile is consumed from LDS, so each block keeps ~a tile of HBM traffic in
// flight at all times (the un-pipelined version spent 83 % of its wave cycles waiting: SQ_WAIT_ANY).
#ifndef MFG_STEP_UNROLL
#define MFG_STEP_UNROLL 3
#endif
// Streamed-once data: non-temporal loads (measured +5 % on the d=21 kernel, +8..12 % on the row kernels).
// Depth-2 register prefetch was tried and lost (3.8-4.9 TB/s): the extra 24 VGPRs cost a block per CU.
typedef float v4f_t __attribute__((ext_vector_type(4)));
#define MFG_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#ifndef MFG_STEP_WAVES
#define MFG_STEP_WAVES 7
#endif
template <int KIND, int D, int PER>
__global__ __launch_bounds__(BLOCK, (PER <= 6 ? MFG_STEP_WAVES : 4)) void k_step_small(const float* __restrict__ pi, const float* __restrict__ P,
                                                      int64_t B, int d_rt, float* __restrict__ pi_next,
                                                      float* __restrict__ reward) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = D ? D : d_rt;
  const int G = WAVE / d, TB = WAVES * G, dd = d * d;
  float* tQ = smem;                  // [TB][d] pi (fp32; widened on the fly: VALU is idle here, LDS bytes are not)
  float* tP = smem + ((TB * d + 3) & ~3);  // [TB][d][d], 16-byte aligned
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int t = lane / d, j = lane - t * d;
  const int p2 = next_pow2(d);
  const int64_t ntiles = (B + TB - 1) / TB;
  v4f_t pre[PER];
  float prepi = 0.0f;
  // prefetch of tile TT into registers (a macro, not a lambda: capturing pre[] by reference sends it to scratch)
#define MFG_STEP_PREFETCH(TT)                                                  \
  {                                                                            \
    const int64_t pb0 = (TT) * TB;                                             \
    const int pnb = (int)((B - pb0) < TB ? (B - pb0) : TB);                    \
    const int pn4 = (pnb * dd) >> 2;                                           \
    const v4f_t* s4 = reinterpret_cast<const v4f_t*>(P + pb0 * dd);            \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                          \
      const int k = tid + u * BLOCK;                                           \
      pre[u] = (k < pn4) ? MFG_STREAM_LOAD(s4 + k) : (v4f_t)(0.0f);            \
    }                                                                          \
    prepi = (tid < pnb * d) ? pi[pb0 * d + tid] : 0.0f;                        \
  }
  if ((int64_t)blockIdx.x < ntiles) MFG_STEP_PREFETCH((int64_t)blockIdx.x)
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * TB;
    const int nb = (int)((B - b0) < TB ? (B - b0) : TB);
    const int n = nb * dd, n4 = n >> 2;
    v4f_t* d4 = reinterpret_cast<v4f_t*>(tP);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int k = tid + u * BLOCK;
      if (k < n4) d4[k] = pre[u];
    }
    for (int k = (n4 << 2) + tid; k < n; k += BLOCK) tP[k] = P[b0 * dd + k];  // ragged last tile only
    if (tid < nb * d) tQ[tid] = prepi;
    __syncthreads();
    if (tile + gridDim.x < ntiles) MFG_STEP_PREFETCH(tile + gridDim.x)
    const int tl = wv * G + t;
    const bool valid = (t < G) && (tl < nb);
    const int tlc = valid ? tl : 0;
    const float* colp = tP + tlc * dd + j;
    const float* qv = tQ + tlc * d;
    double acc = 0.0, s1 = 0.0, s2 = 0.0;
    // rows in groups of col_group_rows(d) (mfg_device.h: three groups of seven at d = 21, the plain row order otherwise)
    const int dr = D ? D : d, grp = col_group_rows(dr);
    for (int g0 = 0; g0 < dr; g0 += grp) {
      double pa = 0.0, p1 = 0.0, p2 = 0.0;
#pragma unroll MFG_STEP_UNROLL
      for (int i = g0; i < g0 + grp && i < (D ? D : d); ++i) {
        const double p = (double)colp[i * d];
        const double qx = (double)qv[i];
        // u = pi_i P_ij is exact in fp64 (two fp32 factors), so acc += u equals the fma, and pi_i P_ij^2 = u p,
        // pi_i^2 P_ij^2 = u^2 are formed inside the fmas with the same single rounding as before: 4 fp64 ops, not 5
        const double u = p * qx;
        pa += u;
        if (KIND != MFG_REWARD_EXTERNAL) {
          p1 = fma(u, p, p1);
          if (KIND == MFG_REWARD_MFG_AC2) p2 = fma(u, u, p2);
        }
      }
      acc = g0 ? acc + pa : pa;
      s1 = g0 ? s1 + p1 : p1;
      s2 = g0 ? s2 + p2 : p2;
    }
    double racc = 0.0;
    if (KIND == MFG_REWARD_MFG_AC2) racc = fma((double)qv[j], s1, -s2);
    if (KIND == MFG_REWARD_SYNTHETIC) racc = -0.5 * s1;
    if (KIND != MFG_REWARD_EXTERNAL) racc = seg_sum(racc, j, d, p2);
    if (valid) {
      pi_next[(b0 + tl) * d + j] = (float)acc;
      if (KIND != MFG_REWARD_EXTERNAL && j == 0) reward[b0 + tl] = (float)racc;
    }
    __syncthreads();
  }
}

#undef MFG_STEP_PREFETCH

// ---------------------------------------------------------------------------------------------
// a3+a4, compile-time small d (21, 15), WAVE-PRIVATE tiles: each wavefront stages the contiguous slab of its own
// G = 64/D trajectories (16-byte loads from the enclosing aligned window, the few bytes of the neighbours that come
// along are never read back), so there is no block-wide barrier at all -- a wave waits only for its OWN prefetch --
// and the per-trajectory reward sum goes through the wave's LDS region (dead after its column walk) instead of a
// shuffle tree.  Same arithmetic and summation order as k_core_small's column pass.
// ---------------------------------------------------------------------------------------------
// Prefetch of tile TT (the G trajectories' slab, as 16-byte words of its enclosing aligned window) into pre[] / prepi.
// MFG_WAVE_PREFETCH: B is a multiple of lcm(G, 4) (the launcher splits off the remainder), so every tile is full and
// the slab is a whole number of 16-byte words: every word of a window lies inside it, the first PER-1 loads are
// unconditional and issue back to back; the last one clamps its lane's word index to the window.  (With per-load lane
// tests, partial register writes or a second code path for partial tiles the compiler serialised the loads with
// s_waitcnt vmcnt(0), or waited for the prefetch right after issuing it in order to copy the whole pre[] tuple through
// a phi -- 2x on the low-occupancy batched kernel.)
#define MFG_WAVE_PREFETCH(TT)                                                             \
  {                                                                                       \
    const int64_t f0 = (TT) * (int64_t)(G * DD);                                          \
    const int pn4 = (int)(((f0 & 3) + (int64_t)(G * DD) + 3) >> 2);                       \
    const v4f_t* src = P4 + (f0 >> 2);                                                    \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                     \
      /* every load unconditional (no phi over pre[]): lanes past the window re-read its last word */ \
      const int k = lane + u * WAVE;                                                      \
      pre[u] = MFG_STREAM_LOAD(src + (((u + 1) * WAVE <= ((G * DD) >> 2)) ? k : (k < pn4 ? k : pn4 - 1))); \
    }                                                                                     \
    prepi = pi[(TT) * (int64_t)(G * D) + (lane < G * D ? lane : G * D - 1)];              \
  }
// MFG_WAVE_PREFETCH_RAG: any B (the tail launch): word addresses clamped to the slab, the 1..3 floats after its last
// whole word are patched into LDS by MFG_WAVE_TAIL.
#define MFG_WAVE_PREFETCH_RAG(TT)                                                             \
  {                                                                                       \
    const int64_t f0 = (TT) * (int64_t)(G * DD);                                          \
    const int64_t a4 = f0 >> 2;                                                           \
    const int png = (int)((B - (TT) * G) < G ? (B - (TT) * G) : G);                       \
    const int pn4 = (int)(((f0 & 3) + (int64_t)png * DD + 3) >> 2);                       \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                     \
      const int k = lane + u * WAVE;                                                      \
      pre[u] = (v4f_t)(0.0f);                                                             \
      /* whole words only, address clamped to the slab: no partial register writes, so the PER loads issue back to  \
         back; the 1..3 floats after the last whole word are patched into LDS by MFG_WAVE_TAIL */                    \
      if (k < pn4) pre[u] = MFG_STREAM_LOAD(P4 + ((a4 + k) < total4 ? (a4 + k) : total4 - 1));                    \
    }                                                                                     \
    prepi = (lane < png * D) ? pi[(TT) * (int64_t)(G * D) + lane] : 0.0f;                 \
  }
// The 1..3 floats after the slab's last whole 16-byte word (B d^2 not a multiple of 4) belong to the LAST tile only: its
// wave copies them into its LDS window after the vector staging (wave-uniform test, off the hot path).
#define MFG_WAVE_TAIL(TT)                                                                  \
  if ((TT) == ntiles - 1 && ((B * DD) & 3)) {                                              \
    const int64_t w0 = total4 << 2;                                                        \
    if (lane < (int)((B * DD) & 3)) wP[(int)(w0 - ((((TT) * (int64_t)(G * DD)) >> 2) << 2)) + lane] = P[w0 + lane]; \
  }
template <int KIND, int D, bool RAG>
__global__ __launch_bounds__(BLOCK, MFG_STEP_WAVES) void k_step_wave(const float* __restrict__ pi, const float* __restrict__ P,
                                                                      int64_t B, float* __restrict__ pi_next,
                                                                      float* __restrict__ reward) {
  constexpr int G = WAVE / D, DD = D * D;
  constexpr int WF = ((G * DD + 6 + 3) / 4) * 4;        // floats of a wave's LDS window (16-byte multiple, + alignment slack)
  constexpr int PER = (WF / 4 + WAVE - 1) / WAVE;       // 16-byte loads per lane per tile
  __shared__ __attribute__((aligned(16))) float sP[WAVES][WF];
  __shared__ float sQ[WAVES][G * D];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);  // wave-uniform by construction: keeps the tile bookkeeping scalar
  const int t = lane / D, j = lane - t * D;
  float* wP = sP[wv];
  float* wQ = sQ[wv];
  const int64_t total4 = (B * DD) >> 2;                 // whole 16-byte words of the slab
  const int64_t ntiles = (B + G - 1) / G;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  v4f_t pre[PER];
  float prepi = 0.0f;
  const v4f_t* P4 = reinterpret_cast<const v4f_t*>(P);
  int64_t tile = (int64_t)blockIdx.x * WAVES + wv;
  if (tile < ntiles) {
    if (RAG) MFG_WAVE_PREFETCH_RAG(tile) else MFG_WAVE_PREFETCH(tile)
  }
  for (; tile < ntiles; tile += nwaves) {
    const int ng = (int)((B - tile * G) < G ? (B - tile * G) : G);
    const int off = (int)((tile * (int64_t)(G * DD)) & 3);   // position of the slab inside its aligned window
    v4f_t* d4 = reinterpret_cast<v4f_t*>(wP);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int k = lane + u * WAVE;
      if (k < WF / 4) d4[k] = pre[u];
    }
    if (lane < G * D) wQ[lane] = prepi;
    if (RAG) MFG_WAVE_TAIL(tile)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (tile + nwaves < ntiles) {
      if (RAG) MFG_WAVE_PREFETCH_RAG(tile + nwaves) else MFG_WAVE_PREFETCH(tile + nwaves)
    }
    const bool valid = (t < G) && (t < ng);
    const int tc = valid ? t : 0;
    const float* colp = wP + off + tc * DD + j;
    const float* qv = wQ + tc * D;
    double acc = 0.0, s1 = 0.0, s2 = 0.0;
    // rows in groups of col_group_rows(D) (mfg_device.h: three groups of seven at d = 21; one group = the plain row order at 15):
    // THE loop of rounds 2-5 with the running sums folded into the totals at the end of a group (all sums start from 0: 0 + u,
    // fma(u, p, 0) and 0 + P0 are exact -- the bits of col_walk_row)
    constexpr int GR = col_group_rows(D);
    double pa = 0.0, p1 = 0.0, p2 = 0.0;
#ifndef MFG_COLWALK_STEP
#define MFG_COLWALK_STEP 2  // developer switch: 2 seven rows per unrolled body (static folds), 3 MFG_STEP_UNROLL rows + a uniform test
#endif
    constexpr int UN = (MFG_COLWALK_STEP == 2 && GR < D) ? (D == 15 ? 8 : GR) : MFG_STEP_UNROLL;  // a whole number of groups per body
#pragma unroll UN
    for (int i = 0; i < D; ++i) {
      const double p = (double)colp[i * D];
      const double qx = (double)qv[i];
      const double u = p * qx;
      pa += u;
      if (KIND != MFG_REWARD_EXTERNAL) {
        p1 = fma(u, p, p1);
        if (KIND == MFG_REWARD_MFG_AC2) p2 = fma(u, u, p2);
      }
      if (GR < D && col_group_end(i, D)) {
        acc += pa;
        s1 += p1;
        s2 += p2;
        pa = p1 = p2 = 0.0;
      }
    }
    if (GR >= D) {
      acc = pa;
      s1 = p1;
      s2 = p2;
    }
    double racc = 0.0;
    if (KIND == MFG_REWARD_MFG_AC2) racc = fma((double)qv[j], s1, -s2);
    if (KIND == MFG_REWARD_SYNTHETIC) racc = s1;
    const int64_t b = tile * G + tc;
    if (valid) pi_next[b * D + j] = (float)acc;
    if (KIND != MFG_REWARD_EXTERNAL) {
      // the wave's tile region is dead now: park the column terms there (8-byte aligned line per trajectory)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      double* line = reinterpret_cast<double*>(wP) + tc * (D + 1);
      if (valid) line[j] = racc;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      if (valid && j == 0) {
        double r0 = 0.0, r1 = 0.0;
        int k = 0;
#pragma unroll 4
        for (; k + 1 < D; k += 2) {
          r0 += line[k];
          r1 += line[k + 1];
        }
        if (k < D) r0 += line[k];
        double r = r0 + r1;
        if (KIND == MFG_REWARD_SYNTHETIC) r *= -0.5;
        reward[b] = (float)r;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// a3+a4, d = 21 / 15, LARGE batches: k_step_wave with the OUTPUT side rebuilt.  tools/micro/step_lab.hip (round 2) showed
// that the walk costs 4 % and the 4.5 % of bytes that are written cost 25 %: the same kernel without its stores streams at
// 6.95 TB/s, with them at 5.5, and with the stores aimed at an L2-resident scratch again at 7.05 -- it is the trickle of
// small writes reaching HBM in between the reads, not the store instructions.  So here a wave
//   * takes KB CONSECUTIVE tiles (KB*G trajectories), parks their pi' / rewards in its own LDS stash and writes them out
//     once per super tile as one contiguous burst of 16-byte stores (KB = 32 at d = 21: 8 KB of pi' per wave),
//   * issues those stores at device scope (sc1: written through the L2 right away instead of trickling out of it line by
//     line whenever the cache replaces one),
//   * and the grid is sized to 8 waves per CU with an equal number of super tiles per wave (the batch costs 42 KB of LDS
//     per block; more resident waves measured slower with the batched stores, fewer starve the read stream).
// Same arithmetic, same summation order, bit-identical outputs.  Measured on the 983 040-transition slab: 5.5 -> 6.4 TB/s.
// A ragged last super tile falls back to per-tile stores.
// ---------------------------------------------------------------------------------------------
// Device-scope (sc1) stores through the raw-buffer intrinsics (a plain C++ store cannot carry a scope; inline asm would
// hide the store from the compiler's hazard / waitcnt bookkeeping).  The base must be wave-uniform.
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x27000);  // raw buffer, 32-bit dword format (gfx9 family)
}
__device__ __forceinline__ void store16_device_scope(__amdgpu_buffer_rsrc_t r, int byte_off, v4f_t v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), r, byte_off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void store4_device_scope(__amdgpu_buffer_rsrc_t r, int byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, byte_off, 0, 16 /* sc1 */);
}
template <int KIND, int D, int KB>
__global__ __launch_bounds__(BLOCK, 2) void k_step_wave_batched(const float* __restrict__ pi, const float* __restrict__ P,
                                                                 int64_t B, float* __restrict__ pi_next,
                                                                 float* __restrict__ reward) {
  constexpr int G = WAVE / D, DD = D * D;
  constexpr int WF = ((G * DD + 6 + 3) / 4) * 4;
  constexpr int PER = (WF / 4 + WAVE - 1) / WAVE;
  constexpr int NO = KB * G * D, NR = KB * G;          // floats of pi' / rewards per super tile
  constexpr bool REW = KIND != MFG_REWARD_EXTERNAL;
  static_assert(NO % 4 == 0 && NR % 4 == 0, "a super tile's outputs must be whole 16-byte words");
  __shared__ __attribute__((aligned(16))) float sP[WAVES][WF];
  __shared__ float sQ[WAVES][G * D];
  __shared__ __attribute__((aligned(16))) float sO[WAVES][NO];
  __shared__ __attribute__((aligned(16))) float sR[WAVES][NR];
  __shared__ double sL[WAVES][G * (D + 1)];  // per-trajectory line of the column terms of the reward
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);  // wave-uniform by construction: keeps the tile bookkeeping scalar
  const int t = lane / D, j = lane - t * D;
  const bool valid = t < G;  // every tile is full (B is a multiple of G)
  const int tc = valid ? t : 0;
  float* wP = sP[wv];
  float* wQ = sQ[wv];
  float* wO = sO[wv];
  float* wR = sR[wv];
  double* line = sL[wv] + tc * (D + 1);
  const int64_t ntiles = (B + G - 1) / G;
  const int64_t nsuper = (ntiles + KB - 1) / KB;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  v4f_t pre[PER];
  float prepi = 0.0f;
  const v4f_t* P4 = reinterpret_cast<const v4f_t*>(P);
  int64_t sup = (int64_t)blockIdx.x * WAVES + wv;
  if (sup * KB < ntiles) MFG_WAVE_PREFETCH(sup * KB)
  for (; sup < nsuper; sup += nwaves) {
    const bool full = (sup + 1) * (int64_t)(KB * G) <= B;  // every trajectory of the super tile exists: batched stores
    int oo = lane, ro = 0;  // running stash offsets of the tile (no per-tile multiply)
    int kk = 0;
#pragma unroll 1
    for (; kk < KB; ++kk, oo += G * D, ro += G) {
      const int64_t tile = sup * KB + kk;
      if (tile >= ntiles) break;  // wave-uniform (last super tile only)
      const int off = (int)((tile * (int64_t)(G * DD)) & 3);
      v4f_t* d4 = reinterpret_cast<v4f_t*>(wP);
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int k = lane + u * WAVE;
        if (k < WF / 4) d4[k] = pre[u];
      }
      if (lane < G * D) wQ[lane] = prepi;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      {
        const int64_t nxt = (kk + 1 < KB && tile + 1 < ntiles) ? tile + 1 : (sup + nwaves) * KB;
        if (nxt < ntiles) MFG_WAVE_PREFETCH(nxt)
      }
      const float* colp = wP + off + tc * DD + j;
      const float* qv = wQ + tc * D;
      double acc = 0.0, s1 = 0.0, s2 = 0.0;
      // The reward of the PREVIOUS tile is summed here, one line entry per row of this tile's walk, by every lane of
      // the trajectory (broadcast reads; the VALU is idle anyway): the per-tile serial phase of k_step_wave -- two wave
      // barriers and 21 dependent adds on one lane -- disappears into the walk.  Same order: even terms, odd terms.
      double r0 = 0.0, r1 = 0.0;
      // rows in groups of col_group_rows(D) (mfg_device.h: three groups of seven at d = 21; the plain row order at 15): the fully
      // unrolled walk of round 2 with the running sums folded at the end of a group (see k_step_wave)
      constexpr int GR = col_group_rows(D);
      double pa = 0.0, p1 = 0.0, p2 = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const double p = (double)colp[i * D];
        const double qx = (double)qv[i];
        const double u = p * qx;
        pa += u;
        if (REW) {
          p1 = fma(u, p, p1);
          if (KIND == MFG_REWARD_MFG_AC2) p2 = fma(u, u, p2);
          if (i & 1) r1 += line[i];
          else r0 += line[i];
        }
        if (GR < D && col_group_end(i, D)) {
          acc += pa;
          s1 += p1;
          s2 += p2;
          pa = p1 = p2 = 0.0;
        }
      }
      if (GR >= D) {
        acc = pa;
        s1 = p1;
        s2 = p2;
      }
      double racc = 0.0;
      if (KIND == MFG_REWARD_MFG_AC2) racc = fma((double)qv[j], s1, -s2);
      if (KIND == MFG_REWARD_SYNTHETIC) racc = s1;
      const int64_t b = tile * G + tc;
      if (full) {
        if (valid) wO[oo] = (float)acc;
      } else if (valid) {
        pi_next[b * D + j] = (float)acc;
      }
      if (REW) {
        if (kk > 0 && valid) {  // the previous tile's reward (stash: every lane of the trajectory writes the same value,
          double r = r0 + r1;   // so the sums above stay in the walk instead of sinking into a one-lane branch)
          if (KIND == MFG_REWARD_SYNTHETIC) r *= -0.5;
          if (full) wR[ro - G + tc] = (float)r;
          else if (j == 0) reward[b - G] = (float)r;
        }
        __builtin_amdgcn_wave_barrier();  // every lane has read the line (LDS operations of a wave execute in order)
        if (valid) line[j] = racc;
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (REW && kk > 0) {
      // the super tile's last reward: the one serial sum left
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      if (valid && j == 0) {
        double r0 = 0.0, r1 = 0.0;
        int k = 0;
#pragma unroll 4
        for (; k + 1 < D; k += 2) {
          r0 += line[k];
          r1 += line[k + 1];
        }
        if (k < D) r0 += line[k];
        double r = r0 + r1;
        if (KIND == MFG_REWARD_SYNTHETIC) r *= -0.5;
        if (full) wR[ro - G + tc] = (float)r;
        else reward[(sup * KB + kk - 1) * G + tc] = (float)r;
      }
    }
    if (full) {
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      const __amdgpu_buffer_rsrc_t ob = out_rsrc(pi_next + sup * (int64_t)NO);
      const v4f_t* s4 = reinterpret_cast<const v4f_t*>(wO);
      for (int k = lane; k < NO / 4; k += WAVE) store16_device_scope(ob, k * 16, s4[k]);
      if (REW) {
        const __amdgpu_buffer_rsrc_t rb = out_rsrc(reward + sup * (int64_t)NR);
        const v4f_t* sr4 = reinterpret_cast<const v4f_t*>(wR);
        for (int k = lane; k < NR / 4; k += WAVE) store16_device_scope(rb, k * 16, sr4[k]);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Fallback for a P pointer that is not 16-byte aligned: scalar staging, no prefetch.
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_step_small_unaligned(const float* __restrict__ pi, const float* __restrict__ P,
                                                                int64_t B, int d, float* __restrict__ pi_next,
                                                                float* __restrict__ reward) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int G = WAVE / d, TB = WAVES * G, dd = d * d;
  double2* tQ = reinterpret_cast<double2*>(smem);
  float* tP = reinterpret_cast<float*>(tQ + TB * d);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int t = lane / d, j = lane - t * d;
  const int p2 = next_pow2(d);
  const int64_t ntiles = (B + TB - 1) / TB;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * TB;
    const int nb = (int)((B - b0) < TB ? (B - b0) : TB);
    for (int k = tid; k < nb * dd; k += BLOCK) tP[k] = P[b0 * dd + k];
    for (int k = tid; k < nb * d; k += BLOCK) {
      const double v = (double)pi[b0 * d + k];
      tQ[k] = make_double2(v, v * v);
    }
    __syncthreads();
    const int tl = wv * G + t;
    const bool valid = (t < G) && (tl < nb);
    const int tlc = valid ? tl : 0;
    const float* colp = tP + tlc * dd + j;
    const double2* qv = tQ + tlc * d;
    double acc = 0.0, s1 = 0.0, s2 = 0.0;
    const int grp = col_group_rows(d);  // (mfg_device.h: pi' of the aligned kernels bit for bit, also at d = 21)
    for (int g0 = 0; g0 < d; g0 += grp) {
      double pa = 0.0, p1 = 0.0, p2 = 0.0;
      for (int i = g0; i < g0 + grp && i < d; ++i) {
        const double p = (double)colp[i * d];
        const double2 q = qv[i];
        const double pp = p * p;
        pa = fma(p, q.x, pa);
        p1 = fma(q.x, pp, p1);
        p2 = fma(q.y, pp, p2);
      }
      acc = g0 ? acc + pa : pa;
      s1 = g0 ? s1 + p1 : p1;
      s2 = g0 ? s2 + p2 : p2;
    }
    double racc = 0.0;
    if (KIND == MFG_REWARD_MFG_AC2) racc = fma(qv[j].x, s1, -s2);
    if (KIND == MFG_REWARD_SYNTHETIC) racc = -0.5 * s1;
    if (KIND != MFG_REWARD_EXTERNAL) racc = seg_sum(racc, j, d, p2);
    if (valid) {
      pi_next[(b0 + tl) * d + j] = (float)acc;
      if (KIND != MFG_REWARD_EXTERNAL && j == 0) reward[b0 + tl] = (float)racc;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// a3+a4, d = 4*LPR (d = 128: LPR = 32, d = 256: LPR = 64): one wavefront per trajectory, every lane issues
// 16-byte loads and one wave instruction covers 64/LPR consecutive rows (1 KiB, fully coalesced).  8-byte
// or half-populated loads reach only ~0.55x of this rate (the d = 128 case of k_step_large).  The lane
// groups that hold the same columns of different rows are combined with xor-shuffles at the end.
// ---------------------------------------------------------------------------------------------
#ifndef MFG_ROWS_UNROLL
#define MFG_ROWS_UNROLL 8   // row groups (1 KiB wave loads) per chunk; two chunks are in flight / in use per wave
#endif
#ifndef MFG_ROWS_BPC
#define MFG_ROWS_BPC 2      // resident blocks per CU the grid is sized for (8 waves per CU)
#endif
#ifndef MFG_ROWS_KT
#define MFG_ROWS_KT 8       // consecutive trajectories per wave whose outputs are written as one burst (large batches)
#endif
#ifndef MFG_ROWS_ABL_NOREW   // timing ablations (tools/variant.sh): drop the reward sums / the output stores
#define MFG_ROWS_ABL_NOREW 0
#endif
#ifndef MFG_ROWS_ABL_NOSTORE
#define MFG_ROWS_ABL_NOSTORE 0
#endif
#ifndef MFG_ROWS_ABL_NOMATH
#define MFG_ROWS_ABL_NOMATH 0
#endif
// Round 2: a CONTINUOUS load stream per wave.  The first version loaded pi, staged it, and only then started the row
// loads of a trajectory -- two serial memory latencies per trajectory during which the wave streamed nothing (12 % of
// the time at d = 128, fitted from the d = 128 / 256 rates) -- and ran 32 waves per CU.  Now the rows are consumed in
// chunks of U row groups from two register buffers (ping-pong): while one chunk is multiplied the next one is in flight,
// and the chunk after a trajectory's last is the FIRST chunk of the wave's next trajectory, issued before the reduction /
// stores of the current one; the next trajectory's state is prefetched into registers at the start of the current one.
// 8 waves per CU (more resident waves measured slower: d = 256 6.5 TB/s at 32 waves/CU, 6.9-7.0 at 8), outputs stored at
// device scope (sc1).  Per-lane summation order is unchanged (rows in increasing order) => bit-identical results.
// KT > 1: a wave takes KT CONSECUTIVE trajectories (a super tile), parks their pi' / rewards in its LDS stash and writes
// them as one contiguous burst (the d = 21 finding: it is the trickle of small writes between the reads that costs
// bandwidth -- the 0.8 % of bytes written here cost 6 % at d = 128).  KT = 1 (small batches): direct stores.
template <int KIND, int LPR, int KT>
__global__ __launch_bounds__(BLOCK, 2) void k_step_rows(const float* __restrict__ pi, const float* __restrict__ P, int64_t B,
                                                        float* __restrict__ pi_next, float* __restrict__ reward) {
  constexpr int d = 4 * LPR, RPW = WAVE / LPR;
  constexpr int U = MFG_ROWS_UNROLL;
  constexpr int CPT = d / RPW / U;   // chunks per trajectory
  constexpr int NQ = d / WAVE;       // state entries per lane
  static_assert(CPT >= 2 && CPT % 2 == 0, "ping-pong needs an even number of chunks per trajectory");
  static_assert(KT == 1 || KT % 4 == 0, "a super tile's rewards must be whole 16-byte words");
  __shared__ double2 qs[WAVES][d];  // (pi_i, pi_i^2) fp64 per wave
  __shared__ __attribute__((aligned(16))) float sO[WAVES][KT > 1 ? KT * d : 4];
  __shared__ __attribute__((aligned(16))) float sR[WAVES][KT > 1 ? KT : 4];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int sub = lane / LPR, c4 = lane - sub * LPR;
  double2* q = qs[wv];
  float* wO = sO[wv];
  float* wR = sR[wv];
  const int64_t nw = (int64_t)gridDim.x * WAVES;
  const int64_t nsup = (B + KT - 1) / KT;
  int64_t sup = (int64_t)blockIdx.x * WAVES + wv;
  if (sup >= nsup) return;
  v4f_t va[U], vb[U];
  float pnx[NQ];
  // chunk c of trajectory bb -> register buffer BUF (U wave-wide 1 KiB loads)
#define MFG_ROWS_ISSUE(BUF, bb, c)                                                                          \
  {                                                                                                         \
    const v4f_t* src = reinterpret_cast<const v4f_t*>(P + (bb) * (int64_t)d * d) + (c) * (U * WAVE) + lane; \
    _Pragma("unroll") for (int u = 0; u < U; ++u) BUF[u] = MFG_STREAM_LOAD(src + u * WAVE);                 \
  }
#define MFG_ROWS_COMPUTE(BUF, c)                                    \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                   \
    if (MFG_ROWS_ABL_NOMATH) {                                      \
      acc[0] += (double)(BUF[u].x + BUF[u].y + BUF[u].z + BUF[u].w); \
      continue;                                                     \
    }                                                               \
    const double2 qq = q[((c) * U + u) * RPW + sub];                \
    const float pv[4] = {BUF[u].x, BUF[u].y, BUF[u].z, BUF[u].w};   \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                 \
      const double p = (double)pv[k];                               \
      acc[k] = fma(p, qq.x, acc[k]);                                \
      if (KIND != MFG_REWARD_EXTERNAL && !MFG_ROWS_ABL_NOREW) {     \
        const double pp = p * p;                                    \
        s1[k] = fma(qq.x, pp, s1[k]);                               \
        if (KIND == MFG_REWARD_MFG_AC2) s2 = fma(qq.y, pp, s2);     \
      }                                                             \
    }                                                               \
  }
  // prologue: state of the first trajectory, its first chunk
  {
    const int64_t b0 = sup * KT;
#pragma unroll
    for (int m = 0; m < NQ; ++m) pnx[m] = pi[b0 * d + lane + m * WAVE];
    MFG_ROWS_ISSUE(va, b0, 0)
  }
  for (; sup < nsup; sup += nw) {
    const bool full = KT > 1 && (sup + 1) * KT <= B;  // every trajectory of the super tile exists: batched stores
#pragma unroll 1
    for (int kk = 0; kk < KT; ++kk) {
      const int64_t b = sup * KT + kk;
      if (b >= B) break;  // wave-uniform (last super tile only)
      // the wave's next trajectory (B = none: the very last chunk issue re-reads this trajectory's chunk 0)
      const int64_t bn = (kk + 1 < KT && b + 1 < B) ? b + 1 : ((sup + nw) * KT < B ? (sup + nw) * KT : B);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int m = 0; m < NQ; ++m) {
        const double v = (double)pnx[m];
        q[lane + m * WAVE] = make_double2(v, v * v);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      if (bn < B) {
#pragma unroll
        for (int m = 0; m < NQ; ++m) pnx[m] = pi[bn * d + lane + m * WAVE];
      }
      double acc[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0}, s2 = 0.0;
#pragma unroll 1
      for (int c = 0; c < CPT; c += 2) {
        MFG_ROWS_ISSUE(vb, b, c + 1)
        MFG_ROWS_COMPUTE(va, c)
        {
          // always issued (no phi over the register buffer): the chunk after the wave's very last one re-reads chunk 0
          const bool more = c + 2 < CPT;
          const int64_t nb = more ? b : (bn < B ? bn : b);
          const int nc = more ? c + 2 : 0;
          MFG_ROWS_ISSUE(va, nb, nc)
        }
        MFG_ROWS_COMPUTE(vb, c + 1)
      }
#pragma unroll
      for (int off = LPR; off < WAVE; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[k] += __shfl_xor(acc[k], off, WAVE);
          if (KIND != MFG_REWARD_EXTERNAL) s1[k] += __shfl_xor(s1[k], off, WAVE);
        }
      }
      if (sub == 0 && (!MFG_ROWS_ABL_NOSTORE || acc[0] == 123.456)) {
        const v4f_t ov = {(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
        if (full) reinterpret_cast<v4f_t*>(wO + kk * d)[c4] = ov;
        else store16_device_scope(out_rsrc(pi_next + b * d), c4 * 16, ov);
      }
      if (KIND != MFG_REWARD_EXTERNAL) {
        double racc = 0.0;
        if (sub == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) racc += (KIND == MFG_REWARD_MFG_AC2) ? q[4 * c4 + k].x * s1[k] : s1[k];
        }
        if (KIND == MFG_REWARD_MFG_AC2) racc -= s2;
        racc = wave_sum(racc);
        if (KIND == MFG_REWARD_SYNTHETIC) racc *= -0.5;
        if (lane == 0 && (!MFG_ROWS_ABL_NOSTORE || racc == 123.456)) {
          if (full) wR[kk] = (float)racc;
          else store4_device_scope(out_rsrc(reward + b), 0, (float)racc);
        }
      }
    }
    if (full) {
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      const __amdgpu_buffer_rsrc_t ob = out_rsrc(pi_next + sup * (int64_t)(KT * d));
      const v4f_t* s4 = reinterpret_cast<const v4f_t*>(wO);
#pragma unroll
      for (int k = lane; k < KT * d / 4; k += WAVE) store16_device_scope(ob, k * 16, s4[k]);
      if (KIND != MFG_REWARD_EXTERNAL && lane < KT / 4)
        store16_device_scope(out_rsrc(reward + sup * (int64_t)KT), lane * 16, reinterpret_cast<const v4f_t*>(wR)[lane]);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
#undef MFG_ROWS_ISSUE
#undef MFG_ROWS_COMPUTE
}

// ---------------------------------------------------------------------------------------------
// a3+a4, large d: one wavefront per trajectory; lane owns VEC consecutive columns in each of R
// 64*VEC-wide column chunks; rows stream from HBM with coalesced VEC*4-byte loads per lane.
// ---------------------------------------------------------------------------------------------
template <int VEC>
struct VecT;
template <>
struct VecT<1> {
  using type = float;
};
template <>
struct VecT<2> {
  using type = float2;
};
template <>
struct VecT<4> {
  using type = float4;
};

template <int VEC, int R, int KIND>
__global__ __launch_bounds__(BLOCK) void k_step_large(const float* __restrict__ pi, const float* __restrict__ P,
                                                      int64_t B, int d, float* __restrict__ pi_next,
                                                      float* __restrict__ reward) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using V = typename VecT<VEC>::type;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  float* pis = smem + wv * d;
  const int64_t nw = (int64_t)gridDim.x * WAVES;
  for (int64_t b = (int64_t)blockIdx.x * WAVES + wv; b < B; b += nw) {
    for (int c = lane; c < d; c += WAVE) pis[c] = pi[b * d + c];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave landed
    double pc[R][VEC], acc[R][VEC], s1[R][VEC];
    int col[R];
#pragma unroll
    for (int m = 0; m < R; ++m) {
      col[m] = (m * WAVE + lane) * VEC;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        pc[m][v] = (col[m] + v < d) ? (double)pis[col[m] + v] : 0.0;
        acc[m][v] = 0.0;
        s1[m][v] = 0.0;
      }
    }
    double s2 = 0.0;  // sum_i pi_i^2 sum_(own columns) p^2
    const float* Pb = P + b * (int64_t)d * d;
#pragma unroll 4
    for (int i = 0; i < d; ++i) {
      const double pii = (double)pis[i];
      const double pii2 = pii * pii;
      const float* row = Pb + (int64_t)i * d;
#pragma unroll
      for (int m = 0; m < R; ++m) {
        if (col[m] < d) {  // VEC divides d on the vector paths, so a chunk is fully in or out
          float pv[VEC];
          *reinterpret_cast<V*>(pv) = *reinterpret_cast<const V*>(row + col[m]);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const double p = (double)pv[v];
            acc[m][v] = fma(p, pii, acc[m][v]);
            if (KIND != MFG_REWARD_EXTERNAL) {
              const double pp = p * p;
              s1[m][v] = fma(pii, pp, s1[m][v]);
              if (KIND == MFG_REWARD_MFG_AC2) s2 = fma(pii2, pp, s2);
            }
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) {
      if (col[m] < d) {
        float pv[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) pv[v] = (float)acc[m][v];
        *reinterpret_cast<V*>(pi_next + b * d + col[m]) = *reinterpret_cast<V*>(pv);
      }
    }
    if (KIND != MFG_REWARD_EXTERNAL) {
      double racc = 0.0;
#pragma unroll
      for (int m = 0; m < R; ++m)
#pragma unroll
        for (int v = 0; v < VEC; ++v) racc += (KIND == MFG_REWARD_MFG_AC2) ? pc[m][v] * s1[m][v] : s1[m][v];
      if (KIND == MFG_REWARD_MFG_AC2) racc -= s2;
      racc = wave_sum(racc);
      if (KIND == MFG_REWARD_SYNTHETIC) racc *= -0.5;
      if (lane == 0) reward[b] = (float)racc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// a6/a8 batch sums: G = [ sum_n delta_n phi(pi_n) | sum delta_n g_n | sum r_n | N ].
// The quadratic block is sum_n delta_n pi_n pi_n^T (upper triangle): each block owns a chunk of
// samples (staged in LDS) x a chunk of 4*BLOCK outputs; partials go to the workspace and are summed
// in a fixed order by k_reduce_partials, so results are run-to-run deterministic.
// ---------------------------------------------------------------------------------------------
constexpr int MFG_GRAD_SMALL_MAX_D = 28;  // k_grad_mfma_small: d + 4 augmented entries fit two 16-wide halves
constexpr int GR_OUT_PER_THREAD = 4;
constexpr int GR_OUT_PER_BLOCK = GR_OUT_PER_THREAD * BLOCK;

struct GradArgs {
  const float* pi;  // sample n=(b,s): pi + b*stride_b + s*d
  int64_t stride_b;
  const double* delta;
  const double* g;
  const float* reward;
  int64_t N;
  int T, d, chunk;  // chunk = samples staged per iteration
  int64_t nsb;      // number of sample-blocks (grid.x)
  double* partial;  // [nsb][F+3]
  // in-kernel finalisation by the last block to finish (k_grad_small with few rows): G, optional parameter update
  int add_reward;     // delta_n <- delta_n + reward_n first (external reward arrived after the rollout); written back
  unsigned* counter;  // zero on entry, zero again on exit; NULL -> separate k_reduce_partials launch
  double* G;
  int accumulate, apply;
  double lr_c, lr_a;
  double *w, *theta, *reward_acc;
};

__global__ __launch_bounds__(BLOCK) void k_grad_partial(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int d = a.d, Q = d * (d + 1) / 2, F = Q + d + 1, FO = F + 3;
  double* dl = reinterpret_cast<double*>(smem_raw);  // [chunk][3] delta, delta*g, reward
  float* sp = reinterpret_cast<float*>(dl + 3 * a.chunk);  // [chunk][d]
  const int tid = threadIdx.x;
  int oi[GR_OUT_PER_THREAD], oj[GR_OUT_PER_THREAD], kind[GR_OUT_PER_THREAD];
  double acc[GR_OUT_PER_THREAD];
#pragma unroll
  for (int u = 0; u < GR_OUT_PER_THREAD; ++u) {
    const int k = blockIdx.y * GR_OUT_PER_BLOCK + u * BLOCK + tid;
    acc[u] = 0.0;
    oi[u] = oj[u] = 0;
    if (k < Q) {
      // invert k = i*d - i(i-1)/2 + (j-i): largest i with start(i) <= k
      int i = (int)(((2.0 * d + 1.0) - sqrt((2.0 * d + 1.0) * (2.0 * d + 1.0) - 8.0 * (double)k)) * 0.5);
      while (i > 0 && feat_idx(i, i, d) > k) --i;
      while (i + 1 < d && feat_idx(i + 1, i + 1, d) <= k) ++i;
      oi[u] = i;
      oj[u] = i + (k - feat_idx(i, i, d));
      kind[u] = 0;
    } else if (k < Q + d) {
      oi[u] = k - Q;
      kind[u] = 1;
    } else if (k < FO) {
      kind[u] = 2 + (k - (Q + d));  // 2 bias, 3 delta*g, 4 reward, 5 count
    } else {
      kind[u] = -1;
    }
  }
  for (int64_t n0 = (int64_t)blockIdx.x * a.chunk; n0 < a.N; n0 += a.nsb * a.chunk) {
    const int cn = (int)((a.N - n0) < a.chunk ? (a.N - n0) : a.chunk);
    __syncthreads();
    for (int k = tid; k < cn * d; k += BLOCK) {
      const int q = k / d, c = k - q * d;
      const int64_t n = n0 + q;
      const int64_t b = n / a.T;
      const int s = (int)(n - b * a.T);
      sp[k] = a.pi[b * a.stride_b + (int64_t)s * d + c];
    }
    for (int q = tid; q < cn; q += BLOCK) {
      const double de = a.delta[n0 + q];
      dl[3 * q] = de;
      dl[3 * q + 1] = a.g ? de * a.g[n0 + q] : 0.0;
      dl[3 * q + 2] = a.reward ? (double)a.reward[n0 + q] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GR_OUT_PER_THREAD; ++u) {
      double s_ = acc[u];
      if (kind[u] == 0) {
        for (int q = 0; q < cn; ++q) s_ = fma(dl[3 * q] * (double)sp[q * d + oi[u]], (double)sp[q * d + oj[u]], s_);
      } else if (kind[u] == 1) {
        for (int q = 0; q < cn; ++q) s_ = fma(dl[3 * q], (double)sp[q * d + oi[u]], s_);
      } else if (kind[u] == 2) {
        for (int q = 0; q < cn; ++q) s_ += dl[3 * q];
      } else if (kind[u] == 3) {
        for (int q = 0; q < cn; ++q) s_ += dl[3 * q + 1];
      } else if (kind[u] == 4) {
        for (int q = 0; q < cn; ++q) s_ += dl[3 * q + 2];
      } else if (kind[u] == 5) {
        s_ += (double)cn;
      }
      acc[u] = s_;
    }
  }
#pragma unroll
  for (int u = 0; u < GR_OUT_PER_THREAD; ++u) {
    const int k = blockIdx.y * GR_OUT_PER_BLOCK + u * BLOCK + tid;
    if (kind[u] >= 0) a.partial[(int64_t)blockIdx.x * FO + k] = acc[u];
  }
}

__global__ void k_add_reward(double* __restrict__ delta, const float* __restrict__ reward, int64_t N) {
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
    delta[n] += (double)reward[n];
}

// Small d (d <= 28; compile-time for the reference's 21 and 15): ALL the batch sums of an update on the fp64 matrix cores.
// Per sample n two augmented vectors of length d + 4 <= 32,
//     a_n = [ delta pi_0 .. delta pi_{d-1} | delta | 1 | 0 | 0 ]        (A operand, "row" index i)
//     b_n = [ pi_0 .. pi_{d-1}             | 1 | g | r | 1 ]            (B operand, "column" index j)
// and C = sum_n a_n b_n^T holds every entry of G = [sum delta phi | sum delta g | sum r | N] in its upper triangle:
//     C[i][j], i <= j < d   = sum delta pi_i pi_j     (quadratic features)      C[i][d]     = sum delta pi_i   (linear)
//     C[d][d]   = sum delta (bias)     C[d][d+1] = sum delta g     C[d+1][d+2] = sum r     C[d+1][d+3] = N.
// v_mfma_f64_16x16x4_f64: a wave keeps the three 16 x 16 tiles (0,0), (0,1), (1,1) of the 32 x 32 product (12 fp64
// accumulators per lane) and retires FOUR samples per K step with three matrix instructions; lane (li = lane & 15,
// lk = lane >> 4) feeds entries li and 16 + li of sample 4 ks + lk, straight from global memory (pi_traj / delta / g /
// reward as the rollout left them; the next batch of K steps is loaded while this one is multiplied).  The round-2 kernel
// kept row i of the sum in d fp64 registers per lane and fetched pi_n[j] from LDS for every FMA: one LDS read per FMA,
// 109 us for the 983 040 samples of the bench rollout against ~20 us of matrix-core time here.
// Partial rows are combined in a fixed order (k_reduce_partials, or in-kernel by the last block for few rows): run-to-run
// deterministic, no floating-point atomics.
// Data path (second version): a wave works through chunks of 64 consecutive samples.  Their pi rows are fetched with d
// fully used load instructions (flat element e = 64 k + lane of the chunk -> sample e / d, entry e % d: consecutive lanes
// read consecutive floats except at trajectory boundaries), delta / g / reward with one load each (lane = sample), all
// into registers while the previous chunk is multiplied, then parked in the wave's own LDS region (compact rows, no
// block barrier) from where the 16 K steps of the chunk read their operands in the matrix layout.  The first version
// loaded the operands directly (5 load instructions per K step, delta / g / reward fetched by 16 lanes each): the
// kernel was bound by the vector-memory issue rate of the CU, 58 us whatever the occupancy.
constexpr int GS_CH = 64;  // samples per chunk = 16 K steps
#ifndef MFG_GS_BPC
#define MFG_GS_BPC 2  // blocks per CU of the launch (2 waves per SIMD: measured)
#endif

template <int D>
struct GradChunk {
  static constexpr int NL = D ? D : MFG_GRAD_SMALL_MAX_D;  // pi loads per lane and chunk
  float pi[NL];
  double de, dg;
  float rr;
};

// n0 = first sample of the chunk (wave uniform); (b0, s0) = its trajectory / step.  Sample n0 + j sits at trajectory
// b0 + (s0 + j) / T, step (s0 + j) % T: small integers, so the division is an fp32 multiply (exact below 2^22).
template <int D, bool WIDE>
__device__ __forceinline__ void grad_chunk_load(GradChunk<D>& c, const GradArgs& a, const double* gp, const float* rp, int d,
                                                int64_t n0, int64_t b0, int s0, int lane, float invT, float inv_d) {
  const int last = (int)((a.N - 1 - n0) < (GS_CH - 1) ? (a.N - 1 - n0) : (GS_CH - 1));  // last live sample of the chunk
  {
    const int64_t n = n0 + (lane < last ? lane : last);  // lane = sample for the per-sample scalars (clamped: masked at use)
    c.de = a.delta[n];
    c.dg = gp[n];
    c.rr = rp[n];
  }
  // The chunk's [64][d] block is contiguous in memory except for the rows the layout skips between trajectories
  // (stride_b - T d floats, the T+1-th state of pi_traj): element e of the block sits at  base + e + q extra,  q = number of
  // trajectory boundaries in front of its sample -- 32-bit arithmetic on a wave-uniform 64-bit base (the (b, s) form cost
  // two 64-bit multiplies and three 64-bit shifts-and-adds per load: 330 of the kernel's 790 VALU instructions per chunk,
  // and f64 VALU work does not overlap the f64 matrix instructions).  Wide strides (WIDE, chosen by the host: skipped part
  // >= 2^23 floats) keep the general form.
  if constexpr (!WIDE) {
    const float* cb = a.pi + b0 * a.stride_b + (int64_t)s0 * d;
    const int extra = (int)(a.stride_b - (int64_t)a.T * d);
#pragma unroll
    for (int k = 0; k < GradChunk<D>::NL; ++k) {
      if (!D && k * WAVE >= GS_CH * d) {                  // run-time d: loads past the chunk are not needed
        c.pi[k] = 0.0f;
        continue;
      }
      int e = k * WAVE + lane;                            // flat element of the chunk's [64][d] block
      int j = (int)(((float)e + 0.5f) * inv_d);           // sample of the chunk (e < 64 * 28: exact in fp32)
      if (j > last) { j = last; e = last * d; }           // past the end of the batch / of a run-time-d chunk: any live entry
      const int q = (int)(((float)(s0 + j) + 0.5f) * invT);
      c.pi[k] = cb[(unsigned)(e + __mul24(q, extra))];
    }
  } else {
#pragma unroll
    for (int k = 0; k < GradChunk<D>::NL; ++k) {
      if (!D && k * WAVE >= GS_CH * d) {
        c.pi[k] = 0.0f;
        continue;
      }
      const int e = k * WAVE + lane;
      int j = (int)(((float)e + 0.5f) * inv_d);
      int col = e - j * d;
      if (j > last) { j = last; col = 0; }
      const int sj = s0 + j;
      const int q = (int)(((float)sj + 0.5f) * invT);
      const int64_t off = (b0 + q) * a.stride_b + (int64_t)((sj - q * a.T) * d + col);
      c.pi[k] = a.pi[off];
    }
  }
}

template <int D, bool WIDE = false>
__global__ __launch_bounds__(BLOCK) void k_grad_mfma_small(GradArgs a) {
  const int d = D ? D : a.d;
  const int Q = d * (d + 1) / 2, F = Q + d + 1, FO = F + 3;
  constexpr int DMAX = D ? D : MFG_GRAD_SMALL_MAX_D;
  // per wave: pi rows [64][d] (+ 16 floats: the hi-half read of the last row may run past it), then per sample
  // (delta, g, reward as double) -- 64 x 3 doubles; the block reduction reuses the space
  constexpr int PI_FL = GS_CH * DMAX + 16;
  constexpr int W_BYTES = ((PI_FL * 4 + 15) / 16) * 16 + GS_CH * 3 * 8;
  constexpr int RED_BYTES = WAVES * 3 * 4 * WAVE * 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[(WAVES * W_BYTES > RED_BYTES) ? WAVES * W_BYTES : RED_BYTES];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = __builtin_amdgcn_readfirstlane(tid / WAVE);
  float* lpi = reinterpret_cast<float*>(smem + (size_t)wv * W_BYTES);
  double* lsc = reinterpret_cast<double*>(smem + (size_t)wv * W_BYTES + ((PI_FL * 4 + 15) / 16) * 16);
  const int li = lane & 15, lk = lane >> 4;
  // lane constants of the augmented entries li (lo half) and 16 + li (hi half)
  auto consts = [&](int idx, float& a_d, double& a_1, float& b_1, double& b_g, double& b_r) {
    a_d = idx == d ? 1.0f : 0.0f;       // a: delta at idx == d
    a_1 = idx == d + 1 ? 1.0 : 0.0;     // a: 1 at idx == d + 1
    b_1 = (idx == d || idx == d + 3) ? 1.0f : 0.0f;
    b_g = idx == d + 1 ? 1.0 : 0.0;
    b_r = idx == d + 2 ? 1.0 : 0.0;
  };
  float ad_lo, b1_lo, ad_hi, b1_hi;
  double a1_lo, a1_hi, bg_lo, br_lo, bg_hi, br_hi;
  consts(li, ad_lo, a1_lo, b1_lo, bg_lo, br_lo);
  consts(16 + li, ad_hi, a1_hi, b1_hi, bg_hi, br_hi);
  const bool pi_lo = li < d, pi_hi = 16 + li < d;
  const int ilo = pi_lo ? li : 0, ihi = pi_hi ? 16 + li : 0;
  // optional inputs: a valid address to load from, and whether the loaded value counts
  const bool has_g = a.g != nullptr, has_r = a.reward != nullptr;
  const double* gp = has_g ? a.g : a.delta;
  const float* rp = has_r ? a.reward : a.pi;
  v4d_t c00 = (v4d_t)(0.0), c01 = (v4d_t)(0.0), c11 = (v4d_t)(0.0);
  const float invT = 1.0f / (float)a.T, inv_d = 1.0f / (float)d;
  const int64_t NC = (a.N + GS_CH - 1) / GS_CH;           // chunks
  const int64_t W = (int64_t)gridDim.x * WAVES;           // waves of the launch
  const int64_t gw = (int64_t)blockIdx.x * WAVES + wv;
  GradChunk<D> nx;
  if (gw < NC) grad_chunk_load<D, WIDE>(nx, a, gp, rp, d, gw * GS_CH, (gw * GS_CH) / a.T, (int)((gw * GS_CH) % a.T), lane, invT, inv_d);
  for (int64_t ch = gw; ch < NC; ch += W) {
    const int64_t n0 = ch * GS_CH;
    // park the chunk in LDS (the previous chunk's reads are complete: wave-local barrier)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < GradChunk<D>::NL; ++k)
      if (D || k * WAVE < GS_CH * d) lpi[k * WAVE + lane] = nx.pi[k];
    {
      const bool ok = n0 + lane < a.N;
      const double rr = (ok && has_r) ? (double)nx.rr : 0.0;
      double de = ok ? nx.de : 0.0;
      if (a.add_reward) {
        de += rr;
        if (ok) const_cast<double*>(a.delta)[n0 + lane] = de;  // the lane that owns sample n writes it back
      }
      lsc[3 * lane] = de;
      lsc[3 * lane + 1] = (ok && has_g) ? nx.dg : 0.0;
      lsc[3 * lane + 2] = rr;
    }
    const int nvalid = (int)((a.N - n0) < GS_CH ? (a.N - n0) : GS_CH);  // samples of this chunk (wave uniform)
    if (ch + W < NC) {
      const int64_t n1 = (ch + W) * GS_CH;
      grad_chunk_load<D, WIDE>(nx, a, gp, rp, d, n1, n1 / a.T, (int)(n1 % a.T), lane, invT, inv_d);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int ks = 0; ks < GS_CH / 4; ++ks) {
      const int j = 4 * ks + lk;
      const float plo = lpi[j * d + ilo], phi = lpi[j * d + ihi];
      const double de = lsc[3 * j], dg = lsc[3 * j + 1];
      const double rr = lsc[3 * j + 2];
      const double one = j < nvalid ? 1.0 : 0.0;               // slots past the last sample hold zeros and count nothing
      double A_lo, B_lo;
      if constexpr (D >= 16) {  // entries 0 .. 15 are all state entries: no constants, no selects
        B_lo = (double)plo;
        A_lo = de * B_lo;
      } else {
        A_lo = fma(de, (double)(pi_lo ? plo : ad_lo), a1_lo * one);
        B_lo = fma(bg_lo, dg, fma(br_lo, rr, (double)(pi_lo ? plo : b1_lo)));
      }
      const double A_hi = fma(de, (double)(pi_hi ? phi : ad_hi), a1_hi * one);
      const double B_hi = fma(bg_hi, dg, fma(br_hi, rr, (double)(pi_hi ? phi : b1_hi)));
      // (timing ablations at the bench shape, 44 us: without these three instructions 24 us, without the chunk loads 38 us --
      //  the matrix-core time, 20 us, ADDS to the rest whatever the occupancy (1 / 2 / 4 blocks per CU: 49 / 44 / 44 us):
      //  the fp64 matrix instructions of this kernel do not hide behind its other work)
      c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(A_lo, B_lo, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(A_lo, B_hi, c01, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(A_hi, B_hi, c11, 0, 0, 0);
    }
  }
  __syncthreads();  // every wave is done with its staging region: the block reduction reuses the space
  double (*red)[3][4][WAVE] = reinterpret_cast<double (*)[3][4][WAVE]>(smem);
  // block reduction in a fixed order: every wave parks its tiles, then each output is added up over the WAVES copies
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    red[wv][0][v][lane] = c00[v];
    red[wv][1][v][lane] = c01[v];
    red[wv][2][v][lane] = c11[v];
  }
  __syncthreads();
  double* out = a.partial + (int64_t)blockIdx.x * FO;
  for (int e = tid; e < 3 * 4 * WAVE; e += BLOCK) {
    const int t = e / (4 * WAVE), v = (e / WAVE) & 3, l = e & (WAVE - 1);
    // D[i][j] of a tile: lane l holds row (l >> 4) + 4 v, column l & 15 (f64 MFMA layout)
    const int gi = (t == 2 ? 16 : 0) + 4 * v + (l >> 4), gj = (t == 0 ? 0 : 16) + (l & 15);
    int k = -1;
    if (gj < d) {
      if (gi <= gj) k = feat_idx(gi, gj, d);
    } else if (gj == d) {
      if (gi <= d) k = Q + gi;            // linear terms, then the bias at gi == d
    } else if (gj == d + 1) {
      if (gi == d) k = F;                 // sum delta g
    } else if (gj == d + 2) {
      if (gi == d + 1) k = F + 1;         // sum r
    } else if (gj == d + 3) {
      if (gi == d + 1) k = F + 2;         // N
    }
    if (k >= 0) {
      double tsum = red[0][t][v][l];
#pragma unroll
      for (int q = 1; q < WAVES; ++q) tsum += red[q][t][v][l];
      out[k] = tsum;
    }
  }
  if (!a.counter) return;
  // Few rows (small batches, per-step updates): the last block to finish sums the rows in a fixed order, writes G
  // and, when asked, applies the parameter update -- one launch instead of three dependent ones.
  __shared__ int s_last;
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = (atomicAdd(a.counter, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  double* fin = &red[0][0][0][0];  // FO <= 32 * 33 / 2 + ... < 3 * 4 * 64 * WAVES doubles
  const int nrows = (int)gridDim.x;
  for (int k = tid; k < FO; k += BLOCK) {
    // plain loads: the agent-scope fence above already invalidated this CU's L1, and nothing here was read before it
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const double* col = a.partial + k;
    int r = 0;
    for (; r + 7 < nrows; r += 8) {
      const double v0 = col[(int64_t)r * FO], v1 = col[(int64_t)(r + 1) * FO], v2 = col[(int64_t)(r + 2) * FO],
                   v3 = col[(int64_t)(r + 3) * FO], v4 = col[(int64_t)(r + 4) * FO], v5 = col[(int64_t)(r + 5) * FO],
                   v6 = col[(int64_t)(r + 6) * FO], v7 = col[(int64_t)(r + 7) * FO];
      s0 += v0;
      s1 += v1;
      s2 += v2;
      s3 += v3;
      s0 += v4;
      s1 += v5;
      s2 += v6;
      s3 += v7;
    }
    for (; r < nrows; ++r) s0 += col[(int64_t)r * FO];
    const double tot = (s0 + s1) + (s2 + s3);
    const double gk = a.accumulate ? a.G[k] + tot : tot;
    a.G[k] = gk;
    fin[k] = gk;  // (`red` as tiles was last read before the barriers around the completion counter)
  }
  __syncthreads();
  if (a.apply) {
    // identical arithmetic to k_apply_update
    const double count = fin[F + 2];
    if (count > 0.0) {
      const double inv = 1.0 / count;
      for (int k = tid; k < F; k += BLOCK) a.w[k] = updated_param(a.w[k], a.lr_c, fin[k], inv);
      if (tid == 0) {
        if (a.reward_acc) *a.reward_acc += fin[F + 1] * inv;
        *a.theta = updated_param(*a.theta, a.lr_a, fin[F], inv);
      }
    }
  }
  if (tid == 0) *a.counter = 0u;
}

// ---------------------------------------------------------------------------------------------
// Critic-gradient sums on the fp64 matrix cores for d a multiple of 16 (d >= 64): the one GEMM-shaped piece of the
// path, M = sum_n delta_n pi_n pi_n^T = A B with A = (delta pi)^T [d x N], B = pi [N x d].  v_mfma_f64_16x16x4_f64:
// a wave owns up to 8 upper-triangle 16x16 tiles of M (4 fp64 accumulators per lane per tile), a block stages 32
// samples (fp32 rows + delta) in LDS and runs 8 K-steps of 4 samples over them; operands are widened / scaled on
// the way from LDS (2 LDS reads + 2 cvt + 1 mul per 2 048-flop MFMA instead of 3 LDS reads per FMA in
// k_grad_partial, which ran at 4.5 % of the fp64 peak: 3.1 ms per C3 rollout).  Split-K over grid.x with one partial
// row per x, tiles split over grid.y; the linear / scalar sums ride on the y = 0 blocks.  Deterministic.
// ---------------------------------------------------------------------------------------------
constexpr int GM_KC = 32;    // samples staged per chunk
#ifndef MFG_GM_TPW
#define MFG_GM_TPW 8
#endif
constexpr int GM_TPW = MFG_GM_TPW;    // max tiles per wave

// NPF > 0 (float4 staging, NPF = d / 32 sixteen-byte loads per thread and chunk): the NEXT chunk's rows and deltas are
// fetched into registers before this chunk's matrix instructions and committed to LDS after them, so the staging
// latency (every sample chunk is staged by all blockIdx.y slices) hides behind the MFMAs.  NPF == 0: unpipelined.
template <int NPF>
__global__ __launch_bounds__(BLOCK) void k_grad_mfma(GradArgs a, int tpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int d = a.d, nt = d >> 4, pitch = d + 16;  // pitch = 16 mod 32: the four k-rows of an operand hit distinct banks
  const int Q = d * (d + 1) / 2, F = Q + d + 1, FO = F + 3;
  double* dl = reinterpret_cast<double*>(smem_raw);                  // [KC] delta
  double* red = dl + GM_KC;                                          // [4][BLOCK] scalar reduction scratch
  float* sp = reinterpret_cast<float*>(red + 4 * BLOCK);             // [KC][pitch] pi rows
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int ntiles = nt * (nt + 1) / 2;
  // this wave's tiles: linear ids t0 .. t0+nmine-1 of the row-major upper-triangle tile list
  const int t0 = ((int)blockIdx.y * WAVES + wv) * tpw;
  int nmine = ntiles - t0;
  nmine = nmine < 0 ? 0 : (nmine > tpw ? tpw : nmine);
  int tr[GM_TPW], tc[GM_TPW];
#pragma unroll
  for (int i = 0; i < GM_TPW; ++i) {
    int t = t0 + i, r = 0;
    if (i < nmine) {
      while (t >= nt - r) {  // row r of the tile triangle holds nt - r tiles
        t -= nt - r;
        ++r;
      }
    } else {
      t = 0;
    }
    tr[i] = r;
    tc[i] = r + t;
  }
  v4d_t acc[GM_TPW];
#pragma unroll
  for (int i = 0; i < GM_TPW; ++i) acc[i] = (v4d_t)(0.0);
  const bool side = blockIdx.y == 0;  // also owns the linear and scalar sums
  double lin[4] = {0.0, 0.0, 0.0, 0.0}, s_d = 0.0, s_dg = 0.0, s_r = 0.0, s_n = 0.0;
  const int li = lane & 15, lk = lane >> 4;
  const double invT = 1.0 / (double)a.T;
  // register prefetch of a chunk (NPF > 0): thread -> (row q0 + rpp u, float4 column c4), no divisions per element
  const int dq = d >> 2, rpp = BLOCK / (dq > 0 ? dq : 1);
  const int q0 = tid / dq, c4 = tid - q0 * dq;
  float4 pf[NPF > 0 ? NPF : 1];
  double pf_de = 0.0, pf_dg = 0.0, pf_rr = 0.0;
  bool pf_on = false;
#define MFG_GM_FETCH(n0_)                                                                              \
  {                                                                                                    \
    const int cn_ = (int)((a.N - (n0_)) < GM_KC ? (a.N - (n0_)) : GM_KC);                              \
    _Pragma("unroll") for (int u = 0; u < NPF; ++u) {                                                  \
      const int q = q0 + u * rpp;                                                                      \
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);                                                         \
      if (q < cn_) {                                                                                   \
        const int64_t n = (n0_) + q;                                                                   \
        const int64_t b = (int64_t)(((double)n + 0.5) * invT);                                         \
        pf[u] = *reinterpret_cast<const float4*>(a.pi + b * a.stride_b + (n - b * a.T) * d + 4 * c4);  \
      }                                                                                                \
    }                                                                                                  \
    pf_de = 0.0; pf_dg = 0.0; pf_rr = 0.0; pf_on = false;                                              \
    if (tid < cn_) {                                                                                   \
      const int64_t n = (n0_) + tid;                                                                   \
      pf_de = a.delta[n];                                                                              \
      pf_rr = a.reward ? (double)a.reward[n] : 0.0;                                                    \
      if (a.g) pf_dg = a.g[n];                                                                         \
      pf_on = true;                                                                                    \
    }                                                                                                  \
  }
  if (NPF > 0 && (int64_t)blockIdx.x * GM_KC < a.N) MFG_GM_FETCH((int64_t)blockIdx.x * GM_KC)
  for (int64_t n0 = (int64_t)blockIdx.x * GM_KC; n0 < a.N; n0 += (int64_t)gridDim.x * GM_KC) {
    const int cn = (int)((a.N - n0) < GM_KC ? (a.N - n0) : GM_KC);
    __syncthreads();
    if (NPF > 0) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) *reinterpret_cast<float4*>(sp + (q0 + u * rpp) * pitch + 4 * c4) = pf[u];
      if (tid < GM_KC) {
        if (side && pf_on) {
          s_d += pf_de;
          if (a.g) s_dg = fma(pf_de, pf_dg, s_dg);
          s_r += pf_rr;
          s_n += 1.0;
        }
        dl[tid] = pf_de;
      }
      __syncthreads();
      const int64_t nn = n0 + (int64_t)gridDim.x * GM_KC;
      if (nn < a.N) MFG_GM_FETCH(nn)
    } else {
    if (a.chunk) {
      // rows are 16-byte aligned and BLOCK is a multiple of d/4: thread -> (row, float4 column) without divisions
      for (int q = q0; q < GM_KC; q += rpp) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < cn) {
          const int64_t n = n0 + q;
          const int64_t b = (int64_t)(((double)n + 0.5) * invT);
          v = *reinterpret_cast<const float4*>(a.pi + b * a.stride_b + (n - b * a.T) * d + 4 * c4);
        }
        *reinterpret_cast<float4*>(sp + q * pitch + 4 * c4) = v;  // rows past the end are zero: they add nothing
      }
    } else {
      for (int k = tid; k < GM_KC * d; k += BLOCK) {
        const int q = k / d, c = k - q * d;
        float v = 0.0f;
        if (q < cn) {
          const int64_t n = n0 + q;
          const int64_t b = (int64_t)(((double)n + 0.5) * invT);
          v = a.pi[b * a.stride_b + (n - b * a.T) * d + c];
        }
        sp[q * pitch + c] = v;
      }
    }
    if (tid < GM_KC) {
      double de = 0.0;
      if (tid < cn) {
        const int64_t n = n0 + tid;
        de = a.delta[n];
        const double rr = a.reward ? (double)a.reward[n] : 0.0;
        if (side) {
          s_d += de;
          if (a.g) s_dg = fma(de, a.g[n], s_dg);
          s_r += rr;
          s_n += 1.0;
        }
      }
      dl[tid] = de;
    }
    __syncthreads();
    }
#pragma unroll
    for (int ks = 0; ks < GM_KC / 4; ++ks) {
      const int k = ks * 4 + lk;
      const double dk = dl[k];
      const float* row = sp + k * pitch + li;
#pragma unroll
      for (int i = 0; i < GM_TPW; ++i) {
        if (i < nmine) {
          const double av = dk * (double)row[tr[i] << 4];   // A[i = li][k] = delta_k pi_k[16 r + li]
          const double bv = (double)row[tc[i] << 4];        // B[k][j = li] = pi_k[16 c + li]
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
        }
      }
    }
    if (side) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = tid + u * BLOCK;
        if (c < d) {
          double t = lin[u];
          for (int q = 0; q < GM_KC; ++q) t = fma(dl[q], (double)sp[q * pitch + c], t);
          lin[u] = t;
        }
      }
    }
  }
#undef MFG_GM_FETCH
  // D[i][j] of a tile: lane holds rows i = 4 v + lane / 16, v = 0..3, column j = lane % 16
  double* out = a.partial + (int64_t)blockIdx.x * FO;
#pragma unroll
  for (int i = 0; i < GM_TPW; ++i) {
    if (i < nmine) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int gi = (tr[i] << 4) + 4 * v + lk, gj = (tc[i] << 4) + li;
        if (gi <= gj) out[feat_idx(gi, gj, d)] = acc[i][v];
      }
    }
  }
  if (side) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = tid + u * BLOCK;
      if (c < d) out[Q + c] = lin[u];
    }
    __syncthreads();
    red[tid] = s_d;
    red[BLOCK + tid] = s_dg;
    red[2 * BLOCK + tid] = s_r;
    red[3 * BLOCK + tid] = s_n;
    __syncthreads();
    if (tid < 4) {
      double t = 0.0;
      for (int q = 0; q < GM_KC; ++q) t += red[tid * BLOCK + q];  // only threads < KC hold scalar partials
      out[Q + d + tid] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Round 3: register-blocked form of k_grad_mfma for d = 128 / 256 (C3, C5).  k_grad_mfma gives a wave an arbitrary run of
// tiles, fetches BOTH operands of every matrix instruction from LDS and tests `i < nmine` (a scalar branch, i.e. a basic
// block boundary) in front of each: nothing overlaps, 190-220 cycles per instruction against the ~70 the matrix core needs,
// whatever the tile count per wave or the number of sample slices (variants measured: no better).  Here a wave owns
// RECTANGLES of tiles given at compile time as (rows NA, columns NB, tile mask) -- one K step of 4 samples fetches NA + NB
// operands for up to NA NB instructions, no branch in the K loop, and the fetches of a step issue under the matrix
// instructions of the step before.  The upper triangle is cut so that EVERY wave has the same number of tiles (the
// instruction stream of a SIMD is the bound; 64 x 64 "super tiles" s = 0 .. d/64-1 on the diagonal, (r, c) off it):
//   d = 256, 136 tiles = 8 waves x 17 (grid.y = 2 blocks of 4 waves):
//     y = 0: (0,1) + one tile of diagonal 1 | (0,2) + one | (0,3) + one | diagonal 0 (10 tiles) + the first two rows of diagonal 1 (7)
//     y = 1: (1,2) + one tile of diagonal 3 | (1,3) + one | (2,3) + one | diagonal 2 + the first two rows of diagonal 3
//   d = 128, 36 tiles = 4 waves x 9: half of (0,1) (2 x 4 tiles) + tile (3,3) of a diagonal | ... | diagonal 0 less (3,3) | diagonal 1 less (3,3)
// Staging, split-K over grid.x, partial rows, side sums and the row reduction are those of k_grad_mfma.
// ---------------------------------------------------------------------------------------------
constexpr unsigned GM2_RECT44 = 0xFFFFu, GM2_DIAG44 = 0x8CEFu, GM2_DIAG44M = 0x0CEFu, GM2_TRAP24 = 0xEFu, GM2_RECT24 = 0xFFu,
                   GM2_ONE = 1u;
__host__ __device__ constexpr int gm2_count(unsigned m) { return m == 0 ? 0 : (int)(m & 1u) + gm2_count(m >> 1); }

// one rectangle over one staged chunk
template <int D, int NA, int NB, unsigned MASK>
__device__ __forceinline__ void gm2_step(const float* __restrict__ abase, const float* __restrict__ bbase, int off, double dk,
                                         v4d_t* acc) {
  double av[NA], bv[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) av[i] = dk * (double)abase[off + (i << 4)];  // A[row li][k] = delta_k pi_k[16 (ra + i) + li]
#pragma unroll
  for (int j = 0; j < NB; ++j) bv[j] = (double)bbase[off + (j << 4)];       // B[k][col li] = pi_k[16 (cb + j) + li]
  int t = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if ((MASK >> (i * NB + j)) & 1u) {
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[t], 0, 0, 0);
        ++t;
      }
    }
  }
}
template <int D, int NA0, int NB0, unsigned M0, int NA1, int NB1, unsigned M1>
__device__ __forceinline__ void gm2_chunk(const float* __restrict__ sp, const double* __restrict__ dl, int lk, int li,
                                          const int* ra, const int* cb, v4d_t* acc) {
  constexpr int pitch = D + 16;
  const float* lb = sp + lk * pitch + li;
  const float *a0 = lb + (ra[0] << 4), *b0 = lb + (cb[0] << 4), *a1 = lb + (ra[1] << 4), *b1 = lb + (cb[1] << 4);
#pragma unroll 2
  for (int ks = 0; ks < GM_KC / 4; ++ks) {  // (fully unrolled the compiler hoists all 8 steps' operands: 444 spilled registers)
    const double dk = dl[ks * 4 + lk];
    gm2_step<D, NA0, NB0, M0>(a0, b0, ks * 4 * pitch, dk, acc);
    if constexpr (M1 != 0) gm2_step<D, NA1, NB1, M1>(a1, b1, ks * 4 * pitch, dk, acc + gm2_count(M0));
  }
}
template <int D, int NA, int NB, unsigned MASK>
__device__ __forceinline__ void gm2_store(double* __restrict__ out, int lk, int li, int ra, int cb, const v4d_t* acc) {
  int t = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if ((MASK >> (i * NB + j)) & 1u) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {  // D[i][j] of a tile: lane holds rows 4 v + lane / 16, column lane % 16
          const int gi = ((ra + i) << 4) + 4 * v + lk, gj = ((cb + j) << 4) + li;
          if (gi <= gj) out[feat_idx(gi, gj, D)] = acc[t][v];
        }
        ++t;
      }
    }
  }
}

template <int D>
struct Gm2Geom {
  static constexpr int NY = D == 256 ? 2 : 1, ACCN = D == 256 ? 17 : 9;
};
template <int NA_, int NB_, unsigned MASK_>
struct Gm2Rect {
  static constexpr int NA = NA_, NB = NB_;
  static constexpr unsigned MASK = MASK_;
};
template <int D>
__global__ __launch_bounds__(BLOCK, 2) void k_grad_mfma2(GradArgs a) {
  static_assert(D == 128 || D == 256, "tile maps exist for d = 128 and 256");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int d = D, pitch = D + 16;
  constexpr int Q = d * (d + 1) / 2, F = Q + d + 1, FO = F + 3;
  constexpr int ACCN = Gm2Geom<D>::ACCN;
  double* dl = reinterpret_cast<double*>(smem_raw);       // [KC] delta
  double* red = dl + GM_KC;                               // [4][BLOCK] scalar reduction scratch
  float* sp = reinterpret_cast<float*>(red + 4 * BLOCK);  // [KC][pitch] pi rows
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE), y = (int)blockIdx.y;
  // this wave's two rectangles (tile coordinates of their corners); the last wave of a block owns the diagonal pieces
  const bool diag = D == 256 ? wv == 3 : wv >= 2;
  int ra[2] = {0, 0}, cb[2] = {0, 0};
  if (D == 256) {
    const int sd = 2 * y;  // this block's diagonal super tiles: sd (whole), sd + 1 (split)
    if (diag) {
      ra[0] = cb[0] = 4 * sd;
      ra[1] = cb[1] = 4 * sd + 4;
    } else {
      const int sr = y == 0 ? 0 : (wv == 2 ? 2 : 1), sc = y == 0 ? wv + 1 : (wv == 0 ? 2 : 3);
      ra[0] = 4 * sr;
      cb[0] = 4 * sc;
      ra[1] = 4 * sd + 4 + (wv == 2 ? 3 : 2);  // tiles (2,2) (2,3) (3,3) of super tile sd + 1
      cb[1] = 4 * sd + 4 + (wv == 0 ? 2 : 3);
    }
  } else {
    if (diag) ra[0] = cb[0] = 4 * (wv - 2);
    else {
      ra[0] = 2 * wv;
      cb[0] = 4;
      ra[1] = cb[1] = 4 * wv + 3;  // tile (3,3) of diagonal super tile wv
    }
  }
  const bool side = blockIdx.y == 0;  // also owns the linear and scalar sums
  double lin = 0.0, s_d = 0.0, s_dg = 0.0, s_r = 0.0, s_n = 0.0;  // (d <= BLOCK: one linear column per thread)
  const int li = lane & 15, lk = lane >> 4;
  const double invT = 1.0 / (double)a.T;
  // register prefetch of a chunk: thread -> (row q0 + rpp u, float4 column c4)
  constexpr int dq = d >> 2, rpp = BLOCK / dq, NPF = GM_KC / rpp;
  static_assert(BLOCK % dq == 0 && GM_KC % rpp == 0, "whole rows per staging pass");
  const int q0 = tid / dq, c4 = tid - q0 * dq;
  double* out = a.partial + (int64_t)blockIdx.x * FO;
  // The whole sample loop is instantiated once per ROLE and the role is chosen outside it: with the role test inside the
  // loop the two instantiations' accumulators are merged at every iteration (two full sets live: 254 spilled registers
  // at d = 256).  Both instantiations execute the same barriers.
  auto run = [&](auto r0, auto r1) __attribute__((always_inline)) {
    using R0 = decltype(r0);
    using R1 = decltype(r1);
    v4d_t acc[ACCN];
#pragma unroll
    for (int i = 0; i < ACCN; ++i) acc[i] = (v4d_t)(0.0);
    float4 pf[NPF];
    double pf_de = 0.0, pf_dg = 0.0, pf_rr = 0.0;
    bool pf_on = false;
    auto fetch = [&](int64_t n0_) __attribute__((always_inline)) {
      const int cn_ = (int)((a.N - n0_) < GM_KC ? (a.N - n0_) : GM_KC);
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        const int q = q0 + u * rpp;
        pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);  // rows past the end are zero: they add nothing
        if (q < cn_) {
          const int64_t n = n0_ + q;
          const int64_t b = (int64_t)(((double)n + 0.5) * invT);
          pf[u] = *reinterpret_cast<const float4*>(a.pi + b * a.stride_b + (n - b * a.T) * d + 4 * c4);
        }
      }
      pf_de = 0.0, pf_dg = 0.0, pf_rr = 0.0, pf_on = false;
      if (tid < cn_) {
        const int64_t n = n0_ + tid;
        pf_de = a.delta[n];
        pf_rr = a.reward ? (double)a.reward[n] : 0.0;
        if (a.g) pf_dg = a.g[n];
        pf_on = true;
      }
    };
    if ((int64_t)blockIdx.x * GM_KC < a.N) fetch((int64_t)blockIdx.x * GM_KC);
    for (int64_t n0 = (int64_t)blockIdx.x * GM_KC; n0 < a.N; n0 += (int64_t)gridDim.x * GM_KC) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NPF; ++u) *reinterpret_cast<float4*>(sp + (q0 + u * rpp) * pitch + 4 * c4) = pf[u];
      if (tid < GM_KC) {
        if (side && pf_on) {
          s_d += pf_de;
          if (a.g) s_dg = fma(pf_de, pf_dg, s_dg);
          s_r += pf_rr;
          s_n += 1.0;
        }
        dl[tid] = pf_de;
      }
      __syncthreads();
      const int64_t nn = n0 + (int64_t)gridDim.x * GM_KC;
      if (nn < a.N) fetch(nn);
      gm2_chunk<D, R0::NA, R0::NB, R0::MASK, R1::NA, R1::NB, R1::MASK>(sp, dl, lk, li, ra, cb, acc);
      if (side && tid < d) {
        double t = lin;
#pragma unroll 8
        for (int q = 0; q < GM_KC; ++q) t = fma(dl[q], (double)sp[q * pitch + tid], t);
        lin = t;
      }
    }
    gm2_store<D, R0::NA, R0::NB, R0::MASK>(out, lk, li, ra[0], cb[0], acc);
    if constexpr (R1::MASK != 0) gm2_store<D, R1::NA, R1::NB, R1::MASK>(out, lk, li, ra[1], cb[1], acc + gm2_count(R0::MASK));
  };
  if (D == 256) {
    if (diag) run(Gm2Rect<4, 4, GM2_DIAG44>{}, Gm2Rect<2, 4, GM2_TRAP24>{});
    else run(Gm2Rect<4, 4, GM2_RECT44>{}, Gm2Rect<1, 1, GM2_ONE>{});
  } else {
    if (diag) run(Gm2Rect<4, 4, GM2_DIAG44M>{}, Gm2Rect<1, 1, 0u>{});
    else run(Gm2Rect<2, 4, GM2_RECT24>{}, Gm2Rect<1, 1, GM2_ONE>{});
  }
  if (side) {
    if (tid < d) out[Q + tid] = lin;
    __syncthreads();
    red[tid] = s_d;
    red[BLOCK + tid] = s_dg;
    red[2 * BLOCK + tid] = s_r;
    red[3 * BLOCK + tid] = s_n;
    __syncthreads();
    if (tid < 4) {
      double t = 0.0;
      for (int q = 0; q < GM_KC; ++q) t += red[tid * BLOCK + q];  // only threads < KC hold scalar partials
      out[Q + d + tid] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Critic values of a whole rollout on the fp64 matrix cores (d a multiple of 16, 64 <= d <= 512): the second GEMM-shaped
// piece of the path.  V_n = x_n^T U x_n + b.x_n + c for the N = B (T+1) states a rollout left in pi_traj, i.e. the
// diagonal of X U X^T: a wave owns 16 states (A operand: X[16 x 4] slices from its LDS copy of the rows), sweeps the
// column tiles of the upper triangle U (B operand straight from the L2-resident weight vector, row k of U is contiguous in
// c), and folds each finished 16 x 16 tile Y = X U into  v_n += sum_c (Y_nc + b_c) x_nc.  The 16 lanes that hold one state
// are one DPP row, so the final sum is four DPP steps.  Inside the wave-per-trajectory rollout kernel the same values
// cost a latency-bound triangular loop per state (10 % of the d = 128 training rollout); here they are ~2 nt (nt + 1)
// matrix instructions per 16 states.  Followed by k_td_delta (delta = r + gamma V' - V).
// ---------------------------------------------------------------------------------------------
// Round 2b: the B operand comes from LDS.  Before, every lane fetched its U element from the L2-resident weight vector
// with 64-bit index arithmetic and a select in front of every matrix instruction (~15 VALU + 1 dependent L2 load per
// MFMA, the whole of U re-read by every wave for its 16 states): 3-6x the matrix-core time.  Now the block's four waves
// (64 states) share each 64-row x 16-column piece of U: it is fetched once per block (zero-filled below the diagonal),
// double-buffered in LDS -- the loads of piece n+1 are issued before the matrix instructions of piece n and committed
// after them -- and a matrix instruction costs two LDS reads and a convert.
constexpr int VM_CH = 64;  // rows of U per staged piece
__global__ __launch_bounds__(BLOCK) void k_value_mfma(const float* __restrict__ pi, int64_t stride_b, int64_t N, int TP1, int d,
                                                      const double* __restrict__ w, double* __restrict__ V) {
  extern __shared__ __attribute__((aligned(16))) float smx[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int li = lane & 15, lk = lane >> 4;
  const int nt = d >> 4, pitch = d + 4;  // +4 floats: the 16 state rows of an A operand fall on distinct banks
  const int Q = d * (d + 1) / 2;
  float* xs = smx + (size_t)wv * 16 * pitch;
  double* Bs = reinterpret_cast<double*>(smx + (size_t)WAVES * 16 * pitch);  // [2][VM_CH][16]
  const double invT = 1.0 / (double)TP1;
  const int64_t ngroups = (N + 15) / 16;
  const int64_t npass = (ngroups + WAVES - 1) / WAVES;
  // staging role of this thread: column sc of the tile, rows sr, sr + 16, sr + 32, sr + 48 of the piece
  const int sc = tid & 15, sr = tid >> 4;
  double stg[4];
#define MFG_VM_LOAD(jt_, ch_)                                                              \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                          \
    const int k = VM_CH * (ch_) + sr + 16 * q, c = 16 * (jt_) + sc;                        \
    stg[q] = (k <= c) ? w[k * d - (k * (k - 1)) / 2 + (c - k)] : 0.0;                      \
  }
#define MFG_VM_COMMIT(buf_)                                                                \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) Bs[((buf_) * VM_CH + sr + 16 * q) * 16 + sc] = stg[q];
  for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
    const int64_t n0 = (pass * WAVES + wv) * 16;
    __syncthreads();  // the previous pass is done with Bs
    // stage the 16 state rows (fp32, float4 copies: d is a multiple of 16 and rows are 16-byte aligned when stride_b % 4 == 0)
    for (int e = lane; e < 16 * (d >> 2); e += WAVE) {
      const int q = e / (d >> 2), c4 = e - q * (d >> 2);
      const int64_t n = n0 + q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) {
        const int64_t b = (int64_t)(((double)n + 0.5) * invT);
        v = *reinterpret_cast<const float4*>(pi + b * stride_b + (n - b * TP1) * (int64_t)d + 4 * c4);
      }
      *reinterpret_cast<float4*>(xs + q * pitch + 4 * c4) = v;
    }
    MFG_VM_LOAD(0, 0)
    MFG_VM_COMMIT(0)
    __syncthreads();
    int buf = 0;
    double vs[4] = {0.0, 0.0, 0.0, 0.0};
    for (int jt = 0; jt < nt; ++jt) {
      const int c = (jt << 4) + li;  // this lane's column of the tile
      v4d_t acc = (v4d_t)(0.0);
      const int nch = (16 * (jt + 1) + VM_CH - 1) / VM_CH;
      for (int ch = 0; ch < nch; ++ch) {
        // the next piece of the sweep: issue its loads now, commit them after this piece's matrix instructions
        const bool last_ch = ch + 1 == nch;
        const bool has_next = !last_ch || jt + 1 < nt;
        const int njt = last_ch ? jt + 1 : jt, nc = last_ch ? 0 : ch + 1;
        if (has_next) MFG_VM_LOAD(njt, nc)
        int ksteps = 4 * (jt + 1) - (VM_CH / 4) * ch;  // rows 0 .. 16 jt + 15 of U, four at a time
        if (ksteps > VM_CH / 4) ksteps = VM_CH / 4;
        const float* xa = xs + li * pitch + VM_CH * ch + lk;
        const double* bb = Bs + (buf * VM_CH + lk) * 16 + li;
#pragma unroll 4
        for (int ks = 0; ks < ksteps; ++ks)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xa[4 * ks], bb[64 * ks], acc, 0, 0, 0);  // A[i = li][k], B[k][j = li]
        if (has_next) MFG_VM_COMMIT(buf ^ 1)
        __syncthreads();
        buf ^= 1;
      }
      const double bc = w[Q + c];
#pragma unroll
      for (int v = 0; v < 4; ++v) vs[v] = fma(acc[v] + bc, (double)xs[(4 * v + lk) * pitch + c], vs[v]);  // D[i = 4v + lk][j = li]
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double t = vs[v];
      t += dpp_mov_f64<0xB1, 0xF>(t);
      t += dpp_mov_f64<0x4E, 0xF>(t);
      t += dpp_mov_f64<0x141, 0xF>(t);
      t += dpp_mov_f64<0x140, 0xF>(t);
      const int64_t n = n0 + 4 * v + lk;
      if (li == 0 && n < N) V[n] = t + w[Q + d];
    }
  }
#undef MFG_VM_LOAD
#undef MFG_VM_COMMIT
}

// ---------------------------------------------------------------------------------------------
// Round 3: GEMM-tiled form of k_value_mfma for d a multiple of 64 (C3, C5).  k_value_mfma keeps the FULL rows of a wave's
// 16 states in LDS (66 KB per block at d = 256: one block per CU, one wave per SIMD), one accumulator tile per wave (every
// matrix instruction waits for the one before) and a block barrier every 16 instructions: 226 cycles per instruction
// against ~70.  Here a block owns 128 states and walks the upper triangle of U in 64 x 64 pieces (column block cb, row
// chunk kc <= cb): a piece of U (fp64, zero below the diagonal) and the matching 128 x 64 slab of states (fp32) are staged
// -- fetched into registers under the previous piece's matrix instructions, committed behind them -- and a wave runs
// 2 state groups x 4 column tiles = 8 independent accumulators over the 16 K steps of the piece: 2 + 4 operand reads
// for 8 matrix instructions, 128 instructions between barriers, two blocks per CU.  After a column block's last piece the
// finished tiles Y = X U are folded into  v_n += sum_c (Y_nc + b_c) x_nc  (x: 32 LDS reads per lane per 64
// columns, read back from the slab of the diagonal piece).  The all-zero tiles of a diagonal piece are skipped.
// ---------------------------------------------------------------------------------------------
constexpr int VM2_MB = 128, VM2_KC = 64, VM2_XP = VM2_KC + 2;  // states per block, rows per piece, pitch of the state slab
                                                               // (2 mod 32: the 32 lanes of a half wave hit 32 banks)
__global__ __launch_bounds__(BLOCK, 2) void k_value_mfma2(const float* __restrict__ pi, int64_t stride_b, int64_t N, int TP1, int d,
                                                           const double* __restrict__ w, double* __restrict__ V) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* Us = reinterpret_cast<double*>(smem_raw);                   // [KC][64]
  float* Xs = reinterpret_cast<float*>(Us + VM2_KC * 64);             // [MB][XP]
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int li = lane & 15, lk = lane >> 4;
  const int ncb = d >> 6, Q = d * (d + 1) / 2;
  const double invT = 1.0 / (double)TP1;
  const int64_t npass = (N + VM2_MB - 1) / VM2_MB;
  // staging roles.  U piece: thread -> row ur = tid / 4, sixteen columns from uc = 16 (tid % 4) (the row of U is contiguous in
  // the packed weight vector); state slab: thread -> state xr + 16 u (u < 8), float4 column xc.
  const int ur = tid >> 2, uc = (tid & 3) << 4;
  const int xr = tid >> 4, xc = (tid & 15) << 2;
  double ustg[16];
  float4 xstg[8];
  for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
    const int64_t n0 = pass * VM2_MB;
    // row pointers of this thread's staged states (clamped; rows past N are never written out)
    const float* xrow[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int64_t n = n0 + xr + 16 * u;
      if (n >= N) n = N - 1;
      const int64_t b = (int64_t)(((double)n + 0.5) * invT);
      xrow[u] = pi + b * stride_b + (n - b * TP1) * (int64_t)d;
    }
    auto fetch = [&](int cbk, int kc) __attribute__((always_inline)) {
      const int k = VM2_KC * kc + ur, c0 = 64 * cbk + uc;
      const double* wr = w + ((int64_t)k * d - ((int64_t)k * (k - 1)) / 2 - k);  // U[k][c] = wr[c] for c >= k
#pragma unroll
      for (int q = 0; q < 16; ++q) ustg[q] = (c0 + q >= k) ? wr[c0 + q] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) xstg[u] = *reinterpret_cast<const float4*>(xrow[u] + VM2_KC * kc + xc);
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 16; q += 2) *reinterpret_cast<double2*>(Us + ur * 64 + uc + q) = make_double2(ustg[q], ustg[q + 1]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float* dst = Xs + (xr + 16 * u) * VM2_XP + xc;  // (pitch 66: 8-byte aligned only)
        *reinterpret_cast<float2*>(dst) = make_float2(xstg[u].x, xstg[u].y);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(xstg[u].z, xstg[u].w);
      }
    };
    double vs[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    fetch(0, 0);
    for (int cbk = 0; cbk < ncb; ++cbk) {
      v4d_t acc[2][4];
#pragma unroll
      for (int sg = 0; sg < 2; ++sg)
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) acc[sg][cg] = (v4d_t)(0.0);
      for (int kc = 0; kc <= cbk; ++kc) {
        __syncthreads();  // the previous piece's matrix instructions are done with Us / Xs
        commit();
        __syncthreads();
        // the next piece of the sweep (its loads land under this piece's matrix instructions)
        const bool last_kc = kc == cbk;
        if (!last_kc || cbk + 1 < ncb) fetch(last_kc ? cbk + 1 : cbk, last_kc ? 0 : kc + 1);
        const float* xa = Xs + (wv * 32 + li) * VM2_XP + lk;
        const double* ub = Us + lk * 64 + li;
        if (!last_kc) {
#pragma unroll 2
          for (int ks = 0; ks < VM2_KC / 4; ++ks) {
            const double a0 = (double)xa[4 * ks], a1 = (double)xa[16 * VM2_XP + 4 * ks];  // A[state li][k]
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
              const double bv = ub[4 * ks * 64 + 16 * cg];  // B[k][col li]
              acc[0][cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0][cg], 0, 0, 0);
              acc[1][cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1][cg], 0, 0, 0);
            }
          }
        } else {
          // the piece on the diagonal: rows 16 p .. 16 p + 15 of it are zero left of column tile p -- skip those tiles
#pragma unroll
          for (int pq = 0; pq < 4; ++pq) {
#pragma unroll 2
            for (int ks = 4 * pq; ks < 4 * pq + 4; ++ks) {
              const double a0 = (double)xa[4 * ks], a1 = (double)xa[16 * VM2_XP + 4 * ks];
#pragma unroll
              for (int cg = pq; cg < 4; ++cg) {
                const double bv = ub[4 * ks * 64 + 16 * cg];
                acc[0][cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0][cg], 0, 0, 0);
                acc[1][cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1][cg], 0, 0, 0);
              }
            }
          }
        }
      }
      // fold the finished 32 x 64 block of Y: D[i = 4 v + lk][j = li] of tile (sg, cg).  The state slab of the diagonal
      // piece (still in LDS: the next commit waits behind the barrier) holds exactly the columns of this block.
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        const double bc = w[Q + 64 * cbk + 16 * cg + li];
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const double x = (double)Xs[(wv * 32 + sg * 16 + 4 * v + lk) * VM2_XP + 16 * cg + li];
            vs[sg][v] = fma(acc[sg][cg][v] + bc, x, vs[sg][v]);
          }
        }
      }
    }
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double t = vs[sg][v];
        t += dpp_mov_f64<0xB1, 0xF>(t);
        t += dpp_mov_f64<0x4E, 0xF>(t);
        t += dpp_mov_f64<0x141, 0xF>(t);
        t += dpp_mov_f64<0x140, 0xF>(t);
        const int64_t n = n0 + wv * 32 + sg * 16 + 4 * v + lk;
        if (li == 0 && n < N) V[n] = t + w[Q + d];
      }
    }
  }
}

// delta[b, s] = r[b, s] + gd(s) V[b, s+1] - V[b, s],  gd = gamma (mfg_ac2.py:505) or the running gamma^s (ac_irl.py:691);
// reward == NULL: the IRL form without the reward (added by the gradient kernel once the network has run).
__global__ void k_td_delta(const double* __restrict__ V, const float* __restrict__ reward, int64_t B, int T, double gamma,
                           int discount_pow, double* __restrict__ delta) {
  const int64_t n_tot = B * T;
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < n_tot; n += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = n / T;
    const int s = (int)(n - b * T);
    double gd = gamma;
    if (discount_pow) {
      gd = 1.0;
      for (int q = 0; q < s; ++q) gd *= gamma;  // the running product of the in-kernel form, same rounding
    }
    const double r = reward ? (double)reward[n] : 0.0;
    delta[n] = r + gd * V[b * (T + 1) + s + 1] - V[b * (T + 1) + s];
  }
}

// Sum nsb partial rows in a fixed order: block = 64 slices x 16 outputs (16 consecutive doubles = one 128-byte line per
// slice); slice s adds rows s, s+64, ... (eight loads in flight), the 64 slice sums are combined in slice order through
// LDS.  (Round 2: 16 slices x 64 outputs -- four blocks for the 256 outputs of d = 21, each thread a chain of 4-6
// dependent L2 round trips: 4.8-6 us for the 342-512 rows of a per-step update; now FO / 16 blocks and one or two rounds.)
constexpr int RP_SLICES = 64, RP_OUT = 16;
// `ap` != NULL (single-GPU training rollout, accumulate == 0): the parameter update rides along -- the number of samples
// is known on the host (count), so every output updates its own parameter without waiting for another block's sum:
// k < F: w[k] += lr_c G[k] / count; k == F: theta += lr_a G[F] / count; k == F+1: *reward_acc += G[F+1] / count
// (the arithmetic of k_apply_update).
struct ReduceApply {
  double lr_c, lr_a, count;
  double *w, *theta, *reward_acc;
  int on;
};
__global__ __launch_bounds__(RP_SLICES* RP_OUT) void k_reduce_partials(const double* __restrict__ partial, int64_t nsb, int64_t FO,
                                                                      int accumulate, double* __restrict__ G, ReduceApply ap) {
  __shared__ double red[RP_SLICES][RP_OUT + 1];
  const int lo = threadIdx.x & (RP_OUT - 1), sl = threadIdx.x / RP_OUT;
  const int64_t k = (int64_t)blockIdx.x * RP_OUT + lo;
  // the value this output updates (parameter / accumulator / running G): read FIRST, under the row reads -- read where it is
  // used it was one more dependent L2 round trip at the end of a kernel that is nothing but such round trips
  double old_val = 0.0, old_G = 0.0;
  if (sl == 0 && k < FO) {
    const int64_t F = FO - 3;
    if (accumulate) old_G = G[k];
    if (ap.on) {
      if (k < F) old_val = ap.w[k];
      else if (k == F) old_val = *ap.theta;
      else if (k == F + 1 && ap.reward_acc) old_val = *ap.reward_acc;
    }
  }
  double s = 0.0;
  if (k < FO) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t p = sl;
    for (; p + 7 * RP_SLICES < nsb; p += 8 * RP_SLICES) {
      const double v0 = partial[p * FO + k], v1 = partial[(p + RP_SLICES) * FO + k], v2 = partial[(p + 2 * RP_SLICES) * FO + k],
                   v3 = partial[(p + 3 * RP_SLICES) * FO + k], v4 = partial[(p + 4 * RP_SLICES) * FO + k],
                   v5 = partial[(p + 5 * RP_SLICES) * FO + k], v6 = partial[(p + 6 * RP_SLICES) * FO + k],
                   v7 = partial[(p + 7 * RP_SLICES) * FO + k];
      s0 += v0;
      s1 += v1;
      s2 += v2;
      s3 += v3;
      s0 += v4;
      s1 += v5;
      s2 += v6;
      s3 += v7;
    }
    // tail: up to seven rows, loaded together
    double t[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) t[u] = (p + u * RP_SLICES < nsb) ? partial[(p + u * RP_SLICES) * FO + k] : 0.0;
    s0 += t[0];
    s1 += t[1];
    s2 += t[2];
    s3 += t[3];
    s0 += t[4];
    s1 += t[5];
    s2 += t[6];
    s = (s0 + s1) + (s2 + s3);
  }
  red[sl][lo] = s;
  __syncthreads();
  if (sl == 0 && k < FO) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
    for (int q = 0; q < RP_SLICES; q += 4) {
      t0 += red[q][lo];
      t1 += red[q + 1][lo];
      t2 += red[q + 2][lo];
      t3 += red[q + 3][lo];
    }
    const double tot = (t0 + t1) + (t2 + t3);
    const double gk = accumulate ? old_G + tot : tot;
    G[k] = gk;
    if (ap.on) {
      const int64_t F = FO - 3;
      const double inv = 1.0 / ap.count;
      if (k < F) ap.w[k] = updated_param(old_val, ap.lr_c, gk, inv);
      else if (k == F) *ap.theta = updated_param(old_val, ap.lr_a, gk, inv);
      else if (k == F + 1 && ap.reward_acc) *ap.reward_acc = old_val + gk * inv;
    }
  }
}

// The row reduction + update of the IRL env step as a launch of its own (after an episode's LAST step; the other steps' reductions
// ride in the next step kernel, k_core_small<.., STEP>): a wave per column, rows_column_sum -- the same order, the same bits.
__global__ __launch_bounds__(BLOCK) void k_reduce_rows_apply(const double* __restrict__ rows, int nrows, int64_t FO, double* __restrict__ G,
                                                            double lr_c, double lr_a, double count, double* __restrict__ w,
                                                            const double* theta_in, double* theta_out,
                                                            double* __restrict__ reward_acc) {
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int64_t k = (int64_t)blockIdx.x * WAVES + wv, F = FO - 3;
  if (k >= FO) return;
  double old_val = 0.0;
  if (lane == 0) {
    if (k < F) old_val = w[k];
    else if (k == F) old_val = *theta_in;
    else if (k == F + 1 && reward_acc) old_val = *reward_acc;
  }
  const double gk = rows_column_sum(rows, nrows, FO, k, lane);
  if (lane == 0) {
    const double inv = 1.0 / count;
    G[k] = gk;
    if (k < F) w[k] = updated_param(old_val, lr_c, gk, inv);
    else if (k == F) *theta_out = updated_param(old_val, lr_a, gk, inv);
    else if (k == F + 1 && reward_acc) *reward_acc = old_val + gk * inv;
  }
}

// ---------------------------------------------------------------------------------------------
// host side: launch helpers
// ---------------------------------------------------------------------------------------------
static int grid_for(int64_t work_items, int per_block, int blocks_per_cu) {
  int64_t g = (work_items + per_block - 1) / per_block;
  const int64_t cap = (int64_t)num_cus() * blocks_per_cu;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

constexpr size_t MFG_WS_CONTROL_BYTES = 64;  // control block at the START of the workspace (completion counter), kept zero:
                                              // a fixed place, whatever N a call is made with; partial rows follow it
constexpr int MFG_GRAD_FUSE_MAX_ROWS = 32;   // in-kernel finalisation up to this many partial rows

struct ApplyArgs {
  double lr_c, lr_a;
  double *w, *theta, *reward_acc;
};

// Large-d rollouts evaluate the critic values of all states in one matrix-core pass after the rollout (k_value_mfma): room
// for V[B (T+1)] <= 2 N doubles behind the partial rows of the gradient sums.
static bool value_batch_ok(int d) { return d > WAVE && d % 16 == 0 && d <= MFG_MAX_D; }
static size_t value_buffer_bytes(int64_t N, int d) { return value_batch_ok(d) ? (size_t)(2 * N) * 8 : 0; }

static void grad_geometry(int64_t N, int d, int* chunk, int64_t* nsb, int* nob) {
  const int64_t FO = mfg_num_features(d) + 3;
  int ch = 8192 / d;
  if (ch > 64) ch = 64;
  if (ch < 8) ch = 8;
  *chunk = ch;
  int64_t sb = (N + ch - 1) / ch;
#ifndef MFG_GRAD_WS_MB
#define MFG_GRAD_WS_MB 72  // partial rows of the batch sums: 256 rows at d = 256 (one per CU; 32 MB until round 3)
#endif
  int64_t cap = (int64_t)((long long)MFG_GRAD_WS_MB << 20) / (FO * 8);
  if (cap > 1024) cap = 1024;
  if (cap < 16) cap = 16;
  if (sb > cap) sb = cap;
  if (sb < 1) sb = 1;
  *nsb = sb;
  *nob = (int)((FO + GR_OUT_PER_BLOCK - 1) / GR_OUT_PER_BLOCK);
}

// `apply` (optional): also perform the parameter update; *applied tells whether it was done in-kernel.
static int launch_grad(const float* pi, int64_t stride_b, const double* delta, const double* g, const float* reward,
                       int64_t N, int T, int d, double* G, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
                       const ApplyArgs* apply = nullptr, bool* applied = nullptr, bool add_reward = false) {
  if (applied) *applied = false;
  const int64_t FO = mfg_num_features(d) + 3;
  int chunk, nob;
  int64_t nsb;
  grad_geometry(N, d, &chunk, &nsb, &nob);
  const size_t need = (size_t)(nsb * FO * 8) + MFG_WS_CONTROL_BYTES;  // (the value buffer, if any, lies behind)
  if (ws_bytes < need) return fail(MFG_EWORKSPACE, "%s: need %lld bytes, have %lld", "workspace", (long long)need,
                                   (long long)ws_bytes);
  // the parameter update, when asked for, rides in the kernel that finishes the sums (fixed-order; needs accumulate == 0
  // so that the sample count is the host-known N)
  ReduceApply rap{};
  if (apply && !accumulate) {
    rap.on = 1;
    rap.lr_c = apply->lr_c;
    rap.lr_a = apply->lr_a;
    rap.count = (double)N;
    rap.w = apply->w;
    rap.theta = apply->theta;
    rap.reward_acc = apply->reward_acc;
  }
  GradArgs a{};
  a.pi = pi;
  a.stride_b = stride_b;
  a.delta = delta;
  a.g = g;
  a.reward = reward;
  a.N = N;
  a.T = T;
  a.d = d;
  a.chunk = chunk;
  a.nsb = nsb;
  a.partial = reinterpret_cast<double*>((char*)ws + MFG_WS_CONTROL_BYTES);
  // (the small-d kernel folds delta += reward into its own load: every sample is read by exactly one wave there; the
  //  other kernels read a sample from several blocks, so the update is a separate elementwise launch first)
  const bool small = d <= MFG_GRAD_SMALL_MAX_D;
  if (add_reward && !small)
    hipLaunchKernelGGL(k_add_reward, dim3(grid_for(N, 256, 8)), dim3(256), 0, st, const_cast<double*>(delta), reward, N);
  if (small) {
    a.add_reward = add_reward ? 1 : 0;
    // one partial row per block; nsb rows fit the workspace by construction (grad_geometry).  A wave works through chunks of
    // 64 samples: one chunk per wave while the batch is small, then more chunks per wave (two blocks per CU at most)
    const int64_t NC = (N + GS_CH - 1) / GS_CH;  // chunks of 64 samples, one wave each at a time
    int64_t blocks = (NC + WAVES - 1) / WAVES;
    const int64_t cap = (int64_t)num_cus() * MFG_GS_BPC;
    if (blocks > cap) blocks = cap;
    if (blocks > nsb) blocks = nsb;
    if (blocks < 1) blocks = 1;
    a.nsb = blocks;
    // few rows (small batches, per-step updates): the last block to finish sums them (and applies the update) itself
    const bool fuse = blocks <= MFG_GRAD_FUSE_MAX_ROWS;
    if (fuse) {
      a.counter = reinterpret_cast<unsigned*>(ws);
      a.G = G;
      a.accumulate = accumulate;
      if (apply) {
        a.apply = 1;
        a.lr_c = apply->lr_c;
        a.lr_a = apply->lr_a;
        a.w = apply->w;
        a.theta = apply->theta;
        a.reward_acc = apply->reward_acc;
        if (applied) *applied = true;
      }
    }
    const bool wide = stride_b - (int64_t)T * d >= (1 << 23);  // floats skipped between trajectories: 32-bit offsets inside a chunk?
    if (wide) hipLaunchKernelGGL((k_grad_mfma_small<0, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, a);
    else if (d == 21) hipLaunchKernelGGL((k_grad_mfma_small<21>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, a);
    else if (d == 15) hipLaunchKernelGGL((k_grad_mfma_small<15>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_grad_mfma_small<0>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, a);
    if (fuse) return check_launch("grad_mfma_small");
    hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((FO + RP_OUT - 1) / RP_OUT)), dim3(RP_SLICES * RP_OUT), 0, st,
                       (const double*)a.partial, blocks, FO, accumulate, G, rap);
    if (applied && rap.on) *applied = true;
    return check_launch("grad_mfma_small");
  }
  if (d % 16 == 0 && d >= 64 && d <= 4 * BLOCK) {
    const int nt = d / 16, ntiles = nt * (nt + 1) / 2;
    const int ny = (ntiles + WAVES * GM_TPW - 1) / (WAVES * GM_TPW);
    const int tpw = (ntiles + ny * WAVES - 1) / (ny * WAVES);
    a.chunk = (BLOCK % (d / 4) == 0 && (((uintptr_t)pi & 15) == 0) && (stride_b % 4 == 0)) ? 1 : 0;  // float4 staging
    const size_t lds_m = (size_t)GM_KC * 8 + (size_t)4 * BLOCK * 8 + (size_t)GM_KC * (d + 16) * 4;
    const int npf = a.chunk ? d / 32 : 0;
#ifndef MFG_GRAD_MFMA_OLD
    if (a.chunk && (d == 128 || d == 256)) {
      // two resident blocks per CU (two waves per SIMD: the second hides the first one's staging and barriers)
#ifndef MFG_GM2_BPC128
#define MFG_GM2_BPC128 3  // 166 registers: three blocks per CU fit (measured 253 -> 236 us at C3)
#endif
      const int64_t want = d == 256 ? (int64_t)num_cus() * 2 / Gm2Geom<256>::NY : (int64_t)num_cus() * MFG_GM2_BPC128 / Gm2Geom<128>::NY;
      if (nsb > want) nsb = want;
      a.nsb = nsb;
      if (d == 256) hipLaunchKernelGGL((k_grad_mfma2<256>), dim3((unsigned)nsb, (unsigned)Gm2Geom<256>::NY), dim3(BLOCK), lds_m, st, a);
      else hipLaunchKernelGGL((k_grad_mfma2<128>), dim3((unsigned)nsb, (unsigned)Gm2Geom<128>::NY), dim3(BLOCK), lds_m, st, a);
    } else
#endif
    switch (npf) {
      case 2: hipLaunchKernelGGL((k_grad_mfma<2>), dim3((unsigned)nsb, (unsigned)ny), dim3(BLOCK), lds_m, st, a, tpw); break;
      case 4: hipLaunchKernelGGL((k_grad_mfma<4>), dim3((unsigned)nsb, (unsigned)ny), dim3(BLOCK), lds_m, st, a, tpw); break;
      case 8: hipLaunchKernelGGL((k_grad_mfma<8>), dim3((unsigned)nsb, (unsigned)ny), dim3(BLOCK), lds_m, st, a, tpw); break;
      case 16: hipLaunchKernelGGL((k_grad_mfma<16>), dim3((unsigned)nsb, (unsigned)ny), dim3(BLOCK), lds_m, st, a, tpw); break;
      default: hipLaunchKernelGGL((k_grad_mfma<0>), dim3((unsigned)nsb, (unsigned)ny), dim3(BLOCK), lds_m, st, a, tpw); break;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((FO + RP_OUT - 1) / RP_OUT)), dim3(RP_SLICES * RP_OUT), 0, st,
                       (const double*)a.partial, nsb, FO, accumulate, G, rap);
    if (applied && rap.on) *applied = true;
    return check_launch("grad_mfma");
  }
  const size_t lds = (size_t)chunk * 3 * 8 + (size_t)chunk * d * 4;
  hipLaunchKernelGGL(k_grad_partial, dim3((unsigned)nsb, (unsigned)nob), dim3(BLOCK), lds, st, a);
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((FO + RP_OUT - 1) / RP_OUT)), dim3(RP_SLICES * RP_OUT), 0, st,
                     (const double*)a.partial, nsb, FO, accumulate, G, rap);
  if (applied && rap.on) *applied = true;
  return check_launch("grad_reduce");
}

// ---- h(z) table (mfg_device.h): module-resident, fitted once per device in fp64 -------------------------
__device__ float4 g_htab[HTAB_N];

__global__ void k_init_htab(float4* __restrict__ tab) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= HTAB_N) return;
  double y[4];
  for (int q = 0; q < 4; ++q) {
    const double z = (double)HTAB_ZMIN + ((double)k + (double)q / 3.0) / (double)HTAB_PER_UNIT;
    double sp, sg;
    softplus_sigmoid(z, sp, sg);
    y[q] = digamma_pos(sp) * sg * INV_LN2;  // log2 units, see mfg_device.h
  }
  // cubic through f = 0, 1/3, 2/3, 1 (Newton forward differences, t = 3 f)
  const double d1 = y[1] - y[0], d2 = y[2] - 2.0 * y[1] + y[0], d3 = y[3] - 3.0 * y[2] + 3.0 * y[1] - y[0];
  tab[k] = make_float4((float)y[0], (float)(3.0 * (d1 - 0.5 * d2 + d3 / 3.0)), (float)(9.0 * (0.5 * d2 - 0.5 * d3)),
                       (float)(27.0 * d3 / 6.0));
}

// Device address of the table; the first call on a device fits it (one launch + one device sync, ever).
static const float4* htab_ptr() {
  static std::mutex mu;
  static const float4* ptr[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_htab)) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(k_init_htab, dim3((HTAB_N + 255) / 256), dim3(256), 0, 0, (float4*)p);
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    ptr[dev] = (const float4*)p;
  }
  return ptr[dev];
}

// ---- device status word (mfg_status): one host-mapped 32-bit word per device.  Kernels OR condition bits into it through
// its device address (a plain store from one lane: the conditions are functions of theta alone, so every launch that sees
// the condition writes the same bits); the host reads it through the host address without synchronising.
struct StatusWord {
  unsigned* host = nullptr;
  unsigned* dev = nullptr;
};
// ---- contexts (mfg_ctx_t, include/mfg_hip.h): the mutable state a launch touches -- the status word, an RCCL communicator --
// owned by an opaque object instead of the process.  A thread binds a context (mfg_ctx_bind) and every entry point called
// from that thread acts on it; with none bound the device's default context (the process-wide word of ABI 14) is used.
struct mfg_ctx {
  int device = -1;
  StatusWord sw;
  void* comm = nullptr;  // adopted RCCL communicator (mfg_ctx_adopt_comm), destroyed with the context
};
// The binding is per THREAD, the object's lifetime is not: a context may be destroyed by another thread than the one(s) it is
// bound on (a garbage collector drops the last reference wherever it runs).  Every live context is therefore registered with
// a generation id, a thread's binding remembers (pointer, generation), and a destroy bumps a global epoch: the next entry
// point of a thread that finds the epoch moved re-validates its binding under the registry lock and, if its context is gone
// (or the address now belongs to a younger one), falls back to the device's default word instead of touching freed memory.
// The hot path (no destroy since the last look) is one relaxed-cost atomic load.
static thread_local mfg_ctx* g_ctx = nullptr;
static thread_local uint64_t g_ctx_gen = 0, g_ctx_seen_epoch = 0;
static std::mutex g_ctx_mu;
static std::unordered_map<mfg_ctx*, uint64_t> g_ctx_live;  // live contexts -> generation id (under g_ctx_mu)
static uint64_t g_ctx_next_gen = 1;                          // (under g_ctx_mu)
static std::atomic<uint64_t> g_ctx_epoch{1};                 // bumped by every mfg_ctx_destroy
static bool ctx_is_live(mfg_ctx* c) {
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  return g_ctx_live.find(c) != g_ctx_live.end();
}
// the calling thread's bound context; NULL if none, or if it has been destroyed (by any thread) since it was bound
static mfg_ctx* bound_ctx() {
  if (!g_ctx) return nullptr;
  if (g_ctx_epoch.load(std::memory_order_acquire) != g_ctx_seen_epoch) {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    const auto it = g_ctx_live.find(g_ctx);
    if (it == g_ctx_live.end() || it->second != g_ctx_gen) g_ctx = nullptr;
    g_ctx_seen_epoch = g_ctx_epoch.load(std::memory_order_acquire);
  }
  return g_ctx;
}
extern "C" int mfg_dist_destroy(void* comm);
static bool alloc_status_word(StatusWord* out) {
  void* h = nullptr;
  void* d = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return false;
  memset(h, 0, 64);
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipHostFree(h);
    return false;
  }
  out->host = (unsigned*)h;
  out->dev = (unsigned*)d;
  return true;
}
static StatusWord status_word() {
  static std::mutex mu;
  static StatusWord sw[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return StatusWord{};
  if (mfg_ctx* c = bound_ctx()) return c->device == dev ? c->sw : StatusWord{};  // (a context of another device: refused by the callers)
  std::lock_guard<std::mutex> lock(mu);
  if (!sw[dev].host && !alloc_status_word(&sw[dev])) return StatusWord{};
  return sw[dev];
}
static int status_error(unsigned bits) {
  if (bits & MFG_STATUS_MIXED_RANGE)
    return fail(MFG_ERANGE, "%s", "an earlier mixed-precision sampling launch ran with |theta| (1/2 + |shift|) > 86 (or theta not "
                                  "finite): the fp32 factors of the separable exponential left their range and that launch's "
                                  "outputs are NaN; use MFG_PRECISION_F64 / precision='f64' for such policies, then mfg_clear_status()");
  return fail(MFG_ERANGE, "device status word = 0x%x", bits);
}

// mfg_train_rollouts_dist: the sticky status word is read ONCE in front of the episode loop and once behind it, not per
// enqueue -- the host read races the device by a different number of episodes on every rank, and a rank that stops
// enqueueing leaves its peers' all-reduces without a partner (the private communicator has no watchdog).
static thread_local bool g_status_check_deferred = false;

static int launch_core(const CoreArgs& a_in, bool sample, bool td, int precision, hipStream_t st) {
  CoreArgs a = a_in;
  {
    const StatusWord sw = status_word();
    if (!sw.host) return fail(MFG_ELAUNCH, "%s", "status word allocation failed");
    const unsigned bits = g_status_check_deferred ? 0u : *(volatile unsigned*)sw.host;
    // sticky until mfg_clear_status() -- for the launches the condition concerns: MFG_STATUS_MIXED_RANGE is a property of
    // mixed-precision SAMPLING (theta beyond the range of its fp32 factors); strict-precision launches and launches on given
    // actions have no such limit and go ahead whatever another instance / thread on this device ran into
    const unsigned blocking = (sample && precision == MFG_PRECISION_MIXED) ? bits : (bits & ~(unsigned)MFG_STATUS_MIXED_RANGE);
    if (blocking) return status_error(blocking);
    a.status = sw.dev;
  }
#ifdef MFG_TIMING
  {
    const char* e = getenv("MFG_TIMING_BUF");
    a.dbg = e ? (unsigned long long*)strtoull(e, nullptr, 16) : nullptr;
  }
#endif
  if (td && precision == MFG_PRECISION_MIXED) {
    a.htab = htab_ptr();
    if (!a.htab) return fail(MFG_ELAUNCH, "%s", "h(z) table initialisation failed");
  }
  int rc;
  if (a.d <= WAVE) rc = launch_core_small(a, sample, td, precision == MFG_PRECISION_MIXED, num_cus(), st);
  else if (precision == MFG_PRECISION_MIXED) rc = launch_core_large_mixed(a, sample, td, num_cus(), st);
  else rc = launch_core_large_f64(a, sample, td, num_cus(), st);
  if (rc != MFG_OK) return fail(rc, "%s: d=%lld > %lld", "core", (long long)a.d, (long long)MFG_MAX_D);
  return check_launch("core");
}

// Per-step updates at the packed sizes (T == 1, d = 21 / 15, in-kernel reward): the step kernel itself leaves one partial
// row of the batch sums per tile (k_core_small<..., SUMS>), so an update is [step kernel | row reduction (+ apply)]
// instead of [step kernel | gradient kernel with its own finish].  Used while every tile has its own resident block
// (MFG_CORE_SUMS_MAX_ROWS tiles = 6 144 trajectories at d = 21) and the workspace has room for the rows; otherwise the
// caller takes the two-kernel path.
constexpr int64_t MFG_CORE_SUMS_MAX_ROWS = 512;  // = blocks resident at the SUMS variant's two waves per SIMD (256 CUs x 2)
static int64_t core_sums_rows(int d, int64_t B) {
  if (!(d == 21 || d == 15)) return 0;
  const int TB = WAVES * (WAVE / d);
  const int64_t nt = (B + TB - 1) / TB;
  return nt <= MFG_CORE_SUMS_MAX_ROWS ? nt : 0;
}
static bool core_sums_ok(int d, int64_t B, int T, int reward_kind, const void* ws, size_t ws_bytes) {
  if (T != 1 || reward_kind == MFG_REWARD_EXTERNAL || !ws) return false;
  const int64_t nt = core_sums_rows(d, B);
  return nt > 0 && ws_bytes >= (size_t)(nt * (mfg_num_features(d) + 3) * 8) + MFG_WS_CONTROL_BYTES;
}
// the row reduction (+ optional parameter update) that follows a SUMS launch
static int reduce_core_sums(int d, int64_t B, double* G, int accumulate, void* ws, const ApplyArgs* apply, hipStream_t st) {
  const int64_t FO = mfg_num_features(d) + 3, nt = core_sums_rows(d, B);
  ReduceApply rap{};
  if (apply && !accumulate) {
    rap.on = 1;
    rap.lr_c = apply->lr_c;
    rap.lr_a = apply->lr_a;
    rap.count = (double)B;
    rap.w = apply->w;
    rap.theta = apply->theta;
    rap.reward_acc = apply->reward_acc;
  }
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((FO + RP_OUT - 1) / RP_OUT)), dim3(RP_SLICES * RP_OUT), 0, st,
                     (const double*)((char*)ws + MFG_WS_CONTROL_BYTES), nt, FO, accumulate, G, rap);
  return check_launch("core_sums");
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char* mfg_last_error(void) { return g_err; }
int mfg_abi_version(void) { return 17; }

int mfg_init(void) {
  if (!htab_ptr()) return fail(MFG_ELAUNCH, "%s", "mfg_init: no HIP device / table initialisation failed");
  if (!status_word().host) return fail(MFG_ELAUNCH, "%s", "mfg_init: status word allocation failed");
  return MFG_OK;
}

int mfg_set_core_mapping(int mode) { return core_mapping_set(mode); }

int mfg_status(unsigned* bits_host) {
  const StatusWord sw = status_word();
  if (!sw.host) return fail(MFG_ELAUNCH, "%s", "status word allocation failed");
  const unsigned bits = *(volatile unsigned*)sw.host;
  if (bits_host) *bits_host = bits;
  return bits ? status_error(bits) : MFG_OK;
}

int mfg_clear_status(void) {
  const StatusWord sw = status_word();
  if (!sw.host) return fail(MFG_ELAUNCH, "%s", "status word allocation failed");
  *(volatile unsigned*)sw.host = 0u;
  return MFG_OK;
}

int mfg_ctx_create(mfg_ctx_t** ctx_out) {
  REQUIRE(ctx_out, "null pointer");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(MFG_ELAUNCH, "%s", "mfg_ctx_create: no HIP device");
  if (!htab_ptr()) return fail(MFG_ELAUNCH, "%s", "mfg_ctx_create: table initialisation failed");  // (shared, immutable, per device)
  mfg_ctx* c = new (std::nothrow) mfg_ctx();
  if (!c) return fail(MFG_ELAUNCH, "%s", "mfg_ctx_create: out of memory");
  c->device = dev;
  if (!alloc_status_word(&c->sw)) {
    delete c;
    return fail(MFG_ELAUNCH, "%s", "mfg_ctx_create: status word allocation failed");
  }
  {
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    g_ctx_live[c] = g_ctx_next_gen++;
  }
  *ctx_out = c;
  return MFG_OK;
}

int mfg_ctx_destroy(mfg_ctx_t* ctx) {
  if (!ctx) return MFG_OK;
  {
    // out of the registry first, then the epoch: a thread that still has it bound (this one or any other) drops the binding
    // at its next entry point (bound_ctx) instead of dereferencing the freed object
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (g_ctx_live.erase(ctx) == 0) return fail(MFG_EINVAL, "%s", "mfg_ctx_destroy: not a live context (destroyed twice?)");
    g_ctx_epoch.fetch_add(1, std::memory_order_release);
  }
  if (g_ctx == ctx) g_ctx = nullptr;
  int rc = MFG_OK;
  if (ctx->comm) rc = mfg_dist_destroy(ctx->comm);
  if (ctx->sw.host) (void)hipHostFree(ctx->sw.host);
  delete ctx;
  return rc;
}

int mfg_ctx_bind(mfg_ctx_t* ctx) {
  uint64_t gen = 0, epoch = 0;
  if (ctx) {
    {
      std::lock_guard<std::mutex> lock(g_ctx_mu);
      const auto it = g_ctx_live.find(ctx);
      if (it == g_ctx_live.end()) return fail(MFG_EINVAL, "%s", "mfg_ctx_bind: not a live context (already destroyed?)");
      gen = it->second;
      epoch = g_ctx_epoch.load(std::memory_order_acquire);
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev != ctx->device)
      return fail(MFG_EINVAL, "mfg_ctx_bind: the context belongs to device %d, the current device is %d", ctx->device, dev);
  }
  g_ctx = ctx;
  g_ctx_gen = gen;
  g_ctx_seen_epoch = epoch;
  return MFG_OK;
}

mfg_ctx_t* mfg_ctx_current(void) { return bound_ctx(); }

int mfg_ctx_status(mfg_ctx_t* ctx, unsigned* bits_host) {
  REQUIRE(ctx && ctx_is_live(ctx) && ctx->sw.host, "null or destroyed context");
  const unsigned bits = *(volatile unsigned*)ctx->sw.host;
  if (bits_host) *bits_host = bits;
  return bits ? status_error(bits) : MFG_OK;
}

int mfg_ctx_clear_status(mfg_ctx_t* ctx) {
  REQUIRE(ctx && ctx_is_live(ctx) && ctx->sw.host, "null or destroyed context");
  *(volatile unsigned*)ctx->sw.host = 0u;
  return MFG_OK;
}

int mfg_ctx_adopt_comm(mfg_ctx_t* ctx, void* comm) {
  REQUIRE(ctx && ctx_is_live(ctx), "null or destroyed context");
  ctx->comm = comm;
  return MFG_OK;
}

void* mfg_ctx_comm(mfg_ctx_t* ctx) { return (ctx && ctx_is_live(ctx)) ? ctx->comm : nullptr; }

int mfg_device_info(int* cu_count_host, char* arch_host, int arch_len) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess)
    return fail(MFG_ELAUNCH, "%s", "no HIP device");
  if (cu_count_host) *cu_count_host = p.multiProcessorCount;
  if (arch_host && arch_len > 0) {
    strncpy(arch_host, p.gcnArchName, (size_t)arch_len - 1);
    arch_host[arch_len - 1] = 0;
  }
  return MFG_OK;
}

int64_t mfg_num_features(int d) { return (int64_t)d * (d + 1) / 2 + d + 1; }

int64_t mfg_feature_index(int i, int j, int d) {
  if (i > j) {
    const int t = i;
    i = j;
    j = t;
  }
  return (int64_t)i * d - ((int64_t)i * (i - 1)) / 2 + (j - i);
}

size_t mfg_workspace_bytes(int64_t N, int d) {
  if (N < 1 || d < 1) return 0;
  int chunk, nob;
  int64_t nsb;
  grad_geometry(N, d, &chunk, &nsb, &nob);
  const int64_t rows_sums = core_sums_rows(d, N);  // T = 1 updates with the sums fused into the step kernel: a row per tile
  if (rows_sums > nsb) nsb = rows_sums;
  if (d <= 32) {  // IRL steps with the sums fused into the reward-network launch: a row per block of eight samples, <= 512
    const int64_t rows_rn = (N + 7) / 8 < 512 ? (N + 7) / 8 : 512;
    if (rows_rn > nsb) nsb = rows_rn;
  }
  return (size_t)(nsb * (mfg_num_features(d) + 3) * 8) + MFG_WS_CONTROL_BYTES + value_buffer_bytes(N, d);
}

#define CHECK_BD()                                        \
  REQUIRE(B >= 0, "B < 0");                               \
  REQUIRE(d >= 1, "d < 1");                               \
  if (d > MFG_MAX_D) return fail(MFG_EUNSUPPORTED, "%s: d=%lld > %lld", "shape", (long long)d, (long long)MFG_MAX_D); \
  if (B == 0) return MFG_OK;

int mfg_gather_start(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d, float* pi0,
                     mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(mat_pi0 && idx && pi0 && num_start > 0, "null pointer / empty table");
  hipLaunchKernelGGL(k_gather_start, dim3(grid_for(B * d, 256, 8)), dim3(256), 0, S(stream), mat_pi0, num_start, idx, B,
                     d, pi0);
  return check_launch("gather_start");
}

int mfg_draw_start(const float* mat_pi0, int64_t num_start, int64_t B, int d, uint64_t seed, uint32_t step, uint64_t traj_offset,
                   int32_t* idx, float* pi0, mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(num_start > 0 && num_start <= 0x7FFFFFFF, "empty / oversized start-state table");
  REQUIRE(idx || pi0, "nothing to write");
  REQUIRE(!pi0 || mat_pi0, "null start-state table");
  hipLaunchKernelGGL(k_draw_start, dim3(grid_for(pi0 ? B * d : B, 256, 8)), dim3(256), 0, S(stream), mat_pi0, num_start, B, d,
                     seed, step, traj_offset, idx, pi0);
  return check_launch("draw_start");
}

int mfg_alpha(const float* pi, int64_t B, int d, const double* theta, double shift, double* alpha, double* alpha_deriv,
              mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(pi && theta && (alpha || alpha_deriv), "null pointer");
  hipLaunchKernelGGL(k_alpha, dim3(grid_for(B * d * d, 256, 8)), dim3(256), 0, S(stream), pi, B, d, theta, shift, alpha,
                     alpha_deriv);
  return check_launch("alpha");
}

int mfg_dirichlet_from_gamma(const float* y, int64_t B, int d, float* P, mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(y && P, "null pointer");
  hipLaunchKernelGGL(k_dirichlet_from_gamma, dim3(grid_for(B * d, WAVES, 8)), dim3(BLOCK), 0, S(stream), y, B * d, d, P);
  return check_launch("dirichlet_from_gamma");
}

int mfg_philox_raw(uint64_t seed, uint32_t first_ctr, uint32_t c1, uint32_t c2, uint32_t c3, int64_t n, uint32_t* out,
                   mfg_stream_t stream) {
  REQUIRE(n >= 0 && (out || n == 0), "null pointer");
  if (n == 0) return MFG_OK;
  hipLaunchKernelGGL(k_philox_raw, dim3(grid_for(n, 256, 8)), dim3(256), 0, S(stream), seed, first_ctr, c1, c2, c3, n, out);
  return check_launch("philox_raw");
}

// developer knob (tools/step_probe.py): MFG_STEP_BATCH = 0 | 8 | 16 | 32 forces the tiles per super tile of the d = 21 / 15
// given-P kernel (0 = per-tile stores); unset / anything else = chosen from the batch size
static int step_batch_override() {
#ifdef MFG_DEV_KNOBS  // developer builds only (tools/variant.sh ... "-DMFG_DEV_KNOBS"): the shipped library reads no environment
  static const int v = [] {
    const char* e = getenv("MFG_STEP_BATCH");
    if (!e || !*e) return -1;
    const int k = atoi(e);
    return (k == 0 || k == 8 || k == 16 || k == 32) ? k : -1;
  }();
  return v;
#else
  return -1;
#endif
}

int mfg_step_given_P(const float* pi, const float* P, int64_t B, int d, int reward_kind, float* pi_next, float* reward,
                     mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(pi && P && pi_next, "null pointer");
  REQUIRE(reward_kind >= 0 && reward_kind <= 2, "bad reward_kind");
  if (!reward) reward_kind = MFG_REWARD_EXTERNAL;
  hipStream_t st = S(stream);
  if (d <= WAVE) {
    const int G = WAVE / d, TB = WAVES * G;
    const size_t lds_pre = (size_t)TB * d * d * 4 + (size_t)((TB * d + 3) & ~3) * 4;   // prefetching kernel (fp32 pi)
    const size_t lds_una = (size_t)TB * d * d * 4 + (size_t)TB * d * 16;               // unaligned fallback (fp64 pi, pi^2)
    const size_t lds = (((uintptr_t)P & 15) == 0) ? lds_pre : lds_una;
    int bpc = (int)((160 * 1024) / (lds + 64));
    if (bpc > 8) bpc = 8;
    if (bpc < 1) bpc = 1;
    const int grid = grid_for(B, TB, bpc);
    const bool aligned = (((uintptr_t)P & 15) == 0);
    const int per = (int)(((int64_t)TB * d * d / 4 + BLOCK - 1) / BLOCK);  // 16-byte loads per thread per tile
#define STEP_SMALL(K, DD, PP) \
  hipLaunchKernelGGL((k_step_small<K, DD, PP>), dim3(grid), dim3(BLOCK), lds, st, pi, P, B, d, pi_next, reward)
#define STEP_SMALL_D(DD, PP)                 \
  switch (reward_kind) {                     \
    case 0: STEP_SMALL(0, DD, PP); break;    \
    case 1: STEP_SMALL(1, DD, PP); break;    \
    default: STEP_SMALL(2, DD, PP); break;   \
  }
    if (!aligned) {
      switch (reward_kind) {
        case 0: hipLaunchKernelGGL((k_step_small_unaligned<0>), dim3(grid), dim3(BLOCK), lds, st, pi, P, B, d, pi_next, reward); break;
        case 1: hipLaunchKernelGGL((k_step_small_unaligned<1>), dim3(grid), dim3(BLOCK), lds, st, pi, P, B, d, pi_next, reward); break;
        default: hipLaunchKernelGGL((k_step_small_unaligned<2>), dim3(grid), dim3(BLOCK), lds, st, pi, P, B, d, pi_next, reward); break;
      }
    } else if (d == 21 || d == 15) {
#ifdef MFG_STEP_BLOCK_TILES
      if (d == 21) { STEP_SMALL_D(21, 6) } else { STEP_SMALL_D(15, 4) }
#else
      // The main launch covers B4 = B - B mod lcm(G, 4) trajectories (full tiles only and a whole number of 16-byte words
      // of P, so its prefetch needs no partial-tile / end-of-slab handling); the remaining < 12 trajectories go through
      // the ragged-capable instantiation.
      const int G = WAVE / d;
      const int64_t unit = (G % 4 == 0) ? G : (G % 2 == 0 ? 2 * G : 4 * G);
      const int64_t B4 = B - B % unit;
#define STEP_WAVE(K, DD, RAG, GRID, PI, PP, NB, PN, RW) \
  hipLaunchKernelGGL((k_step_wave<K, DD, RAG>), dim3(GRID), dim3(BLOCK), 0, st, PI, PP, NB, PN, RW)
#define STEP_WAVE_K(DD, RAG, GRID, PI, PP, NB, PN, RW)                    \
  switch (reward_kind) {                                                   \
    case 0: STEP_WAVE(0, DD, RAG, GRID, PI, PP, NB, PN, RW); break;        \
    case 1: STEP_WAVE(1, DD, RAG, GRID, PI, PP, NB, PN, RW); break;        \
    default: STEP_WAVE(2, DD, RAG, GRID, PI, PP, NB, PN, RW); break;       \
  }
      if (B4 > 0) {
        // Large batches: consecutive-tile super tiles with batched device-scope stores (k_step_wave_batched), as soon as
        // every one of the 8 waves per CU gets >= 2 super tiles; KB = tiles per super tile, the largest that qualifies.
        const int64_t ntiles = (B4 + G - 1) / G;
        const int64_t nw_target = (int64_t)num_cus() * 2 * WAVES;
        int kb = 0;
        for (int cand : {32, 16, 8}) {  // >= 2 super tiles per wave, and the last round >= 90 % full
          const int64_t ns = (ntiles + cand - 1) / cand, rr = (ns + nw_target - 1) / nw_target;
          if (!kb && rr >= 2 && 10 * ns >= 9 * rr * nw_target) kb = cand;
        }
        const int forced = step_batch_override();  // developer knob: MFG_STEP_BATCH = 0 (never) | 8 | 16 | 32
        if (forced >= 0) kb = forced;
        const bool out16 = (((uintptr_t)pi_next & 15) == 0) && (((uintptr_t)reward & 15) == 0);
        if (kb && out16) {
          const int64_t nsuper = (ntiles + kb - 1) / kb;
          const int64_t rounds = (nsuper + nw_target - 1) / nw_target;
          const int64_t nwv = (nsuper + rounds - 1) / rounds;
          const int gb = (int)((nwv + WAVES - 1) / WAVES);
#define STEP_WAVE_B(K, DD, KB) \
  hipLaunchKernelGGL((k_step_wave_batched<K, DD, KB>), dim3(gb), dim3(BLOCK), 0, st, pi, P, B4, pi_next, reward)
#define STEP_WAVE_BK(DD, KB)                    \
  switch (reward_kind) {                        \
    case 0: STEP_WAVE_B(0, DD, KB); break;      \
    case 1: STEP_WAVE_B(1, DD, KB); break;      \
    default: STEP_WAVE_B(2, DD, KB); break;     \
  }
          if (d == 21) {
            if (kb == 32) { STEP_WAVE_BK(21, 32) } else if (kb == 16) { STEP_WAVE_BK(21, 16) } else { STEP_WAVE_BK(21, 8) }
          } else {
            if (kb == 32) { STEP_WAVE_BK(15, 32) } else if (kb == 16) { STEP_WAVE_BK(15, 16) } else { STEP_WAVE_BK(15, 8) }
          }
#undef STEP_WAVE_BK
#undef STEP_WAVE_B
        } else {
          const int gw = grid_for(B4, G * WAVES, MFG_STEP_WAVES);
          if (d == 21) { STEP_WAVE_K(21, false, gw, pi, P, B4, pi_next, reward) }
          else { STEP_WAVE_K(15, false, gw, pi, P, B4, pi_next, reward) }
        }
      }
      if (B > B4) {
        const float* pi_t = pi + B4 * d;
        const float* P_t = P + B4 * d * d;   // 16-byte aligned: B4 is a multiple of 4
        float* pn_t = pi_next + B4 * d;
        float* rw_t = reward ? reward + B4 : nullptr;
        const int64_t Bt = B - B4;
        if (d == 21) { STEP_WAVE_K(21, true, 1, pi_t, P_t, Bt, pn_t, rw_t) }
        else { STEP_WAVE_K(15, true, 1, pi_t, P_t, Bt, pn_t, rw_t) }
      }
#undef STEP_WAVE_K
#undef STEP_WAVE
#endif
    }
    else if (per <= 4) { STEP_SMALL_D(0, 4) }
    else if (per <= 8) { STEP_SMALL_D(0, 8) }
    else { STEP_SMALL_D(0, 16) }
#undef STEP_SMALL_D
#undef STEP_SMALL
  } else {
    const size_t lds = (size_t)WAVES * d * 4;
    const int grid = grid_for(B, WAVES, (d == 128 || d == 256) ? MFG_ROWS_BPC : 8);
    const bool a16 = (((uintptr_t)P & 15) == 0) && (((uintptr_t)pi_next & 15) == 0);
    int vec = 1;
    if (a16 && d % 4 == 0) vec = 4;
    else if (a16 && d % 2 == 0) vec = 2;
    // keep at most 2 chunks per lane on the vector paths, fall back to narrower vectors otherwise
    int R = (d + WAVE * vec - 1) / (WAVE * vec);
#define STEP_ROWS(L, KT, GRID)                                                                                                \
  switch (reward_kind) {                                                                                                       \
    case 0: hipLaunchKernelGGL((k_step_rows<0, L, KT>), dim3(GRID), dim3(BLOCK), 0, st, pi, P, B, pi_next, reward); break;     \
    case 1: hipLaunchKernelGGL((k_step_rows<1, L, KT>), dim3(GRID), dim3(BLOCK), 0, st, pi, P, B, pi_next, reward); break;     \
    default: hipLaunchKernelGGL((k_step_rows<2, L, KT>), dim3(GRID), dim3(BLOCK), 0, st, pi, P, B, pi_next, reward); break;    \
  }
    if (a16 && (d == 128 || d == 256)) {
      // super tiles of MFG_ROWS_KT consecutive trajectories (batched output bursts) once every resident wave gets one
      const bool out16 = (((uintptr_t)pi_next & 15) == 0) && (((uintptr_t)reward & 15) == 0);
      const int64_t nw_target = (int64_t)num_cus() * MFG_ROWS_BPC * WAVES;
      // ... and the last round is >= 90 % full (a half-empty round of 8-trajectory units costs more than the bursts save:
      // 20 000 trajectories ran at 0.70 of peak as 2 500 super tiles over two rounds, 0.78 one trajectory at a time)
      int kt = 1;
      if (out16) {
        const int64_t ns8 = (B + MFG_ROWS_KT - 1) / MFG_ROWS_KT, rr8 = (ns8 + nw_target - 1) / nw_target;
        if (ns8 >= nw_target && 10 * ns8 >= 9 * rr8 * nw_target) kt = MFG_ROWS_KT;
      }
      const int64_t nsup = (B + kt - 1) / kt;
      const int64_t rounds = (nsup + nw_target - 1) / nw_target;
      const int gr = (int)(((nsup + rounds - 1) / rounds + WAVES - 1) / WAVES);
      if (d == 128) {
        if (kt > 1) { STEP_ROWS(32, MFG_ROWS_KT, gr) } else { STEP_ROWS(32, 1, gr) }
      } else {
        if (kt > 1) { STEP_ROWS(64, MFG_ROWS_KT, gr) } else { STEP_ROWS(64, 1, gr) }
      }
      return check_launch("step_given_P");
    }
#undef STEP_ROWS
#define STEP_LARGE(V, RR, K) \
  hipLaunchKernelGGL((k_step_large<V, RR, K>), dim3(grid), dim3(BLOCK), lds, st, pi, P, B, d, pi_next, reward)
#define STEP_LARGE_K(V, RR)                      \
  switch (reward_kind) {                         \
    case 0: STEP_LARGE(V, RR, 0); break;         \
    case 1: STEP_LARGE(V, RR, 1); break;         \
    default: STEP_LARGE(V, RR, 2); break;        \
  }
    if (vec == 4 && R == 1) { STEP_LARGE_K(4, 1) }
    else if (vec == 4 && R == 2) { STEP_LARGE_K(4, 2) }
    else if (vec == 2 && R == 1) { STEP_LARGE_K(2, 1) }
    else if (vec == 2 && R == 2) { STEP_LARGE_K(2, 2) }
    else if (vec == 2 && R == 3) { STEP_LARGE_K(2, 3) }
    else if (vec == 2 && R == 4) { STEP_LARGE_K(2, 4) }
    else {
      R = (d + WAVE - 1) / WAVE;
      switch (R) {
        case 2: STEP_LARGE_K(1, 2) break;
        case 3: STEP_LARGE_K(1, 3) break;
        case 4: STEP_LARGE_K(1, 4) break;
        case 5: STEP_LARGE_K(1, 5) break;
        case 6: STEP_LARGE_K(1, 6) break;
        case 7: STEP_LARGE_K(1, 7) break;
        case 8: STEP_LARGE_K(1, 8) break;
        default: return fail(MFG_EUNSUPPORTED, "%s: d=%lld > %lld", "step", (long long)d, (long long)MFG_MAX_D);
      }
    }
#undef STEP_LARGE_K
#undef STEP_LARGE
  }
  return check_launch("step_given_P");
}

int mfg_value(const float* pi, const double* w, int64_t B, int d, double* value, mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(pi && w && value, "null pointer");
  hipLaunchKernelGGL(k_value, dim3(grid_for(B, WAVES, 8)), dim3(BLOCK), 0, S(stream), pi, w, B, d, value);
  return check_launch("value");
}

int mfg_features(const float* pi, int64_t B, int d, double* phi, mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(pi && phi, "null pointer");
  hipLaunchKernelGGL(k_features, dim3(grid_for(B * d * d, 256, 8)), dim3(256), 0, S(stream), pi, B, d, phi);
  return check_launch("features");
}

int mfg_backward_value(const float* P, int64_t B, int T, int d, double* V, double* diff_l1, double* diff_jsd,
                       mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(T >= 1, "T < 1");
  REQUIRE(P && V && diff_l1, "null pointer");
  hipLaunchKernelGGL(k_backward_value, dim3(grid_for(B, WAVES, 8)), dim3(BLOCK), (size_t)WAVES * 2 * d * 8, S(stream), P,
                     B, T, d, V, diff_l1, diff_jsd);
  return check_launch("backward_value");
}

int mfg_jsd(const float* p, const float* q, int64_t B, int d, double* out, mfg_stream_t stream) {
  REQUIRE(B >= 0 && d >= 1, "bad shape");
  if (B == 0) return MFG_OK;
  REQUIRE(p && q && out, "null pointer");
  hipLaunchKernelGGL(k_jsd, dim3(grid_for(B, WAVES, 8)), dim3(BLOCK), 0, S(stream), p, q, B, d, out);
  return check_launch("jsd");
}

int mfg_policy_logpdf(const float* pi, const float* P, int64_t N, int d, const double* thetas, int K, double shift,
                      double alpha_scale, double alpha_floor, double p_floor, double* out, mfg_stream_t stream) {
  const int64_t B = N;
  CHECK_BD();
  REQUIRE(pi && P && thetas && out && K >= 1, "null pointer / K < 1");
  hipLaunchKernelGGL(k_policy_logpdf, dim3(grid_for(N * K, WAVES, 8)), dim3(BLOCK), 0, S(stream), pi, P, N, d, thetas, K,
                     shift, alpha_scale, alpha_floor, p_floor, out);
  return check_launch("policy_logpdf");
}

#define CHECK_PRECISION() REQUIRE(precision == MFG_PRECISION_F64 || precision == MFG_PRECISION_MIXED, "bad precision")

int mfg_sample_dirichlet(const float* pi, int64_t B, int d, const double* theta, double shift, double alpha_scale,
                         uint64_t seed, uint32_t step, uint64_t traj_offset, int precision, float* P,
                         mfg_stream_t stream) {
  CHECK_BD();
  CHECK_PRECISION();
  REQUIRE(pi && theta && P, "null pointer");
  CoreArgs a{};
  a.pi0 = pi;
  a.theta = theta;
  a.shift = shift;
  a.alpha_scale = alpha_scale;
  a.gamma = 1.0;
  a.B = B;
  a.d = d;
  a.T = 1;
  a.reward_kind = MFG_REWARD_EXTERNAL;
  a.seed = seed;
  a.first_step = step;
  a.traj_offset = traj_offset;
  a.P_out = P;
  return launch_core(a, true, false, precision, S(stream));
}

int mfg_score(const float* pi_alpha, const float* P, int64_t B, int d, const double* theta, double shift,
              int precision, double* g, mfg_stream_t stream) {
  CHECK_BD();
  CHECK_PRECISION();
  REQUIRE(pi_alpha && P && theta && g, "null pointer");
  CoreArgs a{};
  a.pi0 = pi_alpha;
  a.P_in = P;
  a.theta = theta;
  a.shift = shift;
  a.gamma = 1.0;
  a.B = B;
  a.d = d;
  a.T = 1;
  a.reward_kind = MFG_REWARD_EXTERNAL;
  a.g = g;
  return launch_core(a, false, true, precision, S(stream));
}

int mfg_td_pg_accumulate(const float* pi, const float* pi_next, const float* P, const float* reward, const double* w,
                         const double* theta, double shift, double gamma_or_discount, int64_t B, int d, int precision,
                         double* delta, double* g, double* G, int accumulate, void* workspace,
                         size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_BD();
  CHECK_PRECISION();
  REQUIRE(pi && pi_next && P && reward && w && theta && delta && g, "null pointer");
  CoreArgs a{};
  a.pi0 = pi;
  a.P_in = P;
  a.pi_next_in = pi_next;
  a.reward_in = reward;
  a.theta = theta;
  a.w = w;
  a.shift = shift;
  a.gamma = gamma_or_discount;
  a.B = B;
  a.d = d;
  a.T = 1;
  a.reward_kind = MFG_REWARD_EXTERNAL;
  a.delta = delta;
  a.g = g;
  int rc = launch_core(a, false, true, precision, S(stream));
  if (rc != MFG_OK || !G) return rc;
  REQUIRE(workspace, "workspace is null");
  return launch_grad(pi, d, delta, g, reward, B, 1, d, G, accumulate, workspace, workspace_bytes, S(stream));
}

int mfg_apply_update(const double* G, int d, double lr_critic, double lr_actor, double* w, double* theta,
                     double* reward_acc, mfg_stream_t stream) {
  REQUIRE(G && w && theta && d >= 1, "null pointer");
  const int64_t F = mfg_num_features(d);
  hipLaunchKernelGGL(k_apply_update, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, S(stream), G, F, lr_critic,
                     lr_actor, w, theta, reward_acc);
  return check_launch("apply_update");
}

// After a TD rollout that skipped the in-kernel values (large d): V of all B (T+1) states, then delta.
static int launch_values_and_delta(const float* pi_traj, int64_t B, int T, int d, const double* w, const float* reward,
                                   double gamma, int discount_pow, double* delta, void* ws, size_t ws_bytes, hipStream_t st) {
  const int64_t N = B * T, NV = B * (int64_t)(T + 1);
  int chunk, nob;
  int64_t nsb;
  grad_geometry(N, d, &chunk, &nsb, &nob);
  const size_t off = (size_t)(nsb * (mfg_num_features(d) + 3) * 8) + MFG_WS_CONTROL_BYTES;
  if (!ws || ws_bytes < off + (size_t)NV * 8)
    return fail(MFG_EWORKSPACE, "%s: need %lld bytes, have %lld", "values", (long long)(off + (size_t)NV * 8), (long long)ws_bytes);
  double* V = reinterpret_cast<double*>((char*)ws + off);
  const size_t lds = (size_t)WAVES * 16 * (d + 4) * 4 + (size_t)2 * VM_CH * 16 * 8;  // state rows + two pieces of U
  int64_t groups = (NV + 15) / 16;
  int64_t blocks = (groups + WAVES - 1) / WAVES;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
#ifndef MFG_VALUE_MFMA_OLD
  if (d % 64 == 0) {
    const size_t lds2 = (size_t)VM2_KC * 64 * 8 + (size_t)VM2_MB * VM2_XP * 4;
    int64_t blocks2 = (NV + VM2_MB - 1) / VM2_MB;
    if (blocks2 > (int64_t)num_cus() * 2) blocks2 = (int64_t)num_cus() * 2;
    hipLaunchKernelGGL(k_value_mfma2, dim3((unsigned)blocks2), dim3(BLOCK), lds2, st, pi_traj, (int64_t)(T + 1) * d, NV, T + 1, d, w, V);
  } else
#endif
  hipLaunchKernelGGL(k_value_mfma, dim3((unsigned)blocks), dim3(BLOCK), lds, st, pi_traj, (int64_t)(T + 1) * d, NV, T + 1, d, w, V);
  hipLaunchKernelGGL(k_td_delta, dim3(grid_for(N, 256, 8)), dim3(256), 0, st, (const double*)V, reward, B, T, gamma, discount_pow,
                     delta);
  return check_launch("values");
}
// whether a TD rollout can defer its values to k_value_mfma
static bool defer_values(int d, int64_t B, int T, const float* pi_traj, const void* ws, size_t ws_bytes) {
  if (!value_batch_ok(d) || !pi_traj || !ws || (((uintptr_t)pi_traj & 15) != 0)) return false;
  return ws_bytes >= mfg_workspace_bytes(B * T, d);
}

int mfg_rollout(const float* pi0, int64_t B, int d, int T, const double* theta, double shift, double alpha_scale,
                const double* w, double gamma, int reward_kind, uint64_t seed, uint32_t first_step, uint64_t traj_offset,
                int flags, float* pi_traj, float* pi_last, float* reward, double* delta, double* g, float* P_out,
                double* G, int accumulate, void* workspace, size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(T >= 1, "T < 1");
  REQUIRE(pi0 && theta, "null pointer");
  REQUIRE(reward_kind >= 0 && reward_kind <= 2, "bad reward_kind");
  const bool td = (flags & MFG_ROLLOUT_TD) != 0;
  const bool ext = reward_kind == MFG_REWARD_EXTERNAL;
  REQUIRE(!(flags & MFG_ROLLOUT_WRITE_P) || P_out, "WRITE_P without P_out");
  REQUIRE(!td || (w && delta && g && pi_traj && (reward || ext)), "TD rollout needs w, delta, g, reward, pi_traj");
  REQUIRE(!(ext && td && G), "external reward: the batch sums need the reward, call mfg_grad_accumulate afterwards");
  if (ext) reward = nullptr;
  CoreArgs a{};
  a.pi0 = pi0;
  a.theta = theta;
  a.w = td ? w : nullptr;
  a.shift = shift;
  a.alpha_scale = alpha_scale;
  a.gamma = gamma;
  a.B = B;
  a.d = d;
  a.T = T;
  a.reward_kind = reward_kind;
  a.discount_pow = (flags & MFG_ROLLOUT_DISCOUNT_POW) ? 1 : 0;
  a.seed = seed;
  a.first_step = first_step;
  a.traj_offset = traj_offset;
  a.pi_traj = pi_traj;
  a.pi_next_out = pi_last;
  a.reward_out = reward;
  a.delta = delta;
  a.g = g;
  a.P_out = (flags & MFG_ROLLOUT_WRITE_P) ? P_out : nullptr;
  const int precision = (flags & MFG_ROLLOUT_F64) ? MFG_PRECISION_F64 : MFG_PRECISION_MIXED;
  const bool deferred = td && defer_values(d, B, T, pi_traj, workspace, workspace_bytes);
  if (deferred) a.w = nullptr;  // the kernel then leaves delta alone; values + delta follow on the matrix cores
  const bool sums_in_core = td && G && core_sums_ok(d, B, T, reward_kind, workspace, workspace_bytes);
  if (sums_in_core) a.part_rows = reinterpret_cast<double*>((char*)workspace + MFG_WS_CONTROL_BYTES);
  int rc = launch_core(a, true, td, precision, S(stream));
  if (rc != MFG_OK || !td) return rc;
  if (sums_in_core) return reduce_core_sums(d, B, G, accumulate, workspace, nullptr, S(stream));
  if (deferred) {
    rc = launch_values_and_delta(pi_traj, B, T, d, w, reward, gamma, a.discount_pow, delta, workspace, workspace_bytes, S(stream));
    if (rc != MFG_OK) return rc;
  }
  if (!G) return rc;
  REQUIRE(workspace, "workspace is null");
  return launch_grad(pi_traj, (int64_t)(T + 1) * d, delta, g, reward, B * T, T, d, G, accumulate, workspace,
                     workspace_bytes, S(stream));
}

// learning-rate multipliers of the reference schedule in episode number `episode` (mfg_ac2.py:511-522: 1/(episode+1) and
// 1/((episode+1) ln ln(episode+20)); ac_irl.py:697-708 with its 1-indexed episode); 1, 1 when `constant`.  The same
// double arithmetic as parallel.lr_scales of the host classes (libm log), so native and per-episode runs agree bit for bit.
static void lr_schedule(int64_t episode, int constant, double* sc, double* sa) {
  if (constant) {
    *sc = 1.0;
    *sa = 1.0;
    return;
  }
  const double e1 = (double)(episode + 1);
  *sc = 1.0 / e1;
  *sa = 1.0 / (e1 * log(log((double)(episode + 20))));
}

// the previous update of a multi-rank job, all-reduced but not applied yet (mfg_train_rollout_deferred)
struct DeferredUpdate {
  const double* G;
  double lr_c, lr_a;
  double* reward_acc;
  double *theta_out, *w_out;
};

// one training update per episode: [rollout kernel (start rows: drawn in the kernel when idx == NULL, else gathered) |
// values + delta (large d) | batch sums | row reduction (+ update)]
static int train_rollout_impl(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d, int T, double* theta,
                              double shift, double alpha_scale, double* w, double gamma, int reward_kind, uint64_t seed,
                              uint32_t first_step, uint64_t traj_offset, int flags, double lr_critic, double lr_actor,
                              float* pi_traj, float* pi_last, float* reward, double* delta, double* g, double* G,
                              double* reward_acc, void* workspace, size_t workspace_bytes, hipStream_t st,
                              const DeferredUpdate* du = nullptr) {
  CoreArgs a{};
  if (du) {
    if (d <= WAVE) {
      // packed kernel: the update rides in the weight staging of this rollout
      a.pend_G = du->G;
      a.pend_lr_c = du->lr_c;
      a.pend_lr_a = du->lr_a;
      a.pend_reward_acc = du->reward_acc;
      a.theta_out = du->theta_out;
      a.w_out = du->w_out;
    } else {
      // wave-per-trajectory kernels read the weights from memory throughout: apply the update out of place first
      const int64_t F = mfg_num_features(d);
      hipLaunchKernelGGL(k_apply_update_oop, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, du->G, F, du->lr_c, du->lr_a,
                         (const double*)w, (const double*)theta, du->w_out, du->theta_out, du->reward_acc);
      theta = du->theta_out;
      w = du->w_out;
    }
  }
  a.pi0 = mat_pi0;
  a.start_idx = idx;
  a.start_draw = idx ? 0 : 1;
  a.num_start = num_start;
  a.theta = theta;
  a.w = w;
  a.shift = shift;
  a.alpha_scale = alpha_scale;
  a.gamma = gamma;
  a.B = B;
  a.d = d;
  a.T = T;
  a.reward_kind = reward_kind;
  a.discount_pow = (flags & MFG_ROLLOUT_DISCOUNT_POW) ? 1 : 0;
  a.seed = seed;
  a.first_step = first_step;
  a.traj_offset = traj_offset;
  a.pi_traj = pi_traj;
  a.pi_next_out = pi_last;
  a.reward_out = reward;
  a.delta = delta;
  a.g = g;
  const int precision = (flags & MFG_ROLLOUT_F64) ? MFG_PRECISION_F64 : MFG_PRECISION_MIXED;
  const bool deferred = defer_values(d, B, T, pi_traj, workspace, workspace_bytes);
  if (deferred) a.w = nullptr;
  int rc = launch_core(a, true, true, precision, st);
  if (rc != MFG_OK) return rc;
  if (deferred) {
    rc = launch_values_and_delta(pi_traj, B, T, d, w, reward, gamma, a.discount_pow, delta, workspace, workspace_bytes, st);
    if (rc != MFG_OK) return rc;
  }
  const ApplyArgs ap{lr_critic, lr_actor, w, theta, reward_acc};
  const bool want_apply = (flags & MFG_TRAIN_APPLY) != 0;
  bool applied = false;
  rc = launch_grad(pi_traj, (int64_t)(T + 1) * d, delta, g, reward, B * T, T, d, G, 0, workspace, workspace_bytes, st,
                   want_apply ? &ap : nullptr, &applied);
  if (rc != MFG_OK) return rc;
  if (want_apply && !applied) {
    const int64_t F = mfg_num_features(d);
    hipLaunchKernelGGL(k_apply_update, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, G, F, lr_critic, lr_actor, w, theta,
                       reward_acc);
  }
  return check_launch("train_rollout");
}

#define CHECK_TRAIN_ROLLOUT()                                                                                        \
  CHECK_BD();                                                                                                        \
  REQUIRE(T >= 1, "T < 1");                                                                                          \
  REQUIRE(mat_pi0 && num_start > 0 && num_start <= 0x7FFFFFFF, "null / empty / oversized start-state table");        \
  REQUIRE(theta && w && pi_traj && reward && delta && g && G && workspace, "null pointer");                          \
  REQUIRE(reward_kind == MFG_REWARD_MFG_AC2 || reward_kind == MFG_REWARD_SYNTHETIC, "needs an in-kernel reward")

int mfg_train_rollout(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d, int T, double* theta,
                      double shift, double alpha_scale, double* w, double gamma, int reward_kind, uint64_t seed,
                      uint32_t first_step, uint64_t traj_offset, int flags, double lr_critic, double lr_actor,
                      float* pi_traj, float* pi_last, float* reward, double* delta, double* g, double* G,
                      double* reward_acc, void* workspace, size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_TRAIN_ROLLOUT();
  return train_rollout_impl(mat_pi0, num_start, idx, B, d, T, theta, shift, alpha_scale, w, gamma, reward_kind, seed, first_step,
                            traj_offset, flags, lr_critic, lr_actor, pi_traj, pi_last, reward, delta, g, G, reward_acc, workspace,
                            workspace_bytes, S(stream));
}

int mfg_train_rollouts(const float* mat_pi0, int64_t num_start, int64_t B, int d, int T, int64_t episodes, int64_t first_episode,
                       int constant, double* theta, double shift, double alpha_scale, double* w, double gamma, int reward_kind,
                       uint64_t seed, uint32_t first_step, uint64_t traj_offset, int flags, double lr_critic, double lr_actor,
                       float* pi_traj, float* pi_last, float* reward, double* delta, double* g, double* G, double* reward_acc,
                       void* workspace, size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_TRAIN_ROLLOUT();
  REQUIRE(episodes >= 0 && first_episode >= 0, "bad episode range");
  REQUIRE((uint64_t)first_step + (uint64_t)episodes * (uint64_t)T <= 0xFFFFFFFFull, "Philox step counter would wrap");
  for (int64_t k = 0; k < episodes; ++k) {
    double sc, sa;
    lr_schedule(first_episode + k, constant, &sc, &sa);
    const int rc = train_rollout_impl(mat_pi0, num_start, nullptr, B, d, T, theta, shift, alpha_scale, w, gamma, reward_kind, seed,
                                      first_step + (uint32_t)(k * T), traj_offset, flags | MFG_TRAIN_APPLY, lr_critic * sc,
                                      lr_actor * sa, pi_traj, pi_last, reward, delta, g, G, reward_acc ? reward_acc + k : nullptr,
                                      workspace, workspace_bytes, S(stream));
    if (rc != MFG_OK) return rc;
  }
  return MFG_OK;
}

int mfg_train_rollout_deferred(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d, int T,
                               const double* theta, const double* w, const double* G_pending, double lr_critic_pending,
                               double lr_actor_pending, double* reward_acc_pending, double* theta_out, double* w_out, double shift,
                               double alpha_scale, double gamma, int reward_kind, uint64_t seed, uint32_t first_step,
                               uint64_t traj_offset, int flags, float* pi_traj, float* pi_last, float* reward, double* delta,
                               double* g, double* G, void* workspace, size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_TRAIN_ROLLOUT();
  REQUIRE(!(flags & MFG_TRAIN_APPLY), "deferred update: MFG_TRAIN_APPLY makes no sense here");
  REQUIRE(!G_pending || (theta_out && w_out && theta_out != theta && w_out != w), "pending update needs separate output parameters");
  const DeferredUpdate du{G_pending, lr_critic_pending, lr_actor_pending, reward_acc_pending, theta_out, w_out};
  return train_rollout_impl(mat_pi0, num_start, idx, B, d, T, const_cast<double*>(theta), shift, alpha_scale, const_cast<double*>(w),
                            gamma, reward_kind, seed, first_step, traj_offset, flags, 0.0, 0.0, pi_traj, pi_last, reward, delta, g,
                            G, nullptr, workspace, workspace_bytes, S(stream), G_pending ? &du : nullptr);
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU episode loop, native: RCCL called from this library (one all-reduce of G per update).
// The RCCL entry points are resolved at run time from the librccl.so the process already holds (PyTorch's, so that one RCCL
// instance serves the job); the library has no link-time dependency on it and reports MFG_EUNSUPPORTED where it is absent.
// ---------------------------------------------------------------------------------------------
}  // extern "C" (the helpers below have C++ linkage)
namespace {
struct RcclApi {
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, mfg_rccl_id_t, int) = nullptr;   // ncclUniqueId is passed BY VALUE (128 bytes)
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
const RcclApi& rccl() {
  static RcclApi api = [] {
    RcclApi a;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);          // already in the process (torch)?
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);        // a stand-alone host program: load the system's
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return a;
    a.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<int (*)(void**, int, mfg_rccl_id_t, int)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
    a.CommAbort = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommAbort"));
    a.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce;
    return a;
  }();
  return api;
}
int rccl_fail(const char* what, int rc) {
  const RcclApi& r = rccl();
  return fail(MFG_ELAUNCH, "%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
}
constexpr int RCCL_SUM = 0, RCCL_FLOAT64 = 8;  // ncclSum, ncclFloat64 (rccl.h)
}  // namespace
extern "C" {

int mfg_dist_available(void) { return rccl().ok ? 1 : 0; }

int mfg_dist_unique_id(mfg_rccl_id_t* id_host) {
  REQUIRE(id_host, "null pointer");
  const RcclApi& r = rccl();
  if (!r.ok) return fail(MFG_EUNSUPPORTED, "%s", "librccl.so is not available in this process");
  const int rc = r.GetUniqueId(id_host);
  return rc == 0 ? MFG_OK : rccl_fail("ncclGetUniqueId", rc);
}

int mfg_dist_init(const mfg_rccl_id_t* id_host, int nranks, int rank, void** comm_out) {
  REQUIRE(id_host && comm_out && nranks >= 1 && rank >= 0 && rank < nranks, "bad arguments");
  const RcclApi& r = rccl();
  if (!r.ok) return fail(MFG_EUNSUPPORTED, "%s", "librccl.so is not available in this process");
  void* comm = nullptr;
  const int rc = r.CommInitRank(&comm, nranks, *id_host, rank);   // collective: every rank of the job calls it (current device)
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  *comm_out = comm;
  return MFG_OK;
}

int mfg_dist_destroy(void* comm) {
  if (!comm) return MFG_OK;
  const RcclApi& r = rccl();
  if (!r.ok) return fail(MFG_EUNSUPPORTED, "%s", "librccl.so is not available in this process");
  const int rc = r.CommDestroy(comm);
  return rc == 0 ? MFG_OK : rccl_fail("ncclCommDestroy", rc);
}

int mfg_dist_abort(void* comm) {
  if (!comm) return MFG_OK;
  const RcclApi& r = rccl();
  if (!r.ok || !r.CommAbort) return fail(MFG_EUNSUPPORTED, "%s", "ncclCommAbort is not available in this process");
  const int rc = r.CommAbort(comm);
  return rc == 0 ? MFG_OK : rccl_fail("ncclCommAbort", rc);
}

int mfg_dist_all_reduce(void* comm, double* G, int64_t n, mfg_stream_t stream) {
  REQUIRE(comm && G && n >= 1, "bad arguments");
  const RcclApi& r = rccl();
  if (!r.ok) return fail(MFG_EUNSUPPORTED, "%s", "librccl.so is not available in this process");
  const int rc = r.AllReduce(G, G, (size_t)n, RCCL_FLOAT64, RCCL_SUM, comm, S(stream));
  return rc == 0 ? MFG_OK : rccl_fail("ncclAllReduce", rc);
}

int mfg_train_rollouts_dist(void* comm, const float* mat_pi0, int64_t num_start, int64_t B, int d, int T, int64_t episodes,
                            int64_t first_episode, int constant, double* theta, double* w, double* theta_alt, double* w_alt,
                            double shift, double alpha_scale, double gamma, int reward_kind, uint64_t seed, uint32_t first_step,
                            uint64_t traj_offset, int flags, double lr_critic, double lr_actor, float* pi_traj, float* pi_last,
                            float* reward, double* delta, double* g, double* G, double* reward_acc, void* workspace,
                            size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_TRAIN_ROLLOUT();
  REQUIRE(comm && theta_alt && w_alt && theta_alt != theta && w_alt != w, "needs a communicator and a second parameter set");
  REQUIRE(episodes >= 0 && first_episode >= 0, "bad episode range");
  REQUIRE(!(flags & MFG_TRAIN_APPLY), "MFG_TRAIN_APPLY makes no sense here");
  REQUIRE((uint64_t)first_step + (uint64_t)episodes * (uint64_t)T <= 0xFFFFFFFFull, "Philox step counter would wrap");
  const RcclApi& r = rccl();
  if (!r.ok) return fail(MFG_EUNSUPPORTED, "%s", "librccl.so is not available in this process");
  if (episodes == 0) return MFG_OK;
  const int64_t F = mfg_num_features(d);
  hipStream_t st = S(stream);
  // Every rank must enqueue the SAME number of all-reduces.  The sticky status word is therefore examined here, before the
  // first enqueue (a condition raised by an earlier call: every rank that shares the history refuses alike), and again
  // behind the loop; inside the loop launches go ahead whatever the device reports meanwhile (the outputs of a launch in the
  // reported condition are NaN on every rank alike, theta is replicated).  A rank-local failure inside the loop (a launch
  // error) aborts the communicator, so the peers' pending collectives fail instead of waiting for this rank for ever.
  {
    const bool mixed = !(flags & MFG_ROLLOUT_F64);
    unsigned bits = 0;
    if (mfg_status(&bits) != MFG_OK && mixed && (bits & MFG_STATUS_MIXED_RANGE)) return MFG_ERANGE;
  }
  struct DeferGuard {
    DeferGuard() { g_status_check_deferred = true; }
    ~DeferGuard() { g_status_check_deferred = false; }
  } defer_guard;
  double *tc = theta, *wc = w, *tn = theta_alt, *wn = w_alt;   // current / next parameter set
  double plc = 0.0, pla = 0.0;
  double* pacc = nullptr;
  bool pending = false;
  for (int64_t k = 0; k < episodes; ++k) {
    const DeferredUpdate du{G, plc, pla, pacc, tn, wn};
    int rc = train_rollout_impl(mat_pi0, num_start, nullptr, B, d, T, tc, shift, alpha_scale, wc, gamma, reward_kind, seed,
                                first_step + (uint32_t)(k * T), traj_offset, flags, 0.0, 0.0, pi_traj, pi_last, reward, delta, g, G,
                                nullptr, workspace, workspace_bytes, st, pending ? &du : nullptr);
    if (rc != MFG_OK) {
      // (the communicator is dead afterwards: MFG_ECOMM tells the caller to forget the handle; mfg_last_error keeps the cause)
      if (r.CommAbort) {
        (void)r.CommAbort(comm);
        return MFG_ECOMM;
      }
      return rc;
    }
    if (pending) {  // the rollout left the updated parameters in the other set
      double* t = tc; tc = tn; tn = t;
      t = wc; wc = wn; wn = t;
    }
    rc = r.AllReduce(G, G, (size_t)(F + 3), RCCL_FLOAT64, RCCL_SUM, comm, st);   // the ONE exchange of the update
    if (rc != 0) {
      (void)rccl_fail("ncclAllReduce", rc);
      if (r.CommAbort) {
        (void)r.CommAbort(comm);
        return MFG_ECOMM;
      }
      return MFG_ELAUNCH;
    }
    double sc, sa;
    lr_schedule(first_episode + k, constant, &sc, &sa);
    plc = lr_critic * sc;
    pla = lr_actor * sa;
    pacc = reward_acc ? reward_acc + k : nullptr;
    pending = true;
  }
  // the last update, and the parameters back in the caller's primary set
  hipLaunchKernelGGL(k_apply_update, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, G, F, plc, pla, wc, tc, pacc);
  if (tc != theta) {
    if (hipMemcpyAsync(theta, tc, sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(w, wc, (size_t)F * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(MFG_ELAUNCH, "%s", "train_rollouts_dist: parameter copy failed");
  }
  const int lrc = check_launch("train_rollouts_dist");
  if (lrc != MFG_OK) return lrc;
  if (!(flags & MFG_ROLLOUT_F64)) {   // what the device has reported so far (everything is enqueued: symmetric by construction)
    unsigned bits = 0;
    if (mfg_status(&bits) != MFG_OK && (bits & MFG_STATUS_MIXED_RANGE)) return MFG_ERANGE;
  }
  return MFG_OK;
}

int mfg_grad_accumulate(const float* pi, int64_t stride_b, double* delta, const double* g, const float* reward, int64_t B,
                        int T, int d, int add_reward, double* G, int accumulate, void* workspace, size_t workspace_bytes,
                        mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(T >= 1 && stride_b >= (int64_t)T * d, "bad T / stride_b");
  REQUIRE(pi && delta && G && workspace, "null pointer");
  REQUIRE(!add_reward || reward, "add_reward without reward");
  return launch_grad(pi, stride_b, delta, g, reward, B * T, T, d, G, accumulate, workspace, workspace_bytes, S(stream),
                     nullptr, nullptr, add_reward != 0);
}

int mfg_grad_apply(const float* pi, int64_t stride_b, double* delta, const double* g, const float* reward, int64_t B, int T,
                   int d, int add_reward, double* G, double lr_critic, double lr_actor, double* w, double* theta,
                   double* reward_acc, void* workspace, size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(T >= 1 && stride_b >= (int64_t)T * d, "bad T / stride_b");
  REQUIRE(pi && delta && G && workspace && w && theta, "null pointer");
  REQUIRE(!add_reward || reward, "add_reward without reward");
  const ApplyArgs ap{lr_critic, lr_actor, w, theta, reward_acc};
  bool applied = false;
  int rc = launch_grad(pi, stride_b, delta, g, reward, B * T, T, d, G, 0, workspace, workspace_bytes, S(stream), &ap, &applied,
                       add_reward != 0);
  if (rc != MFG_OK) return rc;
  if (!applied) {
    const int64_t F = mfg_num_features(d);
    hipLaunchKernelGGL(k_apply_update, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, S(stream), G, F, lr_critic, lr_actor,
                       w, theta, reward_acc);
  }
  return check_launch("grad_apply");
}

static int train_episode_impl(float* pi_io, float* pi_scratch, int64_t B, int d, int T, double* theta, double shift,
                              double alpha_scale, double* w, double gamma, int reward_kind, uint64_t seed, uint32_t first_step,
                              uint64_t traj_offset, int precision, double lr_critic, double lr_actor, float* reward,
                              double* delta, double* g, double* G, double* reward_acc, void* workspace, size_t workspace_bytes,
                              hipStream_t st) {
  float* cur = pi_io;
  float* nxt = pi_scratch;
  for (int s = 0; s < T; ++s) {
    CoreArgs a{};
    a.pi0 = cur;
    a.theta = theta;
    a.w = w;
    a.shift = shift;
    a.alpha_scale = alpha_scale;
    a.gamma = gamma;
    a.B = B;
    a.d = d;
    a.T = 1;
    a.reward_kind = reward_kind;
    a.seed = seed;
    a.first_step = first_step + (uint32_t)s;
    a.traj_offset = traj_offset;
    a.pi_next_out = nxt;
    a.reward_out = reward;
    a.delta = delta;
    a.g = g;
    const bool sums_in_core = core_sums_ok(d, B, 1, reward_kind, workspace, workspace_bytes);
    if (sums_in_core) a.part_rows = reinterpret_cast<double*>((char*)workspace + MFG_WS_CONTROL_BYTES);
    int rc = launch_core(a, true, true, precision, st);
    if (rc != MFG_OK) return rc;
    const ApplyArgs ap{lr_critic, lr_actor, w, theta, reward_acc};
    bool applied = false;
    if (sums_in_core) {
      rc = reduce_core_sums(d, B, G, 0, workspace, &ap, st);
      applied = true;
    } else {
      rc = launch_grad(cur, d, delta, g, reward, B, 1, d, G, 0, workspace, workspace_bytes, st, &ap, &applied);
    }
    if (rc != MFG_OK) return rc;
    if (!applied) {
      const int64_t F = mfg_num_features(d);
      hipLaunchKernelGGL(k_apply_update, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, G, F, lr_critic, lr_actor,
                         w, theta, reward_acc);
    }
    float* t = cur;
    cur = nxt;
    nxt = t;
  }
  if (cur != pi_io && hipMemcpyAsync(pi_io, cur, (size_t)B * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail(MFG_ELAUNCH, "%s", "train_episode: final state copy failed");
  return check_launch("train_episode");
}


#define CHECK_TRAIN_EPISODE()                                                                                      \
  CHECK_BD();                                                                                                      \
  CHECK_PRECISION();                                                                                               \
  REQUIRE(T >= 1, "T < 1");                                                                                        \
  REQUIRE(pi_io && pi_scratch && theta && w && reward && delta && g && G && workspace, "null pointer");            \
  REQUIRE(reward_kind == MFG_REWARD_MFG_AC2 || reward_kind == MFG_REWARD_SYNTHETIC, "needs an in-kernel reward")

int mfg_train_episode(float* pi_io, float* pi_scratch, int64_t B, int d, int T, double* theta, double shift,
                      double alpha_scale, double* w, double gamma, int reward_kind, uint64_t seed, uint32_t first_step,
                      uint64_t traj_offset, int precision, double lr_critic, double lr_actor, float* reward, double* delta,
                      double* g, double* G, double* reward_acc, void* workspace, size_t workspace_bytes,
                      mfg_stream_t stream) {
  CHECK_TRAIN_EPISODE();
  return train_episode_impl(pi_io, pi_scratch, B, d, T, theta, shift, alpha_scale, w, gamma, reward_kind, seed, first_step,
                            traj_offset, precision, lr_critic, lr_actor, reward, delta, g, G, reward_acc, workspace,
                            workspace_bytes, S(stream));
}

int mfg_train_episodes(const float* mat_pi0, int64_t num_start, float* pi_io, float* pi_scratch, int64_t B, int d, int T,
                       int64_t episodes, int64_t first_episode, int constant, double* theta, double shift, double alpha_scale,
                       double* w, double gamma, int reward_kind, uint64_t seed, uint32_t first_step, uint64_t traj_offset,
                       int precision, double lr_critic, double lr_actor, float* reward, double* delta, double* g, double* G,
                       double* reward_acc, void* workspace, size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_TRAIN_EPISODE();
  REQUIRE(mat_pi0 && num_start > 0 && num_start <= 0x7FFFFFFF, "null / empty / oversized start-state table");
  REQUIRE(episodes >= 0 && first_episode >= 0, "bad episode range");
  REQUIRE((uint64_t)first_step + (uint64_t)episodes * (uint64_t)T <= 0xFFFFFFFFull, "Philox step counter would wrap");
  for (int64_t k = 0; k < episodes; ++k) {
    const uint32_t step0 = first_step + (uint32_t)(k * T);
    hipLaunchKernelGGL(k_draw_start, dim3(grid_for(B * d, 256, 8)), dim3(256), 0, S(stream), mat_pi0, num_start, B, d, seed, step0,
                       traj_offset, (int32_t*)nullptr, pi_io);
    double sc, sa;
    lr_schedule(first_episode + k, constant, &sc, &sa);
    const int rc = train_episode_impl(pi_io, pi_scratch, B, d, T, theta, shift, alpha_scale, w, gamma, reward_kind, seed, step0,
                                      traj_offset, precision, lr_critic * sc, lr_actor * sa, reward, delta, g, G,
                                      reward_acc ? reward_acc + k : nullptr, workspace, workspace_bytes, S(stream));
    if (rc != MFG_OK) return rc;
  }
  return MFG_OK;
}

int mfg_train_rollout_irl(const float* mat_pi0, int64_t num_start, const int32_t* idx, int64_t B, int d, int T, double* theta,
                          double shift, double alpha_scale, double* w, double gamma, uint64_t seed, uint32_t first_step,
                          uint64_t traj_offset, int flags, double lr_critic, double lr_actor, const mfg_reward_net_t* net,
                          uint64_t rn_key, uint64_t rn_sample_offset, float* pi_traj, float* pi_last, float* P, float* reward,
                          double* delta, double* g, double* G, double* reward_acc, void* workspace, size_t workspace_bytes,
                          mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(T >= 1, "T < 1");
  REQUIRE(mat_pi0 && num_start > 0 && num_start <= 0x7FFFFFFF, "null / empty / oversized start-state table");
  REQUIRE(theta && w && net && pi_traj && P && reward && delta && g && G && workspace, "null pointer");
  REQUIRE(B * (int64_t)T <= 0x7FFFFFFF, "B * T too large");
  hipStream_t st = S(stream);
  CoreArgs a{};
  a.pi0 = mat_pi0;
  a.start_idx = idx;
  a.start_draw = idx ? 0 : 1;
  a.num_start = num_start;
  a.theta = theta;
  a.w = w;
  a.shift = shift;
  a.alpha_scale = alpha_scale;
  a.gamma = gamma;
  a.B = B;
  a.d = d;
  a.T = T;
  a.reward_kind = MFG_REWARD_EXTERNAL;  // delta = discount V(pi') - V(pi); the reward joins it in the gradient kernel
  a.discount_pow = (flags & MFG_ROLLOUT_DISCOUNT_POW) ? 1 : 0;
  a.seed = seed;
  a.first_step = first_step;
  a.traj_offset = traj_offset;
  a.pi_traj = pi_traj;
  a.pi_next_out = pi_last;
  a.delta = delta;
  a.g = g;
  a.P_out = P;
  const int precision = (flags & MFG_ROLLOUT_F64) ? MFG_PRECISION_F64 : MFG_PRECISION_MIXED;
  int rc = launch_core(a, true, true, precision, st);
  if (rc != MFG_OK) return rc;
  // ONE reward-network pass over all B*T transitions; the states are read in place from pi_traj (rows b (T+1) + t)
  rc = reward_net_forward_sums(pi_traj, P, B * (int64_t)T, d, net->k1, net->f2, net->k2, net->n3, net->n4, net->conv1_w, net->conv1_b,
                               net->conv2_w, net->conv2_b, net->fc3_w, net->fc3_b, net->fc4_w, net->fc4_b, net->out_w, net->out_b,
                               net->keep_prob, rn_key, rn_sample_offset, reward, nullptr, nullptr, stream, T);
  if (rc != MFG_OK) return rc;
  const ApplyArgs ap{lr_critic, lr_actor, w, theta, reward_acc};
  const bool want_apply = (flags & MFG_TRAIN_APPLY) != 0;
  bool applied = false;
  rc = launch_grad(pi_traj, (int64_t)(T + 1) * d, delta, g, reward, B * T, T, d, G, 0, workspace, workspace_bytes, st,
                   want_apply ? &ap : nullptr, &applied, true);
  if (rc != MFG_OK) return rc;
  if (want_apply && !applied) {
    const int64_t F = mfg_num_features(d);
    hipLaunchKernelGGL(k_apply_update, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, G, F, lr_critic, lr_actor, w, theta,
                       reward_acc);
  }
  return check_launch("train_rollout_irl");
}

}  // extern "C"
// mat_pi0 != NULL: the start states are DRAWN from the table [num_start,d] (the draw of mfg_draw_start at step = first_step) --
// inside the first step kernel where the two-launch flow serves -- and pi_io is an output only
static int train_episode_irl_impl(const float* mat_pi0, int64_t num_start, float* pi_io, float* pi_scratch, int64_t B, int d, int T,
                                  double* theta, double shift, double alpha_scale, double* w, double gamma, uint64_t seed,
                                  uint32_t first_step, uint64_t traj_offset, int precision, double lr_critic, double lr_actor,
                                  const mfg_reward_net_t* net, uint64_t rn_seed, uint64_t rn_call0, uint64_t rn_sample_offset,
                                  float* P, float* reward, double* delta, double* g, double* G, double* reward_acc, void* workspace,
                                  size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_BD();
  CHECK_PRECISION();
  REQUIRE(T >= 1, "T < 1");
  REQUIRE(pi_io && pi_scratch && theta && w && net && P && reward && delta && g && G && workspace, "null pointer");
  REQUIRE(!mat_pi0 || (num_start > 0 && num_start <= 0x7FFFFFFF), "empty / oversized start-state table");
  hipStream_t st = S(stream);
  float* cur = pi_io;
  float* nxt = pi_scratch;
  double discount = 1.0;  // running gamma^t of ac_irl.py:691, :710
  {
    // Two launches per env step where the matrix-core reward-network kernel serves (d = 21 / 15, n_fc3 <= 16):
    //   step kernel (STEP variant): sampling + transition + score with theta formed from the PREVIOUS step's partial rows by
    //     every wave; the grid's last blocks reduce those rows and publish w, theta, G, the return;
    //   reward network: r, the TD error delta = r + discount V(pi') - V(pi) from the updated w, this step's partial rows.
    // The row reduction -- a launch of its own between two dependent launches before -- leaves the critical path.
    const int64_t FO = mfg_num_features(d) + 3;
    // workspace: control block | column F of the rows, contiguous [nrows] | the rows [nrows][FO]   (nrows <= 256: one per block)
    const int64_t max_rows = (B + 15) / 16 < 256 ? (B + 15) / 16 : 256;
    const int64_t room = workspace_bytes >= MFG_WS_CONTROL_BYTES + (size_t)max_rows * (FO + 1) * 8 ? max_rows : 0;
    if (d <= WAVE && reward_net_sums_td_ready(B, d, net->k1, net->f2, net->k2, net->n3, net->n4, net->fc3_w, room)) {
      double* rows_buf = reinterpret_cast<double*>((char*)workspace + MFG_WS_CONTROL_BYTES) + max_rows;
      double* th_slot = reinterpret_cast<double*>((char*)workspace + 16);  // two slots: theta after odd / even steps
      const double* th_in = theta;
      int nrows = 0;
      if (mat_pi0 && (T & 1)) {  // drawn start states: the buffers alternate so that the LAST step writes pi_io (no copy)
        cur = pi_scratch;
        nxt = pi_io;
      }
      for (int s = 0; s < T; ++s) {
        CoreArgs a{};
        a.pi0 = cur;
        if (s == 0 && mat_pi0) {  // the first step kernel draws its start states itself and leaves them in `cur` for the network
          a.pi0 = mat_pi0;
          a.num_start = num_start;
          a.start_draw = 1;
          a.pi_start_out = cur;
          a.step_nrows = -1;
        }
        a.theta = th_in;
        a.w = nullptr;  // no value part here
        a.shift = shift;
        a.alpha_scale = alpha_scale;
        a.gamma = discount;
        a.B = B;
        a.d = d;
        a.T = 1;
        a.reward_kind = MFG_REWARD_EXTERNAL;
        a.seed = seed;
        a.first_step = first_step + (uint32_t)s;
        a.traj_offset = traj_offset;
        a.pi_next_out = nxt;
        a.g = g;
        a.P_out = P;
        if (s > 0) {  // (the first step has nothing to reduce: the plain kernel, without its value part)
          a.step_G = G;
          a.pend_lr_c = lr_critic;
          a.pend_lr_a = lr_actor;
          a.w_out = w;
          a.pend_reward_acc = reward_acc;
          a.step_rows = rows_buf;
          a.step_nrows = nrows;
          a.theta_out = th_slot + (s & 1);
        }
        int rc = launch_core(a, true, true, precision, st);
        if (rc != MFG_OK) return rc;
        if (s > 0) th_in = th_slot + (s & 1);
        const uint64_t key = rn_seed ^ ((rn_call0 + (uint64_t)s + 1ull) * 0x9E3779B97F4A7C15ull);
        RnSums sm{};
        sm.g = g;
        sm.delta_out = delta;
        sm.part_rows = rows_buf;
        sm.max_rows = room;
        sm.td_w = w;
        sm.state_next = nxt;
        sm.td_gamma = discount;
        sm.col_f = rows_buf - max_rows;
        int rows = 0;
        rc = reward_net_forward_sums(cur, P, B, d, net->k1, net->f2, net->k2, net->n3, net->n4, net->conv1_w, net->conv1_b,
                                     net->conv2_w, net->conv2_b, net->fc3_w, net->fc3_b, net->fc4_w, net->fc4_b, net->out_w,
                                     net->out_b, net->keep_prob, key, rn_sample_offset, reward, &sm, &rows, stream);
        if (rc != MFG_OK) return rc;
        if (rows != (int)max_rows) return fail(MFG_ELAUNCH, "%s", "train_episode_irl: the reward-network launch left no partial rows");
        nrows = rows;
        discount *= gamma;
        float* t = cur;
        cur = nxt;
        nxt = t;
      }
      hipLaunchKernelGGL(k_reduce_rows_apply, dim3((unsigned)((FO + WAVES - 1) / WAVES)), dim3(BLOCK), 0, st, (const double*)rows_buf,
                         nrows, FO, G, lr_critic, lr_actor, (double)B, w, th_in, theta, reward_acc);
      if (cur != pi_io && hipMemcpyAsync(pi_io, cur, (size_t)B * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return fail(MFG_ELAUNCH, "%s", "train_episode_irl: final state copy failed");
      return check_launch("train_episode_irl");
    }
  }
  if (mat_pi0) {
    hipLaunchKernelGGL(k_draw_start, dim3(grid_for(B * d, 256, 8)), dim3(256), 0, st, mat_pi0, num_start, B, d, seed, first_step,
                       traj_offset, (int32_t*)nullptr, pi_io);
    const int rc = check_launch("draw_start");
    if (rc != MFG_OK) return rc;
  }
  for (int s = 0; s < T; ++s) {
    CoreArgs a{};
    a.pi0 = cur;
    a.theta = theta;
    a.w = w;
    a.shift = shift;
    a.alpha_scale = alpha_scale;
    a.gamma = discount;
    a.B = B;
    a.d = d;
    a.T = 1;
    a.reward_kind = MFG_REWARD_EXTERNAL;  // delta = discount V(pi') - V(pi); the reward joins it in the gradient kernel
    a.seed = seed;
    a.first_step = first_step + (uint32_t)s;
    a.traj_offset = traj_offset;
    a.pi_next_out = nxt;
    a.delta = delta;
    a.g = g;
    a.P_out = P;
    int rc = launch_core(a, true, true, precision, st);
    if (rc != MFG_OK) return rc;
    const uint64_t key = rn_seed ^ ((rn_call0 + (uint64_t)s + 1ull) * 0x9E3779B97F4A7C15ull);
    // reward network; at the packed sizes the same launch folds delta = delta0 + r and leaves the partial rows of the batch
    // sums (one per block of eight samples), so the update is a row reduction instead of a gradient kernel
    const int64_t FO = mfg_num_features(d) + 3;
    const int64_t room = workspace_bytes > MFG_WS_CONTROL_BYTES ? (int64_t)((workspace_bytes - MFG_WS_CONTROL_BYTES) / (size_t)(FO * 8)) : 0;
    const RnSums sm{delta, g, delta, reinterpret_cast<double*>((char*)workspace + MFG_WS_CONTROL_BYTES), room};
    int rows = 0;
    rc = reward_net_forward_sums(cur, P, B, d, net->k1, net->f2, net->k2, net->n3, net->n4, net->conv1_w, net->conv1_b,
                                 net->conv2_w, net->conv2_b, net->fc3_w, net->fc3_b, net->fc4_w, net->fc4_b, net->out_w,
                                 net->out_b, net->keep_prob, key, rn_sample_offset, reward, &sm, &rows, stream);
    if (rc != MFG_OK) return rc;
    const ApplyArgs ap{lr_critic, lr_actor, w, theta, reward_acc};
    bool applied = false;
    if (rows > 0) {
      ReduceApply rap{};
      rap.on = 1;
      rap.lr_c = lr_critic;
      rap.lr_a = lr_actor;
      rap.count = (double)B;
      rap.w = w;
      rap.theta = theta;
      rap.reward_acc = reward_acc;
      hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((FO + RP_OUT - 1) / RP_OUT)), dim3(RP_SLICES * RP_OUT), 0, st,
                         (const double*)sm.part_rows, (int64_t)rows, FO, 0, G, rap);
      applied = true;
      rc = check_launch("irl_sums");
    } else {
      rc = launch_grad(cur, d, delta, g, reward, B, 1, d, G, 0, workspace, workspace_bytes, st, &ap, &applied, true);
    }
    if (rc != MFG_OK) return rc;
    if (!applied) {
      const int64_t F = mfg_num_features(d);
      hipLaunchKernelGGL(k_apply_update, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, G, F, lr_critic, lr_actor,
                         w, theta, reward_acc);
    }
    discount *= gamma;
    float* t = cur;
    cur = nxt;
    nxt = t;
  }
  if (cur != pi_io && hipMemcpyAsync(pi_io, cur, (size_t)B * d * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail(MFG_ELAUNCH, "%s", "train_episode_irl: final state copy failed");
  return check_launch("train_episode_irl");
}

extern "C" {
int mfg_train_episode_irl(float* pi_io, float* pi_scratch, int64_t B, int d, int T, double* theta, double shift,
                          double alpha_scale, double* w, double gamma, uint64_t seed, uint32_t first_step,
                          uint64_t traj_offset, int precision, double lr_critic, double lr_actor,
                          const mfg_reward_net_t* net, uint64_t rn_seed, uint64_t rn_call0, uint64_t rn_sample_offset,
                          float* P, float* reward, double* delta, double* g, double* G, double* reward_acc, void* workspace,
                          size_t workspace_bytes, mfg_stream_t stream) {
  return train_episode_irl_impl(nullptr, 0, pi_io, pi_scratch, B, d, T, theta, shift, alpha_scale, w, gamma, seed, first_step,
                                traj_offset, precision, lr_critic, lr_actor, net, rn_seed, rn_call0, rn_sample_offset, P, reward,
                                delta, g, G, reward_acc, workspace, workspace_bytes, stream);
}

int mfg_train_episode_irl_draw(const float* mat_pi0, int64_t num_start, float* pi_out, float* pi_scratch, int64_t B, int d, int T,
                               double* theta, double shift, double alpha_scale, double* w, double gamma, uint64_t seed,
                               uint32_t first_step, uint64_t traj_offset, int precision, double lr_critic, double lr_actor,
                               const mfg_reward_net_t* net, uint64_t rn_seed, uint64_t rn_call0, uint64_t rn_sample_offset,
                               float* P, float* reward, double* delta, double* g, double* G, double* reward_acc, void* workspace,
                               size_t workspace_bytes, mfg_stream_t stream) {
  if (!mat_pi0) return fail(MFG_EINVAL, "%s", "train_episode_irl_draw: null start-state table");
  return train_episode_irl_impl(mat_pi0, num_start, pi_out, pi_scratch, B, d, T, theta, shift, alpha_scale, w, gamma, seed,
                                first_step, traj_offset, precision, lr_critic, lr_actor, net, rn_seed, rn_call0, rn_sample_offset,
                                P, reward, delta, g, G, reward_acc, workspace, workspace_bytes, stream);
}

}  // extern "C"
