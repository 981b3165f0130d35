// HIP kernels + C ABI of the MFG hot path for MI355X (gfx950 / CDNA4, wave64).
// See include/mfg_hip.h for the contract and DESIGN.md for the layout / roofline notes.
//
// Two work decompositions:
//   small d (d <= 64): G = 64/d trajectories packed per wavefront; the d x d action matrix of each
//       trajectory is staged in LDS per wavefront.  HBM-bound step kernel: lane = (trajectory, column j).
//       Compute-bound sampler / TD kernels: lane = (trajectory, row i) so that all per-row Dirichlet
//       quantities (row sum of gamma variates, sum_j alpha_ij, ...) stay lane-local.
//   large d (d > 64): one wavefront per trajectory, lanes own columns, rows are streamed from HBM
//       with coalesced loads; per-row quantities use wavefront reductions.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mfg_hip.h"
#include "mfg_device.h"

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

static int num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / WAVE;

// ---------------------------------------------------------------------------------------------
// trivial kernels: gather, alpha, features, dirichlet_from_gamma, philox_raw, apply_update
// ---------------------------------------------------------------------------------------------
__global__ void k_gather_start(const float* __restrict__ mat, const int32_t* __restrict__ idx, int64_t B, int d,
                               float* __restrict__ out) {
  const int64_t n = B * d;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / d;
    const int j = (int)(e - b * d);
    out[e] = mat[(int64_t)idx[b] * d + j];
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
                               double* __restrict__ theta) {
  const double count = G[F + 2];
  if (!(count > 0.0)) return;
  const double inv = 1.0 / count;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < F; k += (int64_t)gridDim.x * blockDim.x)
    w[k] += lr_c * (G[k] * inv);
  if (blockIdx.x == 0 && threadIdx.x == 0) *theta += lr_a * (G[F] * inv);
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
// a5: V(pi) = phi(pi).w, one wavefront per trajectory, lanes own columns c, rows i <= c.
// w rows are contiguous in k for fixed i, so the loads are coalesced (L2 resident).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double value_wave(const float* pis /*LDS or global, d floats*/, const double* __restrict__ w,
                                             int d, int lane) {
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
// (contiguous in HBM) is copied flat with 16-byte loads into LDS; lane = (trajectory, column j)
// accumulates pi'_j and the column's share of the reward in fp64, one segmented reduction at the end.
// ---------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_step_small(const float* __restrict__ pi, const float* __restrict__ P,
                                                      int64_t B, int d, int vec_ok, float* __restrict__ pi_next,
                                                      float* __restrict__ reward) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int G = WAVE / d, TB = WAVES * G, dd = d * d;
  float* tP = smem;                 // [TB][d][d]
  float* tPi = smem + TB * dd;      // [TB][d]
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int t = lane / d, j = lane - t * d;
  const int p2 = next_pow2(d);
  const int64_t ntiles = (B + TB - 1) / TB;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * TB;
    const int nb = (int)((B - b0) < TB ? (B - b0) : TB);
    const int n = nb * dd;
    const float* src = P + b0 * dd;
    if (vec_ok) {
      const int n4 = n >> 2;
      const float4* s4 = reinterpret_cast<const float4*>(src);
      float4* d4 = reinterpret_cast<float4*>(tP);
      for (int k = tid; k < n4; k += BLOCK) d4[k] = s4[k];
      for (int k = (n4 << 2) + tid; k < n; k += BLOCK) tP[k] = src[k];
    } else {
      for (int k = tid; k < n; k += BLOCK) tP[k] = src[k];
    }
    for (int k = tid; k < nb * d; k += BLOCK) tPi[k] = pi[b0 * d + k];
    __syncthreads();
    const int tl = wv * G + t;
    const bool valid = (t < G) && (tl < nb);
    const int tlc = valid ? tl : 0;
    const float* rowp = tP + tlc * dd + j;
    const float* pv = tPi + tlc * d;
    const double pj = (double)pv[j];
    double acc = 0.0, racc = 0.0;
#pragma unroll 4
    for (int i = 0; i < d; ++i) {
      const double p = (double)rowp[i * d];
      const double pii = (double)pv[i];
      acc = fma(p, pii, acc);
      if (KIND == MFG_REWARD_MFG_AC2) racc = fma(pii * (pj - pii), p * p, racc);
      if (KIND == MFG_REWARD_SYNTHETIC) racc = fma(pii, p * p, racc);
    }
    if (KIND != MFG_REWARD_EXTERNAL) {
      racc = seg_sum(racc, j, d, p2);
      if (KIND == MFG_REWARD_SYNTHETIC) racc *= -0.5;
    }
    if (valid) {
      pi_next[(b0 + tl) * d + j] = (float)acc;
      if (KIND != MFG_REWARD_EXTERNAL && j == 0) reward[b0 + tl] = (float)racc;
    }
    __syncthreads();
  }
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
    double pc[R][VEC], acc[R][VEC];
    int col[R];
#pragma unroll
    for (int m = 0; m < R; ++m) {
      col[m] = (m * WAVE + lane) * VEC;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        pc[m][v] = (col[m] + v < d) ? (double)pis[col[m] + v] : 0.0;
        acc[m][v] = 0.0;
      }
    }
    double racc = 0.0;
    const float* Pb = P + b * (int64_t)d * d;
#pragma unroll 4
    for (int i = 0; i < d; ++i) {
      const double pii = (double)pis[i];
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
            if (KIND == MFG_REWARD_MFG_AC2) racc = fma(pii * (pc[m][v] - pii), p * p, racc);
            if (KIND == MFG_REWARD_SYNTHETIC) racc = fma(pii, p * p, racc);
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
      racc = wave_sum(racc);
      if (KIND == MFG_REWARD_SYNTHETIC) racc *= -0.5;
      if (lane == 0) reward[b] = (float)racc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// Core actor-critic kernels: sampling (a1+a2), transition/reward (a3+a4), value (a5), TD error
// (a6), score (a7) -- T steps with fixed (theta, w), state kept on chip.
// ---------------------------------------------------------------------------------------------
struct CoreArgs {
  const float* pi0;         // [B,d]
  const float* pi_alpha;    // GIVEN: state the concentrations are computed from (NULL -> pi0)
  const float* P_in;        // GIVEN: [B,d,d]
  const float* pi_next_in;  // GIVEN: [B,d] (may be NULL when no delta is wanted)
  const float* reward_in;   // external reward [B*T] or NULL
  const double* theta;
  const double* w;          // NULL -> no value / delta
  double shift, alpha_scale, gamma;
  int64_t B;
  int d, T, reward_kind, discount_pow;
  uint64_t seed;
  uint32_t first_step;
  uint64_t traj_offset;
  float* pi_traj;     // [B,T+1,d] or NULL
  float* pi_next_out; // [B,d] final state or NULL
  float* reward_out;  // [B,T] or NULL
  double* delta;      // [B,T] or NULL
  double* g;          // [B,T] or NULL
  float* P_out;       // [B,T,d,d] or NULL
};

__device__ __forceinline__ double reward_term(int kind, double pii, double pj, double p) {
  // contribution of element (i,j) BEFORE the factor pi_i (kind 0) / -0.5 pi_i (kind 1)
  return kind == MFG_REWARD_MFG_AC2 ? (pj - pii) * p * p : p * p;
}

// small d: lane = (trajectory t, row i).  LDS per block: tile[TB][d][dp] (gamma variates, then P),
// pis[TB][d], pin[TB][d], wl[F] (critic weights, fp64).
template <bool SAMPLE, bool TD>
__global__ __launch_bounds__(BLOCK) void k_core_small(CoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int d = a.d, dd = d * d, dp = d | 1, T = a.T;
  const int G = WAVE / d, TB = WAVES * G;
  const int Q = d * (d + 1) / 2, F = Q + d + 1;
  const bool want_v = TD && a.w != nullptr;
  double* wl = reinterpret_cast<double*>(smem_raw);                   // [F] (only if want_v)
  float* tile = reinterpret_cast<float*>(wl + (want_v ? F : 0));      // [TB][d][dp]
  float* pis = tile + TB * d * dp;                                    // [TB][d]
  float* pin = pis + TB * d;                                          // [TB][d]
  float* pal = pin + TB * d;                                          // [TB][d] (GIVEN with pi_alpha)
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int t = lane / d, i = lane - t * d;
  const int p2 = next_pow2(d);
  const double theta = *a.theta;
  const float inv_d = 1.0f / (float)d;
  if (want_v) {
    for (int k = tid; k < F; k += BLOCK) wl[k] = a.w[k];
  }
  const int64_t ntiles = (a.B + TB - 1) / TB;
  for (int64_t tileid = blockIdx.x; tileid < ntiles; tileid += gridDim.x) {
    const int64_t b0 = tileid * TB;
    const int nb = (int)((a.B - b0) < TB ? (a.B - b0) : TB);
    const int tl = wv * G + t;
    const bool valid = (t < G) && (tl < nb);
    const int tlc = valid ? tl : 0;
    const int64_t b = b0 + tlc;
    float pi_i = a.pi0[b * d + i];
    if (valid && a.pi_traj) a.pi_traj[b * (int64_t)(T + 1) * d + i] = pi_i;
    double v_cur = 0.0, discount = 1.0;
    bool have_v = false;
    for (int s = 0; s < T; ++s) {
      __syncthreads();
      if (valid) pis[tlc * d + i] = pi_i;
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
      __syncthreads();
      float* trow = tile + (tlc * d + i) * dp;
      const float* pv = pis + tlc * d;
      const float* pav = (!SAMPLE && a.pi_alpha) ? pal + tlc * d : pv;
      const double pai = (double)pav[i];
      const double pid = (double)pi_i;
      double A = 0.0, D = 0.0, Ssum = 0.0, gacc = 0.0, racc = 0.0;
      if (valid) {
      for (int j = 0; j < d; ++j) {
        double al = 0.0, ad = 0.0;
        if (SAMPLE || TD) {
          const double x = (double)pav[j] - pai - a.shift;
          double sg;
          softplus_sigmoid(theta * x, al, sg);
          ad = x * sg;
        }
        double lnv = 0.0;
        if (SAMPLE) {
          float y = gamma_mt((float)(al * a.alpha_scale), a.seed, (uint32_t)(i * d + j), a.first_step + (uint32_t)s,
                             a.traj_offset + (uint64_t)b);
          if (y == 0.0f) y = ZERO_GAMMA_REPLACEMENT;
          Ssum += (double)y;
          trow[j] = y;
          if (TD) lnv = log((double)y);
        } else {
          const double p = (double)trow[j];
          if (TD) lnv = (p == 0.0) ? LOG_ZERO_P : log(p);
          racc += reward_term(a.reward_kind, pid, (double)pv[j], p);
        }
        if (TD) {
          A += al;
          D += ad;
          gacc = fma(-digamma_pos(al) + lnv, ad, gacc);
        }
      }
      if (SAMPLE) {
        // normalise the row: P_ij = fl32(y_ij / S_i); the reward uses the stored fp32 P
        const double invS = 1.0 / Ssum;
        for (int j = 0; j < d; ++j) {
          const float p32 = (float)((double)trow[j] * invS);
          trow[j] = p32;
          racc += reward_term(a.reward_kind, pid, (double)pv[j], (double)p32);
        }
        if (TD) gacc -= log(Ssum) * D;
      }
      if (TD) gacc = fma(digamma_pos(A), D, gacc);
      }  // valid
      __syncthreads();
      float pi_n;
      if (SAMPLE) {
        // pi'_i = sum_k pi_k P_ki : column read of the tile (consecutive lanes, consecutive banks)
        double acc = 0.0;
        const float* tcol = tile + tlc * d * dp + i;
        for (int k = 0; k < d; ++k) acc = fma((double)tcol[k * dp], (double)pv[k], acc);
        pi_n = (float)acc;
        if (valid) pin[tlc * d + i] = pi_n;
        if (a.P_out) {
          // coalesced copy-out of the block's P tile
          const int n = nb * dd;
          float* dst = a.P_out + (b0 * (int64_t)T) * dd;  // trajectory-major [B,T,d,d]
          for (int k = tid; k < n; k += BLOCK) {
            const int row = (int)(((float)k + 0.5f) * inv_d);  // tl*d + i
            const int colj = k - row * d;
            const int tl2 = (int)(((float)row + 0.5f) * inv_d);
            const int ii = row - tl2 * d;
            dst[((int64_t)tl2 * T + s) * dd + ii * d + colj] = tile[row * dp + colj];
          }
        }
      } else {
        pi_n = a.pi_next_in ? pin[tlc * d + i] : 0.0f;
      }
      double r;
      if (a.reward_kind == MFG_REWARD_EXTERNAL) {
        r = a.reward_in ? (double)a.reward_in[b * T + s] : 0.0;
      } else {
        r = seg_sum(pid * racc, i, d, p2);
        if (a.reward_kind == MFG_REWARD_SYNTHETIC) r *= -0.5;
      }
      if (valid && i == 0 && a.reward_out) a.reward_out[b * T + s] = (float)r;
      if (TD) {
        const double gsum = seg_sum(gacc, i, d, p2);
        if (valid && i == 0 && a.g) a.g[b * T + s] = gsum;
        if (want_v) {
          __syncthreads();  // pin complete
          if (!have_v) {
            double col = 0.0;
            for (int k = 0; k <= i; ++k) col = fma(wl[feat_idx(k, i, d)], (double)pv[k], col);
            v_cur = seg_sum(pid * (col + wl[Q + i]), i, d, p2) + wl[Q + d];
            have_v = true;
          }
          const float* pn = pin + tlc * d;
          double col = 0.0;
          for (int k = 0; k <= i; ++k) col = fma(wl[feat_idx(k, i, d)], (double)pn[k], col);
          const double v_next = seg_sum((double)pi_n * (col + wl[Q + i]), i, d, p2) + wl[Q + d];
          const double gd = a.discount_pow ? discount : a.gamma;
          const double del = r + gd * v_next - v_cur;
          if (valid && i == 0 && a.delta) a.delta[b * T + s] = del;
          v_cur = v_next;
          discount *= a.gamma;
        }
      }
      if (valid && a.pi_traj) a.pi_traj[(b * (int64_t)(T + 1) + s + 1) * d + i] = pi_n;
      pi_i = pi_n;
    }
    if (valid && a.pi_next_out) a.pi_next_out[b * d + i] = pi_i;
  }
}

// large d: one wavefront per trajectory, lane owns columns c = lane + 64 m (m < R).
template <int R, bool SAMPLE, bool TD>
__global__ __launch_bounds__(BLOCK) void k_core_large(CoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = a.d, T = a.T;
  const int64_t dd = (int64_t)d * d;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  float* pis = smem + wv * 3 * d;  // current state
  float* pin = pis + d;            // next state
  float* pal = pin + d;            // state for alpha (GIVEN with pi_alpha)
  const bool want_v = TD && a.w != nullptr;
  const double theta = *a.theta;
  const int64_t nw = (int64_t)gridDim.x * WAVES;
  for (int64_t b = (int64_t)blockIdx.x * WAVES + wv; b < a.B; b += nw) {
    float pc[R];
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int c = lane + m * WAVE;
      pc[m] = c < d ? a.pi0[b * d + c] : 0.0f;
      if (c < d && a.pi_traj) a.pi_traj[b * (int64_t)(T + 1) * d + c] = pc[m];
    }
    double v_cur = 0.0, discount = 1.0;
    bool have_v = false;
    for (int s = 0; s < T; ++s) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const int c = lane + m * WAVE;
        if (c < d) {
          pis[c] = pc[m];
          if (!SAMPLE && a.pi_next_in) pin[c] = a.pi_next_in[b * d + c];
          if (!SAMPLE && a.pi_alpha) pal[c] = a.pi_alpha[b * d + c];
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      const float* pav = (!SAMPLE && a.pi_alpha) ? pal : pis;
      double pcd[R], pad[R], acc[R];
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const int c = lane + m * WAVE;
        pcd[m] = (double)pc[m];
        pad[m] = c < d ? (double)pav[c] : 0.0;
        acc[m] = 0.0;
      }
      double racc = 0.0, gacc = 0.0, guni = 0.0;
      const float* Pb = SAMPLE ? nullptr : a.P_in + b * dd;
      float* Po = (SAMPLE && a.P_out) ? a.P_out + (b * (int64_t)T + s) * dd : nullptr;
      for (int i = 0; i < d; ++i) {
        const double pii = (double)pis[i];
        const double pai = (double)pav[i];
        float y[R];
        double Ssum = 0.0, A = 0.0, D = 0.0;
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const int c = lane + m * WAVE;
          y[m] = 0.0f;
          if (c < d) {
            double al = 0.0, ad = 0.0;
            if (SAMPLE || TD) {
              const double x = pad[m] - pai - a.shift;
              double sg;
              softplus_sigmoid(theta * x, al, sg);
              ad = x * sg;
            }
            double lnv = 0.0;
            if (SAMPLE) {
              float yy = gamma_mt((float)(al * a.alpha_scale), a.seed, (uint32_t)(i * d + c),
                                  a.first_step + (uint32_t)s, a.traj_offset + (uint64_t)b);
              if (yy == 0.0f) yy = ZERO_GAMMA_REPLACEMENT;
              y[m] = yy;
              Ssum += (double)yy;
              if (TD) lnv = log((double)yy);
            } else {
              y[m] = Pb[(int64_t)i * d + c];
              if (TD) lnv = (y[m] == 0.0f) ? LOG_ZERO_P : log((double)y[m]);
            }
            if (TD) {
              A += al;
              D += ad;
              gacc = fma(-digamma_pos(al) + lnv, ad, gacc);
            }
          }
        }
        double invS = 1.0;
        if (SAMPLE) {
          Ssum = wave_sum(Ssum);
          invS = 1.0 / Ssum;
        }
        if (TD) {
          A = wave_sum(A);
          D = wave_sum(D);
          guni += digamma_pos(A) * D;
          if (SAMPLE) guni -= log(Ssum) * D;
        }
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const int c = lane + m * WAVE;
          if (c < d) {
            const float p32 = SAMPLE ? (float)((double)y[m] * invS) : y[m];
            const double p = (double)p32;
            if (Po) Po[(int64_t)i * d + c] = p32;
            acc[m] = fma(p, pii, acc[m]);
            racc += pii * reward_term(a.reward_kind, pii, pcd[m], p);
          }
        }
      }
      // next state
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
        r = wave_sum(racc);
        if (a.reward_kind == MFG_REWARD_SYNTHETIC) r *= -0.5;
      }
      if (lane == 0 && a.reward_out) a.reward_out[b * T + s] = (float)r;
      if (TD) {
        const double gsum = wave_sum(gacc) + guni;
        if (lane == 0 && a.g) a.g[b * T + s] = gsum;
        if (want_v) {
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

// ---------------------------------------------------------------------------------------------
// a6/a8 batch sums: G = [ sum_n delta_n phi(pi_n) | sum delta_n g_n | sum r_n | N ].
// The quadratic block is sum_n delta_n pi_n pi_n^T (upper triangle): each block owns a chunk of
// samples (staged in LDS) x a chunk of 4*BLOCK outputs; partials go to the workspace and are summed
// in a fixed order by k_reduce_partials, so results are run-to-run deterministic.
// ---------------------------------------------------------------------------------------------
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

__global__ void k_reduce_partials(const double* __restrict__ partial, int64_t nsb, int64_t FO, int accumulate,
                                  double* __restrict__ G) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < FO; k += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int64_t p = 0; p < nsb; ++p) s += partial[p * FO + k];
    G[k] = accumulate ? G[k] + s : s;
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

static void grad_geometry(int64_t N, int d, int* chunk, int64_t* nsb, int* nob) {
  const int64_t FO = mfg_num_features(d) + 3;
  int ch = 8192 / d;
  if (ch > 64) ch = 64;
  if (ch < 8) ch = 8;
  *chunk = ch;
  int64_t sb = (N + ch - 1) / ch;
  int64_t cap = (int64_t)(32ll << 20) / (FO * 8);
  if (cap > 1024) cap = 1024;
  if (cap < 16) cap = 16;
  if (sb > cap) sb = cap;
  if (sb < 1) sb = 1;
  *nsb = sb;
  *nob = (int)((FO + GR_OUT_PER_BLOCK - 1) / GR_OUT_PER_BLOCK);
}

static int launch_grad(const float* pi, int64_t stride_b, const double* delta, const double* g, const float* reward,
                       int64_t N, int T, int d, double* G, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  const int64_t FO = mfg_num_features(d) + 3;
  int chunk, nob;
  int64_t nsb;
  grad_geometry(N, d, &chunk, &nsb, &nob);
  if (ws_bytes < (size_t)(nsb * FO * 8)) return fail(MFG_EWORKSPACE, "%s: need %lld bytes, have %lld", "workspace",
                                                     (long long)(nsb * FO * 8), (long long)ws_bytes);
  GradArgs a{pi, stride_b, delta, g, reward, N, T, d, chunk, nsb, (double*)ws};
  const size_t lds = (size_t)chunk * 3 * 8 + (size_t)chunk * d * 4;
  hipLaunchKernelGGL(k_grad_partial, dim3((unsigned)nsb, (unsigned)nob), dim3(BLOCK), lds, st, a);
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((FO + 255) / 256)), dim3(256), 0, st, (const double*)ws, nsb, FO,
                     accumulate, G);
  return check_launch("grad_reduce");
}

static size_t core_small_lds(int d, bool want_v) {
  const int G = WAVE / d, TB = WAVES * G, dp = d | 1;
  const int64_t F = mfg_num_features(d);
  return (want_v ? (size_t)F * 8 : 0) + (size_t)TB * d * dp * 4 + 3 * (size_t)TB * d * 4;
}

template <bool SAMPLE, bool TD>
static int launch_core(const CoreArgs& a, hipStream_t st) {
  const int d = a.d;
  if (d <= WAVE) {
    const bool want_v = TD && a.w != nullptr;
    const size_t lds = core_small_lds(d, want_v);
    const int G = WAVE / d, TB = WAVES * G;
    int bpc = (int)((160 * 1024) / (lds + 256));
    if (bpc > 8) bpc = 8;
    if (bpc < 1) bpc = 1;
    const int grid = grid_for(a.B, TB, bpc);
    hipLaunchKernelGGL((k_core_small<SAMPLE, TD>), dim3(grid), dim3(BLOCK), lds, st, a);
  } else {
    const int R = (d + WAVE - 1) / WAVE;
    const size_t lds = (size_t)WAVES * 3 * d * 4;
    const int grid = grid_for(a.B, WAVES, 8);
#define CORE_LARGE(RR)                                                                                  \
  case RR:                                                                                              \
    hipLaunchKernelGGL((k_core_large<RR, SAMPLE, TD>), dim3(grid), dim3(BLOCK), lds, st, a);             \
    break;
    switch (R) {
      CORE_LARGE(2)
      CORE_LARGE(3)
      CORE_LARGE(4)
      CORE_LARGE(5)
      CORE_LARGE(6)
      CORE_LARGE(7)
      CORE_LARGE(8)
      default:
        return fail(MFG_EUNSUPPORTED, "%s: d=%lld > %lld", "core", (long long)d, (long long)MFG_MAX_D);
    }
#undef CORE_LARGE
  }
  return check_launch("core");
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char* mfg_last_error(void) { return g_err; }
int mfg_abi_version(void) { return 1; }

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
  return (size_t)(nsb * (mfg_num_features(d) + 3) * 8);
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
  hipLaunchKernelGGL(k_gather_start, dim3(grid_for(B * d, 256, 8)), dim3(256), 0, S(stream), mat_pi0, idx, B, d, pi0);
  return check_launch("gather_start");
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

int mfg_step_given_P(const float* pi, const float* P, int64_t B, int d, int reward_kind, float* pi_next, float* reward,
                     mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(pi && P && pi_next, "null pointer");
  REQUIRE(reward_kind >= 0 && reward_kind <= 2, "bad reward_kind");
  if (!reward) reward_kind = MFG_REWARD_EXTERNAL;
  hipStream_t st = S(stream);
  if (d <= WAVE) {
    const int G = WAVE / d, TB = WAVES * G;
    const size_t lds = (size_t)TB * d * d * 4 + (size_t)TB * d * 4;
    int bpc = (int)((160 * 1024) / (lds + 256));
    if (bpc > 8) bpc = 8;
    if (bpc < 1) bpc = 1;
    const int grid = grid_for(B, TB, bpc);
    const int vec_ok = (((uintptr_t)P & 15) == 0) && (((int64_t)TB * d * d) % 4 == 0);
#define STEP_SMALL(K)                                                                                         \
  case K:                                                                                                     \
    hipLaunchKernelGGL((k_step_small<K>), dim3(grid), dim3(BLOCK), lds, st, pi, P, B, d, vec_ok, pi_next, reward); \
    break;
    switch (reward_kind) {
      STEP_SMALL(0)
      STEP_SMALL(1)
      STEP_SMALL(2)
    }
#undef STEP_SMALL
  } else {
    const size_t lds = (size_t)WAVES * d * 4;
    const int grid = grid_for(B, WAVES, 8);
    const bool a16 = (((uintptr_t)P & 15) == 0) && (((uintptr_t)pi_next & 15) == 0);
    int vec = 1;
    if (a16 && d % 4 == 0) vec = 4;
    else if (a16 && d % 2 == 0) vec = 2;
    // keep at most 2 chunks per lane on the vector paths, fall back to narrower vectors otherwise
    int R = (d + WAVE * vec - 1) / (WAVE * vec);
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

int mfg_jsd(const float* p, const float* q, int64_t B, int d, double* out, mfg_stream_t stream) {
  REQUIRE(B >= 0 && d >= 1, "bad shape");
  if (B == 0) return MFG_OK;
  REQUIRE(p && q && out, "null pointer");
  hipLaunchKernelGGL(k_jsd, dim3(grid_for(B, WAVES, 8)), dim3(BLOCK), 0, S(stream), p, q, B, d, out);
  return check_launch("jsd");
}

int mfg_sample_dirichlet(const float* pi, int64_t B, int d, const double* theta, double shift, double alpha_scale,
                         uint64_t seed, uint32_t step, uint64_t traj_offset, float* P, mfg_stream_t stream) {
  CHECK_BD();
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
  return launch_core<true, false>(a, S(stream));
}

int mfg_score(const float* pi_alpha, const float* P, int64_t B, int d, const double* theta, double shift, double* g,
              mfg_stream_t stream) {
  CHECK_BD();
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
  return launch_core<false, true>(a, S(stream));
}

int mfg_td_pg_accumulate(const float* pi, const float* pi_next, const float* P, const float* reward, const double* w,
                         const double* theta, double shift, double gamma_or_discount, int64_t B, int d, double* delta,
                         double* g, double* G, int accumulate, void* workspace, size_t workspace_bytes,
                         mfg_stream_t stream) {
  CHECK_BD();
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
  int rc = launch_core<false, true>(a, S(stream));
  if (rc != MFG_OK || !G) return rc;
  REQUIRE(workspace, "workspace is null");
  return launch_grad(pi, d, delta, g, reward, B, 1, d, G, accumulate, workspace, workspace_bytes, S(stream));
}

int mfg_apply_update(const double* G, int d, double lr_critic, double lr_actor, double* w, double* theta,
                     mfg_stream_t stream) {
  REQUIRE(G && w && theta && d >= 1, "null pointer");
  const int64_t F = mfg_num_features(d);
  hipLaunchKernelGGL(k_apply_update, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, S(stream), G, F, lr_critic,
                     lr_actor, w, theta);
  return check_launch("apply_update");
}

int mfg_rollout(const float* pi0, int64_t B, int d, int T, const double* theta, double shift, double alpha_scale,
                const double* w, double gamma, int reward_kind, uint64_t seed, uint32_t first_step, uint64_t traj_offset,
                int flags, float* pi_traj, float* reward, double* delta, double* g, float* P_out, double* G,
                int accumulate, void* workspace, size_t workspace_bytes, mfg_stream_t stream) {
  CHECK_BD();
  REQUIRE(T >= 1, "T < 1");
  REQUIRE(pi0 && theta, "null pointer");
  REQUIRE(reward_kind == MFG_REWARD_MFG_AC2 || reward_kind == MFG_REWARD_SYNTHETIC, "fused rollout needs an in-kernel reward");
  const bool td = (flags & MFG_ROLLOUT_TD) != 0;
  REQUIRE(!(flags & MFG_ROLLOUT_WRITE_P) || P_out, "WRITE_P without P_out");
  REQUIRE(!td || (w && delta && g && reward && pi_traj), "TD rollout needs w, delta, g, reward, pi_traj");
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
  a.reward_out = reward;
  a.delta = delta;
  a.g = g;
  a.P_out = (flags & MFG_ROLLOUT_WRITE_P) ? P_out : nullptr;
  int rc = td ? launch_core<true, true>(a, S(stream)) : launch_core<true, false>(a, S(stream));
  if (rc != MFG_OK || !td || !G) return rc;
  REQUIRE(workspace, "workspace is null");
  return launch_grad(pi_traj, (int64_t)(T + 1) * d, delta, g, reward, B * T, T, d, G, accumulate, workspace,
                     workspace_bytes, S(stream));
}

}  // extern "C"
