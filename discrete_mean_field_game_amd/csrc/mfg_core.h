// Core actor-critic kernels: sampling (a1+a2), transition/reward (a3+a4), value (a5), TD error (a6),
// score (a7) -- T steps with fixed (theta, w), state kept on chip.  Included by the translation units
// that instantiate them (mfg_core_small.hip, mfg_core_large_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfg_hip.h"
#include "mfg_device.h"

namespace mfg {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / WAVE;

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
  float* pi_traj;      // [B,T+1,d] or NULL
  float* pi_next_out;  // [B,d] final state or NULL
  float* reward_out;   // [B,T] or NULL
  double* delta;       // [B,T] or NULL
  double* g;           // [B,T] or NULL
  float* P_out;        // [B,T,d,d] or NULL
};

// launchers defined in mfg_core_small.hip / mfg_core_large_*.hip; return 0 or MFG_EUNSUPPORTED
int launch_core_small(const CoreArgs& a, bool sample, bool td, bool fast, int num_cus, hipStream_t st);
int launch_core_large_f64(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st);
int launch_core_large_mixed(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st);

__device__ __forceinline__ double reward_term(int kind, double pii, double pj, double p) {
  // contribution of element (i,j) BEFORE the factor pi_i (kind 0) / -0.5 pi_i (kind 1)
  return kind == MFG_REWARD_MFG_AC2 ? (pj - pii) * p * p : p * p;
}

// One matrix element of the policy: concentration, its theta-derivative, (SAMPLE) a gamma variate or
// (GIVEN) the stored probability, and the element's share of the row sums / score.
//   x = pi_j - pi_i - shift.  Returns the gamma variate (SAMPLE) or p_given.
template <bool SAMPLE, bool TD, bool FAST>
__device__ __forceinline__ float policy_elem(const CoreArgs& a, double theta, double x, uint32_t elem, uint32_t step,
                                             uint64_t traj, float p_given, double& A, double& D, double& Ssum,
                                             double& gacc) {
  float y = p_given;
  if (FAST) {
    float al = 0.f, sg = 0.f;
    if (SAMPLE || TD) softplus_sigmoid_fast(theta * x, al, sg);
    float lnv = 0.f;
    if (SAMPLE) {
      y = gamma_mt(al * (float)a.alpha_scale, a.seed, elem, step, traj);
      if (y == 0.0f) y = ZERO_GAMMA_REPLACEMENT;
      Ssum += (double)y;
      if (TD) lnv = __logf(y);
    } else if (TD) {
      lnv = (y == 0.0f) ? (float)LOG_ZERO_P : __logf(y);
    }
    if (TD) {
      const float ad = (float)x * sg;
      A += (double)al;
      D += (double)ad;
      gacc = fma((double)(lnv - digamma_pos_fast(al)), (double)ad, gacc);
    }
  } else {
    double al = 0.0, ad = 0.0;
    if (SAMPLE || TD) {
      double sg;
      softplus_sigmoid(theta * x, al, sg);
      ad = x * sg;
    }
    double lnv = 0.0;
    if (SAMPLE) {
      y = gamma_mt((float)(al * a.alpha_scale), a.seed, elem, step, traj);
      if (y == 0.0f) y = ZERO_GAMMA_REPLACEMENT;
      Ssum += (double)y;
      if (TD) lnv = log((double)y);
    } else if (TD) {
      lnv = (y == 0.0f) ? LOG_ZERO_P : log((double)y);
    }
    if (TD) {
      A += al;
      D += ad;
      gacc = fma(-digamma_pos(al) + lnv, ad, gacc);
    }
  }
  return y;
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
template <bool SAMPLE, bool TD, bool FAST>
__global__ __launch_bounds__(BLOCK) void k_core_small(CoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int d = a.d, dd = d * d, dp = d | 1, T = a.T;
  const int G = WAVE / d, TB = WAVES * G;
  const int Q = d * (d + 1) / 2, F = Q + d + 1;
  const bool want_v = TD && a.w != nullptr;
  double* wl = reinterpret_cast<double*>(smem_raw);
  float* tile = reinterpret_cast<float*>(wl + (want_v ? F : 0));
  float* pis = tile + TB * d * dp;
  float* pin = pis + TB * d;
  float* pal = pin + TB * d;
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
        const uint32_t step = a.first_step + (uint32_t)s;
        const uint64_t traj = a.traj_offset + (uint64_t)b;
        for (int j = 0; j < d; ++j) {
          const double x = (double)pav[j] - pai - a.shift;
          const float pg = SAMPLE ? 0.0f : trow[j];
          const float y = policy_elem<SAMPLE, TD, FAST>(a, theta, x, (uint32_t)(i * d + j), step, traj, pg, A, D, Ssum, gacc);
          if (SAMPLE) trow[j] = y;
          else racc += reward_term(a.reward_kind, pid, (double)pv[j], (double)y);
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
      }
      __syncthreads();
      float pi_n;
      if (SAMPLE) {
        // pi'_i = sum_k pi_k P_ki : column read of the tile (consecutive lanes -> consecutive banks)
        double acc = 0.0;
        const float* tcol = tile + tlc * d * dp + i;
        for (int k = 0; k < d; ++k) acc = fma((double)tcol[k * dp], (double)pv[k], acc);
        pi_n = (float)acc;
        if (valid) pin[tlc * d + i] = pi_n;
        if (a.P_out) {
          // coalesced copy-out of the block's P tile into [B,T,d,d]
          const int n = nb * dd;
          float* dst = a.P_out + (b0 * (int64_t)T) * dd;
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

inline size_t core_small_lds(int d, bool want_v) {
  const int G = WAVE / d, TB = WAVES * G, dp = d | 1;
  const size_t F = (size_t)d * (d + 1) / 2 + d + 1;
  return (want_v ? F * 8 : 0) + (size_t)TB * d * dp * 4 + 3 * (size_t)TB * d * 4;
}

// ---------------------------------------------------------------------------------------------
// large d (d > 64): one wavefront per trajectory, lane owns columns c = lane + 64 m (m < R).
// ---------------------------------------------------------------------------------------------
template <int R, bool SAMPLE, bool TD, bool FAST>
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
    const uint64_t traj = a.traj_offset + (uint64_t)b;
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
      const uint32_t step = a.first_step + (uint32_t)s;
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
            const double x = pad[m] - pai - a.shift;
            const float pg = SAMPLE ? 0.0f : Pb[(int64_t)i * d + c];
            y[m] = policy_elem<SAMPLE, TD, FAST>(a, theta, x, (uint32_t)(i * d + c), step, traj, pg, A, D, Ssum, gacc);
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

inline int core_grid(int64_t work_items, int per_block, int blocks_per_cu, int num_cus) {
  int64_t g = (work_items + per_block - 1) / per_block;
  const int64_t cap = (int64_t)num_cus * blocks_per_cu;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

template <bool FAST>
inline int launch_core_large_impl(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st) {
  const int d = a.d;
  const int R = (d + WAVE - 1) / WAVE;
  const size_t lds = (size_t)WAVES * 3 * d * 4;
  const int grid = core_grid(a.B, WAVES, 8, num_cus);
#define MFG_CORE_LARGE_MODE(RR)                                                                              \
  if (sample && td) hipLaunchKernelGGL((k_core_large<RR, true, true, FAST>), dim3(grid), dim3(BLOCK), lds, st, a);        \
  else if (sample) hipLaunchKernelGGL((k_core_large<RR, true, false, FAST>), dim3(grid), dim3(BLOCK), lds, st, a);        \
  else hipLaunchKernelGGL((k_core_large<RR, false, true, FAST>), dim3(grid), dim3(BLOCK), lds, st, a);
  switch (R) {
    case 2: MFG_CORE_LARGE_MODE(2) break;
    case 3: MFG_CORE_LARGE_MODE(3) break;
    case 4: MFG_CORE_LARGE_MODE(4) break;
    case 5: MFG_CORE_LARGE_MODE(5) break;
    case 6: MFG_CORE_LARGE_MODE(6) break;
    case 7: MFG_CORE_LARGE_MODE(7) break;
    case 8: MFG_CORE_LARGE_MODE(8) break;
    default: return MFG_EUNSUPPORTED;
  }
#undef MFG_CORE_LARGE_MODE
  return MFG_OK;
}

}  // namespace mfg
