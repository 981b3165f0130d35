// Device-side building blocks of the MFG hot path (gfx950, wave64).
// Philox4x32-10, Marsaglia-Tsang gamma, fp64 digamma / softplus, segmented wave reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfg {

constexpr int WAVE = 64;
constexpr float ZERO_GAMMA_REPLACEMENT = 1e-20f;   // mfg_ac2.py:244
constexpr double LOG_ZERO_P = -230.25850929940458;  // ln(1e-100), mfg_ac2.py:369

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Counter layout used by every sampler:
//   c0 = element index i*d + j, c1 = env step, c2 = low 32 bits of the global trajectory id,
//   c3 = (high 16 bits of the trajectory id) | (draw block << 16); key = 64-bit seed.
// ---------------------------------------------------------------------------
struct u32x4 {
  uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c.x;
    const uint64_t p1 = (uint64_t)M1 * c.z;
    u32x4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

__device__ __forceinline__ u32x4 philox_elem(uint64_t seed, uint32_t elem, uint32_t step, uint64_t traj,
                                             uint32_t block) {
  u32x4 c;
  c.x = elem;
  c.y = step;
  c.z = (uint32_t)traj;
  c.w = ((uint32_t)(traj >> 32) & 0xFFFFu) | (block << 16);
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// (0,1) open interval from the top 24 bits.
__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-8f; }

// ---------------------------------------------------------------------------
// Gamma(shape a, scale 1), Marsaglia & Tsang (2000) with the U^(1/a) boost for a < 1.
// fp32; one Philox block per attempt.  Returns > 0 or exactly 0 on
// underflow (the caller applies the reference's zero replacement).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gamma_mt(float a, uint64_t seed, uint32_t elem, uint32_t step, uint64_t traj) {
  const bool small = a < 1.0f;
  const float a1 = small ? a + 1.0f : a;
  const float dd = a1 - (1.0f / 3.0f);
  const float c = rsqrtf(9.0f * dd);
  float v = 1.0f;
  float boost_u = 1.0f;
  // One Philox block per attempt: normal = Box-Muller(r.x, r.y), accept uniform = r.z; r.w of block 0
  // is the boost uniform.  Acceptance is > 95 % so the loop almost never iterates.
  for (uint32_t block = 0; block < 64; ++block) {
    const u32x4 r = philox_elem(seed, elem, step, traj, block);
    if (block == 0) boost_u = u01(r.w);
    const float x = sqrtf(-2.0f * __logf(u01(r.x))) * __builtin_amdgcn_cosf(u01(r.y));
    const float u = u01(r.z);
    const float t = c * x;
    if (t <= -1.0f) continue;
    // eps = v - 1 with v = (1+t)^3, formed without cancellation
    const float eps = t * (3.0f + t * (3.0f + t));
    const float x2 = x * x;
    bool acc = u < 1.0f - 0.0331f * x2 * x2;
    if (!acc) {
      // log(v) - eps: series for small eps (large shapes), direct otherwise
      float lme;
      if (fabsf(eps) < 0.125f) {
        const float e2 = eps * eps;
        lme = e2 * (-0.5f + eps * (1.0f / 3.0f + eps * (-0.25f + eps * (0.2f + eps * (-1.0f / 6.0f +
              eps * (1.0f / 7.0f + eps * (-0.125f)))))));
      } else {
        lme = __logf(1.0f + eps) - eps;
      }
      acc = __logf(u) < 0.5f * x2 + dd * lme;
    }
    if (acc) {
      v = 1.0f + eps;
      break;
    }
  }
  float y = dd * v;
  if (small) y *= __powf(boost_u, 1.0f / a);
  return y;
}

// ---------------------------------------------------------------------------
// fp64 special functions
// ---------------------------------------------------------------------------
// digamma for x > 0: recurrence to x >= 8 through one rational step, then the asymptotic series.
__device__ __forceinline__ double digamma_pos(double x) {
  double corr = 0.0;
  if (x < 8.0) {
    // sum_{k=0..7} 1/(x+k) = q'(x)/q(x), q = prod (x+k): all terms positive, no cancellation.
    double q = x, qp = 1.0;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const double xk = x + (double)k;
      qp = fma(qp, xk, q);
      q = q * xk;
    }
    corr = qp / q;
    x += 8.0;
  }
  const double inv = 1.0 / x;
  const double inv2 = inv * inv;
  // Bernoulli series  B_2k / (2k x^2k), k = 1..7
  // Horner on t = inv2:  t (1/12 - t/120 + t^2/252 - t^3/240 + t^4/132 - t^5 691/32760 + t^6/12)
  const double s = inv2 * (1.0 / 12.0 - inv2 * (1.0 / 120.0 - inv2 * (1.0 / 252.0 - inv2 * (1.0 / 240.0 - inv2 *
      (1.0 / 132.0 - inv2 * (691.0 / 32760.0 - inv2 * (1.0 / 12.0)))))));
  return log(x) - 0.5 * inv - s - corr;
}

// alpha = ln(1 + e^z) and sigmoid(z), z = theta * x   (mfg_ac2.py:228, :233-234)
__device__ __forceinline__ void softplus_sigmoid(double z, double& sp, double& sg) {
  const double e = exp(z);
  sp = log1p(e);
  sg = e / (1.0 + e);
}

// ---------------------------------------------------------------------------
// Mixed precision ("MFG_PRECISION_MIXED"): fp32 hardware transcendentals (v_exp_f32 / v_log_f32 /
// v_rcp_f32, ~1 ulp), fp64 only for the sums.  Measured effect on the score g: ~1e-6 relative.
// ---------------------------------------------------------------------------
// alpha = log1p(e^z), sigmoid(z) with z = theta*x given in fp64: the low part of z is folded in so
// that e carries fp32 (not |z| * 2^-24) relative error; log1p by Kahan's log(u) * e/(u-1).
__device__ __forceinline__ void softplus_sigmoid_fast(double z, float& sp, float& sg) {
  const float zh = (float)z;
  const float zl = (float)(z - (double)zh);
  float e = __expf(zh);
  e = fmaf(e, zl, e);
  const float u = 1.0f + e;
  sg = e * __builtin_amdgcn_rcpf(u);
  const float um1 = u - 1.0f;
  sp = (um1 == 0.0f) ? e : __logf(u) * (e * __builtin_amdgcn_rcpf(um1));
}

__device__ __forceinline__ float digamma_pos_fast(float x) {
  float corr = 0.0f;
  if (x < 8.0f) {
    float q = x, qp = 1.0f;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float xk = x + (float)k;
      qp = fmaf(qp, xk, q);
      q = q * xk;
    }
    corr = qp * __builtin_amdgcn_rcpf(q);
    x += 8.0f;
  }
  const float inv = __builtin_amdgcn_rcpf(x);
  const float inv2 = inv * inv;
  const float s = inv2 * (1.0f / 12.0f - inv2 * (1.0f / 120.0f - inv2 * (1.0f / 252.0f)));
  return __logf(x) - 0.5f * inv - s - corr;
}

// ---------------------------------------------------------------------------
// wave-level reductions
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

// Sum over a segment of `len` consecutive lanes starting at lane (lane - pos); `pos` = position of
// this lane inside its segment, `p2` = smallest power of two >= len.  Every lane of the wave must
// call this.  The total is returned to every lane of the segment.
template <typename T>
__device__ __forceinline__ T seg_sum(T v, int pos, int len, int p2) {
  int cur = len;
  for (int off = p2 >> 1; off > 0; off >>= 1) {
    const T o = __shfl_down(v, off, WAVE);
    if (pos < off && pos + off < cur) v += o;
    cur = cur < off ? cur : off;
  }
  const int lane = (int)(threadIdx.x & (WAVE - 1));
  return __shfl(v, lane - pos, WAVE);
}

__host__ __device__ __forceinline__ int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// k(i,j) for i <= j: row-major upper triangle (mfg_ac2.py:333).
__host__ __device__ __forceinline__ int feat_idx(int i, int j, int d) { return i * d - (i * (i - 1)) / 2 + (j - i); }

}  // namespace mfg
