// Device-side building blocks of the MFG hot path (gfx950, wave64).
// Philox4x32-10, Marsaglia-Tsang gamma, fp64 digamma / softplus, segmented wave reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfg {

constexpr int WAVE = 64;
typedef double v4d_t __attribute__((ext_vector_type(4)));  // accumulators of v_mfma_f64_16x16x4_f64
constexpr float ZERO_GAMMA_REPLACEMENT = 1e-20f;   // mfg_ac2.py:244
constexpr double LOG_ZERO_P = -230.25850929940458;  // ln(1e-100), mfg_ac2.py:369
constexpr double LN2 = 0.6931471805599453, INV_LN2 = 1.4426950408889634;

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

// (0,1] from the top 24 bits (the +0.5 keeps log(u) finite).
__device__ __forceinline__ float u01(uint32_t r) { return __builtin_fmaf((float)(r >> 8), 5.9604644775390625e-8f, 2.98023223876953125e-8f); }

// Raw hardware transcendentals (v_log_f32 / v_exp_f32: base 2, ~1 ulp, no denormal fix-ups: every
// argument below is a normal number).
__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// ---------------------------------------------------------------------------
// Gamma(shape a, scale 1), Marsaglia & Tsang (2000) with the U^(1/a) boost for a < 1.  fp32.
// Matrix elements are sampled in QUADS (see "Quad sampler" below): one Philox block gives two Box-Muller pairs
// (cos / sin of the same radius) and four 12-bit acceptance integers.
// ---------------------------------------------------------------------------
// The quad sampler's normals are kept in units of K = sqrt(2 ln 2): Box-Muller gives x = sqrt(-2 ln u) cos(phi) =
// K sqrt(-log2 u) cos(phi), and everything the hot path does with x is scale covariant (t = c x, the squeeze threshold in
// (x t)^2), so the hot path works with xs = x / K and a pre-scaled c -- the multiplication by -2 ln 2 in front of the square
// root (one instruction per Box-Muller pair) is folded into constants.  The exact continuation restores x = K xs.
constexpr float BM_K = 1.1774100225154747f;       // sqrt(2 ln 2)
constexpr float BM_K2 = 1.3862943611198906f;      // 2 ln 2
struct GammaState {
  float a, dd, c;  // hot path (gamma_setup_hot): c = K / sqrt(9 d), applied to xs; gamma_setup: c = 1 / sqrt(9 d), applied to x
  bool small;
};

__device__ __forceinline__ void gamma_setup(GammaState& g, float a) {
  g.a = a;
  g.small = a < 1.0f;
  const float a1 = g.small ? a + 1.0f : a;
  g.dd = a1 - (1.0f / 3.0f);
  g.c = __builtin_amdgcn_rsqf(9.0f * g.dd);
}
// Hot-path variant of the quad sampler: (d, c) for the shape as it is; shapes < 1 are flagged and re-set-up with the
// boosted shape a + 1 in the cold continuation (gamma_fix), so the hot path carries no select for them.
__device__ __forceinline__ void gamma_setup_hot(GammaState& g, float a) {
  g.a = a;
  g.small = a < 1.0f;
  g.dd = a - (1.0f / 3.0f);
  g.c = __builtin_amdgcn_rsqf((9.0f / BM_K2) * g.dd);  // K / sqrt(9 d); garbage for a < 1/3: such elements never use the hot-path result
}
// The same from the concentration alpha and the scale (shape a = alpha * scale), one instruction shorter: d and 9 d are
// one fma each of alpha, the small-shape flag is read off d (a < 1 <=> d < 2/3), and the shape itself -- which only the
// cold small-shape continuation needs -- is d + 1/3.
//   scale9k = 9 scale / (2 ln 2): c = K / sqrt(9 d) = rsq(9 d / K^2)
__device__ __forceinline__ void gamma_setup_hot(GammaState& g, float alpha, float scale, float scale9k) {
  g.dd = __builtin_fmaf(alpha, scale, -(1.0f / 3.0f));
  g.small = g.dd < (2.0f / 3.0f);
  g.a = g.dd + (1.0f / 3.0f);  // dead on the hot path
  g.c = __builtin_amdgcn_rsqf(__builtin_fmaf(alpha, scale9k, -3.0f / BM_K2));
}

// Marsaglia-Tsang acceptance for normal x and uniform u; v = (1 + c x)^3.
__device__ __forceinline__ bool mt_accept(const GammaState& g, float x, float u, float& v) {
  const float t = g.c * x;
  const float eps = t * (3.0f + t * (3.0f + t));  // v - 1, formed without cancellation
  const float x2 = x * x;
  v = 1.0f + eps;
  bool acc = u < 1.0f - 0.0331f * x2 * x2;
  if (!acc) {
    // log(v) - eps: series for small eps (large shapes), direct otherwise
    float lme;
    if (__builtin_fabsf(eps) < 0.125f) {
      const float e2 = eps * eps;
      lme = e2 * (-0.5f + eps * (1.0f / 3.0f + eps * (-0.25f + eps * (0.2f + eps * (-1.0f / 6.0f +
            eps * (1.0f / 7.0f + eps * (-0.125f)))))));
    } else {
      lme = fast_ln(v) - eps;
    }
    acc = fast_ln(u) < 0.5f * x2 + g.dd * lme;
  }
  return acc && (t > -1.0f);
}

__device__ __forceinline__ float box_muller_radius(uint32_t r) {
  return __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r)));  // sqrt(-2 ln u)
}

// ---------------------------------------------------------------------------
// Quad sampler (round 2): ONE Philox block serves FOUR matrix elements (two Box-Muller pairs), and the acceptance
// test of an element is decided from 12 random bits in all but ~1e-3 of the cases -- without giving up exactness.
//
//  * Bits of the block r = (x, y, z, w):  pair h (h = 0, 1): radius uniform = top 24 bits of r.x / r.y, angle = 16 bits of
//    r.z (high / low half); acceptance: 48 bits = r.w and the low bytes of r.x, r.y, cut into four 12-bit integers k.
//    The true acceptance uniform of an element is u = (k + u') / 4096 with u' uniform in (0,1), drawn ONLY when the 12
//    leading bits do not decide the test.
//  * Squeeze for the shapes the policy actually has (hundreds ... tens of thousands):  with t = c x, d = 1/(9 c^2) the
//    Marsaglia-Tsang exponent is  x^2/2 + d (ln v - (v - 1)) = 3 d [ln(1+t) - t + t^2/2 - t^3/3]
//    = -3 d int_0^t s^3/(1+s) ds >= -(3/4) d t^4 / (1 - |t|) >= -1.5 d t^4 = -x^2 t^2 / 6  for |t| <= 1/2,
//    so  u < 1 - x^2 (0.19 t^2 + 1e-6)  implies acceptance (0.19 > 1/6 and the 1e-6 absorb the rounding of c, x).  The
//    bound is missed with probability ~0.06 / shape (MT's generic 1 - 0.0331 x^4 squeeze fails for every |x| > 2 and
//    sent the whole wave through the log test on every draw).
//  * Exact path (|t| > 1/2, squeeze undecided, or k in the top cell): u' from block 1 of the ELEMENT's own counter, the
//    original MT test; on rejection fresh normals from blocks 2, 3, ... of that counter.  Shapes < 1: boost uniform
//    from block 0xFFFF of the element's counter.
// Counters: the quad block is (elem of the quad's first element, block 0); everything else is keyed by the element.
// ---------------------------------------------------------------------------
struct QuadRand {
  float radu[2];   // radius uniforms in (0,1)
  float ang[2];    // angles in revolutions, (0,1)
  float kf[4];     // acceptance integers as floats: 0 .. 2^KBITS(e) - 1
};
// Bit layout of the quad block r = (x, y, z, w) (round 4):
//   radius uniform of pair h: the top 20 bits of r.x / r.y        ((k + 1/2) 2^-20: Box-Muller radii up to 5.4)
//   angle of pair h:          high / low half of r.z               (16 bits)
//   acceptance integers:      k0 = r.w >> 16, k1 = r.w & 0xFFFF    (16 bits: elements 0, 1)
//                             k2 = r.x & 0xFFF, k3 = r.y & 0xFFF   (12 bits: elements 2, 3)
// Every field is one shift / mask / half-word select away from its word (round 2 cut 4 x 12 bits out of r.w and the low
// BYTES of r.x, r.y: 8 instructions per quad for k2, k3 alone), and the two 16-bit integers are undecided 16 times less
// often: the wave-uniform exact-path branch was taken by 3.4 % of the pairs (any of 128 elements with k in the top cell of
// 4096) at ~110 instructions each -- 1.9 instructions per element on average; now 1.8 % of the pairs.
__host__ __device__ constexpr int quad_kbits(int e) { return e < 2 ? 16 : 12; }

__device__ __forceinline__ void quad_rand(QuadRand& q, uint64_t seed, uint32_t elem0, uint32_t step, uint64_t traj) {
#ifdef MFG_ABL_PHILOX  // timing ablation only (tools/ablate.sh): a two-multiply hash instead of the Philox block
  u32x4 r;
  r.x = (elem0 + step) * 2654435761u ^ (uint32_t)traj;
  r.y = r.x * 2246822519u + 12345u;
  r.z = r.y ^ (r.x >> 7);
  r.w = r.z * 3266489917u;
#else
  const u32x4 r = philox_elem(seed, elem0, step, traj, 0);
#endif
  q.ang[0] = __builtin_fmaf((float)(r.z >> 16), 1.52587890625e-5f, 7.62939453125e-6f);
  q.ang[1] = __builtin_fmaf((float)(r.z & 0xFFFFu), 1.52587890625e-5f, 7.62939453125e-6f);
  q.radu[0] = __builtin_fmaf((float)(r.x >> 12), 9.5367431640625e-7f, 4.76837158203125e-7f);   // (k + 1/2) 2^-20
  q.radu[1] = __builtin_fmaf((float)(r.y >> 12), 9.5367431640625e-7f, 4.76837158203125e-7f);
  q.kf[0] = (float)(r.w >> 16);
  q.kf[1] = (float)(r.w & 0xFFFFu);
  q.kf[2] = (float)(r.x & 0xFFFu);
  q.kf[3] = (float)(r.y & 0xFFFu);
}

// Exact continuation for one element: returns v = (1 + c x)^3 of the accepted draw.
//   kscale = 2^-KBITS of the element's acceptance integer kf (quad_kbits)
__device__ __forceinline__ float gamma_exact_path(const GammaState& g, float x, float kf, float kscale, uint64_t seed,
                                                  uint32_t elem, uint32_t step, uint64_t traj) {
  float v = 1.0f;
  {
    const u32x4 r = philox_elem(seed, elem, step, traj, 1);
    const float u = (kf + u01(r.x)) * kscale;  // the element's full-precision acceptance uniform
    if (mt_accept(g, x, u, v)) return v;
  }
  for (uint32_t block = 2; block < 64; ++block) {
    const u32x4 r = philox_elem(seed, elem, step, traj, block);
    const float xn = box_muller_radius(r.x) * __builtin_amdgcn_cosf(u01(r.y));
    if (mt_accept(g, xn, u01(r.z), v)) return v;
  }
  return 1.0f;  // never reached in practice
}

// Branch-free hot half of the quad sampler: v = (1 + c x)^3 and whether the KB leading acceptance bits already decide
// the draw.
//   (kf + 1) / 2^KB <= 1 - x^2 (0.19 t^2 + 1e-6)   <=>   kf <= 2^KB - 1 - 0.19 2^KB (x t)^2 - 2^KB 1e-6 x^2.  Box-Muller radii
//   stay below 5.89 (24-bit uniforms; 5.4 with the 20-bit uniforms of the v2 layout), so the last term is < 2^KB 3.5e-5: a
//   constant keeps the bound (conservatively) and the test needs (x t)^2 only -- two instructions fewer than forming t^2,
//   x^2 and the inner fma.   KB = 12: kf <= 4094.85 - 778.24 (x t)^2;   KB = 16: kf <= 65532.7 - 12451.84 (x t)^2.
//   The hot path holds xs = x / K (GammaState): (x t)^2 = 2 ln 2 (xs t)^2, folded into the slope (rounded away from zero).
template <int KB>
struct TryConst {
  static constexpr float slope = KB == 16 ? -12451.84f * 1.3862944f : -778.24f * 1.3862944f;
  static constexpr float top = KB == 16 ? 65532.7f : 4094.85f;
  static constexpr float kscale = KB == 16 ? 1.52587890625e-5f : 2.44140625e-4f;
};
//   x = the normal in units of K (xs), g from gamma_setup_hot
template <int KB = 12>
__device__ __forceinline__ float gamma_try(const GammaState& g, float x, float kf, bool& sure) {
#ifdef MFG_ABL_TRY
  sure = true;
  return 1.0f + 3.0f * g.c * x;
#endif
  const float t = g.c * x;
  const float q = x * t;
  const float thr = __builtin_fmaf(q * q, TryConst<KB>::slope, TryConst<KB>::top);
  sure = (__builtin_fabsf(t) <= 0.5f) && (kf <= thr);
  return 1.0f + t * (3.0f + t * (3.0f + t));
}
// The same, with the "not decided / small shape" condition returned as a WAVE MASK (bit = lane) straight from the compare
// instructions: the sampling loop only asks "any lane?", which is then a scalar test of the mask.  (A per-lane bool that is
// OR-ed over the pair and passed through the ballot builtin compiled to v_cndmask + v_cmp per pair on top of the scalar
// mask arithmetic: 1 VALU instruction per element.)  Unordered compares: a NaN anywhere sends the element to the cold path.
template <int KB>
__device__ __forceinline__ float gamma_try_mask(const GammaState& g, float x, float kf, uint64_t& cold) {
#ifdef MFG_ABL_TRY
  cold = 0;
  return 1.0f + 3.0f * g.c * x;
#endif
  const float t = g.c * x;
  const float q = x * t;
  const float thr = __builtin_fmaf(q * q, TryConst<KB>::slope, TryConst<KB>::top);
  constexpr int FCMP_OLT = 4, FCMP_UGT = 10;
  cold = __builtin_amdgcn_fcmpf(__builtin_fabsf(t), 0.5f, FCMP_UGT) | __builtin_amdgcn_fcmpf(kf, thr, FCMP_UGT) |
         __builtin_amdgcn_fcmpf(g.dd, 2.0f / 3.0f, FCMP_OLT);
  return 1.0f + t * (3.0f + t * (3.0f + t));
}
// Cold half: exact continuation and the shape < 1 boost; returns the variate.  xs = the hot path's normal (units of K), g from
// gamma_setup_hot: the true normal and the true c are restored here.
__device__ __forceinline__ float gamma_fix(const GammaState& g, float xs, float kf, float kscale, bool sure, float v,
                                           uint64_t seed, uint32_t elem, uint32_t step, uint64_t traj) {
  const float x = BM_K * xs;
  if (g.small) {
    // shape < 1: Gamma(a) = Gamma(a + 1) U^(1/a); the boosted draw runs the exact test from the start
    GammaState gb;
    gamma_setup(gb, g.a);
    v = gamma_exact_path(gb, x, kf, kscale, seed, elem, step, traj);
    const u32x4 rb = philox_elem(seed, elem, step, traj, 0xFFFFu);
    float y = gb.dd * v * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u01(rb.x)) * __builtin_amdgcn_rcpf(g.a));
    if (y == 0.0f) y = ZERO_GAMMA_REPLACEMENT;  // underflow of the boost (mfg_ac2.py:244)
    return y;
  }
  if (!sure) {
    GammaState gt = g;
    gt.c = g.c * (1.0f / BM_K);
    v = gamma_exact_path(gt, x, kf, kscale, seed, elem, step, traj);
  }
  return g.dd * v;
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
// alpha = log1p(e^z), sigmoid(z) for z = zh + zl (zl = low-order part of theta*x, so that e carries fp32
// relative error instead of |z| * 2^-24).  log1p as in softplus_sigmoid_e.
__device__ __forceinline__ void softplus_sigmoid_fast(float zh, float zl, float& sp, float& sg) {
  float e = fast_exp(zh);
  e = __builtin_fmaf(e, zl, e);
  const float u = 1.0f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  sg = e * r;
  sp = __builtin_fmaf(e - (u - 1.0f), r, fast_ln(u));  // log1p through the rounded sum + rounding-error correction (softplus_sigmoid_e)
}

// Round 2: the exponential is SEPARABLE, e^{theta (pi_j - pi_i - shift)} = E_j F_i with E_j = e^{theta (pi_j - 1/2)},
// F_i = e^{-theta (pi_i + shift - 1/2)}: the lane that owns state entry i evaluates E_i, F_i once per env step (fp64
// argument, hardware exp2 on its fp32 head, first-order correction for the tail: ~1e-7 relative) and an element costs ONE
// multiply instead of (hi/lo product, v_exp, correction).  Both factors are centred on pi = 1/2 (state entries lie in
// [0, 1]), so they stay inside the fp32 range while |theta| (1/2 + |shift|) <= 86.  There is NO per-element fall-back
// beyond that: the sampling kernels report MFG_STATUS_MIXED_RANGE (report_sep_range, include/mfg_hip.h) and their outputs
// are NaN; precision f64 has no limit.
constexpr double SEP_CENTRE = 0.5;
constexpr double SEP_LIMIT = 86.0;  // ln(fp32 max) = 88.7, ln(fp32 min normal) = -87.3; margin for the fp32 head / tail split
__device__ __forceinline__ void report_sep_range(unsigned* status, double theta, double shift) {
  // written as !(x <= limit) so that a NaN theta is reported too; one lane of the launch stores the bit (host-mapped word)
  if (status && blockIdx.x == 0 && threadIdx.x == 0 && !(fabs(theta) * (SEP_CENTRE + fabs(shift)) <= SEP_LIMIT))
    __hip_atomic_fetch_or(status, 1u /* MFG_STATUS_MIXED_RANGE */, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float exp_f64arg(double z) {
  const double zl2 = z * 1.4426950408889634;
  const float zh = (float)zl2;
  const float zt = (float)(zl2 - (double)zh);
  const float e = __builtin_amdgcn_exp2f(zh);
  return __builtin_fmaf(e, zt * 0.69314718055994531f, e);
}

// log1p(e) for 0 <= e <= 1/4: e * (degree-6 minimax polynomial of log1p(e)/e), 9.3e-8 relative in fp32 arithmetic
// (replaces the atanh series and its reciprocal).
__device__ __forceinline__ float log1p_small(float e) {
  float p = __builtin_fmaf(e, 0.0707516148686409f, -0.145447239279747f);
  p = __builtin_fmaf(e, p, 0.19665537774562836f);
  p = __builtin_fmaf(e, p, -0.24972063302993774f);
  p = __builtin_fmaf(e, p, 0.33332204818725586f);
  p = __builtin_fmaf(e, p, -0.4999998211860657f);
  p = __builtin_fmaf(e, p, 1.0f);
  return e * p;
}

// alpha = log1p(e), sigmoid = e / (1 + e) from e = e^z.
__device__ __forceinline__ void softplus_sigmoid_e(float e, float& sp, float& sg) {
#ifdef MFG_ABL_SETUP
  sg = e * 0.5f;
  sp = e + 0.3f;
  return;
#endif
#ifdef MFG_LOG1P_POLY  // the first form: degree-6 polynomial below 1/4, ln(1 + e) above, one select (6 more instructions)
  const float u = 1.0f + e;
  sg = e * __builtin_amdgcn_rcpf(u);
  float lp = log1p_small(e), ln = fast_ln(u);
  asm("" : "+v"(lp), "+v"(ln));  // both sides evaluated: the select is a v_cndmask, not an exec-masked if/else that would
                                 // cut the basic block (the sampling loop interleaves four elements' chains)
  sp = (e < 0.25f) ? lp : ln;
#else
  // log1p(e) = ln(u) + (e - (u - 1)) / u with u = fl32(1 + e): the hardware log of the ROUNDED sum plus the first-order
  // correction for what the rounding dropped (u - 1 is exact, so the numerator is the exact rounding error of 1 + e).
  // Branch-free over the whole range -- for e < 2^-24 it returns e -- and it shares the reciprocal with the sigmoid:
  // max relative error 1.95e-7 over e in [2^-24, 4] (tools/micro/log1p_check.hip), 8 instructions against 14.
  const float u = 1.0f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  sg = e * r;
  sp = __builtin_fmaf(e - (u - 1.0f), r, fast_ln(u));
#endif
}

// fp64 helpers of the mixed-precision per-row epilogue (one call per matrix ROW per step, but IEEE fp64 division / log
// expansions are ~30-45 instructions each at 4 cycles): reciprocal = v_rcp_f64 + two Newton steps; log = exponent * ln 2 +
// hardware log2 of the fp32 head of the mantissa + first-order tail (absolute error ~4e-8, against rows sums of 1e3-1e5).
__device__ __forceinline__ double fast_rcp_f64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_log_f64(double x) {
  const double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  const int ex = __builtin_amdgcn_frexp_exp(x);
  const float mh = (float)m;
  const float ml = (float)(m - (double)mh);
  const float lg2 = __builtin_amdgcn_logf(mh);
  return fma((double)ex + (double)lg2, 0.6931471805599453, (double)(ml * __builtin_amdgcn_rcpf(mh)));
}
// digamma for x > 0, mixed precision (|error| < 1e-9 + the fast log's 4e-8): asymptotic series from x >= 4 (six
// Bernoulli terms: the first omitted one is 3e-10 at x = 4), a FOUR-step recurrence below that.  Row sums A_i of the
// concentrations are 5 .. 15 at the reference policies, so the recurrence (and its reciprocal) is skipped wave-wide
// almost always; everything is a short fp64 chain (the per-row epilogue is latency bound).
__device__ __forceinline__ double digamma_pos_mixed(double x) {
  double corr = 0.0;
  if (x < 4.0) {
    // sum_{k<4} 1/(x+k) = q'(x)/q(x), q = x (x+1) (x+2) (x+3)
    const double a = x * (x + 3.0);          // x^2 + 3x
    const double q = a * (a + 2.0);          // (x^2+3x)(x^2+3x+2)
    const double qp = (2.0 * x + 3.0) * (2.0 * a + 2.0);
    corr = qp * fast_rcp_f64(q);
    x += 4.0;
  }
  const double inv = fast_rcp_f64(x);
  const double t = inv * inv;
  // t (1/12 - t/120 + t^2/252 - t^3/240 + t^4/132 - t^5 691/32760)
  const double s = t * (1.0 / 12.0 - t * (1.0 / 120.0 - t * (1.0 / 252.0 - t * (1.0 / 240.0 - t * (1.0 / 132.0 -
                   t * (691.0 / 32760.0))))));
  return fast_log_f64(x) - 0.5 * inv - s - corr;
}

// 1 / x as fp32 for an fp64 x (row normaliser): hardware reciprocal of the fp32 head + one Newton step, ~1e-7 relative.
__device__ __forceinline__ float fast_rcp_f32_of_f64(double x) {
  const float xf = (float)x;
  float r = __builtin_amdgcn_rcpf(xf);
  r = __builtin_fmaf(__builtin_fmaf(-xf, r, 1.0f), r, r);
  return r;
}

// Split-constant helper: z = theta * (pj - pi - shift) evaluated in fp32 with the rounding of the
// product and of the constants carried in zl.
struct ThetaSplit {
  float th, tl, sh, c0;  // theta = th + tl, shift = sh + sl, c0 = th * sl
  float thn;             // th * (h-table intervals per unit of z): the table index is one fma of x
};
__device__ __forceinline__ ThetaSplit theta_split(double theta, double shift) {
  ThetaSplit t;
  t.th = (float)theta;
  t.tl = (float)(theta - (double)t.th);
  t.sh = (float)shift;
  t.c0 = t.th * (float)(shift - (double)t.sh);
  t.thn = t.th * 16.0f;  // HTAB_PER_UNIT (a power of two: exact)
  return t;
}
__device__ __forceinline__ void theta_times_x(const ThetaSplit& t, float pj, float pi, float& x, float& zh, float& zl) {
  x = (pj - pi) - t.sh;
  zh = t.th * x;
  zl = __builtin_fmaf(t.th, x, -zh) + __builtin_fmaf(t.tl, x, -t.c0);
}

// ---------------------------------------------------------------------------
// (The table stores h / ln 2: the mixed kernels sum the per-element score terms  log2(y) alpha' - x h / ln 2  in log2
// units -- the hardware logarithm as it is, no multiply by ln 2 per element -- and scale the sum by ln 2 once per row.)
// h(z) = psi(softplus(z)) * sigmoid(z): the score's -psi(alpha_ij) alpha'_ij term equals -x_ij h(z_ij) with
// z = theta x, and h is a smooth bounded function of ONE variable (h -> -1 for z -> -inf, ~ln z for z -> inf).
// Mixed precision evaluates it from a table of per-interval cubics (fitted in fp64 by k_init_htab through
// 4 equispaced points of each interval, |error| < 1e-8 + fp32 rounding) instead of a 40-instruction
// digamma per matrix element: index + one 16-byte load + 3 FMAs.  The table lives in global memory (28 KB, of
// which a policy touches a few KB; L1/L2 resident): the kernels are VALU-issue bound, the load rides on the
// otherwise idle memory pipe.  It spans the whole fp32 range of e^z, so there is no out-of-table branch.
// ---------------------------------------------------------------------------
constexpr float HTAB_ZMIN = -24.0f;                        // h(z) = -1 + O(e^z) below
constexpr float HTAB_ZMAX = 88.0f;                         // fp32 range of e^z (the mixed mode's own limit)
constexpr int HTAB_PER_UNIT = 16;                          // intervals per unit of z
constexpr int HTAB_N = (88 + 24) * HTAB_PER_UNIT;          // 1 792 intervals over [-24, 88)

// h(theta x) from x and thn = theta * HTAB_PER_UNIT (ThetaSplit): the interval coordinate is ONE fma of x.
__device__ __forceinline__ float htab_eval(const float4* __restrict__ tab, float x, float thn) {
#ifdef MFG_ABL_HTAB
  return x * thn * 0.1f;
#endif
  static_assert(HTAB_PER_UNIT == 16, "ThetaSplit::thn assumes 16 intervals per unit of z");
  float t = __builtin_fmaf(x, thn, -HTAB_ZMIN * (float)HTAB_PER_UNIT);
  t = __builtin_amdgcn_fmed3f(t, 0.0f, (float)HTAB_N - 0.001f);  // clamp: one instruction
  const unsigned k = (unsigned)t;                                 // truncation == floor (t >= 0)
  const float f = __builtin_amdgcn_fractf(t);
  const float4 c = tab[k];
  return __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(c.w, f, c.z), f, c.y), f, c.x);
}

// ---------------------------------------------------------------------------
// wave-level reductions
// ---------------------------------------------------------------------------
// DPP move of a 64-bit value (two 32-bit DPP movs); lanes outside row_mask / without a source get 0.
// ROW_MASK == 0xF with a permutation inside rows: every lane has a source, so the "old" operand is dead; bound_ctrl with
// full masks lets the compiler drop it (a live "old" costs one extra v_mov per half: 36 instructions per row of the
// large-d kernels).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (ROW_MASK == 0xF) {
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);  // bound_ctrl + full masks: no tied "old" register
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  } else {
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  }
  return __hiloint2double(hi, lo);
}

// Wave-wide fp64 sum on the DPP path (no LDS-pipe ds_bpermute): butterfly inside each row of 16 lanes
// (quad_perm, row_half_mirror, row_mirror), then row_bcast:15 / row_bcast:31 accumulate the four row
// totals into lane 63, which is broadcast through an SGPR.  Fixed order => deterministic.
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_mov_f64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_mov_f64<0x140, 0xF>(v);  // row_mirror
  v += dpp_mov_f64<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_mov_f64<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}


// Three independent wave sums, step-interleaved: between a lane's write of a partial sum and the DPP read of it sit
// the other two chains, so the DPP read-after-write hazard costs no s_nop (18 per row in the large-d kernels).
__device__ __forceinline__ void wave_sum3_dpp(double& a, double& b, double& c) {
#define MFG_SUM3_STEP(CTRL, MASK)                 \
  {                                               \
    const double ta = dpp_mov_f64<CTRL, MASK>(a); \
    const double tb = dpp_mov_f64<CTRL, MASK>(b); \
    const double tc = dpp_mov_f64<CTRL, MASK>(c); \
    a += ta;                                      \
    b += tb;                                      \
    c += tc;                                      \
  }
  MFG_SUM3_STEP(0xB1, 0xF)
  MFG_SUM3_STEP(0x4E, 0xF)
  MFG_SUM3_STEP(0x141, 0xF)
  MFG_SUM3_STEP(0x140, 0xF)
  MFG_SUM3_STEP(0x142, 0xA)
  MFG_SUM3_STEP(0x143, 0xC)
#undef MFG_SUM3_STEP
  a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), 63), __builtin_amdgcn_readlane(__double2loint(a), 63));
  b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(b), 63), __builtin_amdgcn_readlane(__double2loint(b), 63));
  c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(c), 63), __builtin_amdgcn_readlane(__double2loint(c), 63));
}

// fp32 DPP move (building block of the fp32 row / wave sums: each butterfly step is ONE v_add_f32 with a DPP operand,
// against two moves and an fp64 add for a double).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_f32(float v) {
  const int x = __float_as_int(v);
  const int r = (ROW_MASK == 0xF) ? __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true)
                                  : __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xF, false);
  return __int_as_float(r);
}
// ---------------------------------------------------------------------------
// Transposed row sums (round 3; mixed mode of the wave-per-trajectory kernels).  The 64 lanes of a wave each hold a
// partial sum of KB different matrix rows (KB = 2, 4, 8; rows arrive in pairs).  Reducing every row on its own costs
// 6 DPP adds + a read-lane per sum; here the rows of a batch end up PACKED in one register and share the last steps:
//   * pair merge on lane bit 3 (row_pair_merge): lanes with the bit clear take a + ror8(a), the others b + ror8(b) -- the
//     second add is bank-masked, so two rows cost two instructions and leave one register;
//   * two plain butterfly steps (lane bits 2 and 0), then ONE select deposits the pair into the lanes whose bits (2, 0)
//     spell the pair's index in the batch (row_pair_deposit, mask in an SGPR pair);
//   * once per batch (row_batch_finish): lane bit 1, and bits 4 / 5 through v_permlane16_swap / v_permlane32_swap (gfx950).
// Afterwards lane l holds the total of the batch's row  bit3(l) + 2 bit0(l) + 4 bit2(l)  (masked to KB - 1).
// 5 instructions per sum per row pair + ~9 per batch, against 14 per pair.
// ---------------------------------------------------------------------------
template <int NS>
__device__ __forceinline__ void row_pair_merge(const float* a, const float* b, float* m) {
#pragma unroll
  for (int q = 0; q < NS; ++q) m[q] = a[q] + dpp_mov_f32<0x128, 0xF>(a[q]);  // row_ror:8
#pragma unroll
  for (int q = 0; q < NS; ++q)  // (the s_nop covers the VALU-write -> DPP-read distance the compiler cannot see into)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(m[q]) : "v"(b[q]));
#pragma unroll
  for (int q = 0; q < NS; ++q) m[q] += dpp_mov_f32<0x141, 0xF>(m[q]);  // row_half_mirror: lane bit 2 (and 0 <-> 1 swapped)
#pragma unroll
  for (int q = 0; q < NS; ++q) m[q] += dpp_mov_f32<0xB1, 0xF>(m[q]);   // quad_perm [1,0,3,2]: lane bit 0
}
// lanes of pair p (of KB / 2): bits (2, 0) of the lane index == p
template <int KB>
__device__ __forceinline__ uint64_t row_pair_mask(int p) {
  if constexpr (KB == 8) return 0x0505050505050505ull << ((p & 1) + ((p >> 1) & 1) * 4);
  else if constexpr (KB == 4) return 0x5555555555555555ull << (p & 1);
  else return ~0ull;
}
template <int NS>
__device__ __forceinline__ void row_pair_deposit(float* acc, const float* m, uint64_t mask) {
#pragma unroll
  for (int q = 0; q < NS; ++q) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(acc[q]) : "v"(m[q]), "s"(mask));
}
template <int NS>
__device__ __forceinline__ void row_batch_finish(float* x) {
#pragma unroll
  for (int q = 0; q < NS; ++q) x[q] += dpp_mov_f32<0x4E, 0xF>(x[q]);  // quad_perm [2,3,0,1]: lane bit 1
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[q]), __float_as_uint(x[q]), false, false);
    x[q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // lane bit 4
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[q]), __float_as_uint(x[q]), false, false);
    x[q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // lane bit 5
  }
}
// The three sums of a TD batch (S, A, D) finished TOGETHER: the swaps that fold lane bits 4 and 5 take TWO registers, so
// S and A share the bit-4 step and the pair shares the bit-5 step with D -- 10 instructions instead of 27 (the same swap
// with both operands equal needs a copy first), and ONE register comes back: lane l holds, for its row row_batch_row(l),
// S if bits (5, 4) are (0, 0), A if (0, 1), D if bit 5 is set.
__device__ __forceinline__ float row_batch_finish3(const float* x) {
  float s = x[0] + dpp_mov_f32<0x4E, 0xF>(x[0]);  // quad_perm [2,3,0,1]: lane bit 1
  float a = x[1] + dpp_mov_f32<0x4E, 0xF>(x[1]);
  float d = x[2] + dpp_mov_f32<0x4E, 0xF>(x[2]);
  const auto r4 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(a), false, false);
  const float sa = __uint_as_float(r4[0]) + __uint_as_float(r4[1]);  // even 16-lane rows: S over bit 4, odd rows: A
  const auto rd = __builtin_amdgcn_permlane16_swap(__float_as_uint(d), __float_as_uint(d), false, false);
  const float d4 = __uint_as_float(rd[0]) + __uint_as_float(rd[1]);
  const auto r5 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sa), __float_as_uint(d4), false, false);
  return __uint_as_float(r5[0]) + __uint_as_float(r5[1]);            // lower half: S / A over bit 5, upper half: D
}
// slot of the packed value in the per-row (A, D, S) triple, and whether this lane is the one that publishes it
__device__ __forceinline__ int row_batch_slot3(int lane) { return (lane & 32) ? 1 : ((lane & 16) ? 0 : 2); }
template <int KB>
__device__ __forceinline__ bool row_batch_owner3(int lane) {
  // bit 1 clear (one of the two copies), not the (bit 5, bit 4) = (1, 1) copy of D, index bits the batch does not use clear
  return (lane & 2) == 0 && (lane & 0x30) != 0x30 && (lane & (KB == 8 ? 0 : (KB == 4 ? 0x04 : 0x05))) == 0;
}
// in-batch row index k -> the first lane that holds its total after row_batch_finish, and a lane's row
__device__ __forceinline__ int row_batch_lane(int k) { return ((k & 1) << 3) | ((k >> 1) & 1) | (((k >> 2) & 1) << 2); }
template <int KB>
__device__ __forceinline__ int row_batch_row(int lane) {
  return (((lane >> 3) & 1) | ((lane & 1) << 1) | (((lane >> 2) & 1) << 2)) & (KB - 1);
}

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

// Row of the start-state table for a drawn index (mfg_ac2.py:466-469).  The reference draws randint(num_start_samples),
// so a valid index is always inside the table; an index from a stale / foreign draw is clamped instead of read out of
// bounds (the host classes keep num_start_samples in step with the table, see mfg_ac2.py mat_pi0 setter).
__host__ __device__ __forceinline__ int64_t start_row(int32_t idx, int64_t num_start) {
  const int64_t r = (int64_t)idx;
  return r < 0 ? 0 : (r >= num_start ? num_start - 1 : r);
}

// Start-state draw on the device (mfg_ac2.py:466, ac_irl.py:655: idx_row = randint(num_start_samples), one per trajectory
// and episode).  Batched runs draw it from the same counter-based generator as the actions, so no host RNG, no index
// upload and -- with several ranks -- no broadcast sits in front of an episode:
//   counter = (c0 = START_DRAW_ELEM, c1 = Philox step of the episode's FIRST env step, c2 | c3 = global trajectory id, draw
//   block 0); row = floor(x * num_start / 2^32) with x the first word of the block (Lemire's multiply-shift; the bias is
//   below num_start / 2^32).  START_DRAW_ELEM lies outside the element ids i*d + j < 512^2 of the action draws, so the
//   streams never collide; keyed by the GLOBAL trajectory id the draw does not depend on launch geometry or world size.
constexpr uint32_t START_DRAW_ELEM = 0xFFFFFFFFu;
__host__ __device__ __forceinline__ int64_t start_draw_row(uint64_t seed, uint32_t step, uint64_t traj, int64_t num_start) {
  u32x4 c;
  c.x = START_DRAW_ELEM;
  c.y = step;
  c.z = (uint32_t)traj;
  c.w = (uint32_t)(traj >> 32) & 0xFFFFu;
  const u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (int64_t)(((uint64_t)r.x * (uint64_t)num_start) >> 32);
}

// One parameter of the update  p += lr * G_k / count  (mfg_ac2.py:511-522 for the batch mean): ONE definition, used by the
// stand-alone update kernel, the update fused into the row reduction and the update folded into the next rollout's weight
// staging, so that every path produces the same bits.  `inv` = 1 / count.
__host__ __device__ __forceinline__ double updated_param(double p, double lr, double gk, double inv) { return fma(lr, gk * inv, p); }

// Column k of `nsb` partial rows [nsb][FO] of batch sums, added up by ONE wavefront in a fixed order: lane L adds rows L, L + 64,
// ... in row order (four loads in flight, unconditional, from clamped addresses), the 64 lane sums go through wave_sum_dpp; every
// lane returns the total.  ONE definition for the blocks that publish the update and for every sampling wave that forms the
// updated theta on its own (IRL env step, mfg_train_episode_irl): the same bits everywhere.
__device__ __forceinline__ double rows_column_sum(const double* __restrict__ rows, int nsb, int64_t FO, int64_t k, int lane) {
  double s = 0.0;
  for (int p0 = 0; p0 < nsb; p0 += 4 * WAVE) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * WAVE + lane;
      v[u] = rows[(int64_t)(p < nsb ? p : 0) * FO + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s += (p0 + u * WAVE + lane < nsb) ? v[u] : 0.0;
  }
  return wave_sum_dpp(s);
}

// Column walk of the transition / reward sums (pi'_j = sum_i pi_i P_ij and the two reward sums of column j): the d rows are added
// in groups of col_group_rows(d) consecutive rows (the last group may be shorter) -- every group starts from its own first row,
// the group sums are folded in group order.  d = 21: THREE groups of seven; d = 15: FOUR groups of 4, 4, 4, 3 -- so that the
// packed kernels (a lane walks all rows of its column) and the one-trajectory-per-wave kernel (k_core_row3: three / four lanes
// per column, one group each) share ONE summation tree: the results of a launch do not depend on which lane mapping it picked,
// i.e. on the batch a rank happens to hold (world-size invariance, tests/test_gpu_fullsize.py, tests/test_gpu_row3.py).
// Every other d: one group = the plain row order of rounds 1-5.
#ifdef MFG_COLGROUP_OFF  // developer builds only (A/B timing against the plain row order of rounds 1-5; other bits at d = 21 / 15)
__host__ __device__ constexpr int col_group_rows(int d) { return d > 0 ? d : 1; }
#else
__host__ __device__ constexpr int col_group_rows(int d) { return d == 21 ? 7 : (d == 15 ? 4 : (d > 0 ? d : 1)); }
#endif
// last row of a group: where the running sums of the walk are folded into the totals
__host__ __device__ constexpr bool col_group_end(int k, int d) { return k % col_group_rows(d) == col_group_rows(d) - 1 || k == d - 1; }
// One row of the walk folded into the running sums of its group (first = the group's first row): u = pi_i P_ij is exact in fp64.
__device__ __forceinline__ void col_walk_row(bool first, double u, double p, double& pa, double& p1, double& p2) {
  if (first) {
    pa = u;
    p1 = u * p;
    p2 = u * u;
  } else {
    pa += u;
    p1 = fma(u, p, p1);
    p2 = fma(u, u, p2);
  }
}

// k(i,j) for i <= j: row-major upper triangle (mfg_ac2.py:333).
__host__ __device__ __forceinline__ int feat_idx(int i, int j, int d) { return i * d - (i * (i - 1)) / 2 + (j - i); }

}  // namespace mfg
