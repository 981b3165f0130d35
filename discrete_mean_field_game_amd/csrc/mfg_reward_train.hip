// Reward-network TRAINING step of the max-ent IRL experiments on the device (gfx950).
//
// Replaces, for one call of AC_IRL.update_reward (reference ac_irl.py:804-846): the batch assembly from the Python lists
// of (state, action) pairs, `sess.run(r_train_op)` = forward of networks.py:46-81 over the sampled demonstration and
// generated transitions, the guided-cost-learning loss of ac_irl.py:390-413
//     L = -(1/N_demo) sum r_demo  +  log( (1/M) sum_traj exp( sum_t r_gen ) )  [+ l1_l2(fc3_w) + l1_l2(fc4_w)],
// its gradient, and one tf.train.AdamOptimizer step (ac_irl.py:417-418).
//
// The batch is tiny (5 + 5 trajectories x 15 transitions in the reference) and the step is latency bound, so the design
// minimises dependent launches and host work instead of per-sample throughput:
//   * the trajectories live in device-resident stores ([n,15,d] states, [n,15,d,d] actions); a batch is a list of store
//     rows passed BY VALUE in the kernel arguments -- no gather kernel, no index upload;
//   * launch 1 (k_rn_train_sample, one 256-thread block per transition): forward with every activation kept in LDS /
//     registers, then the backward pass of THAT sample for d r / d params.  The loss couples samples only through the scalar
//     dL/dr_n (the soft-max over the generated trajectories), and the gradient is linear in it:
//         dL/dp = sum_n c_n  d r_n / d p,   c_n = -1/N_demo (demo),  softmax_j(S)_traj(n) (generated),
//     so the per-sample Jacobian row needs no second pass over the network.  fc3_w (97 % of the parameters) is kept
//     factored: the block stores its fc3 input a2 [2 d^2] and dz3 [n3], not their outer product;
//   * launch 2 (k_rn_train_combine, one thread per parameter): c_n from the 150 rewards, the weighted sum over samples (for
//     fc3_w the small GEMM  sum_n (c_n dz3_n[k]) a2_n[i]), the l1_l2 gradient, and the Adam update in place; block 0 writes
//     [loss, first, second, reg] to a device slot that the host reads only when it prints them.
// Multi-GPU (replicated reward net): launch 2 stops at the gradient (MFG_RN_TRAIN_GRAD_ONLY), the caller
// all-reduces it and calls mfg_reward_net_adam.
//
// fp32 like the TF graph.  Dropout masks: the counter-based draw of the forward kernel (Philox key = seed, counter =
// (unit, layer 3 | 4, sample index n within the batch, block 0)), so the NumPy restatement can replay them.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfg_hip.h"
#include "mfg_core.h"
#include "mfg_rn_common.h"

namespace mfg {

constexpr int RT_BLOCK = 256, RT_WAVES = RT_BLOCK / WAVE;
constexpr int RT_MAX_TRAJ = MFG_RN_TRAIN_MAX_TRAJ;  // per batch half (demonstrations / generated)
constexpr int RT_MAXN = 32;                          // n3, n4
constexpr int RT_PP = 4;                             // pixels per thread, d <= 32
constexpr int RT_KC = 8;                             // fc3 units whose weight columns a thread holds in registers

struct RtLayout {
  int o_c1w, o_c1b, o_c2w, o_c2b, o_w3, o_b3, o_w4, o_b4, o_wo, o_bo, np;
  int a2;  // fc3 inputs = f2 d^2
  int ns;  // parameters outside fc3_w
};
__host__ __device__ inline RtLayout rt_layout(int d, int k1, int f2, int k2, int n3, int n4) {
  RtLayout L;
  L.a2 = f2 * d * d;
  L.o_c1w = 0;
  L.o_c1b = L.o_c1w + k1 * k1;
  L.o_c2w = L.o_c1b + 1;
  L.o_c2b = L.o_c2w + f2 * k2 * k2;
  L.o_w3 = L.o_c2b + f2;
  L.o_b3 = L.o_w3 + n3 * L.a2;
  L.o_w4 = L.o_b3 + n3;
  L.o_b4 = L.o_w4 + n4 * (n3 + d);
  L.o_wo = L.o_b4 + n4;
  L.o_bo = L.o_wo + n4;
  L.np = L.o_bo + 1;
  L.ns = L.np - n3 * L.a2;
  return L;
}
// index of parameter p (outside fc3_w) in a sample's small Jacobian row
__device__ __forceinline__ int rt_small(const RtLayout& L, int p) { return p < L.o_w3 ? p : p - (L.o_b3 - L.o_w3); }

struct RtArgs {
  const float* params;
  int d, k1, f2, k2, n3, n4;
  const float *demo_state, *demo_action, *gen_state, *gen_action;
  int steps, n_demo, n_gen;
  float keep_prob;
  int l1l2;
  uint64_t seed;
  float *r, *a2, *dz3, *js, *reg;  // workspace: [N], [N][a2], [N][n3], [N][ns], [1]
  int32_t rows[2 * RT_MAX_TRAJ];   // store rows of the batch: demonstrations, then generated
};

// ---------------------------------------------------------------------------------------------------------------------
// launch 1: one block per transition
// ---------------------------------------------------------------------------------------------------------------------
__device__ void rn_train_reg_block(const struct RtArgs& a, const RtLayout& L);

// K1 / K2 / F2 > 0: compile-time conv geometry (the reference's 5 / 3 / 2: taps unroll, weights come as scalar loads,
// the weight-gradient accumulators stay in registers); 0 = run-time geometry, tap by tap (any odd k <= 7, f2 <= 2).
// QA: fc3 inputs per thread (4 covers f2 d^2 <= 1024, i.e. d <= 22 with two filters; 8 everything up to d = 32)
template <int K1, int K2, int F2, int QA>
__global__ __launch_bounds__(RT_BLOCK) void k_rn_train_sample(RtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[RT_WAVES][64];
  __shared__ float s_h3[RT_MAXN], s_h4[RT_MAXN], s_dz3[RT_MAXN], s_dz4[RT_MAXN], s_state[32], s_dzo;
  const int d = a.d, dd = d * d, n3 = a.n3, n4 = a.n4;
  const int k1 = K1 ? K1 : a.k1, k2 = K2 ? K2 : a.k2, f2 = F2 ? F2 : a.f2;
  const int h1 = k1 / 2, h2 = k2 / 2, W1 = d + 2 * h1, W2 = d + 2 * h2;
  const RtLayout L = rt_layout(d, k1, f2, k2, n3, n4);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int n = blockIdx.x;
  if (n == (a.n_demo + a.n_gen) * a.steps) {  // the extra block
    rn_train_reg_block(a, L);
    return;
  }
  // which transition: trajectory j of the batch, step t
  const int j = n / a.steps, t = n - j * a.steps;
  const bool demo = j < a.n_demo;
  const int64_t tr = (int64_t)a.rows[demo ? j : RT_MAX_TRAJ + (j - a.n_demo)] * a.steps + t;
  const float* state = (demo ? a.demo_state : a.gen_state) + tr * d;
  const float* act = (demo ? a.demo_action : a.gen_action) + tr * dd;
  const float* P = a.params;
  // LDS carve: padded input | padded conv1 map | conv2 map (NHWC flat) | f2 padded dz2 maps | every weight except fc3_w
  float* tin = smem;
  float* a1p = tin + W1 * W1;
  float* a2s = a1p + W2 * W2;
  float* dz2p = a2s + L.a2;
  const int lds_n = W1 * W1 + W2 * W2 + L.a2 + f2 * W2 * W2;
  float* sw = smem + lds_n;  // small weights, indexed by rt_small(parameter)
  // Every global read of this block is issued HERE, before the first use (the step is a chain of short dependent phases:
  // a read issued where it is needed costs its full ~1 us latency every time -- the first version of this kernel spent 30 us
  // that way, 20 of them in the two fc3 loops and in fc4 reading weights one by one from L2):
  //   * the fc3 weight columns of this thread's inputs go to registers, RT_KC units at a time (kept for the backward pass
  //     when n3 <= RT_KC; their latency hides behind the convolutions);
  //   * all other weights (~180 floats) go to LDS.
  const float* W3 = P + L.o_w3;
  float w3r[RT_KC][QA];
#pragma unroll
  for (int kk = 0; kk < RT_KC; ++kk)
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int i = tid + q * RT_BLOCK;
      w3r[kk][q] = (kk < n3 && i < L.a2) ? W3[(int64_t)kk * L.a2 + i] : 0.0f;
    }
  for (int k = tid; k < L.ns; k += RT_BLOCK) sw[k] = P[k < L.o_w3 ? k : k + (L.o_b3 - L.o_w3)];
  for (int k = tid; k < lds_n; k += RT_BLOCK) smem[k] = 0.0f;
  if (tid < d) s_state[tid] = state[tid];
  int py[RT_PP], px[RT_PP];
  float actv[RT_PP];
#pragma unroll
  for (int q = 0; q < RT_PP; ++q) {
    const int p = tid + q * RT_BLOCK;
    const int pc = p < dd ? p : 0;
    py[q] = pc / d;
    px[q] = pc - py[q] * d;
    actv[q] = p < dd ? act[p] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RT_PP; ++q)
    if (tid + q * RT_BLOCK < dd) tin[(py[q] + h1) * W1 + px[q] + h1] = actv[q];
  __syncthreads();
  // weights of the convolutions: compile-time geometry -> scalar loads from read-only memory (uniform addresses, no vector or
  // LDS instruction per tap); run-time geometry -> broadcast reads of the LDS copy
  RnConstF c1w_g = (RnConstF)(P + L.o_c1w), c2w_g = (RnConstF)(P + L.o_c2w);
  const float* c1w_l = sw + L.o_c1w;
  const float* c2w_l = sw + L.o_c2w;
  auto c1w = [&](int k) __attribute__((always_inline)) { return K1 ? c1w_g[k] : c1w_l[k]; };
  auto c2w = [&](int k) __attribute__((always_inline)) { return K1 ? c2w_g[k] : c2w_l[k]; };
  // ---- forward: conv1 + ReLU
  float a1v[RT_PP];
#pragma unroll
  for (int q = 0; q < RT_PP; ++q) {
    a1v[q] = 0.0f;
    if (tid + q * RT_BLOCK < dd) {
      float s = sw[L.o_c1b];
      const float* tp = tin + py[q] * W1 + px[q];
#pragma unroll
      for (int dy = 0; dy < (K1 ? K1 : k1); ++dy)
#pragma unroll
        for (int dx = 0; dx < (K1 ? K1 : k1); ++dx) s = fmaf(tp[dy * W1 + dx], c1w(dy * k1 + dx), s);
      a1v[q] = fmaxf(s, 0.0f);
      a1p[(py[q] + h2) * W2 + px[q] + h2] = a1v[q];
    }
  }
  __syncthreads();
  // ---- conv2 + ReLU -> NHWC flat
#pragma unroll
  for (int q = 0; q < RT_PP; ++q) {
    const int p = tid + q * RT_BLOCK;
    if (p < dd) {
      const float* tp = a1p + py[q] * W2 + px[q];
#pragma unroll
      for (int c = 0; c < (F2 ? F2 : 2); ++c) {
        if (c < f2) {
          float s = sw[L.o_c2b + c];
#pragma unroll
          for (int dy = 0; dy < (K2 ? K2 : k2); ++dy)
#pragma unroll
            for (int dx = 0; dx < (K2 ? K2 : k2); ++dx)
              s = fmaf(tp[dy * W2 + dx], c2w(c * k2 * k2 + dy * k2 + dx), s);
          a2s[p * f2 + c] = fmaxf(s, 0.0f);
        }
      }
    }
  }
  __syncthreads();
  // ---- fc3: thread owns inputs i = tid + 256 q
  float a2v[QA];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int i = tid + q * RT_BLOCK;
    a2v[q] = i < L.a2 ? a2s[i] : 0.0f;
  }
  // units in chunks of RT_KC: chunk 0 is already in registers; the sums of a chunk are reduced together (independent chains)
  for (int k0 = 0; k0 < n3; k0 += RT_KC) {
    float part[RT_KC];
#pragma unroll
    for (int kk = 0; kk < RT_KC; ++kk) {
      float sacc = 0.0f;
      if (k0 == 0) {
#pragma unroll
        for (int q = 0; q < QA; ++q) sacc = fmaf(a2v[q], w3r[kk][q], sacc);
      } else if (k0 + kk < n3) {
        float wv[QA];
#pragma unroll
        for (int q = 0; q < QA; ++q) {
          const int i = tid + q * RT_BLOCK;
          wv[q] = i < L.a2 ? W3[(int64_t)(k0 + kk) * L.a2 + i] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < QA; ++q) sacc = fmaf(a2v[q], wv[q], sacc);
      }
      part[kk] = sacc;
    }
    // wave sums on the DPP path, four at a time (24 adds per group; a shuffle tree is six LDS-pipe permutes per sum)
#pragma unroll
    for (int g4 = 0; g4 < RT_KC; g4 += 4) {
      float p4[4] = {part[g4], part[g4 + 1], part[g4 + 2], part[g4 + 3]};
      wave_sum4_to_lane63(p4);
      if (lane == 63) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k0 + g4 + u < n3) red[wv][k0 + g4 + u] = p4[u];
      }
    }
  }
  __syncthreads();
  const bool drop = a.keep_prob < 1.0f;
  const float inv_keep = drop ? 1.0f / a.keep_prob : 1.0f;
  if (tid < n3) {
    float z = sw[rt_small(L, L.o_b3 + tid)];
#pragma unroll
    for (int w = 0; w < RT_WAVES; ++w) z += red[w][tid];
    float h = fmaxf(z, 0.0f);
    if (drop) {
      const u32x4 r = philox_elem(a.seed, (uint32_t)tid, 3u, (uint64_t)n, 0);
      h = (u01(r.x) <= a.keep_prob) ? h * inv_keep : 0.0f;
    }
    s_h3[tid] = h;
  }
  __syncthreads();
  // ---- fc4 over [h3, state] (networks.py:72)
  const int in4 = n3 + d;
  if (tid < n4) {
    const float* w = sw + rt_small(L, L.o_w4) + tid * in4;
    float z = sw[rt_small(L, L.o_b4 + tid)];
    for (int k = 0; k < n3; ++k) z = fmaf(s_h3[k], w[k], z);
    for (int k = 0; k < d; ++k) z = fmaf(s_state[k], w[n3 + k], z);
    float h = fmaxf(z, 0.0f);
    if (drop) {
      const u32x4 r = philox_elem(a.seed, (uint32_t)tid, 4u, (uint64_t)n, 0);
      h = (u01(r.x) <= a.keep_prob) ? h * inv_keep : 0.0f;
    }
    s_h4[tid] = h;
  }
  __syncthreads();
  float* js = a.js + (int64_t)n * L.ns;
  if (tid == 0) {
    float z = sw[rt_small(L, L.o_bo)];
    for (int m = 0; m < n4; ++m) z = fmaf(s_h4[m], sw[rt_small(L, L.o_wo) + m], z);
    const float r = tanhf(z);
    a.r[n] = r;
    const float dzo = 1.0f - r * r;  // d r / d z_out
    s_dzo = dzo;
    js[rt_small(L, L.o_bo)] = dzo;
  }
  __syncthreads();
  // ---- backward of THIS sample's reward (dL/dr = 1; the combine kernel scales the row)
  const float dzo = s_dzo;
  if (tid < n4) {
    const float dz = s_h4[tid] > 0.0f ? dzo * sw[rt_small(L, L.o_wo) + tid] * inv_keep : 0.0f;  // h4 > 0 <=> pre-activation > 0 and unit kept
    s_dz4[tid] = dz;
    js[rt_small(L, L.o_wo + tid)] = dzo * s_h4[tid];
    js[rt_small(L, L.o_b4 + tid)] = dz;
  }
  __syncthreads();
  for (int e = tid; e < n4 * in4; e += RT_BLOCK) {
    const int m = e / in4, k = e - m * in4;
    js[rt_small(L, L.o_w4 + e)] = s_dz4[m] * (k < n3 ? s_h3[k] : s_state[k - n3]);
  }
  if (tid < n3) {
    float dh = 0.0f;
    for (int m = 0; m < n4; ++m) dh = fmaf(s_dz4[m], sw[rt_small(L, L.o_w4) + m * in4 + tid], dh);
    const float dz = s_h3[tid] > 0.0f ? dh * inv_keep : 0.0f;
    s_dz3[tid] = dz;
    js[rt_small(L, L.o_b3 + tid)] = dz;
    a.dz3[(int64_t)n * n3 + tid] = dz;
  }
  __syncthreads();
  // ---- fc3 backward: d a2, stored factored (a2, dz3) for the combine kernel; dz2 into the padded maps
  float gb2[2] = {0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int i = tid + q * RT_BLOCK;
    if (i < L.a2) {
      float da = 0.0f;
#pragma unroll
      for (int kk = 0; kk < RT_KC; ++kk) da = fmaf(kk < n3 ? s_dz3[kk] : 0.0f, w3r[kk][q], da);
      for (int k0 = RT_KC; k0 < n3; k0 += RT_KC) {   // (n3 > RT_KC: re-read, RT_KC loads in flight)
        float wv[RT_KC];
#pragma unroll
        for (int kk = 0; kk < RT_KC; ++kk) wv[kk] = k0 + kk < n3 ? W3[(int64_t)(k0 + kk) * L.a2 + i] : 0.0f;
#pragma unroll
        for (int kk = 0; kk < RT_KC; ++kk) da = fmaf(k0 + kk < n3 ? s_dz3[k0 + kk] : 0.0f, wv[kk], da);
      }
      a.a2[(int64_t)n * L.a2 + i] = a2v[q];
      const float dz = a2v[q] > 0.0f ? da : 0.0f;
      const int pix = i / f2, c = i - pix * f2;
      const int y = pix / d, x = pix - y * d;
      dz2p[c * W2 * W2 + (y + h2) * W2 + x + h2] = dz;
      gb2[c] += dz;
    }
  }
  __syncthreads();
  // ---- conv2 weight gradients, d a1, conv1 weight gradients
  if constexpr (K1 > 0) {
    constexpr int NW2 = F2 * K2 * K2, NW1 = K1 * K1, NACC = NW2 + F2 + NW1 + 1;
    static_assert(NACC <= 64, "reduction scratch");
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0f;
#pragma unroll
    for (int c = 0; c < F2; ++c) acc[NW2 + c] = gb2[c];
#pragma unroll
    for (int q = 0; q < RT_PP; ++q) {
      if (tid + q * RT_BLOCK < dd) {
        const int y = py[q], x = px[q];
        float da1 = 0.0f;
#pragma unroll
        for (int c = 0; c < F2; ++c) {
          const float* zc = dz2p + c * W2 * W2;
          const float dzc = zc[(y + h2) * W2 + x + h2];
          const float* ap = a1p + y * W2 + x;
#pragma unroll
          for (int dy = 0; dy < K2; ++dy)
#pragma unroll
            for (int dx = 0; dx < K2; ++dx) {
              acc[c * K2 * K2 + dy * K2 + dx] = fmaf(dzc, ap[dy * W2 + dx], acc[c * K2 * K2 + dy * K2 + dx]);
              // a1[y,x] feeds a2[c, y - dy + h2, x - dx + h2] through tap (dy, dx)
              da1 = fmaf(c2w(c * K2 * K2 + dy * K2 + dx), zc[(y - dy + 2 * h2) * W2 + (x - dx + 2 * h2)], da1);
            }
        }
        const float dz1 = a1v[q] > 0.0f ? da1 : 0.0f;
        const float* tp = tin + y * W1 + x;
#pragma unroll
        for (int dy = 0; dy < K1; ++dy)
#pragma unroll
          for (int dx = 0; dx < K1; ++dx) acc[NW2 + F2 + dy * K1 + dx] = fmaf(dz1, tp[dy * W1 + dx], acc[NW2 + F2 + dy * K1 + dx]);
        acc[NW2 + F2 + NW1] += dz1;
      }
    }
#pragma unroll
    for (int k = 0; k < NACC; k += 4) {
      float p4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p4[u] = k + u < NACC ? acc[k + u] : 0.0f;
      wave_sum4_to_lane63(p4);
      if (lane == 63) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k + u < NACC) red[wv][k + u] = p4[u];
      }
    }
    __syncthreads();
    if (tid < NACC) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < RT_WAVES; ++w) s += red[w][tid];
      // acc order: conv2_w | conv2_b | conv1_w | conv1_b  ->  parameter order conv1_w | conv1_b | conv2_w | conv2_b
      const int p = tid < NW2 + F2 ? L.o_c2w + tid : L.o_c1w + (tid - NW2 - F2);
      js[rt_small(L, p)] = s;
    }
  } else {
    // run-time geometry: one block reduction per tap
    float dz1v[RT_PP];
#pragma unroll
    for (int q = 0; q < RT_PP; ++q) {
      dz1v[q] = 0.0f;
      if (tid + q * RT_BLOCK < dd) {
        const int y = py[q], x = px[q];
        float da1 = 0.0f;
        for (int c = 0; c < f2; ++c)
          for (int dy = 0; dy < k2; ++dy)
            for (int dx = 0; dx < k2; ++dx)
              da1 = fmaf(c2w(c * k2 * k2 + dy * k2 + dx),
                         dz2p[c * W2 * W2 + (y - dy + 2 * h2) * W2 + (x - dx + 2 * h2)], da1);
        dz1v[q] = a1v[q] > 0.0f ? da1 : 0.0f;
      }
    }
    const int nw2 = f2 * k2 * k2, nw1 = k1 * k1, ntap = nw2 + f2 + nw1 + 1;
    for (int e0 = 0; e0 < ntap; e0 += 64) {
      const int ne = ntap - e0 < 64 ? ntap - e0 : 64;
      for (int ee = 0; ee < ne; ++ee) {
        const int e = e0 + ee;
        float s = 0.0f;
        if (e >= nw2 && e < nw2 + f2) s = gb2[e - nw2];
        else {
#pragma unroll
          for (int q = 0; q < RT_PP; ++q) {
            if (tid + q * RT_BLOCK < dd) {
              const int y = py[q], x = px[q];
              if (e < nw2) {
                const int c = e / (k2 * k2), tp = e - c * k2 * k2, dy = tp / k2, dx = tp - dy * k2;
                s = fmaf(dz2p[c * W2 * W2 + (y + h2) * W2 + x + h2], a1p[(y + dy) * W2 + x + dx], s);
              } else if (e < nw2 + f2 + nw1) {
                const int tp = e - nw2 - f2, dy = tp / k1, dx = tp - dy * k1;
                s = fmaf(dz1v[q], tin[(y + dy) * W1 + x + dx], s);
              } else s += dz1v[q];
            }
          }
        }
        s = wave_sum(s);
        if (lane == 0) red[wv][ee] = s;
      }
      __syncthreads();
      if (tid < ne) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < RT_WAVES; ++w) s += red[w][tid];
        const int e = e0 + tid;
        const int p = e < nw2 + f2 ? L.o_c2w + e : L.o_c1w + (e - nw2 - f2);
        js[rt_small(L, p)] = s;
      }
      __syncthreads();
    }
  }
}

// the regulariser's value l1_l2(fc3_w) + l1_l2(fc4_w) (weights are read-only in launch 1): its own block, next to the samples
__device__ void rn_train_reg_block(const RtArgs& a, const RtLayout& L) {
  __shared__ float redr[RT_WAVES];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const float* P = a.params;
  float s = 0.0f;
  if (a.l1l2) {
    for (int p = L.o_w3 + tid; p < L.o_b3; p += RT_BLOCK) s += fabsf(P[p]) + 0.5f * P[p] * P[p];
    for (int p = L.o_w4 + tid; p < L.o_b4; p += RT_BLOCK) s += fabsf(P[p]) + 0.5f * P[p] * P[p];
  }
  s = wave_sum(s);
  if (lane == 0) redr[wv] = s;
  __syncthreads();
  if (tid == 0) a.reg[0] = redr[0] + redr[1] + redr[2] + redr[3];
}

// ---------------------------------------------------------------------------------------------------------------------
// launch 2: coefficients, gradient, Adam
// ---------------------------------------------------------------------------------------------------------------------
struct RtCombineArgs {
  float* params;
  float *m, *v;  // Adam moments [np]
  float* grad;   // [np] out (may be NULL when the update is applied here)
  float* stats;  // [4] loss, first, second, reg (may be NULL)
  const float *r, *a2, *dz3, *js, *reg;
  int d, k1, f2, k2, n3, n4;
  int steps, n_demo, n_gen;
  float demo_scale;  // 1 / num_demo_samples (ac_irl.py:390)
  int l1l2, apply;
  float lr_t, beta1, beta2, eps;  // lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t)  (tf.train.AdamOptimizer)
};

__device__ __forceinline__ float adam_param(float p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
  m = fmaf(b1, m, (1.0f - b1) * g);
  v = fmaf(b2, v, (1.0f - b2) * g * g);
  return p - lr_t * m / (sqrtf(v) + eps);
}

// block = RT_CP consecutive parameters x RT_WAVES sample slices: wave s sums the samples n = s, s + 4, ... of the block's 64
// parameters (coalesced: lane = parameter), RT_CU loads in flight per lane (the first round issued before anything else); the four slice sums meet in LDS and are added in
// slice order (fixed association: bit-reproducible).  The first version ran one thread per parameter over all 150 samples
// with a load and an integer division per iteration: 30 us, all of it L2 latency.
constexpr int RT_CP = WAVE, RT_CU = 38, RT_DZ = 8;  // RT_CU: one round covers 4 x 38 = 152 samples (the reference batch is 150; a second round of reads cost 4 us)
__global__ __launch_bounds__(RT_BLOCK) void k_rn_train_combine(RtCombineArgs a) {
  __shared__ float s_c[2 * RT_MAX_TRAJ];  // dL/dr per trajectory: demonstrations, then generated
  __shared__ float s_S[RT_MAX_TRAJ];
  __shared__ float s_stat[2];
  __shared__ double s_part[RT_WAVES][RT_CP];
  extern __shared__ __attribute__((aligned(16))) float s_cn[];  // [N] dL/dr per sample | [N][n3] c_n dz3_n
  const RtLayout L = rt_layout(a.d, a.k1, a.f2, a.k2, a.n3, a.n4);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int nd = a.n_demo, ng = a.n_gen, T = a.steps, n3 = a.n3;
  const int N = (nd + ng) * T;
  float* s_cdz = s_cn + N;
  const int p0 = blockIdx.x * RT_CP;
  const bool any_w3 = p0 + RT_CP > L.o_w3 && p0 < L.o_b3;
  const int p = p0 + lane;
  const bool live = p < L.np, w3 = live && p >= L.o_w3 && p < L.o_b3;
  const int k3 = w3 ? (p - L.o_w3) / L.a2 : 0;
  const float* src = w3 ? a.a2 + ((p - L.o_w3) - k3 * L.a2) : a.js + (live ? rt_small(L, p) : 0);
  const int64_t stride = w3 ? L.a2 : L.ns;
  // Every read that does not depend on the coefficients is issued HERE, before the coefficient phase waits for the rewards:
  // the rewards, the block's share of dz3, and the FIRST round of this lane's Jacobian column (the kernel is a latency chain:
  // rewards -> soft-max -> weighted sums -> Adam; 18 us in the version that started the column reads after the soft-max).
  // (UNCONDITIONAL loads from clamped addresses, masked where they are used: a load under a lane predicate compiles to a
  //  branch with a full wait behind it -- the first version of this block issued its 38 column reads one after the other)
  float r_mine[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int n = tid + u * RT_BLOCK;
    r_mine[u] = a.r[n < N ? n : 0];
  }
  float dz_mine[RT_DZ];
#pragma unroll
  for (int u = 0; u < RT_DZ; ++u) {
    const int e = tid + u * RT_BLOCK;
    dz_mine[u] = a.dz3[e < N * n3 ? e : 0];
  }
  float x0[RT_CU];
#pragma unroll
  for (int u = 0; u < RT_CU; ++u) {
    const int n = wv + u * RT_WAVES;
    x0[u] = src[(int64_t)(n < N ? n : 0) * stride];
  }
  const int pc = live ? p : 0;
  const float w_old = a.params[pc];
  const float m_old = a.m ? a.m[pc] : 0.0f, v_old = a.v ? a.v[pc] : 0.0f;   // (uniform pointers: scalar branches)
  // S_j = sum_t r[j, t] in step order (fixed association per trajectory), the first term's sum on another wave
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (tid + u * RT_BLOCK < N) s_cn[tid + u * RT_BLOCK] = r_mine[u];
  __syncthreads();
  if (tid < ng) {
    float sacc = 0.0f;
    for (int t = 0; t < T; ++t) sacc += s_cn[(nd + tid) * T + t];
    s_S[tid] = sacc;
  }
  if (tid < nd) s_c[tid] = -a.demo_scale;
  if (wv == 1) {  // first term: sum of the demonstration rewards, one wave (lane-strided partial sums in index order + a DPP tree)
    float sd = 0.0f;
    for (int n = lane; n < nd * T; n += WAVE) sd += s_cn[n];
    sd = wave_sum_f32_dpp(sd);
    if (lane == 0) s_stat[1] = sd;
  }
  __syncthreads();
  if (wv == 0) {  // soft-max over the generated trajectories (<= 64: one per lane) and log-mean-exp, wave-parallel
    const float Sj = lane < ng ? s_S[lane] : -INFINITY;
    float mx = Sj;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, WAVE));
    const float ej = lane < ng ? expf(Sj - mx) : 0.0f;
    const float z = wave_sum_f32_dpp(ej);
    if (lane < ng) s_c[nd + lane] = ej / z;
    if (lane == 0) s_stat[0] = ng ? mx + logf(z / (float)ng) : 0.0f;  // = log( 1/M sum exp S_j )
  }
  __syncthreads();
  // per-sample coefficient, and c_n dz3_n for the factored fc3_w gradient
  for (int n = tid; n < N; n += RT_BLOCK) s_cn[n] = s_c[n / T];
  __syncthreads();
  if (any_w3) {
#pragma unroll
    for (int u = 0; u < RT_DZ; ++u) {
      const int e = tid + u * RT_BLOCK;
      if (e < N * n3) s_cdz[e] = s_cn[e / n3] * dz_mine[u];
    }
    for (int e = tid + RT_DZ * RT_BLOCK; e < N * n3; e += RT_BLOCK) s_cdz[e] = s_cn[e / n3] * a.dz3[e];  // (batches beyond 256 x RT_DZ entries)
    __syncthreads();
  }
  // the sum over the batch runs in fp64: its terms (demonstrations -, generated +) cancel to a small net value, and an
  // fp32 running sum would leave ~1e-7 of the LARGEST partial sum in it.  Wave s owns the samples s, s + 4, ... of the
  // block's 64 parameters; the four slice sums are added in slice order (fixed association: bit-reproducible).
  double gs = 0.0;
  if (live) {
#pragma unroll
    for (int u = 0; u < RT_CU; ++u) {
      const int n = wv + u * RT_WAVES;
      if (n < N) gs = fma((double)(w3 ? s_cdz[n * n3 + k3] : s_cn[n]), (double)x0[u], gs);
    }
    for (int n0 = wv + RT_CU * RT_WAVES; n0 < N; n0 += RT_WAVES * RT_CU) {
      float x[RT_CU];
#pragma unroll
      for (int u = 0; u < RT_CU; ++u) {
        const int n = n0 + u * RT_WAVES;
        x[u] = src[(int64_t)(n < N ? n : 0) * stride];
      }
#pragma unroll
      for (int u = 0; u < RT_CU; ++u) {
        const int n = n0 + u * RT_WAVES;
        if (n < N) gs = fma((double)(w3 ? s_cdz[n * n3 + k3] : s_cn[n]), (double)x[u], gs);
      }
    }
  }
  s_part[wv][lane] = gs;
  __syncthreads();
  if (wv == 0 && live) {
    gs = ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
    const float w = w_old;
    if (a.l1l2 && ((p >= L.o_w3 && p < L.o_b3) || (p >= L.o_w4 && p < L.o_b4)))
      gs += (double)((w > 0.0f ? 1.0f : (w < 0.0f ? -1.0f : 0.0f)) + w);  // d/dw (|w| + w^2 / 2)
    const float g = (float)gs;
    if (a.grad) a.grad[p] = g;
    if (a.apply) {
      float m = m_old, v = v_old;
      a.params[p] = adam_param(w, g, m, v, a.lr_t, a.beta1, a.beta2, a.eps);
      a.m[p] = m;
      a.v[p] = v;
    }
  }
  if (blockIdx.x == 0 && tid == 0 && a.stats) {
    const float first = -a.demo_scale * s_stat[1], second = s_stat[0], reg = a.l1l2 ? a.reg[0] : 0.0f;
    a.stats[0] = first + second + reg;
    a.stats[1] = first;
    a.stats[2] = second;
    a.stats[3] = reg;
  }
}

__global__ __launch_bounds__(RT_BLOCK) void k_rn_adam(float* params, float* m, float* v, const float* grad, int64_t n, float lr_t,
                                                      float b1, float b2, float eps) {
  const int64_t p = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
  if (p < n) {
    float mm = m[p], vv = v[p];
    params[p] = adam_param(params[p], grad[p], mm, vv, lr_t, b1, b2, eps);
    m[p] = mm;
    v[p] = vv;
  }
}

static bool rt_shape_ok(int d, int k1, int f2, int k2, int n3, int n4) {
  return d >= 1 && d <= 32 && f2 >= 1 && f2 <= 2 && n3 >= 1 && n3 <= RT_MAXN && n4 >= 1 && n4 <= RT_MAXN && (k1 & 1) && (k2 & 1) &&
         k1 >= 1 && k1 <= 7 && k2 >= 1 && k2 <= 7;
}
static float adam_lr_t(double lr, double b1, double b2, int64_t step) {
  return (float)(lr * sqrt(1.0 - pow(b2, (double)step)) / (1.0 - pow(b1, (double)step)));
}

}  // namespace mfg

using namespace mfg;

extern "C" {

int64_t mfg_reward_net_num_params(int d, int k1, int f2, int k2, int n3, int n4) {
  if (d < 1 || k1 < 1 || f2 < 1 || k2 < 1 || n3 < 1 || n4 < 1) return -1;
  return rt_layout(d, k1, f2, k2, n3, n4).np;
}

int mfg_reward_net_param_offsets(int d, int k1, int f2, int k2, int n3, int n4, int64_t* offsets_host) {
  if (!offsets_host || d < 1 || k1 < 1 || f2 < 1 || k2 < 1 || n3 < 1 || n4 < 1)
    return set_error(MFG_EINVAL, "reward_net_param_offsets: bad argument");
  const RtLayout L = rt_layout(d, k1, f2, k2, n3, n4);
  const int o[11] = {L.o_c1w, L.o_c1b, L.o_c2w, L.o_c2b, L.o_w3, L.o_b3, L.o_w4, L.o_b4, L.o_wo, L.o_bo, L.np};
  for (int k = 0; k < 11; ++k) offsets_host[k] = o[k];
  return MFG_OK;
}

size_t mfg_reward_net_train_workspace_bytes(int d, int k1, int f2, int k2, int n3, int n4, int64_t n_transitions) {
  if (d < 1 || k1 < 1 || f2 < 1 || k2 < 1 || n3 < 1 || n4 < 1 || n_transitions < 0) return 0;
  const RtLayout L = rt_layout(d, k1, f2, k2, n3, n4);
  return (size_t)(n_transitions * (1 + L.a2 + n3 + L.ns) + 4) * sizeof(float);
}

int mfg_reward_net_train_step(float* params, float* adam_m, float* adam_v, int d, int k1, int f2, int k2, int n3, int n4,
                              const float* demo_state, const float* demo_action, const int32_t* demo_rows_host, int n_demo,
                              const float* gen_state, const float* gen_action, const int32_t* gen_rows_host, int n_gen, int steps,
                              int demo_divisor, float keep_prob, int l1l2, uint64_t seed, double lr, double beta1, double beta2,
                              double eps, int64_t adam_step, int flags, float* grad, float* stats, void* workspace,
                              size_t workspace_bytes, mfg_stream_t stream) {
  if (!params || !workspace || n_demo < 0 || n_gen < 0 || steps < 1 || demo_divisor < 1 || (n_demo && (!demo_state || !demo_action || !demo_rows_host)) ||
      (n_gen && (!gen_state || !gen_action || !gen_rows_host)))
    return set_error(MFG_EINVAL, "reward_net_train_step: null pointer / bad count");
  const bool apply = !(flags & MFG_RN_TRAIN_GRAD_ONLY);
  if (apply && (!adam_m || !adam_v || adam_step < 1)) return set_error(MFG_EINVAL, "reward_net_train_step: Adam state missing / step < 1");
  if (!apply && !grad) return set_error(MFG_EINVAL, "reward_net_train_step: MFG_RN_TRAIN_GRAD_ONLY needs grad");
  if (!(keep_prob > 0.0f && keep_prob <= 1.0f)) return set_error(MFG_EINVAL, "reward_net_train_step: keep_prob must be in (0,1]");
  if (!rt_shape_ok(d, k1, f2, k2, n3, n4))
    return set_error(MFG_EUNSUPPORTED, "reward_net_train_step: supported d <= 32, f2 <= 2, n_fc <= 32, odd kernels <= 7");
  if (n_demo > RT_MAX_TRAJ || n_gen > RT_MAX_TRAJ)
    return set_error(MFG_EUNSUPPORTED, "reward_net_train_step: at most MFG_RN_TRAIN_MAX_TRAJ trajectories per batch half");
  const int64_t N = (int64_t)(n_demo + n_gen) * steps;
  if (N == 0) return set_error(MFG_EINVAL, "reward_net_train_step: empty batch");
  // the combine kernel keeps one coefficient per transition and c_n dz3_n [N][n3] in LDS and eight rewards per thread in registers
  if (N > 8 * RT_BLOCK || (size_t)N * (size_t)(1 + n3) * sizeof(float) > 60 * 1024)
    return set_error(MFG_EUNSUPPORTED, "reward_net_train_step: batch too large ((n_demo + n_gen) * steps <= 2048 and * (1 + n_fc3) * 4 B <= 60 KB)");
  if (workspace_bytes < mfg_reward_net_train_workspace_bytes(d, k1, f2, k2, n3, n4, N))
    return set_error(MFG_EWORKSPACE, "reward_net_train_step: workspace too small (mfg_reward_net_train_workspace_bytes)");
  const RtLayout L = rt_layout(d, k1, f2, k2, n3, n4);
  RtArgs a{};
  a.params = params;
  a.d = d; a.k1 = k1; a.f2 = f2; a.k2 = k2; a.n3 = n3; a.n4 = n4;
  a.demo_state = demo_state; a.demo_action = demo_action; a.gen_state = gen_state; a.gen_action = gen_action;
  a.steps = steps; a.n_demo = n_demo; a.n_gen = n_gen;
  a.keep_prob = keep_prob;
  a.l1l2 = l1l2 ? 1 : 0;
  a.seed = seed;
  float* ws = (float*)workspace;
  a.reg = ws;          // [4] (16-byte slot)
  a.r = ws + 4;
  a.dz3 = a.r + N;
  a.js = a.dz3 + N * n3;
  a.a2 = a.js + N * L.ns;
  for (int k = 0; k < n_demo; ++k) {
    if (demo_rows_host[k] < 0) return set_error(MFG_EINVAL, "reward_net_train_step: negative store row");
    a.rows[k] = demo_rows_host[k];
  }
  for (int k = 0; k < n_gen; ++k) {
    if (gen_rows_host[k] < 0) return set_error(MFG_EINVAL, "reward_net_train_step: negative store row");
    a.rows[RT_MAX_TRAJ + k] = gen_rows_host[k];
  }
  hipStream_t st = (hipStream_t)stream;
  const int h1 = k1 / 2, h2 = k2 / 2, W1 = d + 2 * h1, W2 = d + 2 * h2;
  const size_t lds = (size_t)(W1 * W1 + W2 * W2 + L.a2 + f2 * W2 * W2 + L.ns) * sizeof(float);
  if (k1 == 5 && k2 == 3 && f2 == 2)
    {
    if (L.a2 <= 4 * RT_BLOCK) hipLaunchKernelGGL((k_rn_train_sample<5, 3, 2, 4>), dim3((unsigned)N + 1), dim3(RT_BLOCK), lds, st, a);
    else hipLaunchKernelGGL((k_rn_train_sample<5, 3, 2, 8>), dim3((unsigned)N + 1), dim3(RT_BLOCK), lds, st, a);
  }
  else
    hipLaunchKernelGGL((k_rn_train_sample<0, 0, 0, 8>), dim3((unsigned)N + 1), dim3(RT_BLOCK), lds, st, a);
  RtCombineArgs c{};
  c.params = params; c.m = adam_m; c.v = adam_v; c.grad = grad; c.stats = stats;
  c.r = a.r; c.a2 = a.a2; c.dz3 = a.dz3; c.js = a.js; c.reg = a.reg;
  c.d = d; c.k1 = k1; c.f2 = f2; c.k2 = k2; c.n3 = n3; c.n4 = n4;
  c.steps = steps; c.n_demo = n_demo; c.n_gen = n_gen;
  c.demo_scale = 1.0f / (float)demo_divisor;
  c.l1l2 = a.l1l2;
  c.apply = apply ? 1 : 0;
  if (apply) c.lr_t = adam_lr_t(lr, beta1, beta2, adam_step);
  c.beta1 = (float)beta1; c.beta2 = (float)beta2; c.eps = (float)eps;
  hipLaunchKernelGGL(k_rn_train_combine, dim3((unsigned)((L.np + RT_CP - 1) / RT_CP)), dim3(RT_BLOCK), (size_t)N * (1 + n3) * sizeof(float), st, c);
  return hipGetLastError() == hipSuccess ? MFG_OK : set_error(MFG_ELAUNCH, "reward_net_train_step: launch failed");
}

int mfg_reward_net_adam(float* params, float* adam_m, float* adam_v, const float* grad, int64_t n, double lr, double beta1,
                        double beta2, double eps, int64_t adam_step, mfg_stream_t stream) {
  if (!params || !adam_m || !adam_v || !grad || n < 0 || adam_step < 1) return set_error(MFG_EINVAL, "reward_net_adam: bad argument");
  if (n == 0) return MFG_OK;
  hipLaunchKernelGGL(k_rn_adam, dim3((unsigned)((n + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, (hipStream_t)stream, params, adam_m,
                     adam_v, grad, n, adam_lr_t(lr, beta1, beta2, adam_step), (float)beta1, (float)beta2, (float)eps);
  return hipGetLastError() == hipSuccess ? MFG_OK : set_error(MFG_ELAUNCH, "reward_net_adam: launch failed");
}

}  // extern "C"
