// The IRL reward network (reference networks.py:46-81) evaluated INSIDE the packed step kernel k_core_small<..., RN>
// (mfg_core.h), for the reference's geometry (5x5 conv, 1 filter -> 3x3 conv, 2 filters -> FC n3 -> [., state] FC n4 -> 1,
// tanh) at d = 21 / 15.
//
// Why inside: an IRL env step (ac_irl.py:674-691: sample -> transition -> r = reward_net(pi, P) -> delta) was three dependent
// launches -- step kernel with the actions written out (7.2 MB at B = 4 096), reward-network kernel reading them back (one
// sample per wave, a third of its 12 us in the action fetch and prologue), row reduction.  The action matrix of a trajectory
// is already in the LDS tile of the wave that sampled it; here that wave runs the network on its G = 3 (d = 21) / 4 (d = 15)
// tiles right behind the column pass, and the reward joins delta in the same step: P goes to HBM only when the caller asks
// for it (MFG_ROLLOUT_WRITE_P), an env step with per-step updates is two launches.
//
// Mapping (that of k_reward_net_mfma's convolution phase): lane = (strip r of RUN columns, row y), the rows of a strip in
// consecutive lanes -- the run's own row comes from the tile (RUN + 4 reads, scaled by 1 / S_y when the tile holds the
// un-normalised variates), the rows above and below from the neighbouring lanes (DPP wave shifts), the side columns of the
// conv1 map from the neighbouring strips (two lane permutes).  FC3 is evaluated by the wave on its own (the stand-alone
// kernel splits K over 16 waves and pays two block barriers per 16 samples -- a block here has 12): a lane's 2 RUN inputs
// meet their weights as RUN 8-byte LDS reads per unit (fc3_w staged once per block, <= 32 KB), four units advance together
// through one block of 24 DPP adds.  FC4 / output as in the stand-alone kernel.  Dropout uniforms of all G samples of a wave
// come from ONE Philox evaluation (lane = (sample, slot), n3 + n4 <= 16), with the counters of mfg_reward_net_forward.
#pragma once
#include "mfg_rn_common.h"

namespace mfg {

struct RnFusedArgs {
  int on;  // != 0: evaluate the network in the step kernel (reward_kind must be MFG_REWARD_EXTERNAL)
  int n3, n4;
  float keep_prob;
  const float *c1w, *c1b, *c2w, *c2b, *w3, *b3, *w4, *b4, *wo, *bo;
  // dropout masks of env step s of the launch, trajectory b:  Philox key = seed ^ ((call0 + 1 + s call_stride) * golden),
  // sample counter = sample_offset + b sample_stride_b + s sample_stride_s   (per-step launches: 1, 1, 0 -- the keys of T
  // separate mfg_reward_net_forward calls; a whole rollout scored by ONE call over [B*T] transitions: 0, T, 1)
  uint64_t seed, call0, sample_offset;
  int call_stride, sample_stride_b, sample_stride_s;
};

template <int D>
struct RnFusedGeom {
  static constexpr int RUN = D == 21 ? 7 : 5, RPR = 3, K1 = 5, K2 = 3, F2 = 2, H1 = 2, H2 = 1, KK = F2 * D * D;
  static_assert(RUN * RPR == D && D * RPR <= WAVE, "runs tile a row exactly and fit one wavefront");
};

// floats of LDS behind the step kernel's own regions: fc3_w | small weights | reward per trajectory of the tile | uniforms
__host__ __device__ inline size_t rn_fused_small_floats(int d, int n3, int n4) { return ((size_t)(n4 * (n3 + d) + 2 * n4 + 1 + n3) + 3) & ~(size_t)3; }
inline size_t rn_fused_lds_floats(int d, int n3, int n4, int tb, int waves) {
  const size_t w3 = ((size_t)n3 * 2 * d * d + 3) & ~(size_t)3;
  return w3 + rn_fused_small_floats(d, n3, n4) + (size_t)((tb + 3) & ~3) + (size_t)waves * WAVE;
}
// shapes the fused evaluation supports (else the caller keeps the separate launches)
inline bool rn_fused_supported(int d, int k1, int f2, int k2, int n3, int n4, const void* fc3_w) {
  return (d == 21 || d == 15) && k1 == 5 && k2 == 3 && f2 == 2 && n3 >= 1 && n4 >= 1 && n3 + n4 <= 16 &&
         (size_t)n3 * 2 * d * d * 4 <= 32 * 1024 && (((uintptr_t)fc3_w) & 7) == 0;
}

// block-wide staging of the weights (call before the kernel's first block barrier)
template <int D, int NT>
__device__ __forceinline__ void rn_fused_stage(const RnFusedArgs& rn, float* s_w3, float* s_sm, int tid) {
  constexpr int KK = RnFusedGeom<D>::KK;
  const int n3 = rn.n3, n4 = rn.n4, nin = n3 + D;
  // fc3_w: 8-byte pieces (the tensor starts at a multiple of 8 bytes only), ALL of a thread's loads in flight together (a
  // per-step launch pays this latency on its critical path: four rounds of four loads cost 6 800 ticks, see phase_timing_irl)
  const int n2 = (n3 * KK) >> 1;
  const float2* src = reinterpret_cast<const float2*>(rn.w3);
  float2* dst = reinterpret_cast<float2*>(s_w3);
  constexpr int NV = 16;  // pieces per thread and round: 32 KB / 8 B / 256 threads
  for (int k0 = 0; k0 < n2; k0 += NV * NT) {
    float2 v[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int k = k0 + u * NT + tid;
      v[u] = src[k < n2 ? k : 0];
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int k = k0 + u * NT + tid;
      if (k < n2) dst[k] = v[u];
    }
  }
  float* s_w4 = s_sm;
  float* s_b4 = s_w4 + n4 * nin;
  float* s_wo = s_b4 + n4;
  float* s_bo = s_wo + n4;
  float* s_b3 = s_bo + 1;
  for (int k = tid; k < n4 * nin; k += NT) s_w4[k] = rn.w4[k];
  if (tid < n4) {
    s_b4[tid] = rn.b4[tid];
    s_wo[tid] = rn.wo[tid];
  }
  if (tid == 0) s_bo[0] = rn.bo[0];
  if (tid < n3) s_b3[tid] = rn.b3[tid];
}

// Dropout uniforms of the wave's samples: lane = (sample smp = lane / 16, slot = lane % 16); slot < n3: FC3 unit slot, then
// the FC4 units.  `sample0` = sample counter of the wave's first trajectory, `stride` between its trajectories.
__device__ __forceinline__ void rn_fused_uniforms(const RnFusedArgs& rn, uint64_t key, uint64_t sample0, uint64_t stride, float* s_uw,
                                                  int lane) {
  const int slot = lane & 15, smp = lane >> 4;
  const bool fc3 = slot < rn.n3;
  const uint32_t elem = (uint32_t)(fc3 ? slot : slot - rn.n3);
  s_uw[lane] = u01(philox_elem(key, elem, fc3 ? 3u : 4u, sample0 + (uint64_t)smp * stride, 0).x);
}

// r(state, action) of ALL trajectories of a wave in one pass.  Lane = (trajectory t, row y) -- the step kernel's own mapping:
// the lane convolves the row it sampled.  Its row (D values, normalised by 1 / S_y on the way in) sits in registers with a
// zero halo of two columns; the rows above and below come from the neighbouring lanes (DPP wave shifts; the lanes at the
// ends of a trajectory skip the taps that would reach outside the image -- their neighbours belong to another trajectory),
// the conv1 map of the whole row stays in registers, so conv2 needs no exchange along the row at all.  Every lane carries
// D = 21 (15) independent accumulators: the wave is alone on its SIMD at the batch sizes of the IRL configuration, and what it
// needs is independent work per instruction, not occupancy.  (First version: one trajectory after the other with lane =
// (strip of 7 columns, row) -- the stand-alone kernel's mapping --, 3 passes of ~1 300 dependent-ish instructions: 38 600 ticks
// per step, four times the sampling loop.)
//   FC3: the row's 2 D inputs meet their weights as D 8-byte LDS reads per unit (lanes of different trajectories read the
//   same address: broadcast), four units at a time; the per-row partials of a unit go through the wave's own tile region
//   (scratch: its rows are in registers by then) and are added per (trajectory, unit) in row order -- a fixed association.
//   FC4 / output: lanes (t, unit) / (t, 0), a few dozen dependent fused multiply-adds.
//   trow  : this lane's tile row [D] (un-normalised variates or the normalised action), `inv` the factor that normalises it
//   valid : the lane holds a row of a live trajectory (else it contributes zeros and its results are dropped)
//   st    : state [D] of this lane's trajectory (fp32, LDS)
//   scr   : the wave's scratch (>= G (n3 D + 32) floats; may alias the tile rows: every lane's row is in registers before
//           the first write)
//   s_uw  : the wave's dropout uniforms [G][16] (rn_fused_uniforms) -- read only when keep_prob < 1
//   r_out : [G] rewards of the wave's trajectories (written by lane (t, 0) of every live trajectory)
template <int D>
__device__ __attribute__((noinline)) void rn_fused_eval_wave(const float* trow, float inv, bool valid, int t, int i, const float* st,
                                                   const float* s_w3, const float* s_sm, const float* s_uw, float* scr,
                                                   const RnFusedArgs& rn, float* r_out) {
  using Gm = RnFusedGeom<D>;
  constexpr int K1 = Gm::K1, K2 = Gm::K2, H1 = Gm::H1, H2 = Gm::H2, KK = Gm::KK, G = WAVE / D;
  const int n3 = rn.n3, n4 = rn.n4, nin = n3 + D;
  const float* s_w4 = s_sm;
  const float* s_b4 = s_w4 + n4 * nin;
  const float* s_wo = s_b4 + n4;
  const float* s_bo = s_wo + n4;
  const float* s_b3 = s_bo + 1;
  int yv = i;
  asm volatile("" : "+v"(yv));  // (row predicates are formed where they are used, not hoisted into scalar-register pairs)
  RnConstF c1w_s = (RnConstF)rn.c1w, c1b_s = (RnConstF)rn.c1b, c2w_s = (RnConstF)rn.c2w, c2b_s = (RnConstF)rn.c2b;
  // ---- the lane's row, zero halo
  constexpr int W1 = D + 2 * H1;
  float xw[W1];
#pragma unroll
  for (int k = 0; k < D; ++k) xw[k + H1] = trow[k];
#pragma unroll
  for (int k = 0; k < H1; ++k) xw[k] = xw[W1 - 1 - k] = 0.0f;
#pragma unroll
  for (int k = 0; k < D; ++k) xw[k + H1] = valid ? xw[k + H1] * inv : 0.0f;
  // ---- conv1 5x5 + ReLU
  float c1[D];
  {
    float w1[K1 * K1];
#pragma unroll
    for (int k = 0; k < K1 * K1; ++k) w1[k] = c1w_s[k];
    const float b1 = c1b_s[0];
#pragma unroll
    for (int x = 0; x < D; ++x) c1[x] = b1;
#pragma unroll
    for (int x = 0; x < D; ++x)
#pragma unroll
      for (int dx = 0; dx < K1; ++dx) c1[x] = fmaf(xw[x + dx], w1[H1 * K1 + dx], c1[x]);
    float xs[W1];
#pragma unroll
    for (int k = 0; k < W1; ++k) xs[k] = xw[k];
#pragma unroll
    for (int e = 1; e <= H1; ++e) {  // rows y - e (the halo columns are zero in every lane: not shifted)
#pragma unroll
      for (int k = H1; k < D + H1; ++k) xs[k] = lane_below(xs[k]);
      if (yv >= e) {
#pragma unroll
        for (int x = 0; x < D; ++x)
#pragma unroll
          for (int dx = 0; dx < K1; ++dx) c1[x] = fmaf(xs[x + dx], w1[(H1 - e) * K1 + dx], c1[x]);
      }
    }
#pragma unroll
    for (int k = 0; k < W1; ++k) xs[k] = xw[k];
#pragma unroll
    for (int e = 1; e <= H1; ++e) {  // rows y + e
#pragma unroll
      for (int k = H1; k < D + H1; ++k) xs[k] = lane_above(xs[k]);
      if (yv + e < D) {
#pragma unroll
        for (int x = 0; x < D; ++x)
#pragma unroll
          for (int dx = 0; dx < K1; ++dx) c1[x] = fmaf(xs[x + dx], w1[(H1 + e) * K1 + dx], c1[x]);
      }
    }
  }
  // ---- conv2 3x3, two filters (packed FMA: same input, weight pair) + ReLU
  rn_v2f_t a2[D];
  {
    static_assert(H2 == 1, "one halo column");
    constexpr int W2 = D + 2;
    rn_v2f_t w2[K2 * K2];
#pragma unroll
    for (int k = 0; k < K2 * K2; ++k) w2[k] = rn_v2f_t{c2w_s[k], c2w_s[K2 * K2 + k]};
    const rn_v2f_t b2 = {c2b_s[0], c2b_s[1]};
    float m0[W2], ms[W2];
    m0[0] = m0[W2 - 1] = 0.0f;
#pragma unroll
    for (int x = 0; x < D; ++x) m0[x + 1] = valid ? relu_f32(c1[x]) : 0.0f;
#pragma unroll
    for (int x = 0; x < D; ++x) a2[x] = b2;
#pragma unroll
    for (int x = 0; x < D; ++x)
#pragma unroll
      for (int dx = 0; dx < K2; ++dx) a2[x] = __builtin_elementwise_fma(rn_v2f_t{m0[x + dx], m0[x + dx]}, w2[H2 * K2 + dx], a2[x]);
    ms[0] = ms[W2 - 1] = 0.0f;
#pragma unroll
    for (int k = 1; k <= D; ++k) ms[k] = lane_below(m0[k]);
    if (yv >= 1) {
#pragma unroll
      for (int x = 0; x < D; ++x)
#pragma unroll
        for (int dx = 0; dx < K2; ++dx) a2[x] = __builtin_elementwise_fma(rn_v2f_t{ms[x + dx], ms[x + dx]}, w2[(H2 - 1) * K2 + dx], a2[x]);
    }
#pragma unroll
    for (int k = 1; k <= D; ++k) ms[k] = lane_above(m0[k]);
    if (yv + 1 < D) {
#pragma unroll
      for (int x = 0; x < D; ++x)
#pragma unroll
        for (int dx = 0; dx < K2; ++dx) a2[x] = __builtin_elementwise_fma(rn_v2f_t{ms[x + dx], ms[x + dx]}, w2[(H2 + 1) * K2 + dx], a2[x]);
    }
#pragma unroll
    for (int x = 0; x < D; ++x) {
      a2[x][0] = relu_f32(a2[x][0]);
      a2[x][1] = relu_f32(a2[x][1]);
    }
  }
  // every lane's row is in registers: the tile rows may be overwritten from here on
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // ---- FC3 partials of this row: unit u -> scr[(t n3 + u) D + y]   (NHWC inputs: (y D + x) 2 + channel)
  const int tc = t < G ? t : G - 1;
  float* s_fc = scr;
  float* s_h = scr + G * n3 * D;  // [G][16] FC3 activations, then [G][16] output-unit products
  float* s_o = s_h + G * 16;
  {
    const float2* wbase = reinterpret_cast<const float2*>(s_w3) + i * D;
    for (int u0 = 0; u0 < n3; u0 += 4) {
      float p4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int x = 0; x < D; ++x) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float2 wv2 = wbase[(u0 + u < n3 ? u0 + u : n3 - 1) * (KK / 2) + x];
          p4[u] = fmaf(a2[x][0], wv2.x, p4[u]);
          p4[u] = fmaf(a2[x][1], wv2.y, p4[u]);
        }
        // (a fence every three columns: left alone the compiler issues all 4 D weight reads of the group up front -- 168
        //  registers -- and spills the step kernel's own state around them)
        if (x % 3 == 2) asm volatile("" ::: "memory");
      }
      if (valid) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u0 + u < n3) s_fc[(tc * n3 + u0 + u) * D + i] = p4[u];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  const bool drop = rn.keep_prob < 1.0f;
  const float inv_keep = 1.0f / rn.keep_prob;
  // ---- lane (t, u < n3): unit u of trajectory t -- the D row partials in row order, bias, ReLU (+ dropout)
  if (valid && i < n3) {
    const float* src = s_fc + (tc * n3 + i) * D;
    float h = 0.0f;
#pragma unroll
    for (int y2 = 0; y2 < D; ++y2) h += src[y2];
    h = fmaxf(h + s_b3[i], 0.0f);
    if (drop) h = (s_uw[tc * 16 + i] <= rn.keep_prob) ? h * inv_keep : 0.0f;
    s_h[tc * 16 + i] = h;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // ---- lane (t, m < n4): FC4 unit m over [h3, state] + ReLU (+ dropout), times its output weight
  if (valid && i < n4) {
    const float* w = s_w4 + i * nin;
    float z4 = s_b4[i];
    for (int k = 0; k < n3; ++k) z4 = fmaf(s_h[tc * 16 + k], w[k], z4);
#pragma unroll
    for (int k = 0; k < D; ++k) z4 = fmaf(st[k], w[n3 + k], z4);
    float h4 = fmaxf(z4, 0.0f);
    if (drop) h4 = (s_uw[tc * 16 + ((n3 + i) & 15)] <= rn.keep_prob) ? h4 * inv_keep : 0.0f;
    s_o[tc * 16 + i] = h4 * s_wo[i];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  if (valid && i == 0) {
    float z = s_bo[0];
    for (int m = 0; m < n4; ++m) z += s_o[tc * 16 + m];
    r_out[tc] = tanhf(z);
  }
}

}  // namespace mfg
