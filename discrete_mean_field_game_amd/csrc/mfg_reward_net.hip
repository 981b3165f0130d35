// Batched forward pass of the IRL reward network r(pi, P).
//
// Reference: networks.py:46-81 (r_net_dropout_l1l2 and its three siblings), evaluated per env step at
// ac_irl.py:683 with batch 1.  Architecture (f1 = 1, k1 = 5, f2 = 2, k2 = 3 are fixed by ac_irl.py:251-267):
//   action [d,d] -> conv 5x5 (1 filter, SAME, ReLU) -> conv 3x3 (f2 filters, SAME, ReLU) -> NHWC flatten
//   -> FC n3 ReLU (-> dropout) -> concat state [d] -> FC n4 ReLU (-> dropout) -> FC 1, tanh.
// One wavefront evaluates one (state, action) sample entirely on chip: the action tile and the conv1
// feature map live in LDS (zero halos), conv2 activations in registers, FC3 is a wave reduction against
// weights staged in LDS.  fp32 like the reference's TF graph.  The MIOpen path through PyTorch needs
// ~20 launches and 1.6 ms for 65 536 samples; this kernel is one launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/mfg_hip.h"
#include "mfg_core.h"
#include "mfg_rn_common.h"

namespace mfg {

struct RewardNetArgs {
  const float* state;   // [B,d]
  const float* action;  // [B,d,d]
  int64_t B;
  int d, k1, f2, k2, n3, n4;
  const float *c1w, *c1b;  // [k1*k1], [1]
  const float *c2w, *c2b;  // [f2][k2*k2], [f2]
  const float *w3, *b3;    // [n3][f2*d*d] (input index (pixel*f2 + channel): TF NHWC flatten), [n3]
  const float *w4, *b4;    // [n4][n3+d], [n4]
  const float *wo, *bo;    // [n4], [1]
  float keep_prob;         // 1 -> no dropout
  uint64_t seed, sample_offset;
  float* reward;  // [B]
  int w3_in_lds;
  // SUMS variant (IRL step with per-step updates, ac_irl.py:683-708): delta_b = delta0_b + r_b is written to delta_out and
  // the block leaves one partial row [sum delta phi(state) | sum delta g | sum r | count] of ITS samples in part_rows
  const double* delta0;
  const double* gsc;
  double* delta_out;
  double* part_rows;  // [gridDim.x][F+3]
  // matrix-core SUMS kernel: the TD error's value part is formed HERE, delta0_b = sum_k w_k (gamma phi_k(next_b) - phi_k(state_b))
  // (ac_irl.py:686-691 with the critic of mfg_ac2.py:333-347), from the features the batch sums need anyway -- the step kernel
  // that precedes this launch then needs theta only, and the row reduction of the previous env step runs inside it
  const double* td_w;       // [F] critic weights
  const float* state_next;  // [B,d]
  double td_gamma;          // discount^t of this env step
  double* col_f;            // [gridDim.x] column F (the actor's sum) of the partial rows once more, contiguous: what every
                            // sampling wave of the next step kernel reads
  // states inside a rollout's pi_traj [B', T+1, d]: sample n = (b', t) reads row b' (T+1) + t (state_T = T; 0: plain [B,d])
  int state_T;
};
__device__ __forceinline__ int64_t rn_state_row(const RewardNetArgs& a, int64_t b) {
  if (a.state_T <= 0) return b;
  const int64_t q = b / a.state_T;
  return q * (a.state_T + 1) + (b - q * a.state_T);
}

#ifndef MFG_RN_WAVES
#define MFG_RN_WAVES 8
#endif
#ifndef MFG_RN_P21
#define MFG_RN_P21 25  // LDS row pitch of the padded tiles in the run-mapped kernel, d = 21 (>= 25)
#endif
#ifndef MFG_RN_P15
#define MFG_RN_P15 19  // same, d = 15 (>= 19)
#endif
#ifndef MFG_RN_BPC
#define MFG_RN_BPC 2
#endif
#ifndef MFG_RN_LDS_MIN
#define MFG_RN_LDS_MIN 16  // samples per block from which the FC3 weights are staged in LDS
#endif
// 8 waves share one LDS copy of the FC3 weights (28 KB at d = 21, n3 = 8): 76 KB per block, 2 blocks per CU.
constexpr int RN_WAVES = MFG_RN_WAVES, RN_BLOCK = RN_WAVES * WAVE, RN_MAXF2 = 2, RN_MAXN = 32;

// PPMAX = max pixels per lane (ceil(d*d/64)).  K1 / K2 / F2 > 0: compile-time conv geometry (the reference always
// uses k1 = 5, k2 = 3, f2 = 2, ac_irl.py:251-267): taps unroll, LDS reads get immediate offsets and can be issued
// together; with run-time bounds every tap is a dependent ~100-cycle LDS round trip (the first version of this
// kernel spent 30 us per sample that way).  0 = generic run-time value.
// D > 0: compile-time d (21 and 15, the reference's two sizes): pixel -> (row, column) needs no run-time division and
// every tap of the convolutions is an immediate LDS offset (run-time d spent ~600 integer instructions per sample on
// index arithmetic, more than the network's own FMAs).
template <int PPMAX, int K1, int K2, int F2, int D>
__global__ __launch_bounds__(RN_BLOCK) void k_reward_net(RewardNetArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = D ? D : a.d, dd = d * d, n3 = a.n3, n4 = a.n4;
  const int k1 = K1 ? K1 : a.k1, k2 = K2 ? K2 : a.k2, f2 = F2 ? F2 : a.f2;
  const int h1 = k1 / 2, h2 = k2 / 2;
  const int W1 = d + 2 * h1, W2 = d + 2 * h2;  // padded widths of the input / conv1 maps
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  // LDS carve: small weights | fc3 weights (optional) | per-wave tiles
  float* sc1 = smem;                   // k1*k1 + 1
  float* sc2 = sc1 + k1 * k1 + 1;      // f2*k2*k2 + f2
  float* s4 = sc2 + f2 * k2 * k2 + f2; // n4*(n3+d) + n4 + n4 + 1 + n3
  const int n_s4 = n4 * (n3 + d) + 2 * n4 + 1 + n3;
  int off = (k1 * k1 + 1) + (f2 * k2 * k2 + f2) + n_s4;
  off = (off + 3) & ~3;
  float* s3 = smem + off;  // n3 * f2 * dd (when w3_in_lds)
  if (a.w3_in_lds) off += n3 * f2 * dd;
  off = (off + 3) & ~3;
  float* tin = smem + off + wv * (W1 * W1 + W2 * W2);  // padded input tile of this wave
  float* tc1 = tin + W1 * W1;                          // padded conv1 map of this wave
  for (int k = tid; k < k1 * k1; k += RN_BLOCK) sc1[k] = a.c1w[k];
  if (tid == 0) sc1[k1 * k1] = a.c1b[0];
  for (int k = tid; k < f2 * k2 * k2; k += RN_BLOCK) sc2[k] = a.c2w[k];
  for (int k = tid; k < f2; k += RN_BLOCK) sc2[f2 * k2 * k2 + k] = a.c2b[k];
  float* s_w4 = s4;
  float* s_b4 = s_w4 + n4 * (n3 + d);
  float* s_wo = s_b4 + n4;
  float* s_bo = s_wo + n4;
  float* s_b3 = s_bo + 1;
  for (int k = tid; k < n4 * (n3 + d); k += RN_BLOCK) s_w4[k] = a.w4[k];
  for (int k = tid; k < n4; k += RN_BLOCK) {
    s_b4[k] = a.b4[k];
    s_wo[k] = a.wo[k];
  }
  if (tid == 0) s_bo[0] = a.bo[0];
  for (int k = tid; k < n3; k += RN_BLOCK) s_b3[k] = a.b3[k];
  if (a.w3_in_lds) {
    // batched copy: 4 independent 16-byte loads in flight per thread (a plain loop serialises ~30 round trips)
    const int n4 = (n3 * f2 * dd) >> 2;
    const float4* src4 = reinterpret_cast<const float4*>(a.w3);
    float4* dst4 = reinterpret_cast<float4*>(s3);
    for (int k0 = 0; k0 < n4; k0 += 4 * RN_BLOCK) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * RN_BLOCK + tid;
        v[u] = (k < n4) ? src4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * RN_BLOCK + tid;
        if (k < n4) dst4[k] = v[u];
      }
    }
    for (int k = (n4 << 2) + tid; k < n3 * f2 * dd; k += RN_BLOCK) s3[k] = a.w3[k];
  }
  for (int k = lane; k < W1 * W1 + W2 * W2; k += WAVE) tin[k] = 0.0f;  // zero halos (interiors are rewritten)
  __syncthreads();
  const float* w3 = a.w3_in_lds ? s3 : a.w3;
  const float inv_keep = 1.0f / a.keep_prob;
  const bool drop = a.keep_prob < 1.0f;
  const int64_t nw = (int64_t)gridDim.x * RN_WAVES;
  // pixel slots of this lane (p = lane + 64 q): offsets into the two padded tiles, computed once
  int o1[PPMAX], o2[PPMAX];
#pragma unroll
  for (int q = 0; q < PPMAX; ++q) {
    const int p = lane + q * WAVE;
    const int pc = p < dd ? p : 0;
    const int y = pc / d, x = pc - y * d;
    o1[q] = y * W1 + x;
    o2[q] = y * W2 + x;
  }
  for (int64_t b = (int64_t)blockIdx.x * RN_WAVES + wv; b < a.B; b += nw) {
    const float* act = a.action + b * dd;
    // 1. action -> padded LDS tile
    float av[PPMAX];
#pragma unroll
    for (int q = 0; q < PPMAX; ++q) av[q] = (lane + q * WAVE < dd) ? act[lane + q * WAVE] : 0.0f;
#pragma unroll
    for (int q = 0; q < PPMAX; ++q)
      if (lane + q * WAVE < dd) tin[o1[q] + h1 * W1 + h1] = av[q];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // 2. conv1 (cross-correlation, SAME) + ReLU -> padded conv1 map
#pragma unroll
    for (int q = 0; q < PPMAX; ++q) {
      if (lane + q * WAVE < dd) {
        float s = sc1[k1 * k1];
        const float* tp = tin + o1[q];
#pragma unroll
        for (int dy = 0; dy < (K1 ? K1 : k1); ++dy)
#pragma unroll
          for (int dx = 0; dx < (K1 ? K1 : k1); ++dx) s = fmaf(tp[dy * W1 + dx], sc1[dy * k1 + dx], s);
        tc1[o2[q] + h2 * W2 + h2] = fmaxf(s, 0.0f);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // 3. conv2 + ReLU -> registers act2[pixel slot][channel]
    float act2[PPMAX][RN_MAXF2];
#pragma unroll
    for (int q = 0; q < PPMAX; ++q) {
#pragma unroll
      for (int c = 0; c < RN_MAXF2; ++c) act2[q][c] = 0.0f;
      if (lane + q * WAVE < dd) {
#pragma unroll
        for (int c = 0; c < RN_MAXF2; ++c) {
          if (c < f2) {
            float s = sc2[f2 * k2 * k2 + c];
            const float* tp = tc1 + o2[q];
#pragma unroll
            for (int dy = 0; dy < (K2 ? K2 : k2); ++dy)
#pragma unroll
              for (int dx = 0; dx < (K2 ? K2 : k2); ++dx) s = fmaf(tp[dy * W2 + dx], sc2[c * k2 * k2 + dy * k2 + dx], s);
            act2[q][c] = fmaxf(s, 0.0f);
          }
        }
      }
    }
    // dropout uniforms of this sample, all at once: lane o < 32 draws unit o of FC3, lane 32+o unit o of FC4 (same
    // Philox counters as a per-unit draw; computing them one by one, wave-uniformly, was a third of the kernel)
    float u_drop = 0.0f;
    if (drop) {
      const u32x4 r = philox_elem(a.seed, (uint32_t)(lane & 31), lane < 32 ? 3u : 4u, a.sample_offset + (uint64_t)b, 0);
      u_drop = u01(r.x);
    }
    // 4. FC3 + ReLU (+ dropout): every lane ends up with all n3 activations it needs for FC4
    float h3_mine = 0.0f;  // lane o < n3 keeps h3[o]
#pragma unroll 2
    for (int o = 0; o < n3; ++o) {
      const float* wrow = w3 + (int64_t)o * f2 * dd;
      float s = 0.0f;
#pragma unroll
      for (int q = 0; q < PPMAX; ++q) {
        const int p = lane + q * WAVE;
        if (p < dd) {
#pragma unroll
          for (int c = 0; c < RN_MAXF2; ++c)
            if (c < f2) s = fmaf(act2[q][c], wrow[p * f2 + c], s);
        }
      }
      s = wave_sum(s);
      float h = fmaxf(s + s_b3[o], 0.0f);
      if (drop) h = (__shfl(u_drop, o, WAVE) <= a.keep_prob) ? h * inv_keep : 0.0f;
      if (lane == o) h3_mine = h;
    }
    // 5. FC4 over [h3, state] + ReLU (+ dropout): lane o < n4
    float h4 = 0.0f;
    {
      const int o = lane < n4 ? lane : 0;
      float s = s_b4[o];
      for (int k = 0; k < n3; ++k) s = fmaf(__shfl(h3_mine, k, WAVE), s_w4[o * (n3 + d) + k], s);
      const float* st = a.state + rn_state_row(a, b) * d;
      for (int k = 0; k < d; ++k) s = fmaf(st[k], s_w4[o * (n3 + d) + n3 + k], s);
      h4 = fmaxf(s, 0.0f);
      if (drop) h4 = (__shfl(u_drop, 32 + o, WAVE) <= a.keep_prob) ? h4 * inv_keep : 0.0f;
      h4 = (lane < n4) ? h4 * s_wo[o] : 0.0f;
    }
    // 6. output unit, tanh
    const float z = wave_sum(h4) + s_bo[0];
    if (lane == 0) a.reward[b] = tanhf(z);
    __builtin_amdgcn_wave_barrier();
  }
}


// ---------------------------------------------------------------------------------------------
// Run-mapped kernel for the reference geometry (k1 = 5, k2 = 3, f2 = 2) at compile-time d (21 and 15).
// The pixel-per-lane kernel above is LDS bound (SQ_LDS_IDX_ACTIVE 78 % of the CU cycles, 28 % of them bank conflicts,
// VALU 37 %): every tap of every pixel is its own LDS read.  Here a lane owns a horizontal RUN of pixels of one row
// (RPR runs per row, RUN*RPR = d: 63 lanes at d = 21, 45 at d = 15); for each kernel row it reads the RUN+k-1 inputs
// under its run once and slides the taps over them in registers: 55 + 27 LDS reads per sample instead of 175 + 63.
// The run's 2*RUN FC3 inputs are contiguous in the NHWC-flattened weight rows (8-byte reads), the FC3 reductions run on
// the DPP path instead of ds_bpermute, and the conv weights sit in scalar registers.
// ---------------------------------------------------------------------------------------------
// (dpp_mov_f32 lives in mfg_device.h)
// (wave_sum_f32_dpp / wave_sum4_to_lane63: mfg_rn_common.h)
template <int D, int RUN, int RPR, int P1, int P2>
struct RunsGeom {
  static_assert(RUN * RPR == D && D * RPR <= WAVE, "runs must tile a row exactly and fit one wavefront");
  static constexpr int K1 = 5, K2 = 3, F2 = 2, H1 = 2, H2 = 1, DD = D * D;
  static constexpr int T1 = (D + 2 * H1) * P1, T2 = (D + 2 * H2) * P2;  // floats per padded tile (pitches P1, P2)
  static constexpr int PP = (DD + WAVE - 1) / WAVE;
  static size_t lds_floats(int n3, int n4, bool w3_in_lds) {
    size_t fl = (size_t)(n4 * (n3 + D) + 2 * n4 + 1 + n3);
    fl = (fl + 3) & ~(size_t)3;
    if (w3_in_lds) fl += (size_t)n3 * F2 * DD;
    fl = (fl + 3) & ~(size_t)3;
    return fl + (size_t)RN_WAVES * (T1 + T2);
  }
};

// SUMS: the IRL env step needs, right after the rewards, the batch sums of the TD update (a6 / a8) over the same samples.
// Here every wave folds its samples into FO = F + 3 running sums spread over its lanes (entry k = lane + 64 q: one
// fp64 fma per entry and sample, operands from a (D + 1)-float LDS line [state | 1]); the block's eight waves are added
// up in wave order at the end and leave ONE partial row.  The separate gradient kernel of the step (a launch, a re-read
// of the states, a fence-and-last-block finish: 10.6 us at B = 4 096) shrinks to the row reduction.
template <int D, int RUN, int RPR, int P1, int P2, bool SUMS = false>
__global__ __launch_bounds__(RN_BLOCK) void k_reward_net_runs(RewardNetArgs a) {
  using Gm = RunsGeom<D, RUN, RPR, P1, P2>;
  constexpr int K1 = Gm::K1, K2 = Gm::K2, F2 = Gm::F2, H1 = Gm::H1, H2 = Gm::H2, DD = Gm::DD, PP = Gm::PP;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n3 = a.n3, n4 = a.n4, nin = n3 + D;  // FC4 input = [h3 (n3), state (D)], nin <= 64
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  float* s_w4 = smem;              // [n4][nin]
  float* s_b4 = s_w4 + n4 * nin;
  float* s_wo = s_b4 + n4;
  float* s_bo = s_wo + n4;
  float* s_b3 = s_bo + 1;
  int off = n4 * nin + 2 * n4 + 1 + n3;
  off = (off + 3) & ~3;
  float* s3 = smem + off;
  if (a.w3_in_lds) off += n3 * F2 * DD;
  off = (off + 3) & ~3;
  float* tin = smem + off + wv * (Gm::T1 + Gm::T2);
  float* tc1 = tin + Gm::T1;
  for (int k = tid; k < n4 * nin; k += RN_BLOCK) s_w4[k] = a.w4[k];
  for (int k = tid; k < n4; k += RN_BLOCK) {
    s_b4[k] = a.b4[k];
    s_wo[k] = a.wo[k];
  }
  if (tid == 0) s_bo[0] = a.bo[0];
  for (int k = tid; k < n3; k += RN_BLOCK) s_b3[k] = a.b3[k];
  if (a.w3_in_lds) {
    const int nq = (n3 * F2 * DD) >> 2;
    const float4* src4 = reinterpret_cast<const float4*>(a.w3);
    float4* dst4 = reinterpret_cast<float4*>(s3);
    for (int k0 = 0; k0 < nq; k0 += 4 * RN_BLOCK) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * RN_BLOCK + tid;
        v[u] = (k < nq) ? src4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * RN_BLOCK + tid;
        if (k < nq) dst4[k] = v[u];
      }
    }
    for (int k = (nq << 2) + tid; k < n3 * F2 * DD; k += RN_BLOCK) s3[k] = a.w3[k];
  }
  for (int k = lane; k < Gm::T1 + Gm::T2; k += WAVE) tin[k] = 0.0f;  // zero halos (interiors are rewritten)
  // conv weights and biases: ONE gather per wave (lane t holds entry t of [c1w | c1b | c2w | c2b], 46 values), then
  // v_readlane into scalar registers.  Plain `a.c1w[k]` reads are re-issued as vector loads for every sample (the
  // compiler cannot prove that the reward store does not alias them) and sat on the critical path.
  constexpr int NW1 = K1 * K1, NW2 = F2 * K2 * K2;
  static_assert(NW1 + 1 + NW2 + F2 <= WAVE, "conv parameters must fit one wavefront");
  float wtab;
  {
    const float* src = lane < NW1 ? a.c1w + lane
                     : lane == NW1 ? a.c1b
                     : lane < NW1 + 1 + NW2 ? a.c2w + (lane - NW1 - 1)
                     : a.c2b + (lane < NW1 + 1 + NW2 + F2 ? lane - NW1 - 1 - NW2 : 0);
    wtab = *src;
  }
  float w1[NW1], w2[F2][K2 * K2];
#pragma unroll
  for (int k = 0; k < NW1; ++k) w1[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), k));
  const float b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), NW1));
#pragma unroll
  for (int c = 0; c < F2; ++c)
#pragma unroll
    for (int k = 0; k < K2 * K2; ++k)
      w2[c][k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), NW1 + 1 + c * K2 * K2 + k));
  const float b20 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), NW1 + 1 + NW2));
  const float b21 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), NW1 + 2 + NW2));
  __syncthreads();
  const float* w3g = a.w3;
  const bool w3_lds = a.w3_in_lds != 0;
  const float inv_keep = 1.0f / a.keep_prob;
  const bool drop = a.keep_prob < 1.0f;
  // this lane's run: row y, columns x0 .. x0+RUN-1
  const bool active = lane < D * RPR;
  const int y = active ? lane / RPR : 0, x0 = active ? (lane - y * RPR) * RUN : 0;
  const float* win1 = tin + y * P1 + x0;                // top-left of the conv1 window in the padded input tile
  float* out1 = tc1 + (y + H2) * P2 + x0 + H2;          // this run inside the padded conv1 map
  const float* win2 = tc1 + y * P2 + x0;                // top-left of the conv2 window
  const int w3off = (y * D + x0) * F2;                  // the run's 2*RUN inputs inside an FC3 weight row
  int o1[PP];
#pragma unroll
  for (int q = 0; q < PP; ++q) {
    const int p = lane + q * WAVE;
    const int pc = p < DD ? p : 0;
    o1[q] = (pc / D + H1) * P1 + pc % D + H1;
  }
  const int64_t nw = (int64_t)gridDim.x * RN_WAVES;
  int64_t b = (int64_t)blockIdx.x * RN_WAVES + wv;
  // SUMS: this lane's entries k = lane + 64 q of the row: coefficient kind (0 delta, 1 delta g, 2 r, 3 one, -1 none) and
  // the two factors' positions in the line [state (D) | 1]
  constexpr int Qs = D * (D + 1) / 2, Fs = Qs + D + 1, FO = Fs + 3, NPL = (FO + WAVE - 1) / WAVE;
  float* xs = smem + off + RN_WAVES * (Gm::T1 + Gm::T2) + wv * 32;  // (only allocated for SUMS launches)
  int e_ia[NPL], e_ib[NPL], e_cm[NPL];
  double e_acc[NPL];
  if constexpr (SUMS) {
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const int k = lane + q * WAVE;
      e_acc[q] = 0.0;
      e_ia[q] = e_ib[q] = D;
      e_cm[q] = -1;
      if (k < Qs) {
        int i = 0;
        while (i + 1 < D && feat_idx(i + 1, i + 1, D) <= k) ++i;  // row of the upper triangle that holds entry k
        e_ia[q] = i;
        e_ib[q] = i + (k - feat_idx(i, i, D));
        e_cm[q] = 0;
      } else if (k < Qs + D) {
        e_ia[q] = k - Qs;
        e_cm[q] = 0;
      } else if (k == Qs + D) {
        e_cm[q] = 0;
      } else if (k < FO) {
        e_cm[q] = k - Fs + 1;  // F: delta g, F + 1: r, F + 2: count
      }
    }
    if (lane == 0) xs[D] = 1.0f;
  }
  // software prefetch: the next sample's action (and state entry) is in flight while this one is evaluated
  float av[PP], st_mine = 0.0f;
  double d0_next = 0.0, g_next = 0.0;
#pragma unroll
  for (int q = 0; q < PP; ++q) av[q] = (b < a.B && lane + q * WAVE < DD) ? a.action[b * DD + lane + q * WAVE] : 0.0f;
  if (b < a.B && lane >= n3 && lane < nin) st_mine = a.state[rn_state_row(a, b) * D + (lane - n3)];
  if (SUMS && b < a.B) {
    d0_next = a.delta0[b];
    g_next = a.gsc[b];
  }
  for (; b < a.B; b += nw) {
    // 1. action -> padded LDS tile (coalesced global read, pixel p = lane + 64 q)
#pragma unroll
    for (int q = 0; q < PP; ++q)
      if (lane + q * WAVE < DD) tin[o1[q]] = av[q];
    const float st_cur = st_mine;
    const double d0_cur = d0_next, g_cur = g_next;
    if (SUMS && lane >= n3 && lane < nin) xs[lane - n3] = st_cur;
    {
      const int64_t bn = b + nw;
#pragma unroll
      for (int q = 0; q < PP; ++q) av[q] = (bn < a.B && lane + q * WAVE < DD) ? a.action[bn * DD + lane + q * WAVE] : 0.0f;
      if (bn < a.B && lane >= n3 && lane < nin) st_mine = a.state[rn_state_row(a, bn) * D + (lane - n3)];
      if (SUMS && bn < a.B) {
        d0_next = a.delta0[bn];
        g_next = a.gsc[bn];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // 2. conv1 5x5 (cross-correlation, SAME) + ReLU over the run
    float c1[RUN];
#pragma unroll
    for (int k = 0; k < RUN; ++k) c1[k] = b1;
#pragma unroll
    for (int dy = 0; dy < K1; ++dy) {
      float row[RUN + K1 - 1];
#pragma unroll
      for (int t = 0; t < RUN + K1 - 1; ++t) row[t] = win1[dy * P1 + t];
#pragma unroll
      for (int k = 0; k < RUN; ++k)
#pragma unroll
        for (int dx = 0; dx < K1; ++dx) c1[k] = fmaf(row[k + dx], w1[dy * K1 + dx], c1[k]);
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < RUN; ++k) out1[k] = fmaxf(c1[k], 0.0f);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // 3. conv2 3x3, two filters + ReLU
    float a2[RUN][F2];
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
      a2[k][0] = b20;
      a2[k][1] = b21;
    }
#pragma unroll
    for (int dy = 0; dy < K2; ++dy) {
      float row[RUN + K2 - 1];
#pragma unroll
      for (int t = 0; t < RUN + K2 - 1; ++t) row[t] = win2[dy * P2 + t];
#pragma unroll
      for (int k = 0; k < RUN; ++k)
#pragma unroll
        for (int dx = 0; dx < K2; ++dx) {
          a2[k][0] = fmaf(row[k + dx], w2[0][dy * K2 + dx], a2[k][0]);
          a2[k][1] = fmaf(row[k + dx], w2[1][dy * K2 + dx], a2[k][1]);
        }
    }
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
      a2[k][0] = active ? fmaxf(a2[k][0], 0.0f) : 0.0f;
      a2[k][1] = active ? fmaxf(a2[k][1], 0.0f) : 0.0f;
    }
    // dropout uniforms of this sample (lane o < 32: FC3 unit o, lane 32+o: FC4 unit o)
    float u_drop = 0.0f;
    if (drop) {
      const u32x4 r = philox_elem(a.seed, (uint32_t)(lane & 31), lane < 32 ? 3u : 4u, a.sample_offset + (uint64_t)b, 0);
      u_drop = u01(r.x);
    }
    // 4. FC3 + ReLU (+ dropout); lane o < n3 keeps unit o, lanes n3 .. n3+D-1 hold the state: `x4` is FC4's input
    float x4 = (lane >= n3 && lane < nin) ? st_cur : 0.0f;
#pragma unroll 2
    for (int o = 0; o < n3; ++o) {
      float s = 0.0f;
      if (w3_lds) {
        const float2* wr = reinterpret_cast<const float2*>(s3 + o * F2 * DD + w3off);
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
          const float2 wv2 = wr[k];
          s = fmaf(a2[k][0], wv2.x, s);
          s = fmaf(a2[k][1], wv2.y, s);
        }
      } else {
        const float2* wr = reinterpret_cast<const float2*>(w3g + (int64_t)o * F2 * DD + w3off);
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
          const float2 wv2 = wr[k];
          s = fmaf(a2[k][0], wv2.x, s);
          s = fmaf(a2[k][1], wv2.y, s);
        }
      }
      s = wave_sum_f32_dpp(s);
      float h = fmaxf(s + s_b3[o], 0.0f);
      if (drop) h = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(u_drop), o)) <= a.keep_prob) ? h * inv_keep : 0.0f;
      if (lane == o) x4 = h;
    }
    // 5. FC4 over [h3, state] + ReLU (+ dropout), 6. output unit: lane-parallel products, one DPP sum per unit
    float z = s_bo[0];
    for (int o = 0; o < n4; ++o) {
      const float wgt = lane < nin ? s_w4[o * nin + lane] : 0.0f;
      float h4 = fmaxf(wave_sum_f32_dpp(x4 * wgt) + s_b4[o], 0.0f);
      if (drop) h4 = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(u_drop), 32 + o)) <= a.keep_prob) ? h4 * inv_keep : 0.0f;
      z = fmaf(h4, s_wo[o], z);
    }
    const float rwd = tanhf(z);
    if (lane == 0) a.reward[b] = rwd;
    if constexpr (SUMS) {
      const double rr = (double)rwd, de = d0_cur + rr;  // delta = r + discount V(pi') - V(pi)   (ac_irl.py:691)
      if (lane == 0) a.delta_out[b] = de;
      const double dgc = de * g_cur;
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        const double x = (double)xs[e_ia[q]] * (double)xs[e_ib[q]];
        const double coef = e_cm[q] == 0 ? de : (e_cm[q] == 1 ? dgc : (e_cm[q] == 2 ? rr : (e_cm[q] == 3 ? 1.0 : 0.0)));
        e_acc[q] = fma(coef, x, e_acc[q]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (SUMS) {
    // (a geometry with D > 31, or conv kernels so small that the tiles are shorter than the rows, would overflow LDS silently)
    static_assert(D + 1 <= 32, "the per-wave line [state | 1] of the SUMS variant holds 32 floats");
    static_assert(RN_WAVES * FO * 8 <= RN_WAVES * (Gm::T1 + Gm::T2) * 4, "the waves' partial rows reuse the tile region");
    __syncthreads();  // every wave is done with its tiles: they now hold the waves' rows
    double* rows = reinterpret_cast<double*>(smem + off);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const int k = lane + q * WAVE;
      if (k < FO) rows[wv * FO + k] = e_acc[q];
    }
    __syncthreads();
    for (int k = tid; k < FO; k += RN_BLOCK) {
      double t = rows[k];
#pragma unroll
      for (int w_ = 1; w_ < RN_WAVES; ++w_) t += rows[w_ * FO + k];
      a.part_rows[(int64_t)blockIdx.x * FO + k] = t;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// FC3 on the matrix cores (round 4; reference networks.py:67-69: the [2 d^2] -> [n_fc3] layer is 70 % of the network's
// multiply-adds).  Across SAMPLES the layer is a GEMM  H[16 samples x n3] = ACT[16 x 2d^2] . W3^T[2d^2 x n3]  -- but a
// wave that owns 16 samples would run their convolutions one after the other (no good at B = 4 096, where every wave
// has ONE sample).  So the block is 16 waves = 16 samples and the GEMM is split along K:
//   phase 1  wave s evaluates the convolutions of sample s -- lane = (strip of RUN columns, row), the rows of a strip in
//            consecutive lanes: a lane reads its own row from the LDS tile and takes the rows above / below from the
//            neighbouring lanes (DPP wave shifts), the side columns of the conv1 map from the neighbouring strips (lane
//            permutes) -- and leaves the 2 d^2 activations of its sample as row s of an LDS matrix (pitch = 2 mod 32: the
//            operand reads of a 32-lane group touch 32 distinct banks);
//   phase 2  wave w owns the k-slice [w KW, (w+1) KW) of ALL 16 samples: NSTEP = KW / 4 v_mfma_f32_16x16x4_f32 with
//            A = activations (one ds_read_b32 per step) and B = ITS slice of the FC3 weights, which it fetched ONCE at
//            kernel start and keeps in NSTEP registers -- the weights are neither staged in LDS nor re-read from L2 per
//            sample (the run-mapped kernel reads all 28 KB per sample: 115 MB of L2 traffic per 4 096-sample launch);
//            the 16 x 16 partial product (4 registers) goes to LDS;
//   phase 3  wave s adds the 16 partials of its sample (4 LDS reads + two lane swaps, a fixed order), and finishes
//            FC3's ReLU / dropout, FC4 (four units at a time), the output unit (and the SUMS fold).
// Two block barriers per 16 samples.  fp32 multiply-add on the matrix cores (no reduced precision; on gfx950 the same rate as
// the vector unit: what is saved is the LDS traffic and the per-unit reductions).  n3 <= 16, FC3 weights 8-byte aligned.
// ---------------------------------------------------------------------------------------------
#ifndef MFG_RN_MFMA
#define MFG_RN_MFMA 1
#endif
#ifndef MFG_RM_P21
#define MFG_RM_P21 43  // tile pitch of the matrix-core kernel at d = 21: lane (strip r, row y) reads at y P + 7 r -- with
#endif                 // P = 11 mod 32 the 32 lanes of a group hit 32 different banks (any other pitch below 53: 2 or 3 deep)
#ifndef MFG_RM_P15
#define MFG_RM_P15 21
#endif
constexpr int RM_WAVES = 16, RM_BLOCK = RM_WAVES * WAVE, RM_RED = 260;  // RM_RED: floats per wave's partial (= 4 mod 64 x 4)
typedef float rn_v4f_u __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte global access at any 4-byte address

// Entry k of the row [sum delta phi | sum delta g | sum r | count]: positions of its two factors in the line [state | 1]
// and the coefficient kind (0 delta, 1 delta g, 2 r, 3 one, -1 none), packed ia | ib << 8 | (kind + 1) << 16; built at
// compile time (mfg_ac2.py:333 feature order: row-major upper triangle, then the linear terms, then the bias).
template <int D>
struct SumsTable {
  static constexpr int Qs = D * (D + 1) / 2, Fs = Qs + D + 1, FO = Fs + 3, N = ((FO + WAVE - 1) / WAVE) * WAVE;
  uint32_t e[N];
  constexpr SumsTable() : e{} {
    for (int k = 0; k < N; ++k) {
      int ia = D, ib = D, cm = -1;
      if (k < Qs) {
        int i = 0;
        while (i + 1 < D && (i + 1) * D - ((i + 1) * i) / 2 <= k) ++i;
        ia = i;
        ib = i + (k - (i * D - (i * (i - 1)) / 2));
        cm = 0;
      } else if (k < Qs + D) {
        ia = k - Qs;
        cm = 0;
      } else if (k == Qs + D) {
        cm = 0;
      } else if (k < FO) {
        cm = k - Fs + 1;
      }
      e[k] = (uint32_t)ia | ((uint32_t)ib << 8) | ((uint32_t)(cm + 1) << 16);
    }
  }
};

constexpr bool rows_hit_distinct_even_banks(int stride) {  // 16 rows `stride` floats apart: 16 different even banks of 32
  unsigned seen = 0;
  for (int n = 0; n < 16; ++n) {
    const int b = (stride * n) % 32;
    if ((b & 1) || ((seen >> b) & 1u)) return false;
    seen |= 1u << b;
  }
  return true;
}

template <int D, int RUN, int RPR, int P1>
struct MfmaGeom {
  static_assert(RUN * RPR == D && D * RPR <= WAVE, "runs must tile a row exactly and fit one wavefront");
  static constexpr int K1 = 5, K2 = 3, F2 = 2, H1 = 2, H2 = 1, DD = D * D;
  static constexpr int PP = (DD + WAVE - 1) / WAVE;
  static constexpr int K = F2 * DD;                         // FC3 inputs
  static constexpr int NSTEP = (K + 4 * RM_WAVES - 1) / (4 * RM_WAVES);  // matrix instructions per wave and group
  static constexpr int KW = 4 * NSTEP, KP = RM_WAVES * KW;  // k's per wave, padded K
  // ds_read_b32 is served in two 32-lane groups over 32 banks: the A-operand read of a group touches samples 0..15 at
  // k, k + 1 -> banks (s PITCH + {0, 1}) mod 32 must be 32 different ones: PITCH = 2 mod 32
  static constexpr int PITCH = ((KP - 2 + 31) / 32) * 32 + 2;
  static_assert(PITCH >= KP && (PITCH & 31) == 2, "activation rows: room for the padded K, conflict-free operand reads");
  // the wave's tile: the D rows of the action with a zero halo of H1 columns on either side (no rows above or below: the
  // vertical neighbours come from the neighbouring lanes), pitch P1; before the first group it is the scratch of the
  // weight transposition (16 rows of KW + 2)
  static_assert(P1 >= D + 2 * H1, "tile pitch");
  static constexpr int T1 = D * P1, WS = KW + 2, TS = ((T1 > 16 * WS ? T1 : 16 * WS) + 3) & ~3;
  // SUMS: two lines [state | 1], [next state | 1] per wave and the critic weights as doubles, zero-padded to whole waves of entries
  static constexpr int SUMS_WPAD = ((D * (D + 1) / 2 + D + 1 + 3 + WAVE - 1) / WAVE) * WAVE;
  static size_t lds_floats(int n3, int n4, bool sums) {
    size_t fl = (size_t)(n4 * (n3 + D) + 2 * n4 + 1 + n3);
    fl = (fl + 3) & ~(size_t)3;
    return fl + (size_t)RM_WAVES * (PITCH + RM_RED + TS) + 2 * RM_WAVES * 16 + (sums ? RM_WAVES * 64 + 2 * SUMS_WPAD : 0);
  }
};

// (RnConstF, relu_f32, lane_below / lane_above: mfg_rn_common.h)

// Developer build (-DMFG_RN_STAMPS, tools/rn_stamps.py): shader-clock stamps of the phases of blocks 0 and 100, every wave,
// first group; read back with mfg_debug_rn_stamps.
#ifdef MFG_RN_STAMPS
__device__ unsigned long long rn_stamps[2 * RM_WAVES * 16];
#define RN_STAMP(i)                                                                                          \
  if ((blockIdx.x == 0 || blockIdx.x == 100) && lane == 0 && first_pass)                                     \
    rn_stamps[((blockIdx.x ? 1 : 0) * RM_WAVES + wv) * 16 + (i)] = __builtin_readcyclecounter();
#else
#define RN_STAMP(i)
#endif

template <int D, int RUN, int RPR, int P1, bool SUMS>
__global__ __launch_bounds__(RM_BLOCK) void k_reward_net_mfma(RewardNetArgs a) {
  using Gm = MfmaGeom<D, RUN, RPR, P1>;
  constexpr int K1 = Gm::K1, K2 = Gm::K2, F2 = Gm::F2, H1 = Gm::H1, H2 = Gm::H2, DD = Gm::DD, PP = Gm::PP;
  constexpr int KK = Gm::K, NSTEP = Gm::NSTEP, KW = Gm::KW, PITCH = Gm::PITCH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n3 = a.n3, n4 = a.n4, nin = n3 + D;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);  // (uniform: per-sample scalars come through the scalar cache)
  bool first_pass = true;
  (void)first_pass;
  RN_STAMP(0)
  float* s_w4 = smem;
  float* s_b4 = s_w4 + n4 * nin;
  float* s_wo = s_b4 + n4;
  float* s_bo = s_wo + n4;
  float* s_b3 = s_bo + 1;
  int off = n4 * nin + 2 * n4 + 1 + n3;
  off = (off + 3) & ~3;
  float* acts = smem + off;                          // [16 samples][PITCH]
  float* red = acts + RM_WAVES * PITCH;              // [16 k-slices][RM_RED]: register r of lane l at r*64 + l
  float* tiles = red + RM_WAVES * RM_RED;
  float* tin = tiles + wv * Gm::TS;
  // the first group's action is in flight under everything below
  const int64_t ngroups = (a.B + RM_WAVES - 1) / RM_WAVES;
  int64_t g = blockIdx.x;
  int64_t b = g * RM_WAVES + wv;
  float av[PP], st_mine = 0.0f, sn_mine = 0.0f;
  double g_next = 0.0;
  // (every load of the prologue is UNCONDITIONAL, from a clamped address, and masked afterwards: loads under branches make
  //  the compiler wait for all outstanding loads at the first use of any of them -- including the weight slice below)
  {
    const int64_t bc = b < a.B ? b : 0;
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      const bool ok = b < a.B && lane + q * WAVE < DD;
      const float v = a.action[bc * DD + (lane + q * WAVE < DD ? lane + q * WAVE : 0)];
      av[q] = ok ? v : 0.0f;
    }
    const bool st_ok = b < a.B && lane >= n3 && lane < nin;
    const float sv = a.state[rn_state_row(a, bc) * D + (lane >= n3 && lane < nin ? lane - n3 : 0)];
    st_mine = st_ok ? sv : 0.0f;
    if constexpr (SUMS) {
      const float nv = a.state_next[bc * D + (lane >= n3 && lane < nin ? lane - n3 : 0)];
      sn_mine = st_ok ? nv : 0.0f;
      g_next = a.gsc[bc];
    }
  }
  // This wave's slice of the FC3 weights, rows n < 16 x k in [wv KW, (wv + 1) KW): fetched as whole row pieces (16-byte
  // lanes -- rows start at multiples of 8 bytes only, global loads need no more --, RPI rows per instruction: 2 instructions
  // at d = 21, n3 = 8), transposed into the B-operand layout through the wave's own tile region before that is initialised
  // -- wave-local, no block barrier -- and kept in NSTEP registers for every group.
  // (The CU's address path takes ~16 cycles per vector-memory instruction whatever its width, and a launch of the per-step
  //  IRL update has ONE sample per wave: the prologue's instruction count is its cost.  Fetched directly in the operand
  //  layout the slice is 14 gathers per wave -- 6 000 cycles per block in front of the other waves' action loads or, issued
  //  later, in front of the last waves' convolutions; staged for the whole block in LDS it costs two block barriers.)
  // Rows >= n3 repeat row n3 - 1 (columns nobody reads); a piece that would pass the end of its row is fetched from the
  // row's last 16 bytes: its real entries (KK = 2 mod 4: two) are moved to the front, the padded k's behind them meet zero
  // activations -- any finite weight will do.
  constexpr int J4 = KW / 4, LPR = J4 > 8 ? 16 : 8, RPI = WAVE / LPR, NLD = 16 / RPI, WS = Gm::WS;
  static_assert(J4 <= LPR && (KK & 3) == 2 && (KW & 3) == 0 && 16 * WS <= Gm::TS && rows_hit_distinct_even_banks(WS),
                "weight scratch: fits the tile region; the operand read of a 32-lane group (16 rows x 2 k's) hits 32 banks");
  rn_v4f_u w3r[NLD];
  const int w3j = lane & (LPR - 1), w3nr = lane / LPR;
  const int w3k = wv * KW + 4 * w3j;                       // first k of this lane's piece
  const bool w3_half = w3k == KK - 2;                      // the piece holds the row's last two entries
  {
    const int kc = (w3j < J4 && w3k <= KK - 4) ? w3k : KK - 4;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int n = u * RPI + w3nr;
      if (u * RPI < n3) w3r[u] = *reinterpret_cast<const rn_v4f_u*>(a.w3 + (n < n3 ? n : n3 - 1) * KK + kc);
      else w3r[u] = w3r[0];
    }
  }
  // (All global reads of the prologue are issued before the first use of any: a launch of the per-step IRL update has one
  //  sample per wave, so the prologue is on the critical path -- with the parameter copies, a conv-weight gather and the SUMS
  //  table each waiting for their own loads it took 10 800 of the kernel's 31 000 cycles.)
  constexpr int NW1 = K1 * K1;
  // the small parameters and the SUMS table are the same for every wave: wave 0 fetches them, the others take them from LDS
  constexpr int Qs = D * (D + 1) / 2, Fs = Qs + D + 1, FO = Fs + 3, NPL = (FO + WAVE - 1) / WAVE;
  const int nw4 = n4 * nin;  // <= 32 * 37
  constexpr int NP4 = (32 * (16 + D) + WAVE - 1) / WAVE;
  uint32_t* s_tab = reinterpret_cast<uint32_t*>(red);  // (the partial-product buffer is idle until the first matrix phase)
  if (wv == 0) {
    float pw4[NP4];
#pragma unroll
    for (int q = 0; q < NP4; ++q) pw4[q] = q * WAVE < nw4 ? a.w4[lane + q * WAVE < nw4 ? lane + q * WAVE : 0] : 0.0f;
    const float pb4 = a.b4[lane < n4 ? lane : 0], pwo = a.wo[lane < n4 ? lane : 0], pb3 = a.b3[lane < n3 ? lane : 0];
    const float pbo = a.bo[0];
    double pwt[NPL];
    if constexpr (SUMS) {
#pragma unroll
      for (int q = 0; q < NPL; ++q) pwt[q] = a.td_w[lane + q * WAVE < Fs ? lane + q * WAVE : 0];
    }
    if constexpr (SUMS) {
      static constexpr SumsTable<D> tab{};
#pragma unroll
      for (int q = 0; q < NPL; ++q) s_tab[lane + q * WAVE] = tab.e[lane + q * WAVE];
    }
#pragma unroll
    for (int q = 0; q < NP4; ++q)
      if (lane + q * WAVE < nw4) s_w4[lane + q * WAVE] = pw4[q];
    if constexpr (SUMS) {
      double* s_wtd0 = reinterpret_cast<double*>(tiles + RM_WAVES * Gm::TS + 2 * RM_WAVES * 16 + RM_WAVES * 64);
#pragma unroll
      for (int q = 0; q < NPL; ++q) s_wtd0[lane + q * WAVE] = lane + q * WAVE < Fs ? pwt[q] : 0.0;
    }
    if (lane < n4) {
      s_b4[lane] = pb4;
      s_wo[lane] = pwo;
    }
    if (lane == 0) s_bo[0] = pbo;
    if (lane < n3) s_b3[lane] = pb3;
  }
  // the weight pieces -> scratch rows [unit][KW + 2] in the wave's tile region -> NSTEP operand registers (wave-local)
  float wreg[NSTEP];
  {
    if (w3j < J4) {
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        float* dst = tin + (u * RPI + w3nr) * WS + 4 * w3j;  // (8-byte aligned: WS is even)
        const rn_v4f_u v = w3r[u];
        *reinterpret_cast<float2*>(dst) = w3_half ? make_float2(v[2], v[3]) : make_float2(v[0], v[1]);
        *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wreg[s] = tin[(lane & 15) * WS + 4 * s + (lane >> 4)];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  // LDS initialisation: zero halos of the tile (interiors are rewritten per group); the padded k's of every activation row
  // stay zero (a row's real entries are rewritten per group; the rows of samples beyond B hold whatever LDS held -- row m of
  // A only reaches row m of the product)
  for (int k = lane; k < Gm::TS; k += WAVE) tin[k] = 0.0f;
  for (int k = tid; k < RM_WAVES * (PITCH - KK); k += RM_BLOCK) acts[(k / (PITCH - KK)) * PITCH + KK + k % (PITCH - KK)] = 0.0f;
  RN_STAMP(1)
  __syncthreads();
  uint32_t tabv[NPL];
  if constexpr (SUMS) {
#pragma unroll
    for (int q = 0; q < NPL; ++q) tabv[q] = s_tab[lane + q * WAVE];
  }
  RN_STAMP(2)
  const float inv_keep = 1.0f / a.keep_prob;
  const bool drop = a.keep_prob < 1.0f;
  // Lane -> run: strip r = lane / D (columns r RUN ... + RUN - 1), row y = lane % D -- the rows of a strip sit in
  // CONSECUTIVE lanes, so the rows above and below come from the neighbouring lanes with one DPP move per value (the
  // lanes at the ends of a strip skip the taps that would reach outside the image: their neighbours belong to another
  // strip).  Only the run's own row is read from LDS: 11 + 0 reads per sample instead of 55 + 27, and the conv1 map needs
  // no tile at all (its side columns come from the strips left and right with two lane permutes).  The last lane(s) beyond
  // RPR D compute on the last run's addresses and store nothing.
  const bool active = lane < D * RPR;
  const int la = active ? lane : D * RPR - 1;
  const int rs = la / D, y = la - rs * D, x0 = rs * RUN;
  const float* win1 = tin + y * P1 + x0;                // the run's row in the padded tile: columns x0 - H1 .. x0 + RUN - 1 + H1
  float2* act_out = reinterpret_cast<float2*>(acts + wv * PITCH + (y * D + x0) * F2);  // the run's 2 RUN inputs of FC3
  const float* act_in = acts + (lane & 15) * PITCH + wv * KW + (lane >> 4);             // A operand: sample lane % 16
  float* red_out = red + wv * RM_RED + lane;
  // phase 3 reads the partials of sample wv: lane -> unit n = lane % 16, k-slices 4 (lane / 16) ... + 3
  const float* red_in = red + (4 * (lane >> 4)) * RM_RED + (wv & 3) * 64 + 16 * (wv >> 2) + (lane & 15);
  int o1[PP];
#pragma unroll
  for (int q = 0; q < PP; ++q) {
    const int p = lane + q * WAVE;
    const int pc = p < DD ? p : 0;
    o1[q] = (pc / D) * P1 + pc % D + H1;
  }
  float* s_udrop = tiles + RM_WAVES * Gm::TS;  // [2][16 samples][16 slots]
  float* xs = s_udrop + 2 * RM_WAVES * 16 + wv * 64;  // (only allocated for SUMS launches): [state | 1], at + 32 [next state | 1]
  const double* s_wtd = reinterpret_cast<const double*>(s_udrop + 2 * RM_WAVES * 16 + RM_WAVES * 64);  // critic weights
  double e_acc[NPL];
  if constexpr (SUMS) {
    // (the run-mapped kernel derives the entries with a search per lane: ~800 instructions per wave, as much as a sample's
    //  convolutions; here they stay packed in one register each and are unpacked at the fold)
#pragma unroll
    for (int q = 0; q < NPL; ++q) e_acc[q] = 0.0;
    if (lane == 0) xs[D] = 1.0f;
    if (lane == 1) xs[32 + D] = 1.0f;
  }
  const bool shared_u = n3 + n4 <= 16;
  int u_par = 0;
  RN_STAMP(3)
  for (; g < ngroups; g += gridDim.x) {  // (block-uniform trip count: the barriers below are taken by all 16 waves)
    b = g * RM_WAVES + wv;
    const bool valid = b < a.B;
    // Lane predicates (lane < n3, y >= 1, ...) are formed where they are used, from copies of the lane's coordinates that
    // the compiler cannot see through: hoisted out of the loop they are ~30 lane masks = 60 scalar registers that do not
    // fit -- spilled to the lanes of a register they cost ~330 instructions of set-up on the critical path of a
    // one-pass launch and ~60 lane moves per pass.
    int ln = lane, yv = y, rv = rs;
    asm volatile("" : "+v"(ln), "+v"(yv), "+v"(rv));
    // 1. action -> padded LDS tile; the next group's sample is fetched under this one's evaluation
#pragma unroll
    for (int q = 0; q < PP; ++q)
      if (ln + q * WAVE < DD) tin[o1[q]] = av[q];
    const float st_cur = st_mine;
    const double g_cur = g_next;
    if (SUMS && ln >= n3 && ln < nin) {
      xs[lane - n3] = st_cur;
      xs[32 + lane - n3] = sn_mine;
    }
    {
      // (one uniform branch, unconditional loads from clamped addresses: pixels beyond DD are never written to the tile, the
      //  state entry is masked where it is used)
      const int64_t bn = b + (int64_t)gridDim.x * RM_WAVES;
      if (bn < a.B) {
        const float* src = a.action + bn * DD;
#pragma unroll
        for (int q = 0; q < PP; ++q) av[q] = src[(q + 1) * WAVE <= DD ? lane + q * WAVE : (ln + q * WAVE < DD ? lane + q * WAVE : 0)];
        st_mine = a.state[rn_state_row(a, bn) * D + (ln >= n3 && ln < nin ? lane - n3 : 0)];
        if constexpr (SUMS) {
          sn_mine = a.state_next[bn * D + (ln >= n3 && ln < nin ? lane - n3 : 0)];
          g_next = a.gsc[bn];
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // value part of the TD error from the lane's feature entries: delta0 = sum_k w_k (gamma phi_k(next) - phi_k(state)); the wave
    // sum is not needed before the fold at the end of the pass
    double d0_cur = 0.0;
    if constexpr (SUMS) {
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        if (q * WAVE < Fs) {
          const int ia = tabv[q] & 0xFFu, ib = (tabv[q] >> 8) & 0xFFu;
          const double x = (double)xs[ia] * (double)xs[ib], xn = (double)xs[32 + ia] * (double)xs[32 + ib];
          acc = fma(s_wtd[lane + q * WAVE], fma(a.td_gamma, xn, -x), acc);   // (weights beyond F are staged as zeros)
        }
      }
      d0_cur = wave_sum_dpp(acc);
    }
    // 2. conv1 5x5 + ReLU over the run: the run's own row from LDS, rows y -+ 1, y -+ 2 from the lanes below / above.
    // The weights of a convolution come in through the scalar cache right before it (read-only memory, uniform addresses:
    // scalar loads, no vector instruction) -- all 46 of them held for the whole loop do not fit next to the kernel's
    // pointers, every spilled one costs a lane move per use (142 in the first version of this loop), and broadcasting them
    // from a register costs a v_readlane each (53 per sample).
    RnConstF c1w_s = (RnConstF)a.c1w, c1b_s = (RnConstF)a.c1b, c2w_s = (RnConstF)a.c2w, c2b_s = (RnConstF)a.c2b;
    asm volatile("" : "+s"(c1w_s), "+s"(c1b_s), "+s"(c2w_s), "+s"(c2b_s));  // (the loads stay in the loop)
    RN_STAMP(12)
    float c1[RUN];
    {
      float w1[NW1];
#pragma unroll
      for (int k = 0; k < NW1; ++k) w1[k] = c1w_s[k];
      const float b1 = c1b_s[0];
      constexpr int W = RUN + K1 - 1;
      float x0r[W], xs_[W];
#pragma unroll
      for (int t = 0; t < W; ++t) x0r[t] = win1[t];
#pragma unroll
      for (int k = 0; k < RUN; ++k) c1[k] = b1;
#pragma unroll
      for (int k = 0; k < RUN; ++k)
#pragma unroll
        for (int dx = 0; dx < K1; ++dx) c1[k] = fmaf(x0r[k + dx], w1[H1 * K1 + dx], c1[k]);
      RN_STAMP(13)
#pragma unroll
      for (int t = 0; t < W; ++t) xs_[t] = x0r[t];
#pragma unroll
      for (int e = 1; e <= H1; ++e) {  // rows y - e
#pragma unroll
        for (int t = 0; t < W; ++t) xs_[t] = lane_below(xs_[t]);
        if (yv >= e) {
#pragma unroll
          for (int k = 0; k < RUN; ++k)
#pragma unroll
            for (int dx = 0; dx < K1; ++dx) c1[k] = fmaf(xs_[k + dx], w1[(H1 - e) * K1 + dx], c1[k]);
        }
      }
      RN_STAMP(14)
#pragma unroll
      for (int t = 0; t < W; ++t) xs_[t] = x0r[t];
#pragma unroll
      for (int e = 1; e <= H1; ++e) {  // rows y + e
#pragma unroll
        for (int t = 0; t < W; ++t) xs_[t] = lane_above(xs_[t]);
        if (yv + e < D) {
#pragma unroll
          for (int k = 0; k < RUN; ++k)
#pragma unroll
            for (int dx = 0; dx < K1; ++dx) c1[k] = fmaf(xs_[k + dx], w1[(H1 + e) * K1 + dx], c1[k]);
        }
      }
    }
    RN_STAMP(4)
    // 3. conv2 3x3, two filters + ReLU -> row wv of the activation matrix (NHWC order: pixel * 2 + channel)
    // (the two filters of a pixel advance together in one packed FMA: same input, weight pair (filter 0, filter 1))
    rn_v2f_t a2[RUN];
    {
      static_assert(H2 == 1 && F2 == 2, "one column from either neighbouring strip; two filters per packed instruction");
      constexpr int W = RUN + K2 - 1;
      rn_v2f_t w2[K2 * K2];
#pragma unroll
      for (int k = 0; k < K2 * K2; ++k) w2[k] = rn_v2f_t{c2w_s[k], c2w_s[K2 * K2 + k]};
      const rn_v2f_t b2 = {c2b_s[0], c2b_s[1]};
      float m0[W], ms[W];
#pragma unroll
      for (int k = 0; k < RUN; ++k) m0[k + 1] = relu_f32(c1[k]);
      // the columns left and right of the run: the last / first value of the same row in the neighbouring strips
      const float lft = __shfl(m0[RUN], (lane - D) & (WAVE - 1), WAVE), rgt = __shfl(m0[1], (lane + D) & (WAVE - 1), WAVE);
      m0[0] = rv > 0 ? lft : 0.0f;
      m0[W - 1] = rv + 1 < RPR ? rgt : 0.0f;
#pragma unroll
      for (int k = 0; k < RUN; ++k) a2[k] = b2;
#pragma unroll
      for (int k = 0; k < RUN; ++k)
#pragma unroll
        for (int dx = 0; dx < K2; ++dx) a2[k] = __builtin_elementwise_fma(rn_v2f_t{m0[k + dx], m0[k + dx]}, w2[H2 * K2 + dx], a2[k]);
#pragma unroll
      for (int t = 0; t < W; ++t) ms[t] = lane_below(m0[t]);
      if (yv >= 1) {
#pragma unroll
        for (int k = 0; k < RUN; ++k)
#pragma unroll
          for (int dx = 0; dx < K2; ++dx)
            a2[k] = __builtin_elementwise_fma(rn_v2f_t{ms[k + dx], ms[k + dx]}, w2[(H2 - 1) * K2 + dx], a2[k]);
      }
#pragma unroll
      for (int t = 0; t < W; ++t) ms[t] = lane_above(m0[t]);
      if (yv + 1 < D) {
#pragma unroll
        for (int k = 0; k < RUN; ++k)
#pragma unroll
          for (int dx = 0; dx < K2; ++dx)
            a2[k] = __builtin_elementwise_fma(rn_v2f_t{ms[k + dx], ms[k + dx]}, w2[(H2 + 1) * K2 + dx], a2[k]);
      }
    }
    if (ln < D * RPR) {
#pragma unroll
      for (int k = 0; k < RUN; ++k) act_out[k] = make_float2(relu_f32(a2[k][0]), relu_f32(a2[k][1]));
    }
    // Dropout uniforms (unit o of FC3 / FC4 of sample n <- Philox4x32-10 with counter (o, 3 | 4, sample_offset + n, 0)): a
    // Philox block per lane costs ~100 instructions per wave whatever the number of lanes that need one, and a sample needs
    // n3 + n4 of them.  With n3 + n4 <= 16 the waves 0..3 (one per SIMD) draw for four samples each -- lane = (sample,
    // slot), slot < n3: FC3 unit, then the FC4 units -- and leave them in LDS (two buffers, by group parity: a wave may be a
    // whole pass ahead of the slowest reader); otherwise every wave draws for its own sample (lane o < 32: FC3 unit o, lane
    // 32 + o: FC4 unit o).
    float u_drop = 0.0f;
    float* s_u = s_udrop + u_par * (RM_WAVES * 16);
    u_par ^= 1;
    if (drop && (!shared_u || wv < 4)) {
      const int slot = ln & 15, smp = 4 * wv + (ln >> 4);
      const bool fc3 = shared_u ? slot < n3 : ln < 32;
      const uint32_t elem = shared_u ? (uint32_t)(fc3 ? slot : slot - n3) : (uint32_t)(lane & 31);
      const uint64_t traj = a.sample_offset + (uint64_t)(shared_u ? g * RM_WAVES + smp : b);
      uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
      asm volatile("" : "+s"(k0), "+s"(k1));  // (the ten round keys are formed here, not held in 20 scalar registers across the loop)
      u_drop = u01(philox_elem(((uint64_t)k1 << 32) | k0, elem, fc3 ? 3u : 4u, traj, 0).x);
      if (shared_u) s_u[smp * 16 + slot] = u_drop;
    }
    RN_STAMP(5)
    __syncthreads();  // the 16 activation rows are complete
    RN_STAMP(6)
    // 4. FC3, this wave's k-slice of all 16 samples: C[sample 4 (lane / 16) + r][unit lane % 16]
    {
      rn_v4f_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
      float av3[NSTEP];
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) av3[s] = act_in[4 * s];
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av3[s], wreg[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) red_out[r * 64] = acc[r];
    }
    RN_STAMP(7)
    __syncthreads();  // the 16 partial products are complete (and every wave is done reading the activations)
    RN_STAMP(8)
    // sample wv, unit lane % 16: the 16 k-slices in a fixed order
    float h3 = (red_in[0] + red_in[RM_RED]) + (red_in[2 * RM_RED] + red_in[3 * RM_RED]);
    {
      const auto r4 = __builtin_amdgcn_permlane16_swap(__float_as_uint(h3), __float_as_uint(h3), false, false);
      h3 = __uint_as_float(r4[0]) + __uint_as_float(r4[1]);  // lane bit 4
      const auto r5 = __builtin_amdgcn_permlane32_swap(__float_as_uint(h3), __float_as_uint(h3), false, false);
      h3 = __uint_as_float(r5[0]) + __uint_as_float(r5[1]);  // lane bit 5
    }
    // ReLU (+ dropout); lane o < n3 keeps unit o, lanes n3 .. n3+D-1 hold the state: `x4` is FC4's input
    float x4 = (ln >= n3 && ln < nin) ? st_cur : 0.0f;
    {
      float h = fmaxf(h3 + s_b3[ln < n3 ? ln : 0], 0.0f);
      if (drop) {
        const float uo = shared_u ? s_u[wv * 16 + (lane & 15)] : u_drop;  // (own draw: lane o < 32 drew unit o of FC3)
        h = (uo <= a.keep_prob) ? h * inv_keep : 0.0f;
      }
      if (ln < n3) x4 = h;
    }
    // 5. FC4 over [h3, state] + ReLU (+ dropout), 6. output unit
    // Four units at a time.  Their products come from four unconditional LDS reads in flight together (lanes >= nin hold
    // x4 = 0 and read the last weight; units >= n4 repeat unit n4 - 1 and are dropped below), their wave sums advance in
    // step, and the rest of the layer is lane-parallel: lane u < 4 takes the sum, bias, dropout uniform and output weight
    // of unit o0 + u, the four contributions to z are added across the quad.  (One unit after the other, with guarded
    // reads, the layer was ~250 instructions in four dependent chains per sample -- with four waves per SIMD the phase is
    // bound by instruction issue.)
    float z = 0.0f;  // lanes 0..3: partial sums of z; the rest stays 0
    const int lc = ln < nin ? ln : nin - 1;
    for (int o0 = 0; o0 < n4; o0 += 4) {
      float p4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p4[u] = x4 * s_w4[(o0 + u < n4 ? o0 + u : n4 - 1) * nin + lc];
      wave_sum4_to_lane63(p4);
      const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p4[0]), 63));
      const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p4[1]), 63));
      const float t2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p4[2]), 63));
      const float t3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p4[3]), 63));
      const int o = o0 + (ln & 3);
      const bool mine = ln < 4 && o < n4;
      const int oc = o < n4 ? o : n4 - 1;
      const float tot = (ln & 2) ? ((ln & 1) ? t3 : t2) : ((ln & 1) ? t1 : t0);
      float h4 = fmaxf(tot + s_b4[oc], 0.0f);
      if (drop) {
        // (own draw: unit o's uniform was drawn by lane 32 + o)
        const float uo = shared_u ? s_u[wv * 16 + ((n3 + oc) & 15)] : __shfl(u_drop, (32 + oc) & 63, WAVE);
        h4 = (uo <= a.keep_prob) ? h4 * inv_keep : 0.0f;
      }
      z = mine ? fmaf(h4, s_wo[oc], z) : z;
    }
    z += dpp_mov_f32<0xB1, 0xF>(z);  // quad_perm [1,0,3,2]
    z += dpp_mov_f32<0x4E, 0xF>(z);  // quad_perm [2,3,0,1]: lanes 0..3 hold the sum
    z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(z))) + s_bo[0];
    const float rwd = tanhf(z);
    if (valid && ln == 0) a.reward[b] = rwd;
    RN_STAMP(9)
    if constexpr (SUMS) {
      if (valid) {
        const double rr = (double)rwd, de = d0_cur + rr;  // delta = r + discount V(pi') - V(pi)   (ac_irl.py:691)
        if (ln == 0) a.delta_out[b] = de;
        const double dgc = de * g_cur;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
          const int cm = (int)(tabv[q] >> 16) - 1;
          const double x = (double)xs[tabv[q] & 0xFFu] * (double)xs[(tabv[q] >> 8) & 0xFFu];
          // (entries below Fs all carry delta: known per q at compile time, no selects)
          const double coef = (q + 1) * WAVE <= Fs ? de : (cm == 0 ? de : (cm == 1 ? dgc : (cm == 2 ? rr : (cm == 3 ? 1.0 : 0.0))));
          e_acc[q] = fma(coef, x, e_acc[q]);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    RN_STAMP(10)
    first_pass = false;
  }
  if constexpr (SUMS) {
    static_assert(D + 1 <= 32, "the per-wave line [state | 1] of the SUMS variant holds 32 floats");
    static_assert(RM_WAVES * FO * 8 <= RM_WAVES * Gm::TS * 4, "the waves' partial rows reuse the tile region");
    __syncthreads();
    double* rows = reinterpret_cast<double*>(tiles);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const int k = lane + q * WAVE;
      if (k < FO) rows[wv * FO + k] = e_acc[q];
    }
    __syncthreads();
    for (int k = tid; k < FO; k += RM_BLOCK) {
      double t = rows[k];
#pragma unroll
      for (int w_ = 1; w_ < RM_WAVES; ++w_) t += rows[w_ * FO + k];
      a.part_rows[(int64_t)blockIdx.x * FO + k] = t;
      if (k == Fs) a.col_f[blockIdx.x] = t;
    }
  }
#ifdef MFG_RN_STAMPS
  first_pass = true;
#endif
  RN_STAMP(11)
}

// dynamic LDS above 64 KB needs the attribute, which applies to the CURRENT device: once per device, result kept -- a device
// where it failed takes the run-mapped kernels
template <int D, int RUN, int RPR, int P1>
static bool mfma_lds_attribute() {
  static std::mutex attr_mu;
  static signed char attr_state[64] = {0};   // 0 = not tried, 1 = ok, -1 = failed
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(attr_mu);
  if (attr_state[dev] == 0) {
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_reward_net_mfma<D, RUN, RPR, P1, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_reward_net_mfma<D, RUN, RPR, P1, false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    attr_state[dev] = (e1 == hipSuccess && e2 == hipSuccess) ? 1 : -1;
    (void)hipGetLastError();
  }
  return attr_state[dev] > 0;
}

template <int D, int RUN, int RPR, int P1>
static int launch_reward_net_mfma(const RewardNetArgs& a, bool want_sums, int64_t max_rows, int* rows_out, hipStream_t st) {
  using Gm = MfmaGeom<D, RUN, RPR, P1>;
  int64_t grid = (a.B + RM_WAVES - 1) / RM_WAVES;
  if (grid > 256) grid = 256;  // one 16-wave block per CU (LDS: 137 KB at d = 21)
  const bool sums = want_sums && grid <= max_rows;
  const size_t lds = Gm::lds_floats(a.n3, a.n4, sums) * 4;
  // (return 1: the caller falls through to the run-mapped kernels)
  if (!mfma_lds_attribute<D, RUN, RPR, P1>()) return 1;
  if (sums) {
    hipLaunchKernelGGL((k_reward_net_mfma<D, RUN, RPR, P1, true>), dim3((unsigned)grid), dim3(RM_BLOCK), lds, st, a);
    *rows_out = (int)grid;
  } else {
    hipLaunchKernelGGL((k_reward_net_mfma<D, RUN, RPR, P1, false>), dim3((unsigned)grid), dim3(RM_BLOCK), lds, st, a);
  }
  return 0;
}

}  // namespace mfg

using namespace mfg;

namespace mfg {
static bool mfma_shape_ok(int d, int k1, int f2, int k2, int n3, const float* fc3_w) {
  return MFG_RN_MFMA && k1 == 5 && k2 == 3 && f2 == 2 && n3 <= 16 && (d == 21 || d == 15) && (((uintptr_t)fc3_w & 7) == 0);
}
bool reward_net_sums_td_ready(int64_t B, int d, int k1, int f2, int k2, int n3, int n4, const float* fc3_w, int64_t max_rows) {
  if (B < 1 || n3 < 1 || n4 < 1 || n4 > RN_MAXN || !mfma_shape_ok(d, k1, f2, k2, n3, fc3_w)) return false;
  int64_t grid = (B + RM_WAVES - 1) / RM_WAVES;
  if (grid > 256) grid = 256;
  if (grid > max_rows) return false;
  return d == 21 ? mfma_lds_attribute<21, 7, 3, MFG_RM_P21>() : mfma_lds_attribute<15, 5, 3, MFG_RM_P15>();
}

// sums != NULL: ask for the SUMS variant; *rows_out = partial rows written (0: this shape has no SUMS kernel -- the plain
// forward ran and the caller takes the separate gradient kernel)
int reward_net_forward_sums(const float* state, const float* action, int64_t B, int d, int k1, int f2, int k2, int n3, int n4,
                            const float* conv1_w, const float* conv1_b, const float* conv2_w, const float* conv2_b,
                            const float* fc3_w, const float* fc3_b, const float* fc4_w, const float* fc4_b,
                            const float* out_w, const float* out_b, float keep_prob, uint64_t seed, uint64_t sample_offset,
                            float* reward, const RnSums* sums, int* rows_out, mfg_stream_t stream, int state_T) {
  if (rows_out) *rows_out = 0;
  if (B < 0 || d < 1 || !state || !action || !reward || !conv1_w || !conv1_b || !conv2_w || !conv2_b || !fc3_w ||
      !fc3_b || !fc4_w || !fc4_b || !out_w || !out_b)
    return set_error(MFG_EINVAL, "reward_net: null pointer / bad shape");
  if (!(keep_prob > 0.0f && keep_prob <= 1.0f)) return set_error(MFG_EINVAL, "reward_net: keep_prob must be in (0,1]");
  if (d > 32 || f2 < 1 || f2 > RN_MAXF2 || n3 < 1 || n3 > RN_MAXN || n4 < 1 || n4 > RN_MAXN || (k1 & 1) == 0 ||
      (k2 & 1) == 0 || k1 > 7 || k2 > 7)
    return set_error(MFG_EUNSUPPORTED, "reward_net: supported d <= 32, f2 <= 2, n_fc <= 32, odd kernels <= 7");
  if (B == 0) return MFG_OK;
  RewardNetArgs a{state, action, B, d, k1, f2, k2, n3, n4, conv1_w, conv1_b, conv2_w, conv2_b, fc3_w, fc3_b,
                  fc4_w, fc4_b, out_w, out_b, keep_prob, seed, sample_offset, reward, 0, nullptr, nullptr, nullptr, nullptr};
  a.state_T = state_T;
  if (sums) {
    a.delta0 = sums->delta0;
    a.gsc = sums->g;
    a.delta_out = sums->delta_out;
    a.part_rows = sums->part_rows;
    a.td_w = sums->td_w;
    a.state_next = sums->state_next;
    a.td_gamma = sums->td_gamma;
    a.col_f = sums->col_f;
  }
  const int dd = d * d;
  const int W1 = d + 2 * (k1 / 2), W2 = d + 2 * (k2 / 2);
  size_t fl = (size_t)(k1 * k1 + 1) + (size_t)(f2 * k2 * k2 + f2) + (size_t)(n4 * (n3 + d) + 2 * n4 + 1 + n3);
  fl = (fl + 3) & ~(size_t)3;
  const size_t w3fl = (size_t)n3 * f2 * dd;
  int64_t grid = (B + RN_WAVES - 1) / RN_WAVES;
  if (grid > 256 * MFG_RN_BPC) grid = 256 * MFG_RN_BPC;
  // stage the FC3 weights in LDS only when a block amortises the copy over enough samples (and the pointer is
  // 16-byte aligned); otherwise they are read straight from L2 (coalesced, 28 KB at d = 21).
  // (Round 4, tried for the one-sample-per-wave launches of the per-step IRL update, B = 4 096: fetch the weights into
  //  registers under the convolutions and commit them to LDS in front of the first FC3 -- 19.7 -> 20.5 us per launch and
  //  0.527 -> 0.625 ms per 15-step episode: the block barrier in front of FC3 makes all eight waves wait for the slowest
  //  convolution, which costs more than the L2 round trips it removes.  Not kept.)
  const int64_t samples_per_block = (B + grid - 1) / grid;
  a.w3_in_lds = (w3fl * 4 <= 64 * 1024 && samples_per_block >= MFG_RN_LDS_MIN && (((uintptr_t)fc3_w & 15) == 0)) ? 1 : 0;
  if (a.w3_in_lds) fl += w3fl;
  fl = (fl + 3) & ~(size_t)3;
  fl += (size_t)RN_WAVES * (W1 * W1 + W2 * W2);
  const size_t lds = fl * 4;
  const int pp = (dd + WAVE - 1) / WAVE;
  hipStream_t st = (hipStream_t)stream;
  const bool ref_geom = (k1 == 5 && k2 == 3 && f2 == 2);
#define RN_LAUNCH(PP)                                                                                                 \
  if (ref_geom) hipLaunchKernelGGL((k_reward_net<PP, 5, 3, 2, 0>), dim3((unsigned)grid), dim3(RN_BLOCK), lds, st, a); \
  else hipLaunchKernelGGL((k_reward_net<PP, 0, 0, 0, 0>), dim3((unsigned)grid), dim3(RN_BLOCK), lds, st, a);
  // w3 rows are read as float2 at even offsets: needs an 8-byte aligned fc3_w when it is not staged in LDS
  const bool runs_ok = ref_geom && (a.w3_in_lds || (((uintptr_t)fc3_w & 7) == 0));
  const bool want_sums = sums && sums->delta0 && sums->g && sums->delta_out && sums->part_rows && grid <= sums->max_rows;
  // (the matrix-core SUMS kernel forms the TD error itself: it wants the critic weights and the next states, not delta0)
  const bool sums_ptrs = sums && sums->td_w && sums->state_next && sums->col_f && sums->g && sums->delta_out && sums->part_rows;
  const bool mfma_ok = mfma_shape_ok(d, k1, f2, k2, n3, fc3_w);
  bool mfma_done = false;
  if (mfma_ok) {
    int rows = 0;
    const int rc = d == 21 ? launch_reward_net_mfma<21, 7, 3, MFG_RM_P21>(a, sums_ptrs, sums_ptrs ? sums->max_rows : 0, &rows, st)
                           : launch_reward_net_mfma<15, 5, 3, MFG_RM_P15>(a, sums_ptrs, sums_ptrs ? sums->max_rows : 0, &rows, st);
    if (rc == 0) {
      mfma_done = true;
      if (rows_out) *rows_out = rows;
    }
  }
  if (mfma_done) {
  } else if (runs_ok && d == 21) {
    using Gm = RunsGeom<21, 7, 3, MFG_RN_P21, MFG_RN_P21>;
    if (want_sums) {
      hipLaunchKernelGGL((k_reward_net_runs<21, 7, 3, MFG_RN_P21, MFG_RN_P21, true>), dim3((unsigned)grid), dim3(RN_BLOCK),
                         (Gm::lds_floats(n3, n4, a.w3_in_lds != 0) + RN_WAVES * 32) * 4, st, a);
      *rows_out = (int)grid;
    } else {
      hipLaunchKernelGGL((k_reward_net_runs<21, 7, 3, MFG_RN_P21, MFG_RN_P21>), dim3((unsigned)grid), dim3(RN_BLOCK),
                         Gm::lds_floats(n3, n4, a.w3_in_lds != 0) * 4, st, a);
    }
  } else if (runs_ok && d == 15) {
    using Gm = RunsGeom<15, 5, 3, MFG_RN_P15, MFG_RN_P15>;
    if (want_sums) {
      hipLaunchKernelGGL((k_reward_net_runs<15, 5, 3, MFG_RN_P15, MFG_RN_P15, true>), dim3((unsigned)grid), dim3(RN_BLOCK),
                         (Gm::lds_floats(n3, n4, a.w3_in_lds != 0) + RN_WAVES * 32) * 4, st, a);
      *rows_out = (int)grid;
    } else {
      hipLaunchKernelGGL((k_reward_net_runs<15, 5, 3, MFG_RN_P15, MFG_RN_P15>), dim3((unsigned)grid), dim3(RN_BLOCK),
                         Gm::lds_floats(n3, n4, a.w3_in_lds != 0) * 4, st, a);
    }
  } else if (pp <= 4) { RN_LAUNCH(4) }
  else if (pp <= 7) { RN_LAUNCH(7) }
  else { RN_LAUNCH(16) }
#undef RN_LAUNCH
  return hipGetLastError() == hipSuccess ? MFG_OK : set_error(MFG_ELAUNCH, "reward_net: launch failed");
}
}  // namespace mfg

extern "C" int mfg_reward_net_forward(const float* state, const float* action, int64_t B, int d, int k1, int f2, int k2,
                                      int n3, int n4, const float* conv1_w, const float* conv1_b, const float* conv2_w,
                                      const float* conv2_b, const float* fc3_w, const float* fc3_b, const float* fc4_w,
                                      const float* fc4_b, const float* out_w, const float* out_b, float keep_prob,
                                      uint64_t seed, uint64_t sample_offset, float* reward, mfg_stream_t stream) {
  return mfg::reward_net_forward_sums(state, action, B, d, k1, f2, k2, n3, n4, conv1_w, conv1_b, conv2_w, conv2_b, fc3_w, fc3_b,
                                      fc4_w, fc4_b, out_w, out_b, keep_prob, seed, sample_offset, reward, nullptr, nullptr, stream, 0);
}

#ifdef MFG_RN_STAMPS
extern "C" int mfg_debug_rn_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mfg::rn_stamps), sizeof(mfg::rn_stamps)) == hipSuccess ? 0 : -1;
}
#endif
