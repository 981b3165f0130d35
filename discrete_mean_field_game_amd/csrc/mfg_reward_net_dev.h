// Device-side evaluator of the IRL reward network r(pi, P) (networks.py:46-81) for ONE sample per wavefront, "run"
// mapping (reference geometry k1 = 5, k2 = 3, f2 = 2; compile-time d = 21 / 15).  Shared by the stand-alone forward kernel
// (mfg_reward_net.hip, k_reward_net_runs) and by the IRL step kernel that evaluates the network on the action tile while
// it is still in LDS (mfg_core.h, k_core_small<..., RN>): both call RnRunsEval::eval, so their rewards are bit-identical.
//
// A lane owns a horizontal RUN of pixels of one row (RPR runs per row, RUN * RPR = d: 63 lanes at d = 21, 45 at d = 15);
// for each kernel row it reads the RUN + k - 1 inputs under its run once and slides the taps over them in registers
// (55 + 27 LDS reads per sample instead of one per tap).  The run's 2 RUN FC3 inputs are contiguous in the NHWC-flattened
// weight rows (8-byte reads), the FC3 / FC4 reductions run on the DPP path, the conv weights sit in scalar registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfg_device.h"

namespace mfg {

// Weights of the network as the device sees them (PyTorch layouts, see include/mfg_hip.h mfg_reward_net_forward).
struct RnWeights {
  int n3, n4;
  const float *c1w, *c1b;  // [k1*k1], [1]
  const float *c2w, *c2b;  // [f2][k2*k2], [f2]
  const float *w3, *b3;    // [n3][f2*d*d] (input index (pixel*f2 + channel): TF NHWC flatten), [n3]
  const float *w4, *b4;    // [n4][n3+d], [n4]
  const float *wo, *bo;    // [n4], [1]
  float keep_prob;         // 1 -> no dropout
  uint64_t seed, sample_offset;
};

__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
  v += dpp_mov_f32<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f32<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f32<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_mov_f32<0x140, 0xF>(v);  // row_mirror
  v += dpp_mov_f32<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_mov_f32<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int D, int RUN, int RPR, int P1, int P2>
struct RunsGeom {
  static_assert(RUN * RPR == D && D * RPR <= WAVE, "runs must tile a row exactly and fit one wavefront");
  static constexpr int K1 = 5, K2 = 3, F2 = 2, H1 = 2, H2 = 1, DD = D * D;
  static constexpr int T1 = (D + 2 * H1) * P1, T2 = (D + 2 * H2) * P2;  // floats per padded tile (pitches P1, P2)
  static constexpr int PP = (DD + WAVE - 1) / WAVE;
  // floats of dynamic LDS: block-shared small weights [+ FC3 weights] + the padded tiles of `nwaves` waves
  __host__ __device__ static size_t shared_floats(int n3, int n4, bool w3_in_lds) {
    size_t fl = (size_t)(n4 * (n3 + D) + 2 * n4 + 1 + n3);
    fl = (fl + 3) & ~(size_t)3;
    if (w3_in_lds) fl += (size_t)n3 * F2 * DD;
    return (fl + 3) & ~(size_t)3;
  }
  static size_t lds_floats(int n3, int n4, bool w3_in_lds, int nwaves) {
    return shared_floats(n3, n4, w3_in_lds) + (size_t)nwaves * (T1 + T2);
  }
};

template <int D, int RUN, int RPR, int P1, int P2>
struct RnRunsEval {
  using Gm = RunsGeom<D, RUN, RPR, P1, P2>;
  static constexpr int K1 = Gm::K1, K2 = Gm::K2, F2 = Gm::F2, H1 = Gm::H1, H2 = Gm::H2, DD = Gm::DD, PP = Gm::PP;
  static constexpr int NW1 = K1 * K1, NW2 = F2 * K2 * K2;
  static_assert(NW1 + 1 + NW2 + F2 <= WAVE, "conv parameters must fit one wavefront");
  // block-shared LDS
  float *s_w4, *s_b4, *s_wo, *s_bo, *s_b3, *s3;
  // this wave's padded tiles
  float *tin, *tc1;
  // conv weights / biases (wave uniform: scalar registers)
  float w1[NW1], w2[F2][K2 * K2], b1, b20, b21;
  // lane constants
  bool active;
  const float* win1;
  float* out1;
  const float* win2;
  int w3off, o1[PP];
  int n3, n4, nin;
  float inv_keep, keep_prob;
  bool drop, w3_lds;
  const float* w3g;

  // Stage the block-shared small weights (and, if asked, the FC3 weights) into `smem`; `nthreads` threads of the block take
  // part.  Returns the floats used.  The caller must __syncthreads() before the first eval().
  __device__ __forceinline__ int stage_shared(const RnWeights& a, float* smem, int tid, int nthreads, bool w3_in_lds) {
    n3 = a.n3;
    n4 = a.n4;
    nin = n3 + D;  // FC4 input = [h3 (n3), state (D)], nin <= 64
    s_w4 = smem;   // [n4][nin]
    s_b4 = s_w4 + n4 * nin;
    s_wo = s_b4 + n4;
    s_bo = s_wo + n4;
    s_b3 = s_bo + 1;
    int off = n4 * nin + 2 * n4 + 1 + n3;
    off = (off + 3) & ~3;
    s3 = smem + off;
    if (w3_in_lds) off += n3 * F2 * DD;
    off = (off + 3) & ~3;
    for (int k = tid; k < n4 * nin; k += nthreads) s_w4[k] = a.w4[k];
    for (int k = tid; k < n4; k += nthreads) {
      s_b4[k] = a.b4[k];
      s_wo[k] = a.wo[k];
    }
    if (tid == 0) s_bo[0] = a.bo[0];
    for (int k = tid; k < n3; k += nthreads) s_b3[k] = a.b3[k];
    if (w3_in_lds) {
      const int nq = (n3 * F2 * DD) >> 2;
      const float4* src4 = reinterpret_cast<const float4*>(a.w3);
      float4* dst4 = reinterpret_cast<float4*>(s3);
      for (int k0 = 0; k0 < nq; k0 += 4 * nthreads) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * nthreads + tid;
          v[u] = (k < nq) ? src4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * nthreads + tid;
          if (k < nq) dst4[k] = v[u];
        }
      }
      for (int k = (nq << 2) + tid; k < n3 * F2 * DD; k += nthreads) s3[k] = a.w3[k];
    }
    w3_lds = w3_in_lds;
    w3g = a.w3;
    keep_prob = a.keep_prob;
    inv_keep = 1.0f / a.keep_prob;
    drop = a.keep_prob < 1.0f;
    return off;
  }

  // Per-wave set-up: zero the halos of the wave's tiles (interiors are rewritten per sample), gather the conv weights
  // into scalar registers, derive the lane's run.  `tiles` = this wave's T1 + T2 floats.
  __device__ __forceinline__ void init_wave(const RnWeights& a, float* tiles, int lane) {
    tin = tiles;
    tc1 = tin + Gm::T1;
    for (int k = lane; k < Gm::T1 + Gm::T2; k += WAVE) tin[k] = 0.0f;
    // conv weights and biases: ONE gather per wave (lane t holds entry t of [c1w | c1b | c2w | c2b], 46 values), then
    // v_readlane into scalar registers.  Plain `a.c1w[k]` reads are re-issued as vector loads for every sample (the
    // compiler cannot prove that the reward store does not alias them) and sat on the critical path.
    float wtab;
    {
      const float* src = lane < NW1 ? a.c1w + lane
                       : lane == NW1 ? a.c1b
                       : lane < NW1 + 1 + NW2 ? a.c2w + (lane - NW1 - 1)
                       : a.c2b + (lane < NW1 + 1 + NW2 + F2 ? lane - NW1 - 1 - NW2 : 0);
      wtab = *src;
    }
#pragma unroll
    for (int k = 0; k < NW1; ++k) w1[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), k));
    b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), NW1));
#pragma unroll
    for (int c = 0; c < F2; ++c)
#pragma unroll
      for (int k = 0; k < K2 * K2; ++k)
        w2[c][k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), NW1 + 1 + c * K2 * K2 + k));
    b20 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), NW1 + 1 + NW2));
    b21 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wtab), NW1 + 2 + NW2));
    // this lane's run: row y, columns x0 .. x0+RUN-1
    active = lane < D * RPR;
    const int y = active ? lane / RPR : 0, x0 = active ? (lane - y * RPR) * RUN : 0;
    win1 = tin + y * P1 + x0;                // top-left of the conv1 window in the padded input tile
    out1 = tc1 + (y + H2) * P2 + x0 + H2;    // this run inside the padded conv1 map
    win2 = tc1 + y * P2 + x0;                // top-left of the conv2 window
    w3off = (y * D + x0) * F2;               // the run's 2*RUN inputs inside an FC3 weight row
#pragma unroll
    for (int q = 0; q < PP; ++q) {
      const int p = lane + q * WAVE;
      const int pc = p < DD ? p : 0;
      o1[q] = (pc / D + H1) * P1 + pc % D + H1;
    }
  }

  // r(state, action) of one sample.  av[q] = action pixel lane + 64 q (row-major d x d); st = the state entry of lanes
  // n3 .. n3+D-1 (FC4's input vector is [h3 | state]), ignored elsewhere; sample = global sample index (dropout counter).
  __device__ __forceinline__ float eval(const RnWeights& a, const float (&av)[PP], float st, uint64_t sample, int lane) {
    // 1. action -> padded LDS tile (pixel p = lane + 64 q)
#pragma unroll
    for (int q = 0; q < PP; ++q)
      if (lane + q * WAVE < DD) tin[o1[q]] = av[q];
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
      const u32x4 r = philox_elem(a.seed, (uint32_t)(lane & 31), lane < 32 ? 3u : 4u, sample, 0);
      u_drop = u01(r.x);
    }
    // 4. FC3 + ReLU (+ dropout); lane o < n3 keeps unit o, lanes n3 .. n3+D-1 hold the state: `x4` is FC4's input
    float x4 = (lane >= n3 && lane < nin) ? st : 0.0f;
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
      if (drop) h = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(u_drop), o)) <= keep_prob) ? h * inv_keep : 0.0f;
      if (lane == o) x4 = h;
    }
    // 5. FC4 over [h3, state] + ReLU (+ dropout), 6. output unit: lane-parallel products, one DPP sum per unit
    float z = s_bo[0];
    for (int o = 0; o < n4; ++o) {
      const float wgt = lane < nin ? s_w4[o * nin + lane] : 0.0f;
      float h4 = fmaxf(wave_sum_f32_dpp(x4 * wgt) + s_b4[o], 0.0f);
      if (drop) h4 = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(u_drop), 32 + o)) <= keep_prob) ? h4 * inv_keep : 0.0f;
      z = fmaf(h4, s_wo[o], z);
    }
    return tanhf(z);
  }
};

}  // namespace mfg
