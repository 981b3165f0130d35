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

#include "../../include/mfg_hip.h"
#include "mfg_core.h"

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
};

constexpr int RN_BLOCK = 256, RN_WAVES = 4, RN_MAXF2 = 2, RN_MAXN = 32;

// PPMAX = max pixels per lane (ceil(d*d/64)).  K1 / K2 / F2 > 0: compile-time conv geometry (the reference always
// uses k1 = 5, k2 = 3, f2 = 2, ac_irl.py:251-267): taps unroll, LDS reads get immediate offsets and can be issued
// together; with run-time bounds every tap is a dependent ~100-cycle LDS round trip (the first version of this
// kernel spent 30 us per sample that way).  0 = generic run-time value.
template <int PPMAX, int K1, int K2, int F2>
__global__ __launch_bounds__(RN_BLOCK) void k_reward_net(RewardNetArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = a.d, dd = d * d, n3 = a.n3, n4 = a.n4;
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
  for (int64_t b = (int64_t)blockIdx.x * RN_WAVES + wv; b < a.B; b += nw) {
    const float* act = a.action + b * dd;
    // 1. action -> padded LDS tile
    for (int p = lane; p < dd; p += WAVE) {
      const int y = p / d, x = p - y * d;
      tin[(y + h1) * W1 + x + h1] = act[p];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // 2. conv1 (cross-correlation, SAME) + ReLU -> padded conv1 map
    for (int p = lane; p < dd; p += WAVE) {
      const int y = p / d, x = p - y * d;
      float s = sc1[k1 * k1];
      const float* tp = tin + y * W1 + x;
#pragma unroll
      for (int dy = 0; dy < (K1 ? K1 : k1); ++dy)
#pragma unroll
        for (int dx = 0; dx < (K1 ? K1 : k1); ++dx) s = fmaf(tp[dy * W1 + dx], sc1[dy * k1 + dx], s);
      tc1[(y + h2) * W2 + x + h2] = fmaxf(s, 0.0f);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // 3. conv2 + ReLU -> registers act2[pixel slot][channel]
    float act2[PPMAX][RN_MAXF2];
#pragma unroll
    for (int q = 0; q < PPMAX; ++q) {
      const int p = lane + q * WAVE;
#pragma unroll
      for (int c = 0; c < RN_MAXF2; ++c) act2[q][c] = 0.0f;
      if (p < dd) {
        const int y = p / d, x = p - y * d;
#pragma unroll
        for (int c = 0; c < RN_MAXF2; ++c) {
          if (c < f2) {
            float s = sc2[f2 * k2 * k2 + c];
            const float* tp = tc1 + y * W2 + x;
#pragma unroll
            for (int dy = 0; dy < (K2 ? K2 : k2); ++dy)
#pragma unroll
              for (int dx = 0; dx < (K2 ? K2 : k2); ++dx) s = fmaf(tp[dy * W2 + dx], sc2[c * k2 * k2 + dy * k2 + dx], s);
            act2[q][c] = fmaxf(s, 0.0f);
          }
        }
      }
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
      if (drop) {
        const u32x4 r = philox_elem(a.seed, (uint32_t)o, 3u, a.sample_offset + (uint64_t)b, 0);
        h = (u01(r.x) <= a.keep_prob) ? h * inv_keep : 0.0f;
      }
      if (lane == o) h3_mine = h;
    }
    // 5. FC4 over [h3, state] + ReLU (+ dropout): lane o < n4
    float h4 = 0.0f;
    {
      const int o = lane < n4 ? lane : 0;
      float s = s_b4[o];
      for (int k = 0; k < n3; ++k) s = fmaf(__shfl(h3_mine, k, WAVE), s_w4[o * (n3 + d) + k], s);
      const float* st = a.state + b * d;
      for (int k = 0; k < d; ++k) s = fmaf(st[k], s_w4[o * (n3 + d) + n3 + k], s);
      h4 = fmaxf(s, 0.0f);
      if (drop) {
        const u32x4 r = philox_elem(a.seed, (uint32_t)o, 4u, a.sample_offset + (uint64_t)b, 0);
        h4 = (u01(r.x) <= a.keep_prob) ? h4 * inv_keep : 0.0f;
      }
      h4 = (lane < n4) ? h4 * s_wo[o] : 0.0f;
    }
    // 6. output unit, tanh
    const float z = wave_sum(h4) + s_bo[0];
    if (lane == 0) a.reward[b] = tanhf(z);
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace mfg

using namespace mfg;

extern "C" int mfg_reward_net_forward(const float* state, const float* action, int64_t B, int d, int k1, int f2, int k2,
                                      int n3, int n4, const float* conv1_w, const float* conv1_b, const float* conv2_w,
                                      const float* conv2_b, const float* fc3_w, const float* fc3_b, const float* fc4_w,
                                      const float* fc4_b, const float* out_w, const float* out_b, float keep_prob,
                                      uint64_t seed, uint64_t sample_offset, float* reward, mfg_stream_t stream) {
  if (B < 0 || d < 1 || !state || !action || !reward || !conv1_w || !conv1_b || !conv2_w || !conv2_b || !fc3_w ||
      !fc3_b || !fc4_w || !fc4_b || !out_w || !out_b)
    return set_error(MFG_EINVAL, "reward_net: null pointer / bad shape");
  if (!(keep_prob > 0.0f && keep_prob <= 1.0f)) return set_error(MFG_EINVAL, "reward_net: keep_prob must be in (0,1]");
  if (d > 32 || f2 < 1 || f2 > RN_MAXF2 || n3 < 1 || n3 > RN_MAXN || n4 < 1 || n4 > RN_MAXN || (k1 & 1) == 0 ||
      (k2 & 1) == 0 || k1 > 7 || k2 > 7)
    return set_error(MFG_EUNSUPPORTED, "reward_net: supported d <= 32, f2 <= 2, n_fc <= 32, odd kernels <= 7");
  if (B == 0) return MFG_OK;
  RewardNetArgs a{state, action, B, d, k1, f2, k2, n3, n4, conv1_w, conv1_b, conv2_w, conv2_b, fc3_w, fc3_b,
                  fc4_w, fc4_b, out_w, out_b, keep_prob, seed, sample_offset, reward, 0};
  const int dd = d * d;
  const int W1 = d + 2 * (k1 / 2), W2 = d + 2 * (k2 / 2);
  size_t fl = (size_t)(k1 * k1 + 1) + (size_t)(f2 * k2 * k2 + f2) + (size_t)(n4 * (n3 + d) + 2 * n4 + 1 + n3);
  fl = (fl + 3) & ~(size_t)3;
  const size_t w3fl = (size_t)n3 * f2 * dd;
  int64_t grid = (B + RN_WAVES - 1) / RN_WAVES;
  if (grid > 256 * 3) grid = 256 * 3;
  // stage the FC3 weights in LDS only when a block amortises the copy over enough samples (and the pointer is
  // 16-byte aligned); otherwise they are read straight from L2 (coalesced, 28 KB at d = 21)
  const int64_t samples_per_block = (B + grid - 1) / grid;
  a.w3_in_lds = (w3fl * 4 <= 64 * 1024 && samples_per_block >= 16 && (((uintptr_t)fc3_w & 15) == 0)) ? 1 : 0;
  if (a.w3_in_lds) fl += w3fl;
  fl = (fl + 3) & ~(size_t)3;
  fl += (size_t)RN_WAVES * (W1 * W1 + W2 * W2);
  const size_t lds = fl * 4;
  const int pp = (dd + WAVE - 1) / WAVE;
  hipStream_t st = (hipStream_t)stream;
  const bool ref_geom = (k1 == 5 && k2 == 3 && f2 == 2);
#define RN_LAUNCH(PP)                                                                                              \
  if (ref_geom) hipLaunchKernelGGL((k_reward_net<PP, 5, 3, 2>), dim3((unsigned)grid), dim3(RN_BLOCK), lds, st, a); \
  else hipLaunchKernelGGL((k_reward_net<PP, 0, 0, 0>), dim3((unsigned)grid), dim3(RN_BLOCK), lds, st, a);
  if (pp <= 4) { RN_LAUNCH(4) }
  else if (pp <= 7) { RN_LAUNCH(7) }
  else { RN_LAUNCH(16) }
#undef RN_LAUNCH
  return hipGetLastError() == hipSuccess ? MFG_OK : set_error(MFG_ELAUNCH, "reward_net: launch failed");
}
