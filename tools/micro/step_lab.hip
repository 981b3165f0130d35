// Developer lab: where does the d = 21 given-P kernel (k_step_wave) lose bandwidth?  Variants of the shipped wave-private
// tile kernel with ONE aspect switched (column walk, output stores, tile assignment, batched stores), timed on the
// bench's 1.73 GB action slab.  Results are only timings: the ablated variants compute nothing meaningful.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/step_lab.hip -o tools/micro/step_lab
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#ifndef LAB_UNROLL
#define LAB_UNROLL 3
#endif
typedef float v4f_t __attribute__((ext_vector_type(4)));
constexpr int WAVE = 64;

// WALK: 0 = one LDS read per lane, 1 = the full fp64 column walk
// STORE: 0 = none, 1 = direct per tile (shipped), 2 = K consecutive tiles batched through an LDS line, 16-byte stores
// K: consecutive tiles per wave (1 = interleaved assignment as shipped)
// REW: 0 = no reward reduction / store, 1 = shipped
template <int FLAV>
__device__ __forceinline__ void store4(v4f_t* p, v4f_t v) {
  if (FLAV == 0) *p = v;
  else if (FLAV == 1) __builtin_nontemporal_store(v, p);
  else if (FLAV == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (FLAV == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (FLAV == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  else if (FLAV == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int WALK, int STORE, int K, int REW, int OCC, int FLAV = 0, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, (OCC * WAVES + 3) / 4) void k_lab(const float* __restrict__ pi, const float* __restrict__ P, int64_t B,
                                                    float* __restrict__ pi_next, float* __restrict__ reward, int64_t region64) {
  constexpr int D = 21, G = WAVE / D, DD = D * D;
  constexpr int WF = ((G * DD + 6 + 3) / 4) * 4;
  constexpr int PER = (WF / 4 + WAVE - 1) / WAVE;
  __shared__ __attribute__((aligned(16))) float sP[WAVES][WF];
  __shared__ float sQ[WAVES][G * D];
  __shared__ __attribute__((aligned(16))) float sO[WAVES][(STORE == 2 || STORE == 5) ? K * G * D + 4 : 4];
  __shared__ __attribute__((aligned(16))) float sR[WAVES][(STORE == 2 || STORE == 5) ? ((K * G + 3) / 4) * 4 : 4];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const int t = lane / D, j = lane - t * D;
  float* wP = sP[wv];
  float* wQ = sQ[wv];
  const int64_t ntiles = (B + G - 1) / G;
  const int64_t nsuper = ntiles / K;  // lab: B is a multiple of K*G
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  v4f_t pre[PER];
  float prepi = 0.0f;
  const v4f_t* P4 = reinterpret_cast<const v4f_t*>(P);
#define LAB_PREFETCH(TT)                                                    \
  {                                                                         \
    const int64_t f0 = (TT) * (int64_t)(G * DD);                            \
    const int64_t a4 = f0 >> 2;                                             \
    const int pn4 = (int)(((f0 & 3) + (int64_t)G * DD + 3) >> 2);           \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                       \
      const int k = lane + u * WAVE;                                        \
      pre[u] = (v4f_t)(0.0f);                                               \
      if (k < pn4) pre[u] = __builtin_nontemporal_load(P4 + a4 + k);        \
    }                                                                       \
    prepi = (lane < G * D) ? pi[(TT) * (int64_t)(G * D) + lane] : 0.0f;     \
  }
  int64_t sup = (int64_t)blockIdx.x * WAVES + wv;
  if (sup < nsuper) LAB_PREFETCH(sup * K)
  float keep = 0.0f;
  for (; sup < nsuper; sup += nwaves) {
#pragma unroll 1
    for (int kk = 0; kk < K; ++kk) {
      const int64_t tile = sup * K + kk;
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
        const int64_t nxt = (kk + 1 < K) ? tile + 1 : (sup + nwaves) * K;
        if (nxt < ntiles && (kk + 1 < K || sup + nwaves < nsuper)) LAB_PREFETCH(nxt)
      }
      const bool valid = t < G;
      const int tc = valid ? t : 0;
      const float* colp = wP + off + tc * DD + j;
      const float* qv = wQ + tc * D;
      double acc = 0.0, s1 = 0.0, s2 = 0.0;
      if (WALK) {
#pragma unroll LAB_UNROLL
        for (int i = 0; i < D; ++i) {
          const double p = (double)colp[i * D];
          const double qx = (double)qv[i];
          const double u = p * qx;
          acc += u;
          s1 = fma(u, p, s1);
          s2 = fma(u, u, s2);
        }
      } else {
        acc = (double)colp[0];
        s1 = acc;
        s2 = acc;
      }
      double racc = fma((double)qv[j], s1, -s2);
      const int64_t b = tile * G + tc;
      if (STORE == 1) {
        if (valid) pi_next[b * D + j] = (float)acc;
      } else if (STORE == 3) {  // small per-wave buffer (stays in L2)
        if (valid) pi_next[((int64_t)blockIdx.x * WAVES + wv) * 64 + lane] = (float)acc;
      } else if (STORE == 4) {  // non-temporal
        if (valid) __builtin_nontemporal_store((float)acc, pi_next + b * D + j);
      } else if (STORE == 7) {  // 256-byte aligned store per tile into a cyclic region of region64 x 256 B
        pi_next[(tile % region64) * 64 + lane] = (float)acc;
      } else if (STORE == 8) {  // every second tile stores (half the write bytes)
        if (valid && (tile & 1)) pi_next[b * D + j] = (float)acc;
      } else if (STORE == 9) {  // every fourth tile stores
        if (valid && (tile & 3) == 0) pi_next[b * D + j] = (float)acc;
      } else if (STORE == 6) {  // one full 256-byte line pair per tile (wrong layout; 64 lanes x 4 B aligned)
        pi_next[tile * 64 + lane] = (float)acc;
      } else if (STORE == 2 || STORE == 5) {
        if (valid) sO[wv][kk * G * D + lane] = (float)acc;
      } else {
        keep += (float)acc;
      }
      if (REW) {
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
          const double r = r0 + r1;
          if (STORE == 1 || STORE == 3 || STORE == 4 || STORE == 6) reward[b] = (float)r;
          else if (STORE == 2 || STORE == 5) sR[wv][kk * G + tc] = (float)r;
          else keep += (float)r;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
      } else {
        keep += (float)racc;
      }
    }
    if (STORE == 2 || STORE == 5) {
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      constexpr int N4 = K * G * D / 4;  // K multiple of 4 -> exact
      // STORE 5: every super tile's output padded to a multiple of 128 B (wrong layout, full aligned lines only)
      constexpr int OSTRIDE = STORE == 5 ? ((K * G * D * 4 + 127) / 128) * 32 : K * G * D;
      v4f_t* o4 = reinterpret_cast<v4f_t*>(pi_next + sup * (int64_t)OSTRIDE);
      const v4f_t* s4 = reinterpret_cast<const v4f_t*>(sO[wv]);
      for (int k = lane; k < N4; k += WAVE) store4<FLAV>(o4 + k, s4[k]);
      if (REW) {
        constexpr int R4 = K * G / 4;
        v4f_t* r4 = reinterpret_cast<v4f_t*>(reward + sup * (int64_t)(K * G));
        const v4f_t* sr4 = reinterpret_cast<const v4f_t*>(sR[wv]);
        if (lane < R4) store4<FLAV>(r4 + lane, sr4[lane]);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (keep == 123.456f) pi_next[0] = keep;
#undef LAB_PREFETCH
}

__global__ void k_fill(float* x, int64_t n, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    x[i] = scale * (float)(h >> 8) * (1.0f / 16777216.0f);
  }
}
template <typename F>
static double time_us(F launch, int reps = 20) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int r = 0; r < 10; ++r) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return 1e3 * ms / reps;
}

template <int WALK, int STORE, int K, int REW, int OCC, int FLAV = 0, int WAVES = 4>
static void run(const char* name, int cus, int bpc, const float* pi, const float* P, int64_t B, float* pn, float* rw, int64_t region64 = 1) {
  const int grid = cus * bpc;
  const double us = time_us([&] {
    hipLaunchKernelGGL((k_lab<WALK, STORE, K, REW, OCC, FLAV, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, pi, P, B, pn, rw, region64);
  });
  const double bytes = 1936.0 * (double)B;
  printf("%-58s %2d blocks/CU  %7.1f us  %.2f TB/s  (%.3f of 8)\n", name, bpc, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int64_t B = 983040;
  float *pi, *P, *pn, *rw;
  (void)hipMalloc(&pi, B * 21 * 4);
  (void)hipMalloc(&P, B * 441 * 4 + 64);
  (void)hipMalloc(&pn, B * 32 * 4);
  (void)hipMalloc(&rw, B * 4);
  (void)hipMemset(pi, 0, B * 21 * 4);
  (void)hipMemset(P, 0, B * 441 * 4 + 64);
  if (getenv("LAB_RANDOM")) {
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, P, B * 441, 2.0f / 21.0f);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, pi, B * 21, 2.0f / 21.0f);
    (void)hipDeviceSynchronize();
  }
  printf("%s, %d CUs, B = %lld, d = 21\n", p.gcnArchName, cus, (long long)B);
  for (int rep = 0; rep < 2; ++rep) {
    run<1, 2, 40, 1, 2, 3, 4>("full, K=40 batched sc1, 4-wave blocks", cus, 2, pi, P, B, pn, rw);
    run<1, 0, 40, 1, 2, 3, 4>("K=40 walk + reward, NO stores, 4-wave blocks", cus, 2, pi, P, B, pn, rw);
    run<1, 0, 40, 0, 2, 3, 4>("K=40 walk only, NO reward line, NO stores", cus, 2, pi, P, B, pn, rw);
  }
  return 0;
}
