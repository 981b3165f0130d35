// Developer micro-benchmark: does interleaving independent Philox4x32-10 blocks in one wave pay on gfx950?
// NB = blocks evaluated side by side per loop iteration (ILP 2*NB), 4 waves/SIMD like the fused kernel.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
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

constexpr int ITERS = 512;
template <int NB>
__global__ __launch_bounds__(256) void k_philox(uint32_t* out, uint32_t seed) {
  uint32_t acc = 0;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
    u32x4 r[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) r[b] = philox4x32_10(u32x4{tid, (uint32_t)(it * NB + b), acc, 7u}, seed, 99u);
#pragma unroll
    for (int b = 0; b < NB; ++b) acc ^= r[b].x ^ r[b].y ^ r[b].z ^ r[b].w;   // serial only through one xor per iteration
  }
  out[tid] = acc;
}

template <int NB>
static void run(uint32_t* buf, int cus, int wps, double ghz) {
  const int blocks = cus * wps;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_philox<NB>), dim3(blocks), dim3(256), 0, 0, buf, 1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k_philox<NB>), dim3(blocks), dim3(256), 0, 0, buf, 1u);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double blocks_per_simd = (double)wps * ITERS * NB;  // wave-level Philox blocks per SIMD
  printf("NB=%d  %d waves/SIMD  %.3f ms  %.1f cycles per Philox block per SIMD\n", NB, wps, ms, ms * 1e-3 * ghz * 1e9 / blocks_per_simd);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  uint32_t* buf;
  (void)hipMalloc((void**)&buf, (size_t)cus * 8 * 256 * 4);
  for (int wps : {2, 4}) {
    run<1>(buf, cus, wps, ghz);
    run<2>(buf, cus, wps, ghz);
    run<4>(buf, cus, wps, ghz);
  }
  return 0;
}
