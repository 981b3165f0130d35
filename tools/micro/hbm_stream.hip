// Developer micro-benchmark: hand-written streaming-read ceilings on this box (context for roofline.frac).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_stream.hip -o tools/micro/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <initializer_list>

typedef float v4f __attribute__((ext_vector_type(4)));

// grid-stride 16-byte loads, U independent loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ x, int64_t n4, float* out) {
  v4f acc = (v4f)(0.0f);
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = base + (int64_t)u * 256;
      v[u] = (k < n4) ? (NT ? __builtin_nontemporal_load(x + k) : x[k]) : (v4f)(0.0f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[0] = s;  // keep the loads alive
}

// the given-P pattern: persistent blocks, contiguous TILE-byte tiles, register prefetch, LDS staging + barrier
template <int PER, bool STAGE>
__global__ __launch_bounds__(256) void k_tiles(const v4f* __restrict__ x, int64_t ntiles, int tile4, float* out) {
  extern __shared__ v4f lds[];
  v4f pre[PER], acc = (v4f)(0.0f);
  const int tid = threadIdx.x;
  auto fetch = [&](int64_t t) {
    const v4f* s = x + t * tile4;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int k = tid + u * 256;
      pre[u] = (k < tile4) ? __builtin_nontemporal_load(s + k) : (v4f)(0.0f);
    }
  };
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (STAGE) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int k = tid + u * 256;
        if (k < tile4) lds[k] = pre[u];
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int u = 0; u < PER; ++u) acc += pre[u];
    }
    if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
    if (STAGE) {
      acc += lds[(tid * 7) % tile4];
      __syncthreads();
    }
  }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[0] = s;
}


// same tile pattern, but the loads write straight into LDS (global_load_lds_dwordx4, no VGPR hop, no ds_write):
// double-buffered tiles, one wave instruction moves 64 x 16 B = 1 KiB to M0-based LDS addresses
__global__ __launch_bounds__(256) void k_tiles_dma(const v4f* __restrict__ x, int64_t ntiles, int tile4, float* out) {
  extern __shared__ v4f lds[];  // [2][tile4 rounded up to 64]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int CH = 6;                    // chunks per wave per tile: always issued, so vmcnt(CH) is exact
  const int pitch = 4 * CH * 64;           // float4 per buffer (24 KiB); lanes past the tile load a dummy address
  v4f acc = (v4f)(0.0f);
  auto fetch = [&](int64_t t, int buf) {
    const v4f* s = x + t * tile4;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int c = wv + 4 * u;
      const int k = c * 64 + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + (k < tile4 ? k : 0)),
                                       (__attribute__((address_space(3))) void*)(lds + buf * pitch + c * 64), 16, 0, 0);
    }
  };
  int buf = 0;
  if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x, 0);
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (t + gridDim.x < ntiles) fetch(t + gridDim.x, buf ^ 1);
    // wait for the CURRENT tile only: the prefetch just issued may stay in flight
    if (t + gridDim.x < ntiles) __builtin_amdgcn_s_waitcnt(0x0f70 | 6);  // vmcnt(6): this wave's 6 newest loads pending
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    acc += lds[buf * pitch + (tid * 7) % tile4];
    __syncthreads();
    buf ^= 1;
  }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[0] = s;
}

template <typename F>
static double time_ms(F launch, int reps = 10) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const size_t bytes = 1734082560ull / 16 * 16;  // the bench's 1.73 GB action slab
  void *x, *out;
  (void)hipMalloc(&x, bytes);
  (void)hipMalloc(&out, 256);
  (void)hipMemset(x, 0, bytes);
  const int64_t n4 = bytes / 16;
  printf("%s, %d CUs, slab %.2f GB\n", p.gcnArchName, cus, bytes / 1e9);
  for (int bpc : {4, 8, 16, 32}) {
    const int grid = cus * bpc;
    double t;
    t = time_ms([&] { hipLaunchKernelGGL((k_read<4, false>), dim3(grid), dim3(256), 0, 0, (const v4f*)x, n4, (float*)out); });
    printf("grid-stride U=4      %2d blocks/CU  %.2f TB/s\n", bpc, bytes / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((k_read<4, true>), dim3(grid), dim3(256), 0, 0, (const v4f*)x, n4, (float*)out); });
    printf("grid-stride U=4 nt   %2d blocks/CU  %.2f TB/s\n", bpc, bytes / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((k_read<8, true>), dim3(grid), dim3(256), 0, 0, (const v4f*)x, n4, (float*)out); });
    printf("grid-stride U=8 nt   %2d blocks/CU  %.2f TB/s\n", bpc, bytes / t / 1e9);
  }
  const int tile4 = 12 * 441 / 4 * 4 / 4;  // 12 trajectories x 441 floats = 1323 float4
  const int64_t ntiles = n4 / tile4;
  for (int bpc : {4, 6, 7, 8}) {
    const int grid = cus * bpc;
    double t;
    t = time_ms([&] { hipLaunchKernelGGL((k_tiles<6, false>), dim3(grid), dim3(256), 0, 0, (const v4f*)x, ntiles, tile4, (float*)out); });
    printf("tiles 21 KB, regs    %2d blocks/CU  %.2f TB/s\n", bpc, (double)ntiles * tile4 * 16 / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((k_tiles<6, true>), dim3(grid), dim3(256), tile4 * 16, 0, (const v4f*)x, ntiles, tile4, (float*)out); });
    printf("tiles 21 KB, LDS+bar %2d blocks/CU  %.2f TB/s\n", bpc, (double)ntiles * tile4 * 16 / t / 1e9);
  }
  for (int bpc : {2, 3}) {
    const int grid = cus * bpc;
    const int pitch = 4 * 6 * 64;
    const double t = time_ms([&] { hipLaunchKernelGGL(k_tiles_dma, dim3(grid), dim3(256), 2 * pitch * 16, 0, (const v4f*)x, ntiles, tile4, (float*)out); });
    printf("tiles 21 KB, LDS DMA x2 %2d blocks/CU  %.2f TB/s\n", bpc, (double)ntiles * tile4 * 16 / t / 1e9);
  }
  return 0;
}
