// Developer micro-benchmark: issue cost (cycles per wave instruction per SIMD) of the VALU instructions the fused
// rollout kernel leans on.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#ifndef MFG_CHAINS
#define MFG_CHAINS 8
#endif
constexpr int ITERS = 4096, CHAINS = MFG_CHAINS;

#define OP_KERNEL(NAME, TYPE, INIT, BODY)                                            \
  __global__ __launch_bounds__(256) void NAME(TYPE* out, TYPE seed) {                \
    TYPE v[CHAINS];                                                                  \
    _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) v[c] = INIT;                  \
    for (int it = 0; it < ITERS; ++it) {                                             \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { TYPE x = v[c]; BODY; v[c] = x; } \
    }                                                                                \
    TYPE s = v[0];                                                                   \
    _Pragma("unroll") for (int c = 1; c < CHAINS; ++c) s += v[c];                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                  \
  }

OP_KERNEL(k_fma_f32, float, seed + (float)(threadIdx.x + c), x = fmaf(x, 1.0001f, 0.5f))
OP_KERNEL(k_fma_f64, double, seed + (double)(threadIdx.x + c), x = fma(x, 1.0001, 0.5))
OP_KERNEL(k_add_f64, double, seed + (double)(threadIdx.x + c), x = x + 1.5)
OP_KERNEL(k_mul_f64, double, seed + (double)(threadIdx.x + c), x = x * 1.0000001)
OP_KERNEL(k_mul_lo_u32, uint32_t, seed + threadIdx.x + c, x = x * 0xD2511F53u)
OP_KERNEL(k_mul_hi_u32, uint32_t, seed + threadIdx.x + c, x = __umulhi(x, 0xD2511F53u) + 1u)
OP_KERNEL(k_mad_u64_u32, uint32_t, seed + threadIdx.x + c,
          { uint64_t p = (uint64_t)x * 0xD2511F53u; x = (uint32_t)(p >> 32) ^ (uint32_t)p; })
OP_KERNEL(k_xor_u32, uint32_t, seed + threadIdx.x + c, x = (x ^ 0x9E3779B9u) + 1u)
OP_KERNEL(k_exp2_f32, float, seed + 0.001f * (float)(threadIdx.x + c), x = __builtin_amdgcn_exp2f(x) - 1.0f)
OP_KERNEL(k_log2_f32, float, seed + 2.0f + (float)(threadIdx.x + c), x = __builtin_amdgcn_logf(x) + 3.0f)
OP_KERNEL(k_rcp_f32, float, seed + 2.0f + (float)(threadIdx.x + c), x = __builtin_amdgcn_rcpf(x) + 1.0f)
OP_KERNEL(k_rsq_f32, float, seed + 2.0f + (float)(threadIdx.x + c), x = __builtin_amdgcn_rsqf(x) + 1.0f)
OP_KERNEL(k_sqrt_f32, float, seed + 2.0f + (float)(threadIdx.x + c), x = __builtin_amdgcn_sqrtf(x) + 1.0f)
OP_KERNEL(k_sin_f32, float, seed + 0.01f * (float)(threadIdx.x + c), x = __builtin_amdgcn_sinf(x) + 0.1f)
OP_KERNEL(k_cvt_f64_f32, double, seed + (double)(threadIdx.x + c), x = (double)((float)x) + 1.0)
OP_KERNEL(k_rcp_f64, double, seed + 2.0 + (double)(threadIdx.x + c), x = __builtin_amdgcn_rcp(x) + 1.0)

typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk_fma_f32(float* out, float seed) {
  v2f v[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) v[c] = v2f{seed + (float)(threadIdx.x + c), seed + 1.0f};
  const v2f m{1.0001f, 0.9999f}, a{0.5f, 0.25f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = __builtin_elementwise_fma(v[c], m, a);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += v[c].x + v[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T, typename K>
static void run(const char* name, K kern, int ops_per_body, void* buf, T seed, double ghz, int cus, int wps = 8) {
  const int blocks = cus * wps;  // wps blocks x 4 waves per CU = wps waves per SIMD resident
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (T*)buf, seed);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (T*)buf, seed);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instrs_per_simd = (double)blocks * 4 / (cus * 4) * ITERS * CHAINS;  // per measured body
  const double cycles = ms * 1e-3 * ghz * 1e9;
  printf("%-18s %d waves/SIMD %8.3f ms   %6.2f cycles per body (body = %d target op%s + glue)\n", name, wps, ms,
         cycles / wave_instrs_per_simd, ops_per_body, ops_per_body > 1 ? "s" : "");
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("%s: %d CUs, %.2f GHz (nominal)\n", p.gcnArchName, cus, ghz);
  void* buf;
  hipMalloc(&buf, (size_t)cus * 8 * 256 * 8);
  run<float>("v_fma_f32", k_fma_f32, 1, buf, 1.0f, ghz, cus);
  run<float>("v_pk_fma_f32", k_pk_fma_f32, 1, buf, 1.0f, ghz, cus);
  run<double>("v_fma_f64", k_fma_f64, 1, buf, 1.0, ghz, cus);
  run<double>("v_add_f64", k_add_f64, 1, buf, 1.0, ghz, cus);
  run<double>("v_mul_f64", k_mul_f64, 1, buf, 1.0, ghz, cus);
  run<uint32_t>("v_xor+add", k_xor_u32, 2, buf, 1u, ghz, cus);
  run<uint32_t>("v_mul_lo_u32", k_mul_lo_u32, 1, buf, 1u, ghz, cus);
  run<uint32_t>("v_mul_hi_u32+add", k_mul_hi_u32, 2, buf, 1u, ghz, cus);
  run<uint32_t>("v_mad_u64_u32+xor", k_mad_u64_u32, 2, buf, 1u, ghz, cus);
  run<float>("v_exp_f32+sub", k_exp2_f32, 2, buf, 0.5f, ghz, cus);
  run<float>("v_log_f32+add", k_log2_f32, 2, buf, 0.5f, ghz, cus);
  run<float>("v_rcp_f32+add", k_rcp_f32, 2, buf, 0.5f, ghz, cus);
  run<float>("v_rsq_f32+add", k_rsq_f32, 2, buf, 0.5f, ghz, cus);
  run<float>("v_sqrt_f32+add", k_sqrt_f32, 2, buf, 0.5f, ghz, cus);
  run<float>("v_sin_f32+add", k_sin_f32, 2, buf, 0.5f, ghz, cus);
  run<double>("cvt f64<->f32+add", k_cvt_f64_f32, 3, buf, 1.0, ghz, cus);
  run<double>("v_rcp_f64+add", k_rcp_f64, 2, buf, 1.0, ghz, cus);
  for (int wps : {1, 2, 3, 4, 6}) {
    run<float>("v_fma_f32", k_fma_f32, 1, buf, 1.0f, ghz, cus, wps);
    run<uint32_t>("v_xor+add", k_xor_u32, 2, buf, 1u, ghz, cus, wps);
    run<double>("v_fma_f64", k_fma_f64, 1, buf, 1.0, ghz, cus, wps);
    run<uint32_t>("v_mad_u64_u32+xor", k_mad_u64_u32, 2, buf, 1u, ghz, cus, wps);
    run<float>("v_log_f32+add", k_log2_f32, 2, buf, 0.5f, ghz, cus, wps);
  }
  return 0;
}
