// Developer micro-benchmark: does VALU work hide behind v_mfma_f64_16x16x4_f64 on gfx950?  Per loop iteration 3 independent
// matrix instructions and NV independent VALU instructions (fp32 FMA or fp64 FMA), interleaved in program order, at 2 waves
// per SIMD.  Cycles per iteration per SIMD: flat in NV = overlap, rising from NV = 0 = serialised.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int ITERS = 1024;
template <int NV, bool F64, int GROUP = 1>  // GROUP matrix-instruction triples back to back, then their GROUP * NV VALU instructions
__global__ __launch_bounds__(256) void k(double* out, double s) {
  v4d c[3];
  for (int i = 0; i < 3; ++i) c[i] = (v4d)(0.0);
  double a = s + threadIdx.x, b = s * 0.5 + threadIdx.x;
  float f[8];
  double g[8];
  for (int i = 0; i < 8; ++i) { f[i] = (float)threadIdx.x + i; g[i] = (double)threadIdx.x + i; }
  for (int it = 0; it < ITERS / GROUP; ++it) {
    if (GROUP == 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV / 3; ++v) {
          if (F64) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(g[(i * (NV / 3) + v) & 7]) : "v"(b));
          else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[(i * (NV / 3) + v) & 7]) : "v"((float)s));
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < GROUP; ++u)
#pragma unroll
        for (int i = 0; i < 3; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < NV * GROUP; ++v) {
        if (F64) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(g[v & 7]) : "v"(b));
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[v & 7]) : "v"((float)s));
      }
    }
  }
  double t = 0;
  for (int i = 0; i < 3; ++i) t += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  for (int i = 0; i < 8; ++i) t += f[i] + g[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <int NV, bool F64, int GROUP = 1>
void run(double* buf, int cus, double ghz, int wps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)0; hipLaunchKernelGGL((k<NV, F64, GROUP>), dim3(cus * wps), dim3(256), 0, 0, buf, 1.0); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL((k<NV, F64, GROUP>), dim3(cus * wps), dim3(256), 0, 0, buf, 1.0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s VALU per 3 MFMA = %3d  group %2d  waves/SIMD=%d  %.3f ms  %.0f cycles per 3 MFMA per SIMD\n", F64 ? "fp64" : "fp32", NV, GROUP, wps, ms,
         ms * 1e-3 * ghz * 1e9 / ((double)wps * ITERS));
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  double* buf; (void)hipMalloc(&buf, (size_t)p.multiProcessorCount * 8 * 256 * 8);
  double ghz = p.clockRate * 1e-6;
  for (int wps : {1, 2}) {
    run<0, false>(buf, p.multiProcessorCount, ghz, wps); run<12, false>(buf, p.multiProcessorCount, ghz, wps);
    run<24, false>(buf, p.multiProcessorCount, ghz, wps); run<48, false>(buf, p.multiProcessorCount, ghz, wps);
    run<96, false>(buf, p.multiProcessorCount, ghz, wps);
    run<12, true>(buf, p.multiProcessorCount, ghz, wps); run<24, true>(buf, p.multiProcessorCount, ghz, wps);
    run<48, true>(buf, p.multiProcessorCount, ghz, wps);
    run<24, false, 4>(buf, p.multiProcessorCount, ghz, wps); run<24, false, 16>(buf, p.multiProcessorCount, ghz, wps);
    run<24, true, 4>(buf, p.multiProcessorCount, ghz, wps); run<24, true, 16>(buf, p.multiProcessorCount, ghz, wps);
    run<12, false, 16>(buf, p.multiProcessorCount, ghz, wps);
  }
  return 0;
}
