// Developer micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (cycles per instruction per SIMD) with 1..4
// independent accumulators and 1..4 waves per SIMD.  hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_rate.hip -o tools/micro/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2048;
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, double s) {
  v4d c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = (v4d)(0.0);
  double a = s + threadIdx.x, b = s * 0.5 + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  double t = 0;
  for (int i = 0; i < NACC; ++i) t += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
template <int NACC>
void run(double* buf, int cus, double ghz, int wps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(cus * wps), dim3(256), 0, 0, buf, 1.0); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(cus * wps), dim3(256), 0, 0, buf, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)wps * ITERS * NACC;  // MFMAs per SIMD
  printf("acc=%d waves/SIMD=%d  %.3f ms  %.1f cycles per MFMA per SIMD  (%.1f TFLOP/s fp64)\n", NACC, wps, ms, ms * 1e-3 * ghz * 1e9 / n,
         (double)cus * 4 * n * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  double* buf; hipMalloc(&buf, (size_t)p.multiProcessorCount * 8 * 256 * 8);
  double ghz = p.clockRate * 1e-6;
  for (int wps : {1, 2, 4}) { run<1>(buf, p.multiProcessorCount, ghz, wps); run<3>(buf, p.multiProcessorCount, ghz, wps); run<4>(buf, p.multiProcessorCount, ghz, wps); }
  return 0;
}
