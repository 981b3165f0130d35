// Developer check: accuracy of  log1p(e) = ln2 * v_log_f32(u) + (e - (u - 1)) * v_rcp_f32(u),  u = fl32(1 + e),
// against the fp64 log1p over e in [2^-24, 4] (hardware log near 1 + first-order correction of the rounding of u).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* e, float* out, float* out_poly, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = e[i];
  const float u = 1.0f + x;
  const float r = __builtin_amdgcn_rcpf(u);
  const float d = x - (u - 1.0f);
  out[i] = fmaf(d, r, __builtin_amdgcn_logf(u) * 0.69314718055994531f);
  out_poly[i] = __builtin_amdgcn_logf(u) * 0.69314718055994531f;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = ldexpf(1.0f, -24) * expf(logf(4.0f * 16777216.0f) * (float)i / (float)n);
  float *de, *d1, *d2;
  hipMalloc(&de, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
  hipMemcpy(de, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, de, d1, d2, n);
  std::vector<float> o1(n), o2(n);
  hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(o2.data(), d2, n * 4, hipMemcpyDeviceToHost);
  double worst = 0, worst_e = 0, worst_plain = 0;
  for (int i = 0; i < n; ++i) {
    const double ref = log1p((double)h[i]);
    const double r1 = fabs((double)o1[i] - ref) / ref, r2 = fabs((double)o2[i] - ref) / ref;
    if (r1 > worst) { worst = r1; worst_e = h[i]; }
    if (h[i] < 0.25f && r2 > worst_plain) worst_plain = r2;
  }
  printf("corrected: max rel err %.3e at e = %.6g;  plain log(1+e) for e < 1/4: max rel err %.3e\n", worst, worst_e, worst_plain);
  return 0;
}
