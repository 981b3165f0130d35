/*
 * Plain-C consumer of the C ABI (include/mfg_hip.h): no Python, no torch, no C++.
 * One forward-RL update at the reference's problem size, the way a host program written in any language with a C
 * FFI would drive the library: allocate device buffers with the HIP runtime, zero the workspace once, enqueue
 * mfg_gather_start -> mfg_rollout (fused TD rollout + batch gradients) -> mfg_apply_update on a stream, then the
 * native episode loop mfg_train_rollouts (device-side start draws, schedule in C); read
 * theta and the mean reward back.  tests/test_c_consumer.py compares the printed numbers with the Python binding.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/abi_consumer.c \
 *       -Ldiscrete_mean_field_game_amd/csrc -lmfg_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,... -o abi_consumer
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "mfg_hip.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define MFG_OK_(x)                                                                 \
  do {                                                                             \
    int r_ = (x);                                                                  \
    if (r_ != MFG_OK) {                                                            \
      fprintf(stderr, "%s:%d mfg error %d: %s\n", __FILE__, __LINE__, r_, mfg_last_error()); \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char** argv) {
  const int d = 21, T = 15, num_start = 8;
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
  const int64_t F = mfg_num_features(d);
  int cus = 0;
  char arch[64];
  MFG_OK_(mfg_device_info(&cus, arch, (int)sizeof arch));
  MFG_OK_(mfg_init());
  /* this program's library state (the status word of its launches) lives in a context of its own, bound to this thread */
  mfg_ctx_t* ctx = NULL;
  MFG_OK_(mfg_ctx_create(&ctx));
  MFG_OK_(mfg_ctx_bind(ctx));

  /* deterministic inputs that the Python side can rebuild: start states, start indices, critic weights */
  float* mat_h = (float*)malloc(sizeof(float) * num_start * d);
  int32_t* idx_h = (int32_t*)malloc(sizeof(int32_t) * B);
  double* w_h = (double*)malloc(sizeof(double) * F);
  for (int s = 0; s < num_start; ++s) {
    double sum = 0.0;
    for (int j = 0; j < d; ++j) sum += (double)((s * 31 + j * 17) % 97 + 1);
    for (int j = 0; j < d; ++j) mat_h[s * d + j] = (float)((double)((s * 31 + j * 17) % 97 + 1) / sum);
  }
  for (int64_t b = 0; b < B; ++b) idx_h[b] = (int32_t)((b * 7 + 3) % num_start);
  for (int64_t k = 0; k < F; ++k) w_h[k] = (double)((k * 13) % 101) / 101.0;
  double theta_h = 8.86349;

  float *mat, *pi0, *pi_traj, *pi_last, *reward;
  int32_t* idx;
  double *theta, *w, *delta, *g, *G, *racc;
  void* ws;
  const size_t ws_bytes = mfg_workspace_bytes(B * T, d);
  HIP_OK(hipMalloc((void**)&mat, sizeof(float) * num_start * d));
  HIP_OK(hipMalloc((void**)&idx, sizeof(int32_t) * B));
  HIP_OK(hipMalloc((void**)&pi0, sizeof(float) * B * d));
  HIP_OK(hipMalloc((void**)&pi_traj, sizeof(float) * B * (T + 1) * d));
  HIP_OK(hipMalloc((void**)&pi_last, sizeof(float) * B * d));
  HIP_OK(hipMalloc((void**)&reward, sizeof(float) * B * T));
  HIP_OK(hipMalloc((void**)&delta, sizeof(double) * B * T));
  HIP_OK(hipMalloc((void**)&g, sizeof(double) * B * T));
  HIP_OK(hipMalloc((void**)&theta, sizeof(double)));
  HIP_OK(hipMalloc((void**)&w, sizeof(double) * F));
  HIP_OK(hipMalloc((void**)&G, sizeof(double) * (F + 3)));
  HIP_OK(hipMalloc((void**)&racc, sizeof(double)));
  HIP_OK(hipMalloc(&ws, ws_bytes));
  HIP_OK(hipMemset(ws, 0, ws_bytes)); /* once: the control block must start out zero (mfg_hip.h) */
  HIP_OK(hipMemset(racc, 0, sizeof(double)));
  HIP_OK(hipMemcpy(mat, mat_h, sizeof(float) * num_start * d, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(idx, idx_h, sizeof(int32_t) * B, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(w, w_h, sizeof(double) * F, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(theta, &theta_h, sizeof(double), hipMemcpyHostToDevice));

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  for (int episode = 0; episode < 2; ++episode) { /* two updates, lr/(episode+1) like mfg_ac2.py:514 */
    MFG_OK_(mfg_gather_start(mat, num_start, idx, B, d, pi0, st));
    MFG_OK_(mfg_rollout(pi0, B, d, T, theta, 0.16, 12000.0, w, 1.0, MFG_REWARD_MFG_AC2, /*seed*/ 42u,
                        /*first_step*/ (uint32_t)(episode * T), /*traj_offset*/ 0u, MFG_ROLLOUT_TD, pi_traj, pi_last,
                        reward, delta, g, NULL, G, 0, ws, ws_bytes, st));
    MFG_OK_(mfg_apply_update(G, d, 0.1 / (episode + 1), 0.001 / (episode + 1), w, theta, racc, st));
  }
  HIP_OK(hipStreamSynchronize(st));
  { /* numeric sanitiser of the boundary: a launch that met theta outside the mixed-precision range would have reported it */
    unsigned status_bits = 1u;
    MFG_OK_(mfg_ctx_status(ctx, &status_bits));
    if (status_bits != 0u || mfg_ctx_current() != ctx) return 3;
  }

  double racc_h = 0.0, count = 0.0;
  HIP_OK(hipMemcpy(&theta_h, theta, sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&racc_h, racc, sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(w_h, w, sizeof(double) * F, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&count, G + F + 2, sizeof(double), hipMemcpyDeviceToHost));
  double wsum = 0.0;
  for (int64_t k = 0; k < F; ++k) wsum += w_h[k];

  /* The episode loop itself (mfg_ac2.py:460-526) as ONE native call: three more episodes, start states drawn on the
   * device (mfg_draw_start semantics: Philox keyed by seed / the episode's first step / trajectory id), learning rates
   * lr / (e + 1), lr / ((e + 1) ln ln (e + 20)) for e = 2, 3, 4, one return accumulator per episode. */
  double* racc3;
  HIP_OK(hipMalloc((void**)&racc3, 3 * sizeof(double)));
  HIP_OK(hipMemset(racc3, 0, 3 * sizeof(double)));
  MFG_OK_(mfg_train_rollouts(mat, num_start, B, d, T, /*episodes*/ 3, /*first_episode*/ 2, /*constant*/ 0, theta, 0.16, 12000.0, w,
                             1.0, MFG_REWARD_MFG_AC2, /*seed*/ 42u, /*first_step*/ (uint32_t)(2 * T), /*traj_offset*/ 0u,
                             /*flags*/ 0, 0.1, 0.001, pi_traj, pi_last, reward, delta, g, G, racc3, ws, ws_bytes, st));
  HIP_OK(hipStreamSynchronize(st));
  double theta2 = 0.0, racc3_h[3];
  HIP_OK(hipMemcpy(&theta2, theta, sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(racc3_h, racc3, 3 * sizeof(double), hipMemcpyDeviceToHost));

  printf("{\"abi\": %d, \"arch\": \"%s\", \"cus\": %d, \"B\": %lld, \"theta\": %.17g, \"w_sum\": %.17g, "
         "\"mean_reward_acc\": %.17g, \"count\": %.0f, \"theta_after_native_loop\": %.17g, "
         "\"native_loop_rewards\": [%.17g, %.17g, %.17g]}\n",
         mfg_abi_version(), arch, cus, (long long)B, theta_h, wsum, racc_h, count, theta2, racc3_h[0], racc3_h[1], racc3_h[2]);
  MFG_OK_(mfg_ctx_destroy(ctx)); /* (also unbinds it from this thread) */
  if (mfg_ctx_current() != NULL) return 3;
  return 0;
}
