/*
 * tests/c_abi/c_abi_check.c -- the C-ABI boundary used the way a non-Python caller would:
 * plain C, include/mfa.h, the HIP runtime for device memory, nothing else.  It follows the flow of the
 * reference's validateProblemSize (Tests/FlashAttentionTests/Attention/SquareAttentionTest.swift:214-555):
 * descriptor -> three kernels -> ten buffers -> three dispatches -> compare with the CPU Network
 * (here the oracle, linked as test infrastructure).  Exit code 0 = all six outputs within tolerance.
 *
 *   build: see tests/test_c_abi.py (gcc -std=c11 ... -lmfa_hip -loracle_network -lamdhip64)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfa.h"

/* oracle (test infrastructure) */
void oracle_network_init(unsigned long long seed, int R, int C, int D, float *Q, float *K, float *V, float *dO);
int oracle_network_run(int R, int C, int D, const float *Q, const float *K, const float *V, const float *dO,
                       float *O, float *L, float *Dt, float *dV, float *dK, float *dQ, int num_threads);

#define CHECK_MFA(call)                                                                        \
  do {                                                                                         \
    mfa_status st_ = (call);                                                                   \
    if (st_ != MFA_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, mfa_last_error_string()); return 2; } \
  } while (0)
#define CHECK_HIP(call)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 3; } \
  } while (0)

static double max_abs_diff(const float *a, const float *b, size_t n, double scale_b) {
  double m = 0;
  for (size_t i = 0; i < n; ++i) {
    double d = fabs((double)a[i] - (double)b[i] * scale_b);
    if (d > m || d != d) m = d;
  }
  return m;
}

int main(int argc, char **argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 200, C = argc > 2 ? atoi(argv[2]) : 333, D = argc > 3 ? atoi(argv[3]) : 80;
  const size_t nq = (size_t)R * D, nk = (size_t)C * D;
  float *Q = malloc(nq * 4), *K = malloc(nk * 4), *V = malloc(nk * 4), *dO = malloc(nq * 4);
  float *O = malloc(nq * 4), *L = malloc(R * 4), *Dt = malloc(R * 4), *dV = malloc(nk * 4), *dK = malloc(nk * 4), *dQ = malloc(nq * 4);
  oracle_network_init(42, R, C, D, Q, K, V, dO);
  if (oracle_network_run(R, C, D, Q, K, V, dO, O, L, Dt, dV, dK, dQ, 0) != 0) return 4;

  /* AttentionDescriptor (AttentionDescriptor.swift:10-27), FP32, nothing transposed */
  mfa_attention_descriptor desc;
  mfa_attention_descriptor_init(&desc);
  desc.hasMatrixDimensions = 1; desc.row = R; desc.column = C; desc.head = (uint16_t)D;
  desc.hasTransposeState = 1;

  /* ten device buffers at AttentionOperand.bufferBinding 0-9 */
  const size_t bytes[MFA_BUFFER_SLOTS] = {nq * 4, nk * 4, nk * 4, nq * 4, R * 4, R * 4, nq * 4, nk * 4, nk * 4, nq * 4};
  const float *init[MFA_BUFFER_SLOTS] = {Q, K, V, NULL, NULL, NULL, dO, NULL, NULL, NULL};
  void *buf[MFA_BUFFER_SLOTS];
  for (int s = 0; s < MFA_BUFFER_SLOTS; ++s) {
    CHECK_HIP(hipMalloc(&buf[s], bytes[s]));
    if (init[s]) CHECK_HIP(hipMemcpy(buf[s], init[s], bytes[s], hipMemcpyHostToDevice));
    else CHECK_HIP(hipMemset(buf[s], 0xFF, bytes[s]));   /* NaN poison */
  }
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));

  mfa_launch_params params;
  mfa_launch_params_init(&params);
  params.row = R; params.column = C;
  for (int type = MFA_FORWARD; type <= MFA_BACKWARD_KEY_VALUE; ++type) {   /* forward -> dQ -> dK/dV */
    mfa_attention_kernel_descriptor kd;
    mfa_attention_kernel *kernel;
    CHECK_MFA(mfa_attention_descriptor_kernel_descriptor(&desc, type, &kd));
    CHECK_MFA(mfa_attention_kernel_create(&kd, &kernel));
    uint16_t par, trav, head;
    CHECK_MFA(mfa_attention_kernel_block_dimensions(kernel, &par, &trav, &head));
    printf("type %d: %s  block (%u, %u, %u)  threads %u  LDS %u B\n", type, mfa_attention_kernel_variant(kernel),
           par, trav, head, mfa_attention_kernel_threadgroup_size(kernel),
           mfa_attention_kernel_threadgroup_memory_allocation(kernel));
    CHECK_MFA(mfa_attention_kernel_launch(kernel, buf, &params, (void *)stream));
    mfa_attention_kernel_destroy(kernel);
  }
  CHECK_HIP(hipStreamSynchronize(stream));

  float *gO = malloc(nq * 4), *gL = malloc(R * 4), *gD = malloc(R * 4), *gdV = malloc(nk * 4), *gdK = malloc(nk * 4), *gdQ = malloc(nq * 4);
  CHECK_HIP(hipMemcpy(gO, buf[3], nq * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gL, buf[4], R * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gD, buf[5], R * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gdV, buf[7], nk * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gdK, buf[8], nk * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gdQ, buf[9], nq * 4, hipMemcpyDeviceToHost));

  /* storage scaling undone as the reference's test does (SquareAttentionTest.swift:408-413):
     L is stored in base-2 units, D pre-multiplied by 1/sqrt(D) */
  const double tol = 2e-5;
  const double e[6] = {max_abs_diff(gO, O, nq, 1.0), max_abs_diff(gL, L, R, 1.44269504089),
                       max_abs_diff(gD, Dt, R, 1.0 / sqrt((double)D)), max_abs_diff(gdV, dV, nk, 1.0),
                       max_abs_diff(gdK, dK, nk, 1.0), max_abs_diff(gdQ, dQ, nq, 1.0)};
  const char *names[6] = {"O", "L", "D", "dV", "dK", "dQ"};
  int bad = 0;
  for (int i = 0; i < 6; ++i) {
    printf("%-3s max |err| = %.3e\n", names[i], e[i]);
    if (!(e[i] <= tol * (i == 1 ? 1.44269504089 : 1.0))) bad = 1;
  }
  /* error path: never aborts */
  mfa_attention_descriptor empty;
  mfa_attention_descriptor_init(&empty);
  mfa_attention_kernel_descriptor kd;
  if (mfa_attention_descriptor_kernel_descriptor(&empty, MFA_FORWARD, &kd) != MFA_ERR_INCOMPLETE_DESCRIPTOR) bad = 1;
  printf(bad ? "FAILED\n" : "C ABI OK (%d x %d x %d)\n", R, C, D);
  return bad;
}
