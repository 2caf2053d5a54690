// tools/probe_mfma_peak.hip -- what the matrix pipe sustains on THIS chip for v_mfma_f32_32x32x16_bf16:
// back-to-back MFMAs on register operands (no memory traffic at all), zeros versus N(0,1) operands, for
// launches long enough (tens of ms) for the power management to settle.  The random-operand figure is the
// practical ceiling an attention kernel fed with random Q/K/V can approach.  Developer microbenchmark.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_peak.hip -o tools/probe_mfma_peak.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ inline float hash_normal(unsigned x) {   // cheap N(0,1)-ish value per (lane, index)
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  const float u1 = ((x & 0xFFFF) + 1) / 65537.0f, u2 = (x >> 16) / 65536.0f;
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

__global__ __launch_bounds__(512) void mfma_loop(float *sink, int iters, int random) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (__bf16)(random ? hash_normal(threadIdx.x * 64 + i * 8 + j + blockIdx.x * 977) : 0.f);
      b[i][j] = (__bf16)(random ? hash_normal(threadIdx.x * 64 + i * 8 + j + 32 + blockIdx.x * 131) : 0.f);
    }
  f32x16 acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + k) & 3], b[k], acc[i], 0, 0, 0);
    if (random && (it & 63) == 63) {   // keep the accumulators finite
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] *= 1e-3f;
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 1234.5f) sink[0] = s;
}

int main() {
  float *sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int random = 0; random < 2; ++random)
    for (int iters : {20000, 200000, 600000}) {
      const int blocks = 256;
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, sink, 100, random);
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, sink, iters, random);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * 8.0 * blocks;
      printf("%s operands  %7d iters x 16 MFMA x 8 waves x %d WGs: %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2.5 PF; implied clock %.2f GHz)\n",
             random ? "N(0,1)" : "zero  ", iters, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0,
             flops / ms / 1e9 / 2500.0 * 2.4);
    }
  return 0;
}
