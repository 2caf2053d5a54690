// tools/probe_f32_mfma.hip -- what does a v_mfma_f32_32x32x2_f32 cost beside the instructions the FP32 attention kernels put
// around it?  One wave per SIMD (256 threads, 160 KiB of LDS per workgroup), 256 workgroups x ITER groups of 8 matrix
// instructions; s_memtime around the loop.  Developer microbenchmark:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Imetal_flash_attention_amd/csrc tools/probe_f32_mfma.hip -o /tmp/probe_f32 && /tmp/probe_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "attn_f32.h"

using namespace mfa;
using namespace mfa::f32k;

// (the compiler-scheduled forms the first version of the kernels used)
template <int N> __device__ __forceinline__ void lds_wait(f32x4 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void lds_wait(f32x4 &a, f32x4 &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
template <int DP> __device__ __forceinline__ uint32_t first_address(int i, int hi) { return i * Geo<DP>::ROWB + ((hi ^ (i & 15)) << 4); }
template <int DP> __device__ __forceinline__ uint32_t second_address(int i, int hi) {
  const int byte = i * Geo<DP>::RB2;
  return 4 * hi * Geo<DP>::ROWB + ((((byte >> 4) ^ (4 * hi)) << 4) | (byte & 15));
}

template <bool EXP> __device__ __forceinline__ void filler(float &x, float fa, float fb) {
  if constexpr (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(fa), "v"(fb));
}

// MODE 0: two accumulators alternate, operands fixed registers
//      1: + the two ds_read_b128 and the counted wait of first_product_pair per group of 8
//      2: one accumulator (first_product of the forward kernel), reads as in the kernel
//      3: four accumulators rotate (second_product), one read per 4
//      4: MODE 1 with the reads issued in the MIDDLE of the group instead of its end
//      5: MODE 0 with two v_mov per group of 8 (VALU beside MFMA)
//      6: three accumulators rotate, one read per 3
//      7: four accumulators in the pattern of MODE 1 (s_a, dp_a, s_b, dp_b: 2 reads per 8)
//      8: four accumulators rotate, per 4: one read + NV VALU instructions (fma / exp mix: the softmax arithmetic as filler)
//      9: two accumulators alternate, per 4: one read + NV VALU
template <int MODE, int NV = 0> __global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cycles, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, q = lane & 31, hi = lane >> 5;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float *>(smem)[i] = 1.0f / (1 + (i & 7));
  __syncthreads();
  const uint32_t base0 = lds_addr(smem) + first_address<128>(q, hi), base1 = base0 + 16384;
  const uint32_t sbase = lds_addr(smem) + second_address<128>(q, hi);
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float b[8];
  for (int i = 0; i < 8; ++i) b[i] = 0.5f + i + lane * 1e-3f;
  f32x4 a0 = {1.f, 2.f, 3.f, 4.f}, a1 = {0.5f, 0.25f, 0.125f, 1.f};
  float fa = 1.0001f, fb = 1e-6f;
  float bq[8];
  f32x16 oacc[4];
  if constexpr (MODE == 10) {
    for (int i = 0; i < 8; ++i) bq[i] = b[i];
    pin_fragments<8>(bq, 1.f);
  }
  if constexpr (MODE == 11)
    for (int k = 0; k < 4; ++k) mfma_zero(oacc[k]);
  f32x4 r0[3], r1[3];
  for (int i = 0; i < 3; ++i) { r0[i] = a0; r1[i] = a1; }
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if constexpr (MODE == 1 || MODE == 4) {
    r0[0] = rd128<0>(base0); r1[0] = rd128<0>(base1);
    r0[1] = rd128<0>(base0 ^ 32); r1[1] = rd128<0>(base1 ^ 32);
    r0[2] = rd128<0>(base0 ^ 64); r1[2] = rd128<0>(base1 ^ 64);
  }
  if constexpr (MODE == 10) {
    r0[0] = rd128<0>(base0); r1[0] = rd128<16384>(base0);
    r0[1] = rd128<0>(base0); r1[1] = rd128<16384>(base0);
    r0[2] = rd128<0>(base0); r1[2] = rd128<16384>(base0);
  }
  if constexpr (MODE == 2 || MODE == 3 || MODE == 11) { r0[0] = rd128<0>(base0); r0[1] = rd128<0>(base0 ^ 32); r0[2] = rd128<0>(base0 ^ 64); }
  for (int it = 0; it < iters; ++it) {
    static_for<3>([&](auto S_) {
      constexpr int S = decltype(S_)::value;
      if constexpr (MODE == 0 || MODE == 5) {
        if constexpr (MODE == 5) {
          asm volatile("v_mov_b32 %0, %1" : "=v"(a0[0]) : "v"(b[S]));
          asm volatile("v_mov_b32 %0, %1" : "=v"(a1[0]) : "v"(b[S + 1]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[0] = mfma(a0[i], b[i], acc[0]); acc[1] = mfma(a1[i], b[4 + i], acc[1]); }
      } else if constexpr (MODE == 1) {
        lds_wait<4>(r0[S], r1[S]);
        const f32x4 v0 = r0[S], v1 = r1[S];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[0] = mfma(v0[i], b[i], acc[0]); acc[1] = mfma(v1[i], b[4 + i], acc[1]); }
        r0[S] = rd128<0>(base0 ^ (((it * 3 + S) & 15) << 5));
        r1[S] = rd128<0>(base1 ^ (((it * 3 + S) & 15) << 5));
      } else if constexpr (MODE == 4) {
        lds_wait<4>(r0[S], r1[S]);
        const f32x4 v0 = r0[S], v1 = r1[S];
        acc[0] = mfma(v0[0], b[0], acc[0]); acc[1] = mfma(v1[0], b[4], acc[1]);
        acc[0] = mfma(v0[1], b[1], acc[0]); acc[1] = mfma(v1[1], b[5], acc[1]);
        f32x4 n0 = rd128<0>(base0 ^ (((it * 3 + S) & 15) << 5));
        acc[0] = mfma(v0[2], b[2], acc[0]); acc[1] = mfma(v1[2], b[6], acc[1]);
        f32x4 n1 = rd128<0>(base1 ^ (((it * 3 + S) & 15) << 5));
        acc[0] = mfma(v0[3], b[3], acc[0]); acc[1] = mfma(v1[3], b[7], acc[1]);
        r0[S] = n0; r1[S] = n1;
      } else if constexpr (MODE == 2) {
        lds_wait<2>(r0[S]);
        const f32x4 v0 = r0[S];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0] = mfma(v0[i], b[i], acc[0]);
        r0[S] = rd128<0>(base0 ^ (((it * 3 + S) & 15) << 5));
        lds_wait<2>(r0[(S + 1) % 3]);
        const f32x4 v1 = r0[(S + 1) % 3];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0] = mfma(v1[i], b[4 + i], acc[0]);
        r0[(S + 1) % 3] = rd128<0>(base0 ^ (((it * 3 + S + 7) & 15) << 5));
      } else if constexpr (MODE == 6) {
        lds_wait<2>(r0[S]);
        const f32x4 v0 = r0[S];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = mfma(v0[i], b[S], acc[i]);
        r0[S] = rd128<0>(base0 ^ (((it * 3 + S) & 15) << 5));
        lds_wait<2>(r0[(S + 1) % 3]);
        const f32x4 v1 = r0[(S + 1) % 3];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = mfma(v1[i], b[S + 4], acc[i]);
        r0[(S + 1) % 3] = rd128<0>(base0 ^ (((it * 3 + S + 5) & 15) << 5));
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = mfma(v1[3], b[S + i], acc[i]);   // (8 per group like the others)
      } else if constexpr (MODE == 7) {
        lds_wait<4>(r0[S], r1[S]);
        const f32x4 v0 = r0[S], v1 = r1[S];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[2 * (i & 1)] = mfma(v0[i], b[i], acc[2 * (i & 1)]); acc[2 * (i & 1) + 1] = mfma(v1[i], b[4 + i], acc[2 * (i & 1) + 1]); }
        r0[S] = rd128<0>(base0 ^ (((it * 3 + S) & 15) << 5));
        r1[S] = rd128<0>(base1 ^ (((it * 3 + S) & 15) << 5));
      } else if constexpr (MODE == 8 || MODE == 9) {
        constexpr int NA = MODE == 8 ? 4 : 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          lds_wait<2>(r0[(S + h) % 3]);
          const f32x4 v0 = r0[(S + h) % 3];
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i % NA] = mfma(v0[i], b[S + 4 * h], acc[i % NA]);
          r0[(S + h) % 3] = rd128<0>(base0 ^ (((it * 3 + S + 5 * h) & 15) << 5));
          static_for<NV>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            filler<i % 6 == 4>(b[(i + 1) & 7], fa, fb);
          });
        }
      } else if constexpr (MODE == 10) {   // the kernels' own statement: two score accumulators in VGPRs, B operands in AGPRs
        const f32x4 v0 = r0[S], v1 = r1[S];
        mfma_group_pair<4>(acc[0], v0, bq, acc[1], v1, bq + 4);
        r0[S] = rd128<0>(base0);
        r1[S] = rd128<16384>(base0);
      } else if constexpr (MODE == 11) {   // ... four output accumulators in AGPRs, A and B in VGPRs
        const f32x4 v0 = r0[S];
        mfma_out<2>(oacc, v0, b[S]);
        r0[S] = rd128<0>(base0);
        const f32x4 v1 = r0[(S + 1) % 3];
        mfma_out<2>(oacc, v1, b[S + 4]);
        r0[(S + 1) % 3] = rd128<512>(base0);
      } else {
        lds_wait<2>(r0[S]);
        const f32x4 v0 = r0[S];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mfma(v0[i], b[S], acc[i]);
        r0[S] = rd128<second_row(S) * 512>(sbase ^ (second_ct(S) << 4));
        lds_wait<2>(r0[(S + 1) % 3]);
        const f32x4 v1 = r0[(S + 1) % 3];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = mfma(v1[i], b[S + 4], acc[i]);
        r0[(S + 1) % 3] = rd128<second_row(S + 4) * 512>(sbase ^ (second_ct(S + 4) << 4));
      }
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]));
  float sum = 0.f;
  if constexpr (MODE == 10) mfma_fence(acc[0], acc[1]);
  if constexpr (MODE == 11) {
    mfma_fence_out<4>(oacc);
    for (int k = 0; k < 4; ++k) acc[k] = oacc[k];
  }
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) sum += acc[k][r];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = sum + r0[0][0] + r1[0][0];
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE, int NV = 0> void run(const char *name, float *out, unsigned long long *cyc, int iters) {
  hipFuncSetAttribute((const void *)probe<MODE, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, NV>), dim3(256), dim3(256), 160 * 1024, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : h) mean += c;
  mean /= 256;
  printf("%-78s %8.1f clocks per matrix instruction\n", name, mean / (iters * 24.0));
}

int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
  const int iters = 2000;
  run<0>("two accumulators alternate, fixed operands", out, cyc, iters);
  run<5>("the same, two v_mov per group of 8", out, cyc, iters);
  run<1>("two accumulators, 2 ds_read_b128 + wait per group of 8 (first_product_pair)", out, cyc, iters);
  run<4>("the same, reads in the middle of the group", out, cyc, iters);
  run<2>("ONE accumulator, 1 ds_read_b128 + wait per 4 (first_product, forward)", out, cyc, iters);
  run<3>("four accumulators rotate, 1 ds_read_b128 + wait per 4 (second_product)", out, cyc, iters);
  run<10>("the kernels' first-product statement (asm, scores in VGPRs, B in AGPRs), immediates", out, cyc, iters);
  run<11>("the kernels' second-product statement (asm, outputs in AGPRs), immediates", out, cyc, iters);
  run<6>("three accumulators rotate, 1 read per 3", out, cyc, iters);
  run<7>("four accumulators (s_a, dp_a, s_b, dp_b), 2 reads + wait per group of 8", out, cyc, iters);
  run<8, 6>("four accumulators rotate, per 4: 1 read + 6 VALU (5 fma, 1 exp)", out, cyc, iters);
  run<8, 12>("four accumulators rotate, per 4: 1 read + 12 VALU", out, cyc, iters);
  run<8, 24>("four accumulators rotate, per 4: 1 read + 24 VALU", out, cyc, iters);
  run<9, 6>("two accumulators alternate, per 4: 1 read + 6 VALU", out, cyc, iters);
  run<9, 12>("two accumulators alternate, per 4: 1 read + 12 VALU", out, cyc, iters);
  return 0;
}
