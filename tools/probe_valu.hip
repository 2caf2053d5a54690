// tools/probe_valu.hip -- issue cost of the VALU instructions the softmax segment is made of, on gfx950.
// Developer microbenchmark (not product).  Each test runs a loop of 16 independent instructions per
// iteration in every wave; reports shader cycles (s_memtime) per wave-instruction with one and two waves
// per SIMD, and mixed tests (MFMA stream beside a VALU stream in the same wave / in the SIMD partner).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_valu.hip -o /tmp/probe_valu && /tmp/probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define REP16(X) X X X X X X X X X X X X X X X X
#define REP8(X) X X X X X X X X

template <int OP>
__global__ void probe(long long *cycles, float *sink, int iters) {
  float r[16];
  f32x2 p[16];
  for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{r[i], r[i] + 1}; }
  f32x16 acc = {0};
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.001f * i); fb[i] = (__bf16)(0.002f * i); }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
    } else if constexpr (OP == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 15]), "v"(p[(i + 2) & 15]));
    } else if constexpr (OP == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
    } else if constexpr (OP == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
    } else if constexpr (OP == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
    } else if constexpr (OP == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
    } else if constexpr (OP == 6) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
    } else if constexpr (OP == 7) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 15]));
    } else if constexpr (OP == 8) {   // 16 dependent MFMAs
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
    } else if constexpr (OP == 9) {   // same wave: each MFMA followed by 4 plain VALU
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %0\n\tv_fma_f32 %2, %2, %0, %1\n\tv_fma_f32 %3, %3, %0, %1"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
      }
    } else if constexpr (OP == 10) {  // same wave: each MFMA followed by 7 plain VALU
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %0\n\tv_fma_f32 %2, %2, %0, %1\n\tv_fma_f32 %3, %3, %0, %1\n\t"
                     "v_fma_f32 %4, %4, %1, %2\n\tv_fma_f32 %5, %5, %2, %0\n\tv_fma_f32 %6, %6, %0, %1"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]));
      }
    } else if constexpr (OP == 11) {  // partner split (512 threads): waves 0-3 MFMA only, waves 4-7 64 plain VALU
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
      }
    } else if constexpr (OP == 12) {  // partner split: waves 0-3 MFMA only, waves 4-7 128 plain VALU
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
      }
    } else if constexpr (OP == 13) {  // partner split: MFMA beside 16 v_exp + 48 plain
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 15]), "v"(r[(i + 2) & 15]));
      }
    } else if constexpr (OP == 14) {  // dependent chain of plain VALU (latency)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[0]));
    } else if constexpr (OP == 15) {  // dependent chain of v_exp (latency)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[0]));
    } else if constexpr (OP == 16) {  // v_exp interleaved with plain (does the trans unit overlap plain VALU?)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[8 + i]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
      }
    } else if constexpr (OP == 17) {  // exp followed by its consumer (trans -> VALU dependency)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[8 + i]) : "v"(r[i]));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += r[i] + p[i][0] + p[i][1] + acc[i];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int OP>
static void run(const char *name, int threads, int per_iter) {
  const int blocks = 256, iters = 2000;
  long long *d;
  float *sink;
  hipMalloc(&d, sizeof(long long) * blocks * 8);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d, sink, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(threads), 0, 0, d, sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int nw = blocks * (threads >> 6);
  std::vector<long long> h(nw);
  hipMemcpy(h.data(), d, sizeof(long long) * nw, hipMemcpyDeviceToHost);
  double lo = 0, hi = 0; int nlo = 0, nhi = 0;
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < (threads >> 6); ++w) {
      if (w < 4) { lo += h[b * (threads >> 6) + w]; ++nlo; } else { hi += h[b * (threads >> 6) + w]; ++nhi; }
    }
  printf("%-44s thr %3d  memtime/iter waves0-3 %8.1f  waves4-7 %8.1f   wall %7.3f ms -> %7.1f ns/iter (%d instr/iter)\n", name, threads,
         lo / nlo / iters, nhi ? hi / nhi / iters : 0.0, ms, ms * 1e6 / iters, per_iter);
  hipFree(d); hipFree(sink);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("16 x v_fma_f32", threads, 16);
    run<1>("16 x v_pk_fma_f32", threads, 16);
    run<2>("16 x v_exp_f32", threads, 16);
    run<3>("16 x v_cvt_pk_bf16_f32", threads, 16);
    run<4>("16 x v_max3_f32", threads, 16);
    run<5>("16 x v_pk_add_f32", threads, 16);
    run<6>("16 x v_pk_mul_f32", threads, 16);
    run<7>("16 x v_add_u32", threads, 16);
    run<8>("16 x mfma_32x32x16_bf16 (dependent)", threads, 16);
    run<9>("16 x (mfma + 4 v_fma) same wave", threads, 80);
    run<10>("16 x (mfma + 7 v_fma) same wave", threads, 128);
    run<14>("16 x v_fma dependent chain", threads, 16);
    run<15>("16 x v_exp dependent chain", threads, 16);
    run<16>("8 x (v_exp, v_fma) independent", threads, 16);
    run<17>("8 x (v_exp, dependent v_add)", threads, 16);
  }
  run<11>("partner: 16 mfma | 64 v_fma", 512, 0);
  run<12>("partner: 16 mfma | 128 v_fma", 512, 0);
  run<13>("partner: 16 mfma | 16 v_exp + 48 v_fma", 512, 0);
  return 0;
}
