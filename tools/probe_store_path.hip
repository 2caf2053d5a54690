// probe_store_path.hip -- how long does a workgroup's burst of 128 KiB of stores take (what the epilogue of attn_fwd16_p4p issues per
// 256-row block: 4 waves x 32 KiB of fp32 O), as a function of HOW MANY compute units burst at the same time and of the store pattern?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_store_path.out tools/probe_store_path.hip && tools/probe_store_path.out
// Patterns (per wave and instruction, 64 lanes x 16 bytes): 0 = the product epilogue's (eight lanes cover one 128-byte line of a row, eight
// rows 512 bytes apart), 1 = one contiguous KiB, 2 = dword stores, two contiguous 128-byte lines (the OROW streams').
// Reported: shader clocks (s_memtime, 100 MHz ticks -> ns) from the first store to s_waitcnt vmcnt(0), mean over the workgroups, and the
// bytes per ns of ONE workgroup.  Round 6, DESIGN.md 4.1 "the block switch".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PATTERN>
__global__ __launch_bounds__(256) void burst(float *out, unsigned long long *ticks, int rounds) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *base = out + (size_t)blockIdx.x * (256 * 128) ;          // this workgroup's 256 rows x 128 floats
  unsigned long long total = 0;
  for (int r = 0; r < rounds; ++r) {
    __builtin_amdgcn_s_sleep(100);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (PATTERN == 0) {
      for (int i = 0; i < 32; ++i) {          // 32-row x 32-column block i of the wave's 64 rows: lanes = (row & 7, 16-byte chunk), four row groups
        const int rb = i >> 2 & 1, db = i & 3, g = i >> 3;
        const int row = wave * 64 + rb * 32 + g * 8 + (lane >> 3), col = db * 32 + (lane & 7) * 4;
        *reinterpret_cast<float4 *>(base + row * 128 + col) = make_float4(r, i, lane, wave);
      }
    } else if (PATTERN == 1) {
      for (int i = 0; i < 32; ++i)
        *reinterpret_cast<float4 *>(base + wave * 8192 + i * 256 + lane * 4) = make_float4(r, i, lane, wave);
    } else {
      for (int i = 0; i < 128; ++i) {
        const int row = wave * 64 + (i >> 2) * 2 + (lane >> 5), col = (i & 3) * 32 + (lane & 31);
        base[row * 128 + col] = (float)(r + i);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    total += t1 - t0;
  }
  if (threadIdx.x == 0) ticks[blockIdx.x] = total;
}

int main() {
  float *out; unsigned long long *ticks;
  const int maxwg = 256, rounds = 50;
  hipMalloc(&out, (size_t)maxwg * 256 * 128 * 4);
  hipMalloc(&ticks, maxwg * 8);
  std::vector<unsigned long long> h(maxwg);
  const char *names[3] = {"product epilogue (8 rows x 128 B per instruction)", "contiguous KiB per instruction", "dword stores, 2 x 128 B per instruction"};
  for (int p = 0; p < 3; ++p)
    for (int nwg : {1, 8, 32, 64, 128, 256}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (p == 0) hipLaunchKernelGGL(burst<0>, dim3(nwg), dim3(256), 0, 0, out, ticks, rounds);
        if (p == 1) hipLaunchKernelGGL(burst<1>, dim3(nwg), dim3(256), 0, 0, out, ticks, rounds);
        if (p == 2) hipLaunchKernelGGL(burst<2>, dim3(nwg), dim3(256), 0, 0, out, ticks, rounds);
        hipDeviceSynchronize();
      }
      hipMemcpy(h.data(), ticks, nwg * 8, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < nwg; ++i) s += (double)h[i];
      const double ns = s / nwg / rounds * 10.0;   // s_memrealtime: 100 MHz
      printf("%-52s %3d workgroups: %8.0f ns per 128 KiB burst  (%6.1f B / ns per workgroup, %7.1f GB/s together)\n", names[p], nwg, ns, 131072.0 / ns, 131072.0 / ns * nwg);
    }
  return 0;
}
