// tools/probe_dispatch.hip -- how does the hardware deal workgroups of UNEQUAL duration to compute units?  (one workgroup per
// compute unit at a time: 160 KiB of LDS each, like the hand-placed kernels.)  Workgroup b spins for dur(b) microseconds and
// records (XCC id, SE / CU id, start, end) in wall-clock ticks; the host prints per-XCD makespans for a few block orders of a
// causal launch (durations = fixed + tiles traversed).  Developer microbenchmark:
//   hipcc --offload-arch=gfx950 -O2 tools/probe_dispatch.hip -o tools/probe_dispatch.out && tools/probe_dispatch.out
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Rec { unsigned xcc, hwid; unsigned long long t0, t1; };

__global__ __launch_bounds__(256) void spin(const unsigned *dur_ticks, Rec *out) {
  extern __shared__ char smem[];
  const unsigned long long t0 = wall_clock64();
  const unsigned long long want = dur_ticks[blockIdx.x];
  if (threadIdx.x == 0) smem[0] = 1;
  while (wall_clock64() - t0 < want) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    out[blockIdx.x] = Rec{xcc & 15u, hwid, t0, (unsigned long long)wall_clock64()};
  }
}

int main() {
  const int heads = 64, nb = 16, n = heads * nb;
  const double fixed_us = 25.0, tile_us = 1.375, ticks_per_us = 100.0;   // wall_clock64: 100 MHz
  auto dur = [&](int b) { return fixed_us + tile_us * 4 * (b + 1); };
  unsigned *d_dur; Rec *d_out;
  hipMalloc(&d_dur, n * sizeof(unsigned)); hipMalloc(&d_out, n * sizeof(Rec));
  hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const char *names[] = {"one head after the other (long first)", "pairs of heads interleaved", "row-block-major over the 8 heads of an XCD",
                         "alternating direction per head"};
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<unsigned> h(n);
    std::vector<int> blk(n);
    for (int bid = 0; bid < n; ++bid) {
      const int slot = bid >> 3;                    // per-XCD sequence (workgroup b runs on XCD b % 8)
      int r;
      if (mode == 0) r = nb - 1 - slot % nb;
      else if (mode == 1) r = nb - 1 - (slot % (2 * nb)) / 2;
      else if (mode == 2) r = nb - 1 - slot / (heads / 8);
      else r = ((slot / nb) & 1) ? slot % nb : nb - 1 - slot % nb;
      blk[bid] = r;
      h[bid] = (unsigned)(dur(r) * ticks_per_us);
    }
    hipMemcpy(d_dur, h.data(), n * sizeof(unsigned), hipMemcpyHostToDevice);
    std::vector<Rec> rec(n);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin, dim3(n), dim3(256), 160 * 1024, 0, d_dur, d_out);
      hipDeviceSynchronize();
      hipMemcpy(rec.data(), d_out, n * sizeof(Rec), hipMemcpyDeviceToHost);
      unsigned long long lo = ~0ull, hi = 0;
      for (auto &x : rec) { lo = std::min(lo, x.t0); hi = std::max(hi, x.t1); }
      best = std::min(best, (hi - lo) / ticks_per_us);
    }
    // did workgroup b land on XCD b % 8?  how many distinct (xcc, hw id) places, and jobs per place (min .. max)
    int on_xcd = 0;
    std::vector<std::pair<unsigned long long, int>> places;
    for (int b = 0; b < n; ++b) on_xcd += (int)(rec[b].xcc == (unsigned)(b & 7));
    std::vector<unsigned long long> key(n);
    for (int b = 0; b < n; ++b) key[b] = ((unsigned long long)rec[b].xcc << 32) | (rec[b].hwid & 0x00F0FF00u);   // SE / SA / CU bits
    std::vector<unsigned long long> uniq(key); std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    int mn = 1 << 30, mx = 0;
    for (auto u : uniq) { const int c = (int)std::count(key.begin(), key.end(), u); mn = std::min(mn, c); mx = std::max(mx, c); }
    double sum = 0; for (int b = 0; b < n; ++b) sum += dur(blk[b]);
    printf("%-46s makespan %7.1f us   ideal %6.1f us   on XCD b%%8: %d / %d   places %zu, workgroups per place %d .. %d\n",
           names[mode], best, sum / 256.0, on_xcd, n, uniq.size(), mn, mx);
  }
  return 0;
}
