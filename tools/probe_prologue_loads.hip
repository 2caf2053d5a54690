// probe_prologue_loads.hip -- what bounds the fragment loads in front of the hand-placed backward statements?  (round 5; standalone:
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_prologue_loads.out tools/probe_prologue_loads.hip && tools/probe_prologue_loads.out)
//
// One workgroup of four waves per compute unit (160 KiB of LDS requested, as the kernels), every wave fetches the B-operand
// fragments of its 64 rows of a [rows][256] 16-bit operand exactly as attn_dq16_p5's prologue does (32 x buffer_load_dwordx4, lane =
// row, 16 bytes from element 16 s + 8 hi), stamps the 100 MHz wall clock before the first load, after the last one is ISSUED and
// when all data has arrived, waits `gap` microseconds (the traversal) and goes to its next unit (rows never touched before).
// Variants: how many workgroups take part, whether their starts are spread, whether the rows were touched a moment ago.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNITS = 6, ROWB = 512, WGROWS = 256;

// mode bits: 1 = P-role-like waves 2, 3 (two more operands, two batches), 2 = second pass over the same rows (warm), 4 = whole rows
// per half-wave instead of one row per lane
__global__ __launch_bounds__(256) void probe(const char *q, const char *g, const char *o, uint32_t rows, uint32_t active_mod,
                                              uint32_t spread_ticks, uint32_t gap_ticks, uint32_t mode, uint32_t *out) {
  extern __shared__ char smem[];
  if (blockIdx.x % active_mod != 0) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, ql = lane & 31, hi = lane >> 5;
  const uint64_t t_start = wall_clock64();
  if (spread_ticks) {   // starts spread over spread_ticks (a hash of the workgroup index)
    const uint64_t until = t_start + (blockIdx.x * 2654435761u >> 8) % spread_ticks;
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(4);
  }
  const uint32_t bytes = rows * ROWB;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(q), 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(g), 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(o), 0, bytes, 0x00020000);
  uint32_t sink = 0;
  for (int u = 0; u < UNITS; ++u) {
    const uint32_t r0 = (uint32_t)(u * gridDim.x + blockIdx.x) * WGROWS + wave * 64;
    for (int pass = 0; pass < ((mode & 2) ? 2 : 1); ++pass) {
      const uint64_t t0 = wall_clock64();
      u32x4 x[32];
      const __amdgpu_buffer_rsrc_t res = (wave >= 2 && (mode & 1)) ? gres : qres;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int b = i >> 4, s = i & 15;
        const uint32_t off = (mode & 4) ? (r0 + b * 32 + 2 * s + hi) * ROWB + ql * 16 : (r0 + b * 32 + ql) * ROWB + (16 * s + 8 * hi) * 2;
        x[i] = __builtin_amdgcn_raw_buffer_load_b128(res, off, 0, 0);
      }
      asm volatile("" ::: "memory");
      const uint64_t t1 = wall_clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint64_t t2 = wall_clock64();
#pragma unroll
      for (int i = 0; i < 32; ++i) sink ^= x[i][0] ^ x[i][3];
      uint64_t t3 = t2;
      if (wave >= 2 && (mode & 1)) {   // the O rows of the same wave: a second batch
        u32x4 y[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const int b = i >> 4, s = i & 15;
          y[i] = __builtin_amdgcn_raw_buffer_load_b128(ores, (r0 + b * 32 + ql) * ROWB + (16 * s + 8 * hi) * 2, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t3 = wall_clock64();
#pragma unroll
        for (int i = 0; i < 32; ++i) sink ^= y[i][1];
      }
      if (lane == 0) {
        uint32_t *dst = out + (((size_t)blockIdx.x * UNITS + u) * 2 + pass) * 4 * 4 + wave * 4;
        dst[0] = (uint32_t)(t1 - t0); dst[1] = (uint32_t)(t2 - t0); dst[2] = (uint32_t)(t3 - t0); dst[3] = (uint32_t)(t0 - t_start);
      }
    }
    __syncthreads();
    const uint64_t until = wall_clock64() + gap_ticks;
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
  }
  if (sink == 0x12345678u) out[0] = sink;
}

int main() {
  const uint32_t groups = 256, rows = groups * UNITS * WGROWS;
  const size_t bytes = (size_t)rows * ROWB;
  char *q, *g, *o;
  uint32_t *out;
  CHECK(hipMalloc(&q, bytes)); CHECK(hipMalloc(&g, bytes)); CHECK(hipMalloc(&o, bytes));
  const size_t out_words = (size_t)groups * UNITS * 2 * 16;
  CHECK(hipMalloc(&out, out_words * 4));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // something bigger than every cache between the variants: the rows must be cold again
  char *flush;
  const size_t flush_bytes = (size_t)1 << 30;
  CHECK(hipMalloc(&flush, flush_bytes));
  struct V { const char *name; uint32_t active_mod, spread_us, gap_us, mode; };
  const V variants[] = {
      {"all 256 workgroups in lockstep, S-role loads only", 1, 0, 100, 0},
      {"all 256, waves 2 / 3 as the P-role (dO then O)", 1, 0, 100, 1},
      {"1 workgroup of 256 (nothing else on the GPU)", 256, 0, 100, 0},
      {"1 of 256, P-role waves", 256, 0, 100, 1},
      {"32 of 256", 8, 0, 100, 0},
      {"all 256, starts spread over 100 us", 1, 100, 100, 0},
      {"all 256, starts spread over 100 us, P-role waves", 1, 100, 100, 1},
      {"all 256 in lockstep, whole rows per half-wave", 1, 0, 100, 4},
      {"all 256 in lockstep, every unit loaded twice (second pass = warm)", 1, 0, 100, 2},
  };
  printf("# 32 x buffer_load_dwordx4 per wave (32 KiB), 4 waves per workgroup, one workgroup per compute unit, %d units per workgroup, microseconds\n", UNITS);
  for (const V &v : variants) {
    CHECK(hipMemset(flush, 1, flush_bytes));
    CHECK(hipMemset(out, 0, out_words * 4));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(probe, dim3(groups), dim3(256), 160 * 1024, 0, q, g, o, rows, v.active_mod, v.spread_us * 100, v.gap_us * 100, v.mode, out);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> h(out_words);
    CHECK(hipMemcpy(h.data(), out, out_words * 4, hipMemcpyDeviceToHost));
    for (int pass = 0; pass < ((v.mode & 2) ? 2 : 1); ++pass) {
      for (int role = 0; role < ((v.mode & 1) ? 2 : 1); ++role) {
        std::vector<double> iss, arr, arr2;
        for (uint32_t w = 0; w < groups; w += v.active_mod)
          for (int u = 1; u < UNITS; ++u)   // (unit 0: the launch itself)
            for (int wave = 2 * role; wave < 2 * role + 2; ++wave) {
              const uint32_t *d = &h[(((size_t)w * UNITS + u) * 2 + pass) * 16 + wave * 4];
              iss.push_back(d[0] / 100.0); arr.push_back(d[1] / 100.0); arr2.push_back(d[2] / 100.0);
            }
        auto stat = [](std::vector<double> &x) { std::sort(x.begin(), x.end()); double s = 0; for (double e : x) s += e; return s / x.size(); };
        const double mi = stat(iss), ma = stat(arr), m2 = stat(arr2);
        printf("%-68s %s%s last load issued %6.2f  all data %6.2f (median %6.2f, max %6.2f)", v.name, pass ? "pass 2 " : "", role ? "waves 2,3" : "waves 0,1",
               mi, ma, arr[arr.size() / 2], arr.back());
        if (role) printf("  second batch done %6.2f", m2);
        printf("\n");
      }
    }
  }
  return 0;
}
