// tools/probe_soffset.hip -- buffer addressing facts the LDS-DMA fillers depend on (gfx950, round 6; developer probe, not product):
//  (1) is the SGPR offset of a raw buffer load part of the range check (voffset + soffset against num_records), and what happens
//      when voffset is the kernels' out-of-range marker 0xFFFFFF00 and soffset is not zero (32-bit wrap?);
//  (2) for `buffer_load_dword ... offset:N lds`, does the instruction offset move the LDS destination, the global source, or both.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_soffset.hip -o tools/probe_soffset.out && tools/probe_soffset.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void probe(const uint32_t *src, uint32_t nrec, uint32_t *out) {
  __shared__ uint32_t lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = 0xAAAA0000u + i;
  __syncthreads();
  u32x4 res;
  res[0] = (uint32_t)(uintptr_t)src;
  res[1] = (uint32_t)((uintptr_t)src >> 32) & 0xFFFFu;
  res[2] = nrec;
  res[3] = 0x00020000u;
  uint32_t r[6];
  const uint32_t v0 = lane * 4, vbig = 0xFFFFFF00u;
  uint32_t s;
  // (1) plain loads to a register
  s = 0;    asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(r[0]) : "v"(v0), "s"(res), "s"(s) : "memory");
  s = 128;  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(r[1]) : "v"(v0), "s"(res), "s"(s) : "memory");
  s = 1024; asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(r[2]) : "v"(v0), "s"(res), "s"(s) : "memory");
  s = 512;  asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(r[3]) : "v"(vbig), "s"(res), "s"(s) : "memory");
  s = 0;    asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(r[4]) : "v"(vbig), "s"(res), "s"(s) : "memory");
  s = 0;    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:128\n\ts_waitcnt vmcnt(0)" : "=v"(r[5]) : "v"(v0), "s"(res), "s"(s) : "memory");
  for (int i = 0; i < 6; ++i) out[i * 64 + lane] = r[i];
  // (2) LDS-DMA with an instruction offset: M0 = byte address of lds[256]
  const uint32_t m0 = (uint32_t)(uintptr_t)(&lds[256]);   // (LDS aperture: low 32 bits are the LDS byte address)
  s = 0;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 2\n\tbuffer_load_dword %0, %1, %3 offen offset:64 lds\n\ts_waitcnt vmcnt(0)" : : "v"(v0), "s"(res), "s"(m0), "s"(s) : "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[6 * 64 + i] = lds[i];
}

int main() {
  const int n = 4096;
  std::vector<uint32_t> h(n);
  for (int i = 0; i < n; ++i) h[i] = i;   // word i holds i
  uint32_t *src, *out;
  hipMalloc(&src, n * 4);
  hipMalloc(&out, (6 * 64 + 1024) * 4);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  const uint32_t nrec = 256 * 4;   // 256 words in range
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, nrec, out);
  hipDeviceSynchronize();
  std::vector<uint32_t> o(6 * 64 + 1024);
  hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
  const char *names[6] = {"voffset = 4 lane, soffset 0", "voffset = 4 lane, soffset 128 (in range)", "voffset = 4 lane, soffset 1024 (= num_records)",
                          "voffset = 0xFFFFFF00, soffset 512 (wraps into range?)", "voffset = 0xFFFFFF00, soffset 0", "voffset = 4 lane, offset:128"};
  printf("# tools/probe_soffset.hip: raw buffer of %u bytes (word i holds i; beyond: in the allocation but out of num_records)\n", nrec);
  for (int t = 0; t < 6; ++t) printf("%-58s lane 0 -> %u, lane 1 -> %u, lane 32 -> %u, lane 63 -> %u\n", names[t], o[t * 64], o[t * 64 + 1], o[t * 64 + 32], o[t * 64 + 63]);
  printf("LDS-DMA buffer_load_dword offen offset:64 lds, M0 -> lds[256], voffset = 4 lane: LDS words that changed:\n");
  int shown = 0;
  for (int i = 0; i < 1024; ++i)
    if (o[6 * 64 + i] != 0xAAAA0000u + i && shown < 6) { printf("  lds[%d] = %u\n", i, o[6 * 64 + i]); ++shown; }
  int cnt = 0; for (int i = 0; i < 1024; ++i) cnt += o[6 * 64 + i] != 0xAAAA0000u + i;
  printf("  (%d words changed)\n", cnt);
  return 0;
}
