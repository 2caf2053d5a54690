// tools/probe_issue.hip -- what ONE wave per SIMD can issue beside v_mfma_f32_32x32x16_bf16 on gfx950 (round 6).
// Developer microbenchmark (not product).  Every test is a loop of 16 "gaps": one matrix instruction (four accumulators in rotation, so
// no dependent-accumulator stall) followed by the fillers named in the test; reported: shader clocks (s_memtime) per gap, one wave per
// SIMD (256 threads, one workgroup per compute unit).  The questions: how many plain VALU / SALU / LDS instructions hide in a 32-clock
// gap, what a v_exp_f32 costs depending on what follows it, and what the row sum of four packed P values costs as ONE
// v_mfma_f32_4x4x4_16b_bf16 (A = ones) instead of four v_add_f32.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_issue.hip -o tools/probe_issue.out && tools/probe_issue.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define MF(i) "v_mfma_f32_32x32x16_bf16 a[" #i ":" #i "+15], v[40:43], v[44:47], a[" #i ":" #i "+15]\n\t"
#define VA(r) "v_add_f32 v" #r ", v48, v" #r "\n\t"
#define VX(r) "v_exp_f32 v" #r ", v" #r "\n\t"
#define VC(r) "v_cvt_pk_bf16_f32 v" #r ", v48, v49\n\t"
#define VM(r) "v_max3_f32 v" #r ", v48, v49, v" #r "\n\t"
#define SA(r) "s_add_u32 s" #r ", s" #r ", 1\n\t"
#define M4(r) "v_mfma_f32_4x4x4_16b_bf16 v[" #r ":" #r "+3], v[50:51], v[52:53], v[" #r ":" #r "+3]\n\t"
#define D2(r) "v_dot2_f32_bf16 v" #r ", v48, v49, v" #r "\n\t"
#define DR(r) "ds_read_b128 v[" #r ":" #r "+3], v54\n\t"
#define DT(r) "ds_read_b64_tr_b16 v[" #r ":" #r "+1], v54\n\t"
#define WL "s_waitcnt lgkmcnt(0)\n\t"

// a gap = MF(acc) + fillers; four gaps rotate the accumulators a[0:15], a[16:31], a[32:47], a[48:63]
#define GAP4(F) MF(0) F MF(16) F MF(32) F MF(48) F
#define GAP16(F) GAP4(F) GAP4(F) GAP4(F) GAP4(F)

#define CLOBBERS "memory", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", \
  "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
  "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"

#define TESTS(X) \
  X(0, "mfma only", "") \
  X(1, "4 v_add", VA(60) VA(61) VA(62) VA(63)) \
  X(2, "5 v_add", VA(60) VA(61) VA(62) VA(63) VA(64)) \
  X(3, "6 v_add", VA(60) VA(61) VA(62) VA(63) VA(64) VA(65)) \
  X(4, "7 v_add", VA(60) VA(61) VA(62) VA(63) VA(64) VA(65) VA(66)) \
  X(5, "8 v_add", VA(60) VA(61) VA(62) VA(63) VA(64) VA(65) VA(66) VA(67)) \
  X(6, "1 mfma4x4x4", M4(68)) \
  X(7, "2 mfma4x4x4 (two accumulators)", M4(68) M4(72)) \
  X(8, "4 v_add + 1 mfma4x4x4", VA(60) VA(61) VA(62) VA(63) M4(68)) \
  X(9, "5 v_add + 1 mfma4x4x4", VA(60) VA(61) VA(62) VA(63) VA(64) M4(68)) \
  X(10, "3 v_add + 1 mfma4x4x4", VA(60) VA(61) VA(62) M4(68)) \
  X(11, "5 v_add + 2 s_add", VA(60) VA(61) VA(62) VA(63) VA(64) SA(40) SA(41)) \
  X(12, "4 v_add + 2 s_add", VA(60) VA(61) VA(62) VA(63) SA(40) SA(41)) \
  X(13, "3 v_add + 3 s_add (alternating)", VA(60) SA(40) VA(61) SA(41) VA(62) SA(42)) \
  X(14, "7 s_add", SA(40) SA(41) SA(42) SA(43) SA(44) SA(45) SA(46)) \
  X(15, "5 s_add", SA(40) SA(41) SA(42) SA(43) SA(44)) \
  X(16, "2 v_exp + 3 v_add (exp first, back to back)", VX(60) VX(61) VA(62) VA(63) VA(64)) \
  X(17, "2 v_exp + 3 v_add (each exp followed by v_add)", VX(60) VA(62) VX(61) VA(63) VA(64)) \
  X(18, "2 v_exp + 2 v_add + 2 s_add (each exp followed by s_add)", VX(60) SA(40) VX(61) SA(41) VA(62) VA(63)) \
  X(19, "2 v_exp + 2 v_add + 2 s_add (s_add last)", VX(60) VX(61) VA(62) VA(63) SA(40) SA(41)) \
  X(20, "2 v_exp + 1 v_add (exp last: the next mfma follows an exp)", VA(62) VX(60) VX(61)) \
  X(21, "1 v_exp + 4 v_add", VX(60) VA(61) VA(62) VA(63) VA(64)) \
  X(22, "4 v_add + 1 ds_read_b128", VA(60) VA(61) VA(62) VA(63) DR(72)) \
  X(23, "4 v_add + 1 ds_read_b64_tr_b16", VA(60) VA(61) VA(62) VA(63) DT(72)) \
  X(24, "3 v_add + 2 ds_read_b64_tr_b16", VA(60) VA(61) VA(62) DT(72) DT(74)) \
  X(25, "4 v_add + 1 v_dot2_f32_bf16", VA(60) VA(61) VA(62) VA(63) D2(68)) \
  X(26, "2 v_add + 2 v_cvt_pk + 1 v_max3", VA(60) VA(61) VC(62) VC(63) VM(64)) \
  X(27, "softmax mix of one tile gap: exp exp add add cvt max3 + tr read", VX(60) VA(62) VX(61) VA(63) VC(64) VM(65) DT(72)) \
  X(28, "same, row sum as mfma4x4x4 every 2nd gap equivalent: exp exp cvt max3 + tr read + mfma4x4x4", VX(60) VX(61) VC(64) VM(65) DT(72) M4(68)) \
  X(29, "exp exp cvt max3 + tr read (no row sum)", VX(60) VX(61) VC(64) VM(65) DT(72)) \
  X(30, "6 v_add + 1 s_add", VA(60) VA(61) VA(62) VA(63) VA(64) VA(65) SA(40)) \
  X(31, "3 v_add + 1 s_waitcnt lgkmcnt(0) + 1 ds_read_b128", VA(60) VA(61) VA(62) WL DR(72))

template <int T>
__global__ __launch_bounds__(256) void probe(long long *cycles, int iters) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = 0.f;
  __syncthreads();
  asm volatile("v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
               "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0x3f803f80\n\tv_mov_b32 v51, 0x3f803f80\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\t"
               "v_lshlrev_b32 v54, 4, %0\n\t"
               "s_mov_b32 s40, 0\n\ts_mov_b32 s41, 0\n\ts_mov_b32 s42, 0\n\ts_mov_b32 s43, 0\n\ts_mov_b32 s44, 0\n\ts_mov_b32 s45, 0\n\ts_mov_b32 s46, 0\n\t"
               : : "v"((int)(threadIdx.x & 63)) : CLOBBERS);
  for (int r = 60; r < 80; ++r) asm volatile("" ::: "memory");
  asm volatile("v_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\tv_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\t"
               "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\tv_mov_b32 v72, 0\n\tv_mov_b32 v73, 0\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\t" ::: CLOBBERS);
  for (int i = 0; i < 64; i += 1) asm volatile("" ::: "memory");
  asm volatile(
#define Z(i) "v_accvgpr_write_b32 a" #i ", 0\n\t"
      Z(0) Z(1) Z(2) Z(3) Z(4) Z(5) Z(6) Z(7) Z(8) Z(9) Z(10) Z(11) Z(12) Z(13) Z(14) Z(15) Z(16) Z(17) Z(18) Z(19) Z(20) Z(21) Z(22) Z(23) Z(24) Z(25) Z(26) Z(27) Z(28) Z(29) Z(30) Z(31)
      Z(32) Z(33) Z(34) Z(35) Z(36) Z(37) Z(38) Z(39) Z(40) Z(41) Z(42) Z(43) Z(44) Z(45) Z(46) Z(47) Z(48) Z(49) Z(50) Z(51) Z(52) Z(53) Z(54) Z(55) Z(56) Z(57) Z(58) Z(59) Z(60) Z(61) Z(62) Z(63)
#undef Z
      ::: CLOBBERS);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define RUN(id, name, F) if constexpr (T == id) asm volatile(GAP16(F) "s_waitcnt lgkmcnt(0)\n\t" ::: CLOBBERS);
    TESTS(RUN)
#undef RUN
  }
  const long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  long long *d;
  hipMalloc(&d, sizeof(long long) * cus * 4);
  const int iters = 2000;
  std::vector<long long> h(cus * 4);
  printf("# tools/probe_issue.hip: shader clocks per gap (one v_mfma_f32_32x32x16_bf16 + the fillers), one wave per SIMD, %d workgroups, all-zero operands\n", cus);
#define LAUNCH(id, name, F)                                                                              \
  {                                                                                                      \
    for (int rep = 0; rep < 2; ++rep) {                                                                  \
      hipLaunchKernelGGL(probe<id>, dim3(cus), dim3(256), 0, 0, d, iters);                              \
      hipDeviceSynchronize();                                                                            \
    }                                                                                                    \
    hipMemcpy(h.data(), d, sizeof(long long) * cus * 4, hipMemcpyDeviceToHost);                          \
    double s = 0; long long mn = h[0], mx = h[0];                                                        \
    for (auto v : h) { s += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }                             \
    printf("%-2d %-88s %7.2f clk/gap  (min %7.2f max %7.2f)\n", id, name, s / h.size() / iters / 16.0, mn / (iters * 16.0), mx / (iters * 16.0)); \
  }
  TESTS(LAUNCH)
  return 0;
}
