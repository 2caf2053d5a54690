// gemm_kernels.h -- GEMM operator kernels for gfx950.
//
//   C[m][n] = sum_k A(m, k) B(k, n) (+ previous C)      reference: Sources/FlashAttention/GEMM/GEMMKernel/
//   (GEMMKernel+Source.swift:9-84 kernel body, GEMMKernel+Multiply.swift:113-212 the K loop,
//    GEMMKernel+Caching.swift accumulator load/store with the optional previous C).
//
// Two code objects, both 256 work-items = 2 x 2 waves, each wave owning a 64 x 64 block of C as 2 x 2 MFMA
// tiles (the reference's "splits" (2, 2), GEMMDescriptor.swift:207-211, re-derived for 64-wide waves and
// 32 x 32 matrix-core tiles instead of 8 x 8 simdgroup matrices):
//
// * gemm_f32mfma: every storage-precision mix, transpose state, leading dimension, alignment and ragged
//   edge.  Operands are converted to fp32 while staged (element-wise, bounds-checked; BF16 by bit
//   placement, GEMMHeaders.swift:402-409) and multiplied with v_mfma_f32_32x32x2_f32, which is exact fp32
//   FMA arithmetic -- what the reference's FP32 register precision asks for.  Roofline: 157 TFLOP/s.
// * gemm_16: A and B in the same 16-bit type (any leading dimension, any K).  16-byte chunks are copied to LDS
//   unchanged (bounds-checked buffer loads, zeros outside the matrix) into one of the two images the
//   attention kernels use -- k-contiguous rows for ds_read_b128 when the operand's memory is k-major,
//   [x/32][k][32] for ds_read_b64_tr_b16 when it is not -- and multiplied with v_mfma_f32_32x32x16.
//   Roofline: 2.5 PFLOP/s spec, 1.56 PFLOP/s sustained on random operands.
//
// Both accumulate in fp32 whatever the C storage type (the reference accumulates in FP16 when A, B and C
// are all FP16, GEMMDescriptor.swift:188-193; fp32 is at least as accurate), then add the previous C if
// asked and store with the reference's rounding: FP16 round-to-nearest, BF16 truncation.
#pragma once
#include <type_traits>
#include "attn_fwd16_common.h"

namespace mfa {

struct GemmArgs {
  const void *A, *B;
  void *C;
  uint32_t M, N, K, ldA, ldB, ldC;
  int32_t precA, precB, precC;
  int32_t transA, transB, loadC;
  uint64_t bsA, bsB, bsC;   // batch strides, elements
};

// epilogue shared by both kernels: acc[mb][nb] holds C rows (bm + 32 mb + crow(r, hi)), column bn + 32 nb + (lane & 31)
template <int MT, int NT>
__device__ __forceinline__ void gemm_store(const GemmArgs &g, char *C, const f32x16 (&acc)[MT][NT], int row0, int col0, int lane) {
  const int i = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
      const uint32_t col = col0 + 32 * nb + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t row = row0 + 32 * mb + crow(r, hi);
        if (row < g.M && col < g.N) {
          const int64_t idx = (int64_t)row * g.ldC + col;
          float v = acc[mb][nb][r];
          if (g.loadC) v += load_elem(C, idx, g.precC);
          store_elem(C, idx, g.precC, v);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// general kernel: fp32 arithmetic, element-wise staging
// ------------------------------------------------------------------------------------------------
constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_F32_BK = 16, GEMM_F32_BKP = GEMM_F32_BK + 4;

__global__ __launch_bounds__(256) void gemm_f32mfma(const GemmArgs g) {
  constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = GEMM_F32_BK, BKP = GEMM_F32_BKP, PER = BM * BK / 256;
  __shared__ __attribute__((aligned(16))) float As[2][BM * BKP];   // [m][k], k contiguous, pitch 20 floats (80 B:
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * BKP];   //  16 rows of a ds_read_b128 group hit 16 distinct 16-byte slots)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;
  const uint32_t bm = blockIdx.y * BM, bn = blockIdx.x * BN;
  const char *A = (const char *)g.A + (uint64_t)blockIdx.z * g.bsA * elem_size(g.precA);
  const char *B = (const char *)g.B + (uint64_t)blockIdx.z * g.bsB * elem_size(g.precB);
  char *C = (char *)g.C + (uint64_t)blockIdx.z * g.bsC * elem_size(g.precC);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // per-thread staging slots: the index that is contiguous in MEMORY runs fastest over the threads
  int64_t aoff[PER], boff[PER];
  int alds[PER], blds[PER], ak[PER], bk[PER];
  bool aok[PER], bok[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int idx = tid + 256 * e;
    const int m = g.transA ? idx % BM : idx / BK, ka = g.transA ? idx / BM : idx % BK;
    const int n = g.transB ? idx / BK : idx % BN, kb = g.transB ? idx % BK : idx / BN;
    aok[e] = bm + m < g.M;
    bok[e] = bn + n < g.N;
    ak[e] = ka;
    bk[e] = kb;
    aoff[e] = g.transA ? (int64_t)ka * g.ldA + (bm + m) : (int64_t)(bm + m) * g.ldA + ka;
    boff[e] = g.transB ? (int64_t)(bn + n) * g.ldB + kb : (int64_t)kb * g.ldB + (bn + n);
    alds[e] = m * BKP + ka;
    blds[e] = n * BKP + kb;
  }
  const int64_t astep = g.transA ? (int64_t)BK * g.ldA : BK, bstep = g.transB ? BK : (int64_t)BK * g.ldB;
  float ra[PER], rb[PER];
  auto gload = [&](uint32_t k0) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      ra[e] = (aok[e] && k0 + ak[e] < g.K) ? load_elem(A, aoff[e], g.precA) : 0.f;
      rb[e] = (bok[e] && k0 + bk[e] < g.K) ? load_elem(B, boff[e], g.precB) : 0.f;
      aoff[e] += astep;
      boff[e] += bstep;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      As[buf][alds[e]] = ra[e];
      Bs[buf][blds[e]] = rb[e];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const uint32_t nk = (g.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (uint32_t kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * BK);   // next tile in flight while this one is multiplied
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = *reinterpret_cast<const f32x4 *>(&As[buf][(wm + 32 * t + i) * BKP + 8 * j + 4 * hi]);
        b[t] = *reinterpret_cast<const f32x4 *>(&Bs[buf][(wn + 32 * t + i) * BKP + 8 * j + 4 * hi]);
      }
      // the MFMA contracts the k of lane-half 0 with the k of lane-half 1: element e of both halves
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][e], b[nb][e], acc[mb][nb], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }
  gemm_store<2, 2>(g, C, acc, bm + wm, bn + wn, lane);
}

// ------------------------------------------------------------------------------------------------
// 16-bit kernels: A and B in one 16-bit type
// ------------------------------------------------------------------------------------------------
// Block = (WR x WC) waves, each owning (MT x NT) MFMA tiles of 32 x 32:  BM = 32 WR MT, BN = 32 WC NT, BK = 64.
//   <2, 2, 2, 2>: 128 x 128, 256 work-items, 64 KiB LDS  -- small problems (more workgroups)
//   <2, 4, 4, 2>: 256 x 256, 512 work-items, 128 KiB LDS -- large problems: half the LDS staging bytes per
//                 MFMA (ds_write_b128 moves only ~79 B/clk/CU, MI355X_MICROARCH.md LDS table: at 128 x 128
//                 the staging writes alone take ~80 % of the matrix time), 0.75 instead of 1 fragment read
//                 per MFMA and 32 instead of 16 MFMAs per wave between barriers.
// LDS image of one operand tile (X rows of M or N, BK = 64 deep):
//   KMAJOR (memory is k-contiguous: A not transposed, B transposed): [x][64 k], 128-byte rows, 16-byte chunk
//     index XOR-swizzled by (x >> 1) & 7 (16 rows of a read group -> 16 distinct 16-byte slots); fragment = ds_read_b128.
//   XMAJOR (memory is x-contiguous: A transposed, B not transposed): [x / 32][64 k][32 x] (64-byte rows);
//     fragment = two ds_read_b64_tr_b16, the V^T gather of the attention forward kernel.
constexpr int GEMM16_BK = 64;

template <int WR, int WC, int MT, int NT> constexpr int gemm16_lds_bytes() { return 2 * (32 * WR * MT + 32 * WC * NT) * GEMM16_BK * 2; }

// AKM / BKM: the operand's image kind (k-major rows or [x/32][k][32]) is a compile-time property, so the k loop is
// one straight-line block (as run-time flags every fragment read was a branch and hipcc could not schedule LDS
// reads against MFMAs across them).
// DMA: tiles go global -> LDS directly (buffer_load_dwordx4 ... lds): instruction e of wave w fills the 1 KiB of an
// image at 16-byte positions (w * CH + e) * 64 + lane, so each lane fetches the chunk that BELONGS at its position
// (the swizzle is applied on the source address).  No staging registers, no ds_write_b128 (whose VGPR -> LDS
// transfer costs ~13 cycles per wave-instruction and overlaps nothing).  Needs whole, 16-byte aligned chunks: the
// host selects it only for K % 8 == 0 and aligned operands.
template <typename T, int WR, int WC, int MT, int NT, bool AKM, bool BKM, bool DMA = false>
__global__ __launch_bounds__(WR * WC * 64) void gemm_16(const GemmArgs g) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  constexpr int BM = 32 * WR * MT, BN = 32 * WC * NT, BK = GEMM16_BK, NTHR = WR * WC * 64;
  constexpr int ATILE = BM * BK * 2, BTILE = BN * BK * 2, STAGE = ATILE + BTILE;
  constexpr int ACH = BM * 8 / NTHR, BCH = BN * 8 / NTHR;   // 16-byte chunks per thread and tile
  static_assert(BM * 8 % NTHR == 0 && BN * 8 % NTHR == 0, "tiles must divide evenly over the workgroup");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A | B)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, hi = lane >> 5;
  const uint32_t bm = blockIdx.y * BM, bn = blockIdx.x * BN;
  const int wm = (wave / WC) * (MT * 32), wn = (wave % WC) * (NT * 32);
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const uint32_t ldA2 = g.ldA * 2, ldB2 = g.ldB * 2;
  // buffer resources bound the matrices (rows x pitch): chunks outside read as zeros
  // (sizes rounded up to whole dwords: the range check is per dword, and with an odd element count the
  // dword that holds the last element ends 2 bytes past the matrix -- still inside the same 4-byte word)
  const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>((const char *)g.A + (uint64_t)blockIdx.z * g.bsA * 2), 0, ((g.transA ? g.K : g.M) * ldA2 + 3u) & ~3u, 0x00020000);
  const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>((const char *)g.B + (uint64_t)blockIdx.z * g.bsB * 2), 0, ((g.transB ? g.N : g.K) * ldB2 + 3u) & ~3u, 0x00020000);
  char *C = (char *)g.C + (uint64_t)blockIdx.z * g.bsC * elem_size(g.precC);

  // staging, chunk id = tid + NTHR e:
  //   KMAJOR: chunk = (x, c) with c = 8-element group along k (8 per row);     global (x0 + x) * ld + k0 + 8c
  //   XMAJOR: chunk = (k, c) with c = 8-element group along x (X / 8 per row); global (k0 + k) * ld + x0 + 8c
  constexpr bool akm = AKM, bkm = BKM;   // host: AKM = !transA, BKM = transB
  uint32_t aoff[ACH], boff[BCH], alds[ACH], blds[BCH];
  uint32_t akk[ACH], bkk[BCH];   // k of the chunk inside a tile (for the k-bound test)
#pragma unroll
  for (int e = 0; e < ACH; ++e) {
    const int idx = DMA ? (wave * ACH + e) * 64 + lane : tid + NTHR * e;   // DMA: 16-byte position in the image
    if (akm) {
      const int x = idx >> 3, c = DMA ? (idx & 7) ^ ((x >> 1) & 7) : idx & 7;
      aoff[e] = (bm + x < g.M) ? (bm + x) * ldA2 + c * 16 : OOB;
      alds[e] = x * 128 + ((c ^ ((x >> 1) & 7)) * 16);
      akk[e] = 8 * c;
    } else {
      const int k = DMA ? (idx >> 2) % BK : idx / (BM / 8), c = DMA ? ((idx >> 2) / BK) * 4 + (idx & 3) : idx % (BM / 8);
      aoff[e] = (bm + 8 * c < g.M) ? k * ldA2 + (bm + 8 * c) * 2 : OOB;
      alds[e] = ((c >> 2) * BK + k) * 64 + (c & 3) * 16;
      akk[e] = k;
    }
  }
#pragma unroll
  for (int e = 0; e < BCH; ++e) {
    const int idx = DMA ? (wave * BCH + e) * 64 + lane : tid + NTHR * e;
    if (bkm) {
      const int x = idx >> 3, c = DMA ? (idx & 7) ^ ((x >> 1) & 7) : idx & 7;
      boff[e] = (bn + x < g.N) ? (bn + x) * ldB2 + c * 16 : OOB;
      blds[e] = x * 128 + ((c ^ ((x >> 1) & 7)) * 16);
      bkk[e] = 8 * c;
    } else {
      const int k = DMA ? (idx >> 2) % BK : idx / (BN / 8), c = DMA ? ((idx >> 2) / BK) * 4 + (idx & 3) : idx % (BN / 8);
      boff[e] = (bn + 8 * c < g.N) ? k * ldB2 + (bn + 8 * c) * 2 : OOB;
      blds[e] = ((c >> 2) * BK + k) * 64 + (c & 3) * 16;
      bkk[e] = k;
    }
  }
  // k-bound: a KMAJOR chunk at k >= K still lies inside the resource (it reads the next row), so the test is
  // explicit; a chunk that straddles K (K % 8 != 0, last k tile only) keeps its first K - k elements.
  // Rows need no 16-byte alignment: gfx950 under ROCm serves unaligned buffer loads (odd leading dimensions
  // just cost extra memory transactions).
  auto keep_first = [](u32x4 r, int n) {   // n in 1..7 sixteen-bit elements
#pragma unroll
    for (int d = 0; d < 4; ++d) r[d] = (2 * d + 1 < n) ? r[d] : (2 * d < n ? (r[d] & 0xFFFFu) : 0u);
    return r;
  };
  u32x4 ra[DMA ? 1 : ACH], rb[DMA ? 1 : BCH];
  auto dma = [&](uint32_t k0, int buf) {   // tile at k0 -> stage buf (DMA only)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    char *abase = smem + buf * STAGE + wave * (ACH * 1024), *bbase = smem + buf * STAGE + ATILE + wave * (BCH * 1024);
#pragma unroll
    for (int e = 0; e < ACH; ++e) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass of hipcc does not know this device builtin
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_ptr)(abase + e * 1024), 16, k0 + akk[e] < g.K ? aoff[e] : OOB, 0, 0, 0);
#endif
      aoff[e] = __builtin_elementwise_add_sat(aoff[e], akm ? (uint32_t)BK * 2 : (uint32_t)BK * ldA2);
    }
#pragma unroll
    for (int e = 0; e < BCH; ++e) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bres, (lds_ptr)(bbase + e * 1024), 16, k0 + bkk[e] < g.K ? boff[e] : OOB, 0, 0, 0);
#endif
      boff[e] = __builtin_elementwise_add_sat(boff[e], bkm ? (uint32_t)BK * 2 : (uint32_t)BK * ldB2);
    }
  };
  auto gload = [&](uint32_t k0) {
    const bool tail = k0 + BK > g.K && (g.K & 7) != 0;   // wave-uniform: only the last k tile of a ragged K
#pragma unroll
    for (int e = 0; e < ACH; ++e) {
      const int idx = tid + NTHR * e;
      const uint32_t ka = akm ? k0 + 8 * (idx & 7) : k0 + idx / (BM / 8);
      ra[e] = __builtin_amdgcn_raw_buffer_load_b128(ares, ka < g.K ? aoff[e] : OOB, 0, 0);
      if (tail && akm && ka < g.K && ka + 8 > g.K) ra[e] = keep_first(ra[e], (int)(g.K - ka));
      aoff[e] = __builtin_elementwise_add_sat(aoff[e], akm ? (uint32_t)BK * 2 : (uint32_t)BK * ldA2);
    }
#pragma unroll
    for (int e = 0; e < BCH; ++e) {
      const int idx = tid + NTHR * e;
      const uint32_t kb = bkm ? k0 + 8 * (idx & 7) : k0 + idx / (BN / 8);
      rb[e] = __builtin_amdgcn_raw_buffer_load_b128(bres, kb < g.K ? boff[e] : OOB, 0, 0);
      if (tail && bkm && kb < g.K && kb + 8 > g.K) rb[e] = keep_first(rb[e], (int)(g.K - kb));
      boff[e] = __builtin_elementwise_add_sat(boff[e], bkm ? (uint32_t)BK * 2 : (uint32_t)BK * ldB2);
    }
  };
  auto lstore = [&](int buf) {
    char *base = smem + buf * STAGE;
#pragma unroll
    for (int e = 0; e < ACH; ++e) *reinterpret_cast<u32x4 *>(base + alds[e]) = ra[e];
#pragma unroll
    for (int e = 0; e < BCH; ++e) *reinterpret_cast<u32x4 *>(base + ATILE + blds[e]) = rb[e];
  };
  auto lstore_chunk = [&](int buf, int e) {   // chunk e of the A chunks followed by the B chunks
    char *base = smem + buf * STAGE;
    if (e < ACH) *reinterpret_cast<u32x4 *>(base + alds[e < ACH ? e : 0]) = ra[e < ACH ? e : 0];
    else *reinterpret_cast<u32x4 *>(base + ATILE + blds[e >= ACH ? e - ACH : 0]) = rb[e >= ACH ? e - ACH : 0];
  };
  // fragment of k-step s (16 k) for the 32 rows x0..x0+31 of a tile image
  const int n16 = lane & 15;
  // transposing gather in NATURAL k order (element j of lane-half hi = k 16 s + 8 hi + j, the order of the
  // k-major ds_read_b128 fragment it may be paired with): the 16-lane groups address rows (n16 >> 2) + 8 hi
  // and, for the second read, 4 rows further.  (The attention kernels use rows + 4 hi / + 8, a permuted
  // order that matches their score registers; here the two operands of one MFMA can come from both images.)
  const int tr_lane = ((n16 >> 2) + 8 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;
  auto fragment = [&](const char *img, bool kmajor, int x0, int s) -> v8 {
    if (kmajor) {
      const int x = x0 + i, c = 2 * s + hi;
      return *reinterpret_cast<const v8 *>(img + x * 128 + ((c ^ ((x >> 1) & 7)) * 16));
    }
    const char *p = img + ((x0 >> 5) * BK + 16 * s) * 64 + tr_lane;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * 64));
    return __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  // DMA mode reads its fragments through inline asm: behind an LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in
  // front of every LDS read it cannot prove disjoint from the DMA's destination (all transposing reads), which
  // drains the prefetch of the next tile.  The asm reads are invisible to that pass; their own completion is
  // awaited by frag_wait below (an asm whose operands are the fragments, so no consumer can move above it).
  auto lds_addr = [](const char *p) { return (uint32_t)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char *)p; };
  // An asm fragment stays in the registers its read(s) named until frag_wait has listed them: a k-major fragment
  // is one 128-bit value, a transposed one two 64-bit halves that are joined only afterwards (nothing but the
  // awaited registers themselves may sit between a read and its wait).
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  struct AsmFrag { u32x4 q; u32x2 lo, hi; };
  auto fragment_asm = [&](const char *img, auto kmajor, int x0, int s) -> AsmFrag {
    AsmFrag f;
    if constexpr (decltype(kmajor)::value) {
      const int x = x0 + i, c = 2 * s + hi;
      asm volatile("ds_read_b128 %0, %1" : "=v"(f.q) : "v"(lds_addr(img + x * 128 + ((c ^ ((x >> 1) & 7)) * 16))));
    } else {
      const uint32_t a = lds_addr(img + ((x0 >> 5) * BK + 16 * s) * 64 + tr_lane);
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(f.hi) : "v"(a));
    }
    return f;
  };
  auto frag_value = [&](const AsmFrag &f, auto kmajor) -> v8 {
    if constexpr (decltype(kmajor)::value) return __builtin_bit_cast(v8, f.q);
    else return __builtin_bit_cast(v8, __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3));
  };
  constexpr int FRAG_OPS = MT * (AKM ? 1 : 2) + NT * (BKM ? 1 : 2);   // LDS instructions per k-step of fragment reads
  static_assert((MT == 4 && NT == 2) || !DMA, "frag_wait lists the fragments of a 4 x 2 wave tile");
  // s_waitcnt with the A fragments as in/out operands, then an empty asm that lists the B fragments (asm volatile
  // statements keep their order): no consumer of either can be scheduled above the wait
  auto frag_wait = [&](AsmFrag (&a)[MT], AsmFrag (&b)[NT], auto more) {
    constexpr int N = decltype(more)::value ? FRAG_OPS : 0;   // the next step's reads were issued after these: LDS returns in order
    if constexpr (AKM)
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0].q), "+v"(a[1].q), "+v"(a[2].q), "+v"(a[3].q) : "n"(N < 15 ? N : 15));
    else
      asm volatile("s_waitcnt lgkmcnt(%8)"
                   : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(a[1].lo), "+v"(a[1].hi), "+v"(a[2].lo), "+v"(a[2].hi), "+v"(a[3].lo), "+v"(a[3].hi)
                   : "n"(N < 15 ? N : 15));
    if constexpr (BKM) asm volatile("" : "+v"(b[0].q), "+v"(b[1].q));
    else asm volatile("" : "+v"(b[0].lo), "+v"(b[0].hi), "+v"(b[1].lo), "+v"(b[1].hi));
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb)
#pragma unroll
    for (int nb = 0; nb < NT; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const uint32_t nk = (g.K + BK - 1) / BK;
  if constexpr (DMA) {
    dma(0, 0);
  } else {
    gload(0);
    lstore(0);
    __syncthreads();
  }
  for (uint32_t kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if constexpr (DMA) {
      // tile kt has had a whole iteration to land; behind the barrier nobody reads the other stage any more
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      __syncthreads();
      if (kt + 1 < nk) dma((kt + 1) * BK, buf ^ 1);
    } else {
      if (kt + 1 < nk) gload((kt + 1) * BK);
    }
    const char *Ai = smem + buf * STAGE, *Bi = Ai + ATILE;
    v8 fa[2][MT], fb[2][NT];   // fragments of k-step s + 1 are requested before the matrix instructions of step s
    AsmFrag ga[2][DMA ? MT : 1], gb[2][DMA ? NT : 1];   // (DMA: as asm fragments)
    constexpr std::integral_constant<bool, AKM> akc{};
    constexpr std::integral_constant<bool, BKM> bkc{};
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      if constexpr (DMA) ga[0][t] = fragment_asm(Ai, akc, wm + 32 * t, 0);
      else fa[0][t] = fragment(Ai, akm, wm + 32 * t, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (DMA) gb[0][t] = fragment_asm(Bi, bkc, wn + 32 * t, 0);
      else fb[0][t] = fragment(Bi, bkm, wn + 32 * t, 0);
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      v8 (&a)[MT] = fa[s & 1];
      v8 (&b)[NT] = fb[s & 1];
      if (s + 1 < BK / 16) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          if constexpr (DMA) ga[(s + 1) & 1][t] = fragment_asm(Ai, akc, wm + 32 * t, s + 1);
          else fa[(s + 1) & 1][t] = fragment(Ai, akm, wm + 32 * t, s + 1);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if constexpr (DMA) gb[(s + 1) & 1][t] = fragment_asm(Bi, bkc, wn + 32 * t, s + 1);
          else fb[(s + 1) & 1][t] = fragment(Bi, bkm, wn + 32 * t, s + 1);
        }
        __builtin_amdgcn_sched_barrier(0x406);   // LDS and matrix instructions keep this order; vector / scalar work may move
      }
      if constexpr (DMA) {
        if (s + 1 < BK / 16) frag_wait(ga[s & 1], gb[s & 1], std::true_type{});
        else frag_wait(ga[s & 1], gb[s & 1], std::false_type{});
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t] = frag_value(ga[s & 1][t], akc);
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = frag_value(gb[s & 1][t], bkc);
      }
      // The staging writes of the next k tile go between the matrix instructions of the last k-step instead of
      // one burst in front of the barrier (a wide LDS store occupies the store path for ~13 cycles and loads do
      // not overlap it).  The other stage has no readers in this iteration; past the last tile the registers hold
      // stale data that nobody reads.
#pragma unroll
      for (int mb = 0; mb < MT; ++mb)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
          if (!DMA && s == BK / 16 - 1) {
            constexpr int PER = (ACH + BCH) / (MT * NT);
            static_assert((ACH + BCH) % (MT * NT) == 0, "staging chunks must spread evenly over the last k-step");
#pragma unroll
            for (int e = 0; e < PER; ++e) lstore_chunk(buf ^ 1, (mb * NT + nb) * PER + e);
          }
          acc[mb][nb] = F::mfma(a[mb], b[nb], acc[mb][nb]);
        }
    }
    if constexpr (!DMA) __syncthreads();
  }
  gemm_store<MT, NT>(g, C, acc, bm + wm, bn + wn, lane);
}

} // namespace mfa
