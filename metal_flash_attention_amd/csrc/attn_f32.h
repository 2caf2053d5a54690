// attn_f32.h -- the three attention kernels for the FP32 production case (BASELINE config 3): every operand FP32, row-major,
// 16-byte aligned, head dimension a multiple of 4 and <= DP (64 or 128).  Same arithmetic as the general kernels of
// attn_generic.h (all products on v_mfma_f32_32x32x2_f32, exact online softmax), which keep every other layout / type / mask;
// what differs is how the operands travel:
//   * the traversal-side tiles (K, V / Q, dO: 32 rows x DP floats) go from memory to LDS by LDS-DMA, two (forward) or three
//     (backward) stages, one barrier per tile; bounds-checked buffer resources zero-fill ragged rows and the columns past D
//     (what the reference gets from simdgroup_event::async_copy, GEMMHeaders.swift:166-193);
//   * the cached left-hand operands (+Caching.swift:18-281) are read from memory straight into their fragment registers;
//   * every LDS read is a ds_read_b128 that feeds four matrix instructions.  The contraction index of the first product of
//     a pair and the head-dimension index of the second are both PERMUTED so that the four values lie next to each other in a
//     row-major tile: step t of a first product contracts d = 8 (t >> 2) + 4 hi + (t & 3); accumulator block db of a second
//     product holds d = NDB i + db on lane i (NDB = DP / 32) -- a lane then owns NDB consecutive floats of an output row and
//     stores them to memory directly;
//   * rows of a tile are 16 DP / 4 chunks of 16 bytes, chunk c of row r stored at c ^ (r & 15): both read patterns are
//     conflict-free and the image is what a linear LDS-DMA of permuted global addresses produces.
// The reads are inline asm with counted waits (hipcc drains vmcnt in front of LDS reads it cannot prove disjoint from an
// LDS-DMA in flight, attn_fwd16_v3.h).
//
// v_mfma_f32_32x32x2_f32 runs on the vector ALU: an ordinary VALU instruction beside it is NOT hidden -- a lone v_xor costs
// ~15 clocks, softmax arithmetic in a batch 6.5-11 per instruction, LDS reads / waits / SALU nothing
// (tools/probe_f32_mfma.hip, profiles/r04_f32/probe_f32_mfma.txt).  Hence:
//   * no address arithmetic in the loop: every read address is a register computed once, the stage / operand / row parts are
//     immediate offsets (the tile loop is unrolled over the stages); the LDS-DMA's per-lane offsets are constants, the
//     bounds-checked resource advances instead (scalar ALU);
//   * scale and subtract ride on the matrix instructions: the cached fragments are pre-multiplied by the softmax scale and
//     the accumulator of a first product STARTS at -L / -D (srcC of its first instruction; forward: -m as one more contraction
//     step, ones . (-m)), so the softmax arithmetic left per score is one exponential plus one add (forward) or one multiply
//     (backward);
//   * masks (ragged edge, causal diagonal) sit behind wave-uniform branches: only the tiles that need them pay;
//   * staging is issued in the matrix instructions' shadow: an LDS-DMA instruction alone costs ~45 clocks of issue, ~12 between
//     two groups of matrix instructions (phase clocks: MFA_F32_PROF=1 in the developer library, tools/f32_perf.py).
#pragma once
#include "attn_common.h"
#include "attn_fwd16_common.h"   // Fwd16Grid, fwd16_decode_block (XCD-aware workgroup order)
#include <type_traits>

namespace mfa {
namespace f32k {

typedef __attribute__((address_space(3))) void *lds_ptr;
constexpr uint32_t OOB = 0xFFFFFF00u;
constexpr int BT = 32;   // rows of a traversal-side tile

__device__ __forceinline__ uint32_t lds_addr(const char *p) {
  return (uint32_t)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char *)p;
}
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
template <int OFF> __device__ __forceinline__ f32x4 rd128(uint32_t addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF> __device__ __forceinline__ f32x2 rd64(uint32_t addr) {
  f32x2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// lane l <-> lane l ^ 32 (v_permlane32_swap; the s_nop covers the VALU-write -> permlane-read hazard inside the asm string)
__device__ __forceinline__ void half_swap(float x, float *a, float *b) {
  uint32_t b0 = __builtin_bit_cast(uint32_t, x), b1 = b0;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(b0), "+v"(b1));
  *a = __builtin_bit_cast(float, b0);
  *b = __builtin_bit_cast(float, b1);
}
__device__ __forceinline__ float half_max(float x) { float a, b; half_swap(x, &a, &b); return fmaxf(a, b); }
__device__ __forceinline__ float half_add(float x) { float a, b; half_swap(x, &a, &b); return a + b; }

// The matrix instructions are inline asm with explicit register files: hipcc's allocator, left to itself, parks fragments and
// addresses in accumulation registers and moves them back per use (a v_accvgpr_read each -- VALU time, see below), and copies
// whole accumulators at the loop's back-edge.  Here: cached fragments (the B operand of a first product) and the accumulators
// of a second product live in AGPRs for the whole kernel; the LDS-fed A operand, the score accumulators (which the softmax
// arithmetic reads) and P / dS (the B operand of a second product) in VGPRs.  C and D of an instruction share one file.
// asm volatile keeps the stream in program order.  What hipcc no longer does for these instructions: the wait states between
// a matrix instruction's write and a VALU / memory read of it -- mfma_fence() (19 for the 16-pass instruction, gfx940 rules).
// (one asm statement per GROUP of instructions: hipcc puts a wait state between asm statements that touch the same registers)
#define MFA_MFMA "v_mfma_f32_32x32x2_f32 "
// four steps of ONE first product: d += a[i] . b[i] (VGPR accumulators, B in AGPRs)
// WAIT: the statement starts with s_waitcnt lgkmcnt(WAIT) -- the A operands are LDS reads in flight (at most WAIT younger reads
// may still be pending; a separate wait statement costs a wait state between the two asm statements)
template <int WAIT> __device__ __forceinline__ void mfma_group(f32x16 &d, const f32x4 &a, const float *b) {
  asm volatile("s_waitcnt lgkmcnt(%9)\n\t" MFA_MFMA "%0, %1, %5, %0\n\t" MFA_MFMA "%0, %2, %6, %0\n\t" MFA_MFMA "%0, %3, %7, %0\n\t" MFA_MFMA "%0, %4, %8, %0"
               : "+v"(d) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "a"(b[0]), "a"(b[1]), "a"(b[2]), "a"(b[3]), "n"(WAIT));
}
// the bias step of the forward: d = ones . bias (all VGPR)
__device__ __forceinline__ void mfma_bias(f32x16 &d, float a, float b) {   // (s_nop: as in mfma_zero, `b` may be fresh)
  asm volatile("s_nop 1\n\t" MFA_MFMA "%0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
// four steps of TWO first products, alternating: d0 += a0[i] . b0[i], d1 += a1[i] . b1[i]
template <int WAIT> __device__ __forceinline__ void mfma_group_pair(f32x16 &d0, const f32x4 &a0, const float *b0, f32x16 &d1, const f32x4 &a1, const float *b1) {
  asm volatile("s_waitcnt lgkmcnt(%18)\n\t" MFA_MFMA "%0, %2, %10, %0\n\t" MFA_MFMA "%1, %6, %14, %1\n\t" MFA_MFMA "%0, %3, %11, %0\n\t" MFA_MFMA "%1, %7, %15, %1\n\t"
               MFA_MFMA "%0, %4, %12, %0\n\t" MFA_MFMA "%1, %8, %16, %1\n\t" MFA_MFMA "%0, %5, %13, %0\n\t" MFA_MFMA "%1, %9, %17, %1"
               : "+v"(d0), "+v"(d1)
               : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]),
                 "a"(b0[0]), "a"(b0[1]), "a"(b0[2]), "a"(b0[3]), "a"(b1[0]), "a"(b1[1]), "a"(b1[2]), "a"(b1[3]), "n"(WAIT));
}
// the same with start values: d0 = a0 . b0 + c0, d1 = a1 . b1 + c1 for the first step
template <int WAIT>
__device__ __forceinline__ void mfma_group_pair_init(f32x16 &d0, const f32x4 &a0, const float *b0, const f32x16 &c0, f32x16 &d1, const f32x4 &a1, const float *b1,
                                                     const f32x16 &c1) {
  asm volatile("s_waitcnt lgkmcnt(%20)\n\t" MFA_MFMA "%0, %2, %10, %18\n\t" MFA_MFMA "%1, %6, %14, %19\n\t" MFA_MFMA "%0, %3, %11, %0\n\t" MFA_MFMA "%1, %7, %15, %1\n\t"
               MFA_MFMA "%0, %4, %12, %0\n\t" MFA_MFMA "%1, %8, %16, %1\n\t" MFA_MFMA "%0, %5, %13, %0\n\t" MFA_MFMA "%1, %9, %17, %1"
               : "=&v"(d0), "=&v"(d1)
               : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]),
                 "a"(b0[0]), "a"(b0[1]), "a"(b0[2]), "a"(b0[3]), "a"(b1[0]), "a"(b1[1]), "a"(b1[2]), "a"(b1[3]), "v"(c0), "v"(c1), "n"(WAIT));
}
// one step of a second product over its NDB accumulator blocks (AGPR accumulators, A and B in VGPRs): acc[db] += a[db] . p
template <int WAIT> __device__ __forceinline__ void mfma_out(f32x16 *acc, const f32x4 &a, float p) {
  asm volatile("s_waitcnt lgkmcnt(%9)\n\t" MFA_MFMA "%0, %4, %8, %0\n\t" MFA_MFMA "%1, %5, %8, %1\n\t" MFA_MFMA "%2, %6, %8, %2\n\t" MFA_MFMA "%3, %7, %8, %3"
               : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(p), "n"(WAIT));
}
template <int WAIT> __device__ __forceinline__ void mfma_out(f32x16 *acc, const f32x2 &a, float p) {
  asm volatile("s_waitcnt lgkmcnt(%5)\n\t" MFA_MFMA "%0, %2, %4, %0\n\t" MFA_MFMA "%1, %3, %4, %1"
               : "+a"(acc[0]), "+a"(acc[1]) : "v"(a[0]), "v"(a[1]), "v"(p), "n"(WAIT));
}
// one step of TWO second products, alternating
template <int WAIT> __device__ __forceinline__ void mfma_out_pair(f32x16 *acc0, const f32x4 &a0, float p0, f32x16 *acc1, const f32x4 &a1, float p1) {
  asm volatile("s_waitcnt lgkmcnt(%18)\n\t" MFA_MFMA "%0, %8, %16, %0\n\t" MFA_MFMA "%4, %12, %17, %4\n\t" MFA_MFMA "%1, %9, %16, %1\n\t" MFA_MFMA "%5, %13, %17, %5\n\t"
               MFA_MFMA "%2, %10, %16, %2\n\t" MFA_MFMA "%6, %14, %17, %6\n\t" MFA_MFMA "%3, %11, %16, %3\n\t" MFA_MFMA "%7, %15, %17, %7"
               : "+a"(acc0[0]), "+a"(acc0[1]), "+a"(acc0[2]), "+a"(acc0[3]), "+a"(acc1[0]), "+a"(acc1[1]), "+a"(acc1[2]), "+a"(acc1[3])
               : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(p0), "v"(p1), "n"(WAIT));
}
template <int WAIT> __device__ __forceinline__ void mfma_out_pair(f32x16 *acc0, const f32x2 &a0, float p0, f32x16 *acc1, const f32x2 &a1, float p1) {
  asm volatile("s_waitcnt lgkmcnt(%10)\n\t" MFA_MFMA "%0, %4, %8, %0\n\t" MFA_MFMA "%2, %6, %9, %2\n\t" MFA_MFMA "%1, %5, %8, %1\n\t" MFA_MFMA "%3, %7, %9, %3"
               : "+a"(acc0[0]), "+a"(acc0[1]), "+a"(acc1[0]), "+a"(acc1[1]) : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(p0), "v"(p1), "n"(WAIT));
}
// an accumulator block in AGPRs, zeroed by the matrix pipe itself (0 . 0 + 0): every definition and use of the block is then an
// asm operand of the "a" class, and hipcc keeps it there across the loop (a block it zeroes itself starts out in VGPRs and is
// copied in and out around every asm statement)
// (s_nop: the zero operand has just been written by a VALU instruction -- hipcc does not know the asm reads it as a matrix operand)
__device__ __forceinline__ void mfma_zero(f32x16 &d) {
  asm volatile("s_nop 4\n\t" MFA_MFMA "%0, %1, %1, 0" : "=&a"(d) : "v"(0.f));
}
__device__ __forceinline__ void mfma_fence(f32x16 &a, f32x16 &b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void mfma_fence(f32x16 &a) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a)); }
template <int N> __device__ __forceinline__ void mfma_fence_out(f32x16 *acc) {
  if constexpr (N == 4) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
  else asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[0]), "+a"(acc[1]));
}
// a fragment array, multiplied by `factor`, moves to the accumulation registers (where every later use -- "a" operands -- keeps it)
template <int N> __device__ __forceinline__ void pin_fragments(float *f, float factor) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float x = f[i] * factor;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(f[i]) : "v"(x));
  }
}
// (VALU results feed the B operand of the next matrix instructions)
__device__ __forceinline__ void valu_fence(f32x16 &a) { asm volatile("s_nop 4" : "+v"(a)); }
__device__ __forceinline__ void valu_fence(f32x16 &a, f32x16 &b) { asm volatile("s_nop 4" : "+v"(a), "+v"(b)); }

template <int DP> struct Geo {
  static_assert(DP == 64 || DP == 128, "head-dimension buckets of the FP32 kernels");
  static constexpr int ROWB = DP * 4;            // bytes per tile row
  static constexpr int CPR = DP / 4;             // 16-byte chunks per row
  static constexpr int TILE = BT * ROWB;         // bytes per tile
  static constexpr int NI = CPR / 8;             // LDS-DMA instructions per wave and tile (4 waves x 64 lanes x 16 bytes each)
  static constexpr int NDB = DP / 32;            // accumulator blocks of a second product
  static constexpr int NG = DP / 8;              // b128 reads of a first product (four contraction steps each)
  static constexpr int RB2 = NDB * 4;            // bytes a lane reads per step of a second product
};

// resource over the rows of one operand of one (head, batch): `rows` x ld floats, reads past the end return zero
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_resource(const OperandView &v, uint32_t head, uint32_t batch, int rows) {
  return __builtin_amdgcn_make_buffer_rsrc(operand_base(v, head, batch), 0, (uint32_t)rows * (uint32_t)v.ld * 4u, 0x00020000);
}

// cached fragments of a first product's right-hand operand: f[4 T + i] = X[row][8 T + 4 hi + i]
template <int DP> __device__ __forceinline__ void load_fragments(float *f, const __amdgpu_buffer_rsrc_t &res, uint32_t rowoff, bool valid, int hi, int D) {
  // all NG loads go out back to back (offsets by mask arithmetic: a conditional here becomes a branch around each load, with a
  // wait behind it -- the loads of a prologue then queue up one memory latency after the other)
  f32x4 raw[Geo<DP>::NG];
  static_for<Geo<DP>::NG>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    const int d0 = 8 * T + 4 * hi;
    const uint32_t keep = (uint32_t)-(int32_t)(valid & (d0 < D));   // all ones / zero
    const uint32_t off = ((rowoff + d0 * 4) & keep) | (OOB & ~keep);
    // (the whole vector is cast: __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index, hipcc 7.2)
    raw[T] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res, off, 0, 0));
  });
  static_for<Geo<DP>::NG>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[4 * T + i] = raw[T][i];
  });
}

// the tile stager of one operand: per-lane byte offsets (inside the tile) of the NI chunks this lane moves per tile -- constants;
// the bounds-checked resource of the tile advances instead (tile_resource: scalar arithmetic)
template <int DP> struct Stager {
  uint32_t off[Geo<DP>::NI];
  __device__ __forceinline__ void init(int wave, int lane, uint32_t ld, int D) {
    typedef Geo<DP> G;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
      const int p = (wave * G::NI + i) * 64 + lane;
      const int r = p / G::CPR, c = (p % G::CPR) ^ (r & 15);
      off[i] = (c * 4 < D) ? r * ld * 4u + c * 16 : OOB;
    }
  }
  // one of the NI pieces (an LDS-DMA instruction costs ~45 clocks of issue when nothing runs beside it, none in the shadow of a
  // matrix instruction: the kernels deal the pieces out between the steps of a second product)
  template <int I> __device__ __forceinline__ void piece(const __amdgpu_buffer_rsrc_t &res, char *base, int wave) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (I < Geo<DP>::NI) __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (lds_ptr)(base + (wave * Geo<DP>::NI + I) * 1024), 16, off[I], 0, 0, 0);
#endif
  }
  // tile -> LDS image at `base` (workgroup-relative bytes); `res` covers the rows from the tile's first one on
  __device__ __forceinline__ void issue(const __amdgpu_buffer_rsrc_t &res, char *base, int wave) const {
    typedef Geo<DP> G;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass of hipcc does not know this device builtin
      __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (lds_ptr)(base + (wave * G::NI + i) * 1024), 16, off[i], 0, 0, 0);
#endif
    }
  }
};
// resource over rows [row0, rows) of an operand (ld floats per row) of one (head, batch): reads past the end return zero
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_resource(const char *base, uint32_t ld, int rows, int row0) {
  const int left = rows > row0 ? rows - row0 : 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base) + (int64_t)row0 * ld * 4, 0, (uint32_t)left * ld * 4u, 0x00020000);
}


// read addresses of a lane (bytes, the workgroup's LDS base included; stage, operand and row offsets are immediates)
//   first product:  row i, chunk (2 T + hi) ^ (i & 15), T = 0 .. NG-1
//   second product: row crow(t, hi), NDB floats at column NDB i: its chunk ^ (row & 15), where row & 15 = 4 hi ^ ct(t),
//                   ct(t) = (t & 3) | 8 ((t >> 2) & 1) -- eight different addresses, + ((t & 3) + 8 (t >> 2)) ROWB as an immediate
constexpr int second_ct(int t) { return (t & 3) | (8 * ((t >> 2) & 1)); }
constexpr int second_index(int t) { return (t & 3) + 4 * ((t >> 2) & 1); }
constexpr int second_row(int t) { return (t & 3) + 8 * (t >> 2); }
template <int DP> struct Addresses {
  uint32_t first[8];    // T & 7; bit 8 of the address is clear, so groups 8 .. 15 (DP = 128) are these + 256 as an immediate
  uint32_t second[8];
  __device__ __forceinline__ void init(uint32_t lbase, int i, int hi) {
    typedef Geo<DP> G;
    const uint32_t f = lbase + i * G::ROWB + ((hi ^ (i & 15)) << 4);
#pragma unroll
    for (int T = 0; T < 8; ++T) first[T] = f ^ (T << 5);
    const int byte = i * G::RB2;
    const uint32_t s = lbase + 4 * hi * G::ROWB + ((((byte >> 4) ^ (4 * hi)) << 4) | (byte & 15));
#pragma unroll
    for (int c = 0; c < 8; ++c) second[c] = s ^ (((c & 3) | (8 * (c >> 2))) << 4);
  }
};

// acc = init + X_tile (first pattern, at immediate offset OFF) . f : NG reads, four matrix instructions each, RING reads ahead
// (forward: the start value -m arrives as one extra contraction step, ones . bias -- a matrix instruction instead of a block of
// sixteen registers that every change of m would have to rewrite)
// `between(T)` runs after every group (T an integral_constant): the next tile's LDS-DMA pieces, issued in the matrix instructions' shadow
template <int DP, int OFF, int RING = 4, typename Between>
__device__ __forceinline__ f32x16 first_product(float ones, float bias, const Addresses<DP> &ad, const float *f, Between &&between) {
  typedef Geo<DP> G;
  f32x4 ring[RING];
  f32x16 acc;
  static_for<RING>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    ring[T] = rd128<OFF + (T >> 3) * 256>(ad.first[T & 7]);
  });
  mfma_bias(acc, ones, bias);
  static_for<G::NG>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    constexpr int pending = (G::NG - 1 - T) < (RING - 1) ? (G::NG - 1 - T) : (RING - 1);
    const f32x4 v = ring[T % RING];
    mfma_group<pending>(acc, v, f + 4 * T);
    if constexpr (T + RING < G::NG) ring[T % RING] = rd128<OFF + ((T + RING) >> 3) * 256>(ad.first[(T + RING) & 7]);
    between(T_);
  });
  return acc;
}

// two first products side by side (the matrix instructions alternate): acc0 = init0 + X0 . f0, acc1 = init1 + X1 . f1
// (INIT = false: the accumulators hold their start values already -- acc += ...)
struct FirstRings { static constexpr int RING = 3; f32x4 r0[RING], r1[RING]; };
// the first RING read pairs of two first products (tiles at immediate offsets OFF0, OFF1)
template <int DP, int OFF0, int OFF1> __device__ __forceinline__ void first_prefetch_pair(FirstRings &fr, const Addresses<DP> &ad) {
  static_for<FirstRings::RING>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    fr.r0[T] = rd128<OFF0 + (T >> 3) * 256>(ad.first[T & 7]);
    fr.r1[T] = rd128<OFF1 + (T >> 3) * 256>(ad.first[T & 7]);
  });
}
// ... and the products themselves (the prefetch has been issued; nothing younger than it is pending)
template <int DP, int OFF0, int OFF1, bool INIT = true>
__device__ __forceinline__ void first_body_pair(f32x16 &acc0, const f32x16 &init0, const float *f0, f32x16 &acc1, const f32x16 &init1, const float *f1,
                                                FirstRings &fr, const Addresses<DP> &ad) {
  typedef Geo<DP> G;
  constexpr int RING = FirstRings::RING;
  static_for<G::NG>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    constexpr int left = G::NG - 1 - T;
    constexpr int pending = 2 * (left < (RING - 1) ? left : (RING - 1));
    const f32x4 v0 = fr.r0[T % RING], v1 = fr.r1[T % RING];
    if constexpr (T == 0 && INIT) mfma_group_pair_init<pending>(acc0, v0, f0, init0, acc1, v1, f1, init1);
    else mfma_group_pair<pending>(acc0, v0, f0 + 4 * T, acc1, v1, f1 + 4 * T);
    if constexpr (T + RING < G::NG) {
      fr.r0[T % RING] = rd128<OFF0 + ((T + RING) >> 3) * 256>(ad.first[(T + RING) & 7]);
      fr.r1[T % RING] = rd128<OFF1 + ((T + RING) >> 3) * 256>(ad.first[(T + RING) & 7]);
    }
  });
}
// one read of the second pattern: step T of the tile at immediate offset OFF
template <int DP, int OFF, int T> __device__ __forceinline__ auto second_read(const Addresses<DP> &ad) {
  typedef Geo<DP> G;
  if constexpr (G::NDB == 4) return rd128<OFF + second_row(T) * G::ROWB>(ad.second[second_index(T)]);
  else return rd64<OFF + second_row(T) * G::ROWB>(ad.second[second_index(T)]);
}
template <int DP, int N = 4> struct SecondRing {
  typedef std::conditional_t<Geo<DP>::NDB == 4, f32x4, f32x2> Vec;
  static constexpr int RING = N;
  Vec v[RING];
};
// the first RING reads of a second product (issued early: they are in flight during the softmax arithmetic)
template <int DP, int OFF, int N> __device__ __forceinline__ void second_prefetch(SecondRing<DP, N> &ring, const Addresses<DP> &ad) {
  static_for<N>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    ring.v[T] = second_read<DP, OFF, T>(ad);
  });
}
// the same for two tiles read side by side: a0 b0 a1 b1 ... (second_product_pair waits for them in that order); HALF = 0, 1:
// the first / second half of the ring (the caller may put a wait between them: the counter holds 15)
template <int DP, int OFF0, int OFF1, int HALF> __device__ __forceinline__ void second_prefetch_pair(SecondRing<DP> &r0, SecondRing<DP> &r1, const Addresses<DP> &ad) {
  constexpr int H = SecondRing<DP>::RING / 2;
  static_for<H>([&](auto T_) {
    constexpr int T = decltype(T_)::value + HALF * H;
    r0.v[T] = second_read<DP, OFF0, T>(ad);
    r1.v[T] = second_read<DP, OFF1, T>(ad);
  });
}
// acc[db] += X_tile^T (second pattern) . p : 16 reads, NDB matrix instructions each
// `ahead()` (HOOKED: it issues EXTRA reads) runs once the last read of this product is on its way -- after step 15 - RING; the
// remaining steps then allow EXTRA more pending reads (all of them younger than this product's)
// `between(T)` runs after every step (T an integral_constant: staging pieces in the matrix instructions' shadow)
template <int DP, int OFF, int RING, int EXTRA = 0, typename Ahead, typename Between>
__device__ __forceinline__ void second_product(f32x16 *acc, SecondRing<DP, RING> &ring, const Addresses<DP> &ad, const f32x16 &p, Ahead &&ahead, Between &&between) {
  typedef Geo<DP> G;
  static_for<16>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    constexpr int pending = ((15 - T) < (RING - 1) ? (15 - T) : (RING - 1)) + (T > 15 - RING ? EXTRA : 0);
    const auto v = ring.v[T % RING];
    mfma_out<pending>(acc, v, p[T]);
    if constexpr (T + RING < 16) ring.v[T % RING] = second_read<DP, OFF, T + RING>(ad);
    if constexpr (T == 15 - RING) ahead();
    between(T_);
  });
}
template <int DP, int OFF, int RING> __device__ __forceinline__ void second_product(f32x16 *acc, SecondRing<DP, RING> &ring, const Addresses<DP> &ad, const f32x16 &p) {
  second_product<DP, OFF, RING, 0>(acc, ring, ad, p, []() {}, [](auto) {});
}
// two second products side by side (dV and dK): reads of tile 0 and tile 1 alternate; `ahead` / `between` / EXTRA as in second_product
template <int DP, int OFF0, int OFF1, int EXTRA = 0, typename Ahead, typename Between>
__device__ __forceinline__ void second_product_pair(f32x16 *acc0, const f32x16 &p0, f32x16 *acc1, const f32x16 &p1, SecondRing<DP> &r0, SecondRing<DP> &r1,
                                                    const Addresses<DP> &ad, Ahead &&ahead, Between &&between) {
  typedef Geo<DP> G;
  constexpr int RING = SecondRing<DP>::RING;
  static_for<16>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    constexpr int left = 15 - T;
    constexpr int pending = 2 * (left < (RING - 1) ? left : (RING - 1)) + (T > 15 - RING ? EXTRA : 0);
    const auto v0 = r0.v[T % RING];
    const auto v1 = r1.v[T % RING];
    mfma_out_pair<pending>(acc0, v0, p0[T], acc1, v1, p1[T]);
    if constexpr (T + RING < 16) {
      r0.v[T % RING] = second_read<DP, OFF0, T + RING>(ad);
      r1.v[T % RING] = second_read<DP, OFF1, T + RING>(ad);
    }
    if constexpr (T == 15 - RING) ahead();
    between(T_);
  });
}

// a lane's share of an output row block: acc[db][r] = X[row of the lane][NDB crow(r, hi) + db]
template <int DP> __device__ __forceinline__ void store_rows(const f32x16 *acc, const OperandView &v, uint32_t head, uint32_t batch, int64_t row, int rows,
                                                             int hi, int D, float factor) {
  typedef Geo<DP> G;
  if (row >= rows) return;
  float *dst = reinterpret_cast<float *>(operand_base(v, head, batch)) + row * v.ld;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = G::NDB * crow(r, hi);
    if (d < D) {
      if constexpr (G::NDB == 4) *reinterpret_cast<f32x4 *>(dst + d) = f32x4{acc[0][r] * factor, acc[1][r] * factor, acc[2][r] * factor, acc[3][r] * factor};
      else *reinterpret_cast<f32x2 *>(dst + d) = f32x2{acc[0][r] * factor, acc[1][r] * factor};
    }
  }
}
__device__ __forceinline__ f32x16 splat16(float x) { return f32x16{x, x, x, x, x, x, x, x, x, x, x, x, x, x, x, x}; }

template <int DP> constexpr int lds_bytes() { return 2 /*stages*/ * 2 /*operands*/ * Geo<DP>::TILE; }
template <int DP> constexpr int lds_bytes_dq() { return 3 /*stages*/ * 2 /*operands*/ * Geo<DP>::TILE; }
template <int DP> constexpr int lds_bytes_dkv() { return lds_bytes_dq<DP>() + 3 /*stages*/ * 512; }   // + the L and D slices of a tile
constexpr int ROWS = 128;   // rows (forward, dQ) or columns (dK/dV) of a workgroup: four waves x 32
constexpr float RESCALE_ABOVE = 8.f;   // forward: O, l and the reference maximum are re-based when a row's maximum grew by more than 2^8

// ----------------------------------------------------------------------------------------------
// forward: O = softmax(Q K^T / sqrt(D)) V,  L = m + log2(l)       (+Source.swift:158-200)
// grid: 1-D, ceil(R / 128) x heads x batches in fwd16_decode_block's order; 256 threads; two workgroups per compute unit
// ----------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256, 2) void attn_f32_fwd(const KernelArgs a, const Fwd16Grid grid) {
  typedef Geo<DP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &rblk, &head, &batch);
  if (a.causal) rblk = grid.rowBlocks - 1 - rblk;   // later row blocks traverse more keys: start them first
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * ROWS;
  if (r0 >= R) return;
  const int64_t row = r0 + wave * 32 + q;
  const bool causal = a.causal != 0;

  const __amdgpu_buffer_rsrc_t qres = rows_resource(a.op[SLOT_Q], head, batch, R);
  const char *kbase = operand_base(a.op[SLOT_K], head, batch), *vbase = operand_base(a.op[SLOT_V], head, batch);
  const uint32_t ldk = (uint32_t)a.op[SLOT_K].ld, ldv = (uint32_t)a.op[SLOT_V].ld;
  // Q fragments, pre-multiplied by log2(e) / sqrt(D): the scores leave the matrix instructions in base-2 units
  float qf[DP / 2];
  load_fragments<DP>(qf, qres, (uint32_t)row * (uint32_t)a.op[SLOT_Q].ld * 4u, row < R, hi, D);
  pin_fragments<DP / 2>(qf, a.scale2);

  f32x16 o[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db) mfma_zero(o[db]);
  // online softmax (+Softmax.swift:267-324) around a REFERENCE maximum m: the score accumulator starts at -m, so a tile's
  // scores arrive as s - m; m follows the row maximum when that grew by more than RESCALE_ABOVE (O and l are re-based then,
  // onlineCorrectO) -- the first tile always sets it
  float m = 0.f, l = 0.f;
  const float ones = hi == 0 ? 1.f : 0.f;   // contraction step (hi): A = (1, 0), B = (-m, 0)
  float bias = 0.f;

  // causal extension: row r sees column c iff c <= r + coff; columns past the last row's limit are never visited
  const int coff = causal_offset(R, C);
  const int Cend = causal ? (int)min((int64_t)C, min((int64_t)R, r0 + ROWS) + coff) : C;
  const int nt = (Cend + BT - 1) / BT;
  Stager<DP> ks, vs;
  ks.init(wave, lane, ldk, D);
  vs.init(wave, lane, ldv, D);
  Addresses<DP> ad;
  ad.init(lds_addr(smem), q, hi);
  const int limit = (int)row + coff;
  const int wavelimit = (int)r0 + wave * 32 + coff;   // the wave's smallest limit: tiles reaching past it mask
  if (nt > 0) {
    ks.issue(tile_resource(kbase, ldk, C, 0), smem, wave);
    vs.issue(tile_resource(vbase, ldv, C, 0), smem + G::TILE, wave);
  }
  auto tile = [&](auto STAGE_, int j) {
    constexpr int STAGE = decltype(STAGE_)::value;
    constexpr int KOFF = STAGE * 2 * G::TILE, VOFF = KOFF + G::TILE, NEXT = (STAGE ^ 1) * 2 * G::TILE;
    const int c0 = j * BT;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of tile j have landed
    __syncthreads();                      // ... everybody's; and every wave has finished tile j - 1, whose stage is written next
    // S^T = K Q^T - m : lane holds query `row`, keys c0 + crow(r, hi)
    const bool more = j + 1 < nt;
    const __amdgpu_buffer_rsrc_t kres1 = tile_resource(kbase, ldk, C, c0 + BT), vres1 = tile_resource(vbase, ldv, C, c0 + BT);
    f32x16 s = first_product<DP, KOFF, 3>(ones, bias, ad, qf, [&](auto T_) {
      constexpr int T = decltype(T_)::value;
      if constexpr (T < 2 * G::NI) {
        if (more) {
          if constexpr (T % 2 == 0) ks.template piece<T / 2>(kres1, smem + NEXT, wave);
          else vs.template piece<T / 2>(vres1, smem + NEXT + G::TILE, wave);
        }
      }
    });
    SecondRing<DP, 3> vring;
    second_prefetch<DP, VOFF>(vring, ad);
    mfma_fence(s);
    if (c0 + BT > C) { // maskAttentionMatrixEdge, +Softmax.swift:228-260
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) >= C) s[r] = mask_value();
    }
    if (causal && c0 + BT - 1 > wavelimit) {    // same mechanism, applied to the columns the row may not see
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) > limit) s[r] = mask_value();
    }
    // onlineReduceMaximum, +Softmax.swift:267-288 (relative to m)
    float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
    mx = half_max(fmaxf(mx, s[15]));
    const bool grow = j == 0 || mx > RESCALE_ABOVE;
    if (__builtin_amdgcn_ballot_w64(grow) != 0) {   // wave-uniform: rare after the first tiles
      const float delta = grow ? mx : 0.f;
      const float corr = j == 0 ? 1.f : fast_exp2(-delta);   // onlineCorrectO, +Softmax.swift:290-301 (first tile: O = l = 0)
      m += delta;
      l *= corr;
      mfma_fence_out<G::NDB>(o);
#pragma unroll
      for (int db = 0; db < G::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] -= delta;
      bias = hi == 0 ? -m : 0.f;
    }
    // softmax + onlineReduceSum, +Softmax.swift:304-324, :406-417
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(s[r]); psum += s[r]; }
    l += psum;
    valu_fence(s);
    // O^T += V^T P^T with the key index permuted: step t uses key crow(t, hi)
    second_product<DP, VOFF>(o, vring, ad, s);
  };
  for (int j = 0; j < nt; j += 2) {
    tile(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nt) tile(std::integral_constant<int, 1>{}, j + 1);
  }

  mfma_fence_out<G::NDB>(o);
  const float l_tot = half_add(l);
  const float inv = (m > -1e37f && l_tot > 0.f) ? 1.0f / l_tot : 0.f;   // +Source.swift:165-171 (0: a row whose every block is masked out)
  store_rows<DP>(o, a.op[SLOT_O], head, batch, row, R, hi, D, inv);
  if (hi == 0 && row < R)  // L = m + log2(l), +Caching.swift:373-377
    reinterpret_cast<float *>(operand_base(a.op[SLOT_L], head, batch))[row] = m + log2f(l_tot);
}

// ----------------------------------------------------------------------------------------------
// backward dQ: D = rowsum(dO*O)/sqrt(D); dQ = sum_c dS K                (+Source.swift:202-242)
// same grid; one workgroup per compute unit (Q and dO fragments, the dQ accumulators: 192 registers before any buffer)
// ----------------------------------------------------------------------------------------------
// PROF (developer builds, timing only): phase clocks of wave 0 -- per tile, in shader clocks -- replace dQ[row r0][0..4] (tools/f32_perf.py --prof)
template <int DP, bool PROF = false>
__global__ __launch_bounds__(256) void attn_f32_dq(const KernelArgs a, const Fwd16Grid grid) {
  typedef Geo<DP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &rblk, &head, &batch);
  if (a.causal) rblk = grid.rowBlocks - 1 - rblk;   // later row blocks traverse more keys: start them first
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * ROWS;
  if (r0 >= R) return;
  const int64_t row = r0 + wave * 32 + q;
  const float scale = a.scale;
  const bool causal = a.causal != 0;

  const __amdgpu_buffer_rsrc_t qres = rows_resource(a.op[SLOT_Q], head, batch, R);
  const __amdgpu_buffer_rsrc_t gres = rows_resource(a.op[SLOT_dO], head, batch, R);
  const __amdgpu_buffer_rsrc_t ores = rows_resource(a.op[SLOT_O], head, batch, R);
  const char *kbase = operand_base(a.op[SLOT_K], head, batch), *vbase = operand_base(a.op[SLOT_V], head, batch);
  const uint32_t ldk = (uint32_t)a.op[SLOT_K].ld, ldv = (uint32_t)a.op[SLOT_V].ld;
  float qf[DP / 2], gf[DP / 2];
  load_fragments<DP>(qf, qres, (uint32_t)row * (uint32_t)a.op[SLOT_Q].ld * 4u, row < R, hi, D);
  load_fragments<DP>(gf, gres, (uint32_t)row * (uint32_t)a.op[SLOT_dO].ld * 4u, row < R, hi, D);
  // computeD, +Softmax.swift:32-221: D = (sum_d dO*O) * 1/sqrt(D); the two half-waves split the head dimension
  float dsum = 0.f;
  {
    float of[DP / 2];
    load_fragments<DP>(of, ores, (uint32_t)row * (uint32_t)a.op[SLOT_O].ld * 4u, row < R, hi, D);
#pragma unroll
    for (int t = 0; t < DP / 2; ++t) dsum += of[t] * gf[t];
    dsum = half_add(dsum);
  }
  float Lrow = 0.f;
  if (row < R) Lrow = reinterpret_cast<const float *>(operand_base(a.op[SLOT_L], head, batch))[row];
  pin_fragments<DP / 2>(qf, a.scale2);   // Q pre-multiplied by log2(e) / sqrt(D): S' = S * scale2
  pin_fragments<DP / 2>(gf, 1.f);
  // the accumulators of the first products start at -L and -sum(dO o O): S' - L and dP - D / scale leave the matrix
  // instructions; dS = P (dP - D / scale), the factor 1 / sqrt(D) is applied to dQ once at the end (+Softmax.swift:409-427)
  const f32x16 negL = splat16(-Lrow), negD = splat16(-dsum);

  f32x16 acc[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db) mfma_zero(acc[db]);

  const int coff = causal_offset(R, C);
  const int Cend = causal ? (int)min((int64_t)C, min((int64_t)R, r0 + ROWS) + coff) : C;
  const int nt = (Cend + BT - 1) / BT;
  Stager<DP> ks, vs;
  ks.init(wave, lane, ldk, D);
  vs.init(wave, lane, ldv, D);
  // three stages of (K, V) tiles: tile j in stage j % 3.  The barrier of a tile sits in its MIDDLE (between the first products
  // and the softmax arithmetic): behind it tile j + 1 is complete for everybody and everybody has left tile j - 1, whose stage
  // receives tile j + 2 -- so no LDS read waits behind a barrier: the first reads of tile j + 1 go out during the last steps of
  // tile j's second product.  (stage 2 lies beyond the 64 KiB an immediate offset reaches: a second set of addresses)
  constexpr int STAGEB = 2 * G::TILE;
  Addresses<DP> ad, ad2;
  ad.init(lds_addr(smem), q, hi);
  ad2.init(lds_addr(smem) + 2 * STAGEB, q, hi);
  const int limit = (int)row + coff;
  const int wavelimit = (int)r0 + wave * 32 + coff;
  ks.issue(tile_resource(kbase, ldk, C, 0), smem, wave);
  vs.issue(tile_resource(vbase, ldv, C, 0), smem + G::TILE, wave);
  if (nt > 1) {
    ks.issue(tile_resource(kbase, ldk, C, BT), smem + STAGEB, wave);
    vs.issue(tile_resource(vbase, ldv, C, BT), smem + STAGEB + G::TILE, wave);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  FirstRings fr;
  first_prefetch_pair<DP, 0, G::TILE>(fr, ad);
  uint64_t clk[5] = {0, 0, 0, 0, 0};
  auto stamp = [&](int k, uint64_t &t) {
    if constexpr (PROF) {
      const uint64_t now = __builtin_amdgcn_s_memtime();
      clk[k] += now - t;
      t = now;
    }
  };
  auto tile = [&](auto STAGE_, int j) {
    constexpr int STAGE = decltype(STAGE_)::value, NEXT1 = (STAGE + 1) % 3, NEXT2 = (STAGE + 2) % 3;
    constexpr int KOFF = STAGE == 2 ? 0 : STAGE * STAGEB, VOFF = KOFF + G::TILE;
    constexpr int K1OFF = NEXT1 == 2 ? 0 : NEXT1 * STAGEB, V1OFF = K1OFF + G::TILE;
    const Addresses<DP> &here = STAGE == 2 ? ad2 : ad, &next = NEXT1 == 2 ? ad2 : ad;
    const int c0 = j * BT;
    // S'^T = K Q'^T - L, dP^T = V dO^T - D / scale
    f32x16 s, dp;
    uint64_t t = PROF ? __builtin_amdgcn_s_memtime() : 0;
    first_body_pair<DP, KOFF, VOFF>(s, negL, qf, dp, negD, gf, fr, here);
    SecondRing<DP> kring;
    second_prefetch<DP, KOFF>(kring, here);
    stamp(0, t);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of tile j + 1 have landed
    __syncthreads();
    stamp(1, t);
    stamp(2, t);
    mfma_fence(s, dp);
    // P = exp2(S' - L); dS = P * (dP - D / scale).  Padded columns: K, V rows are zero, so dS * K contributes nothing (as in
    // the reference, where the zero padding comes from the async copy, +Accumulate.swift:330-346)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]) * dp[r];
    if (c0 + BT > C) {   // ragged last tile: a padded column has S' = -L, and a row with L < -128 (base 2) would make exp2 = inf and
#pragma unroll           // inf x 0 = NaN in the whole dQ row -- dS of a column beyond C is zero by definition (wave-uniform branch)
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) >= C) s[r] = 0.f;
    }
    if (causal && c0 + BT - 1 > wavelimit) {   // masked column: P = 0, hence dS = 0
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) > limit) s[r] = 0.f;
    }
    valu_fence(s);
    stamp(3, t);
    // dQ^T += K^T dS^T, key index permuted as in forward; its last steps send the first reads of tile j + 1 ahead
    // ... and tile j + 2 is requested piece by piece between its first steps
    const bool more = j + 2 < nt;
    const __amdgpu_buffer_rsrc_t kres2 = tile_resource(kbase, ldk, C, c0 + 2 * BT), vres2 = tile_resource(vbase, ldv, C, c0 + 2 * BT);
    second_product<DP, KOFF, 4, 2 * FirstRings::RING>(acc, kring, here, s, [&]() { first_prefetch_pair<DP, K1OFF, V1OFF>(fr, next); }, [&](auto T_) {
      constexpr int T = decltype(T_)::value;
      if constexpr (T < 2 * G::NI) {
        if (more) {
          if constexpr (T % 2 == 0) ks.template piece<T / 2>(kres2, smem + NEXT2 * STAGEB, wave);
          else vs.template piece<T / 2>(vres2, smem + NEXT2 * STAGEB + G::TILE, wave);
        }
      }
    });
    stamp(4, t);
  };
  for (int j = 0; j < nt; j += 3) {
    tile(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nt) tile(std::integral_constant<int, 1>{}, j + 1);
    if (j + 2 < nt) tile(std::integral_constant<int, 2>{}, j + 2);
  }
  // (the last tile sent reads ahead for a tile that does not exist)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr.r0[0]), "+v"(fr.r0[1]), "+v"(fr.r0[2]), "+v"(fr.r1[0]), "+v"(fr.r1[1]), "+v"(fr.r1[2]));
  mfma_fence_out<G::NDB>(acc);
  store_rows<DP>(acc, a.op[SLOT_dQ], head, batch, row, R, hi, D, scale);
  if (hi == 0 && row < R)   // +Caching.swift:381-413
    reinterpret_cast<float *>(operand_base(a.op[SLOT_D], head, batch))[row] = dsum * scale;
  if constexpr (PROF) {
    if (tid == 0) {
      float *dst = reinterpret_cast<float *>(operand_base(a.op[SLOT_dQ], head, batch)) + r0 * a.op[SLOT_dQ].ld;
      for (int k = 0; k < 5; ++k) dst[k] = (float)clk[k] / (float)nt;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// backward dK/dV: dV = sum_r P^T dO ; dK = sum_r dS^T Q, parallel over columns     (+Source.swift:244-293)
// grid: 1-D, ceil(C / 128) x heads x batches; one workgroup per compute unit
// ----------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void attn_f32_dkv(const KernelArgs a, const Fwd16Grid grid) {
  typedef Geo<DP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, kc = lane & 31, hi = lane >> 5;
  uint32_t cblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &cblk, &head, &batch);
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t c0 = (int64_t)cblk * ROWS;
  if (c0 >= C) return;
  const int64_t col = c0 + wave * 32 + kc;
  const bool causal = a.causal != 0;

  const __amdgpu_buffer_rsrc_t kres = rows_resource(a.op[SLOT_K], head, batch, C);
  const __amdgpu_buffer_rsrc_t vres = rows_resource(a.op[SLOT_V], head, batch, C);
  const char *qbase = operand_base(a.op[SLOT_Q], head, batch), *gbase = operand_base(a.op[SLOT_dO], head, batch);
  const char *lbase = operand_base(a.op[SLOT_L], head, batch), *dbase = operand_base(a.op[SLOT_D], head, batch);
  const uint32_t ldq = (uint32_t)a.op[SLOT_Q].ld, ldg = (uint32_t)a.op[SLOT_dO].ld;
  // K pre-multiplied by -log2(e) / sqrt(D) and V by -1 / sqrt(D): with accumulators that start at L and D (the slices of the
  // tile's rows, read from LDS straight into the accumulator registers) the first products deliver L - S' and D - dP / sqrt(D):
  // P = exp2(-(L - S')), dS = -P (D - dP / sqrt(D))       (+Softmax.swift:409-427); dK accumulates +P (...) and is negated when stored
  float kf[DP / 2], vf[DP / 2];
  load_fragments<DP>(kf, kres, (uint32_t)col * (uint32_t)a.op[SLOT_K].ld * 4u, col < C, hi, D);
  load_fragments<DP>(vf, vres, (uint32_t)col * (uint32_t)a.op[SLOT_V].ld * 4u, col < C, hi, D);
  pin_fragments<DP / 2>(kf, -a.scale2);
  pin_fragments<DP / 2>(vf, -a.scale);

  f32x16 dk[G::NDB], dv[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db) { mfma_zero(dk[db]); mfma_zero(dv[db]); }

  // causal extension: rows above the workgroup's first column minus the offset see none of its columns
  const int coff = causal_offset(R, C);
  const int rstart = causal ? (int)(max((int64_t)0, c0 - coff) / BT) * BT : 0;
  const int nt = (R - rstart + BT - 1) / BT;
  Stager<DP> qs, gs;
  qs.init(wave, lane, ldq, D);
  gs.init(wave, lane, ldg, D);
  // three stages of (Q, dO) tiles and (L, D) slices, the barrier in the middle of a tile, reads and staging sent ahead from the
  // second products: as in attn_f32_dq
  constexpr int STAGEB = 2 * G::TILE;
  Addresses<DP> ad, ad2;
  ad.init(lds_addr(smem), kc, hi);
  ad2.init(lds_addr(smem) + 2 * STAGEB, kc, hi);
  // L and D slices along the traversal dimension (+Softmax.swift:356-381, :472-503): 32 floats each per tile, staged like the
  // tiles (wave 0: L, wave 1: D; one dword per lane, the upper 32 lanes' rows belong to the next tile and are not read)
  char *ldst = smem + 3 * STAGEB;
  const uint32_t ldread = lds_addr(ldst) + hi * 16;
  auto issue_ld = [&](int stage, int row0) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int left = R > row0 ? R - row0 : 0;
    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(lbase) + (int64_t)row0 * 4, 0, (uint32_t)left * 4u, 0x00020000),
                                               (lds_ptr)(ldst + stage * 512), 4, (uint32_t)lane * 4u, 0, 0, 0);
    if (wave == 1)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(dbase) + (int64_t)row0 * 4, 0, (uint32_t)left * 4u, 0x00020000),
                                               (lds_ptr)(ldst + stage * 512 + 256), 4, (uint32_t)lane * 4u, 0, 0, 0);
#endif
  };
  // the wave's largest column: rows whose limit lies below it mask (causal)
  const int wavecol = (int)c0 + wave * 32 + 31, icol = (int)col;
  qs.issue(tile_resource(qbase, ldq, R, rstart), smem, wave);
  gs.issue(tile_resource(gbase, ldg, R, rstart), smem + G::TILE, wave);
  issue_ld(0, rstart);
  if (nt > 1) {
    qs.issue(tile_resource(qbase, ldq, R, rstart + BT), smem + STAGEB, wave);
    gs.issue(tile_resource(gbase, ldg, R, rstart + BT), smem + STAGEB + G::TILE, wave);
    issue_ld(1, rstart + BT);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  FirstRings fr;
  first_prefetch_pair<DP, 0, G::TILE>(fr, ad);
  // rows rr0 + crow(r, hi) of L and D: four reads of four each, into the registers the first products accumulate in
  f32x4 lv[4], dvv[4];
  static_for<4>([&](auto g_) {
    constexpr int g = decltype(g_)::value;
    lv[g] = rd128<32 * g>(ldread);
    dvv[g] = rd128<256 + 32 * g>(ldread);
  });
  auto tile = [&](auto STAGE_, int j) {
    constexpr int STAGE = decltype(STAGE_)::value, NEXT1 = (STAGE + 1) % 3, NEXT2 = (STAGE + 2) % 3;
    constexpr int QOFF = STAGE == 2 ? 0 : STAGE * STAGEB, GOFF = QOFF + G::TILE;
    constexpr int Q1OFF = NEXT1 == 2 ? 0 : NEXT1 * STAGEB, G1OFF = Q1OFF + G::TILE;
    const Addresses<DP> &here = STAGE == 2 ? ad2 : ad, &next = NEXT1 == 2 ? ad2 : ad;
    const int rr0 = rstart + j * BT;
    // L - S' (S = Q K^T, not swapped: lane holds key `col`, rows rr0 + crow(r, hi)); D - dP / sqrt(D) (dP = dO V^T): the slices
    // (the youngest reads in flight) ARE the accumulators' start values
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lv[0]), "+v"(lv[1]), "+v"(lv[2]), "+v"(lv[3]), "+v"(dvv[0]), "+v"(dvv[1]), "+v"(dvv[2]), "+v"(dvv[3]));
    f32x16 s = __builtin_shufflevector(__builtin_shufflevector(lv[0], lv[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(lv[2], lv[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                       0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    f32x16 dp = __builtin_shufflevector(__builtin_shufflevector(dvv[0], dvv[1], 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(dvv[2], dvv[3], 0, 1, 2, 3, 4, 5, 6, 7),
                                        0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    first_body_pair<DP, QOFF, GOFF, false>(s, s, kf, dp, dp, vf, fr, here);
    SecondRing<DP> gring, qring;
    second_prefetch_pair<DP, GOFF, QOFF, 0>(gring, qring, here);
    second_prefetch_pair<DP, GOFF, QOFF, 1>(gring, qring, here);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of tile j + 1 have landed
    __syncthreads();
    mfma_fence(s, dp);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(-s[r]);
      s[r] = p * dp[r];   // = -dS: the sign is applied to dK once at the end
      dp[r] = p;
    }
    if (causal && wavecol > rr0 + coff) {   // masked: P = 0
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (icol > rr0 + crow(r, hi) + coff) { s[r] = 0.f; dp[r] = 0.f; }
    }
    valu_fence(s, dp);
    // dV^T += dO^T P ; dK^T += Q^T dS   (row index permuted; padded rows of Q / dO are zero); tile j + 2 is requested between
    // the first steps, the last steps send the first reads of tile j + 1 ahead
    const bool more = j + 2 < nt;
    const __amdgpu_buffer_rsrc_t qres2 = tile_resource(qbase, ldq, R, rr0 + 2 * BT), gres2 = tile_resource(gbase, ldg, R, rr0 + 2 * BT);
    second_product_pair<DP, GOFF, QOFF, 2 * FirstRings::RING>(dv, dp, dk, s, gring, qring, here, [&]() { first_prefetch_pair<DP, Q1OFF, G1OFF>(fr, next); }, [&](auto T_) {
      constexpr int T = decltype(T_)::value;
      if constexpr (T < 2 * G::NI) {
        if (more) {
          if constexpr (T % 2 == 0) qs.template piece<T / 2>(qres2, smem + NEXT2 * STAGEB, wave);
          else gs.template piece<T / 2>(gres2, smem + NEXT2 * STAGEB + G::TILE, wave);
        }
      } else if constexpr (T == 2 * G::NI) {
        if (more) issue_ld(NEXT2, rr0 + 2 * BT);
      }
    });
    // the slices of tile j + 1 (behind the reads sent ahead: fourteen in flight)
    static_for<4>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      lv[g] = rd128<NEXT1 * 512 + 32 * g>(ldread);
      dvv[g] = rd128<NEXT1 * 512 + 256 + 32 * g>(ldread);
    });
  };
  for (int j = 0; j < nt; j += 3) {
    tile(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nt) tile(std::integral_constant<int, 1>{}, j + 1);
    if (j + 2 < nt) tile(std::integral_constant<int, 2>{}, j + 2);
  }
  // (the last tile sent reads ahead for a tile that does not exist)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr.r0[0]), "+v"(fr.r0[1]), "+v"(fr.r0[2]), "+v"(fr.r1[0]), "+v"(fr.r1[1]), "+v"(fr.r1[2]), "+v"(lv[0]), "+v"(lv[1]),
               "+v"(lv[2]), "+v"(lv[3]), "+v"(dvv[0]), "+v"(dvv[1]), "+v"(dvv[2]), "+v"(dvv[3]));
  mfma_fence_out<G::NDB>(dv);
  mfma_fence_out<G::NDB>(dk);
  store_rows<DP>(dv, a.op[SLOT_dV], head, batch, col, C, hi, D, 1.f);
  store_rows<DP>(dk, a.op[SLOT_dK], head, batch, col, C, hi, D, -1.f);
}

// ---- host side: does a launch of the general kernel's variant go to these kernels?  (the general kernels' launchers ask)
inline bool rows_ok(const OperandView &v, uint32_t rows) {
  return v.ptr && v.precision == PREC_FP32 && !v.transposed &&
         ((reinterpret_cast<uintptr_t>(v.ptr) | (uint64_t)v.ld * 4 | (uint64_t)v.headStride * 4 | (uint64_t)v.batchStride * 4) & 15) == 0 &&
         (uint64_t)rows * (uint64_t)v.ld * 4 < 0xFFFFFF00ull;   // one bounds-checked resource per (head, batch)
}
inline bool vector_ok(const OperandView &v) { return v.ptr && v.precision == PREC_FP32; }
// type: 0 forward, 1 backwardQuery, 2 backwardKeyValue (mfa.h); DP: the variant's head block
inline bool serves(int type, int DP, const KernelArgs &a) {
  if ((DP != 64 && DP != 128) || a.mask || (a.D & 3) || a.D > (uint32_t)DP || a.R == 0 || a.C == 0) return false;
  if (!rows_ok(a.op[SLOT_Q], a.R) || !rows_ok(a.op[SLOT_K], a.C) || !rows_ok(a.op[SLOT_V], a.C) || !vector_ok(a.op[SLOT_L])) return false;
  if (type == 0) return rows_ok(a.op[SLOT_O], a.R);
  if (!rows_ok(a.op[SLOT_dO], a.R) || !vector_ok(a.op[SLOT_D])) return false;
  if (type == 1) return rows_ok(a.op[SLOT_O], a.R) && rows_ok(a.op[SLOT_dQ], a.R);
  return rows_ok(a.op[SLOT_dK], a.C) && rows_ok(a.op[SLOT_dV], a.C);
}

}  // namespace f32k
}  // namespace mfa
