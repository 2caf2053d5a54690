// attn_fwd16_common.h -- what the 16-bit matrix-core kernels share (fragment types, the 16-bit store rounding, whole-row stores of a
// block laid out in LDS, the K image swizzle, the XCD-aware block decode, half-wave exchanges).  Until round 6 these lived in
// attn_fwd16.h / attn_fwd16_v2.h beside the two superseded forward kernels, which therefore stayed on the product include path; the
// kernels are now developer-library sources (dev/attn_fwd16.h, dev/attn_fwd16_v2.h: read those headers for the fragment maps and LDS
// images every later kernel inherited).
#pragma once
#include "attn_common.h"

namespace mfa {

template <typename T> struct Frag16;
template <> struct Frag16<__bf16> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag16<_Float16> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// two fp32 values -> one dword of the 16-bit storage type, with the reference's store rounding: BF16 by
// truncation (GEMMHeaders.swift:461-471, +Caching.swift:395-401), FP16 round-to-nearest
template <typename T> __device__ __forceinline__ uint32_t pack16(float a, float b) {
  if constexpr (__is_same(T, __bf16)) {
    return (__builtin_bit_cast(uint32_t, a) >> 16) | (__builtin_bit_cast(uint32_t, b) & 0xFFFF0000u);
  } else {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, h);
  }
}

// Whole-row store of a 32-row x D block that a wave has laid out in LDS ([32][D + 4] floats) to a row-major
// operand kept in FP32 or -- fused output cast, SURVEY.md section 8f rank 2 -- in the 16-bit type T.
// `bound` = number of valid rows of the operand (per batch entry), `Dr` = real head dimension.
template <typename T, int D>
__device__ __forceinline__ void store_block_rows(const float *Os, char *base, int prec, uint32_t ld, int64_t r0, int64_t bound,
                                                 int Dr, int lane, float scale = 1.0f) {
  constexpr int OLD = D + 4, CPRO = D / 4;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const uint32_t esz = prec == PREC_FP32 ? 4u : 2u;
  const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(base, 0, (uint32_t)bound * ld * esz, 0x00020000);
#pragma unroll
  for (int i = 0; i < 32 * CPRO / 64; ++i) {
    const int id = lane + i * 64;
    const int rr = id / CPRO, c = id % CPRO;
    float4 val = *reinterpret_cast<const float4 *>(Os + rr * OLD + c * 4);
    val.x *= scale; val.y *= scale; val.z *= scale; val.w *= scale;
    const bool ok = r0 + rr < bound && c * 4 < Dr;
    if (prec == PREC_FP32) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), res, ok ? (uint32_t)(r0 + rr) * ld * 4 + c * 16 : OOB, 0, 0);
    } else {
      const u32x2 h = {pack16<T>(val.x, val.y), pack16<T>(val.z, val.w)};
      __builtin_amdgcn_raw_buffer_store_b64(h, res, ok ? (uint32_t)(r0 + rr) * ld * 2 + c * 8 : OOB, 0, 0);
    }
  }
}

// 16-byte chunk index swizzle of the row-major K image (ds_read_b128 is conflict-free when the 16
// rows of a lane group land on 16 distinct 16-B slots of the 256-B bank row)
template <int D> __device__ __forceinline__ int kswz(int row, int chunk) {
  if constexpr (D >= 128 && (D & (D - 1)) == 0) return chunk ^ (row & 15);
  else if constexpr (D == 64) return chunk ^ ((row >> 1) & 7);
  else return chunk ^ ((row >> 2) & 3);  // D == 32, and the buckets whose row is not a power of two (96, 160, 192 elements:
                                         // the XOR must stay inside the row, chunks per row are a multiple of four)
}

// the row-dependent XOR mask of kswz (kswz(row, c) == c ^ kswz_mask(row)); the swizzle is an involution
template <int D> __device__ __forceinline__ int kswz_mask(int row) { return kswz<D>(row, 0); }

template <int D> constexpr int fwd16_lds_bytes() { return 2 /*buffers*/ * 2 /*K,V*/ * 64 * D * 2; }

// grid: 1-D, (row blocks) x heads x batches flattened; see fwd16_decode_block for the XCD-aware order
struct Fwd16Grid {
  uint32_t rowBlocks, heads, batches;
  // column-parallel ("split-KV") launches only: the key range is cut into `splits` pieces, each
  // workgroup writes un-normalised partial results into the caller's workspace
  //   wsO  [splits][heads*batches][R][D] fp32,  wsML [splits][heads*batches][R][2] = (m, l)
  uint32_t splits;
  float *wsO;
  float *wsML;
};

// (per-lane form: `bid` may differ from lane to lane -- the persistent forward kernel builds its block table with it)
__device__ __forceinline__ void fwd16_decode_block_lane(const Fwd16Grid &g, uint32_t bid, uint32_t *rb,
                                                        uint32_t *head, uint32_t *batch) {
  // Hardware places workgroup b on XCD b % 8 (observed; used for speed only).  All row blocks of
  // one (head, batch) share K and V, so we give each XCD whole heads: its private 4 MiB L2 then
  // holds the K/V of the few heads it is working on.
  // Within an XCD the heads come in PAIRS whose blocks alternate: (block 0, head a), (block 0, head b), (block 1, head a), ...
  // The L2 working set is the same two heads (32 compute units = two heads x 16 blocks of 256 rows at N = 4096), but a causal
  // launch -- whose blocks shrink (or grow) along the block index -- is dealt out longest-first ACROSS the pair: with one head
  // after the other, the in-order dispatcher left the compute units 15 % apart at the end (simulated: 331 against 287 units
  // of time for 8 heads x 16 blocks on 32 units, profiles/r04_causal_dispatch_order.txt); dense launches do not care.
  const uint32_t nh = g.heads * g.batches;
  uint32_t hb, r;
  if ((nh & 7u) == 0) {
    const uint32_t xcd = bid & 7u, slot = bid >> 3;
    const uint32_t hpx = nh >> 3;                          // heads of this XCD
    const uint32_t grp = slot / (2u * g.rowBlocks), t = slot % (2u * g.rowBlocks);
    uint32_t hx;
    if (2u * grp + 1u < hpx) { hx = 2u * grp + (t & 1u); r = t >> 1; }
    else { hx = 2u * grp; r = t; }                         // (an odd head count leaves the last head on its own)
    hb = hx * 8u + xcd;
  } else {
    hb = bid / g.rowBlocks;
    r = bid % g.rowBlocks;
  }
  *rb = r;
  *head = hb % g.heads;
  *batch = hb / g.heads;
}

// the block of a WORKGROUP (`bid` wave-uniform): the integer divisions run on the vector ALU, and hipcc does not move their
// results back to scalar registers by itself when an asm statement asks for an "s" operand derived from them
__device__ __forceinline__ void fwd16_decode_block(const Fwd16Grid &g, uint32_t bid, uint32_t *rb,
                                                   uint32_t *head, uint32_t *batch) {
  uint32_t r, h, b;
  fwd16_decode_block_lane(g, bid, &r, &h, &b);
  *rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
  *head = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
  *batch = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
}


template <int D, int NW, int RB, int RING = 3, int KPADB = 0> constexpr int fwd16v2_lds_bytes() {
  constexpr int ring = RING * (2 * 64 * D * 2 + 64 * KPADB);
  constexpr int epi = NW * RB * 32 * (D + 4) * 4;
  return ring > epi ? ring : epi;
}

// Exchange between the two half-waves (lane l <-> lane l^32) with v_permlane32_swap: lanes 32-63
// of the first operand trade places with lanes 0-31 of the second, so {a, b} = {own, partner} in
// some order on every lane.  Inline asm on purpose: with hipcc (ROCm 7.2) the second result of
// __builtin_amdgcn_permlane32_swap reads the FIRST operand's register (both extracts alias).  The
// s_nop covers the VALU-write -> permlane-read hazard inside the asm string.
__device__ __forceinline__ void half_swap(float x, float *a, float *b) {
  uint32_t b0 = __builtin_bit_cast(uint32_t, x), b1 = b0;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(b0), "+v"(b1));
  *a = __builtin_bit_cast(float, b0);
  *b = __builtin_bit_cast(float, b1);
}
__device__ __forceinline__ float half_swap_max(float x) {
  float a, b;
  half_swap(x, &a, &b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float half_swap_add(float x) {
  float a, b;
  half_swap(x, &a, &b);
  return a + b;
}

} // namespace mfa
