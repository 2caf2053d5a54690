// attn_common.h -- device-side argument block and helpers shared by every attention kernel.
//
// gfx950 (CDNA4) only.  MFMA fragment maps used throughout (wave64, lane l):
//   v_mfma_f32_32x32x2_f32   A[i=l&31][k=l>>5]            B[k=l>>5][j=l&31]
//   v_mfma_f32_32x32x16_bf16 A[i=l&31][k=8*(l>>5)..+7]    B[k=8*(l>>5)..+7][j=l&31]
//   C/D (both):              C[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31], r in [0,16)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfa {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

enum : int { PREC_FP32 = 0, PREC_FP16 = 1, PREC_BF16 = 2 };

// One operand as bound at an AttentionOperand.bufferBinding slot
// (reference: Sources/FlashAttention/Attention/AttentionOperand.swift:52-71).
struct OperandView {
  void *ptr;
  int64_t ld;          // leading dimension, elements
  int64_t headStride;  // elements
  int64_t batchStride; // elements
  int32_t precision;   // PREC_*
  int32_t transposed;  // 0: [seq][D] row-major, 1: [D][seq]
};

// Slot order: Q0 K1 V2 O3 L4 D5 dO6 dV7 dK8 dQ9.
enum : int { SLOT_Q = 0, SLOT_K, SLOT_V, SLOT_O, SLOT_L, SLOT_D, SLOT_dO, SLOT_dV, SLOT_dK, SLOT_dQ, SLOT_COUNT };

struct KernelArgs {
  OperandView op[SLOT_COUNT];
  uint32_t R, C, D;
  float scale;   // 1/sqrt(D)                 (+Softmax.swift:17-26, derivative: true)
  float scale2;  // log2(e)/sqrt(D)           (+Softmax.swift:17-26, derivative: false)
  // extension (not in the reference): causal mask, row r sees column c iff c <= r + (C - R)
  int32_t causal;
  // extension: per-batch-entry sequence lengths (device arrays of `batches` entries, or null): entry b
  // uses the first rowLen[b] rows and colLen[b] columns of its R x C problem; the rest is padding
  const uint32_t *rowLen, *colLen;
  // extension: block mask, one bit per (256 rows x 128 columns) block, row-major bitmap with `maskWords`
  // 32-bit words per row block; strides in words (0 = one mask shared by all heads / batch entries)
  const uint32_t *mask;
  uint32_t maskWords;
  int64_t maskHeadStride, maskBatchStride;
};
constexpr int MASK_BLOCK_ROWS = 256, MASK_BLOCK_COLUMNS = 128;

// row index inside a 32x32 MFMA C/D tile held by (register r, half hi)
__device__ __forceinline__ constexpr int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
// truncation: what the reference does when it stores BF16 (GEMMHeaders.swift:461-471,
// AttentionKernel+Caching.swift:395-401)
__device__ __forceinline__ uint16_t f32_to_bf16_trunc(float f) {
  return (uint16_t)(__builtin_bit_cast(uint32_t, f) >> 16);
}

__device__ __forceinline__ float load_elem(const void *p, int64_t idx, int prec) {
  if (prec == PREC_FP32) return ((const float *)p)[idx];
  if (prec == PREC_FP16) return (float)((const _Float16 *)p)[idx];
  return bf16_bits_to_f32(((const uint16_t *)p)[idx]);
}
__device__ __forceinline__ void store_elem(void *p, int64_t idx, int prec, float v) {
  if (prec == PREC_FP32) ((float *)p)[idx] = v;
  else if (prec == PREC_FP16) ((_Float16 *)p)[idx] = (_Float16)v;
  else ((uint16_t *)p)[idx] = f32_to_bf16_trunc(v);
}

__device__ __forceinline__ int elem_size(int prec) { return prec == PREC_FP32 ? 4 : 2; }

// base pointer of (head, batch) for an operand
__device__ __forceinline__ char *operand_base(const OperandView &v, uint32_t head, uint32_t batch) {
  const int64_t off = (int64_t)head * v.headStride + (int64_t)batch * v.batchStride;
  return (char *)v.ptr + off * elem_size(v.precision);
}

// the problem size of one batch entry (variable-length extension); rows / columns beyond it are never
// loaded (bounds-checked resources shrink with it) nor stored
__device__ __forceinline__ void batch_lengths(const KernelArgs &a, uint32_t batch, int &R, int &C) {
  if (a.rowLen) R = min(R, (int)a.rowLen[batch]);
  if (a.colLen) C = min(C, (int)a.colLen[batch]);
}

// causal launches: row r sees column c iff c <= r + causal_offset(R, C).  The host validates column >= row; per-batch lengths
// (device arrays, resolved by batch_lengths above) can still make an entry's C smaller than its R -- the offset is clamped at 0
// there (row r then sees columns <= min(r, C - 1)), so that no row is left without a visible key (include/mfa.h, rowLengths).
__device__ __forceinline__ int causal_offset(int R, int C) { return max(C - R, 0); }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// mask value for padded attention-matrix columns, AttentionKernel+Softmax.swift:242-243:
// (0.875 / log2(e)) * -max(float)
__device__ __forceinline__ constexpr float mask_value() { return -(0.875f / 1.44269504089f) * 3.402823466e+38f; }

} // namespace mfa
