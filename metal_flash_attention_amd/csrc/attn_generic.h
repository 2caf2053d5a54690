// attn_generic.h -- the three attention kernels in their GENERAL form for gfx950:
//   any R, C, any D <= DP, any of {FP32, FP16, BF16} per operand in memory, any transpose state,
//   any leading dimension.  All arithmetic is fp32: operands are converted while they are staged
//   into LDS and both contractions of every kernel run on v_mfma_f32_32x32x2_f32 (exact fp32 fma
//   chains, 157 TF peak).  This is the FP32 production path (BASELINE config 3) and the
//   correctness path for every shape / layout / dtype mix the 16-bit fast kernels do not take.
//
// What the reference does, restated for CDNA4 (reference: Sources/FlashAttention/Attention/
// AttentionKernel/):
//   * loopForward / loopBackwardQuery / loopBackwardKeyValue (+Source.swift:158-293) become the
//     three __global__ templates below; one wave owns 32 rows (fwd, dQ) or 32 columns (dK/dV) --
//     the MFMA 32x32 tile -- instead of one simdgroup owning 8.
//   * "cached" operands (+Caching.swift:18-281) live in VGPRs for the whole kernel (CACHE=true);
//     with CACHE=false the left-hand operands stay in LDS and are re-read every traversal step
//     (the role device memory plays in +OuterProduct.swift:133-171).  Accumulators always stay in
//     registers: 512 VGPR+AGPR per lane hold 32 x 256 fp32 twice over.
//   * zero padding of ragged tiles, which the reference gets from simdgroup_event::async_copy
//     (GEMMHeaders.swift:166-193), is done by the staging loop; stores are guarded.
//   * the contraction index of the second GEMM of each pair is PERMUTED so that the C/D layout of
//     the first MFMA is directly the B operand of the second (no cross-lane traffic): in step t
//     lanes 0-31 contribute key/row crow(t,0) and lanes 32-63 key/row crow(t,1).
#pragma once
#include "attn_common.h"

namespace mfa {

// ----------------------------------------------------------------------------------------------
// Staging: global (any dtype / layout) -> LDS fp32 tile [ROWS][DP+1], zero padded.
// ----------------------------------------------------------------------------------------------
// float4 chunk e of a [ROWS][DP] tile -> (row, first column).  Eight consecutive lanes take one
// 128-byte run of a row (coalesced), the next eight lanes the same run of the NEXT row: with the odd
// row stride DP+1 the 32 lanes of a half-wave then hit 32 distinct LDS banks when they scatter their
// four components with ds_write_b32 (lanes four floats apart in ONE row would be a 4-way conflict).
template <int ROWS, int DP>
__device__ __forceinline__ void tile_chunk(int e, int *n, int *d) {
  if constexpr (ROWS % 4 == 0) {
    const int g = e >> 3, c8 = e & 7;          // g: 128-byte run index, c8: float4 within the run
    const int quad = g & 3, rest = g >> 2;     // four runs of four consecutive rows share a d-block
    constexpr int RUNS = DP / 32;              // runs per row
    const int dblk = rest % RUNS, rowq = rest / RUNS;
    *n = rowq * 4 + quad;
    *d = dblk * 32 + c8 * 4;
  } else {
    *n = e / (DP / 4);
    *d = (e % (DP / 4)) * 4;
  }
}

// block mask (extension): bit (row block of 256 rows, column block of 128 columns), see attn_common.h.  The
// general kernels skip inactive blocks tile by tile (and then stage synchronously: no register prefetch)
__device__ __forceinline__ const uint32_t *mask_base(const KernelArgs &a, uint32_t head, uint32_t batch) {
  return a.mask ? a.mask + (int64_t)head * a.maskHeadStride + (int64_t)batch * a.maskBatchStride : nullptr;
}
__device__ __forceinline__ bool mask_bit(const uint32_t *m, uint32_t words, int64_t row, int64_t col) {
  const uint32_t cb = (uint32_t)(col / MASK_BLOCK_COLUMNS);
  return ((m[(uint64_t)(row / MASK_BLOCK_ROWS) * words + (cb >> 5)] >> (cb & 31)) & 1u) != 0;
}

// Register-staged variant of the fp32 fast path (prefetch: issue the loads of tile j+1 before the
// arithmetic of tile j, write them to LDS after it).
template <int ROWS, int DP, int NT> struct TileRegsF32 {
  static constexpr int N = (ROWS * DP / 4 + NT - 1) / NT;
  float4 v[N];
};
__device__ __forceinline__ bool f32_fast_path(const OperandView &v, const char *base, int D) {
  return !v.transposed && v.precision == PREC_FP32 && ((reinterpret_cast<uintptr_t>(base) & 15) == 0) &&
         (D & 3) == 0 && (v.ld & 3) == 0;
}
template <int ROWS, int DP, int NT>
__device__ __forceinline__ void tile_load_f32(TileRegsF32<ROWS, DP, NT> &r, const OperandView &v, const char *base,
                                              int64_t n0, int64_t N, int D, int tid) {
  constexpr int V4 = DP / 4;
#pragma unroll
  for (int i = 0; i < TileRegsF32<ROWS, DP, NT>::N; ++i) {
    const int e = tid + i * NT;
    int n, d;
    tile_chunk<ROWS, DP>(e, &n, &d);
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < ROWS * V4 && n0 + n < N && d < D) val = *reinterpret_cast<const float4 *>(base + ((n0 + n) * v.ld + d) * 4);
    r.v[i] = val;
  }
}
template <int ROWS, int DP, int NT>
__device__ __forceinline__ void tile_store_f32(float *__restrict__ lds, const TileRegsF32<ROWS, DP, NT> &r, int tid) {
  constexpr int LD = DP + 1, V4 = DP / 4;
#pragma unroll
  for (int i = 0; i < TileRegsF32<ROWS, DP, NT>::N; ++i) {
    const int e = tid + i * NT;
    int n, d;
    tile_chunk<ROWS, DP>(e, &n, &d);
    if (e < ROWS * V4) {
      float *dst = lds + n * LD + d;
      dst[0] = r.v[i].x; dst[1] = r.v[i].y; dst[2] = r.v[i].z; dst[3] = r.v[i].w;
    }
  }
}

template <int ROWS, int DP, int NT>
__device__ __forceinline__ void stage_tile(float *__restrict__ lds, const OperandView &v,
                                           const char *base, int64_t n0, int64_t N, int D, int tid) {
  constexpr int LD = DP + 1;
  const int prec = v.precision;
  const int64_t ld = v.ld;
  if (!v.transposed) {
    const bool al16 = ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    if (prec == PREC_FP32 && al16 && (D & 3) == 0 && (ld & 3) == 0) {
      constexpr int V4 = DP / 4;
      for (int e = tid; e < ROWS * V4; e += NT) {
        int n, d;
        tile_chunk<ROWS, DP>(e, &n, &d);
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + n < N && d < D) val = *reinterpret_cast<const float4 *>(base + ((n0 + n) * ld + d) * 4);
        float *dst = lds + n * LD + d;
        dst[0] = val.x; dst[1] = val.y; dst[2] = val.z; dst[3] = val.w;
      }
    } else if (prec != PREC_FP32 && al16 && (D & 7) == 0 && (ld & 7) == 0) {
      constexpr int V8 = DP / 8;
      for (int e = tid; e < ROWS * V8; e += NT) {
        const int n = e / V8, d = (e % V8) * 8;
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (n0 + n < N && d < D) raw = *reinterpret_cast<const u32x4 *>(base + ((n0 + n) * ld + d) * 2);
        float *dst = lds + n * LD + d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w = raw[j];
          if (prec == PREC_BF16) {
            dst[2 * j] = __builtin_bit_cast(float, w << 16);
            dst[2 * j + 1] = __builtin_bit_cast(float, w & 0xFFFF0000u);
          } else {
            const f16x2 h = __builtin_bit_cast(f16x2, w);
            dst[2 * j] = (float)h[0];
            dst[2 * j + 1] = (float)h[1];
          }
        }
      }
    } else {
      for (int e = tid; e < ROWS * DP; e += NT) {
        const int n = e / DP, d = e % DP;
        float val = 0.f;
        if (n0 + n < N && d < D) val = load_elem(base, (n0 + n) * ld + d, prec);
        lds[n * LD + d] = val;
      }
    }
  } else {
    // column-major: element (n, d) at d*ld + n  (AttentionKernel.swift:189-195); lanes along n
    for (int e = tid; e < ROWS * DP; e += NT) {
      const int n = e % ROWS, d = e / ROWS;
      float val = 0.f;
      if (n0 + n < N && d < D) val = load_elem(base, (int64_t)d * ld + (n0 + n), prec);
      lds[n * LD + d] = val;
    }
  }
}

// LDS fp32 tile [ROWS][DP+1] -> global (any dtype / layout), guarded.
template <int ROWS, int DP, int NT>
__device__ __forceinline__ void store_tile(const float *__restrict__ lds, const OperandView &v,
                                           char *base, int64_t n0, int64_t N, int D, int tid) {
  constexpr int LD = DP + 1;
  const int prec = v.precision;
  const int64_t ld = v.ld;
  if (!v.transposed) {
    const bool al16 = ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    if (prec == PREC_FP32 && al16 && (D & 3) == 0 && (ld & 3) == 0) {
      constexpr int V4 = DP / 4;
      for (int e = tid; e < ROWS * V4; e += NT) {
        const int n = e / V4, d = (e % V4) * 4;
        if (n0 + n < N && d < D) {
          const float *src = lds + n * LD + d;
          *reinterpret_cast<float4 *>(base + ((n0 + n) * ld + d) * 4) = make_float4(src[0], src[1], src[2], src[3]);
        }
      }
    } else {
      for (int e = tid; e < ROWS * DP; e += NT) {
        const int n = e / DP, d = e % DP;
        if (n0 + n < N && d < D) store_elem(base, (n0 + n) * ld + d, prec, lds[n * LD + d]);
      }
    }
  } else {
    for (int e = tid; e < ROWS * DP; e += NT) {
      const int n = e % ROWS, d = e / ROWS;
      if (n0 + n < N && d < D) store_elem(base, (int64_t)d * ld + (n0 + n), prec, lds[n * LD + d]);
    }
  }
}

__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int generic_max(int a, int b) { return a > b ? a : b; }

// LDS floats needed by each kernel (host uses the same formulas)
template <int DP, int NW, bool CACHE> constexpr int generic_fwd_lds_floats() {
  return CACHE ? generic_max(NW * 32, 64) * (DP + 1) : (NW * 32 + 64) * (DP + 1);
}
// QONLY (dQ): Q cached in registers, dO left in LDS; SEQ (dK/dV): Q and dO tiles share ONE LDS buffer (staged in turn) --
// the two arrangements that let a 32-row block of a 384-wide head fit the 160 KiB of LDS and the 512 registers
template <int DP, int NW, bool CACHE, bool QONLY = false> constexpr int generic_dq_lds_floats() {
  return CACHE ? generic_max(NW * 32, 64) * (DP + 1) : QONLY ? (NW * 32 + 64) * (DP + 1) : (2 * NW * 32 + 64) * (DP + 1);
}
template <int DP, int NW, bool CACHE, bool SEQ = false> constexpr int generic_dkv_lds_floats() {
  return (CACHE ? generic_max(NW * 32, 64) * (DP + 1) : SEQ ? (2 * NW * 32 + 32) * (DP + 1) : (2 * NW * 32 + 64) * (DP + 1)) + 64;
}

// ----------------------------------------------------------------------------------------------
// forward: O = softmax(Q K^T / sqrt(D)) V,  L = m + log2(l)       (+Source.swift:158-200)
// grid = (ceil(R / (32*NW)), heads, batches); block = 64*NW.
// ----------------------------------------------------------------------------------------------
// Two workgroups per compute unit for DP <= 128 (LDS: 2 x 66 KiB): capping the kernel at 256 registers costs
// a 28-byte spill outside the loop and buys the second resident workgroup: 2.71 -> 2.29 ms at N=4096 D=128 fp32,
// 32 heads (the same cap on the dQ kernel spills 188 bytes and loses 10 %, so it stays uncapped).
template <int DP, int NW, bool CACHE, bool MASKED = false>
__global__ __launch_bounds__(NW * 64, (DP <= 128 ? 2 : 1)) void attn_generic_fwd(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = DP + 1, BR = NW * 32, BC = 32, NT = NW * 64, NDB = DP / 32, NS = DP / 2;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  const uint32_t head = blockIdx.y, batch = blockIdx.z;
  const int64_t r0 = (int64_t)blockIdx.x * BR;
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);

  float *Qs = smem;
  float *Ks = CACHE ? smem : smem + BR * LD;
  float *Vs = Ks + BC * LD;

  stage_tile<BR, DP, NT>(Qs, a.op[SLOT_Q], operand_base(a.op[SLOT_Q], head, batch), r0, R, D, tid);
  __syncthreads();
  const float *qrow = Qs + (wave * 32 + q) * LD + hi;
  float qf[CACHE ? NS : 1];
  if constexpr (CACHE) {
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = qrow[2 * s];
    __syncthreads();
  }

  f32x16 o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m = -3.402823466e+38f;          // +Caching.swift:310
  float l = 1.401298464e-45f;           // +Caching.swift:311 (denorm_min)

  const char *kbase = operand_base(a.op[SLOT_K], head, batch);
  const char *vbase = operand_base(a.op[SLOT_V], head, batch);

  // fp32 row-major operands (the FP32 production case): the next tile's global loads are issued
  // before the arithmetic of this tile and land in registers; other layouts stage synchronously.
  // causal extension: row r sees column c iff c <= r + coff; columns past the last row's limit are
  // never visited by this workgroup, the tiles on the diagonal are masked element-wise below
  const int coff = causal_offset(R, C);
  const int Cend = a.causal ? (int)min((int64_t)C, min((int64_t)R, r0 + BR) + coff) : C;
  constexpr bool CAN_PREFETCH = (DP <= 128);   // 2 x (32 x DP / 4 / NT) float4 of staging registers
  const uint32_t *mrow = MASKED ? mask_base(a, head, batch) : nullptr;   // MASKED: separate code objects, the dense ones carry no mask code
  const bool prefetch = !MASKED && CAN_PREFETCH && f32_fast_path(a.op[SLOT_K], kbase, D) && f32_fast_path(a.op[SLOT_V], vbase, D);
  TileRegsF32<BC, DP, NT> kregs, vregs;
  if (prefetch) {
    tile_load_f32<BC, DP, NT>(kregs, a.op[SLOT_K], kbase, 0, C, D, tid);
    tile_load_f32<BC, DP, NT>(vregs, a.op[SLOT_V], vbase, 0, C, D, tid);
  }
  for (int c0 = 0; c0 < Cend; c0 += BC) {
    if constexpr (MASKED) {
      if (!mask_bit(mrow, a.maskWords, r0, c0)) continue;   // workgroup-uniform: the block is never loaded
    }
    if (prefetch) {
      tile_store_f32<BC, DP, NT>(Ks, kregs, tid);
      tile_store_f32<BC, DP, NT>(Vs, vregs, tid);
    } else {
      stage_tile<BC, DP, NT>(Ks, a.op[SLOT_K], kbase, c0, C, D, tid);
      stage_tile<BC, DP, NT>(Vs, a.op[SLOT_V], vbase, c0, C, D, tid);
    }
    __syncthreads();
    if (prefetch && c0 + BC < Cend) {
      tile_load_f32<BC, DP, NT>(kregs, a.op[SLOT_K], kbase, c0 + BC, C, D, tid);
      tile_load_f32<BC, DP, NT>(vregs, a.op[SLOT_V], vbase, c0 + BC, C, D, tid);
    }

    // S^T = K Q^T : lane holds query (wave*32+q), keys c0 + crow(r, hi)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float *krow = Ks + q * LD + hi;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      const float bq = CACHE ? qf[t] : qrow[2 * t];
      s = mfma_f32(krow[2 * t], bq, s);
    }
    if (c0 + BC > C) { // maskAttentionMatrixEdge, +Softmax.swift:228-260
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) >= C) s[r] = mask_value();
    }
    if (a.causal) {    // same mechanism, applied to the columns the row may not see
      const int64_t limit = r0 + wave * 32 + q + coff;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) > limit) s[r] = mask_value();
    }
    // onlineReduceMaximum / onlineCorrectO, +Softmax.swift:267-301
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = mx * a.scale2;
    // corr = (m_new > m) ? exp2(m - m_new) : 1.  When no lane of the wave saw its maximum grow the
    // correction is exactly 1 everywhere: skip the O-wide multiply (wave-uniform branch, same results).
    if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {
      float corr = 1.f;
      if (m_new > m) { corr = fast_exp2(m - m_new); m = m_new; }
      l *= corr;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= corr;
    }
    // softmax + onlineReduceSum, +Softmax.swift:304-324, :406-417
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(s[r] * a.scale2 - m); psum += s[r]; }
    l += psum;
    // O^T += V^T P^T with the key index permuted: step t uses key crow(t, hi)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float *vrow = Vs + crow(t, hi) * LD + q;
#pragma unroll
      for (int db = 0; db < NDB; ++db) o[db] = mfma_f32(vrow[32 * db], s[t], o[db]);
    }
    __syncthreads();
  }

  const float l_tot = l + __shfl_xor(l, 32);
  const float inv = (m > -1e37f) ? 1.0f / l_tot : 0.f;   // +Source.swift:165-171 (0: a row whose every block is masked out)
  float *Os = smem;                         // [BR][LD]
  float *orow = Os + (wave * 32 + q) * LD;
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) orow[32 * db + crow(r, hi)] = o[db][r] * inv;
  __syncthreads();
  store_tile<BR, DP, NT>(Os, a.op[SLOT_O], operand_base(a.op[SLOT_O], head, batch), r0, R, D, tid);
  const int64_t row = r0 + wave * 32 + q;
  if (hi == 0 && row < R)  // L = m + log2(l), +Caching.swift:373-377
    store_elem(operand_base(a.op[SLOT_L], head, batch), row, a.op[SLOT_L].precision, m + log2f(l_tot));
}

// ----------------------------------------------------------------------------------------------
// backward dQ: D = rowsum(dO*O)/sqrt(D); dQ = sum_c dS K                (+Source.swift:202-242)
// ----------------------------------------------------------------------------------------------
template <int DP, int NW, bool CACHE, bool MASKED = false, bool QONLY = false>
__global__ __launch_bounds__(NW * 64) void attn_generic_dq(const KernelArgs a) {
  static_assert(!(CACHE && QONLY), "QONLY: Q cached, dO streamed from LDS");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = DP + 1, BR = NW * 32, BC = 32, NT = NW * 64, NDB = DP / 32, NS = DP / 2;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  const uint32_t head = blockIdx.y, batch = blockIdx.z;
  const int64_t r0 = (int64_t)blockIdx.x * BR;
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t row = r0 + wave * 32 + q;

  constexpr bool QC = CACHE || QONLY;   // Q fragments live in registers
  float *Qs = smem;
  float *dOs = QC ? smem : smem + BR * LD;                           // QONLY: dO takes the place of Q once Q is cached
  float *Ks = CACHE ? smem : QONLY ? smem + BR * LD : smem + 2 * BR * LD;
  float *Vs = Ks + BC * LD;

  float qf[QC ? NS : 1], gf[CACHE ? NS : 1];
  const float *qrow = Qs + (wave * 32 + q) * LD + hi;
  const float *grow = dOs + (wave * 32 + q) * LD + hi;
  stage_tile<BR, DP, NT>(Qs, a.op[SLOT_Q], operand_base(a.op[SLOT_Q], head, batch), r0, R, D, tid);
  __syncthreads();
  if constexpr (QC) {
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[s] = qrow[2 * s];
    __syncthreads();
  }
  stage_tile<BR, DP, NT>(dOs, a.op[SLOT_dO], operand_base(a.op[SLOT_dO], head, batch), r0, R, D, tid);
  __syncthreads();
  if constexpr (CACHE) {
#pragma unroll
    for (int s = 0; s < NS; ++s) gf[s] = grow[2 * s];
  }
  // computeD, +Softmax.swift:32-221: D = (sum_d dO*O) * 1/sqrt(D).  O is read straight from
  // memory (once per row); the two half-waves split the head dimension.
  float dterm = 0.f;
  {
    const OperandView &ov = a.op[SLOT_O];
    const char *obase = operand_base(ov, head, batch);
    if (row < R) {
      for (int d = hi; d < D; d += 2) {
        const int64_t idx = ov.transposed ? (int64_t)d * ov.ld + row : row * ov.ld + d;
        dterm += load_elem(obase, idx, ov.precision) * grow[d - hi];
      }
    }
    dterm += __shfl_xor(dterm, 32);
    dterm *= a.scale;
  }
  float Lrow = 0.f;
  if (row < R) Lrow = load_elem(operand_base(a.op[SLOT_L], head, batch), row, a.op[SLOT_L].precision);
  __syncthreads();

  f32x16 acc[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

  const char *kbase = operand_base(a.op[SLOT_K], head, batch);
  const char *vbase = operand_base(a.op[SLOT_V], head, batch);
  // fp32 row-major operands (the FP32 production case): the next tile's global loads are issued
  // before the arithmetic of this tile and land in registers; other layouts stage synchronously.
  // causal extension: row r sees column c iff c <= r + coff; columns past the last row's limit are
  // never visited by this workgroup, the tiles on the diagonal are masked element-wise below
  const int coff = causal_offset(R, C);
  const int Cend = a.causal ? (int)min((int64_t)C, min((int64_t)R, r0 + BR) + coff) : C;
  constexpr bool CAN_PREFETCH = (DP <= 128);   // 2 x (32 x DP / 4 / NT) float4 of staging registers
  const uint32_t *mrow = MASKED ? mask_base(a, head, batch) : nullptr;   // MASKED: separate code objects, the dense ones carry no mask code
  const bool prefetch = !MASKED && CAN_PREFETCH && f32_fast_path(a.op[SLOT_K], kbase, D) && f32_fast_path(a.op[SLOT_V], vbase, D);
  TileRegsF32<BC, DP, NT> kregs, vregs;
  if (prefetch) {
    tile_load_f32<BC, DP, NT>(kregs, a.op[SLOT_K], kbase, 0, C, D, tid);
    tile_load_f32<BC, DP, NT>(vregs, a.op[SLOT_V], vbase, 0, C, D, tid);
  }
  for (int c0 = 0; c0 < Cend; c0 += BC) {
    if constexpr (MASKED) {
      if (!mask_bit(mrow, a.maskWords, r0, c0)) continue;   // workgroup-uniform: the block is never loaded
    }
    if (prefetch) {
      tile_store_f32<BC, DP, NT>(Ks, kregs, tid);
      tile_store_f32<BC, DP, NT>(Vs, vregs, tid);
    } else {
      stage_tile<BC, DP, NT>(Ks, a.op[SLOT_K], kbase, c0, C, D, tid);
      stage_tile<BC, DP, NT>(Vs, a.op[SLOT_V], vbase, c0, C, D, tid);
    }
    __syncthreads();
    if (prefetch && c0 + BC < Cend) {
      tile_load_f32<BC, DP, NT>(kregs, a.op[SLOT_K], kbase, c0 + BC, C, D, tid);
      tile_load_f32<BC, DP, NT>(vregs, a.op[SLOT_V], vbase, c0 + BC, C, D, tid);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    const float *krow = Ks + q * LD + hi;
    const float *vrow = Vs + q * LD + hi;
#pragma unroll
    for (int t = 0; t < NS; ++t) s = mfma_f32(krow[2 * t], QC ? qf[t] : qrow[2 * t], s);         // S^T = K Q^T
#pragma unroll
    for (int t = 0; t < NS; ++t) dp = mfma_f32(vrow[2 * t], CACHE ? gf[t] : grow[2 * t], dp);    // dP^T = V dO^T
    // P = exp2(S*scale2 - L); dS = P * (dP*scale - D)     (+Softmax.swift:409-427)
    // Padded columns: K,V rows are zero so dS*K contributes nothing (as in the reference, where
    // the zero padding comes from the async copy, +Accumulate.swift:330-346).
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = fast_exp2(s[r] * a.scale2 - Lrow);
      if (a.causal && c0 + crow(r, hi) > row + coff) p = 0.f;   // masked column: P = 0, hence dS = 0
      s[r] = p * (dp[r] * a.scale - dterm);
    }
    // dQ^T += K^T dS^T, key index permuted as in forward
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float *kr = Ks + crow(t, hi) * LD + q;
#pragma unroll
      for (int db = 0; db < NDB; ++db) acc[db] = mfma_f32(kr[32 * db], s[t], acc[db]);
    }
    __syncthreads();
  }

  float *Ts = smem; // [BR][LD]
  float *trow = Ts + (wave * 32 + q) * LD;
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) trow[32 * db + crow(r, hi)] = acc[db][r];
  __syncthreads();
  store_tile<BR, DP, NT>(Ts, a.op[SLOT_dQ], operand_base(a.op[SLOT_dQ], head, batch), r0, R, D, tid);
  if (hi == 0 && row < R)   // +Caching.swift:381-413 (BF16 by truncation, :395-401)
    store_elem(operand_base(a.op[SLOT_D], head, batch), row, a.op[SLOT_D].precision, dterm);
}

// ----------------------------------------------------------------------------------------------
// backward dK/dV: dV = sum_r P^T dO ; dK = sum_r dS^T Q, parallel over columns
//                                                                      (+Source.swift:244-293)
// grid = (ceil(C / (32*NW)), heads, batches)
// ----------------------------------------------------------------------------------------------
template <int DP, int NW, bool CACHE, bool MASKED = false, bool SEQ = false>
__global__ __launch_bounds__(NW * 64) void attn_generic_dkv(const KernelArgs a) {
  static_assert(!(CACHE && SEQ), "SEQ: K and V stay in LDS, Q and dO tiles take turns in one buffer");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = DP + 1, BCOL = NW * 32, BRW = 32, NT = NW * 64, NDB = DP / 32, NS = DP / 2;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, kc = lane & 31, hi = lane >> 5;
  const uint32_t head = blockIdx.y, batch = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.x * BCOL;
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);

  float *Kst = smem;                                  // [BCOL][LD]
  float *Vst = CACHE ? smem : smem + BCOL * LD;       // [BCOL][LD]
  float *Qs = CACHE ? smem : smem + 2 * BCOL * LD;    // [32][LD]
  float *dOs = SEQ ? Qs : Qs + BRW * LD;              // [32][LD]  (SEQ: the same buffer, staged in turn)
  float *LDs = smem + generic_dkv_lds_floats<DP, NW, CACHE, SEQ>() - 64;  // L[32], D[32]

  float kf[CACHE ? NS : 1], vf[CACHE ? NS : 1];
  const float *krow = Kst + (wave * 32 + kc) * LD + hi;
  const float *vrow = Vst + (wave * 32 + kc) * LD + hi;
  stage_tile<BCOL, DP, NT>(Kst, a.op[SLOT_K], operand_base(a.op[SLOT_K], head, batch), c0, C, D, tid);
  __syncthreads();
  if constexpr (CACHE) {
#pragma unroll
    for (int s = 0; s < NS; ++s) kf[s] = krow[2 * s];
    __syncthreads();
  }
  stage_tile<BCOL, DP, NT>(Vst, a.op[SLOT_V], operand_base(a.op[SLOT_V], head, batch), c0, C, D, tid);
  __syncthreads();
  if constexpr (CACHE) {
#pragma unroll
    for (int s = 0; s < NS; ++s) vf[s] = vrow[2 * s];
    __syncthreads();
  }

  f32x16 dk[NDB], dv[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  const char *qbase = operand_base(a.op[SLOT_Q], head, batch);
  const char *gbase = operand_base(a.op[SLOT_dO], head, batch);
  const char *lbase = operand_base(a.op[SLOT_L], head, batch);
  const char *dbase = operand_base(a.op[SLOT_D], head, batch);
  // causal extension: rows above the workgroup's first column minus the offset see none of its columns
  const int coff = causal_offset(R, C);
  const int rstart = a.causal ? (int)(max((int64_t)0, c0 - coff) / BRW) * BRW : 0;
  constexpr bool CAN_PREFETCH = (DP <= 128) && !SEQ;
  const uint32_t *mbase = MASKED ? mask_base(a, head, batch) : nullptr;
  const bool prefetch = !MASKED && CAN_PREFETCH && f32_fast_path(a.op[SLOT_Q], qbase, D) && f32_fast_path(a.op[SLOT_dO], gbase, D);
  TileRegsF32<BRW, DP, NT> qregs, gregs;
  if (prefetch) {
    tile_load_f32<BRW, DP, NT>(qregs, a.op[SLOT_Q], qbase, rstart, R, D, tid);
    tile_load_f32<BRW, DP, NT>(gregs, a.op[SLOT_dO], gbase, rstart, R, D, tid);
  }
  for (int r0 = rstart; r0 < R; r0 += BRW) {
    if constexpr (MASKED) {
      if (!mask_bit(mbase, a.maskWords, r0, c0)) continue;   // workgroup-uniform
    }
    if (prefetch) {
      tile_store_f32<BRW, DP, NT>(Qs, qregs, tid);
      tile_store_f32<BRW, DP, NT>(dOs, gregs, tid);
    } else {
      stage_tile<BRW, DP, NT>(Qs, a.op[SLOT_Q], qbase, r0, R, D, tid);
      if constexpr (!SEQ) stage_tile<BRW, DP, NT>(dOs, a.op[SLOT_dO], gbase, r0, R, D, tid);
    }
    if (tid < 64) { // L and D slices along the traversal dimension (+Softmax.swift:356-381, :472-503)
      const int rr = tid & 31;
      float val = 0.f;
      if (r0 + rr < R)
        val = (tid < 32) ? load_elem(lbase, r0 + rr, a.op[SLOT_L].precision)
                         : load_elem(dbase, r0 + rr, a.op[SLOT_D].precision);
      LDs[tid] = val;
    }
    __syncthreads();
    if (prefetch && r0 + BRW < R) {
      tile_load_f32<BRW, DP, NT>(qregs, a.op[SLOT_Q], qbase, r0 + BRW, R, D, tid);
      tile_load_f32<BRW, DP, NT>(gregs, a.op[SLOT_dO], gbase, r0 + BRW, R, D, tid);
    }
    // S = Q K^T (not swapped): lane holds key (wave*32+kc), rows r0 + crow(r, hi)
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    const float *qr = Qs + kc * LD + hi;
    const float *gr = dOs + kc * LD + hi;
#pragma unroll
    for (int t = 0; t < NS; ++t) s = mfma_f32(qr[2 * t], CACHE ? kf[t] : krow[2 * t], s);
    if constexpr (SEQ) {   // Q has been multiplied: dO takes its place in the shared buffer
      __syncthreads();
      stage_tile<BRW, DP, NT>(dOs, a.op[SLOT_dO], gbase, r0, R, D, tid);
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) dp = mfma_f32(gr[2 * t], CACHE ? vf[t] : vrow[2 * t], dp);   // dP = dO V^T
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float Lr = LDs[crow(r, hi)];
      const float Dr = LDs[32 + crow(r, hi)];
      p[r] = fast_exp2(s[r] * a.scale2 - Lr);
      if (a.causal && c0 + wave * 32 + kc > r0 + crow(r, hi) + coff) p[r] = 0.f;   // masked: P = 0
      s[r] = p[r] * (dp[r] * a.scale - Dr);
    }
    // dV^T += dO^T P ; dK^T += Q^T dS   (row index permuted; padded rows of Q/dO are zero)
    if constexpr (!SEQ) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float *g2 = dOs + crow(t, hi) * LD + kc;
        const float *q2 = Qs + crow(t, hi) * LD + kc;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          dv[db] = mfma_f32(g2[32 * db], p[t], dv[db]);
          dk[db] = mfma_f32(q2[32 * db], s[t], dk[db]);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float *g2 = dOs + crow(t, hi) * LD + kc;
#pragma unroll
        for (int db = 0; db < NDB; ++db) dv[db] = mfma_f32(g2[32 * db], p[t], dv[db]);
      }
      __syncthreads();   // dO has been multiplied: Q returns for the dK product
      stage_tile<BRW, DP, NT>(Qs, a.op[SLOT_Q], qbase, r0, R, D, tid);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float *q2 = Qs + crow(t, hi) * LD + kc;
#pragma unroll
        for (int db = 0; db < NDB; ++db) dk[db] = mfma_f32(q2[32 * db], s[t], dk[db]);
      }
    }
    __syncthreads();
  }

  float *Ts = smem; // [BCOL][LD], used twice
  float *trow = Ts + (wave * 32 + kc) * LD;
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) trow[32 * db + crow(r, hi)] = dv[db][r];
  __syncthreads();
  store_tile<BCOL, DP, NT>(Ts, a.op[SLOT_dV], operand_base(a.op[SLOT_dV], head, batch), c0, C, D, tid);
  __syncthreads();
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) trow[32 * db + crow(r, hi)] = dk[db][r];
  __syncthreads();
  store_tile<BCOL, DP, NT>(Ts, a.op[SLOT_dK], operand_base(a.op[SLOT_dK], head, batch), c0, C, D, tid);
}

} // namespace mfa
