// attn_paged.h -- the three attention kernels for ANY head dimension (D > 384): D-blocked products with NOTHING cached and
// the accumulators PAGED through the output buffers -- the reference's own scheme for head dimensions its registers cannot
// hold, restated for gfx950:
//
//   reference: AttentionKernel+OuterProduct.swift:133-171, :441-475 (the left-hand operand re-loaded per D block when it is
//   not cached), AttentionKernel+Accumulate.swift:403-469 (an accumulator that is not cached: load the block, scale by the
//   every-iteration factor, multiply-accumulate, apply the last-iteration factor, store back), and the fall-through of the
//   parameter tables to their last row for any larger D (AttentionDescriptor+Parameters.swift:60-65).
//
// Up to D = 384 the accumulators never leave the registers on this chip (attn_generic.h); beyond that this file
// serves every D with one code object per kernel type: a workgroup of 256 work-items owns 32 rows (forward, backwardQuery) or
// 32 keys (backwardKeyValue), walks the other dimension in tiles of 32 and the head dimension in chunks of 64:
//     S (and dP) = sum over chunks of [32 x 64] . [32 x 64]^T          (operands staged in LDS as fp32, any storage type / layout)
//     softmax / derivative on the 32 x 32 tile, 4 entries per work-item, row reductions over 8 lanes
//     O (dQ; dV, dK) chunk by chunk: load [32 x 64] fp32 from the OUTPUT buffer, rescale, += P . V chunk, store
// All arithmetic is fp32 FMA.  This is a correctness path (the reference's tables end at 384 too): O, dQ, dK, dV must be FP32
// -- they are the paging store, exactly as in the reference, where these four are always FP32 (+Precisions.swift:140-143).
#pragma once
#include "attn_generic.h"

namespace mfa {
namespace paged {

constexpr int BR = 32, BC = 32, DC = 64, LD = DC + 1;

struct Grid { uint32_t blocks, heads, batches; };

// element (n, d) of an operand view: [seq][D] row-major, or [D][seq] when transposed (AttentionKernel.swift:189-204)
__device__ __forceinline__ int64_t at(const OperandView &v, int64_t n, int64_t d) { return v.transposed ? d * v.ld + n : n * v.ld + d; }

// stage rows n0 .. n0 + 31, columns d0 .. d0 + 63 of an operand into an fp32 LDS tile, zero padded (the role of
// simdgroup_event::async_copy's clamp_to_zero, GEMMHeaders.swift:166-193)
__device__ __forceinline__ void stage(float (*dst)[LD], const OperandView &v, const char *base, int64_t n0, int64_t nmax, int d0, int D, int tid) {
  for (int e = tid; e < 32 * DC; e += 256) {
    int n, d;
    if (v.transposed) { n = e & 31; d = e >> 5; } else { n = e >> 6; d = e & 63; }   // consecutive work-items along the contiguous axis
    const bool ok = n0 + n < nmax && d0 + d < D;
    dst[n][d] = ok ? load_elem(base, at(v, n0 + n, d0 + d), v.precision) : 0.f;
  }
}

__device__ __forceinline__ float reduce8_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 1)); x = fmaxf(x, __shfl_xor(x, 2)); return fmaxf(x, __shfl_xor(x, 4));
}
__device__ __forceinline__ float reduce8_add(float x) {
  x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); return x + __shfl_xor(x, 4);
}

__device__ __forceinline__ void decode(const Grid &g, uint32_t *blk, uint32_t *head, uint32_t *batch) {
  const uint32_t hb = blockIdx.x / g.blocks;
  *blk = blockIdx.x % g.blocks;
  *head = hb % g.heads;
  *batch = hb / g.heads;
}

// key c of row r is visible: inside the batch entry's keys, at or below the causal diagonal (extension), in an active block
__device__ __forceinline__ bool visible(int64_t row, int64_t col, int C, int coff, bool causal) { return col < C && (!causal || col <= row + coff); }

}  // namespace paged

// ---------------------------------------------------------------------------------------------------------------- forward
// loopForward (+Source.swift:158-200) with O not cached: O is paged through its buffer per key tile
__global__ __launch_bounds__(256) void attn_paged_fwd(const KernelArgs a, const paged::Grid grid) {
  using namespace paged;
  __shared__ float Qs[BR][LD], Ks[BC][LD], Ps[BR][BC + 1];
  const int tid = threadIdx.x, r = tid >> 3, c4 = (tid & 7) * 4, d8 = (tid & 7) * 8;
  uint32_t rblk, head, batch;
  decode(grid, &rblk, &head, &batch);
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * BR, row = r0 + r;
  if (r0 >= R) return;
  const char *qb = operand_base(a.op[SLOT_Q], head, batch), *kb = operand_base(a.op[SLOT_K], head, batch), *vb = operand_base(a.op[SLOT_V], head, batch);
  float *ob = reinterpret_cast<float *>(operand_base(a.op[SLOT_O], head, batch));
  const OperandView &ov = a.op[SLOT_O];
  const uint32_t *mk = mask_base(a, head, batch);
  const int coff = causal_offset(R, C);
  const bool causal = a.causal != 0;
  float m = -3.402823466e+38f, l = 0.f;   // +Caching.swift:310-311
  bool first = true;
  for (int64_t c0 = 0; c0 < C; c0 += BC) {
    if (causal && c0 > min((int64_t)R, r0 + BR) - 1 + coff) break;                    // beyond the diagonal of the block's last row
    if (mk && !mask_bit(mk, a.maskWords, r0, c0)) continue;                            // inactive block: never loaded
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < D; d0 += DC) {                                              // S = Q K^T, blocked over D (+OuterProduct.swift:441-475)
      stage(Qs, a.op[SLOT_Q], qb, r0, R, d0, D, tid);
      stage(Ks, a.op[SLOT_K], kb, c0, C, d0, D, tid);
      __syncthreads();
#pragma unroll 8
      for (int d = 0; d < DC; ++d) {
        const float qv = Qs[r][d];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += qv * Ks[c4 + k][d];
      }
      __syncthreads();
    }
    bool vis[4];
    float mx = -3.402823466e+38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      vis[k] = visible(row, c0 + c4 + k, C, coff, causal);
      s[k] *= a.scale2;
      if (vis[k]) mx = fmaxf(mx, s[k]);
    }
    const float m_new = fmaxf(m, reduce8_max(mx));                                   // onlineReduceMaximum (+Softmax.swift:267-290)
    const float corr = fast_exp2(m - m_new);                                         // onlineCorrectO (:290-301); 0 while m is the start value
    float psum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float p = vis[k] ? fast_exp2(s[k] - m_new) : 0.f;
      Ps[r][c4 + k] = p;
      psum += p;
    }
    l = l * corr + reduce8_add(psum);                                                // onlineReduceSum (:303-324)
    m = m_new;
    __syncthreads();
    for (int d0 = 0; d0 < D; d0 += DC) {                                              // O = O corr + P V, paged (+Accumulate.swift:403-469)
      stage(Ks, a.op[SLOT_V], vb, c0, C, d0, D, tid);
      __syncthreads();
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = d0 + d8 + e;
        acc[e] = (!first && row < R && d < D) ? ob[at(ov, row, d)] * corr : 0.f;
      }
      for (int c = 0; c < BC; ++c) {
        const float p = Ps[r][c];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += p * Ks[c][d8 + e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = d0 + d8 + e;
        if (row < R && d < D) ob[at(ov, row, d)] = acc[e];
      }
      __syncthreads();
    }
    first = false;
  }
  // last-iteration scale 1 / l (+Source.swift:165-171) and L = m + log2 l (+Caching.swift:373-377)
  const float l_tot = l + 1.401298464e-45f;
  const float inv = (!first && l_tot > 1e-30f) ? 1.0f / l_tot : 0.f;
  if (row < R) {
    for (int d = d8; d < D; d += DC) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (d + e < D) ob[at(ov, row, d + e)] = first ? 0.f : ob[at(ov, row, d + e)] * inv;
    }
    if ((tid & 7) == 0) store_elem(operand_base(a.op[SLOT_L], head, batch), row, a.op[SLOT_L].precision, m + log2f(l_tot));
  }
}

// ---------------------------------------------------------------------------------------------------------- backwardQuery
// loopBackwardQuery (+Source.swift:202-242) with dQ paged; computeD first (+Softmax.swift:32-221)
__global__ __launch_bounds__(256) void attn_paged_dq(const KernelArgs a, const paged::Grid grid) {
  using namespace paged;
  __shared__ float Qs[BR][LD], Ks[BC][LD], Gs[BR][LD], Vs[BC][LD], Ss[BR][BC + 1];
  const int tid = threadIdx.x, r = tid >> 3, c4 = (tid & 7) * 4, d8 = (tid & 7) * 8;
  uint32_t rblk, head, batch;
  decode(grid, &rblk, &head, &batch);
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * BR, row = r0 + r;
  if (r0 >= R) return;
  const char *qb = operand_base(a.op[SLOT_Q], head, batch), *kb = operand_base(a.op[SLOT_K], head, batch), *vb = operand_base(a.op[SLOT_V], head, batch);
  const char *gb = operand_base(a.op[SLOT_dO], head, batch), *obase = operand_base(a.op[SLOT_O], head, batch);
  float *qg = reinterpret_cast<float *>(operand_base(a.op[SLOT_dQ], head, batch));
  const OperandView &qgv = a.op[SLOT_dQ], &gv = a.op[SLOT_dO], &ovw = a.op[SLOT_O];
  const uint32_t *mk = mask_base(a, head, batch);
  const int coff = causal_offset(R, C);
  const bool causal = a.causal != 0;
  // D_row = sum_d dO O (1 / sqrt D)
  float dpart = 0.f;
  if (row < R)
    for (int d = tid & 7; d < D; d += 8) dpart += load_elem(gb, at(gv, row, d), gv.precision) * load_elem(obase, at(ovw, row, d), ovw.precision);
  const float dterm = reduce8_add(dpart) * a.scale;
  float lrow = 0.f;
  if (row < R) {
    lrow = load_elem(operand_base(a.op[SLOT_L], head, batch), row, a.op[SLOT_L].precision);
    if ((tid & 7) == 0) store_elem(operand_base(a.op[SLOT_D], head, batch), row, a.op[SLOT_D].precision, dterm);   // +Caching.swift:381-413
  }
  bool first = true;
  for (int64_t c0 = 0; c0 < C; c0 += BC) {
    if (causal && c0 > min((int64_t)R, r0 + BR) - 1 + coff) break;
    if (mk && !mask_bit(mk, a.maskWords, r0, c0)) continue;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < D; d0 += DC) {                                              // S = Q K^T and dP = dO V^T
      stage(Qs, a.op[SLOT_Q], qb, r0, R, d0, D, tid);
      stage(Ks, a.op[SLOT_K], kb, c0, C, d0, D, tid);
      stage(Gs, gv, gb, r0, R, d0, D, tid);
      stage(Vs, a.op[SLOT_V], vb, c0, C, d0, D, tid);
      __syncthreads();
#pragma unroll 8
      for (int d = 0; d < DC; ++d) {
        const float qv = Qs[r][d], gval = Gs[r][d];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += qv * Ks[c4 + k][d]; dp[k] += gval * Vs[c4 + k][d]; }
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                                    // softmax(derivative:) (+Softmax.swift:406-427)
      const float p = visible(row, c0 + c4 + k, C, coff, causal) ? fast_exp2(s[k] * a.scale2 - lrow) : 0.f;
      Ss[r][c4 + k] = p * (dp[k] * a.scale - dterm);                                 // dS = P (dP - D) / sqrt D; dQ = dS K
    }
    __syncthreads();
    for (int d0 = 0; d0 < D; d0 += DC) {                                              // dQ += dS K, paged
      stage(Ks, a.op[SLOT_K], kb, c0, C, d0, D, tid);
      __syncthreads();
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = d0 + d8 + e;
        acc[e] = (!first && row < R && d < D) ? qg[at(qgv, row, d)] : 0.f;
      }
      for (int c = 0; c < BC; ++c) {
        const float x = Ss[r][c];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += x * Ks[c][d8 + e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = d0 + d8 + e;
        if (row < R && d < D) qg[at(qgv, row, d)] = acc[e];
      }
      __syncthreads();
    }
    first = false;
  }
  if (first && row < R)
    for (int d = d8; d < D; d += DC)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (d + e < D) qg[at(qgv, row, d + e)] = 0.f;
}

// ------------------------------------------------------------------------------------------------------- backwardKeyValue
// loopBackwardKeyValue (+Source.swift:244-293) with dV and dK paged; parallel over the keys
__global__ __launch_bounds__(256) void attn_paged_dkv(const KernelArgs a, const paged::Grid grid) {
  using namespace paged;
  __shared__ float Qs[BR][LD], Ks[BC][LD], Gs[BR][LD], Vs[BC][LD], Ps[BR][BC + 1], Ss[BR][BC + 1];
  const int tid = threadIdx.x, r = tid >> 3, c4 = (tid & 7) * 4, d8 = (tid & 7) * 8;
  const int kc = tid >> 3;                 // second role of the work-item: key kc, eight columns of dV / dK
  uint32_t cblk, head, batch;
  decode(grid, &cblk, &head, &batch);
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t c0 = (int64_t)cblk * BC, col = c0 + kc;
  if (c0 >= C) return;
  const char *qb = operand_base(a.op[SLOT_Q], head, batch), *kb = operand_base(a.op[SLOT_K], head, batch), *vb = operand_base(a.op[SLOT_V], head, batch);
  const char *gb = operand_base(a.op[SLOT_dO], head, batch);
  const char *lb = operand_base(a.op[SLOT_L], head, batch), *db = operand_base(a.op[SLOT_D], head, batch);
  float *vg = reinterpret_cast<float *>(operand_base(a.op[SLOT_dV], head, batch)), *kg = reinterpret_cast<float *>(operand_base(a.op[SLOT_dK], head, batch));
  const OperandView &vgv = a.op[SLOT_dV], &kgv = a.op[SLOT_dK];
  const uint32_t *mk = mask_base(a, head, batch);
  const int coff = causal_offset(R, C);
  const bool causal = a.causal != 0;
  bool first = true;
  int64_t rstart = 0;
  if (causal) rstart = max((int64_t)0, c0 - coff) / BR * BR;                           // rows above see none of these keys
  for (int64_t r0 = rstart; r0 < R; r0 += BR) {
    if (mk && !mask_bit(mk, a.maskWords, r0, c0)) continue;
    const int64_t row = r0 + r;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d0 = 0; d0 < D; d0 += DC) {
      stage(Qs, a.op[SLOT_Q], qb, r0, R, d0, D, tid);
      stage(Ks, a.op[SLOT_K], kb, c0, C, d0, D, tid);
      stage(Gs, a.op[SLOT_dO], gb, r0, R, d0, D, tid);
      stage(Vs, a.op[SLOT_V], vb, c0, C, d0, D, tid);
      __syncthreads();
#pragma unroll 8
      for (int d = 0; d < DC; ++d) {
        const float qv = Qs[r][d], gval = Gs[r][d];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += qv * Ks[c4 + k][d]; dp[k] += gval * Vs[c4 + k][d]; }
      }
      __syncthreads();
    }
    float lrow = 0.f, drow = 0.f;                                                    // L and D along the traversal (+Softmax.swift:356-404)
    if (row < R) {
      lrow = load_elem(lb, row, a.op[SLOT_L].precision);
      drow = load_elem(db, row, a.op[SLOT_D].precision);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float p = (row < R && visible(row, c0 + c4 + k, C, coff, causal)) ? fast_exp2(s[k] * a.scale2 - lrow) : 0.f;
      Ps[r][c4 + k] = p;
      Ss[r][c4 + k] = p * (dp[k] * a.scale - drow);
    }
    __syncthreads();
    for (int d0 = 0; d0 < D; d0 += DC) {                                              // dV += P^T dO, dK += dS^T Q, paged
      stage(Gs, a.op[SLOT_dO], gb, r0, R, d0, D, tid);
      stage(Qs, a.op[SLOT_Q], qb, r0, R, d0, D, tid);
      __syncthreads();
      float av[8], ak[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = d0 + d8 + e;
        const bool ok = !first && col < C && d < D;
        av[e] = ok ? vg[at(vgv, col, d)] : 0.f;
        ak[e] = ok ? kg[at(kgv, col, d)] : 0.f;
      }
      for (int rr = 0; rr < BR; ++rr) {
        const float p = Ps[rr][kc], x = Ss[rr][kc];
#pragma unroll
        for (int e = 0; e < 8; ++e) { av[e] += p * Gs[rr][d8 + e]; ak[e] += x * Qs[rr][d8 + e]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = d0 + d8 + e;
        if (col < C && d < D) { vg[at(vgv, col, d)] = av[e]; kg[at(kgv, col, d)] = ak[e]; }
      }
      __syncthreads();
    }
    first = false;
  }
  if (first && col < C)
    for (int d = d8; d < D; d += DC)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (d + e < D) { vg[at(vgv, col, d + e)] = 0.f; kg[at(kgv, col, d + e)] = 0.f; }
}

} // namespace mfa
