// attn_fwd16_v2.h -- software-pipelined forward attention on the 16-bit matrix cores of gfx950.
//
// Same math, fragment maps and LDS images as attn_fwd16.h (read that header first); what changes
// is the schedule, re-derived for the CDNA4 issue model (one in-order instruction stream per wave,
// MFMA executing asynchronously in the SIMD's matrix pipe, 512 registers per lane):
//   * the S^T = K Q^T product of tile j+1 is issued in the same straight-line region as the
//     online softmax of tile j, so the matrix pipe works while the VALU exponentiates; then
//     O^T += V^T P^T of tile j.  Two score tiles are live (s_cur, s_next).
//   * K/V tiles travel global -> VGPR -> LDS through a 3-deep LDS ring with ONE barrier per tile:
//     the tile written in iteration j (tile j+1) replaces tile j-2, whose last reader finished
//     before the barrier of iteration j-1.  Loads for tile j+2 are issued right after the write and
//     have a full iteration to land.
//   * K/V/Q are fetched with bounds-checked buffer loads: rows past the end and head-dimension
//     chunks past D read as zero without any branch (the role of the zero-padding async copy in the
//     reference, GEMMHeaders.swift:166-193).
//   * the O rescale of the online softmax is deferred (THR, log2 units): the running max m is only
//     raised -- and O, l multiplied by exp2(m_old - m_new) -- when some row's block maximum exceeds
//     m by more than THR; otherwise P = exp2(S*scale2 - m) simply runs up to 2^THR.  THR = 0 is the
//     reference's rule (+Softmax.swift:290-301) exactly.  L = m + log2(l) is unaffected.
//   * the row sum l = sum_c P runs on the matrix pipe too (an all-ones A operand times P^T), which
//     also makes it the sum of the ROUNDED P, the reference's rule (+Softmax.swift:308-321); the
//     VALU keeps ~4.3 instructions per score: fma, exp2, 1/2 cvt_pk, 1/2 max3, moves.
//   * the epilogue transposes O through LDS so that every store instruction writes whole rows.
#pragma once
#include "attn_fwd16.h"

namespace mfa {

template <typename T, int D, int NW, int RB, int THR, bool MSUM>
__global__ __launch_bounds__(NW * 64) void attn_fwd16_v2(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BC = 64, NT = NW * 64, NDB = D / 32, NKS = D / 16;
  constexpr int ROWB = D * 2, TILE = BC * D * 2, STAGE = 2 * TILE;
  constexpr int CPR = D / 8, NCH = BC * CPR / NT;
  static_assert(BC * CPR % NT == 0, "tile must divide evenly over the workgroup");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &rblk, &head, &batch);
  int R = a.R, C = a.C;
  const int Dr = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * (NW * RB * 32) + wave * (RB * 32);
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2,
                 ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;

  // bounds-checked buffer descriptors (all inputs wave-uniform: kernel arguments and blockIdx)
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)R * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_K], head, batch), 0, (uint32_t)C * ldk2, 0x00020000);
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_V], head, batch), 0, (uint32_t)C * ldv2, 0x00020000);
  constexpr uint32_t OOB = 0xFFFFFF00u;

  // ---- Q fragments (B operand of S^T = K Q^T)
  v8 qf[RB][NKS];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const uint32_t rowoff = (uint32_t)(r0 + b * 32 + q) * ldq2;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int d0 = 16 * s + 8 * hi;
      const uint32_t off = (d0 < Dr && r0 + b * 32 + q < R) ? rowoff + d0 * 2 : OOB;
      qf[b][s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(qres, off, 0, 0));
    }
  }

  // ---- K/V staging: thread handles chunks id = tid + i*NT (row = id / CPR, c = id % CPR)
  // A chunk past the runtime head dimension must read as zero in EVERY tile: its offset is parked
  // just below 2^32 and the per-tile stride is applied with a saturating add, so it can never
  // wrap back into the buffer.
  uint32_t koff[NCH], voff[NCH], klds[NCH], vlds[NCH];
  const uint32_t kinc = BC * ldk2, vinc = BC * ldv2;   // wave-uniform (SGPRs)
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = tid + i * NT;
    const int row = id / CPR, c = id % CPR;
    const bool valid = c * 8 < Dr;
    koff[i] = valid ? row * ldk2 + c * 16 : OOB;
    voff[i] = valid ? row * ldv2 + c * 16 : OOB;
    klds[i] = row * ROWB + kswz<D>(row, c) * 16;
    vlds[i] = TILE + ((c >> 2) * BC + row) * 64 + (c & 3) * 16;   // V image [D/32][64 keys][32 d]
  }
  u32x4 kreg[NCH], vreg[NCH];
  auto issue_loads = [&]() {   // loads the NEXT tile in sequence, then advances the offsets
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(kres, koff[i], 0, 0);
      vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vres, voff[i], 0, 0);
      koff[i] = __builtin_elementwise_add_sat(koff[i], kinc);
      voff[i] = __builtin_elementwise_add_sat(voff[i], vinc);
    }
  };
  auto write_tiles = [&](int stage) {
    char *base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      *reinterpret_cast<u32x4 *>(base + klds[i]) = kreg[i];
      *reinterpret_cast<u32x4 *>(base + vlds[i]) = vreg[i];
    }
  };

  const int n16 = lane & 15;
  const int vtr_off = TILE + ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;
  int kread[NKS];   // per-lane K fragment offsets inside a stage for key block 0 (block 1: + 32 rows)
#pragma unroll
  for (int t = 0; t < NKS; ++t) kread[t] = q * ROWB + kswz<D>(q, 2 * t + hi) * 16;
  // rows q and q+32 have the same swizzle (32 is a multiple of every swizzle period)

  auto qk = [&](int stage, f32x16 (&s)[RB][2]) {
    const char *Ks = smem + stage * STAGE;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < NKS; ++t) {
        const v8 kf = *reinterpret_cast<const v8 *>(Ks + kb * 32 * ROWB + kread[t]);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          if (t == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[b][kb][r] = 0.f;
          }
          s[b][kb] = F::mfma(kf, qf[b][t], s[b][kb]);
        }
      }
  };

  f32x16 o[RB][NDB];
  float m[RB], l[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    m[b] = -3.402823466e+38f;   // +Caching.swift:310
    l[b] = 0.f;                 // the reference starts at denorm_min (+Caching.swift:311); see epilogue
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[b][db][r] = 0.f;
  }

  f32x16 lsum[MSUM ? RB : 1];
  v8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (T)1.0f;
  if constexpr (MSUM) {
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) lsum[b][r] = 0.f;
  }

  // ---- online softmax, split in three so that the only branch of the loop sits at its top ------
  // (+Softmax.swift:228-324, :406-417)
  auto mask_edge = [&](f32x16 (&s)[RB][2], int c0) {   // maskAttentionMatrixEdge
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (c0 + 32 * kb + crow(r, hi) >= C) s[b][kb][r] = mask_value();
  };
  auto block_max = [&](const f32x16 (&s)[RB][2], float (&m_new)[RB]) {   // onlineReduceMaximum
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      float mx0 = fmaxf(s[b][0][0], s[b][0][1]), mx1 = fmaxf(s[b][1][0], s[b][1][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) {   // fmaxf(fmaxf(a, b), c) -> one v_max3_f32
        mx0 = fmaxf(fmaxf(mx0, s[b][0][r]), s[b][0][r + 1]);
        mx1 = fmaxf(fmaxf(mx1, s[b][1][r]), s[b][1][r + 1]);
      }
      m_new[b] = half_swap_max(fmaxf(mx0, mx1)) * a.scale2;
    }
  };
  auto rescale_if_needed = [&](const float (&m_new)[RB]) {   // onlineCorrectO, deferred by THR
    bool need = false;
#pragma unroll
    for (int b = 0; b < RB; ++b) need |= (m_new[b] > m[b] + (float)THR);
    if (__builtin_amdgcn_ballot_w64(need) != 0) {   // wave-uniform, rare after the first tiles
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const float m_up = fmaxf(m[b], m_new[b]);
        const float corr = fast_exp2(m[b] - m_up);
        m[b] = m_up;
        l[b] *= corr;
        if constexpr (MSUM) {
#pragma unroll
          for (int r = 0; r < 16; ++r) lsum[b][r] *= corr;
        }
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[b][db][r] *= corr;
      }
    }
  };
  auto exponentiate = [&](f32x16 (&s)[RB][2], v8 (&pf)[RB][4]) {   // softmax + onlineReduceSum
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const float mb = m[b];
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(s[b][kb][r] * a.scale2 - mb);
          s[b][kb][r] = p;
          if constexpr (!MSUM) ps[r & 3] += p;
        }
      if constexpr (!MSUM) l[b] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {   // MFMA step u uses registers 8*(u&1)..+7 of key block u>>1
        v8 pk;
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = (T)s[b][u >> 1][8 * (u & 1) + i];
        pf[b][u] = pk;
      }
    }
  };

  auto pv = [&](int stage, const v8 (&pf)[RB][4]) {
    const char *Vs = smem + stage * STAGE + vtr_off;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const char *vp = Vs + (db * BC + 16 * u) * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + 8 * 64));
        const v8 vf = __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int b = 0; b < RB; ++b) o[b][db] = F::mfma(vf, pf[b][u], o[b][db]);
      }
    if constexpr (MSUM) {   // onlineReduceSum on the matrix pipe: l = sum_k 1 * P (every output row equal)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int b = 0; b < RB; ++b) lsum[b] = F::mfma(ones, pf[b][u], lsum[b]);
    }
  };

  // ---- prologue: tile 0 -> stage 0, loads of tile 1 in flight, S(0) and its block maximum
  const int ntiles = (C + BC - 1) / BC;
  const bool ragged = (C & (BC - 1)) != 0;
  issue_loads();
  write_tiles(0);
  issue_loads();
  __syncthreads();
  f32x16 s_cur[RB][2], s_next[RB][2];
  v8 pf[RB][4];
  float m_new[RB];
  qk(0, s_cur);
  if (ntiles == 1 && ragged) mask_edge(s_cur, 0);
  block_max(s_cur, m_new);

  int st_cur = 0, st_next = 1;
  auto iteration = [&](int j, bool last) {
    rescale_if_needed(m_new);
    write_tiles(st_next);        // tile j+1 (replaces tile j-2)
    issue_loads();               // tile j+2 (reads as zero past the end)
    __syncthreads();
    qk(st_next, s_next);
    exponentiate(s_cur, pf);
    pv(st_cur, pf);
    if (last && ragged) mask_edge(s_next, (j + 1) * BC);
    block_max(s_next, m_new);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s_cur[b][kb] = s_next[b][kb];
    st_cur = st_next;
    st_next = (st_next == 2) ? 0 : st_next + 1;
  };
  int j = 0;
  for (; j + 2 < ntiles; ++j) iteration(j, false);
  if (j + 1 < ntiles) iteration(j, true);
  rescale_if_needed(m_new);
  exponentiate(s_cur, pf);
  pv(st_cur, pf);

  // ---- epilogue: O /= l (+Source.swift:165-171), L = m + log2(l) (+Caching.swift:373-377)
  __syncthreads();   // every wave is done with the ring
  constexpr int OLD = D + 4;   // padded row (floats)
  float *Os = reinterpret_cast<float *>(smem) + wave * (RB * 32 * OLD);
  char *lbase = operand_base(a.op[SLOT_L], head, batch);
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const float l_tot = (MSUM ? lsum[b][0] : half_swap_add(l[b])) + 1.401298464e-45f;
    const float inv = 1.0f / l_tot;
    float *orow = Os + (b * 32 + q) * OLD;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) =
            make_float4(o[b][db][4 * g] * inv, o[b][db][4 * g + 1] * inv, o[b][db][4 * g + 2] * inv, o[b][db][4 * g + 3] * inv);
    const int64_t row = r0 + b * 32 + q;
    if (hi == 0 && row < R) store_elem(lbase, row, a.op[SLOT_L].precision, m[b] + log2f(l_tot));
  }
  // each wave reads back its own rows: no workgroup barrier needed, only the LDS write->read wait
  const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_O], head, batch), 0, (uint32_t)R * (uint32_t)a.op[SLOT_O].ld * 4u, 0x00020000);
  const uint32_t ldo4 = (uint32_t)a.op[SLOT_O].ld * 4;
  constexpr int CPRO = D / 4;              // 16-byte chunks per fp32 output row
#pragma unroll
  for (int i = 0; i < RB * 32 * CPRO / 64; ++i) {
    const int id = lane + i * 64;
    const int rr = id / CPRO, c = id % CPRO;
    const float4 val = *reinterpret_cast<const float4 *>(Os + rr * OLD + c * 4);
    const int64_t row = r0 + rr;
    const uint32_t off = (row < R && c * 4 < Dr) ? (uint32_t)row * ldo4 + c * 16 : OOB;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), ores, off, 0, 0);
  }
}

} // namespace mfa
