// attn_fwd16_v4.h -- forward attention, 16-bit matrix cores, ROLE-ALTERNATING waves.
//
// Same math, fragment maps, LDS images, ring, loads and epilogue as attn_fwd16_v3.h (8 waves x 32 query
// rows, 64-key tiles handled as two halves of 32 keys).  What changes is who uses which pipe when.
// In v3 all eight waves run the same instruction stream in phase (one barrier per tile re-aligns
// them), so the two waves that share a SIMD ask for the matrix pipe at the same time and for the VALU
// at the same time: rocprof shows the matrix pipe busy only ~52 % of the time.  Here every wave's
// stream is cut into alternating segments
//     M_h : 8 MFMAs  S^T(h+1) = K(h+1) Q^T   +   8 MFMAs  O^T += V^T(h) P^T(h)        (matrix pipe only)
//     V_h : softmax of S^T(h+1) -> P^T(h+1);  LDS reads of the K(h+2), V(h+1) fragments;
//           on odd h the wave's share of the ring traffic (LDS writes of tile j+2, loads of tile j+3)
// each closed by a workgroup barrier, and waves 4-7 (the SIMD partners of waves 0-3) run ONE SEGMENT
// BEHIND waves 0-3: they pass one extra barrier before their first segment, waves 0-3 one extra after
// their last.  While one wave of a SIMD owns the matrix pipe for 16 back-to-back MFMAs its partner
// exponentiates, reads fragments and stages the ring; then they swap.  (h counts half tiles.)
//
// LDS hazards: M segments never touch LDS, and the V segments of the two groups are never concurrent,
// so every cross-group LDS dependence is separated by a barrier.  Tile T is written in V_{2T-3}
// (slots 4T-5 / 4T-4 for the two groups), first read in V_{2T-2} (slot 4T-3), last read in V_{2T}
// (slot 4T+2): a two-stage ring is already safe, RING = 3 keeps one more tile in flight.
#pragma once
#include "attn_fwd16_v3.h"

namespace mfa {

// OPT bits: 1 = raise the wave's priority inside M segments; 2 = static priority for waves 4-7;
//           4 = PV before QK inside M segments; 8 = fragment reads after the softmax (instead of before)
template <typename T, int D, int NW, int THR, int OPT = 0, int RING = 3, bool SPLIT = false, bool CAUSAL = false>
__global__ __launch_bounds__(NW * 64) void attn_fwd16_v4(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BC = 64, NT = NW * 64, NDB = D / 32, NKS = D / 16;
  constexpr int ROWB = D * 2, TILE = BC * D * 2, STAGE = 2 * TILE;
  constexpr int CPR = D / 8, NCH = BC * CPR / NT;
  static_assert(BC * CPR % NT == 0, "tile must divide evenly over the workgroup");
  static_assert(NW == 8, "two waves per SIMD: the partner of wave w is wave w + 4");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  uint32_t bid = blockIdx.x, split = 0;
  if constexpr (SPLIT) { split = bid % grid.splits; bid /= grid.splits; }
  fwd16_decode_block(grid, bid, &rblk, &head, &batch);
  if constexpr (CAUSAL) rblk = grid.rowBlocks - 1 - rblk;
  int R = a.R, C = a.C;
  const int Dr = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * (NW * 32) + wave * 32;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2,
                 ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)R * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_K], head, batch), 0, (uint32_t)C * ldk2, 0x00020000);
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_V], head, batch), 0, (uint32_t)C * ldv2, 0x00020000);
  constexpr uint32_t OOB = 0xFFFFFF00u;

  // ---- Q fragments (B operand of S^T = K Q^T), in registers for the whole kernel
  v8 qf[NKS];
  {
    const uint32_t rowoff = (uint32_t)(r0 + q) * ldq2;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int d0 = 16 * s + 8 * hi;
      const uint32_t off = (d0 < Dr && r0 + q < R) ? rowoff + d0 * 2 : OOB;
      qf[s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(qres, off, 0, 0));
    }
  }

  // ---- key range (as v3)
  const int tiles_total = (C + BC - 1) / BC;
  const int tile0 = SPLIT ? (int)((uint64_t)split * tiles_total / grid.splits) : 0;
  const int coff = causal_offset(R, C);
  int tiles_visible = tiles_total;
  if constexpr (CAUSAL) {
    const int64_t last_row = min((int64_t)R, ((int64_t)rblk + 1) * (NW * 32)) - 1;
    tiles_visible = (int)min((int64_t)tiles_total, (last_row + coff) / BC + 1);
  }
  const int tile1 = SPLIT ? (int)((uint64_t)(split + 1) * tiles_total / grid.splits) : tiles_visible;

  // ---- ring staging: global -> VGPR -> LDS, every wave moves 1/8 of each tile
  uint32_t koff[NCH], voff[NCH], klds[NCH], vlds[NCH];
  const uint32_t kinc = BC * ldk2, vinc = BC * ldv2;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = tid + i * NT;
    const int row = id / CPR, c = id % CPR;
    const bool valid = c * 8 < Dr;
    koff[i] = valid ? (tile0 * BC + row) * ldk2 + c * 16 : OOB;
    voff[i] = valid ? (tile0 * BC + row) * ldv2 + c * 16 : OOB;
    klds[i] = row * ROWB + kswz<D>(row, c) * 16;
    vlds[i] = TILE + ((c >> 2) * BC + row) * 64 + (c & 3) * 16;
  }
  u32x4 kreg[NCH], vreg[NCH];
  auto issue_loads = [&]() {
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
  int kread[NKS];
#pragma unroll
  for (int t = 0; t < NKS; ++t) kread[t] = q * ROWB + kswz<D>(q, 2 * t + hi) * 16;

  // ---- per-wave state
  f32x16 o[NDB], s;
  float m = -3.402823466e+38f, l = 0.f;   // +Caching.swift:310
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  v8 kf[NKS], vf[2][NDB], pf[2];

  auto seg_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto read_k = [&](int stage, int kb) {
    const char *Ks = smem + stage * STAGE + kb * 32 * ROWB;
#pragma unroll
    for (int t = 0; t < NKS; ++t) kf[t] = *reinterpret_cast<const v8 *>(Ks + kread[t]);
  };
  auto read_v = [&](int stage, int kb) {
    const char *Vs = smem + stage * STAGE + vtr_off + kb * 32 * 64;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const char *vp = Vs + (db * BC + 16 * u) * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + 8 * 64));
        vf[u][db] = __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
      }
  };

  // ---- M segment: matrix pipe only
  auto qk_chain = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NKS; ++t) s = F::mfma(kf[t], qf[t], s);
  };
  auto pv_chain = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < NDB; ++db) o[db] = F::mfma(vf[u][db], pf[u], o[db]);
  };
  auto mseg = [&](bool do_qk, bool do_pv) {
    if constexpr ((OPT & 1) != 0) __builtin_amdgcn_s_setprio(2);
    if constexpr ((OPT & 4) != 0) {
      if (do_pv) pv_chain();
      if (do_qk) qk_chain();
    } else {
      if (do_qk) qk_chain();
      if (do_pv) pv_chain();
    }
    if constexpr ((OPT & 1) != 0) __builtin_amdgcn_s_setprio(0);
    seg_barrier();
  };

  // ---- V segment: softmax of the half tile in `s` (keys c0 .. c0+31) + fragment reads + ring share
  auto softmax_half = [&](int c0) {
    if (c0 + 32 > C) {   // maskAttentionMatrixEdge (+Softmax.swift:228-260), wave-uniform test
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) >= C) s[r] = mask_value();
    }
    if constexpr (CAUSAL) {
      if (c0 + 31 > r0 + coff) {
        const int64_t limit = r0 + q + coff;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (c0 + crow(r, hi) > limit) s[r] = mask_value();
      }
    }
    // onlineReduceMaximum (+Softmax.swift:267-289)
    float mx0 = fmaxf(s[0], s[1]), mx1 = fmaxf(s[2], s[3]);
#pragma unroll
    for (int r = 4; r < 16; r += 4) {
      mx0 = fmaxf(fmaxf(mx0, s[r]), s[r + 1]);
      mx1 = fmaxf(fmaxf(mx1, s[r + 2]), s[r + 3]);
    }
    const float m_new = half_swap_max(fmaxf(mx0, mx1)) * a.scale2;
    // onlineCorrectO (+Softmax.swift:290-306), deferred until the maximum has grown by THR (log2 units)
    if (__builtin_amdgcn_ballot_w64(m_new > m + (float)THR) != 0) {
      const float m_up = fmaxf(m, m_new);
      const float corr = fast_exp2(m - m_up);
      m = m_up;
      l *= corr;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= corr;
    }
    // softmax + onlineReduceSum (+Softmax.swift:308-324, :409-417)
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(s[r] * a.scale2 - m);
      s[r] = p;
      ps[r & 3] += p;
    }
    l += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      v8 pk;
#pragma unroll
      for (int i = 0; i < 8; ++i) pk[i] = (T)s[8 * u + i];
      pf[u] = pk;
    }
  };
  auto vseg = [&](int c0, bool rd_k, int k_stage, int k_kb, int v_stage, int v_kb, bool ring, int w_stage) {
    if constexpr ((OPT & 8) == 0) {
      if (rd_k) read_k(k_stage, k_kb);
      read_v(v_stage, v_kb);
      __builtin_amdgcn_sched_barrier(0);
    }
    softmax_half(c0);
    if constexpr ((OPT & 8) != 0) {
      __builtin_amdgcn_sched_barrier(0);
      if (rd_k) read_k(k_stage, k_kb);
      read_v(v_stage, v_kb);
    }
    if (ring) {
      write_tiles(w_stage);
      issue_loads();
    }
    seg_barrier();
  };

  if constexpr ((OPT & 2) != 0) {
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  }
  // ---- prologue: tile 0 -> stage 0, tile 1 in flight
  const int ntiles = tile1 - tile0;
  issue_loads();
  write_tiles(0);
  issue_loads();
  seg_barrier();
  if (wave >= NW / 2) seg_barrier();   // waves 4-7 run one segment behind their SIMD partners
  read_k(0, 0);
  mseg(true, false);                                              // M_-1: S(0)
  int st = 0, stn = 1 % RING, stw = 2 % RING;                     // stages of tiles j, j+1, j+2
  vseg(tile0 * BC, true, st, 1, st, 0, true, stn);                // V_-1: P(0); K(1), V(0); tile 1 -> LDS
  for (int j = 0; j + 1 < ntiles; ++j) {
    mseg(true, true);                                             // M_2j: S(2j+1), PV(2j)
    vseg((tile0 + j) * BC + 32, true, stn, 0, st, 1, false, 0);   // V_2j: P(2j+1); K(2j+2), V(2j+1)
    mseg(true, true);                                             // M_2j+1: S(2j+2), PV(2j+1)
    vseg((tile0 + j + 1) * BC, true, stn, 1, stn, 0, true, stw);  // V_2j+1: P(2j+2); K(2j+3), V(2j+2); tile j+2 -> LDS
    st = stn;
    stn = stw;
    stw = (stw == RING - 1) ? 0 : stw + 1;
  }
  mseg(true, true);                                               // S(last half), PV(last tile, first half)
  vseg((tile0 + ntiles - 1) * BC + 32, false, 0, 0, st, 1, false, 0);
  mseg(false, true);                                              // PV(last half)
  if (wave < NW / 2) seg_barrier();   // waves 0-3 wait for their partners' last segment: the ring is free

  // ---- epilogue (as v3): O /= l, L = m + log2 l; SPLIT launches publish (O, m, l) of their key range
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + wave * (32 * OLD);
  char *lbase = operand_base(a.op[SLOT_L], head, batch);
  const size_t slab = ((size_t)split * grid.heads * grid.batches + (size_t)batch * grid.heads + head) * (size_t)a.R;
  {
    const float l_tot = half_swap_add(l) + 1.401298464e-45f;
    const float inv = SPLIT ? 1.0f : 1.0f / l_tot;
    float *orow = Os + q * OLD;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) =
            make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
    const int64_t row = r0 + q;
    if (hi == 0 && row < R) {
      if constexpr (SPLIT) {
        grid.wsML[(slab + row) * 2] = m;
        grid.wsML[(slab + row) * 2 + 1] = l_tot;
      } else {
        store_elem(lbase, row, a.op[SLOT_L].precision, m + log2f(l_tot));
      }
    }
  }
  if constexpr (SPLIT)
    store_block_rows<T, D>(Os, reinterpret_cast<char *>(grid.wsO + slab * Dr), PREC_FP32, (uint32_t)Dr, r0, R, Dr, lane);
  else
    store_block_rows<T, D>(Os, operand_base(a.op[SLOT_O], head, batch), a.op[SLOT_O].precision, (uint32_t)a.op[SLOT_O].ld,
                           r0, R, Dr, lane);
}

} // namespace mfa
