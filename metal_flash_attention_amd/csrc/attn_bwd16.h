// attn_bwd16.h -- backward attention on the 16-bit matrix cores of gfx950 (BF16 or FP16 Q, K, V, dO).
//
//   backwardQuery    : D = rowsum(dO o O)/sqrt(D);  dQ = sum_c dS K      (+Source.swift:202-242)
//   backwardKeyValue : dV = sum_r P^T dO;  dK = sum_r dS^T Q             (+Source.swift:244-293)
//   with  P = exp2(S*scale2 - L),  dS = P o (dP/sqrt(D) - D)             (+Softmax.swift:406-427)
// reference: Sources/FlashAttention/Attention/AttentionKernel/AttentionKernel+Source.swift,
//            +OuterProduct.swift, +Accumulate.swift, +Softmax.swift:32-221, +Caching.swift:333-413.
//
// The two kernels stay separate and each recomputes S and P from L, exactly as the reference does to
// avoid FP32 atomics (README.md:11, :39-46): 7 GEMMs instead of 5.
//
// Building blocks are those of attn_fwd16.h: v_mfma_f32_32x32x16, the first product of each pair is
// oriented so that a lane owns one reduction row (dQ: S^T = K Q^T, lane = query; dK/dV: S = Q K^T,
// lane = key, which is the orientation the reference uses too, +Source.swift:245-249), and the
// contraction index of the second product is permuted so the C/D registers of the first ARE its B
// operand.  An operand that is needed both as a row fragment (A of a "outer product") and transposed
// (A of an "accumulate") gets two LDS images: row-major with the 16-byte XOR swizzle (ds_read_b128)
// and [D/32][rows][32] (ds_read_b64_tr_b16).  That is K in backwardQuery and Q, dO in
// backwardKeyValue; each staged chunk is simply written twice.
#pragma once
#include "attn_fwd16_common.h"

namespace mfa {

// dO may be stored in a different 16-bit type than Q, K, V: the reference's low-precision mode keeps
// Q, K, V in FP16 and dO in BF16 (AttentionDescriptor+Precisions.swift:13-17).  One MFMA needs both
// operands in one type, so such a chunk is converted (through fp32, exact for BF16 -> fp32) when loaded.
template <typename T, typename TG>
__device__ __forceinline__ typename Frag16<T>::v8 convert_chunk(u32x4 raw) {
  typedef typename Frag16<T>::v8 v8;
  if constexpr (sizeof(T) == sizeof(TG) && __is_same(T, TG)) {
    return __builtin_bit_cast(v8, raw);
  } else {
    const typename Frag16<TG>::v8 src = __builtin_bit_cast(typename Frag16<TG>::v8, raw);
    v8 out;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (T)(float)src[i];
    return out;
  }
}

// ------------------------------------------------------------------------------------------------
// backwardQuery.  Workgroup = NW waves x 32 query rows; traversal over 64-key tiles.
// LDS stage = K | V.  K is needed both as row fragments (A of S^T = K Q^T) and transposed (A of dQ^T += K^T dS^T):
// ONE image serves both, [D/32][64 keys][32 elements] with the four 16-byte chunks of each 64-byte row
// XOR-swizzled by (key >> 2) & 3 -- conflict-free for ds_read_b128 (16 lanes of a read group land on 16
// distinct slots) and for ds_read_b64_tr_b16 (every lane supplies its own address, so the swizzle folds into
// two lane constants), the layout attn_dkv16_rs.h uses for Q and dO.  V is only read as row fragments and
// keeps the swizzled row-major image.  (An earlier version kept two K images: 3 tiles per stage, which did not
// fit D = 256.)  D <= 128: 8 waves (two per SIMD); D = 256: 4 waves with 512 registers each.
// ------------------------------------------------------------------------------------------------
// (BC: keys per tile, 64; 32 for the head blocks above 256 -- round 6 -- whose epilogue also goes in two halves: EPW waves at a time)
template <int D, int NW> constexpr int dq16_epilogue_waves() { return NW * 32 * (D + 4) * 4 > 160 * 1024 ? NW / 2 : NW; }
template <int D, int NW, int BC = 64> constexpr int dq16_lds_bytes() {
  constexpr int ring = 2 * (BC * D * 2 + BC * (D * 2 + (BC == 32 ? 16 : 0)));
  constexpr int epi = dq16_epilogue_waves<D, NW>() * 32 * (D + 4) * 4;
  return ring > epi ? ring : epi;
}

// SPLIT (traversal-parallel launch for grids that cannot fill the GPU, e.g. the reference's single-head
// benchmark shape): the key range is cut into grid.splits pieces, each workgroup leaves its partial dQ in its
// own fp32 slab of the caller's workspace ([split][head x batch][R][D]) and attn_bwd_combine adds the slabs --
// no atomics, like the reference's refusal of an atomic dQ (README.md:11).  D is written by piece 0.
template <typename T, int D, int NW, typename TG = T, bool CAUSAL = false, bool SPARSE = false, bool SPLIT = false, int BC_ = 64>
__global__ __launch_bounds__(NW * 64) void attn_dq16(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BC = BC_, NKB = BC / 32, NT = NW * 64, NDB = D / 32, NKS = D / 16;
  static_assert(BC == 64 || (BC == 32 && !SPARSE && !SPLIT), "32-key tiles: dense / causal / per-batch lengths only (the mask's column blocks are two 64-key tiles)");
  // (BC = 32, the head blocks above 256 -- round 6, profiles/r06_fwdbwd_bf16_d384_mixed_summary_before_lds_fix.txt: 46 % of the LDS cycles
  // were bank conflicts.  V rows of 640 / 768 bytes start on the same banks (768 = 3 x 256) and the four-chunk XOR cannot spread sixteen
  // of them: rows padded by one 16-byte chunk instead (41 / 49 chunks: sixteen consecutive rows start on sixteen different slots), no
  // swizzle.  The K image's staging writes put sixteen consecutive chunks of ONE row on four slots (its d-blocks are 2048 bytes apart):
  // a group of sixteen lanes now writes four rows x the four chunks of one d-block.)
  constexpr bool WIDE = BC == 32;
  constexpr int ROWB = D * 2 + (WIDE ? 16 : 0), TILE = BC * D * 2, STAGE = TILE + BC * ROWB;
  constexpr int CPR = D / 8, NCH = BC * CPR / NT;
  static_assert(BC * CPR % NT == 0, "tile must divide evenly over the workgroup");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  uint32_t bid = blockIdx.x, split = 0;
  if constexpr (SPLIT) { split = bid % grid.splits; bid /= grid.splits; }
  fwd16_decode_block(grid, bid, &rblk, &head, &batch);
  if constexpr (CAUSAL) rblk = grid.rowBlocks - 1 - rblk;   // later row blocks traverse more keys: start them first
  int R = a.R, C = a.C;
  const int Dr = a.D;
  batch_lengths(a, batch, R, C);
  if (!SPLIT && (int64_t)rblk * (NW * 32) >= R) return;   // padded batch entry: the whole workgroup lies beyond its rows
  const int64_t r0 = (int64_t)rblk * (NW * 32) + wave * 32;
  const int64_t row = r0 + q;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2,
                 ldv2 = (uint32_t)a.op[SLOT_V].ld * 2, ldg2 = (uint32_t)a.op[SLOT_dO].ld * 2,
                 ldo4 = (uint32_t)a.op[SLOT_O].ld * 4;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)R * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_dO], head, batch), 0, (uint32_t)R * ldg2, 0x00020000);
  const bool o32 = a.op[SLOT_O].precision == PREC_FP32;
  const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_O], head, batch), 0, (uint32_t)R * (o32 ? ldo4 : ldo4 / 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_K], head, batch), 0, (uint32_t)C * ldk2, 0x00020000);
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_V], head, batch), 0, (uint32_t)C * ldv2, 0x00020000);

  // ---- cached left-hand operands: Q and dO fragments (B operands), +Caching.swift:316-346
  v8 qf[NKS], gf[NKS];
  float dterm = 0.f;
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const int d0 = 16 * s + 8 * hi;
    const bool ok = d0 < Dr && row < R;
    qf[s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(qres, ok ? (uint32_t)row * ldq2 + d0 * 2 : OOB, 0, 0));
    gf[s] = convert_chunk<T, TG>(__builtin_amdgcn_raw_buffer_load_b128(gres, ok ? (uint32_t)row * ldg2 + d0 * 2 : OOB, 0, 0));
    // computeD (+Softmax.swift:32-221): D = sum_d dO*O, the two half-waves split the head dimension
    if (o32) {
      const uint32_t ooff = ok ? (uint32_t)row * ldo4 + d0 * 4 : OOB;
      const f32x4 o0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ores, ooff, 0, 0));
      const f32x4 o1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ores, ok ? ooff + 16 : OOB, 0, 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) dterm += (float)gf[s][i] * o0[i] + (float)gf[s][4 + i] * o1[i];
    } else {   // O stored in the inputs' 16-bit type (fused output cast of the forward kernel)
      const v8 o8 = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(ores, ok ? (uint32_t)row * (ldo4 / 2) + d0 * 2 : OOB, 0, 0));
#pragma unroll
      for (int i = 0; i < 8; ++i) dterm += (float)gf[s][i] * (float)o8[i];
    }
  }
  dterm = half_swap_add(dterm) * a.scale;
  float Lrow = 0.f;
  if (row < R) Lrow = load_elem(operand_base(a.op[SLOT_L], head, batch), row, a.op[SLOT_L].precision);

  // ---- K/V staging
  uint32_t koff[NCH], voff[NCH], klds[NCH], vlds[NCH];
  const uint32_t kinc = BC * ldk2, vinc = BC * ldv2;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = tid + i * NT;
    const int rr = id / CPR, c = id % CPR;
    const bool valid = c * 8 < Dr;
    koff[i] = valid ? rr * ldk2 + c * 16 : OOB;
    voff[i] = valid ? rr * ldv2 + c * 16 : OOB;
    klds[i] = ((c >> 2) * BC + rr) * 64 + (((c & 3) ^ ((rr >> 2) & 3)) * 16);   // K: [D/32][64][32], chunks swizzled
    vlds[i] = TILE + rr * ROWB + kswz<D>(rr, c) * 16;                           // V: row-major (swizzled)
    if constexpr (WIDE) {
      vlds[i] = TILE + rr * ROWB + c * 16;
      // K: a wave = eight keys x a pair of d-blocks (128 contiguous bytes of a row per load), sixteen lanes = four keys x one d-block's chunks
      const int l = id & 63, unit = id >> 6, dbp = unit % (NDB / 2), rg = unit / (NDB / 2), c4 = l & 3;
      const int kr = 8 * rg + 4 * (l >> 5) + ((l >> 2) & 3), kc = 4 * (2 * dbp + ((l >> 4) & 1)) + c4;
      koff[i] = kc * 8 < Dr ? kr * ldk2 + kc * 16 : OOB;
      klds[i] = ((kc >> 2) * BC + kr) * 64 + ((c4 ^ ((kr >> 2) & 3)) * 16);
    }
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
  // SPARSE (block-mask extension): tiles are not consecutive, so the offsets are those of tile 0 plus tile * pitch
  auto issue_loads_at = [&](int tile) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(kres, __builtin_elementwise_add_sat(koff[i], (uint32_t)tile * kinc), 0, 0);
      vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vres, __builtin_elementwise_add_sat(voff[i], (uint32_t)tile * vinc), 0, 0);
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
  constexpr int WEVERY = 2 * NDB / NCH;   // matrix instructions between two staging writes in the main loop
  static_assert(NCH <= 2 * NDB && (2 * NDB) % NCH == 0, "staging chunks must spread evenly over the dQ accumulation");
  auto write_chunk = [&](int stage, int i) {
    char *base = smem + stage * STAGE;
    *reinterpret_cast<u32x4 *>(base + klds[i]) = kreg[i];
    *reinterpret_cast<u32x4 *>(base + vlds[i]) = vreg[i];
  };
  const int n16 = lane & 15;
  // transposing reads of K: rows (n16 >> 2) + 4 hi and + 8 of a 16-key group; this lane's 8-byte piece is number
  // (n16 & 3) of 32-byte half (lane >> 4) & 1 of the row; (row >> 2) & 3 == hi resp. (hi + 2) & 3
  const int trow = (n16 >> 2) + 4 * hi, tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1), thalf = (n16 & 3) & 1;
  const int tr0 = trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8;
  const int tr1 = (trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8;
  int fread[NKS], kfread[NKS];   // row-fragment offsets of key q of a 32-key block: V (row-major) and K (blocked image)
#pragma unroll
  for (int t = 0; t < NKS; ++t) {
    fread[t] = q * ROWB + (WIDE ? 2 * t + hi : kswz<D>(q, 2 * t + hi)) * 16;
    kfread[t] = ((t >> 1) * BC + q) * 64 + (((2 * (t & 1) + hi) ^ ((q >> 2) & 3)) * 16);
  }
  // (head blocks above 256, BC = 32: the 2 x NKS offsets are TWO lane values per image plus compile-time terms -- the XOR of the
  // swizzle touches the low two chunk bits only -- which hipcc does not find in the tables above: 43 spilled registers at 384)
  const int fr_e = fread[0], fr_o = fread[1], kfr_e = kfread[0], kfr_o = kfread[1];
  auto vfrag_off = [&](int t) { if constexpr (BC == 32) return fr_e + t * 32; else return fread[t]; };   // (no swizzle there: chunk 2 t + hi)
  auto kfrag_off = [&](int t) { if constexpr (BC == 32) return ((t & 1) ? kfr_o : kfr_e) + (t >> 1) * BC * 64; else return kfread[t]; };

  f32x16 dq[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

  // CAUSAL (extension): keys past the last row's limit are never visited; blocks crossing the diagonal
  // get P = 0 element-wise (hence dS = 0)
  const int coff = causal_offset(R, C);
  int ntiles = (C + BC - 1) / BC;
  if constexpr (CAUSAL) {
    const int64_t last_row = min((int64_t)R, ((int64_t)rblk + 1) * (NW * 32)) - 1;
    ntiles = (int)min((int64_t)ntiles, (last_row + coff) / BC + 1);
  }
  // block mask: bit (row block of 256 rows = this workgroup, column block of 128 keys = two tiles)
  const uint32_t *mrow = nullptr;
  if constexpr (SPARSE)
    mrow = a.mask + (int64_t)head * a.maskHeadStride + (int64_t)batch * a.maskBatchStride +
           (uint64_t)(((uint64_t)rblk * (NW * 32)) >> 8) * a.maskWords;
  auto next_active = [&](int t) {   // first active tile >= t (ntiles if none); dense: t itself
    if constexpr (SPARSE) {
      while (t < ntiles && !((mrow[(t >> 1) >> 5] >> ((t >> 1) & 31)) & 1u)) ++t;
    }
    return t;
  };
  int tile_lo = 0;
  if constexpr (SPLIT) {   // this workgroup's piece of the (visible) key tiles
    static_assert(!(SPLIT && SPARSE), "masked launches are not split");
    tile_lo = (int)((uint64_t)split * ntiles / grid.splits);
    ntiles = (int)((uint64_t)(split + 1) * ntiles / grid.splits);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      koff[i] = __builtin_elementwise_add_sat(koff[i], (uint32_t)tile_lo * kinc);
      voff[i] = __builtin_elementwise_add_sat(voff[i], (uint32_t)tile_lo * vinc);
    }
  }
  int j = next_active(tile_lo), stage = 0;
  if (j < ntiles) {
    if constexpr (SPARSE) issue_loads_at(j); else issue_loads();
    write_tiles(0);
  }
  __syncthreads();
  // Q / dO fragments, L and D have no consumer before the loop: without this wait hipcc waits for them in front of
  // the first matrix instructions INSIDE the loop, with counts that in steady state also drain the prefetch of the
  // next tile issued a few instructions earlier (vmcnt counts in order).  s_waitcnt vmcnt(0) (expcnt, lgkmcnt free).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  while (j < ntiles) {
    const char *st = smem + stage * STAGE;
    const int jn = next_active(j + 1);
    if (jn < ntiles) { if constexpr (SPARSE) issue_loads_at(jn); else issue_loads(); }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      // S^T = K Q^T and dP^T = V dO^T for 32 keys: lane = query row, registers = keys crow(r, hi)
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      // the two accumulation chains are independent: alternating them keeps consecutive MFMAs off
      // the same accumulator (a non-MFMA instruction between two dependent MFMAs costs ~43 cycles)
      // the fragments of k-step t + 1 are requested before the matrix instructions of step t; sched_barrier(0x406)
      // holds that order (LDS and matrix instructions stay, vector / scalar work may move)
      v8 kfa[2], vfa[2];
      kfa[0] = *reinterpret_cast<const v8 *>(st + kb * 32 * 64 + kfrag_off(0));
      vfa[0] = *reinterpret_cast<const v8 *>(st + TILE + kb * 32 * ROWB + vfrag_off(0));
#pragma unroll
      for (int t = 0; t < NKS; ++t) {
        if (t + 1 < NKS) {
          kfa[(t + 1) & 1] = *reinterpret_cast<const v8 *>(st + kb * 32 * 64 + kfrag_off(t + 1));
          vfa[(t + 1) & 1] = *reinterpret_cast<const v8 *>(st + TILE + kb * 32 * ROWB + vfrag_off(t + 1));
          __builtin_amdgcn_sched_barrier(0x406);
        }
        s = F::mfma(kfa[t & 1], qf[t], s);
        dp = F::mfma(vfa[t & 1], gf[t], dp);
      }
      // P = exp2(S*scale2 - L); dS = P (dP*scale - D).  Keys past C have zero K and V rows, so their
      // dS multiplies zero K rows below: no mask needed (as in the reference, +Accumulate.swift:330-346).
      v8 dsf[2];
      const int c0 = j * BC + 32 * kb;
      const bool diag = CAUSAL && (c0 + 31 > r0 + coff);   // wave-uniform: this block crosses the diagonal
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        v8 pk;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = 8 * u + i;
          float p = fast_exp2(s[r] * a.scale2 - Lrow);
          if (diag && c0 + crow(r, hi) > row + coff) p = 0.f;
          pk[i] = (T)(p * (dp[r] * a.scale - dterm));
        }
        dsf[u] = pk;
      }
      // dQ^T += K^T dS^T (K transposed by the LDS read; key index permuted as in forward).  The staging writes of
      // the next tile go between the matrix instructions of the second half: issued as one burst in front of the
      // barrier they keep the LDS busy for ~800 cycles (D = 256) during which no wave has anything to run.  The
      // other stage has no readers in this iteration; past the last tile the registers hold stale data that
      // nobody reads.
      // The transposed fragment of product idx + 1 is requested before the matrix instruction of product idx.
      auto read_kt = [&](int idx) -> v8 {   // idx = u * NDB + db
        const char *kp = st + ((idx % NDB) * BC + 32 * kb + 16 * (idx / NDB)) * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(kp + tr0));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(kp + tr1));
        return __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      v8 ktf[2];
      ktf[0] = read_kt(0);
#pragma unroll
      for (int idx = 0; idx < 2 * NDB; ++idx) {
        if (idx + 1 < 2 * NDB) ktf[(idx + 1) & 1] = read_kt(idx + 1);
        if (kb == NKB - 1 && idx % WEVERY == 0) write_chunk(stage ^ 1, idx / WEVERY);
        __builtin_amdgcn_sched_barrier(0x406);
        dq[idx % NDB] = F::mfma(ktf[idx & 1], dsf[idx / NDB], dq[idx % NDB]);
      }
    }
    __syncthreads();
    stage ^= 1;
    j = jn;
  }

  // ---- epilogue: dQ through LDS (whole-row stores); D written pre-scaled (+Caching.swift:381-413)
  constexpr int OLD = D + 4, EPW = dq16_epilogue_waves<D, NW>();
  if ((!SPLIT || split == 0) && hi == 0 && row < R)
    store_elem(operand_base(a.op[SLOT_D], head, batch), row, a.op[SLOT_D].precision, dterm);
  float *Os = reinterpret_cast<float *>(smem) + (wave % EPW) * (32 * OLD);
  float *orow = Os + q * OLD;
#pragma unroll
  for (int pass = 0; pass < NW / EPW; ++pass) {   // (head blocks above 256: the staging rows of all waves do not fit the LDS at once)
    if (pass) __syncthreads();
    if (wave / EPW == pass) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) =
              make_float4(dq[db][4 * g], dq[db][4 * g + 1], dq[db][4 * g + 2], dq[db][4 * g + 3]);
      if constexpr (SPLIT) {
        const size_t slab = ((size_t)split * grid.heads * grid.batches + (size_t)batch * grid.heads + head) * (size_t)a.R;
        store_block_rows<T, D>(Os, reinterpret_cast<char *>(grid.wsO + slab * Dr), PREC_FP32, (uint32_t)Dr, r0, R, Dr, lane);
      } else {
        store_block_rows<T, D>(Os, operand_base(a.op[SLOT_dQ], head, batch), a.op[SLOT_dQ].precision, (uint32_t)a.op[SLOT_dQ].ld,
                               r0, R, Dr, lane);
      }
    }
  }
}

// Sum of the partial results of a traversal-parallel backward launch: out[row][d] = sum over pieces of
// ws[piece][head x batch][row][d].  One wave per row, lane c owns elements 4c..4c+3.  `slot` = destination operand
// (dQ, dK or dV), `rows` = its sequence length, `ws` = first slab of that operand.  HBM-bound.
static __global__ __launch_bounds__(256) void attn_bwd_combine(const KernelArgs a, const Fwd16Grid grid, int slot, uint32_t rows,
                                                               const float *ws) {
  const int lane = threadIdx.x & 63;
  const uint32_t Dr = a.D, HB = grid.heads * grid.batches, S = grid.splits;
  const uint64_t rowid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // over HB * rows
  if (rowid >= (uint64_t)HB * rows) return;
  const uint32_t hb = (uint32_t)(rowid / rows), row = (uint32_t)(rowid % rows);
  const uint32_t head = hb % grid.heads, batch = hb / grid.heads;
  // (round 5, as attn_fwd_combine: latency-bound) CL = D / 4 column lanes x 64 / CL piece groups: every lane walks splits / groups
  // pieces with independent loads, the groups are summed with half-wave exchanges
  uint32_t CL = 1;
  while (CL * 4 < Dr) CL <<= 1;                 // (D <= 256: at most 64 column lanes)
  const uint32_t G = 64 / CL, c = (uint32_t)lane % CL, g = (uint32_t)lane / CL;
  const bool active = c * 4 < Dr;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    for (uint32_t s = g; s < S; s += G) {
      const float4 v = *reinterpret_cast<const float4 *>(ws + (((uint64_t)s * HB + hb) * rows + row) * Dr + c * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  for (uint32_t off = CL; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, (int)off, 64); acc.y += __shfl_xor(acc.y, (int)off, 64);
    acc.z += __shfl_xor(acc.z, (int)off, 64); acc.w += __shfl_xor(acc.w, (int)off, 64);
  }
  if (!active || g != 0) return;
  char *obase = operand_base(a.op[slot], head, batch);
  const int64_t idx = (int64_t)row * a.op[slot].ld + c * 4;
  const int prec = a.op[slot].precision;
  if (prec == PREC_FP32) {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(obase) + idx) = acc;
  } else {
    store_elem(obase, idx, prec, acc.x);
    store_elem(obase, idx + 1, prec, acc.y);
    store_elem(obase, idx + 2, prec, acc.z);
    store_elem(obase, idx + 3, prec, acc.w);
  }
}

// ------------------------------------------------------------------------------------------------
// backwardKeyValue.  Workgroup = NW waves x 32 key columns (one wave per SIMD: K, V fragments and the
// dK, dV accumulators of 32 keys x D take ~200 of the 512 registers); traversal over 32-row tiles of
// Q / dO in a 3-stage LDS ring, one barrier per tile.
// LDS stage = Q row-major | Q transposable | dO row-major | dO transposable | L[32] | D[32].
// Software pipeline (as attn_fwd16_v3.h): S and dP of row block j+1 run on the matrix pipe while the
// VALU turns block j into P and dS; then dV += dO^T P and dK += Q^T dS of block j.  The two
// (S, dP) register sets alternate, so the loop body holds two blocks and no register copies.
// A block past the end of R is all zeros (bounds-checked loads) and contributes exactly nothing
// (P = 1 meets dO = 0, dS = 0), so the block count is simply rounded up to even.
// ------------------------------------------------------------------------------------------------
template <int D, int NW> constexpr int dkv16_lds_bytes() {
  constexpr int ring = 3 * (4 * 32 * D * 2 + 256);
  constexpr int epi = NW * 32 * (D + 4) * 4;
  return ring > epi ? ring : epi;
}

template <typename T, int D, int NW, int PRE = 1, typename TG = T, bool CAUSAL = false>
__global__ __launch_bounds__(NW * 64) void attn_dkv16(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BR = 32, NT = NW * 64, NDB = D / 32, NKS = D / 16;
  constexpr int ROWB = D * 2, TILE = BR * D * 2, STAGE = 4 * TILE + 256;
  constexpr int CPR = D / 8, NCH = BR * CPR / NT;
  static_assert(BR * CPR % NT == 0 && NCH >= 1, "tile must divide evenly over the workgroup");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, kc = lane & 31, hi = lane >> 5;
  uint32_t cblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &cblk, &head, &batch);
  int R = a.R, C = a.C;
  const int Dr = a.D;
  batch_lengths(a, batch, R, C);
  if ((int64_t)cblk * (NW * 32) >= C) return;   // padded batch entry: the whole workgroup lies beyond its keys
  const int64_t c0 = (int64_t)cblk * (NW * 32) + wave * 32;
  const int64_t col = c0 + kc;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2,
                 ldv2 = (uint32_t)a.op[SLOT_V].ld * 2, ldg2 = (uint32_t)a.op[SLOT_dO].ld * 2;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)R * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_dO], head, batch), 0, (uint32_t)R * ldg2, 0x00020000);
  const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_K], head, batch), 0, (uint32_t)C * ldk2, 0x00020000);
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_V], head, batch), 0, (uint32_t)C * ldv2, 0x00020000);
  const char *lbase = operand_base(a.op[SLOT_L], head, batch);
  const char *dbase = operand_base(a.op[SLOT_D], head, batch);

  // ---- cached left-hand operands: K and V fragments (B operands), +Caching.swift:316-321
  v8 kf[NKS], vf[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const int d0 = 16 * s + 8 * hi;
    const bool ok = d0 < Dr && col < C;
    kf[s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(kres, ok ? (uint32_t)col * ldk2 + d0 * 2 : OOB, 0, 0));
    vf[s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(vres, ok ? (uint32_t)col * ldv2 + d0 * 2 : OOB, 0, 0));
  }

  // CAUSAL (extension): rows r with r + (C - R) < (first key of the workgroup) see none of its keys: the
  // traversal starts at the first row block that can; blocks crossing the diagonal get P = 0 element-wise
  const int coff = causal_offset(R, C);
  const int block0 = CAUSAL ? (int)(max((int64_t)0, (int64_t)cblk * (NW * 32) - coff) / 32) : 0;

  // ---- Q / dO staging (two images each) + the L, D slices along the traversal dimension
  uint32_t qoff[NCH], goff[NCH], rlds[NCH], tlds[NCH];
  const uint32_t qinc = BR * ldq2, ginc = BR * ldg2;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = tid + i * NT;
    const int rr = id / CPR, c = id % CPR;
    const bool valid = c * 8 < Dr;
    qoff[i] = valid ? (block0 * BR + rr) * ldq2 + c * 16 : OOB;
    goff[i] = valid ? (block0 * BR + rr) * ldg2 + c * 16 : OOB;
    rlds[i] = rr * ROWB + kswz<D>(rr, c) * 16;                  // row-major (swizzled): Q at +0, dO at +2*TILE
    tlds[i] = TILE + ((c >> 2) * BR + rr) * 64 + (c & 3) * 16;  // transposable: Q at +TILE, dO at +3*TILE
  }
  u32x4 qreg[NCH], greg[NCH];
  float ldreg = 0.f;
  int tile_row0 = block0 * BR;
  auto issue_loads = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      qreg[i] = __builtin_amdgcn_raw_buffer_load_b128(qres, qoff[i], 0, 0);
      greg[i] = __builtin_amdgcn_raw_buffer_load_b128(gres, goff[i], 0, 0);
      qoff[i] = __builtin_elementwise_add_sat(qoff[i], qinc);
      goff[i] = __builtin_elementwise_add_sat(goff[i], ginc);
    }
    if (tid < 64) {   // threads 0-31: L of the tile's rows, 32-63: D   (+Softmax.swift:356-381, :472-503)
      const int rr = tile_row0 + (tid & 31);
      ldreg = 0.f;
      if (rr < R) ldreg = (tid < 32) ? load_elem(lbase, rr, a.op[SLOT_L].precision)
                                     : load_elem(dbase, rr, a.op[SLOT_D].precision);
    }
    tile_row0 += BR;
  };
  auto write_tiles = [&](int stage) {
    char *base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      *reinterpret_cast<u32x4 *>(base + rlds[i]) = qreg[i];
      *reinterpret_cast<u32x4 *>(base + tlds[i]) = qreg[i];
      const u32x4 g16 = __builtin_bit_cast(u32x4, convert_chunk<T, TG>(greg[i]));
      *reinterpret_cast<u32x4 *>(base + 2 * TILE + rlds[i]) = g16;
      *reinterpret_cast<u32x4 *>(base + 2 * TILE + tlds[i]) = g16;
    }
    if (tid < 64) reinterpret_cast<float *>(base + 4 * TILE)[tid] = ldreg;
  };
  const int n16 = lane & 15;
  const int tr_off = ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;
  int fread[NKS];
#pragma unroll
  for (int t = 0; t < NKS; ++t) fread[t] = kc * ROWB + kswz<D>(kc, 2 * t + hi) * 16;

  f32x16 dk[NDB], dv[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  // S = Q K^T and dP = dO V^T for the 32 rows of the tile in `stage`: lane = key column,
  // registers = rows crow(r, hi).  Fragment reads are issued ahead of the MFMAs that use them.
  auto scores = [&](int stage, f32x16 &s, f32x16 &dp) {
    const char *st = smem + stage * STAGE;
    v8 qa[NKS], ga[NKS];
#pragma unroll
    for (int t = 0; t < NKS; ++t) qa[t] = *reinterpret_cast<const v8 *>(st + fread[t]);
#pragma unroll
    for (int t = 0; t < NKS; ++t) ga[t] = *reinterpret_cast<const v8 *>(st + 2 * TILE + fread[t]);
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int t = 0; t < NKS; ++t) {   // alternate the two independent chains
      s = F::mfma(qa[t], kf[t], s);
      dp = F::mfma(ga[t], vf[t], dp);
    }
  };
  // P = exp2(S*scale2 - L), dS = P (dP*scale - D), rounded to the 16-bit type and packed as B operands
  int cur_row0 = block0 * BR;   // first row of the block softmax_grad is working on (advanced by the loop)
  auto softmax_grad = [&](int stage, const f32x16 &s, const f32x16 &dp, v8 (&pf)[2], v8 (&dsf)[2]) {
    const float *Ls = reinterpret_cast<const float *>(smem + stage * STAGE + 4 * TILE) + 4 * hi;
    const bool diag = CAUSAL && (c0 + 31 > cur_row0 + coff);   // wave-uniform
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      v8 pk, dk8;
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        const int g = 2 * u + g2;     // rows 8g + 4hi + {0..3}
        const f32x4 l4 = *reinterpret_cast<const f32x4 *>(Ls + 8 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(Ls + 32 + 8 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * g + i;
          float p = fast_exp2(s[r] * a.scale2 - l4[i]);
          if (diag && col > cur_row0 + crow(r, hi) + coff) p = 0.f;
          pk[4 * g2 + i] = (T)p;
          dk8[4 * g2 + i] = (T)(p * (dp[r] * a.scale - d4[i]));
        }
      }
      pf[u] = pk;
      dsf[u] = dk8;
    }
    cur_row0 += BR;
  };
  // dV^T += dO^T P ; dK^T += Q^T dS  (row index permuted; rows past R have zero Q and dO rows)
  auto accumulate = [&](int stage, const v8 (&pf)[2], const v8 (&dsf)[2]) {
    const char *st = smem + stage * STAGE + tr_off;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const int off = (db * BR + 16 * u) * 64;
        const char *gp = st + 3 * TILE + off;
        const char *qp = st + TILE + off;
        const s16x4 g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(gp));
        const s16x4 g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(gp + 8 * 64));
        const s16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(qp));
        const s16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(qp + 8 * 64));
        dv[db] = F::mfma(__builtin_bit_cast(v8, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7)), pf[u], dv[db]);
        dk[db] = F::mfma(__builtin_bit_cast(v8, __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7)), dsf[u], dk[db]);
      }
  };

  v8 pf[2], dsf[2];
  // One pipeline step with every LDS read of its first half issued up front (one wave per SIMD:
  // nothing else hides LDS latency): row fragments of block j+1 (16 x ds_read_b128) and the
  // transposed fragments for the first 16 rows of block j (16 x ds_read_b64_tr_b16), then the S / dP
  // MFMAs of block j+1 interleaved (by the compiler) with the P / dS arithmetic of block j, then the
  // accumulate MFMAs; the second 16 rows' transposed fragments are requested in between.
  auto fused_step = [&](int st_n, int st_c, f32x16 &s_n, f32x16 &dp_n, const f32x16 &s_c, const f32x16 &dp_c) {
    const char *sn = smem + st_n * STAGE;
    const char *sc = smem + st_c * STAGE + tr_off;
    v8 qa[NKS], ga[NKS], gt[NDB], qt[NDB];
#pragma unroll
    for (int t = 0; t < NKS; ++t) qa[t] = *reinterpret_cast<const v8 *>(sn + fread[t]);
#pragma unroll
    for (int t = 0; t < NKS; ++t) ga[t] = *reinterpret_cast<const v8 *>(sn + 2 * TILE + fread[t]);
    auto load_t = [&](int u) {
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const int off = (db * BR + 16 * u) * 64;
        const char *gp = sc + 3 * TILE + off;
        const char *qp = sc + TILE + off;
        const s16x4 g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(gp));
        const s16x4 g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(gp + 8 * 64));
        const s16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(qp));
        const s16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(qp + 8 * 64));
        gt[db] = __builtin_bit_cast(v8, __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7));
        qt[db] = __builtin_bit_cast(v8, __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    load_t(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) { s_n[r] = 0.f; dp_n[r] = 0.f; }
#pragma unroll
    for (int t = 0; t < NKS; ++t) {   // alternate the two independent chains
      s_n = F::mfma(qa[t], kf[t], s_n);
      dp_n = F::mfma(ga[t], vf[t], dp_n);
    }
    softmax_grad(st_c, s_c, dp_c, pf, dsf);
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      dv[db] = F::mfma(gt[db], pf[0], dv[db]);
      dk[db] = F::mfma(qt[db], dsf[0], dk[db]);
    }
    load_t(1);
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      dv[db] = F::mfma(gt[db], pf[1], dv[db]);
      dk[db] = F::mfma(qt[db], dsf[1], dk[db]);
    }
  };

  const int nblocks = (((R + BR - 1) / BR - block0) + 1) & ~1;   // rounded up to even (zero blocks are harmless)
  issue_loads();
  write_tiles(0);
  issue_loads();
  __syncthreads();
  f32x16 s0, dp0, s1, dp1;
  scores(0, s0, dp0);
  int st_cur = 0, st_next = 1;
  auto advance = [&]() { st_cur = st_next; st_next = (st_next == 2) ? 0 : st_next + 1; };
  for (int j = 0; j < nblocks; j += 2) {
    // block j: uses (s0, dp0); produces (s1, dp1) = scores of block j+1
    write_tiles(st_next);      // tile j+1 (replaces tile j-2, whose readers passed the previous barrier)
    issue_loads();             // tile j+2 (zeros past the end)
    __syncthreads();
    if constexpr (PRE == 1) {
      fused_step(st_next, st_cur, s1, dp1, s0, dp0);
    } else {
      scores(st_next, s1, dp1);
      softmax_grad(st_cur, s0, dp0, pf, dsf);
      accumulate(st_cur, pf, dsf);
    }
    advance();
    // block j+1: uses (s1, dp1); produces (s0, dp0) = scores of block j+2
    write_tiles(st_next);
    issue_loads();
    __syncthreads();
    if constexpr (PRE == 1) {
      fused_step(st_next, st_cur, s0, dp0, s1, dp1);
    } else {
      scores(st_next, s0, dp0);
      softmax_grad(st_cur, s1, dp1, pf, dsf);
      accumulate(st_cur, pf, dsf);
    }
    advance();
  }

  // ---- epilogue: dV then dK through LDS (whole-row stores)
  __syncthreads();
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + wave * (32 * OLD);
  float *orow = Os + kc * OLD;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const int slot = which == 0 ? SLOT_dV : SLOT_dK;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x16 &acc = which == 0 ? dv[db] : dk[db];
        *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) =
            make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
      }
    store_block_rows<T, D>(Os, operand_base(a.op[slot], head, batch), a.op[slot].precision, (uint32_t)a.op[slot].ld, c0, C, Dr, lane);
  }
}

} // namespace mfa
