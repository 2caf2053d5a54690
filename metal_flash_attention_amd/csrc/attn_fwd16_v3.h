// attn_fwd16_v3.h -- forward attention on the 16-bit matrix cores: the product kernel for bf16 / fp16 Q, K, V.
//
// Same math / fragment maps / LDS images as attn_fwd16.h and the same ring, deferred rescale and epilogue as
// attn_fwd16_v2.h.  What is specific to this kernel:
//   * the pipeline step is HALF a K/V tile (32 keys): S^T of the next 32 keys is produced by the matrix pipe
//     while the VALU exponentiates the current 32, then O^T += V^T P^T for the current 32 while the VALU
//     reduces the next block's maximum.  The two half score tiles swap roles every step, so no register copies
//     are needed and only two half tiles of scores are live;
//   * geometry by template: NW waves x RB blocks of 32 query rows.  Product: D <= 128 -> 8 waves x 32 rows (two
//     waves per SIMD, <= 256 registers each); D = 256 -> 4 waves x 32 rows (one per SIMD, 512 registers).  RB = 2
//     (every K / V^T fragment feeds two MFMAs) was measured and lost;
//   * RING: 3 stages with one barrier per tile, or 2 stages with two barriers (D = 256: three would not fit);
//   * PRE: 0 = fragment reads placed by hipcc, 1 / 2 = K / K + first V^T fragments requested before the
//     exponentiation;
//   * SPLIT / CAUSAL / SPARSE: column-parallel pieces through the caller's workspace, causal mask, block mask --
//     separate code objects, so the dense kernel carries none of their code.
#pragma once
#include "attn_fwd16_common.h"
#include <type_traits>

// Tile loads forced inline: at D > 128 hipcc otherwise leaves the transposed code objects' tile loads (issue_loads below: NCH
// pieces x two operands x a gathered and a 16-byte form) as a real FUNCTION called from four sites, its closure in scratch memory
// (.private_segment_fixed_size 400-512, s_swappc_b64 in the loop) -- those code objects ran at ~0.1 PFLOP/s
// (profiles/r03_dev_transposed_streams.txt; with the lambda inline 0.48-0.62, profiles/r04_candidate/time_p5_tr_32heads.txt).
// tools/audit_code_objects.py fails the build check on any s_swappc_b64 or scratch in these objects.
#define MFA_V3_INLINE_LOADS __attribute__((always_inline))

namespace mfa {

// VD (bit mask of schedule options):
//    2 = KPAD: K rows padded by 16 bytes in LDS instead of XOR-swizzled (equally conflict-free for ds_read_b128;
//        the eight fragment addresses of a lane become one register plus immediates): +-0, developer knob;
//    4 = VPIPE: V^T fragments double-buffered in groups of four MFMAs (needs PRE >= 1);
//    8 = WSPREAD: the ds_write_b128 of the next tile issued between those groups (with VPIPE) or in the middle
//        of step A, instead of one burst at the barrier -- +11 % at D = 256 (one wave per SIMD), +-1 % otherwise;
//   32 = LDMA: tiles staged by LDS-DMA, every LDS read of the loop issued through the asm helpers below
//        (needs VPIPE, RING = 3) -- the D = 128 product.
//   (1 was softmax arithmetic on register pairs through inline-asm v_pk_fma_f32 / v_pk_add_f32: -23 % VALU
//   instructions, no measurable gain, and asm consumers of v_exp_f32 results escape hipcc's trans-use hazard
//   handling; 16 was the row sum on the matrix pipe: -2.5 %.  Both removed.)
// LDS reads as inline asm (LDMA schedule): hipcc puts s_waitcnt vmcnt(0) in front of every LDS read it cannot
// prove disjoint from the destination of an LDS-DMA in flight (all transposing reads), which would drain the
// prefetch of the next tile.  The asm reads are invisible to that pass; lds_wait<N> (an asm whose in/out operands
// are the fragments, so no consumer can be scheduled above it) waits until at most N younger LDS reads are pending
// -- LDS returns in order, and asm volatile statements keep their order.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_addr(const char *p) {
  return (uint32_t)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char *)p;
}
__device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
struct TrPair { u32x2_t lo, hi; };   // the two halves stay separate values until lds_wait has seen them: nothing but
                                     // the awaited registers themselves may sit between a read and its wait
template <int OFF> __device__ __forceinline__ TrPair lds_read_tr16_pair(uint32_t addr) {   // rows +0 and +8 of a 16-key group
  TrPair r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.hi) : "v"(addr), "n"(OFF + 8 * 64));
  return r;
}
template <int N> __device__ __forceinline__ void lds_wait(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N < 15 ? N : 15));   // 4-bit counter
}
template <int N> __device__ __forceinline__ void lds_wait(TrPair &a, TrPair &b, TrPair &c, TrPair &d) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi)
               : "n"(N < 15 ? N : 15));
}
__device__ __forceinline__ u32x4 tr_join(const TrPair &p) { return __builtin_shufflevector(p.lo, p.hi, 0, 1, 2, 3); }
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <typename T, int D, int NW, int RB, int THR, int PRE, int ABL = 0, int RING = 3, bool SPLIT = false,
          bool CAUSAL = false, int VD = 0, bool SPARSE = false, int TR = 0>
__global__ __launch_bounds__(NW * 64) void attn_fwd16_v3(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BC = 64, NT = NW * 64, NDB = D / 32, NKS = D / 16;
  constexpr bool KPAD = (VD & 2) != 0;
  constexpr bool VPIPE = (VD & 4) != 0;   // V^T fragments double-buffered in groups of four MFMAs (see step)
  constexpr bool WSPREAD = (VD & 8) != 0; // staging writes of the next tile issued between the matrix instructions of step A
  // LDMA: tiles go global -> LDS directly (buffer_load_dwordx4 ... lds): instruction i of wave w fills the 1 KiB of an
  // image at 16-byte positions (w * NCH + i) * 64 + lane, so each lane fetches the chunk that BELONGS at its position
  // (K swizzle and V sub-tiling applied on the source address).  No staging registers, no ds_write_b128.  Needs
  // 16-byte aligned rows (the host checks) and reads its fragments through the asm helpers above.
  constexpr bool LDMA = (VD & 32) != 0;
  constexpr bool VSPLIT = (VD & 64) != 0;   // developer schedule: V staging writes one sub-tile per half-wave (below)
  // LDMA with RING == 2 (D = 256, where three whole stages do not fit): the K images form a ring of THREE and the
  // V images a ring of TWO (3 x 32 + 2 x 32 KiB = all 160 KiB).  K(j+2) and V(j+1) are requested right behind the
  // barrier of iteration j: both have a whole iteration to land, and the second barrier of the register-staged
  // 2-stage ring (which published the tile written during step A) is not needed.
  constexpr bool KV32 = LDMA && RING == 2;
  // TR != 0: operands stored transposed ([D][sequence], transposeState of AttentionKernelDescriptor.swift:28-42, read in place
  // like AttentionKernel.swift:189-204 does -- no scratch).  Bit 0 = K, bit 1 = V: compile-time, because the LDS image keeps
  // the orientation of the source (16-byte chunks = 8 consecutive KEYS of one head-dimension element) and the two read recipes
  // change places: K^T is read like V (sub-images of 32 keys, ds_read_b64_tr_b16), V^T like K (rows of 64 keys, ds_read_b128).
  // Bit 2 alone = only Q / O may be transposed: those two are honoured at run time (a.op[].transposed) by every TR kernel,
  // outside the loop (Q^T: gathered 16-bit loads, once; O^T: stores straight from the accumulators, whose lanes are
  // consecutive rows).  The causal mask is a run-time flag of these kernels (one code object per transposition pattern).
  constexpr bool KT = (TR & 1) != 0, VT = (TR & 2) != 0;
  static_assert(TR == 0 || (!LDMA && PRE == 0 && !SPLIT && !SPARSE && CAUSAL && RB == 1 && (VD & ~2) == 0),
                "transposed operands: register-staged schedule, fragment reads left to hipcc");
  static_assert(!LDMA || (VPIPE && (RING == 3 || RING == 2) && PRE >= 1 && !KPAD && RB == 1 && NKS % 4 == 0),
                "LDMA: grouped-read schedule");
  constexpr int ROWB = D * 2 + ((KPAD && !KT) ? 16 : 0), KTILE = KT ? BC * D * 2 : BC * ROWB, TILE = BC * D * 2, STAGE = KTILE + TILE;
  constexpr int CPR = D / 8, NCH = BC * CPR / NT;
  // byte offset of K image `stage`; of the stage base the V addressing (which includes + KTILE) starts from
  auto koffs = [](int stage) { return KV32 ? stage * KTILE : stage * STAGE; };
  auto voffs = [](int stage) { return KV32 ? 2 * KTILE + stage * TILE : stage * STAGE; };
  static_assert(BC * CPR % NT == 0, "tile must divide evenly over the workgroup");
  // ABL: timing-only ablations (WRONG RESULTS): 2 = exp2 replaced by an FMA, 3 = one K fragment address, 8 = no
  // per-tile barrier, 20 = no fragment reads from LDS in the loop, 21 = no softmax arithmetic, 22 = 20 + 21, 23 = no row maximum
  constexpr bool NOLDS = (ABL == 20 || ABL == 22), NOSOFTMAX = (ABL == 21 || ABL == 22);

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
  if (!SPLIT && (int64_t)rblk * (NW * RB * 32) >= R) return;   // padded batch entry: the whole workgroup lies beyond its rows
  const int64_t r0 = (int64_t)rblk * (NW * RB * 32) + wave * (RB * 32);
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2,
                 ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;

  const bool qT = TR != 0 && a.op[SLOT_Q].transposed != 0, oT = TR != 0 && a.op[SLOT_O].transposed != 0;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)(qT ? Dr : R) * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_K], head, batch), 0, (uint32_t)(KT ? Dr : C) * ldk2, 0x00020000);
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_V], head, batch), 0, (uint32_t)(VT ? Dr : C) * ldv2, 0x00020000);
  constexpr uint32_t OOB = 0xFFFFFF00u;

  // ---- Q fragments (B operand of S^T = K Q^T), cached in registers for the whole kernel
  v8 qf[RB][NKS];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const uint32_t rowoff = (uint32_t)(r0 + b * 32 + q) * ldq2;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int d0 = 16 * s + 8 * hi;
      const uint32_t off = (d0 < Dr && r0 + b * 32 + q < R) ? rowoff + d0 * 2 : OOB;
      if constexpr (TR != 0) {
        // K^T fragments come out of the transposing read with the contraction index in the order of a 32 x 32 accumulator
        // block's registers -- elements 4 hi + {0..3, 8..11} of a 16-element step instead of 8 hi + {0..7} -- so Q follows
        auto elem = [&](int i) { return KT ? 16 * s + 4 * hi + (i & 3) + 8 * (i >> 2) : d0 + i; };
        if (qT) {   // Q^T: the lane's eight elements lie one leading dimension apart; lanes are consecutive rows
          uint16_t e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i)
            e[i] = __builtin_amdgcn_raw_buffer_load_b16(
                qres, (elem(i) < Dr && r0 + b * 32 + q < R) ? (uint32_t)elem(i) * ldq2 + (uint32_t)(r0 + b * 32 + q) * 2 : OOB, 0, 0);
          const u32x4 w = {e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16),
                           e[6] | ((uint32_t)e[7] << 16)};
          qf[b][s] = __builtin_bit_cast(v8, w);
          continue;
        }
        if constexpr (KT) {
          const bool rowok = r0 + b * 32 + q < R;
          const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(qres, (elem(0) < Dr && rowok) ? rowoff + elem(0) * 2 : OOB, 0, 0);
          const u32x2 up = __builtin_amdgcn_raw_buffer_load_b64(qres, (elem(4) < Dr && rowok) ? rowoff + elem(4) * 2 : OOB, 0, 0);
          const u32x4 w = {lo[0], lo[1], up[0], up[1]};
          qf[b][s] = __builtin_bit_cast(v8, w);
          continue;
        }
      }
      qf[b][s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(qres, off, 0, 0));
    }
  }

  // ---- key range of this workgroup: everything, or piece `split` of `splits` (SPLIT launches)
  const int tiles_total = (C + BC - 1) / BC;
  int tile0 = SPLIT ? (int)((uint64_t)split * tiles_total / grid.splits) : 0;   // (SPARSE: first tile of the current run)
  // CAUSAL (extension): row r sees key c iff c <= r + (C - R); the workgroup stops at the tile that holds
  // the last key its last row may see, tiles that cross the diagonal are masked element-wise.
  const int coff = (TR != 0 && !a.causal) ? 0x3FFFFFFF : causal_offset(R, C);   // TR kernels: the mask is a run-time flag (never reached without it)
  int tiles_visible = tiles_total;
  if constexpr (CAUSAL) {
    const int64_t last_row = min((int64_t)R, ((int64_t)rblk + 1) * (NW * RB * 32)) - 1;
    tiles_visible = (int)min((int64_t)tiles_total, (last_row + coff) / BC + 1);
  }
  int tile1 = SPLIT ? (int)((uint64_t)(split + 1) * tiles_total / grid.splits) : tiles_visible;

  // ---- K/V staging: global -> VGPR -> LDS (LDS-DMA staging measured 12 % slower)
  uint32_t koff[NCH], voff[NCH], klds[NCH], vlds[NCH];
  uint32_t kbase0[SPARSE ? NCH : 1], vbase0[SPARSE ? NCH : 1];   // SPARSE: offsets of tile 0, koff/voff restart per run
  const uint32_t kinc = KT ? BC * 2 : BC * ldk2, vinc = VT ? BC * 2 : BC * ldv2;
  // transposed tiles: chunk id = 8 consecutive keys (8 tc .. 8 tc + 7 of the tile) of head-dimension element id / 8; the tile
  // advances ALONG the rows, so the end of the sequence is not the end of the buffer: lcol / wcol = first key of the tile
  // requested / written next, chunks that begin at or beyond C are not fetched (zeros), the one chunk that straddles C
  // (C % 8 != 0) is cut to size before V^T is written (P = 0 there, but 0 x whatever the padding holds is not 0)
  const int tc8 = (tid & 7) * 8;
  int lcol = tile0 * BC, wcol = tile0 * BC;
  // rows of a transposed operand that do not begin on 16-byte boundaries (leading dimension = an odd sequence length, say):
  // the chunk is gathered by eight 16-bit loads instead of one 128-bit load -- slower, but still the matrix-core kernel
  const bool kGather = KT && (((uintptr_t)operand_base(a.op[SLOT_K], head, batch) | ldk2) & 15) != 0;
  const bool vGather = VT && (((uintptr_t)operand_base(a.op[SLOT_V], head, batch) | ldv2) & 15) != 0;
  auto gather_chunk = [&](const __amdgpu_buffer_rsrc_t &res, uint32_t off, int col) {   // keys col .. col + 7 of one row
    uint16_t e[8];
    const bool row = off < OOB;   // (rows beyond the head dimension: their offset saturates at 2^32 - 1 as the tiles advance,
                                  // and + 2 i would wrap into the buffer)
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_raw_buffer_load_b16(res, (row && col + i < C) ? off + 2 * i : OOB, 0, 0);
    const u32x4 w = {e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16),
                     e[6] | ((uint32_t)e[7] << 16)};
    return w;
  };
  static_assert(TR == 0 || NT % 8 == 0, "");
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int id = tid + i * NT;
    const int row = id / CPR, c = id % CPR;
    const bool valid = c * 8 < Dr;
    koff[i] = valid ? (tile0 * BC + row) * ldk2 + c * 16 : OOB;
    voff[i] = valid ? (tile0 * BC + row) * ldv2 + c * 16 : OOB;
    if constexpr (KT) koff[i] = (id / 8 < Dr) ? (uint32_t)(id / 8) * ldk2 + (uint32_t)(tile0 * BC + tc8) * 2 : OOB;
    if constexpr (VT) voff[i] = (id / 8 < Dr) ? (uint32_t)(id / 8) * ldv2 + (uint32_t)(tile0 * BC + tc8) * 2 : OOB;
    if constexpr (LDMA) {   // 16-byte position p of the image -> the chunk stored there
      const int p = (wave * NCH + i) * 64 + lane;
      const int krow = p / CPR, kc = (p % CPR) ^ kswz_mask<D>(krow);
      const int vkey = (p >> 2) % BC, vc = (p / (BC * 4)) * 4 + (p & 3);
      koff[i] = (kc * 8 < Dr) ? (tile0 * BC + krow) * ldk2 + kc * 16 : OOB;
      voff[i] = (vc * 8 < Dr) ? (tile0 * BC + vkey) * ldv2 + vc * 16 : OOB;
    }
    klds[i] = row * ROWB + (KPAD ? c : kswz<D>(row, c)) * 16;
    vlds[i] = KTILE + ((c >> 2) * BC + row) * 64 + (c & 3) * 16;
    if constexpr (VSPLIT) {
      // the V image is [D / 32 sub-tiles][64 keys][64 bytes]: with (row, c) = (id / CPR, id % CPR) the sixteen lanes of a
      // ds_write_b128 group write the same 64-byte columns of DIFFERENT sub-tiles (4096 bytes apart: the same banks).  Here a
      // half-wave takes ONE sub-tile: 8 keys x 64 bytes = 512 contiguous bytes (D = 64 forward: 9.4 % of the LDS-active cycles
      // were bank conflicts, profiles/r04_fwd_bf16_d64_summary.txt)
      const int vrow = (id / (32 * NDB)) * 8 + ((id >> 2) & 7), vc = ((id >> 5) % NDB) * 4 + (id & 3);
      voff[i] = (vc * 8 < Dr) ? (tile0 * BC + vrow) * ldv2 + vc * 16 : OOB;
      vlds[i] = KTILE + ((vc >> 2) * BC + vrow) * 64 + (vc & 3) * 16;
    }
    // K^T image: [2 blocks of 32 keys][D elements][64 bytes]; V^T image: [D elements][64 keys], chunks XOR-swizzled
    if constexpr (KT) klds[i] = (((id & 7) >> 2) * D + id / 8) * 64 + (id & 3) * 16;
    // (within a 16-key step the keys are stored in the order P^T holds them -- 4 h + {0..3, 8..11} in chunk 2 u + h -- so that
    // one ds_read_b128 is a fragment: the lane's chunk goes out as two halves, vlds = where keys +0..3 go, keys +4..7 one
    // chunk further)
    if constexpr (VT) vlds[i] = KTILE + (id / 8) * 128 + (id & 1) * 8;
    if constexpr (SPARSE) {
      kbase0[i] = valid ? row * ldk2 + c * 16 : OOB;
      vbase0[i] = valid ? row * ldv2 + c * 16 : OOB;
      if constexpr (LDMA) {
        const int p = (wave * NCH + i) * 64 + lane;
        const int krow = p / CPR, kc = (p % CPR) ^ kswz_mask<D>(krow);
        const int vkey = (p >> 2) % BC, vc = (p / (BC * 4)) * 4 + (p & 3);
        kbase0[i] = (kc * 8 < Dr) ? krow * ldk2 + kc * 16 : OOB;
        vbase0[i] = (vc * 8 < Dr) ? vkey * ldv2 + vc * 16 : OOB;
      }
    }
  }
  u32x4 kreg[LDMA ? 1 : NCH], vreg[LDMA ? 1 : NCH];
  typedef __attribute__((address_space(3))) void *lds_ptr;
  auto issue_dma_k = [&](int stage) {   // LDMA: the next K tile in sequence -> K image `stage` (zeros past the end)
    char *base = smem + koffs(stage) + wave * (NCH * 1024);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass of hipcc does not know this device builtin
      __builtin_amdgcn_raw_ptr_buffer_load_lds(kres, (lds_ptr)(base + i * 1024), 16, koff[i], 0, 0, 0);
#endif
      koff[i] = __builtin_elementwise_add_sat(koff[i], kinc);
    }
  };
  auto issue_dma_v = [&](int stage) {
    char *base = smem + voffs(stage) + KTILE + wave * (NCH * 1024);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vres, (lds_ptr)(base + i * 1024), 16, voff[i], 0, 0, 0);
#endif
      voff[i] = __builtin_elementwise_add_sat(voff[i], vinc);
    }
  };
  auto issue_dma = [&](int stage) { issue_dma_k(stage); issue_dma_v(stage); };
  auto issue_loads = [&]() MFA_V3_INLINE_LOADS {
    if constexpr (!LDMA) {
      const bool inside = lcol + tc8 < C;   // (transposed tiles only)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (KT && kGather) kreg[i] = gather_chunk(kres, koff[i], lcol + tc8);
        else kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(kres, (KT && !inside) ? OOB : koff[i], 0, 0);
        if (VT && vGather) vreg[i] = gather_chunk(vres, voff[i], lcol + tc8);
        else vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vres, (VT && !inside) ? OOB : voff[i], 0, 0);
        koff[i] = __builtin_elementwise_add_sat(koff[i], kinc);
        voff[i] = __builtin_elementwise_add_sat(voff[i], vinc);
      }
      lcol += BC;
    }
  };
  auto stage_part = [&](int stage, int i0, int i1) {   // write chunks [i0, i1) of tile j+1, then request them for tile j+2
    if constexpr (!LDMA) {
      char *base = smem + stage * STAGE;
#pragma unroll
      for (int i = i0; i < i1; ++i) {
        *reinterpret_cast<u32x4 *>(base + klds[i]) = kreg[i];
        *reinterpret_cast<u32x4 *>(base + vlds[i]) = vreg[i];
      }
#pragma unroll
      for (int i = i0; i < i1; ++i) {
        kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(kres, koff[i], 0, 0);
        vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vres, voff[i], 0, 0);
        koff[i] = __builtin_elementwise_add_sat(koff[i], kinc);
        voff[i] = __builtin_elementwise_add_sat(voff[i], vinc);
      }
    }
  };
  auto write_tiles = [&](int stage) {
    if constexpr (!LDMA) {
      char *base = smem + stage * STAGE;
      if constexpr (VT) {
        const int nv = C - wcol - tc8;   // keys of this lane's chunks inside the sequence
        if (nv > 0 && nv < 8) {
#pragma unroll
          for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const uint32_t keep = (2 * w + 1 < nv) ? 0xFFFFFFFFu : (2 * w < nv) ? 0x0000FFFFu : 0u;
              vreg[i][w] &= keep;
            }
        }
        wcol += BC;
      }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        *reinterpret_cast<u32x4 *>(base + klds[i]) = kreg[i];
        if constexpr (VT) {
          const int vrow = (tid + i * NT) / 8, c2 = tid & 6;   // chunks c2, c2 + 1 hold the step's keys
          const u32x2 lo = {vreg[i][0], vreg[i][1]}, up = {vreg[i][2], vreg[i][3]};
          *reinterpret_cast<u32x2 *>(base + vlds[i] + kswz<64>(vrow, c2) * 16) = lo;
          *reinterpret_cast<u32x2 *>(base + vlds[i] + kswz<64>(vrow, c2 + 1) * 16) = up;
        } else {
          *reinterpret_cast<u32x4 *>(base + vlds[i]) = vreg[i];
        }
      }
    }
  };
  const int n16 = lane & 15;
  const int vtr_off = KTILE + ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;
  int kread[NKS];
#pragma unroll
  for (int t = 0; t < NKS; ++t) kread[t] = q * ROWB + (KPAD ? 2 * t + hi : kswz<D>(q, 2 * t + hi)) * 16;

  // S^T for the 32 keys of half `kb` of the tile in `stage`: one K fragment feeds RB MFMAs
  auto qk = [&](int stage, int kb, f32x16 (&s)[RB]) {
    const char *Ks = smem + koffs(stage) + (KT ? kb * D * 64 + (vtr_off - KTILE) : kb * 32 * ROWB);
#pragma unroll
    for (int t = 0; t < NKS; ++t) {
      v8 kf;
      if constexpr (KT) {   // key = 32 kb + lane % 32, elements 16 t + 8 hi ..: rows 16 t .. of sub-image kb
        const char *kp = Ks + 16 * t * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(kp));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(kp + 8 * 64));
        kf = __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
      } else {
        kf = NOLDS ? qf[0][(t + 1) % NKS] : *reinterpret_cast<const v8 *>(Ks + kread[ABL == 3 ? 0 : t]);
      }
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        if (t == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
        }
        s[b] = F::mfma(kf, qf[b][t], s[b]);
      }
    }
  };

  f32x16 o[RB][NDB];
  float m[RB], l[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    m[b] = ABL == 23 ? 0.f : -3.402823466e+38f;   // +Caching.swift:310
    l[b] = 0.f;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[b][db][r] = 0.f;
  }

  auto mask_edge = [&](f32x16 (&s)[RB], int c0) {   // maskAttentionMatrixEdge (+Softmax.swift:228-260)
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) >= C) s[b][r] = mask_value();
  };
  // causal mask of the 32 keys starting at c0, skipped (wave-uniform test) when the whole block lies
  // at or below the diagonal for every row of this wave
  auto mask_causal = [&](f32x16 (&s)[RB], int c0) {
    if constexpr (CAUSAL) {
      if (c0 + 31 > r0 + coff) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          const int64_t limit = r0 + b * 32 + q + coff;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (c0 + crow(r, hi) > limit) s[b][r] = mask_value();
        }
      }
    }
  };
  auto block_max = [&](const f32x16 (&s)[RB], float (&m_new)[RB]) {   // onlineReduceMaximum
    if constexpr (NOSOFTMAX) { m_new[0] = 0.f; return; }
    if constexpr (ABL == 23) {   // timing only: what the row maximum costs (the reference maximum stays 0, nothing is ever re-based)
#pragma unroll
      for (int b = 0; b < RB; ++b) m_new[b] = 0.f;
      return;
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      float mx0 = fmaxf(s[b][0], s[b][1]), mx1 = fmaxf(s[b][2], s[b][3]);
#pragma unroll
      for (int r = 4; r < 16; r += 4) {
        mx0 = fmaxf(fmaxf(mx0, s[b][r]), s[b][r + 1]);
        mx1 = fmaxf(fmaxf(mx1, s[b][r + 2]), s[b][r + 3]);
      }
      m_new[b] = half_swap_max(fmaxf(mx0, mx1)) * a.scale2;
    }
  };
  auto rescale_if_needed = [&](const float (&m_new)[RB]) {   // onlineCorrectO, deferred by THR
    bool need = false;
#pragma unroll
    for (int b = 0; b < RB; ++b) need |= (m_new[b] > m[b] + (float)THR);
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        const float m_up = fmaxf(m[b], m_new[b]);
        const float corr = fast_exp2(m[b] - m_up);
        m[b] = m_up;
        l[b] *= corr;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[b][db][r] *= corr;
      }
    }
  };
  auto exponentiate = [&](f32x16 (&s)[RB], v8 (&pf)[RB][2]) {   // softmax + onlineReduceSum
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const float mb = m[b];
      if constexpr (!NOSOFTMAX) {
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = (ABL == 2) ? s[b][r] * a.scale2 - mb : fast_exp2(s[b][r] * a.scale2 - mb);
          s[b][r] = p;
          ps[r & 3] += p;
        }
        l[b] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {   // MFMA step u (16 keys) uses registers 8u .. 8u+7
        v8 pk;
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = (T)s[b][8 * u + i];
        pf[b][u] = pk;
      }
    }
  };
  // O^T += V^T P^T for the 32 keys of half `kb`: one V^T fragment feeds RB MFMAs
  auto pv = [&](int stage, int kb, const v8 (&pf)[RB][2]) {
    const char *Vs = smem + voffs(stage) + vtr_off + kb * 32 * 64;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        v8 vf;
        if constexpr (VT) {
          // element 32 db + lane % 32; keys in the order P^T holds them (registers of an accumulator block): the image is written
          // that way (write_tiles)
          const int vrow = 32 * db + q;
          vf = *reinterpret_cast<const v8 *>(smem + voffs(stage) + KTILE + vrow * 128 + kswz<64>(vrow, 4 * kb + 2 * u + hi) * 16);
        } else {
          const char *vp = Vs + (db * BC + 16 * u) * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + 8 * 64));
          vf = NOLDS ? qf[0][(2 * db + u) % NKS] : __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) o[b][db] = F::mfma(vf, pf[b][u], o[b][db]);
      }
  };

  v8 pf[RB][2];
  // One pipeline step with the LDS fragment reads issued up front (PRE >= 1): all K fragments of
  // the NEXT 32 keys (and with PRE == 2 the first half of the V^T fragments of the current 32) are
  // requested before the VALU starts exponentiating, so their LDS latency is covered by that work
  // instead of being exposed in front of every MFMA (hipcc otherwise issues each ds_read one MFMA
  // ahead of its consumer).  sched_barrier(0) pins the reads above the arithmetic.
  auto step = [&](f32x16 (&s_cur)[RB], f32x16 (&s_next)[RB], int k_stage, int k_kb, int v_stage, int v_kb,
                  bool do_qk, auto spread, int wstage) {
    const char *Ks = smem + k_stage * STAGE + k_kb * 32 * ROWB;
    const char *Vs = smem + v_stage * STAGE + vtr_off + v_kb * 32 * 64;
    v8 kf[NKS];
    if (do_qk) {
#pragma unroll
      for (int t = 0; t < NKS; ++t) kf[t] = *reinterpret_cast<const v8 *>(Ks + kread[t]);
    }
    auto load_v = [&](int u, v8 (&vf)[NDB]) {
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const char *vp = Vs + (db * BC + 16 * u) * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + 8 * 64));
        vf[db] = __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    if constexpr (VPIPE) {
      // One wave per SIMD has nobody to hide its LDS latency: request the V^T fragments of four MFMAs while the
      // previous four run (hipcc otherwise issues each pair of transposed reads one MFMA ahead of its consumer).
      // sched_barrier(0x406) pins matrix and LDS instructions, VALU / SALU / transcendental work may still move.
      constexpr int G = 4, NG = 2 * NDB / G;
      static_assert(RB == 1 && (2 * NDB) % G == 0, "VPIPE: one row block per wave");
      auto load_group = [&](int g, v8 (&vf)[G]) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int idx = g * G + i, u = idx / NDB, db = idx % NDB;
          const char *vp = Vs + (db * BC + 16 * u) * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + 8 * 64));
          vf[i] = __builtin_bit_cast(v8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      };
      v8 vg[2][G];
      load_group(0, vg[0]);
      __builtin_amdgcn_sched_barrier(0);
      exponentiate(s_cur, pf);
      if (do_qk) {
#pragma unroll
        for (int t = 0; t < NKS; ++t) {
          if (t == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_next[0][r] = 0.f;
          }
          s_next[0] = F::mfma(kf[t], qf[0][t], s_next[0]);
        }
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) load_group(g + 1, vg[(g + 1) & 1]);
        if constexpr (decltype(spread)::value) stage_part(wstage, NCH * g / NG, NCH * (g + 1) / NG);
        __builtin_amdgcn_sched_barrier(0x406);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int idx = g * G + i, u = idx / NDB, db = idx % NDB;
          o[0][db] = F::mfma(vg[g & 1][i], pf[0][u], o[0][db]);
        }
        __builtin_amdgcn_sched_barrier(0x406);
      }
      return;
    }
    v8 vf0[NDB], vf1[NDB];
    if constexpr (PRE == 2) load_v(0, vf0);
    __builtin_amdgcn_sched_barrier(0);
    exponentiate(s_cur, pf);
    if (do_qk) {
#pragma unroll
      for (int t = 0; t < NKS; ++t)
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          if (t == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_next[b][r] = 0.f;
          }
          s_next[b] = F::mfma(kf[t], qf[b][t], s_next[b]);
        }
    }
    if constexpr (PRE != 2) load_v(0, vf0);
    load_v(1, vf1);
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int b = 0; b < RB; ++b) o[b][db] = F::mfma(vf0[db], pf[b][0], o[b][db]);
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int b = 0; b < RB; ++b) o[b][db] = F::mfma(vf1[db], pf[b][1], o[b][db]);
  };

  // LDMA pipeline step: the same order as the grouped step above with every LDS read issued through asm -- K fragments
  // of the next 32 keys and the first group of V^T fragments up front, one group of four V^T fragments requested
  // while the previous four are multiplied.
  auto step_dma = [&](f32x16 (&s_cur)[RB], f32x16 (&s_next)[RB], int k_stage, int k_kb, int v_stage, int v_kb) {
    if constexpr (LDMA) {
      constexpr int G = 4, NG = 2 * NDB / G;
      const uint32_t ka = lds_addr(smem + koffs(k_stage) + k_kb * 32 * ROWB);
      const uint32_t va = lds_addr(smem + voffs(v_stage) + vtr_off + v_kb * 32 * 64);
      u32x4 kf[NKS];
      TrPair vg[2][G];
#pragma unroll
      for (int t = 0; t < NKS; ++t) kf[t] = lds_read_b128(ka + kread[t]);
      auto load_group = [&](auto gc, TrPair (&vf)[G]) {
        constexpr int g = decltype(gc)::value;
        static_for<G>([&](auto ic) {
          constexpr int idx = g * G + decltype(ic)::value, u = idx / NDB, db = idx % NDB;
          vf[decltype(ic)::value] = lds_read_tr16_pair<(db * BC + 16 * u) * 64>(va);
        });
      };
      load_group(std::integral_constant<int, 0>{}, vg[0]);
      __builtin_amdgcn_sched_barrier(0);
      exponentiate(s_cur, pf);
#pragma unroll
      for (int t = 0; t < NKS; t += 4)   // 2 G transposing reads were issued after the K fragments (in-order return)
        lds_wait<2 * G>(kf[t], kf[t + 1], kf[t + 2], kf[t + 3]);
#pragma unroll
      for (int t = 0; t < NKS; ++t) {
        if (t == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s_next[0][r] = 0.f;
        }
        s_next[0] = F::mfma(__builtin_bit_cast(v8, kf[t]), qf[0][t], s_next[0]);
      }
      // sched_barrier(0x406): matrix instructions and the asm reads / waits keep this order, VALU / SALU /
      // transcendental work (the exponentiation above, the block maximum that follows) may move across
      __builtin_amdgcn_sched_barrier(0x406);
      static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + 1 < NG) load_group(std::integral_constant<int, g + 1>{}, vg[(g + 1) & 1]);
        TrPair (&vf)[G] = vg[g & 1];
        lds_wait<(g + 1 < NG) ? 2 * G : 0>(vf[0], vf[1], vf[2], vf[3]);
        __builtin_amdgcn_sched_barrier(0x406);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int idx = g * G + i, u = idx / NDB, db = idx % NDB;
          o[0][db] = F::mfma(__builtin_bit_cast(v8, tr_join(vf[i])), pf[0][u], o[0][db]);
        }
        __builtin_amdgcn_sched_barrier(0x406);
      });
    }
  };

  static_assert(!(SPARSE && SPLIT), "block-sparse launches are row-parallel");
  // One contiguous run of key tiles [tile0, tile1): prologue, pipelined loop, tail.  Dense launches make one
  // call; block-sparse launches one per run of active tiles, the online-softmax state (m, l, O) carried
  // across in registers.
  auto traverse = [&]() {
  if constexpr (SPARSE) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      koff[i] = __builtin_elementwise_add_sat(kbase0[i], (uint32_t)tile0 * kinc);
      voff[i] = __builtin_elementwise_add_sat(vbase0[i], (uint32_t)tile0 * vinc);
    }
  }
  // ---- prologue
  const int ntiles = tile1 - tile0;
  const bool ragged = (C & (BC - 1)) != 0 && tile1 == tiles_total;   // only the globally last tile is partial
  if constexpr (KV32) {
    issue_dma_k(0);
    issue_dma_v(0);
    issue_dma_k(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCH) : "memory");       // K(0), V(0) (and the Q fragments) have landed
  } else if constexpr (LDMA) {
    issue_dma(0);      // first tile of the range -> stage 0
    issue_dma(1);      // second tile -> stage 1 (zeros past the end)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NCH) : "memory");   // the first tile (and the Q fragments) have landed
  } else {
    issue_loads();
    write_tiles(0);
    issue_loads();
  }
  __syncthreads();
  f32x16 s0[RB], s1[RB];   // half score tiles (keys 0-31 / 32-63 of a tile); roles alternate
  float m_new[RB];
  qk(0, 0, s0);
  if (ntiles == 1 && ragged) mask_edge(s0, tile0 * BC);
  mask_causal(s0, tile0 * BC);
  block_max(s0, m_new);

  int st_cur_rt = 0, st_next_rt = 1;
  int k_cur_rt = 0;   // KV32: position of K(j) in the ring of three K images (st_cur is V(j)'s in the ring of two)
  // iteration j: s0 = S(tile j, keys 0-31) and its block maximum are ready on entry
  // RING == 3: tile j+1 replaces tile j-2, whose last reader finished before the barrier of the
  // previous iteration, so one barrier per tile suffices.  RING == 2 (head dimensions whose three
  // stages would not fit the 160 KiB LDS): tile j+1 replaces tile j-1, still being read by slower
  // waves until they reach this iteration's first barrier -- write after it, and publish the tile
  // with a second barrier before step B reads it.
  auto iteration = [&](int j, bool next_is_last, int st_cur, int st_next) {
    rescale_if_needed(m_new);
    if constexpr (LDMA) {
      // this wave's part of tile j+1 has had a whole iteration to land; behind the barrier every part is visible
      // and nobody reads tile j-1 any more, so its stage takes tile j+2
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      __syncthreads();
      const int k_cur = KV32 ? k_cur_rt : st_cur, k_next = KV32 ? (k_cur_rt == 2 ? 0 : k_cur_rt + 1) : st_next;
      if constexpr (KV32) {
        issue_dma_k(k_cur == 0 ? 2 : k_cur - 1);   // K(j+2) replaces K(j-1)
        issue_dma_v(st_next);                      // V(j+1) replaces V(j-1)
      } else {
        issue_dma(3 - st_cur - st_next);
      }
      step_dma(s0, s1, k_cur, 1, st_cur, 0);
      mask_causal(s1, (tile0 + j) * BC + 32);
      block_max(s1, m_new);
      rescale_if_needed(m_new);
      step_dma(s1, s0, k_next, 0, st_cur, 1);
      if (next_is_last && ragged) mask_edge(s0, (tile0 + j + 1) * BC);
      mask_causal(s0, (tile0 + j + 1) * BC);
      block_max(s0, m_new);
      return;
    }
    if constexpr (WSPREAD && RING == 3) {
      // tile j+1 is only read from step B on: stage it in the middle of step A, behind this wave's own K fragment
      // reads, and publish it with the barrier between the two steps (every wave past the previous iteration's
      // barrier has finished tile j-2, which the writes replace).  sched_barrier(0x40E): memory instructions keep
      // their place, matrix / vector / scalar work may move across.
      static_assert(PRE == 0 || VPIPE, "WSPREAD: unhoisted schedule or grouped V reads");
      if constexpr (VPIPE) {
        step(s0, s1, st_cur, 1, st_cur, 0, true, std::true_type{}, st_next);
      } else {
        qk(st_cur, 1, s1);
        __builtin_amdgcn_sched_barrier(0x40E);
        stage_part(st_next, 0, NCH);
        __builtin_amdgcn_sched_barrier(0x40E);
        exponentiate(s0, pf);
        pv(st_cur, 0, pf);
      }
      mask_causal(s1, (tile0 + j) * BC + 32);
      block_max(s1, m_new);
      rescale_if_needed(m_new);
      __syncthreads();
      if constexpr (PRE == 0) {
        qk(st_next, 0, s0);
        exponentiate(s1, pf);
        pv(st_cur, 1, pf);
      } else {
        step(s1, s0, st_next, 0, st_cur, 1, true, std::false_type{}, 0);
      }
      if (next_is_last && ragged) mask_edge(s0, (tile0 + j + 1) * BC);
      mask_causal(s0, (tile0 + j + 1) * BC);
      block_max(s0, m_new);
      return;
    }
    if constexpr (WSPREAD && RING == 2) {
      static_assert(VPIPE, "WSPREAD with a 2-stage ring: grouped V reads");
      __syncthreads();
      step(s0, s1, st_cur, 1, st_cur, 0, true, std::true_type{}, st_next);
      mask_causal(s1, (tile0 + j) * BC + 32);
      block_max(s1, m_new);
      rescale_if_needed(m_new);
      __syncthreads();
      step(s1, s0, st_next, 0, st_cur, 1, true, std::false_type{}, 0);
      if (next_is_last && ragged) mask_edge(s0, (tile0 + j + 1) * BC);
      mask_causal(s0, (tile0 + j + 1) * BC);
      block_max(s0, m_new);
      return;
    }
    if constexpr (RING == 3) {
      write_tiles(st_next);          // tile j+1 (replaces tile j-2)
      issue_loads();                 // tile j+2 (reads as zero past the end)
      if constexpr (ABL != 8) __syncthreads();   // ABL 8: timing-only ablation (racy, wrong results)
    } else {
      __syncthreads();
      write_tiles(st_next);
      issue_loads();
    }
    // step A: matrix pipe S(j, keys 32-63) | VALU exp(s0); then PV(keys 0-31) | VALU max(s1)
    if constexpr (PRE == 0) {
      qk(st_cur, 1, s1);
      exponentiate(s0, pf);
      pv(st_cur, 0, pf);
    } else {
      step(s0, s1, st_cur, 1, st_cur, 0, true, std::false_type{}, 0);
    }
    mask_causal(s1, (tile0 + j) * BC + 32);
    block_max(s1, m_new);
    rescale_if_needed(m_new);
    if constexpr (RING == 2) __syncthreads();
    // step B: matrix pipe S(j+1, keys 0-31) | VALU exp(s1); then PV(keys 32-63) | VALU max(s0)
    if constexpr (PRE == 0) {
      qk(st_next, 0, s0);
      exponentiate(s1, pf);
      pv(st_cur, 1, pf);
    } else {
      step(s1, s0, st_next, 0, st_cur, 1, true, std::false_type{}, 0);
    }
    if (next_is_last && ragged) mask_edge(s0, (tile0 + j + 1) * BC);
    mask_causal(s0, (tile0 + j + 1) * BC);
    block_max(s0, m_new);
  };
  auto advance = [&]() {
    st_cur_rt = st_next_rt;
    st_next_rt = (st_next_rt == RING - 1) ? 0 : st_next_rt + 1;
    k_cur_rt = (k_cur_rt == 2) ? 0 : k_cur_rt + 1;
  };
  int j = 0;
  for (; j + 2 < ntiles; ++j) { iteration(j, false, st_cur_rt, st_next_rt); advance(); }
  if (j + 1 < ntiles) { iteration(j, true, st_cur_rt, st_next_rt); advance(); ++j; }
  const int st_cur = st_cur_rt;
  // last tile (j = ntiles-1): both halves, no successor
  rescale_if_needed(m_new);
  if constexpr (KV32) {   // V of the last tile was requested in the last iteration: not yet awaited / published
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  qk(KV32 ? k_cur_rt : st_cur, 1, s1);
  exponentiate(s0, pf);
  pv(st_cur, 0, pf);
  if (ragged) mask_edge(s1, (tile0 + j) * BC + 32);
  mask_causal(s1, (tile0 + j) * BC + 32);
  block_max(s1, m_new);
  rescale_if_needed(m_new);
  exponentiate(s1, pf);
  pv(st_cur, 1, pf);
  };   // traverse

  if constexpr (!SPARSE) {
    traverse();
  } else {
    // Block mask (extension): bit (row block of 256 rows, column block of 128 keys) of the caller's bitmap;
    // the workgroup's rows lie inside one row block, a 64-key tile inside one column block.  Runs of
    // consecutive active tiles are traversed one after the other; inactive tiles are never loaded.
    const uint32_t *mrow = a.mask + (int64_t)head * a.maskHeadStride + (int64_t)batch * a.maskBatchStride +
                           (uint64_t)(((uint64_t)rblk * (NW * RB * 32)) >> 8) * a.maskWords;
    auto active = [&](int tile) { const int cb = tile >> 1; return ((mrow[cb >> 5] >> (cb & 31)) & 1u) != 0; };
    const int tend = tile1;
    bool first = true;
    int t = 0;
    while (t < tend) {
      if (!active(t)) { ++t; continue; }
      int e = t + 1;
      while (e < tend && active(e)) ++e;
      if (!first) __syncthreads();   // every wave is done with the previous run's ring stages
      first = false;
      tile0 = t;
      tile1 = e;
      traverse();
      t = e;
    }
  }

  // ---- epilogue: O /= l (+Source.swift:165-171), L = m + log2(l) (+Caching.swift:373-377).
  // SPLIT launches instead publish the un-normalised (O, m, l) of their key range; attn_fwd_combine
  // merges the pieces (no atomics: every piece has its own slab, like the reference's dQ / dK-dV split).
  if constexpr (LDMA) __builtin_amdgcn_s_waitcnt(0x0F70);   // the run-ahead DMAs (zeros past the end) must not land in the epilogue's buffer
  __syncthreads();   // every wave is done with the ring
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + wave * (RB * 32 * OLD);
  char *lbase = operand_base(a.op[SLOT_L], head, batch);
  const size_t slab = ((size_t)split * grid.heads * grid.batches + (size_t)batch * grid.heads + head) * (size_t)a.R;
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const float l_tot = half_swap_add(l[b]) + 1.401298464e-45f;
    const float inv = SPLIT ? 1.0f : (l_tot > 1e-30f ? 1.0f / l_tot : 0.f);   // a row may see no key at all (block mask, empty batch entry)
    float *orow = Os + (b * 32 + q) * OLD;
    const int64_t row = r0 + b * 32 + q;
    bool stored = false;
    if constexpr (TR != 0) {
      if (oT) {   // O^T ([D][R]): register r of block db is element 32 db + crow(r, hi) of the lane's row -- lanes = consecutive rows
        const int prec = a.op[SLOT_O].precision;
        const uint32_t esz = prec == PREC_FP32 ? 4u : 2u, ldo = (uint32_t)a.op[SLOT_O].ld;
        const __amdgpu_buffer_rsrc_t ores =
            __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_O], head, batch), 0, (uint32_t)Dr * ldo * esz, 0x00020000);
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = 32 * db + crow(r, hi);
            const uint32_t off = (d < Dr && row < R) ? ((uint32_t)d * ldo + (uint32_t)row) * esz : OOB;
            const float val = o[b][db][r] * inv;
            if (prec == PREC_FP32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), ores, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pack16<T>(val, 0.f), ores, off, 0, 0);
          }
        stored = true;
      }
    }
    if (!stored) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) =
              make_float4(o[b][db][4 * g] * inv, o[b][db][4 * g + 1] * inv, o[b][db][4 * g + 2] * inv, o[b][db][4 * g + 3] * inv);
    }
    if (hi == 0 && row < R) {
      if constexpr (SPLIT) {
        grid.wsML[(slab + row) * 2] = m[b];
        grid.wsML[(slab + row) * 2 + 1] = l_tot;
      } else {
        store_elem(lbase, row, a.op[SLOT_L].precision, m[b] + log2f(l_tot));
      }
    }
  }
  // whole-row stores: SPLIT -> fp32 slab of the workspace; otherwise O in FP32 or, fused cast, the 16-bit type
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    if constexpr (SPLIT)
      store_block_rows<T, D>(Os + b * 32 * OLD, reinterpret_cast<char *>(grid.wsO + slab * Dr), PREC_FP32, (uint32_t)Dr,
                             r0 + 32 * b, R, Dr, lane);
    else if (!oT)
      store_block_rows<T, D>(Os + b * 32 * OLD, operand_base(a.op[SLOT_O], head, batch), a.op[SLOT_O].precision,
                             (uint32_t)a.op[SLOT_O].ld, r0 + 32 * b, R, Dr, lane);
  }
}

// Merge the pieces of a column-parallel forward launch.  One wave per query row.  m* = max_s m_s;  w_s = exp2(m_s - m*);
// l* = sum_s w_s l_s;  O = sum_s w_s O_s / l*;  L = m* + log2 l*  -- the online-softmax merge (+Softmax.swift:290-324) applied across
// pieces instead of across tiles.  Round 5: LATENCY-bound, not bandwidth-bound (one head: 16 pieces x 4096 rows x 264 bytes = 17 MB),
// and the first version walked the pieces in two serial loops of dependent loads (~2 x 16 round trips to memory: 20 of the 33 us of
// BASELINE config 2 as written).  Now lane s loads (m_s, l_s) of piece s (splits <= 64) and the wave reduces; for O the wave is
// CL = D / 4 column lanes (four consecutive elements each) x 64 / CL piece groups, every lane walks splits / groups pieces with
// independent loads and the groups are summed with half-wave exchanges: two to three round trips in all.
static __global__ __launch_bounds__(256) void attn_fwd_combine(const KernelArgs a, const Fwd16Grid grid) {
  const int lane = threadIdx.x & 63;
  const uint32_t R = a.R, Dr = a.D, HB = grid.heads * grid.batches, S = grid.splits;
  const uint64_t rowid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // over HB * R
  if (rowid >= (uint64_t)HB * R) return;
  const uint32_t hb = (uint32_t)(rowid / R), row = (uint32_t)(rowid % R);
  const uint32_t head = hb % grid.heads, batch = hb / grid.heads;
  auto slab_of = [&](uint32_t s_) { return ((uint64_t)s_ * HB + hb) * R + row; };
  // (m, l) of piece `lane`
  float ms = -3.402823466e+38f, ls = 0.f;
  if ((uint32_t)lane < S) {
    const float2 ml = *reinterpret_cast<const float2 *>(grid.wsML + slab_of((uint32_t)lane) * 2);
    ms = ml.x; ls = ml.y;
  }
  float mstar = ms;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mstar = fmaxf(mstar, __shfl_xor(mstar, off, 64));
  const float w = (uint32_t)lane < S ? fast_exp2(ms - mstar) : 0.f;
  float lstar = w * ls;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) lstar += __shfl_xor(lstar, off, 64);
  // O: column lane c = lane % CL owns elements 4c .. 4c + 3, group g = lane / CL walks the pieces g, g + G, ...
  uint32_t CL = 1;
  while (CL * 4 < Dr) CL <<= 1;                 // (D <= 256: at most 64 column lanes)
  const uint32_t G = 64 / CL, c = (uint32_t)lane % CL, g = (uint32_t)lane / CL;
  const bool active = c * 4 < Dr;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (uint32_t s0 = 0; s0 < S; s0 += G) {
    const uint32_t s_ = s0 + g;
    const float ws = __shfl(w, (int)(s_ < 64 ? s_ : 63), 64);     // (every lane takes part in the exchange)
    if (s_ < S && active) {
      const float4 v = *reinterpret_cast<const float4 *>(grid.wsO + slab_of(s_) * Dr + c * 4);
      acc.x += ws * v.x; acc.y += ws * v.y; acc.z += ws * v.z; acc.w += ws * v.w;
    }
  }
  for (uint32_t off = CL; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, (int)off, 64); acc.y += __shfl_xor(acc.y, (int)off, 64);
    acc.z += __shfl_xor(acc.z, (int)off, 64); acc.w += __shfl_xor(acc.w, (int)off, 64);
  }
  const float inv = 1.0f / lstar;
  if (active && g == 0) {
    char *obase = operand_base(a.op[SLOT_O], head, batch);
    const int64_t idx = (int64_t)row * a.op[SLOT_O].ld + c * 4;
    const int oprec = a.op[SLOT_O].precision;
    if (oprec == PREC_FP32) {
      *reinterpret_cast<float4 *>(reinterpret_cast<float *>(obase) + idx) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    } else {
      store_elem(obase, idx, oprec, acc.x * inv);
      store_elem(obase, idx + 1, oprec, acc.y * inv);
      store_elem(obase, idx + 2, oprec, acc.z * inv);
      store_elem(obase, idx + 3, oprec, acc.w * inv);
    }
  }
  if (lane == 0)
    store_elem(operand_base(a.op[SLOT_L], head, batch), row, a.op[SLOT_L].precision, mstar + log2f(lstar));
}

} // namespace mfa
