// attn_dkv16_rs.h -- backwardKeyValue on the 16-bit matrix cores, ROLE-SPLIT wave pairs.
//
// Same math and fragment maps as attn_dkv16 (attn_bwd16.h); what changes is how the work of one block of
// 32 keys is laid on the hardware.  attn_dkv16 gives the block to ONE wave: K and V fragments plus the dK
// and dV accumulators take ~200 registers, so only one wave fits per SIMD, and that lone wave has to issue
// 32 MFMAs, ~125 VALU and ~65 LDS instructions per 32-row tile by itself (one VALU per ~7.3 cycles,
// tools/probe_valu.hip): rocprof shows the matrix pipe 33 % busy.  Here the block goes to the two waves
// that share a SIMD (w and w + 4):
//     V-wave (w < 4):  S = Q K^T -> P = exp2(S*scale2 - L) -> dV^T += dO^T P      (K fragments, dV)
//     K-wave (w >= 4): dP = dO V^T -> dS = P (dP*scale - D) -> dK^T += Q^T dS     (V fragments, dK)
// Each wave holds ONE cached operand and ONE accumulator, two waves fit per SIMD, and each issues half of
// the instructions.  The only coupling is P: the V-wave leaves its fp32 P tile (4 KiB) in an LDS exchange
// buffer and the K-wave picks it up ONE STEP LATER, so the ring's barrier per step is the only
// synchronisation.  No recomputation: still 4 GEMMs per (row block, key block), as the reference
// (+Source.swift:244-293).
//
// Software pipeline (per wave, as attn_fwd16_v3.h): the first product of the NEXT row block runs on the
// matrix pipe while the VALU turns the current one into P / dS, then the second product of the current
// block.  At step t:   V-waves: S(t+1) | P(t) | dV += dO^T(t) P(t)
//                      K-waves: dP(t)  | dS(t-1) using P(t-1) | dK += Q^T(t-1) dS(t-1)
//
// LDS: ONE image per operand tile, [D/32][32 rows][32 elements] with the four 16-byte chunks of a 64-byte
// row XOR-swizzled by (row >> 2) & 3.  Both access patterns are conflict-free on it: ds_read_b128 of a
// row fragment (A operand of the first product; 16 lanes of a read group land on 16 distinct 16-byte
// slots) and ds_read_b64_tr_b16 (A operand of the second product; every lane supplies its own address,
// so the swizzle is folded into two lane constants).  attn_dkv16 keeps two images per operand.
// Ring of 4 stages {Q | dO | L[32] D[32]}: row blocks t-1, t, t+1 are live during step t and t+2 is
// written after the step's barrier (it replaces t-2, last read in step t-1).  Then the exchange buffer
// [pair][parity][4][64 lanes x 16 B].
#pragma once
#include "attn_bwd16.h"
#include <type_traits>

namespace mfa {

constexpr int DKV16RS_MAX_ROW_BLOCKS = 4096;   // 256-row blocks a block-sparse launch can list (R <= 1 Mi rows)
// wave pairs per workgroup: four (two waves per SIMD) up to D = 128; at D = 256 one accumulator alone takes 128
// registers, so a wave gets a SIMD to itself (512 registers) and the workgroup holds two pairs
template <int D> constexpr int dkv16rs_pairs() { return D <= 128 ? 4 : 2; }
template <int D> constexpr int dkv16rs_lds_bytes() {
  constexpr int ring = 4 * (2 * 32 * D * 2 + 256) + dkv16rs_pairs<D>() * 2 * 4096 + 2 * DKV16RS_MAX_ROW_BLOCKS + 16;
  constexpr int epi = 2 * dkv16rs_pairs<D>() * 32 * (D + 4) * 4;
  return ring > epi ? ring : epi;
}

// ABL: timing-only ablations (WRONG RESULTS): 1 = no global loads in the loop, 2 = no staging at all,
// 3 = no barrier, 4 = no L/D/P LDS traffic in the arithmetic
// SPLIT: traversal-parallel launch (see attn_dq16): the row blocks are cut into grid.splits pieces, partial dV and
// dK go to fp32 slabs of the caller's workspace (dV slabs first, then dK slabs), attn_bwd_combine adds them.
template <typename T, int D, typename TG = T, bool CAUSAL = false, int ABL = 0, bool SPARSE = false, bool SPLIT = false>
__global__ __launch_bounds__(dkv16rs_pairs<D>() * 128) void attn_dkv16_rs(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPAIR = dkv16rs_pairs<D>(), WGCOLS = NPAIR * 32;
  constexpr int BR = 32, NT = NPAIR * 128, NDB = D / 32, NKS = D / 16, RING = 4;
  constexpr int TILE = BR * D * 2, STAGE = 2 * TILE + 256, XBUF = RING * STAGE;
  constexpr int CPR = D / 8, NCHUNK = BR * CPR;   // 16-byte chunks per operand tile (256 / 512 / 1024 at D = 64 / 128 / 256)
  constexpr int SCH = (NCHUNK + NT - 1) / NT;     // chunks per thread and operand (1; 4 at D = 256)

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave % NPAIR, role = wave / NPAIR;   // role 0: V-wave (dV), role 1: K-wave (dK)
  const int lane = tid & 63, kc = lane & 31, hi = lane >> 5;
  uint32_t cblk, head, batch;
  uint32_t bid = blockIdx.x, split = 0;
  if constexpr (SPLIT) { split = bid % grid.splits; bid /= grid.splits; }
  fwd16_decode_block(grid, bid, &cblk, &head, &batch);
  int R = a.R, C = a.C;
  const int Dr = a.D;
  batch_lengths(a, batch, R, C);
  if (!SPLIT && (int64_t)cblk * WGCOLS >= C) return;   // padded batch entry: the whole workgroup lies beyond its keys
  const int64_t c0 = (int64_t)cblk * WGCOLS + pair * 32;
  const int64_t col = c0 + kc;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldg2 = (uint32_t)a.op[SLOT_dO].ld * 2;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)R * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_dO], head, batch), 0, (uint32_t)R * ldg2, 0x00020000);
  const char *lbase = operand_base(a.op[SLOT_L], head, batch);
  const char *dbase = operand_base(a.op[SLOT_D], head, batch);

  // ---- cached operand of this wave's first product: K (V-wave) or V (K-wave) fragments, B operands
  v8 cf[NKS];
  {
    const int slot = role ? SLOT_V : SLOT_K;
    const uint32_t ld2 = (uint32_t)a.op[slot].ld * 2;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[slot], head, batch), 0, (uint32_t)C * ld2, 0x00020000);
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int d0 = 16 * s + 8 * hi;
      const bool ok = d0 < Dr && col < C;
      cf[s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(res, ok ? (uint32_t)col * ld2 + d0 * 2 : OOB, 0, 0));
    }
  }

  // CAUSAL (extension): the traversal starts at the first row block that sees the workgroup's first key
  const int coff = causal_offset(R, C);
  int block0 = CAUSAL ? (int)(max((int64_t)0, (int64_t)cblk * WGCOLS - coff) / 32) : 0;   // (SPARSE: first row block of the current run)
  int block_end = (R + 31) / 32;
  if constexpr (SPLIT) {   // this workgroup's piece of the row blocks [block0, block_end)
    static_assert(!(SPLIT && SPARSE), "masked launches are not split");
    const int nb = block_end - block0;
    const int lo = block0 + (int)((uint64_t)split * nb / grid.splits);
    block_end = block0 + (int)((uint64_t)(split + 1) * nb / grid.splits);
    block0 = lo;
  }

  // ---- Q / dO staging + L, D slices; SCH 16-byte chunks per thread and operand
  uint32_t qbase0[SCH], gbase0[SCH], qoff[SCH], goff[SCH], wlds[SCH];   // *base0: row block 0
  bool stager[SCH];
#pragma unroll
  for (int i = 0; i < SCH; ++i) {
    const int id = tid + i * NT;
    stager[i] = id < NCHUNK;
    const int srow = id / CPR, sc = id % CPR;
    const bool svalid = stager[i] && sc * 8 < Dr;
    qbase0[i] = svalid ? srow * ldq2 + sc * 16 : OOB;
    gbase0[i] = svalid ? srow * ldg2 + sc * 16 : OOB;
    qoff[i] = __builtin_elementwise_add_sat(qbase0[i], (uint32_t)block0 * BR * ldq2);
    goff[i] = __builtin_elementwise_add_sat(gbase0[i], (uint32_t)block0 * BR * ldg2);
    wlds[i] = ((sc >> 2) * BR + srow) * 64 + (((sc & 3) ^ ((srow >> 2) & 3)) * 16);   // Q at +0, dO at +TILE
  }
  const uint32_t qinc = BR * ldq2, ginc = BR * ldg2;
  u32x4 qreg[SCH], greg[SCH];
  // L / D value of the block being loaded, AS LOADED: no instruction touches it before it is written to LDS a step later
  // (converting at the load made hipcc wait for it -- and for every tile load issued before it -- inside issue_loads)
  // One code path for the three storage types: two 16-bit loads (FP32: the two halves; 16-bit types: the element twice) --
  // a branch per type left hipcc copying in-flight registers at the merge, i.e. waiting again.
  uint16_t ldlo = 0, ldhi = 0;
  // L (wave 0) and D (wave 1) slices of a row block, 32 lanes each: one uniform resource and precision per
  // wave, rows past R read as zero through the resource bounds   (+Softmax.swift:356-381, :472-503)
  const bool ldloader = wave < 2 && lane < 32;
  const int ldslot = wave == 0 ? SLOT_L : SLOT_D;
  const int ldprec = a.op[ldslot].precision;
  const uint32_t ldesz = ldprec == PREC_FP32 ? 4u : 2u;
  const __amdgpu_buffer_rsrc_t ldres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char *>(wave == 0 ? lbase : dbase), 0, (uint32_t)R * ldesz, 0x00020000);
  uint32_t ldoff = (uint32_t)(block0 * BR + lane) * ldesz;
  // SPARSE (block-mask extension): the traversal visits the 32-row blocks of the ACTIVE 256-row blocks only,
  // as one continuous sequence (no pipeline restart at the gaps): step n works on row block blk(n), looked up
  // in a table of active 256-row blocks that thread 0 builds in LDS behind the exchange buffer.
  uint16_t *act = reinterpret_cast<uint16_t *>(smem + XBUF + NPAIR * 2 * 4096 + 16);
  int nact = 0, nload = 0;
  auto blk = [&](int n) { return (int)act[n >> 3] * 8 + (n & 7); };
  auto issue_loads = [&]() {
    if constexpr (SPARSE) {
      const bool in = (nload >> 3) < nact;
      const uint32_t b = in ? (uint32_t)blk(nload) : 0u;
      ++nload;
#pragma unroll
      for (int i = 0; i < SCH; ++i) {
        qoff[i] = in ? __builtin_elementwise_add_sat(qbase0[i], b * BR * ldq2) : OOB;
        goff[i] = in ? __builtin_elementwise_add_sat(gbase0[i], b * BR * ldg2) : OOB;
      }
      ldoff = in ? (b * BR + lane) * ldesz : OOB;
    }
#pragma unroll
    for (int i = 0; i < SCH; ++i) {
      qreg[i] = __builtin_amdgcn_raw_buffer_load_b128(qres, qoff[i], 0, 0);
      greg[i] = __builtin_amdgcn_raw_buffer_load_b128(gres, goff[i], 0, 0);
      qoff[i] = __builtin_elementwise_add_sat(qoff[i], qinc);
      goff[i] = __builtin_elementwise_add_sat(goff[i], ginc);
    }
    if (ldloader) {
      ldlo = __builtin_amdgcn_raw_buffer_load_b16(ldres, ldoff, 0, 0);
      ldhi = __builtin_amdgcn_raw_buffer_load_b16(ldres, ldoff + (ldesz - 2u), 0, 0);
      ldoff += BR * ldesz;
    }
  };
  auto write_tiles = [&](int stage) {
    char *base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < SCH; ++i)
      if (stager[i]) {
        *reinterpret_cast<u32x4 *>(base + wlds[i]) = qreg[i];
        *reinterpret_cast<u32x4 *>(base + TILE + wlds[i]) = __builtin_bit_cast(u32x4, convert_chunk<T, TG>(greg[i]));
      }
    if (ldloader) {
      const float ldval = ldprec == PREC_FP32 ? __builtin_bit_cast(float, (uint32_t)ldlo | ((uint32_t)ldhi << 16))
                        : ldprec == PREC_FP16 ? (float)__builtin_bit_cast(_Float16, ldlo) : bf16_bits_to_f32(ldlo);
      reinterpret_cast<float *>(base + 2 * TILE)[wave * 32 + lane] = ldval;
    }
  };

  // row fragment of k-step t (16 elements from d = 16t; this lane: row kc, elements 16t + 8hi .. +7):
  // d-block t >> 1, chunk 2 (t & 1) + hi of the row, swizzled
  int fread[NKS];
#pragma unroll
  for (int t = 0; t < NKS; ++t) fread[t] = ((t >> 1) * BR + kc) * 64 + (((2 * (t & 1) + hi) ^ ((kc >> 2) & 3)) * 16);
  // transposing read of rows 16u + {0..15} of d-block db: two ds_read_b64_tr_b16, rows (n16>>2) + 4hi and
  // + 8; this lane's 8-byte piece is number (n16 & 3) of 32-byte half (lane >> 4) & 1 of the row
  const int n16 = lane & 15;
  const int trow = (n16 >> 2) + 4 * hi, tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1), thalf = (n16 & 3) & 1;
  const int tr0 = trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8;                  // (row >> 2) & 3 == hi
  const int tr1 = (trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8;      // row + 8
  char *xb = smem + XBUF + pair * (2 * 4096) + lane * 16;   // + parity * 4096 + g * 1024

  f32x16 acc[NDB];   // dV^T (V-wave) or dK^T (K-wave): lane = key, registers = head-dimension rows
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

  int nblocks = block_end - block0;   // (SPARSE: row blocks of the current run)
  auto staging = [&](int t) {   // after the barrier: row block t+2 -> LDS (replaces t-2, last read in step t-1), loads of t+3 (zeros past the end)
    if constexpr (ABL != 3) __syncthreads();
    if constexpr (ABL != 2) write_tiles((t + 2) & 3);
    if constexpr (ABL != 1 && ABL != 2) issue_loads();
  };

  // Everything below is instantiated once per role so that each wave's step is straight-line code (the
  // compiler interleaves the first product's MFMAs with the arithmetic of the previous block only inside
  // one basic block).  Both instances pass the same number of barriers.
  auto run = [&](auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr int img_row = ROLE ? TILE : 0;     // first product:  K-wave dO rows, V-wave Q rows
    constexpr int img_tr = ROLE ? 0 : TILE;      // second product: K-wave Q^T,     V-wave dO^T
    auto read_rows = [&](int stage, v8 (&af)[NKS]) {
      const char *st = smem + stage * STAGE + img_row;
#pragma unroll
      for (int t = 0; t < NKS; ++t) af[t] = *reinterpret_cast<const v8 *>(st + fread[t]);
    };
    auto first_product = [&](const v8 (&af)[NKS], f32x16 &x) {
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = 0.f;
#pragma unroll
      for (int t = 0; t < NKS; ++t) x = F::mfma(af[t], cf[t], x);
    };
    // V-wave: P = exp2(S*scale2 - L) (+Softmax.swift:409-417), fp32 copy to the exchange buffer;
    // K-wave: dS = P (dP*scale - D) (+Softmax.swift:419-427) with the partner's P of one step ago
    auto arithmetic = [&](int cur, const f32x16 &x, v8 (&frag)[2]) {
      const float *Ls = reinterpret_cast<const float *>(smem + (cur & 3) * STAGE + 2 * TILE) + (ROLE ? 32 : 0) + 4 * hi;
      char *xp = xb + (cur & 1) * 4096;
      if constexpr (ROLE == 0) {
        const int row0 = (SPARSE ? blk(cur) : block0 + cur) * BR;
        const bool diag = CAUSAL && (c0 + 31 > row0 + coff);   // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = ABL == 4 ? f32x4{1.f, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4 *>(Ls + 8 * g);
          f32x4 p4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            float p = fast_exp2(x[r] * a.scale2 - l4[i]);
            if (CAUSAL && diag && col > row0 + crow(r, hi) + coff) p = 0.f;
            p4[i] = p;
            frag[g >> 1][4 * (g & 1) + i] = (T)p;
          }
          if constexpr (ABL != 4) *reinterpret_cast<f32x4 *>(xp + g * 1024) = p4;
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 d4 = ABL == 4 ? f32x4{1.f, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4 *>(Ls + 8 * g);
          const f32x4 p4 = ABL == 4 ? f32x4{.1f, .2f, .3f, .4f} : *reinterpret_cast<const f32x4 *>(xp + g * 1024);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            frag[g >> 1][4 * (g & 1) + i] = (T)(p4[i] * (x[r] * a.scale - d4[i]));
          }
        }
      }
    };
    auto load_tr = [&](int stage, int u, v8 (&tf)[NDB]) {
      const char *tp = smem + stage * STAGE + img_tr + 16 * u * 64;
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const char *p = tp + db * BR * 64;
        const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + tr0));
        const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + tr1));
        tf[db] = __builtin_bit_cast(v8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    auto second_product = [&](int cur, const v8 (&tf0)[NDB], const v8 (&frag)[2]) {
      v8 tf1[NDB];
#pragma unroll
      for (int db = 0; db < NDB; ++db) acc[db] = F::mfma(tf0[db], frag[0], acc[db]);
      load_tr(cur & 3, 1, tf1);
#pragma unroll
      for (int db = 0; db < NDB; ++db) acc[db] = F::mfma(tf1[db], frag[1], acc[db]);
    };
    // full step: all fragment reads first, then first product of `nxt` | arithmetic of `cur`, second product of `cur`
    auto full = [&](int t, int nxt, int cur, f32x16 &x_n, const f32x16 &x_c) {
      v8 af[NKS], tf0[NDB], frag[2];
      staging(t);   // (doing the LDS writes and loads after the first product's MFMAs instead: +-0 measured)
      read_rows(nxt & 3, af);
      load_tr(cur & 3, 0, tf0);
      __builtin_amdgcn_sched_barrier(0);
      first_product(af, x_n);
      arithmetic(cur, x_c, frag);
      second_product(cur, tf0, frag);
    };
    auto only_next = [&](int nxt, f32x16 &x_n) {
      v8 af[NKS];
      read_rows(nxt & 3, af);
      first_product(af, x_n);
    };
    auto only_cur = [&](int cur, const f32x16 &x_c) {
      v8 tf0[NDB], frag[2];
      load_tr(cur & 3, 0, tf0);
      arithmetic(cur, x_c, frag);
      second_product(cur, tf0, frag);
    };
    // Steps t = 0 .. nblocks, staging(t) in every one of them for both roles.  V-waves: (next, current) =
    // (t+1, t), full steps 0 .. nblocks-2; K-waves: (t, t-1), full steps 1 .. nblocks-1.  The first and last
    // steps are peeled and the full steps come in pairs, so the two score register sets swap roles without
    // a single register copy (a loop with per-step conditions costs 27 tuple copies per two steps).
    f32x16 xa, xb2;
    if constexpr (ROLE == 0) {
      only_next(0, xa);                          // row blocks 0 and 1 are in LDS after the prologue
      int t = 0;
      for (; t + 2 <= nblocks - 1; t += 2) {
        full(t, t + 1, t, xb2, xa);
        full(t + 1, t + 2, t + 1, xa, xb2);
      }
      if (t <= nblocks - 2) {
        full(t, t + 1, t, xb2, xa); ++t;
        staging(t); only_cur(t, xb2);
      } else {
        staging(t); only_cur(t, xa);
      }
      staging(nblocks);
    } else {
      staging(0); only_next(0, xa);
      int t = 1;
      for (; t + 1 <= nblocks - 1; t += 2) {
        full(t, t, t - 1, xb2, xa);
        full(t + 1, t + 1, t, xa, xb2);
      }
      if (t <= nblocks - 1) {
        full(t, t, t - 1, xb2, xa); ++t;
        staging(t); only_cur(t - 1, xb2);
      } else {
        staging(t); only_cur(t - 1, xa);
      }
    }
  };
  // one contiguous run of row blocks [block0, block0 + nblocks): prologue + this wave's role
  auto traverse_rows = [&]() {
    issue_loads();
    write_tiles(0);
    issue_loads();
    write_tiles(1);
    issue_loads();              // row block 2 in flight
    __syncthreads();
    if (role == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
  };
  if constexpr (!SPARSE) {
    if (nblocks > 0) traverse_rows();
  } else {
    // block mask: bit (row block of 256 rows = 8 steps, column block of 128 keys = this workgroup)
    const uint32_t mcb = (uint32_t)(((uint64_t)cblk * WGCOLS) >> 7);   // 128-column block of the mask this workgroup lies in
    const uint32_t *mcol = a.mask + (int64_t)head * a.maskHeadStride + (int64_t)batch * a.maskBatchStride + (mcb >> 5);
    const int rb_end = ((R + BR - 1) / BR + 7) / 8;
    int *count = reinterpret_cast<int *>(smem + XBUF + NPAIR * 2 * 4096);
    if (tid == 0) {
      int n = 0;
      for (int rb = block0 / 8; rb < rb_end && n < DKV16RS_MAX_ROW_BLOCKS; ++rb)   // (causal: rows before block0 see none of these keys)
        if ((mcol[(uint64_t)rb * a.maskWords] >> (mcb & 31)) & 1u) act[n++] = (uint16_t)rb;
      *count = n;
    }
    __syncthreads();
    nact = *count;
    block0 = 0;             // unused by the sparse addressing
    nblocks = 8 * nact;     // blocks past R inside the last active group are all zeros and contribute nothing
    if (nact > 0) traverse_rows();
  }

  // ---- epilogue: this wave's accumulator through LDS (whole-row stores): dV (V-wave) or dK (K-wave)
  __syncthreads();
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + wave * (32 * OLD);
  float *orow = Os + kc * OLD;
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) =
          make_float4(acc[db][4 * g], acc[db][4 * g + 1], acc[db][4 * g + 2], acc[db][4 * g + 3]);
  const int slot = role ? SLOT_dK : SLOT_dV;
  if constexpr (SPLIT) {
    const size_t hb = (size_t)grid.heads * grid.batches;
    const size_t slab = (((size_t)role * grid.splits + split) * hb + (size_t)batch * grid.heads + head) * (size_t)a.C;
    store_block_rows<T, D>(Os, reinterpret_cast<char *>(grid.wsO + slab * Dr), PREC_FP32, (uint32_t)Dr, c0, C, Dr, lane);
  } else {
    store_block_rows<T, D>(Os, operand_base(a.op[slot], head, batch), a.op[slot].precision, (uint32_t)a.op[slot].ld, c0, C, Dr, lane);
  }
}

} // namespace mfa
