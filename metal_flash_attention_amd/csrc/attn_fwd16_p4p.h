// attn_fwd16_p4p.h -- PERSISTENT form of attn_fwd16_p4 (forward attention, D <= 128, 16-bit Q/K/V, four waves x 64 rows): one
// workgroup per compute unit walks the 256-row blocks blockIdx, blockIdx + gridDim, ... and the block loop is INSIDE the
// generated asm statement (tools/p4pgen.py -> attn_fwd16_p4p_stream.inc; the design and what it removes from the per-block
// cost are in the generator's header and DESIGN.md 4.2.0).
//
//   reference: loopForward + createSetup / createCleanup, Sources/FlashAttention/Attention/AttentionKernel/
//   AttentionKernel+Source.swift:158-200, +Caching.swift:286-425 -- same math as attn_fwd16_p4.h: the stream's tile
//   traversal IS p4gen's; new are the block switch (next block's Q / K / V requested by LDS-DMA under the last two tiles), O / l
//   and L = m + log2 l stored straight from the registers, and the block table.
//
// What is left to hipcc: the block table (64-byte entries in LDS: operand bases of the block's head, first row), the lane
// constants of the LDS-DMA and store addressing, the scalar inputs.  Nothing is live after the statement.
// Dense launches only (no causal mask, no per-batch lengths, no block mask): those keep attn_fwd16_p4.
#pragma once
#include "attn_fwd16_p4.h"
#include "attn_fwd16_p4p_stream.inc"

namespace mfa {
namespace p4p {

constexpr int QIMG = MFA_P4P_QIMG, TABLE = MFA_P4P_TABLE, TABLE_ENTRIES = MFA_P4P_TABLE_ENTRIES, LDS_BYTES = MFA_P4P_LDS_BYTES;

#define MFA_P4P_ENUM(name, f16, fold, o16, l16, causal) S_##name,
enum : int { MFA_P4P_STREAM_LIST(MFA_P4P_ENUM) S_COUNT };
#undef MFA_P4P_ENUM

// (the stream list's last column: & 3: 1 = causal / geometry stream, 2 = column-parallel pieces; & 4: O = P V with lane = column, rows stored
// straight from the registers -- PCfg.orow of tools/p4pgen.py)
struct StreamTraits { bool f16, fold, o16, l16, causal, split, orow; };
constexpr StreamTraits traits(int s) {
#define MFA_P4P_TRAITS(name, f16, fold, o16, l16, kind) if (s == S_##name) return StreamTraits{f16 != 0, fold != 0, o16 != 0, l16 != 0, ((kind) & 3) == 1, ((kind) & 3) == 2, ((kind) & 4) != 0};
  MFA_P4P_STREAM_LIST(MFA_P4P_TRAITS)
#undef MFA_P4P_TRAITS
  return StreamTraits{false, false, false, false, false, false, false};
}

}  // namespace p4p

#define MFA_P4P_RUN_STREAM(STREAM)                                                                                       \
  asm volatile(STREAM                                                                                                    \
               : [lim0] "+v"(lim0), [lim1] "+v"(lim1)                                                                     \
               : [kbase] "v"(kbase), [vbase] "v"(vbase), [kv0] "v"(kv[0]), [kv1] "v"(kv[1]),                              \
                 [kv2] "v"(kv[2]), [kv3] "v"(kv[3]), [vv] "v"(vv), [qv0] "v"(qv[0]), [qv1] "v"(qv[1]), [qv2] "v"(qv[2]),  \
                 [qv3] "v"(qv[3]), [ov0] "v"(ov[0]), [ov1] "v"(ov[1]), [ov2] "v"(ov[2]), [ov3] "v"(ov[3]), [lv] "v"(lv),  \
                 [ewa] "v"(ewa), [era] "v"(era), [qlane] "v"(qlane), [hi4] "v"(hi4),                                      \
                 [nt] "s"(nt), [maskfrom] "s"(maskfrom), [scale2] "s"(a.scale2), [kinc] "s"(kinc), [vinc] "s"(vinc),      \
                 [ldsk] "s"(ldsk), [ldsv] "s"(ldsv), [ldsq] "s"(ldsq), [qrel] "s"(qrel), [nblk] "s"(nblk), [tbl] "s"(tbl), \
                 [wave64] "s"(wave64), [ldq2] "s"(ldq2), [ldo] "s"(ldob), [nrecq] "s"(nrecq), [nreck] "s"(nreck),         \
                 [nrecv] "s"(nrecv), [nreco] "s"(nreco), [nrecl] "s"(nrecl), [dr] "s"(dr), [cflag] "s"(cflag)             \
               : "memory", "vcc", "scc", MFA_ALL_AGPRS, MFA_P4P_OWNED_VGPRS, MFA_P4P_OWNED_SGPRS)

// T: __bf16 or _Float16 (must match the stream); STREAM: p4p::S_*.  `total` = units x heads x batches; workgroup w of G takes the
// units w, w + G, ... in fwd16_decode_block's order (G a multiple of 8: a workgroup stays with the heads of its XCD).  A unit is
// one row block (dense streams) or the PAIR of row blocks (last - i, i) (causal streams: row block i walks ~4 (i + 1) key tiles,
// every pair the same number), the long one first
template <typename T, int STREAM>
__global__ __launch_bounds__(256) void attn_fwd16_p4p(const KernelArgs a, const Fwd16Grid grid, const uint32_t total, const uint32_t stagger) {
  using namespace p4p;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr StreamTraits TR = traits(STREAM);
  static_assert(TR.f16 == __is_same(T, _Float16), "stream and element type disagree");
  constexpr int BC = 64, GROWS = 256;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const int tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  const uint32_t G = gridDim.x, first = blockIdx.x;
  if (first >= total) return;
  const uint32_t nunits = (total - first + G - 1) / G;   // blocks per unit x nunits <= TABLE_ENTRIES (the launcher sizes the grid)

  // ---- block table: what attn_fwd16_p4 decodes per workgroup, once per block of this workgroup.  Entry = Q, K, V, O, L base of the
  // block's head (words 0..9), its first row (10) and -- causal ("geometry") streams, round 6 -- the rows and keys of its batch entry
  // (11, 12: per-batch lengths; the launch's R, C otherwise).  Row blocks beyond an entry's rows are NOT entered: the table is
  // compacted (unit n's entries start behind those of the units before it: counts through LDS, then a prefix sum).
  uint32_t *table = reinterpret_cast<uint32_t *>(smem + TABLE);
  Fwd16Grid dgrid = grid;
  const uint32_t RB = grid.rowBlocks;
  if constexpr (TR.causal) dgrid.rowBlocks = (RB + 1) / 2;
  uint32_t *counts = reinterpret_cast<uint32_t *>(smem);   // (the K ring's first bytes: free until the stream starts)
  uint32_t myrows[2] = {0, 0}, myhead = 0, mybatch = 0;
  int mycount = 0, myR = (int)a.R, myC = (int)a.C;
  uint32_t mypiece = 0;
  if ((uint32_t)tid < nunits) {
    uint32_t r, unit = first + (uint32_t)tid * G;
    if constexpr (TR.split) { mypiece = unit % grid.splits; unit /= grid.splits; }   // SPLIT: a unit is (row block, piece of the key range)
    fwd16_decode_block_lane(dgrid, unit, &r, &myhead, &mybatch);
    if constexpr (TR.causal) batch_lengths(a, mybatch, myR, myC);
    const uint32_t cand[2] = {TR.causal ? RB - 1 - r : r, r};
    const int ncand = (TR.causal && cand[0] != cand[1]) ? 2 : 1;
    for (int w = 0; w < ncand; ++w)
      if ((int64_t)cand[w] * GROWS < myR) myrows[mycount++] = cand[w];
    counts[tid] = (uint32_t)mycount;
  }
  __syncthreads();
  if ((uint32_t)tid < nunits) {
    uint32_t pos = 0;
    for (int i = 0; i < tid; ++i) pos += counts[i];
    uint64_t base[5] = {(uint64_t)(uintptr_t)operand_base(a.op[SLOT_Q], myhead, mybatch), (uint64_t)(uintptr_t)operand_base(a.op[SLOT_K], myhead, mybatch),
                        (uint64_t)(uintptr_t)operand_base(a.op[SLOT_V], myhead, mybatch), (uint64_t)(uintptr_t)operand_base(a.op[SLOT_O], myhead, mybatch),
                        (uint64_t)(uintptr_t)operand_base(a.op[SLOT_L], myhead, mybatch)};
    if constexpr (TR.split) {
      // K / V start at the piece (C / splits keys, a multiple of 128: the launcher checks); O and (m, l) go to the piece's slabs of
      // the caller's workspace: wsO [splits][heads x batches][R][D] fp32, wsML [splits][heads x batches][R][2] (attn_fwd_combine)
      const uint64_t keys = (uint64_t)mypiece * (a.C / grid.splits);
      base[1] += keys * (uint64_t)a.op[SLOT_K].ld * 2;
      base[2] += keys * (uint64_t)a.op[SLOT_V].ld * 2;
      const uint64_t slab = ((uint64_t)mypiece * grid.heads * grid.batches + (uint64_t)mybatch * grid.heads + myhead) * a.R;
      base[3] = (uint64_t)(uintptr_t)(grid.wsO + slab * a.D);
      base[4] = (uint64_t)(uintptr_t)(grid.wsML + slab * 2);
    }
    for (int w = 0; w < mycount; ++w) {
      uint32_t *e = table + 16 * (pos + w);
#pragma unroll
      for (int i = 0; i < 5; ++i) { e[2 * i] = (uint32_t)base[i]; e[2 * i + 1] = (uint32_t)(base[i] >> 32); }
      e[10] = myrows[w] * GROWS;
      e[11] = (uint32_t)myR;
      e[12] = (uint32_t)myC;
    }
    if ((uint32_t)tid == nunits - 1) table[16 * TABLE_ENTRIES - 1] = pos + (uint32_t)mycount;   // blocks of this workgroup (the last table word is never an entry's)
  }
  __syncthreads();
  const uint32_t nblk = __builtin_amdgcn_readfirstlane(table[16 * TABLE_ENTRIES - 1]);
  if (nblk == 0) return;   // (per-batch lengths: every row block of this workgroup's units lies beyond its entry's rows)
  // desynchronise the compute units: blocks that end in lockstep store 32 MB at once and the next block's loads queue
  // behind them (profiles/r02_fwd16p4_block_overhead_persistent_experiment.txt)
  for (uint32_t i = 0; i < (stagger & 0xFFFFu) * ((first >> 3) & 31u); ++i) __builtin_amdgcn_s_sleep(8);   // 512 clocks per step

  // (SPLIT: every workgroup sees its piece as the key range; the division runs on the vector ALU and hipcc does not move its result
  // back to a scalar register by itself when the asm statement asks for "s" operands derived from it)
  const uint32_t R = a.R, C = TR.split ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.C / grid.splits)) : a.C, dr = a.D;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2, ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;
  constexpr uint32_t OSZ = TR.o16 ? 2 : 4, LSZ = TR.split ? 8 : TR.l16 ? 2 : 4;
  const uint32_t ldob = TR.split ? dr * 4 : (uint32_t)a.op[SLOT_O].ld * OSZ;
  const uint32_t nrecq = R * ldq2, nreck = C * ldk2, nrecv = C * ldv2, nreco = R * ldob, nrecl = R * LSZ;
  const uint32_t kinc = BC * ldk2, vinc = BC * ldv2;
  // a block walks an EVEN number of key tiles (an odd count gets one fully masked tile): the two K images and the score-tile
  // parity then line up from block to block.  Tiles from `maskfrom` on hold keys >= C (maskAttentionMatrixEdge, +Softmax.swift:228-260)
  uint32_t nt = (C + BC - 1) / BC;
  nt += nt & 1u;
  const uint32_t maskfrom = C / BC;
  // register r of a lane covers key (r & 3) + 8 (r >> 2) + 4 hi of its 32-key block; dense: every row sees all C keys (causal
  // streams recompute both limits per block: min(C - 1, row + C - R) - 4 hi)
  int lim0 = (int)C - 1 - 4 * hi, lim1 = lim0;
  const uint32_t qlane = q, hi4 = 4 * hi;
  const uint32_t cflag = (uint32_t)__builtin_amdgcn_readfirstlane(a.causal ? 1 : 0);

  // ---- lane parts of the LDS-DMA source offsets (the stream adds the scalar parts: first row of the piece x leading dimension).
  // Piece i of a K-shaped image (K tiles, the wave's Q image): 16-byte position p = i * 64 + lane holds row p >> 4, chunk
  // (p & 15) ^ (row & 15); a V image holds [D/32][64 keys][32 d] sub-tiles (attn_fwd16_p4.h)
  uint32_t kv[4], qv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t kc = (lane & 15) ^ ((4 * i + (lane >> 4)) & 15);
    kv[i] = kc * 8 < dr ? (uint32_t)(lane >> 4) * ldk2 + kc * 16 : OOB;
    qv[i] = kc * 8 < dr ? (uint32_t)(lane >> 4) * ldq2 + kc * 16 : OOB;
  }
  const uint32_t vc = wave * 4 + (lane & 3);
  const uint32_t vv = vc * 8 < dr ? (uint32_t)(lane >> 2) * ldv2 + vc * 16 : OOB;
  // stores: a 32 x 32 block of O^T goes through a 4 KiB slice of LDS (in: lane = row q, 16-byte chunk (2 g + hi) ^ (q & 7) of
  // its 128-byte row; out: lane = (row & 7, chunk)), so that eight lanes cover one 128-byte line of a row; columns >= D are out of range
  const uint32_t row8 = lane >> 3, chunk = lane & 7;
  const uint32_t ewa = (uint32_t)q * 128 + ((hi ^ (q & 7)) << 4), era = row8 * 128 + ((chunk ^ row8) << 4);
  uint32_t ov[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const uint32_t col = 32 * db + 4 * chunk;
    ov[db] = col < dr ? row8 * ldob + col * OSZ : OOB;
    // orow streams: no LDS trip -- register r of accumulator block (rb, db) is column 32 db + q of row 8 (r >> 2) + 4 hi + (r & 3); the row
    // is the store's scalar offset, the lane brings its column and the four rows of the upper half-wave
    if constexpr (TR.orow) ov[db] = 32 * db + (uint32_t)q < dr ? (32 * db + (uint32_t)q) * 4 + (uint32_t)hi * 4 * ldob : OOB;
  }
  const uint32_t lv = hi == 0 ? (uint32_t)q * LSZ : OOB;   // L: one lane per row

  const uint32_t lds0 = lds_addr(smem);
  const uint32_t kbase = lds0 + q * 256 + ((hi ^ (q & 15)) << 4);
  const int n16 = lane & 15;
  const uint32_t vbase = lds0 + p4::VBASE + ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;
  const uint32_t ldsk = lds0 + wave * 4096, ldsv = lds0 + p4::VBASE + wave * 4096;
  const uint32_t qrel = QIMG + wave * 16384, ldsq = lds0 + qrel, tbl = lds0 + TABLE, wave64 = wave * 64;

#define MFA_P4P_RUN(name, f16, fold, o16, l16, causal) if constexpr (STREAM == S_##name) MFA_P4P_RUN_STREAM(MFA_P4P_STREAM_##name);
  MFA_P4P_STREAM_LIST(MFA_P4P_RUN)
#undef MFA_P4P_RUN
}

} // namespace mfa
