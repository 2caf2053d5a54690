// attn_fwd16_p6.h -- forward attention for head dimensions D <= 64 (BASELINE config 2), 16-bit Q/K/V: four waves x 64 rows, one
// wave per SIMD, PERSISTENT (one workgroup per compute unit walks the 256-row blocks), the whole block loop ONE generated asm
// statement (tools/p6gen.py -> attn_fwd16_p6_stream.inc; design in the generator's header and DESIGN.md 4.0):
//   * the row sums of the softmax run in the matrix pipe (L^T += ONES P^T beside O^T += V^T P^T): at D = 64 the kernel is bound
//     by instruction issue, not by matrix time, and 8 matrix instructions replace 64 vector additions per tile;
//   * K / V tiles by LDS-DMA two tiles ahead into rings of four 8 KiB images, the pieces first in phase B;
//   * O^T through a staging area of its own into line-sized stores.
//
//   reference: loopForward + createSetup / createCleanup, Sources/FlashAttention/Attention/AttentionKernel/
//   AttentionKernel+Source.swift:158-200, +Caching.swift:286-425, +Softmax.swift:267-324 (online max / correct / sum); the
//   D <= 64 rows of its mixed forward table (AttentionDescriptor+Parameters.swift:106-121) are the reference's fastest.
//
// FOLD streams serve mixed-precision descriptors (lowPrecisionIntermediates: Q' = Q * log2(e)/sqrt(D) in the 16-bit type, the running
// maximum subtracted inside the matrix pipe, row sums of the 16-bit P in the matrix pipe, FP16 L); EXACT streams the descriptors
// that keep the attention matrix in FP32 registers (scale per score in fp32, fp32 row sums of the unrounded P, FP32 L).  Row-major
// operands; dense launches without per-batch lengths, masks or causal flag -- everything else stays with attn_fwd16_v3.  What is
// left to hipcc: the block table, lane constants, scalar inputs.
#pragma once
#include "attn_fwd16_v3.h"
#include "agpr_list.h"
#include "attn_fwd16_p6_stream.inc"

namespace mfa {
namespace p6 {

constexpr int VRING = MFA_P6_VRING, QIMG = MFA_P6_QIMG, TABLE = MFA_P6_TABLE, TABLE_ENTRIES = MFA_P6_TABLE_ENTRIES, STAGE = MFA_P6_STAGE,
              LDS_BYTES = MFA_P6_LDS_BYTES;

#define MFA_P6_ENUM(name, f16, fold, o16, l16, causal, split) S_##name,
enum : int { MFA_P6_STREAM_LIST(MFA_P6_ENUM) S_COUNT };
#undef MFA_P6_ENUM

struct StreamTraits { bool f16, fold, o16, l16, causal, split; };
constexpr StreamTraits traits(int s) {
#define MFA_P6_TRAITS(name, f16, fold, o16, l16, causal, split) if (s == S_##name) return StreamTraits{f16 != 0, fold != 0, o16 != 0, l16 != 0, causal != 0, split != 0};
  MFA_P6_STREAM_LIST(MFA_P6_TRAITS)
#undef MFA_P6_TRAITS
  return StreamTraits{false, false, false, false, false, false};
}

}  // namespace p6

#define MFA_P6_RUN_STREAM(STREAM)                                                                                        \
  asm volatile(STREAM                                                                                                    \
               : [lim0] "+v"(lim0), [lim1] "+v"(lim1)                                                                     \
               : [kbase] "v"(kbase), [vbase] "v"(vbase), [kv0] "v"(kv[0]), [kv1] "v"(kv[1]), [vv] "v"(vv),                \
                 [qv0] "v"(qv[0]), [qv1] "v"(qv[1]), [ov0] "v"(ov[0]), [ov1] "v"(ov[1]), [lv] "v"(lv),                    \
                 [ewa] "v"(ewa), [era] "v"(era), [qlane] "v"(qlane), [hi4] "v"(hi4),                                      \
                 [nt] "s"(nt), [maskfrom] "s"(maskfrom), [scale2] "s"(a.scale2), [kinc] "s"(kinc), [vinc] "s"(vinc),      \
                 [ldsk] "s"(ldsk), [ldsv] "s"(ldsv), [ldsq] "s"(ldsq), [qrel] "s"(qrel), [ldsst] "s"(ldsst),              \
                 [nblk] "s"(nblk), [tbl] "s"(tbl), [wave64] "s"(wave64), [ldq2] "s"(ldq2), [ldo] "s"(ldob),               \
                 [nrecq] "s"(nrecq), [nreck] "s"(nreck), [nrecv] "s"(nrecv), [nreco] "s"(nreco), [nrecl] "s"(nrecl),      \
                 [cflag] "s"(cflag)                                                                                        \
               : "memory", "vcc", "scc", MFA_ALL_AGPRS, MFA_P6_OWNED_VGPRS, MFA_P6_OWNED_SGPRS)

// T: __bf16 or _Float16 (must match the stream); STREAM: p6::S_*.  `total` = row blocks x heads x batches; workgroup w of G takes
// the blocks w, w + G, ... in fwd16_decode_block's order (G a multiple of 8: a workgroup stays with the heads of its XCD)
template <typename T, int STREAM>
__global__ __launch_bounds__(256) void attn_fwd16_p6(const KernelArgs a, const Fwd16Grid grid, const uint32_t total) {
  using namespace p6;
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
  const uint32_t nunits = (total - first + G - 1) / G;   // blocks per unit x nunits <= TABLE_ENTRIES - 1 (the launcher sizes the grid)

  // ---- block table (64-byte entries: Q, K, V, O, L base of the block's head, first row), once per workgroup.  A unit is one row
  // block (dense streams) or the PAIR of row blocks (last - i, i) (causal streams: row block i walks 4 (i + 1) key tiles, every pair
  // the same number), the long one first
  uint32_t *table = reinterpret_cast<uint32_t *>(smem + TABLE);
  Fwd16Grid dgrid = grid;
  const uint32_t RB = grid.rowBlocks;
  if constexpr (TR.causal) dgrid.rowBlocks = (RB + 1) / 2;
  // (round 6, causal / "geometry" streams: an entry also carries the rows and keys of its batch entry -- words 11, 12: per-batch lengths,
  // or the launch's R, C -- and row blocks beyond an entry's rows are NOT entered: the table is compacted, unit n's entries start
  // behind those of the units before it -- counts through LDS, then a prefix sum; nunits <= 255: one unit per thread)
  uint32_t *counts = reinterpret_cast<uint32_t *>(smem);   // (the K ring's first bytes: free until the stream starts)
  uint32_t myrows[2] = {0, 0}, myhead = 0, mybatch = 0, mypiece = 0;
  int mycount = 0, myR = (int)a.R, myC = (int)a.C;
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
    const uint32_t head = myhead, batch = mybatch, piece = mypiece;
    uint32_t pos = 0;
    for (int i = 0; i < tid; ++i) pos += counts[i];
    uint64_t base[5] = {(uint64_t)(uintptr_t)operand_base(a.op[SLOT_Q], head, batch), (uint64_t)(uintptr_t)operand_base(a.op[SLOT_K], head, batch),
                        (uint64_t)(uintptr_t)operand_base(a.op[SLOT_V], head, batch), (uint64_t)(uintptr_t)operand_base(a.op[SLOT_O], head, batch),
                        (uint64_t)(uintptr_t)operand_base(a.op[SLOT_L], head, batch)};
    if constexpr (TR.split) {
      // K / V start at the piece (C / splits keys, a multiple of 256: the launcher checks); O and (m, l) go to the piece's slabs of
      // the caller's workspace: wsO [splits][heads x batches][R][D] fp32, wsML [splits][heads x batches][R][2] (attn_fwd_combine)
      const uint64_t keys = (uint64_t)piece * (a.C / grid.splits);
      base[1] += keys * (uint64_t)a.op[SLOT_K].ld * 2;
      base[2] += keys * (uint64_t)a.op[SLOT_V].ld * 2;
      const uint64_t slab = ((uint64_t)piece * grid.heads * grid.batches + (uint64_t)batch * grid.heads + head) * a.R;
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

  // (SPLIT: every workgroup sees its piece as the key range; the division runs on the vector ALU and hipcc does not move its result
  // back to a scalar register by itself when the asm statement asks for "s" operands derived from it)
  const uint32_t R = a.R, C = TR.split ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.C / grid.splits)) : a.C, dr = a.D;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2, ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;
  constexpr uint32_t OSZ = TR.o16 ? 2 : 4, LSZ = TR.split ? 8 : TR.l16 ? 2 : 4;
  const uint32_t ldob = TR.split ? dr * 4 : (uint32_t)a.op[SLOT_O].ld * OSZ;
  const uint32_t nrecq = R * ldq2, nreck = C * ldk2, nrecv = C * ldv2, nreco = R * ldob, nrecl = R * LSZ;
  const uint32_t kinc = BC * ldk2, vinc = BC * ldv2;
  // a block walks a MULTIPLE OF FOUR key tiles (the loop body is four tiles = the ring of four K / V images: every ring position is
  // an immediate, a block starts in image 0); the surplus tiles are fully masked.  Tiles from `maskfrom` on hold keys >= C
  // (maskAttentionMatrixEdge, +Softmax.swift:228-260)
  uint32_t nt = ((C + BC - 1) / BC + 3u) / 4u * 4u;
  const uint32_t maskfrom = C / BC;
  // register r of a lane covers key (r & 3) + 8 (r >> 2) + 4 hi of its 32-key block; every row sees all C keys
  int lim0 = (int)C - 1 - 4 * hi, lim1 = lim0;   // (causal streams recompute both limits per block: min(C - 1, row + C - R) - 4 hi)
  const uint32_t qlane = q, hi4 = 4 * hi;
  const uint32_t cflag = (uint32_t)a.causal;   // (0 / 1, mfa_kernel.hip; a kernel argument: a scalar register)

  // ---- lane parts of the LDS-DMA source offsets (the stream adds the scalar parts).  K-shaped images (K tiles, the wave's Q
  // image): rows of 128 bytes, a 1 KiB piece = 8 rows; the 16-byte position (lane & 7) of row 8 i + (lane >> 3) holds chunk
  // (lane & 7) ^ ((row >> 1) & 7), and (row >> 1) & 7 = (4 i + (lane >> 4)) & 7 depends on the piece only through its parity.
  // A V image holds [2][64 keys][32 elements] sub-tiles: wave w fills sub-tile w >> 1, keys 32 (w & 1) + 16 i .. + 15 per piece
  uint32_t kv[2], qv[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const uint32_t c = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);
    kv[par] = c * 8 < dr ? (uint32_t)(lane >> 3) * ldk2 + c * 16 : OOB;
    qv[par] = c * 8 < dr ? (uint32_t)(lane >> 3) * ldq2 + c * 16 : OOB;
  }
  const uint32_t vc = (wave >> 1) * 4 + (lane & 3);
  const uint32_t vv = vc * 8 < dr ? ((uint32_t)(lane >> 2) + 32 * (wave & 1)) * ldv2 + vc * 16 : OOB;
  // stores: a 32 x 32 block of O^T goes through the wave's staging slice (in: lane = row q, 16-byte chunk (2 g + hi) ^ (q & 7) of
  // its 128-byte row; out: lane = (row & 7, chunk)), so that eight lanes cover one 128-byte line of a row; columns >= D are out of range
  const uint32_t row8 = lane >> 3, chunk = lane & 7;
  const uint32_t ewa = (uint32_t)q * 128 + ((hi ^ (q & 7)) << 4), era = row8 * 128 + ((chunk ^ row8) << 4);
  uint32_t ov[2];
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const uint32_t col = 32 * db + 4 * chunk;
    ov[db] = col < dr ? row8 * ldob + col * OSZ : OOB;
  }
  const uint32_t lv = hi == 0 ? (uint32_t)q * LSZ : OOB;   // L: one lane per row

  const uint32_t lds0 = lds_addr(smem);
  const uint32_t kbase = lds0 + q * 128 + ((hi ^ ((q >> 1) & 7)) << 4);
  const int n16 = lane & 15;
  const uint32_t vbase = lds0 + VRING + ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;
  const uint32_t ldsk = lds0 + wave * 2048, ldsv = lds0 + VRING + (wave >> 1) * 4096 + (wave & 1) * 2048;
  const uint32_t qrel = QIMG + wave * 8192, ldsq = lds0 + qrel, tbl = lds0 + TABLE, wave64 = wave * 64, ldsst = lds0 + STAGE + wave * 4096;

#define MFA_P6_RUN(name, f16, fold, o16, l16, causal, split) if constexpr (STREAM == S_##name) MFA_P6_RUN_STREAM(MFA_P6_STREAM_##name);
  MFA_P6_STREAM_LIST(MFA_P6_RUN)
#undef MFA_P6_RUN
}

} // namespace mfa
