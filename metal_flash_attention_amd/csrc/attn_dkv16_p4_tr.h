// attn_dkv16_p4_tr.h -- backwardKeyValue on the hand-placed stream with Q and
// dO stored TRANSPOSED ([D][rows]), read where they lie; K, V, dK, dV either way (run-time flags, outside the statement).
//
// The streams (tools/dkv4gen.py Cfg.tr, MFA_DKV4_TR_STREAM_LIST) are verified on the lane-exact model
// (tests/test_dkv4_stream.py::test_transposed_query_gradient_streams); this wrapper restates what tools/dkv4sim.py hands them: a
// 32-row step's tile in the source orientation is [128 elements][4 chunks of 8 rows ^ (element >> 2) & 3]; Q / dO row fragments
// come from transposing reads -- the contraction index in the order of an accumulator block's registers (4 hi + {0..3, 8..11}), so
// the K' and V fragments are parked in that order -- and the dO^T / Q^T fragments are two 8-byte reads each (addresses ta0..ta3).
// Whole steps only (R % 32 == 0), 16-byte aligned rows of Q^T / dO^T, no per-batch lengths, no block mask, dO in the type of
// Q / K / V, L and D in the stream's storage types: the launcher (attn_bwd16_p4_tr.hip) checks.  Product library since round 4
// (GPU evidence: profiles/r04_candidate/).
#pragma once
#include "attn_dkv16_p4.h"

namespace mfa {
namespace dkv4tr {

#define MFA_DKV4TR_ENUM(name, exact, mix) S_##name,
enum : int { MFA_DKV4_TR_STREAM_LIST(MFA_DKV4TR_ENUM) S_COUNT };
#undef MFA_DKV4TR_ENUM
constexpr bool stream_exact(int s) {
#define MFA_DKV4TR_EXACT(name, exact, mix) if (s == S_##name) return exact != 0;
  MFA_DKV4_TR_STREAM_LIST(MFA_DKV4TR_EXACT)
#undef MFA_DKV4TR_EXACT
  return false;
}
// dO^T is BF16 next to FP16 Q^T, K, V (the reference's own low-precision mix; see stream_mix in attn_dkv16_p4.h)
constexpr bool stream_mix(int s) {
#define MFA_DKV4TR_MIX(name, exact, mix) if (s == S_##name) return mix != 0;
  MFA_DKV4_TR_STREAM_LIST(MFA_DKV4TR_MIX)
#undef MFA_DKV4TR_MIX
  return false;
}

}  // namespace dkv4tr

#define MFA_DKV4TR_TRAVERSE(STREAM)                                                                                      \
  asm volatile(STREAM                                                                                                    \
               : [qoff0] "+v"(qoff[0]), [qoff1] "+v"(qoff[1]), [goff0] "+v"(goff[0]), [goff1] "+v"(goff[1]),              \
                 [ldoff] "+v"(ldoff), [ra0] "+v"(ra0), [ra1] "+v"(ra1), [ta0] "+v"(ta[0]), [ta1] "+v"(ta[1]),             \
                 [ta2] "+v"(ta[2]), [ta3] "+v"(ta[3]),                                                                  \
                 [j] "=&s"(tj), [stg] "=&s"(tstg), [delta] "=&s"(tdelta), [wr] "=&s"(twr), [t0] "=&s"(tt0),               \
                 [t1] "=&s"(tt1), [pa] "=&s"(tpa), [pb] "=&s"(tpb), [pc] "=&s"(tpc), [pd] "=&s"(tpd),                     \
                 [plast] "=&s"(tplast), [ptime] "=&s"(tptime)                                                            \
               : [onesw] "v"(onesw), [tk] "v"(tk), [kvback] "v"(kvback), [qres] "s"(qdesc), [gres] "s"(gdesc),            \
                 [lres] "s"(ldesc), [dres] "s"(ddesc), [nsteps] "s"(nsteps), [rscale] "s"(rscale), [qinc] "s"(qinc),      \
                 [ginc] "s"(ginc), [ldinc] "s"(ldinc), [wr0] "s"(wr0), [ringend] "s"(ringend), [maskuntil] "s"(maskuntil), \
                 [rscale2] "s"(rscale2), [scale2x2] "s"(scale2x2)                                                       \
               : "memory", "vcc", "scc", MFA_ALL_AGPRS, MFA_DKV4_OWNED_VGPRS)

template <typename T, int STREAM, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_dkv16_p4_tr(const KernelArgs a, const Fwd16Grid grid) {
  using namespace dkv4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 128, NKS = 8, NDB = 4, WKEYS = 64, GKEYS = 256, BR = 32, PW = 2;
  constexpr bool EXACT = dkv4tr::stream_exact(STREAM), MIX = dkv4tr::stream_mix(STREAM);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, kc = lane & 31, hi = lane >> 5;
  uint32_t cblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &cblk, &head, &batch);
  const int R = a.R, C = a.C, Dr = a.D;   // (no per-batch lengths; R % 32 == 0)
  if ((int64_t)cblk * GKEYS >= C) return;
  const int64_t c0 = (int64_t)cblk * GKEYS + wave * WKEYS;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldg2 = (uint32_t)a.op[SLOT_dO].ld * 2;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const bool kT = a.op[SLOT_K].transposed != 0, vT = a.op[SLOT_V].transposed != 0, dkT = a.op[SLOT_dK].transposed != 0,
             dvT = a.op[SLOT_dV].transposed != 0;

  // ---- K' and V fragments (B operands: lane = key) in the element order of the transposing reads of Q^T / dO^T
  // (4 hi + {0..3, 8..11} of a 16-element step), parked in LDS for the statement
  {
    const uint32_t ldk2 = (uint32_t)a.op[SLOT_K].ld * 2, ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;
    const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_K], head, batch), 0, (uint32_t)(kT ? Dr : C) * ldk2, 0x00020000);
    const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_V], head, batch), 0, (uint32_t)(vT ? Dr : C) * ldv2, 0x00020000);
    auto load16x8 = [&](const __amdgpu_buffer_rsrc_t &res, bool transposed, uint32_t ld2, int64_t col, int s) {
      auto elem = [&](int i) { return 16 * s + 4 * hi + (i & 3) + 8 * (i >> 2); };
      const bool colok = col < C;
      if (transposed) {
        uint16_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          e[i] = __builtin_amdgcn_raw_buffer_load_b16(res, (colok && elem(i) < Dr) ? (uint32_t)elem(i) * ld2 + (uint32_t)col * 2 : OOB, 0, 0);
        return u32x4{e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16)};
      }
      const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(res, (colok && elem(0) < Dr) ? (uint32_t)col * ld2 + elem(0) * 2 : OOB, 0, 0);
      const u32x2 up = __builtin_amdgcn_raw_buffer_load_b64(res, (colok && elem(4) < Dr) ? (uint32_t)col * ld2 + elem(4) * 2 : OOB, 0, 0);
      return u32x4{lo[0], lo[1], up[0], up[1]};
    };
    char *back = smem + wave * 32768 + lane * 16;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int64_t col = c0 + 32 * kb + kc;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const u32x4 kx = load16x8(kres, kT, ldk2, col, s);
        const u32x4 vx = load16x8(vres, vT, ldv2, col, s);
        if constexpr (EXACT) *reinterpret_cast<u32x4 *>(back + (kb * 8 + s) * 1024) = kx;
        else *reinterpret_cast<u32x4 *>(back + (kb * 8 + s) * 1024) = p4::scale16x8<T>(kx, a.scale2);
        if constexpr (MIX) *reinterpret_cast<u32x4 *>(back + (16 + kb * 8 + s) * 1024) = __builtin_bit_cast(u32x4, convert_chunk<__bf16, T>(vx));
        else *reinterpret_cast<u32x4 *>(back + (16 + kb * 8 + s) * 1024) = vx;
      }
    }
  }

  // ---- traversal range: whole 32-row steps from the first row block that sees the workgroup's first key (CAUSAL) to the end
  const int coff = causal_offset(R, C);
  const int row_first = CAUSAL ? (int)(max((int64_t)0, (int64_t)cblk * GKEYS - coff) / BR) * BR : 0;
  const int nsteps = max(1, (R - row_first) / BR);
  int maskuntil = 0;
  if constexpr (CAUSAL) {
    const int64_t span = c0 + WKEYS - 1 - coff - row_first;
    maskuntil = span > 0 ? (int)((span + BR - 1) / BR) : 0;
    maskuntil = __builtin_amdgcn_readfirstlane(maskuntil);
  }

  // ---- LDS-DMA staging: piece i of wave w fills 16-byte positions (2 w + i) * 64 + lane of a tile
  // ([128 elements][4 chunks of 8 rows], chunk ^ (element >> 2) & 3); a step further = 32 rows along every row of Q^T / dO^T
  uint32_t qoff[2], goff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (PW * wave + i) * 64 + lane;
    const int d = p >> 2, chunk = (p & 3) ^ ((d >> 2) & 3);
    qoff[i] = (d < Dr) ? (uint32_t)d * ldq2 + (uint32_t)(row_first + chunk * 8) * 2 : OOB;
    goff[i] = (d < Dr) ? (uint32_t)d * ldg2 + (uint32_t)(row_first + chunk * 8) * 2 : OOB;
  }
  const uint32_t qinc = BR * 2, ginc = BR * 2;
  const char *qptr = operand_base(a.op[SLOT_Q], head, batch), *gptr = operand_base(a.op[SLOT_dO], head, batch);
  const char *lptr = operand_base(a.op[SLOT_L], head, batch), *dptr = operand_base(a.op[SLOT_D], head, batch);
  const uint32_t lesz = a.op[SLOT_L].precision == PREC_FP32 ? 4u : 2u;
  auto desc = [](const char *p, uint32_t bytes) {
    const uint64_t x = (uint64_t)(uintptr_t)p;
    return u32x4{(uint32_t)x, (uint32_t)(x >> 32) & 0xFFFFu, bytes, 0x00020000u};
  };
  const u32x4 qdesc = desc(qptr, (uint32_t)Dr * ldq2), gdesc = desc(gptr, (uint32_t)Dr * ldg2);
  const u32x4 ldesc = desc(lptr, (uint32_t)R * lesz), ddesc = desc(dptr, (uint32_t)R * lesz);
  uint32_t ldoff = (uint32_t)(row_first + kc) * lesz;
  const uint32_t ldinc = BR * lesz;

  // per-lane LDS read addresses: row fragments (Q, dO) = transposing reads of rows + 0 / + 8 of a 16-element step; dO^T / Q^T
  // fragments = the lane's element row, chunk c at 8 hi
  const uint32_t lds0 = lds_addr(smem);
  const int n16 = lane & 15;
  const int trow = (n16 >> 2) + 4 * hi, tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1), thalf = (n16 & 3) & 1;
  uint32_t ra0 = lds0 + trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8;
  uint32_t ra1 = lds0 + (trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8;
  uint32_t ta[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) ta[c] = lds0 + kc * 64 + ((c ^ ((kc >> 2) & 3)) * 16) + 8 * hi;
  const uint32_t kvback = lds0 + wave * 32768 + lane * 16;
  const uint32_t wr0 = lds0 + wave * (PW * 1024), ringend = lds0 + RING_BYTES;
  const uint32_t onesw = hi ? 0u : (MIX ? 0xC000C000u : (__is_same(T, __bf16) ? 0xBF80BF80u : 0xBC00BC00u));   // (mix: -2.0, one pattern for both types)
  const int tk = (int)(c0 + kc - coff - 4 * hi - row_first);
  constexpr float half = MIX ? 0.5f : 1.0f;
  const float rscale = half / a.scale, rscale2 = EXACT ? half / a.scale2 : half;
  const uint64_t scale2x2 = (uint64_t)__builtin_bit_cast(uint32_t, a.scale2) * 0x100000001ull;

  {
    uint32_t tj, tstg, tdelta, twr, tt0, tt1, tplast, tpa, tpb, tpc, tpd;
    uint64_t tptime;
#define MFA_DKV4TR_RUN(name, exact, mix) if constexpr (STREAM == dkv4tr::S_##name) MFA_DKV4TR_TRAVERSE(MFA_DKV4_STREAM_##name);
    MFA_DKV4_TR_STREAM_LIST(MFA_DKV4TR_RUN)
#undef MFA_DKV4TR_RUN
  }

  // ================= epilogue: dV = P^T dO, dK = scale * dS'^T Q (+Source.swift:286-293) =================
  asm volatile("s_nop 15\n\ts_nop 7" ::: MFA_ALL_AGPRS);
  __syncthreads();
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + wave * (WKEYS * OLD);
  static_for<2>([&](auto oc) {
    constexpr int which = decltype(oc)::value;   // 0: dV, 1: dK
    constexpr int slot = which ? SLOT_dK : SLOT_dV;
    const bool outT = which ? dkT : dvT;
    const int prec = a.op[slot].precision;
    const uint32_t esz = prec == PREC_FP32 ? 4u : 2u, ldx = (uint32_t)a.op[slot].ld;
    const float sc = which ? a.scale : 1.0f;
    const __amdgpu_buffer_rsrc_t tres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[slot], head, batch), 0, (uint32_t)Dr * ldx * esz, 0x00020000);
    static_for<2>([&](auto kbc) {
      constexpr int kb = decltype(kbc)::value;
      const int64_t col = c0 + 32 * kb + kc;
      float *orow = Os + (kb * 32 + kc) * OLD;
      static_for<NDB>([&](auto dc) {
        constexpr int db = decltype(dc)::value;
        float x[16];
        p4::acc_read16<which * 128 + 16 * (2 * db + kb)>(x);
        if (outT) {   // [D][C]: register r of block db is element 32 db + crow(r, hi) of the lane's key
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = 32 * db + crow(r, hi);
            const uint32_t off = (d < Dr && col < C) ? ((uint32_t)d * ldx + (uint32_t)col) * esz : OOB;
            const float val = x[r] * sc;
            if (prec == PREC_FP32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), tres, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pack16<T>(val, 0.f), tres, off, 0, 0);
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) = make_float4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
        }
      });
    });
    if (!outT) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        store_block_rows<T, D>(Os + kb * 32 * OLD, operand_base(a.op[slot], head, batch), prec, ldx, c0 + 32 * kb, C, Dr, lane, sc);
    }
  });
}

} // namespace mfa
