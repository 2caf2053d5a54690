// attn_dq16_p4_tr.h -- backwardQuery on the hand-placed stream with K and V
// stored TRANSPOSED ([D][keys]), read where they lie; Q, dO, O, dQ either way (run-time flags, outside the statement).
//
// The streams (tools/dq4gen.py Cfg.tr, MFA_DQ4_TR_STREAM_LIST) are verified on the lane-exact model
// (tests/test_dq4_stream.py::test_transposed_key_value_streams); this wrapper restates what tools/dq4sim.py hands them: the
// K^T / V^T images keep the source orientation ([2 blocks of 32 keys][128 elements][64 bytes], chunks ^ (element >> 2) & 3), K and V
// row fragments come from transposing reads -- which return the contraction index in the order of an accumulator block's registers
// (4 hi + {0..3, 8..11}), so Q' and dO go into the accumulator registers in that order -- and the K^T fragments of the dQ update
// are two 8-byte reads per fragment (addresses ta0..ta3).  Whole tiles only (C % 64 == 0), 16-byte aligned rows of K^T / V^T,
// no per-batch lengths, no block mask, dO in the type of Q / K / V: the launcher (attn_bwd16_p4_tr.hip) checks.  Product
// library since round 4 (GPU evidence: profiles/r04_candidate/).
#pragma once
#include "attn_dq16_p4.h"

namespace mfa {
namespace dq4tr {

#define MFA_DQ4TR_ENUM(name, exact) S_##name,
enum : int { MFA_DQ4_TR_STREAM_LIST(MFA_DQ4TR_ENUM) S_COUNT };
#undef MFA_DQ4TR_ENUM
constexpr bool stream_exact(int s) {
#define MFA_DQ4TR_EXACT(name, exact) if (s == S_##name) return exact != 0;
  MFA_DQ4_TR_STREAM_LIST(MFA_DQ4TR_EXACT)
#undef MFA_DQ4TR_EXACT
  return false;
}

}  // namespace dq4tr

#define MFA_DQ4TR_TRAVERSE(STREAM)                                                                                       \
  asm volatile(STREAM                                                                                                    \
               : [koff0] "+v"(koff[0]), [koff1] "+v"(koff[1]), [koff2] "+v"(koff[2]), [koff3] "+v"(koff[3]),              \
                 [voff0] "+v"(voff[0]), [voff1] "+v"(voff[1]), [voff2] "+v"(voff[2]), [voff3] "+v"(voff[3]),              \
                 [ka0] "+v"(ka0), [ka1] "+v"(ka1), [ta0] "+v"(ta[0]), [ta1] "+v"(ta[1]), [ta2] "+v"(ta[2]), [ta3] "+v"(ta[3]), \
                 [j] "=&s"(tj), [stg] "=&s"(tstg), [delta] "=&s"(tdelta), [wr] "=&s"(twr), [t0] "=&s"(tt0),               \
                 [t1] "=&s"(tt1), [pa] "=&s"(tpa), [pb] "=&s"(tpb), [pc] "=&s"(tpc), [pd] "=&s"(tpd),                     \
                 [plast] "=&s"(tplast), [ptime] "=&s"(tptime)                                                            \
               : [negl0] "v"(negl0), [negl1] "v"(negl1), [negd0] "v"(negd0), [negd1] "v"(negd1), [lim0] "v"(lim0),        \
                 [lim1] "v"(lim1), [kres] "s"(kdesc), [vres] "s"(vdesc), [nt] "s"(nt), [wnt] "s"(wnt), [kinc] "s"(kinc),   \
                 [vinc] "s"(vinc), [wr0] "s"(wr0), [ringend] "s"(ringend), [maskfrom] "s"(maskfrom),                      \
                 [scale2x2] "s"(scale2x2)                                                                                \
               : "memory", "vcc", "scc", MFA_ALL_AGPRS, MFA_DQ4_OWNED_VGPRS)

template <typename T, int STREAM, bool CAUSAL, typename TG = T>
__global__ __launch_bounds__(256) void attn_dq16_p4_tr(const KernelArgs a, const Fwd16Grid grid) {
  using namespace dq4;
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 128, BC = 64, NKS = 8, NDB = 4, WROWS = 64, GROWS = 256, PW = 4;
  constexpr bool EXACT = dq4tr::stream_exact(STREAM);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &rblk, &head, &batch);
  if constexpr (CAUSAL) rblk = grid.rowBlocks - 1 - rblk;
  const int R = a.R, C = a.C, Dr = a.D;   // (no per-batch lengths; C % 64 == 0)
  if ((int64_t)rblk * GROWS >= R) return;
  const int64_t r0 = (int64_t)rblk * GROWS + wave * WROWS;
  const bool qT = a.op[SLOT_Q].transposed != 0, gT = a.op[SLOT_dO].transposed != 0, oT = a.op[SLOT_O].transposed != 0,
             dqT = a.op[SLOT_dQ].transposed != 0;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2,
                 ldv2 = (uint32_t)a.op[SLOT_V].ld * 2, ldg2 = (uint32_t)a.op[SLOT_dO].ld * 2;
  const bool o32 = a.op[SLOT_O].precision == PREC_FP32;
  const uint32_t oesz = o32 ? 4u : 2u, ldo = (uint32_t)a.op[SLOT_O].ld;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)(qT ? Dr : R) * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_dO], head, batch), 0, (uint32_t)(gT ? Dr : R) * ldg2, 0x00020000);
  const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_O], head, batch), 0, (uint32_t)(oT ? Dr : R) * ldo * oesz, 0x00020000);
  const char *kptr = operand_base(a.op[SLOT_K], head, batch), *vptr = operand_base(a.op[SLOT_V], head, batch);
  const uint64_t kaddr = (uint64_t)(uintptr_t)kptr, vaddr = (uint64_t)(uintptr_t)vptr;
  const u32x4 kdesc = {(uint32_t)kaddr, (uint32_t)(kaddr >> 32) & 0xFFFFu, (uint32_t)Dr * ldk2, 0x00020000u};
  const u32x4 vdesc = {(uint32_t)vaddr, (uint32_t)(vaddr >> 32) & 0xFFFFu, (uint32_t)Dr * ldv2, 0x00020000u};

  // eight 16-bit elements of row `row` of a [rows][D] (or transposed [D][rows]) operand, at the elements `elem(i)`
  auto load16x8 = [&](const __amdgpu_buffer_rsrc_t &res, bool transposed, uint32_t ld2, int64_t row, int s) {
    auto elem = [&](int i) { return 16 * s + 4 * hi + (i & 3) + 8 * (i >> 2); };
    const bool rowok = row < R;
    if (transposed) {
      uint16_t e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        e[i] = __builtin_amdgcn_raw_buffer_load_b16(res, (rowok && elem(i) < Dr) ? (uint32_t)elem(i) * ld2 + (uint32_t)row * 2 : OOB, 0, 0);
      return u32x4{e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16)};
    }
    const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(res, (rowok && elem(0) < Dr) ? (uint32_t)row * ld2 + elem(0) * 2 : OOB, 0, 0);
    const u32x2 up = __builtin_amdgcn_raw_buffer_load_b64(res, (rowok && elem(4) < Dr) ? (uint32_t)row * ld2 + elem(4) * 2 : OOB, 0, 0);
    return u32x4{lo[0], lo[1], up[0], up[1]};
  };
  // the same eight elements of O as floats
  auto load_o8 = [&](int64_t row, int s, float (&o)[8]) {
    auto elem = [&](int i) { return 16 * s + 4 * hi + (i & 3) + 8 * (i >> 2); };
    const bool rowok = row < R;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = rowok && elem(i) < Dr;
      const uint32_t off = oT ? ((uint32_t)elem(i) * ldo + (uint32_t)row) * oesz : ((uint32_t)row * ldo + (uint32_t)elem(i)) * oesz;
      if (o32) o[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ores, ok ? off : OOB, 0, 0));
      else {
        const uint16_t h = __builtin_amdgcn_raw_buffer_load_b16(ores, ok ? off : OOB, 0, 0);
        if constexpr (__is_same(T, __bf16)) o[i] = __builtin_bit_cast(float, (uint32_t)h << 16);
        else o[i] = (float)__builtin_bit_cast(_Float16, h);
      }
    }
  };

  // ---- Q' and dO fragments -> a[128:255] in the element order of the transposing reads; computeD
  float dterm[2] = {0.f, 0.f};
  static_for<2>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const int64_t row = r0 + b * 32 + q;
    static_for<NKS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const u32x4 qx = load16x8(qres, qT, ldq2, row, s);
      const u32x4 gx = load16x8(gres, gT, ldg2, row, s);
      const v8 g8 = convert_chunk<T, TG>(gx);   // (TG = __bf16 next to _Float16 Q / K / V: the reference's own mix, converted when loaded)
      float o[8];
      load_o8(row, s, o);
#pragma unroll
      for (int i = 0; i < 8; ++i) dterm[b] += (float)g8[i] * o[i];
      if constexpr (EXACT) p4::acc_write4<Q_BASE + 4 * (b * 8 + s)>(qx);
      else p4::acc_write4<Q_BASE + 4 * (b * 8 + s)>(p4::scale16x8<T>(qx, a.scale2));
      p4::acc_write4<G_BASE + 4 * (b * 8 + s)>(__builtin_bit_cast(u32x4, g8));
    });
  });
  float negl[2], negd[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int64_t row = r0 + b * 32 + q;
    const float dsum = half_swap_add(dterm[b]);
    float Lrow = 0.f;
    if (row < R) Lrow = load_elem(operand_base(a.op[SLOT_L], head, batch), row, a.op[SLOT_L].precision);
    if (hi == 0 && row < R) store_elem(operand_base(a.op[SLOT_D], head, batch), row, a.op[SLOT_D].precision, dsum * a.scale);
    negl[b] = EXACT ? -Lrow / a.scale2 : -Lrow;
    negd[b] = -dsum;
  }
  const float negl0 = negl[0], negl1 = negl[1], negd0 = negd[0], negd1 = negd[1];

  // ---- traversal range (whole tiles)
  const int tiles_total = C / BC;
  const int coff = causal_offset(R, C);
  int nt = tiles_total;
  if constexpr (CAUSAL) {
    const int64_t last_row = min((int64_t)R, ((int64_t)rblk + 1) * GROWS) - 1;
    nt = (int)min((int64_t)tiles_total, (last_row + coff) / BC + 1);
  }
  int wnt = nt;
  if constexpr (CAUSAL) {
    const int64_t wlast = min((int64_t)R, r0 + WROWS) - 1;
    wnt = wlast >= r0 ? (int)max((int64_t)1, min((int64_t)nt, (wlast + coff) / BC + 1)) : 1;
    wnt = __builtin_amdgcn_readfirstlane(wnt);
  }
  const int minlim = CAUSAL ? (int)min((int64_t)C - 1, r0 + coff) : C - 1;
  const int maskfrom = CAUSAL ? (minlim + 1) / BC : nt;
  int lim0 = C - 1, lim1 = C - 1;
  if constexpr (CAUSAL) {
    lim0 = (int)min((int64_t)C - 1, r0 + q + coff);
    lim1 = (int)min((int64_t)C - 1, r0 + 32 + q + coff);
  }
  lim0 -= 4 * hi;
  lim1 -= 4 * hi;

  // ---- LDS-DMA staging: piece i of wave w fills 16-byte positions (4 w + i) * 64 + lane of a tile
  // ([2 blocks of 32 keys][128 elements][4 chunks of 8 keys], chunk ^ (element >> 2) & 3)
  uint32_t koff[4], voff[4];
  const uint32_t kinc = BC * 2, vinc = BC * 2;   // a tile further = 64 keys along every row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = (wave * PW + i) * 64 + lane;
    const int kb = p >> 9, d = (p >> 2) & 127, chunk = (p & 3) ^ ((d >> 2) & 3);
    koff[i] = (d < Dr) ? (uint32_t)d * ldk2 + (kb * 32 + chunk * 8) * 2 : OOB;
    voff[i] = (d < Dr) ? (uint32_t)d * ldv2 + (kb * 32 + chunk * 8) * 2 : OOB;
  }
  const uint32_t lds0 = lds_addr(smem);
  const int n16 = lane & 15;
  const int trow = (n16 >> 2) + 4 * hi, tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1), thalf = (n16 & 3) & 1;
  // row fragments (K, V): transposing reads of rows + 0 / + 8 of a 16-element step; K^T fragments: the lane's element row, chunk c at 8 hi
  uint32_t ka0 = lds0 + trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8;
  uint32_t ka1 = lds0 + (trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8;
  uint32_t ta[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) ta[c] = lds0 + q * 64 + ((c ^ ((q >> 2) & 3)) * 16) + 8 * hi;
  const uint32_t wr0 = lds0 + wave * (PW * 1024), ringend = lds0 + RING_BYTES;
  const uint64_t scale2x2 = (uint64_t)__builtin_bit_cast(uint32_t, a.scale2) * 0x100000001ull;

  {
    uint32_t tj, tstg, tdelta, twr, tt0, tt1, tplast, tpa, tpb, tpc, tpd;
    uint64_t tptime;
#define MFA_DQ4TR_RUN(name, exact) if constexpr (STREAM == dq4tr::S_##name) MFA_DQ4TR_TRAVERSE(MFA_DQ4_STREAM_##name);
    MFA_DQ4_TR_STREAM_LIST(MFA_DQ4TR_RUN)
#undef MFA_DQ4TR_RUN
  }

  // ================= epilogue: dQ = scale * dS' K (+Source.swift:236-242) =================
  asm volatile("s_nop 15\n\ts_nop 7" ::: MFA_ALL_AGPRS);
  __syncthreads();
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + wave * (WROWS * OLD);
  const int dqprec = a.op[SLOT_dQ].precision;
  const uint32_t dqesz = dqprec == PREC_FP32 ? 4u : 2u, lddq = (uint32_t)a.op[SLOT_dQ].ld;
  const __amdgpu_buffer_rsrc_t dqres =
      __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_dQ], head, batch), 0, (uint32_t)Dr * lddq * dqesz, 0x00020000);
  static_for<2>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const int64_t row = r0 + b * 32 + q;
    float *orow = Os + (b * 32 + q) * OLD;
    static_for<NDB>([&](auto dc) {
      constexpr int db = decltype(dc)::value;
      float x[16];
      p4::acc_read16<16 * (b * 4 + db)>(x);
      if (dqT) {   // dQ^T ([D][R]): register r of block db is element 32 db + crow(r, hi) of the lane's row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = 32 * db + crow(r, hi);
          const uint32_t off = (d < Dr && row < R) ? ((uint32_t)d * lddq + (uint32_t)row) * dqesz : OOB;
          const float val = x[r] * a.scale;
          if (dqprec == PREC_FP32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), dqres, off, 0, 0);
          else __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pack16<T>(val, 0.f), dqres, off, 0, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) = make_float4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
      }
    });
  });
  if (!dqT) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
      store_block_rows<T, D>(Os + b * 32 * OLD, operand_base(a.op[SLOT_dQ], head, batch), a.op[SLOT_dQ].precision,
                             (uint32_t)a.op[SLOT_dQ].ld, r0 + 32 * b, R, Dr, lane, a.scale);
  }
}

} // namespace mfa
