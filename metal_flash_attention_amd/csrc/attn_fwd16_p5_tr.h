// attn_fwd16_p5_tr.h -- DEVELOPER BUILD ONLY: the hand-placed forward kernel of the head-dimension buckets 160 / 192 / 256
// (attn_fwd16_p5.h: four waves x 64 rows, 32-key steps) for K and V stored TRANSPOSED ([D][keys], transposeState of
// AttentionKernelDescriptor.swift:28-42, read where they lie like AttentionKernel.swift:189-204); Q and O either way.
//
// Same statement family with the two read recipes exchanged (tools/f256gen.py, Cfg.tr; lane-exact model: tools/f256sim.py,
// tests/test_f256_stream.py).  Both LDS images keep the orientation of the source, [D elements][4 chunks of 8 keys], the chunk
// index ^ (element >> 2) & 3: K^T is read with ds_read_b64_tr_b16 (the contraction index arrives as 4 hi + {0..3, 8..11} of a
// 16-element step, so the Q' fragments are parked in that order), V^T 8 bytes at a time (four addresses, one per chunk).  A step
// further is 64 bytes along every row.  Whole steps only: launches with column % 32 != 0, rows of K^T / V^T that are not 16-byte
// aligned, per-batch lengths or a block mask are not taken (the launcher returns false and the 8 x 32 kernel's transposed code
// object runs, attn_fwd16_v3.h TR).  The product library launches that code object for every such problem until this kernel has
// been timed against it.
// One transposed operand (K^T alone, V^T alone): the halves of a step's fragment list are independent (tools/f256gen.py, Cfg.tr as a
// bit mask) -- the transposed operand takes the exchanged recipe, the other keeps attn_fwd16_p5.h's; those 24 streams are generated
// at build time (attn_fwd16_p5_tr1_stream.inc, `make DEV=1`), model-verified like the others, NOT yet run on a GPU.
#pragma once
#include "attn_fwd16_p5.h"
#include "attn_fwd16_p5_tr1_stream.inc"

// K^T + V^T (tracked streams) first: their enumerators -- and so the names of their code objects -- do not move
#define MFA_P5TR_ALL_STREAMS(X) MFA_P5_TR_STREAM_LIST(X) MFA_P5_TR1_STREAM_LIST(X)

namespace mfa {
namespace p5tr {

#define MFA_P5TR_ENUM(name, fold, d, pattern) S_##name,
enum : int { MFA_P5TR_ALL_STREAMS(MFA_P5TR_ENUM) S_COUNT };
#undef MFA_P5TR_ENUM
constexpr bool stream_folds(int s) {
#define MFA_P5TR_FOLDS(name, fold, d, pattern) if (s == S_##name) return fold != 0;
  MFA_P5TR_ALL_STREAMS(MFA_P5TR_FOLDS)
#undef MFA_P5TR_FOLDS
  return false;
}
constexpr int stream_bucket(int s) {
#define MFA_P5TR_BUCKET(name, fold, d, pattern) if (s == S_##name) return d;
  MFA_P5TR_ALL_STREAMS(MFA_P5TR_BUCKET)
#undef MFA_P5TR_BUCKET
  return 256;
}
// bit 0 = K, bit 1 = V stored transposed
constexpr int stream_pattern(int s) {
#define MFA_P5TR_PATTERN(name, fold, d, pattern) if (s == S_##name) return pattern;
  MFA_P5TR_ALL_STREAMS(MFA_P5TR_PATTERN)
#undef MFA_P5TR_PATTERN
  return 3;
}

}  // namespace p5tr

// the statement of attn_fwd16_p5 with four V^T addresses
#define MFA_P5TR_TRAVERSE(STREAM)                                                                                        \
  asm volatile(STREAM                                                                                                    \
               : [m0] "+v"(m0), [m1] "+v"(m1), [l0] "+v"(l0), [l1] "+v"(l1), [koff0] "+v"(koff[0]), [koff1] "+v"(koff[1]), \
                 [koff2] "+v"(koff[2]), [koff3] "+v"(koff[3]), [voff0] "+v"(voff[0]), [voff1] "+v"(voff[1]),               \
                 [voff2] "+v"(voff[2]), [voff3] "+v"(voff[3]), [ka0] "+v"(ka0), [ka1] "+v"(ka1), [ta0] "+v"(ta[0]),       \
                 [ta1] "+v"(ta[1]), [ta2] "+v"(ta[2]), [ta3] "+v"(ta[3]), [j] "=&s"(tj), [stg] "=&s"(tstg),               \
                 [delta] "=&s"(tdelta), [deltav] "=&s"(tdeltav), [wr] "=&s"(twr), [pend] "=&s"(tpend), [t0] "=&s"(tt0),   \
                 [t1] "=&s"(tt1), [pa] "=&s"(tpa), [pb] "=&s"(tpb), [plast] "=&s"(tplast), [sv] "=&s"(tsv),               \
                 [ptime] "=&s"(tptime)                                                                                   \
               : [lim0] "v"(lim0), [lim1] "v"(lim1), [onesw] "v"(onesw), [qback] "v"(qback), [kres] "s"(kdesc),           \
                 [vres] "s"(vdesc), [nt] "s"(nt), [wnt] "s"(wnt), [scale2] "s"(a.scale2), [kinc] "s"(kinc), [vinc] "s"(vinc), \
                 [wr0] "s"(wr0), [ringend] "s"(ringend), [maskfrom] "s"(maskfrom)                                         \
               : "memory", "vcc", "scc", MFA_ALL_AGPRS, MFA_P5_OWNED_VGPRS)

// T: __bf16 or _Float16 (must match the stream); CAUSAL only selects the block order and the bounds (as attn_fwd16_p5)
template <typename T, int STREAM, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd16_p5_tr(const KernelArgs a, const Fwd16Grid grid) {
  using namespace p5;
  using p5tr::stream_folds;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = p5tr::stream_bucket(STREAM), BC = 32, NKS = D / 16, NDB = D / 32, WROWS = 64, GROWS = 256;
  constexpr int PW = (D + 63) / 64;   // LDS-DMA pieces per wave and operand tile (D elements x 32 keys x 2 bytes, 1 KiB each, four waves)
  constexpr bool KT = (p5tr::stream_pattern(STREAM) & 1) != 0, VT = (p5tr::stream_pattern(STREAM) & 2) != 0;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane0 = tid & 63;
  uint32_t rblk0, head, batch;
  Fwd16Grid dgrid = grid;
  if constexpr (CAUSAL) dgrid.rowBlocks = (grid.rowBlocks + 1) / 2;   // a workgroup takes the pair of row blocks (last - i, i)
  fwd16_decode_block(dgrid, blockIdx.x, &rblk0, &head, &batch);
  const int Dr = a.D;
  const bool qT = a.op[SLOT_Q].transposed != 0, oT = a.op[SLOT_O].transposed != 0;
#pragma unroll 1
  for (int pass = 0; pass < (CAUSAL ? 2 : 1); ++pass) {
  int lane = lane0;   // (everything a lane derives from its number is recomputed per pass, see attn_fwd16_p5.h)
  if constexpr (CAUSAL) asm volatile("" : "+v"(lane));
  const int q = lane & 31, hi = lane >> 5;
  uint32_t rblk = rblk0;
  if constexpr (CAUSAL) {
    rblk = pass == 0 ? grid.rowBlocks - 1 - rblk0 : rblk0;
    if (pass == 1 && rblk0 == grid.rowBlocks - 1 - rblk0) break;
    if (pass == 1) __syncthreads();
  }
  const int R = a.R, C = a.C;   // (no per-batch lengths here; C % 32 == 0)
  if ((int64_t)rblk * GROWS >= R) continue;
  const int64_t r0 = (int64_t)rblk * GROWS + wave * WROWS;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2,
                 ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;
  const char *kptr = operand_base(a.op[SLOT_K], head, batch), *vptr = operand_base(a.op[SLOT_V], head, batch);
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)(qT ? Dr : R) * ldq2, 0x00020000);
  const uint64_t kaddr = (uint64_t)(uintptr_t)kptr, vaddr = (uint64_t)(uintptr_t)vptr;
  auto uni = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
  const u32x4 kdesc = {uni((uint32_t)kaddr), uni((uint32_t)(kaddr >> 32) & 0xFFFFu), uni((uint32_t)(KT ? Dr : C) * ldk2), 0x00020000u};
  const u32x4 vdesc = {uni((uint32_t)vaddr), uni((uint32_t)(vaddr >> 32) & 0xFFFFu), uni((uint32_t)(VT ? Dr : C) * ldv2), 0x00020000u};
  constexpr uint32_t OOB = 0xFFFFFF00u;

  // ---- Q' fragments (B operand of S^T = K Q'^T: lane = row), in the order the K fragments hold the contraction index: with K^T
  // elements 16 s + 4 hi + {0..3, 8..11} (what its transposing reads return), with row-major K 16 s + 8 hi + {0..7}; parked in LDS
  {
    char *back = smem + wave * 32768 + lane * 16;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t row = r0 + b * 32 + q;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const int d0 = KT ? 16 * s + 4 * hi : 16 * s + 8 * hi;
        u32x4 x;
        if (qT) {   // Q^T [D][R]: one element per load (consecutive lanes = consecutive rows)
          uint32_t e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int d = KT ? d0 + (i & 3) + 8 * (i >> 2) : d0 + i;
            const uint32_t off = (d < Dr && row < R) ? (uint32_t)d * ldq2 + (uint32_t)row * 2 : OOB;
            e[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(qres, off, 0, 0);
          }
          x = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
        } else if constexpr (KT) {
          const uint32_t rowoff = (uint32_t)row * ldq2;
          const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(qres, (d0 < Dr && row < R) ? rowoff + d0 * 2 : OOB, 0, 0);
          const u32x2 up = __builtin_amdgcn_raw_buffer_load_b64(qres, (d0 + 8 < Dr && row < R) ? rowoff + (d0 + 8) * 2 : OOB, 0, 0);
          x = u32x4{lo[0], lo[1], up[0], up[1]};
        } else {
          x = __builtin_amdgcn_raw_buffer_load_b128(qres, (d0 < Dr && row < R) ? (uint32_t)row * ldq2 + d0 * 2 : OOB, 0, 0);
        }
        if constexpr (stream_folds(STREAM)) *reinterpret_cast<u32x4 *>(back + (b * NKS + s) * 1024) = p4::scale16x8<T>(x, a.scale2);
        else *reinterpret_cast<u32x4 *>(back + (b * NKS + s) * 1024) = x;
      }
    }
  }

  // ---- traversal range (as attn_fwd16_p5.h)
  const int tiles_total = C / BC;
  const int coff = C - R;   // CAUSAL (extension): row r sees key c iff c <= r + (C - R)
  int nt = tiles_total;
  if constexpr (CAUSAL) {
    const int64_t last_row = min((int64_t)R, ((int64_t)rblk + 1) * GROWS) - 1;
    nt = (int)min((int64_t)tiles_total, (last_row + coff) / BC + 1);
  }
  int wnt = nt;
  if constexpr (CAUSAL) {
    const int64_t wlast = min((int64_t)R, r0 + WROWS) - 1;
    wnt = wlast >= r0 ? (int)max((int64_t)1, min((int64_t)nt, (wlast + coff) / BC + 1)) : 1;
  }
  const int minlim = CAUSAL ? (int)min((int64_t)C - 1, r0 + coff) : C - 1;
  const int maskfrom = __builtin_amdgcn_readfirstlane(CAUSAL ? (minlim + 1) / BC : nt);
  nt = __builtin_amdgcn_readfirstlane(nt);
  wnt = __builtin_amdgcn_readfirstlane(wnt);
  int lim0 = C - 1, lim1 = C - 1;
  if constexpr (CAUSAL) {
    lim0 = (int)min((int64_t)C - 1, r0 + q + coff);
    lim1 = (int)min((int64_t)C - 1, r0 + 32 + q + coff);
  }
  lim0 -= 4 * hi;
  lim1 -= 4 * hi;

  // ---- LDS-DMA staging: piece i of wave w fills 16-byte positions (PW w + i) * 64 + lane of an image
  // (transposed operand: [D elements][4 chunks of 8 keys], chunk index ^ (element >> 2) & 3, a step further = 32 keys along every
  // row; row-major operand: attn_fwd16_p5.h's [D/32][32 keys][4 chunks], chunk swizzled by (key >> 2) & 3, 32 rows further)
  uint32_t koff[4], voff[4];
  const uint32_t kinc = KT ? BC * 2 : BC * ldk2, vinc = VT ? BC * 2 : BC * ldv2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // (the stream uses the first PW of them)
    const int p = (wave * PW + i) * 64 + lane;
    const int d = p >> 2, chunk = (p & 3) ^ ((d >> 2) & 3);
    const int rkey = (p >> 2) & 31, rchunk = (p >> 7) * 4 + ((p & 3) ^ ((rkey >> 2) & 3));
    if constexpr (KT) koff[i] = (i < PW && d < Dr) ? (uint32_t)d * ldk2 + chunk * 16 : OOB;
    else koff[i] = (i < PW && rchunk * 8 < Dr) ? rkey * ldk2 + rchunk * 16 : OOB;
    if constexpr (VT) voff[i] = (i < PW && d < Dr) ? (uint32_t)d * ldv2 + chunk * 16 : OOB;
    else voff[i] = (i < PW && rchunk * 8 < Dr) ? rkey * ldv2 + rchunk * 16 : OOB;
  }
  const uint32_t lds0 = lds_addr(smem);
  const int n16 = lane & 15;
  // lane terms of a transposing read (K^T image; or the V image of a row-major V): rows (n16 >> 2) + 4 hi and + 8 of a 16-element step
  const int trow = (n16 >> 2) + 4 * hi, tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1), thalf = (n16 & 3) & 1;
  const uint32_t tr0 = trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8, tr1 = (trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8;
  uint32_t ka0, ka1;
  if constexpr (KT) {
    ka0 = lds0 + tr0;
    ka1 = lds0 + tr1;
  } else {   // row-major K: the lane's key row, 16 bytes of k-step 2 t + hi (attn_fwd16_p5.h)
    ka0 = lds0 + q * 64 + ((hi ^ ((q >> 2) & 3)) * 16);
    ka1 = lds0 + q * 64 + (((2 + hi) ^ ((q >> 2) & 3)) * 16);
  }
  // V^T: row lane % 32 of a 32-element block, chunk c (keys 8 c + 4 hi + {0..3}); one step behind K: the ring's last stage
  uint32_t ta[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if constexpr (VT) ta[c] = lds0 + (RING - 1) * STAGE + q * 64 + ((c ^ ((q >> 2) & 3)) * 16) + 8 * hi;
    else ta[c] = c == 0 ? lds0 + (RING - 1) * STAGE + tr0 : c == 1 ? lds0 + (RING - 1) * STAGE + tr1 : 0u;   // (two addresses; ta2 / ta3 unused)
  }
  const uint32_t qback = lds0 + wave * 32768 + lane * 16;
  const uint32_t wr0 = lds0 + wave * (PW * 1024), ringend = lds0 + RING_BYTES;

  constexpr float M_INIT = stream_folds(STREAM) ? 0.f : -3.402823466e+38f;   // (see attn_fwd16_p4.h)
  float m0 = M_INIT, m1 = M_INIT, l0 = 0.f, l1 = 0.f;
  const uint32_t onesw = hi ? 0u : (__is_same(T, __bf16) ? 0xBF80BF80u : 0xBC00BC00u);   // -1.0 in k-slots 0, 1
  if (nt > 0) {
    uint32_t tj, tstg, tdelta, tdeltav, twr, tpend, tt0, tt1, tplast, tpa, tpb;
    uint64_t tsv, tptime;
#define MFA_P5TR_RUN(name, fold, d, pattern) if constexpr (STREAM == p5tr::S_##name) MFA_P5TR_TRAVERSE(MFA_P5_STREAM_##name);
    MFA_P5TR_ALL_STREAMS(MFA_P5TR_RUN)
#undef MFA_P5TR_RUN
  } else {   // no keys: O = 0
    static_for<64>([&](auto ic) { p4::acc_write4<4 * decltype(ic)::value>(u32x4{0u, 0u, 0u, 0u}); });
  }

  // ================= epilogue: O /= l (+Source.swift:165-171), L = m + log2 l (+Caching.swift:373-377) =================
  asm volatile("s_nop 15\n\ts_nop 7" ::: MFA_ALL_AGPRS);   // the last accumulating MFMAs leave the matrix pipe
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // every wave is done with the ring
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + wave * (32 * OLD);
  char *lbase = operand_base(a.op[SLOT_L], head, batch);
  const int oprec = a.op[SLOT_O].precision;
  const uint32_t oesz = oprec == PREC_FP32 ? 4u : 2u, ldo = (uint32_t)a.op[SLOT_O].ld;
  const __amdgpu_buffer_rsrc_t otres =
      __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_O], head, batch), 0, (uint32_t)Dr * ldo * oesz, 0x00020000);
  static_for<2>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const float l_tot = half_swap_add(b == 0 ? l0 : l1) + 1.401298464e-45f;   // +Caching.swift:311
    const float inv = l_tot > 1e-30f ? 1.0f / l_tot : 0.f;   // a row without keys: O = 0
    const int64_t row = r0 + b * 32 + q;
    float *orow = Os + q * OLD;
    static_for<NDB>([&](auto dc) {
      constexpr int db = decltype(dc)::value;
      float x[16];
      p4::acc_read16<16 * (b * 8 + db)>(x);
      if (oT) {   // O^T ([D][R]): register r of block db is element 32 db + crow(r, hi) of the lane's row; lanes = consecutive rows
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = 32 * db + crow(r, hi);
          const uint32_t off = (d < Dr && row < R) ? ((uint32_t)d * ldo + (uint32_t)row) * oesz : OOB;
          const float val = x[r] * inv;
          if (oprec == PREC_FP32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), otres, off, 0, 0);
          else __builtin_amdgcn_raw_buffer_store_b16((uint16_t)pack16<T>(val, 0.f), otres, off, 0, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) =
              make_float4(x[4 * g] * inv, x[4 * g + 1] * inv, x[4 * g + 2] * inv, x[4 * g + 3] * inv);
      }
    });
    if (hi == 0 && row < R) store_elem(lbase, row, a.op[SLOT_L].precision, (b == 0 ? m0 : m1) + log2f(l_tot));
    if (!oT)
      store_block_rows<T, D>(Os, operand_base(a.op[SLOT_O], head, batch), a.op[SLOT_O].precision, (uint32_t)a.op[SLOT_O].ld,
                             r0 + 32 * b, R, Dr, lane);
  });
  }   // pass
}

} // namespace mfa
