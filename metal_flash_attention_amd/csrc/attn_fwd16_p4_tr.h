// attn_fwd16_p4_tr.h -- the hand-placed forward kernel (attn_fwd16_p4.h: four waves x 64 rows, one wave per SIMD) for K and / or V
// stored TRANSPOSED ([D][keys], transposeState of AttentionKernelDescriptor.swift:28-42, read where they lie like
// AttentionKernel.swift:189-204): a stream family per pattern (both, K alone, V alone); Q and O either way (run-time flags,
// outside the statement).
//
// The traversal is the same generated statement with two read recipes exchanged (tools/p4gen.py, Cfg.tr): the LDS images keep the
// orientation of the source -- a 16-byte LDS-DMA chunk is 8 consecutive KEYS of one head-dimension element --
//   K^T image [2 blocks of 32 keys][128 elements][64 bytes]        read with ds_read_b64_tr_b16 (as V is in the row-major kernel)
//   V^T image [128 elements][64 keys], chunks ^ (element >> 1 & 7)  read 8 bytes at a time (P^T holds its keys 4 hi + {0..3, 8..11};
//                                                                   sixteen lanes = sixteen rows land on 32 distinct banks)
// and which chunk lands where is decided here, by the lane offsets of the LDS-DMA pieces (128 bytes further per tile).  The
// transposing read returns the contraction index of S^T = K Q^T in that register order too, so the Q fragments are stored in it.
// The tile advances ALONG the rows, so the end of the sequence is not the end of the buffer: the workgroup's last V^T tile is
// fetched through lane offsets of its own (vlast: out of bounds = zeros for chunks at or beyond key C; P is 0 there, but
// 0 x whatever follows the sequence is not), the scores of what follows K^T's keys are replaced by the edge mask.  Chunks cannot
// be cut: launches with column % 8 != 0, rows of K^T / V^T / Q^T that are not 16-byte aligned, per-batch lengths or a block mask
// run on the 8 x 32 kernel's transposed code object instead (attn_fwd16_v3.h, TR), which the launcher falls back to.
// Verified on the lane-exact model like the other streams (tools/p4sim.py run_block with cfg.tr, tests/test_p4_stream.py).
#pragma once
#include "attn_fwd16_p4.h"

namespace mfa {
namespace p4tr {

#define MFA_P4TR_ENUM(name, fold, pattern) S_##name,
enum : int { MFA_P4_TR_STREAM_LIST(MFA_P4TR_ENUM) S_COUNT };
#undef MFA_P4TR_ENUM
constexpr bool stream_folds(int s) {
#define MFA_P4TR_FOLDS(name, fold, pattern) if (s == S_##name) return fold != 0;
  MFA_P4_TR_STREAM_LIST(MFA_P4TR_FOLDS)
#undef MFA_P4TR_FOLDS
  return false;
}
// bit 0 = K, bit 1 = V stored transposed
constexpr int stream_pattern(int s) {
#define MFA_P4TR_PATTERN(name, fold, pattern) if (s == S_##name) return pattern;
  MFA_P4_TR_STREAM_LIST(MFA_P4TR_PATTERN)
#undef MFA_P4TR_PATTERN
  return 0;
}

}  // namespace p4tr

// the statement of attn_fwd16_p4 plus the last tile's V^T offsets
#define MFA_P4TR_TRAVERSE(STREAM)                                                                                        \
  asm volatile(STREAM                                                                                                    \
               : [m0] "+v"(m0), [m1] "+v"(m1), [l0] "+v"(l0), [l1] "+v"(l1), [koff0] "+v"(koff[0]), [koff1] "+v"(koff[1]), \
                 [koff2] "+v"(koff[2]), [koff3] "+v"(koff[3]), [voff0] "+v"(voff[0]), [voff1] "+v"(voff[1]),               \
                 [voff2] "+v"(voff[2]), [voff3] "+v"(voff[3]), [j] "=&s"(tj), [vrd] "=&s"(tvrd), [vwr] "=&s"(tvwr),       \
                 [pend] "=&s"(tpend), [t0] "=&s"(tt0), [t1] "=&s"(tt1), [pa] "=&s"(tpa), [pw] "=&s"(tpw),                 \
                 [pb] "=&s"(tpb), [plast] "=&s"(tplast), [sv] "=&s"(tsv), [ptime] "=&s"(tptime), [selv] "=&s"(tselv),      \
                 [vta] "=&v"(tvta), [vtb] "=&v"(tvtb)                                                                    \
               : [kbase] "v"(kbase), [vbase] "v"(vbase), [lim0] "v"(lim0), [lim1] "v"(lim1), [onesw] "v"(onesw),          \
                 [vlast0] "v"(vlast[0]), [vlast1] "v"(vlast[1]), [vlast2] "v"(vlast[2]), [vlast3] "v"(vlast[3]),          \
                 [kres] "s"(kdesc), [vres] "s"(vdesc), [nt] "s"(nt), [wnt] "s"(wnt), [scale2] "s"(a.scale2), [kinc] "s"(kinc), \
                 [vinc] "s"(vinc), [ldsk] "s"(ldsk), [ldsv] "s"(ldsv), [maskfrom] "s"(maskfrom), [ntm2] "s"(ntm2)           \
               : "memory", "vcc", "scc", MFA_ALL_AGPRS, MFA_P4_OWNED_VGPRS)

// T: __bf16 or _Float16 (must match the stream); CAUSAL only selects the block order and the bounds (as attn_fwd16_p4)
template <typename T, int STREAM, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd16_p4_tr(const KernelArgs a, const Fwd16Grid grid) {
  using namespace p4;
  using p4tr::stream_folds;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = 128, BC = 64, NKS = 8, NDB = 4, WROWS = 64, GROWS = 256;
  constexpr bool KT = (p4tr::stream_pattern(STREAM) & 1) != 0, VT = (p4tr::stream_pattern(STREAM) & 2) != 0;
  typedef __attribute__((address_space(3))) s16x4 *lds_tr_ptr;

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
  int lane = lane0;   // (everything a lane derives from its number is recomputed per pass, see attn_fwd16_p4.h)
  if constexpr (CAUSAL) asm volatile("" : "+v"(lane));
  const int q = lane & 31, hi = lane >> 5;
  uint32_t rblk = rblk0;
  if constexpr (CAUSAL) {
    rblk = pass == 0 ? grid.rowBlocks - 1 - rblk0 : rblk0;
    if (pass == 1 && rblk0 == grid.rowBlocks - 1 - rblk0) break;
    if (pass == 1) __syncthreads();
  }
  const int R = a.R, C = a.C;   // (no per-batch lengths here; C % 8 == 0)
  if ((int64_t)rblk * GROWS >= R) continue;
  const int64_t r0 = (int64_t)rblk * GROWS + wave * WROWS;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2,
                 ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;
  const char *kptr = operand_base(a.op[SLOT_K], head, batch), *vptr = operand_base(a.op[SLOT_V], head, batch);
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
      operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)(qT ? Dr : R) * ldq2, 0x00020000);
  const uint32_t knrec = (uint32_t)(KT ? Dr : C) * ldk2, vnrec = (uint32_t)(VT ? Dr : C) * ldv2;
  const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(kptr), 0, knrec, 0x00020000);
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vptr), 0, vnrec, 0x00020000);
  const uint64_t kaddr = (uint64_t)(uintptr_t)kptr, vaddr = (uint64_t)(uintptr_t)vptr;
  const u32x4 kdesc = {(uint32_t)kaddr, (uint32_t)(kaddr >> 32) & 0xFFFFu, knrec, 0x00020000u};
  const u32x4 vdesc = {(uint32_t)vaddr, (uint32_t)(vaddr >> 32) & 0xFFFFu, vnrec, 0x00020000u};
  constexpr uint32_t OOB = 0xFFFFFF00u;

  typedef __attribute__((address_space(3))) void *lds_ptr;
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass of hipcc does not know this device builtin
#define MFA_P4_DMA(res, dst, off) __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (lds_ptr)(dst), 16, off, 0, 0, 0)
#else
#define MFA_P4_DMA(res, dst, off) ((void)(res), (void)(dst), (void)(off))
#endif
  // ---- Q tile of the wave (64 rows) by LDS-DMA into an image of its own behind the first tiles: row-major Q in a K-tile-shaped
  // image ([64 rows][16 chunks ^ (row & 15)]), Q^T in a K^T-shaped one ([2 blocks of 32 rows][128 elements][64 bytes])
  char *qimg = smem + QSTAGE + wave * 16384;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = i * 64 + lane;
    uint32_t off;
    if (qT) {
      const int d = (p >> 2) & 127;
      const int64_t row = r0 + (p >> 9) * 32 + (p & 3) * 8;   // first of the chunk's eight rows (rows beyond R: never stored)
      off = (d < Dr && row < R) ? (uint32_t)d * ldq2 + (uint32_t)row * 2 : OOB;
    } else {
      const int row = p >> 4, kc = (p & 15) ^ (row & 15);
      off = (kc * 8 < Dr && r0 + row < R) ? (uint32_t)(r0 + row) * ldq2 + kc * 16 : OOB;
    }
    MFA_P4_DMA(qres, qimg + i * 1024, off);
  }
  asm volatile("" ::: "memory");   // keep the Q loads ahead of the DMA pieces in the memory queue

  // ---- traversal range
  const int tiles_total = (C + BC - 1) / BC;
  const int coff = causal_offset(R, C);   // CAUSAL (extension): row r sees key c iff c <= r + (C - R)
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
  const bool ragged = (C & (BC - 1)) != 0 && nt == tiles_total;   // only the globally last tile is partial
  const int minlim = CAUSAL ? (int)min((int64_t)C - 1, r0 + coff) : C - 1;
  const int maskfrom = (CAUSAL || ragged) ? (minlim + 1) / BC : nt;
  const int ntm2 = nt - 2;
  int lim0 = C - 1, lim1 = C - 1;
  if constexpr (CAUSAL) {
    lim0 = (int)min((int64_t)C - 1, r0 + q + coff);
    lim1 = (int)min((int64_t)C - 1, r0 + 32 + q + coff);
  }
  lim0 -= 4 * hi;
  lim1 -= 4 * hi;

  // ---- LDS-DMA staging: piece i of wave w fills 16-byte positions (4 w + i) * 64 + lane of an image
  // (an operand that is NOT transposed keeps the row-major kernel's images: K rows with the chunk index ^ (row & 15), V in
  // sub-images of 32 elements; 64 rows further per tile, zeros past the end of the buffer)
  uint32_t koff[4], voff[4], vlast[4];
  const uint32_t kinc = KT ? BC * 2 : BC * ldk2, vinc = VT ? BC * 2 : BC * ldv2;   // transposed: a tile further = 64 keys along every row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = (wave * 4 + i) * 64 + lane;
    if constexpr (KT) {
      const int kd = (p >> 2) & 127, kkey = (p >> 9) * 32 + (p & 3) * 8;
      koff[i] = (kd < Dr) ? (uint32_t)kd * ldk2 + kkey * 2 : OOB;
    } else {
      const int krow = p >> 4, kc = (p & 15) ^ (krow & 15);
      koff[i] = (kc * 8 < Dr) ? krow * ldk2 + kc * 16 : OOB;
    }
    if constexpr (VT) {
      const int vd = p >> 3, vkey = ((p & 7) ^ ((vd >> 1) & 7)) * 8;
      voff[i] = (vd < Dr) ? (uint32_t)vd * ldv2 + vkey * 2 : OOB;
      // the workgroup's last tile: chunks at or beyond key C are not fetched
      vlast[i] = (vd < Dr && vkey + BC * (nt - 1) < C) ? voff[i] + (uint32_t)(nt - 1) * vinc : OOB;
    } else {
      const int vkey = (p >> 2) & 63, vc = (p >> 8) * 4 + (p & 3);
      voff[i] = (vc * 8 < Dr) ? vkey * ldv2 + vc * 16 : OOB;
      vlast[i] = voff[i];   // (not read by these streams)
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // K(0) -> K image 0
    MFA_P4_DMA(kres, smem + (wave * 4 + i) * 1024, koff[i]);
    koff[i] = __builtin_elementwise_add_sat(koff[i], kinc);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // V(0) -> V image 0
    MFA_P4_DMA(vres, smem + VBASE + (wave * 4 + i) * 1024, (VT && nt == 1) ? vlast[i] : voff[i]);
    voff[i] = __builtin_elementwise_add_sat(voff[i], vinc);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // K(1) -> K image 1
    MFA_P4_DMA(kres, smem + KSLOT + (wave * 4 + i) * 1024, koff[i]);
    koff[i] = __builtin_elementwise_add_sat(koff[i], kinc);
  }
#undef MFA_P4_DMA

  // per-lane LDS read addresses: the transposing read's lane term (K^T); row lane % 32 of the V^T image, the swizzle's XOR mask
  // and 8 hi (the stream XORs the chunk index in)
  const uint32_t lds0 = lds_addr(smem);
  const int n16 = lane & 15;
  const uint32_t trlane = ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;
  const uint32_t kbase = KT ? lds0 + trlane : lds0 + q * 256 + ((hi ^ (q & 15)) << 4);
  const uint32_t vbase = VT ? lds0 + VBASE + q * 128 + (((q >> 1) & 7) << 4) + 8 * hi : lds0 + VBASE + trlane;
  const uint32_t ldsk = lds0 + wave * 4096, ldsv = lds0 + VBASE + wave * 4096;

  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // the wave's own Q image has landed; the 12 K / V pieces stay in flight
  // Q fragments (B operand of S^T = K Q^T: lane = row) -> a[128:191], in the order the K fragments hold the contraction index:
  // with K^T elements 16 s + 4 hi + {0..3, 8..11} (what a transposing read of the Q^T image returns, two 8-byte reads of the
  // row-major one), with row-major K elements 16 s + 8 hi + {0..7} (one 16-byte read; of the Q^T image: the two transposing
  // reads of rows 8 hi and 8 hi + 4 hold them as 4 hi' + {0..3} -- lanes exchange nothing, the image is simply read at
  // element 16 s + 8 hi + 4 (lane's read half) through per-half lane terms)
  static_for<2>([&](auto bc) {
    static_for<NKS>([&](auto sc) {
      constexpr int b = decltype(bc)::value, s = decltype(sc)::value;
      u32x4 qx;
      if (qT) {
        if constexpr (KT) {
          const char *qp = qimg + (b * 128 + 16 * s) * 64 + trlane;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(qp));
          const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(qp + 8 * 64));
          qx = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7));
        } else {
          // elements 16 s + 8 hi + {0..3} and + {4..7}: rows (n16 >> 2) + 8 hi (+ 4) of the image instead of (n16 >> 2) + 4 hi (+ 8)
          const char *qp = qimg + (b * 128 + 16 * s) * 64 + trlane + 4 * hi * 64;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(qp));
          const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(qp + 4 * 64));
          qx = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      } else if constexpr (KT) {
        const char *qrow = qimg + (b * 32 + q) * 256 + 8 * hi;
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(qrow + (((2 * s) ^ (q & 15)) << 4));
        const u32x2 up = *reinterpret_cast<const u32x2 *>(qrow + (((2 * s + 1) ^ (q & 15)) << 4));
        qx = u32x4{lo[0], lo[1], up[0], up[1]};
      } else {
        qx = *reinterpret_cast<const u32x4 *>(qimg + (b * 32 + q) * 256 + (((2 * s + hi) ^ (q & 15)) << 4));
      }
      if constexpr (stream_folds(STREAM)) acc_write4<Q_BASE + 4 * (b * 8 + s)>(scale16x8<T>(qx, a.scale2));
      else acc_write4<Q_BASE + 4 * (b * 8 + s)>(qx);
    });
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is free again before the statement's first barrier

  constexpr float M_INIT = stream_folds(STREAM) ? 0.f : -3.402823466e+38f;   // (see attn_fwd16_p4.h)
  float m0 = M_INIT, m1 = M_INIT, l0 = 0.f, l1 = 0.f;
  const uint32_t onesw = hi ? 0u : (__is_same(T, __bf16) ? 0xBF80BF80u : 0xBC00BC00u);   // -1.0 in k-slots 0, 1
  {
    uint32_t tj, tvrd, tvwr, tpend, tt0, tt1, tplast, tpa, tpw, tpb, tvta, tvtb;
    uint64_t tsv, tptime, tselv;
#define MFA_P4TR_RUN(name, fold, pattern) if constexpr (STREAM == p4tr::S_##name) MFA_P4TR_TRAVERSE(MFA_P4_STREAM_##name);
    MFA_P4_TR_STREAM_LIST(MFA_P4TR_RUN)
#undef MFA_P4TR_RUN
  }

  // ================= epilogue: O /= l (+Source.swift:165-171), L = m + log2 l (+Caching.swift:373-377) =================
  asm volatile("s_nop 15\n\ts_nop 7" ::: MFA_ALL_AGPRS);   // the last accumulating MFMAs leave the matrix pipe
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // run-ahead DMA tiles have landed
  __syncthreads();   // every wave is done with the ring
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + wave * (WROWS * OLD);
  char *lbase = operand_base(a.op[SLOT_L], head, batch);
  const int oprec = a.op[SLOT_O].precision;
  const uint32_t oesz = oprec == PREC_FP32 ? 4u : 2u, ldo = (uint32_t)a.op[SLOT_O].ld;
  const __amdgpu_buffer_rsrc_t otres =
      __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_O], head, batch), 0, (uint32_t)Dr * ldo * oesz, 0x00020000);
  static_for<2>([&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const float l_tot = half_swap_add(b == 0 ? l0 : l1) + 1.401298464e-45f;   // +Caching.swift:311
    const float inv = l_tot > 1e-30f ? 1.0f / l_tot : 0.f;
    const int64_t row = r0 + b * 32 + q;
    float *orow = Os + (b * 32 + q) * OLD;
    static_for<NDB>([&](auto dc) {
      constexpr int db = decltype(dc)::value;
      float x[16];
      acc_read16<O_BASE + 16 * (b * 4 + db)>(x);
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
  });
  if (!oT) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
      store_block_rows<T, D>(Os + b * 32 * OLD, operand_base(a.op[SLOT_O], head, batch), a.op[SLOT_O].precision,
                             (uint32_t)a.op[SLOT_O].ld, r0 + 32 * b, R, Dr, lane);
  }
  }   // pass
}

} // namespace mfa
