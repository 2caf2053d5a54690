// attn_fwd16.h -- forward attention on the 16-bit matrix cores of gfx950 (BF16 or FP16 inputs).
//
//   O = softmax(Q K^T / sqrt(D)) V   (fp32 out),   L = m + log2(l)
//   reference: loopForward, Sources/FlashAttention/Attention/AttentionKernel/
//              AttentionKernel+Source.swift:158-200 and the pieces it calls (+OuterProduct.swift,
//              +Softmax.swift:228-324, :406-417, +Accumulate.swift, +Caching.swift:302-377)
//
// Structure (one workgroup = NW waves, one wave = RB 32-row query blocks; BC = 64 keys per step):
//   * S^T = K Q^T with v_mfma_f32_32x32x16_{bf16,f16} ("swapped" product): a lane then owns ONE query
//     row -- the online-softmax max / sum are lane-local plus a single cross-half exchange, and the
//     running (m, l) live in two VGPRs.  Q is the B operand and stays in registers (Q "cached").
//   * P is converted in-register (v_cvt_pk) and is DIRECTLY the B operand of O^T += V^T P^T: the
//     contraction index is permuted so that k-slot (8*hi + j) means key 16u + (j&3) + 8(j>>2) + 4hi,
//     which is exactly the set of keys a lane already holds after the first MFMA.
//   * V^T (the A operand of the second product) is gathered from a row-major V tile with
//     ds_read_b64_tr_b16; the LDS image is [D/32][64 keys][32 d] so that each half-wave touches 256
//     contiguous bytes.  K is read with ds_read_b128 from an XOR-swizzled row-major image.
//   * K/V tiles are double buffered in LDS; the global loads of tile j+1 are issued before the
//     MFMAs of tile j and written to LDS after them (one barrier per tile).
//   * ragged edges: rows >= R are clamped on load and guarded on store; keys >= C are zero-filled
//     and masked with the reference's mask value (+Softmax.swift:242-243) on the last tile only.
#pragma once
#include "attn_common.h"

namespace mfa {

template <typename T> struct Frag16;
template <> struct Frag16<__bf16> {
  typedef bf16x8 v8;
  typedef bf16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag16<_Float16> {
  typedef f16x8 v8;
  typedef f16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// two fp32 values -> one dword of the 16-bit storage type, with the reference's store rounding: BF16 by
// truncation (GEMMHeaders.swift:461-471, +Caching.swift:395-401), FP16 round-to-nearest
template <typename T> __device__ __forceinline__ uint32_t pack16(float a, float b) {
  if constexpr (__is_same(T, __bf16)) {
    return (__builtin_bit_cast(uint32_t, a) >> 16) | (__builtin_bit_cast(uint32_t, b) & 0xFFFF0000u);
  } else {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, h);
  }
}

// Whole-row store of a 32-row x D block that a wave has laid out in LDS ([32][D + 4] floats) to a row-major
// operand kept in FP32 or -- fused output cast, SURVEY.md section 8f rank 2 -- in the 16-bit type T.
// `bound` = number of valid rows of the operand (per batch entry), `Dr` = real head dimension.
template <typename T, int D>
__device__ __forceinline__ void store_block_rows(const float *Os, char *base, int prec, uint32_t ld, int64_t r0, int64_t bound,
                                                 int Dr, int lane, float scale = 1.0f) {
  constexpr int OLD = D + 4, CPRO = D / 4;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const uint32_t esz = prec == PREC_FP32 ? 4u : 2u;
  const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(base, 0, (uint32_t)bound * ld * esz, 0x00020000);
#pragma unroll
  for (int i = 0; i < 32 * CPRO / 64; ++i) {
    const int id = lane + i * 64;
    const int rr = id / CPRO, c = id % CPRO;
    float4 val = *reinterpret_cast<const float4 *>(Os + rr * OLD + c * 4);
    val.x *= scale; val.y *= scale; val.z *= scale; val.w *= scale;
    const bool ok = r0 + rr < bound && c * 4 < Dr;
    if (prec == PREC_FP32) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), res, ok ? (uint32_t)(r0 + rr) * ld * 4 + c * 16 : OOB, 0, 0);
    } else {
      const u32x2 h = {pack16<T>(val.x, val.y), pack16<T>(val.z, val.w)};
      __builtin_amdgcn_raw_buffer_store_b64(h, res, ok ? (uint32_t)(r0 + rr) * ld * 2 + c * 8 : OOB, 0, 0);
    }
  }
}

// 16-byte chunk index swizzle of the row-major K image (ds_read_b128 is conflict-free when the 16
// rows of a lane group land on 16 distinct 16-B slots of the 256-B bank row)
template <int D> __device__ __forceinline__ int kswz(int row, int chunk) {
  if constexpr (D >= 128 && (D & (D - 1)) == 0) return chunk ^ (row & 15);
  else if constexpr (D == 64) return chunk ^ ((row >> 1) & 7);
  else return chunk ^ ((row >> 2) & 3);  // D == 32, and the buckets whose row is not a power of two (96, 160, 192 elements:
                                         // the XOR must stay inside the row, chunks per row are a multiple of four)
}

// the row-dependent XOR mask of kswz (kswz(row, c) == c ^ kswz_mask(row)); the swizzle is an involution
template <int D> __device__ __forceinline__ int kswz_mask(int row) { return kswz<D>(row, 0); }

template <int D> constexpr int fwd16_lds_bytes() { return 2 /*buffers*/ * 2 /*K,V*/ * 64 * D * 2; }

// grid: 1-D, (row blocks) x heads x batches flattened; see fwd16_decode_block for the XCD-aware order
struct Fwd16Grid {
  uint32_t rowBlocks, heads, batches;
  // column-parallel ("split-KV") launches only: the key range is cut into `splits` pieces, each
  // workgroup writes un-normalised partial results into the caller's workspace
  //   wsO  [splits][heads*batches][R][D] fp32,  wsML [splits][heads*batches][R][2] = (m, l)
  uint32_t splits;
  float *wsO;
  float *wsML;
};

// (per-lane form: `bid` may differ from lane to lane -- the persistent forward kernel builds its block table with it)
__device__ __forceinline__ void fwd16_decode_block_lane(const Fwd16Grid &g, uint32_t bid, uint32_t *rb,
                                                        uint32_t *head, uint32_t *batch) {
  // Hardware places workgroup b on XCD b % 8 (observed; used for speed only).  All row blocks of
  // one (head, batch) share K and V, so we give each XCD whole heads: its private 4 MiB L2 then
  // holds the K/V of the few heads it is working on.
  // Within an XCD the heads come in PAIRS whose blocks alternate: (block 0, head a), (block 0, head b), (block 1, head a), ...
  // The L2 working set is the same two heads (32 compute units = two heads x 16 blocks of 256 rows at N = 4096), but a causal
  // launch -- whose blocks shrink (or grow) along the block index -- is dealt out longest-first ACROSS the pair: with one head
  // after the other, the in-order dispatcher left the compute units 15 % apart at the end (simulated: 331 against 287 units
  // of time for 8 heads x 16 blocks on 32 units, profiles/r04_causal_dispatch_order.txt); dense launches do not care.
  const uint32_t nh = g.heads * g.batches;
  uint32_t hb, r;
  if ((nh & 7u) == 0) {
    const uint32_t xcd = bid & 7u, slot = bid >> 3;
    const uint32_t hpx = nh >> 3;                          // heads of this XCD
    const uint32_t grp = slot / (2u * g.rowBlocks), t = slot % (2u * g.rowBlocks);
    uint32_t hx;
    if (2u * grp + 1u < hpx) { hx = 2u * grp + (t & 1u); r = t >> 1; }
    else { hx = 2u * grp; r = t; }                         // (an odd head count leaves the last head on its own)
    hb = hx * 8u + xcd;
  } else {
    hb = bid / g.rowBlocks;
    r = bid % g.rowBlocks;
  }
  *rb = r;
  *head = hb % g.heads;
  *batch = hb / g.heads;
}

// the block of a WORKGROUP (`bid` wave-uniform): the integer divisions run on the vector ALU, and hipcc does not move their
// results back to scalar registers by itself when an asm statement asks for an "s" operand derived from them
__device__ __forceinline__ void fwd16_decode_block(const Fwd16Grid &g, uint32_t bid, uint32_t *rb,
                                                   uint32_t *head, uint32_t *batch) {
  uint32_t r, h, b;
  fwd16_decode_block_lane(g, bid, &r, &h, &b);
  *rb = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
  *head = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
  *batch = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
}

template <typename T, int D, int NW, int RB>
__global__ __launch_bounds__(NW * 64) void attn_fwd16(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BC = 64, NT = NW * 64, NDB = D / 32, NKS = D / 16;
  constexpr int ROWB = D * 2;                 // bytes per K row
  constexpr int TILE = BC * D * 2;            // bytes per K (or V) tile
  constexpr int CPR = D / 8;                  // 16-B chunks per row
  constexpr int NCH = BC * CPR / NT;          // chunks per thread per tile
  static_assert(BC * CPR % NT == 0, "tile must divide evenly over the workgroup");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &rblk, &head, &batch);
  int R = a.R, C = a.C;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * (NW * RB * 32) + wave * (RB * 32);

  const char *qbase = operand_base(a.op[SLOT_Q], head, batch);
  const char *kbase = operand_base(a.op[SLOT_K], head, batch);
  const char *vbase = operand_base(a.op[SLOT_V], head, batch);
  const int64_t ldq = a.op[SLOT_Q].ld, ldk = a.op[SLOT_K].ld, ldv = a.op[SLOT_V].ld;
  const int Dr = a.D;  // runtime head dimension (multiple of 8, <= D): chunks beyond it read as zero

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[row][16s + 8hi .. +7]
  v8 qf[RB][NKS];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    int64_t row = r0 + b * 32 + q;
    if (row >= R) row = R - 1;  // clamp (AttentionKernel.swift:224-226)
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int d0 = 16 * s + 8 * hi;
      u32x4 raw = {0u, 0u, 0u, 0u};
      if (d0 < Dr) raw = *reinterpret_cast<const u32x4 *>(qbase + (row * ldq + d0) * 2);
      qf[b][s] = __builtin_bit_cast(v8, raw);
    }
  }

  // ---- staging of K/V tiles: thread handles chunks id = tid + i*NT  (row = id / CPR, c = id % CPR)
  u32x4 kreg[NCH], vreg[NCH];
  auto issue_loads = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * NT;
      const int row = id / CPR, c = id % CPR;
      const int64_t key = (int64_t)c0 + row;
      u32x4 kz = {0u, 0u, 0u, 0u}, vz = {0u, 0u, 0u, 0u};
      if (key < C && c * 8 < Dr) {
        kz = *reinterpret_cast<const u32x4 *>(kbase + (key * ldk + c * 8) * 2);
        vz = *reinterpret_cast<const u32x4 *>(vbase + (key * ldv + c * 8) * 2);
      }
      kreg[i] = kz;
      vreg[i] = vz;
    }
  };
  auto write_tiles = [&](int buf) {
    char *Ks = smem + buf * (2 * TILE);
    char *Vs = Ks + TILE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + i * NT;
      const int row = id / CPR, c = id % CPR;
      *reinterpret_cast<u32x4 *>(Ks + row * ROWB + kswz<D>(row, c) * 16) = kreg[i];
      // V image: [D/32][64 keys][32 d]
      *reinterpret_cast<u32x4 *>(Vs + ((c >> 2) * BC + row) * 64 + (c & 3) * 16) = vreg[i];
    }
  };

  f32x16 o[RB][NDB];
  float m[RB], l[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    m[b] = -3.402823466e+38f;   // +Caching.swift:310
    l[b] = 1.401298464e-45f;    // +Caching.swift:311
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[b][db][r] = 0.f;
  }

  issue_loads(0);
  write_tiles(0);
  __syncthreads();

  // per-lane constant parts of the LDS read addresses
  const int n16 = lane & 15;
  // tr read: lane n of a 16-lane group supplies row (n>>2), columns 4*(n&3)..+3 of a [4][16] block;
  // group (lane>>4): bit0 = d half of the 32-wide d block, bit1 = hi
  const int vtr_off = ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;

  const int ntiles = (C + BC - 1) / BC;
  for (int j = 0; j < ntiles; ++j) {
    const int buf = j & 1;
    const char *Ks = smem + buf * (2 * TILE);
    const char *Vs = Ks + TILE;
    const bool more = (j + 1 < ntiles);
    if (more) issue_loads((j + 1) * BC);

    // ---- S^T = K Q^T: s[b][kb] holds queries (b), keys 64j + 32kb + crow(r, hi)
    f32x16 s[RB][2];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[b][kb][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int krow = 32 * kb + q;
#pragma unroll
      for (int t = 0; t < NKS; ++t) {
        const v8 kf = *reinterpret_cast<const v8 *>(Ks + krow * ROWB + kswz<D>(krow, 2 * t + hi) * 16);
#pragma unroll
        for (int b = 0; b < RB; ++b) s[b][kb] = F::mfma(kf, qf[b][t], s[b][kb]);
      }
    }

    // ---- online softmax (+Softmax.swift:228-324, :406-417), all lane-local except one exchange
    v8 pf[RB][4];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      if (j == ntiles - 1 && (C & (BC - 1)) != 0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (j * BC + 32 * kb + crow(r, hi) >= C) s[b][kb][r] = mask_value();
      }
      float mx = s[b][0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[b][kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = mx * a.scale2;
      if (m_new > m[b]) {   // onlineCorrectO (+Softmax.swift:290-301)
        const float corr = fast_exp2(m[b] - m_new);
        m[b] = m_new;
        l[b] *= corr;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[b][db][r] *= corr;
      }
      float psum = 0.f;
      const float mb = m[b];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(s[b][kb][r] * a.scale2 - mb);
          s[b][kb][r] = p;
          psum += p;
        }
      l[b] += psum;
      // pack P: MFMA step u uses registers 8*(u&1) .. +7 of key block u>>1, in order
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v8 pk;
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = (T)s[b][u >> 1][8 * (u & 1) + i];
        pf[b][u] = pk;
      }
    }

    // ---- O^T += V^T P^T  (A operand gathered by the transposing LDS read)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const char *vp = Vs + (db * BC + 16 * u) * 64 + vtr_off;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + 8 * 64));
        const s16x8 both = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        const v8 vf = __builtin_bit_cast(v8, both);
#pragma unroll
        for (int b = 0; b < RB; ++b) o[b][db] = F::mfma(vf, pf[b][u], o[b][db]);
      }
    }

    if (more) write_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: O /= l (+Source.swift:165-171), L = m + log2(l) (+Caching.swift:373-377)
  char *obase = operand_base(a.op[SLOT_O], head, batch);
  char *lbase = operand_base(a.op[SLOT_L], head, batch);
  const int64_t ldo = a.op[SLOT_O].ld;
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const float l_tot = l[b] + __shfl_xor(l[b], 32);
    const float inv = 1.0f / l_tot;
    const int64_t row = r0 + b * 32 + q;
    if (row < R) {
      float *orow = reinterpret_cast<float *>(obase) + row * ldo;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = 32 * db + 8 * g + 4 * hi;   // crow(4g + i, hi) = i + 8g + 4hi
          if (d0 < Dr)
            *reinterpret_cast<float4 *>(orow + d0) =
                make_float4(o[b][db][4 * g] * inv, o[b][db][4 * g + 1] * inv, o[b][db][4 * g + 2] * inv, o[b][db][4 * g + 3] * inv);
        }
      if (hi == 0) store_elem(lbase, row, a.op[SLOT_L].precision, m[b] + log2f(l_tot));
    }
  }
}

} // namespace mfa
