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
#include "attn_fwd16_common.h"

namespace mfa {

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
