// attn_f32.h -- the three attention kernels for the FP32 production case (BASELINE config 3): every operand FP32, row-major,
// 16-byte aligned, head dimension a multiple of 4 and <= DP (64 or 128).  Same arithmetic as the general kernels of
// attn_generic.h (all products on v_mfma_f32_32x32x2_f32, exact online softmax), which keep every other layout / type / mask;
// what differs is how the operands travel:
//   * the traversal-side tiles (K, V / Q, dO: 32 rows x DP floats) go from memory to LDS by LDS-DMA, two stages, one barrier
//     per tile; bounds-checked buffer resources zero-fill ragged rows and the columns past D (what the reference gets from
//     simdgroup_event::async_copy, GEMMHeaders.swift:166-193);
//   * the cached left-hand operands (+Caching.swift:18-281) are read from memory straight into their fragment registers;
//   * every LDS read is a ds_read_b128 that feeds four matrix instructions.  The contraction index of the first product of
//     a pair and the head-dimension index of the second are both PERMUTED so that the four values lie next to each other in a
//     row-major tile: step t of a first product contracts d = 8 (t >> 2) + 4 hi + (t & 3); accumulator block db of a second
//     product holds d = NDB i + db on lane i (NDB = DP / 32) -- a lane then owns NDB consecutive floats of an output row and
//     stores them to memory directly;
//   * rows of a tile are 16 DP / 4 chunks of 16 bytes, chunk c of row r stored at c ^ (r & 15): both read patterns are
//     conflict-free and the image is what a linear LDS-DMA of permuted global addresses produces.
// The reads are inline asm with counted waits (hipcc drains vmcnt in front of LDS reads it cannot prove disjoint from an
// LDS-DMA in flight, attn_fwd16_v3.h).
#pragma once
#include "attn_common.h"
#include "attn_fwd16.h"   // Fwd16Grid, fwd16_decode_block (XCD-aware workgroup order)
#include <type_traits>

namespace mfa {
namespace f32k {

typedef __attribute__((address_space(3))) void *lds_ptr;
constexpr uint32_t OOB = 0xFFFFFF00u;
constexpr int BT = 32;   // rows of a traversal-side tile

__device__ __forceinline__ uint32_t lds_addr(const char *p) {
  return (uint32_t)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char *)p;
}
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
template <int OFF> __device__ __forceinline__ f32x4 rd128(uint32_t addr) {
  f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF> __device__ __forceinline__ f32x2 rd64(uint32_t addr) {
  f32x2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// at most N younger LDS reads may still be pending when `a` is used (LDS returns in order, asm volatile keeps its order)
template <int N> __device__ __forceinline__ void lds_wait(f32x4 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N < 15 ? N : 15)); }
template <int N> __device__ __forceinline__ void lds_wait(f32x2 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N < 15 ? N : 15)); }
template <int N> __device__ __forceinline__ void lds_wait(f32x4 &a, f32x4 &b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N < 15 ? N : 15));
}
template <int N> __device__ __forceinline__ void lds_wait(f32x2 &a, f32x2 &b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N < 15 ? N : 15));
}

// lane l <-> lane l ^ 32 (v_permlane32_swap; the s_nop covers the VALU-write -> permlane-read hazard inside the asm string)
__device__ __forceinline__ void half_swap(float x, float *a, float *b) {
  uint32_t b0 = __builtin_bit_cast(uint32_t, x), b1 = b0;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(b0), "+v"(b1));
  *a = __builtin_bit_cast(float, b0);
  *b = __builtin_bit_cast(float, b1);
}
__device__ __forceinline__ float half_max(float x) { float a, b; half_swap(x, &a, &b); return fmaxf(a, b); }
__device__ __forceinline__ float half_add(float x) { float a, b; half_swap(x, &a, &b); return a + b; }

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int DP> struct Geo {
  static_assert(DP == 64 || DP == 128, "head-dimension buckets of the FP32 kernels");
  static constexpr int ROWB = DP * 4;            // bytes per tile row
  static constexpr int CPR = DP / 4;             // 16-byte chunks per row
  static constexpr int TILE = BT * ROWB;         // bytes per tile
  static constexpr int NI = CPR / 8;             // LDS-DMA instructions per wave and tile (4 waves x 64 lanes x 16 bytes each)
  static constexpr int NDB = DP / 32;            // accumulator blocks of a second product
  static constexpr int NG = DP / 8;              // b128 reads of a first product (four contraction steps each)
  static constexpr int RB2 = NDB * 4;            // bytes a lane reads per step of a second product
};

// resource over the rows of one operand of one (head, batch): `rows` x ld floats, reads past the end return zero
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_resource(const OperandView &v, uint32_t head, uint32_t batch, int rows) {
  return __builtin_amdgcn_make_buffer_rsrc(operand_base(v, head, batch), 0, (uint32_t)rows * (uint32_t)v.ld * 4u, 0x00020000);
}

// cached fragments of a first product's right-hand operand: f[4 T + i] = X[row][8 T + 4 hi + i]
template <int DP> __device__ __forceinline__ void load_fragments(float *f, const __amdgpu_buffer_rsrc_t &res, uint32_t rowoff, bool valid, int hi, int D) {
  static_for<Geo<DP>::NG>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    const int d0 = 8 * T + 4 * hi;
    // (the whole vector is cast: __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the index, hipcc 7.2)
    const f32x4 raw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(res, (valid && d0 < D) ? rowoff + d0 * 4 : OOB, 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) f[4 * T + i] = raw[i];
  });
}

// the tile stager of one operand: per-lane global offsets of the NI chunks this lane moves per tile
template <int DP> struct Stager {
  uint32_t off[Geo<DP>::NI];
  uint32_t inc;
  __device__ __forceinline__ void init(int wave, int lane, uint32_t ld, int D, uint32_t row0) {
    typedef Geo<DP> G;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
      const int p = (wave * G::NI + i) * 64 + lane;
      const int r = p / G::CPR, c = (p % G::CPR) ^ (r & 15);
      off[i] = (c * 4 < D) ? (row0 + r) * ld * 4u + c * 16 : OOB;
    }
    inc = BT * ld * 4u;
  }
  // the next tile in sequence -> LDS image at `base` (workgroup-relative bytes)
  __device__ __forceinline__ void issue(const __amdgpu_buffer_rsrc_t &res, char *base, int wave) {
    typedef Geo<DP> G;
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass of hipcc does not know this device builtin
      __builtin_amdgcn_raw_ptr_buffer_load_lds(res, (lds_ptr)(base + (wave * G::NI + i) * 1024), 16, off[i], 0, 0, 0);
#endif
      off[i] = __builtin_elementwise_add_sat(off[i], inc);
    }
  }
};

// read addresses of a lane inside a tile (bytes, without the tile's base)
//   first product:  row i, chunk (2 T + hi) ^ (i & 15)            = first ^ (T << 5)
//   second product: row crow(t, hi), NDB floats at column NDB i   = second ^ (ct(t) << 4) + ((t & 3) + 8 (t >> 2)) ROWB,
//                   ct(t) = (t & 3) | 8 ((t >> 2) & 1)  (the row's swizzle bits that do not depend on the lane)
template <int DP> __device__ __forceinline__ uint32_t first_address(int i, int hi) { return i * Geo<DP>::ROWB + ((hi ^ (i & 15)) << 4); }
template <int DP> __device__ __forceinline__ uint32_t second_address(int i, int hi) {
  const int byte = i * Geo<DP>::RB2;
  return 4 * hi * Geo<DP>::ROWB + ((((byte >> 4) ^ (4 * hi)) << 4) | (byte & 15));
}
constexpr int second_ct(int t) { return (t & 3) | (8 * ((t >> 2) & 1)); }
constexpr int second_row(int t) { return (t & 3) + 8 * (t >> 2); }

// acc (+)= X_tile (first pattern) . f : NG reads, four matrix instructions each, reads RING groups ahead
template <int DP, typename Acc> __device__ __forceinline__ void first_product(Acc &acc, uint32_t base, const float *f) {
  typedef Geo<DP> G;
  constexpr int RING = 4;
  f32x4 ring[RING];
  static_for<RING>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    ring[T] = rd128<0>(base ^ (T << 5));
  });
  static_for<G::NG>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    constexpr int pending = (G::NG - 1 - T) < (RING - 1) ? (G::NG - 1 - T) : (RING - 1);
    lds_wait<pending>(ring[T % RING]);
    const f32x4 v = ring[T % RING];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = mfma(v[i], f[4 * T + i], acc);
    if constexpr (T + RING < G::NG) ring[T % RING] = rd128<0>(base ^ ((T + RING) << 5));
  });
}

// two first products side by side (independent accumulators: the matrix instructions alternate)
template <int DP> __device__ __forceinline__ void first_product_pair(f32x16 &acc0, uint32_t base0, const float *f0, f32x16 &acc1, uint32_t base1,
                                                                     const float *f1) {
  typedef Geo<DP> G;
  constexpr int RING = 3;
  f32x4 r0[RING], r1[RING];
  static_for<RING>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    r0[T] = rd128<0>(base0 ^ (T << 5));
    r1[T] = rd128<0>(base1 ^ (T << 5));
  });
  static_for<G::NG>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    constexpr int left = G::NG - 1 - T;
    constexpr int pending = 2 * (left < (RING - 1) ? left : (RING - 1));
    lds_wait<pending>(r0[T % RING], r1[T % RING]);
    const f32x4 v0 = r0[T % RING], v1 = r1[T % RING];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc0 = mfma(v0[i], f0[4 * T + i], acc0);
      acc1 = mfma(v1[i], f1[4 * T + i], acc1);
    }
    if constexpr (T + RING < G::NG) {
      r0[T % RING] = rd128<0>(base0 ^ ((T + RING) << 5));
      r1[T % RING] = rd128<0>(base1 ^ ((T + RING) << 5));
    }
  });
}

// one read of the second pattern: step t of the tile at `base` (+ the lane's second_address)
template <int DP, int T> __device__ __forceinline__ auto second_read(uint32_t base) {
  typedef Geo<DP> G;
  if constexpr (G::NDB == 4) return rd128<second_row(T) * G::ROWB>(base ^ (second_ct(T) << 4));
  else return rd64<second_row(T) * G::ROWB>(base ^ (second_ct(T) << 4));
}
template <int DP> struct SecondRing {
  typedef std::conditional_t<Geo<DP>::NDB == 4, f32x4, f32x2> Vec;
  static constexpr int RING = 4;
  Vec v[RING];
};
// the first RING reads of a second product (issued early: they are in flight during the softmax arithmetic)
template <int DP> __device__ __forceinline__ void second_prefetch(SecondRing<DP> &ring, uint32_t base) {
  static_for<SecondRing<DP>::RING>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    ring.v[T] = second_read<DP, T>(base);
  });
}
// the same for two tiles read side by side: a0 b0 a1 b1 ... (second_product_pair waits for them in that order); HALF = 0, 1:
// the first / second half of the ring (the caller may put a wait between them: the counter holds 15)
template <int DP, int HALF> __device__ __forceinline__ void second_prefetch_pair(SecondRing<DP> &r0, uint32_t base0, SecondRing<DP> &r1, uint32_t base1) {
  constexpr int H = SecondRing<DP>::RING / 2;
  static_for<H>([&](auto T_) {
    constexpr int T = decltype(T_)::value + HALF * H;
    r0.v[T] = second_read<DP, T>(base0);
    r1.v[T] = second_read<DP, T>(base1);
  });
}
// acc[db] += X_tile^T (second pattern) . p : 16 reads, NDB matrix instructions each.  `extra` = LDS reads issued after the
// prefetch that are still allowed to be pending (none of them older than the ring's)
template <int DP> __device__ __forceinline__ void second_product(f32x16 *acc, SecondRing<DP> &ring, uint32_t base, const f32x16 &p) {
  typedef Geo<DP> G;
  constexpr int RING = SecondRing<DP>::RING;
  static_for<16>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    constexpr int pending = (15 - T) < (RING - 1) ? (15 - T) : (RING - 1);
    lds_wait<pending>(ring.v[T % RING]);
    const auto v = ring.v[T % RING];
#pragma unroll
    for (int db = 0; db < G::NDB; ++db) acc[db] = mfma(v[db], p[T], acc[db]);
    if constexpr (T + RING < 16) ring.v[T % RING] = second_read<DP, T + RING>(base);
  });
}
// two second products side by side (dV and dK): reads of tile a and tile b alternate
template <int DP> __device__ __forceinline__ void second_product_pair(f32x16 *acc0, uint32_t base0, const f32x16 &p0, f32x16 *acc1, uint32_t base1,
                                                                      const f32x16 &p1, SecondRing<DP> &r0, SecondRing<DP> &r1) {
  typedef Geo<DP> G;
  constexpr int RING = SecondRing<DP>::RING;
  static_for<16>([&](auto T_) {
    constexpr int T = decltype(T_)::value;
    constexpr int left = 15 - T;
    constexpr int pending = 2 * (left < (RING - 1) ? left : (RING - 1));
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0.v[T % RING]), "+v"(r1.v[T % RING]) : "n"(pending < 15 ? pending : 15));
    const auto v0 = r0.v[T % RING];
    const auto v1 = r1.v[T % RING];
#pragma unroll
    for (int db = 0; db < G::NDB; ++db) {
      acc0[db] = mfma(v0[db], p0[T], acc0[db]);
      acc1[db] = mfma(v1[db], p1[T], acc1[db]);
    }
    if constexpr (T + RING < 16) {
      r0.v[T % RING] = second_read<DP, T + RING>(base0);
      r1.v[T % RING] = second_read<DP, T + RING>(base1);
    }
  });
}

// a lane's share of an output row block: acc[db][r] = X[row of the lane][NDB crow(r, hi) + db]
template <int DP> __device__ __forceinline__ void store_rows(const f32x16 *acc, const OperandView &v, uint32_t head, uint32_t batch, int64_t row, int rows,
                                                             int hi, int D, float factor) {
  typedef Geo<DP> G;
  if (row >= rows) return;
  float *dst = reinterpret_cast<float *>(operand_base(v, head, batch)) + row * v.ld;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = G::NDB * crow(r, hi);
    if (d < D) {
      if constexpr (G::NDB == 4) *reinterpret_cast<f32x4 *>(dst + d) = f32x4{acc[0][r] * factor, acc[1][r] * factor, acc[2][r] * factor, acc[3][r] * factor};
      else *reinterpret_cast<f32x2 *>(dst + d) = f32x2{acc[0][r] * factor, acc[1][r] * factor};
    }
  }
}

template <int DP> constexpr int lds_bytes() { return 2 /*stages*/ * 2 /*operands*/ * Geo<DP>::TILE; }
template <int DP> constexpr int lds_bytes_dkv() { return lds_bytes<DP>() + 2 /*stages*/ * 512; }   // + the L and D slices of a tile
constexpr int ROWS = 128;   // rows (forward, dQ) or columns (dK/dV) of a workgroup: four waves x 32

// ----------------------------------------------------------------------------------------------
// forward: O = softmax(Q K^T / sqrt(D)) V,  L = m + log2(l)       (+Source.swift:158-200)
// grid: 1-D, ceil(R / 128) x heads x batches in fwd16_decode_block's order; 256 threads; two workgroups per compute unit
// ----------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256, 2) void attn_f32_fwd(const KernelArgs a, const Fwd16Grid grid) {
  typedef Geo<DP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &rblk, &head, &batch);
  if (a.causal) rblk = grid.rowBlocks - 1 - rblk;   // later row blocks traverse more keys: start them first
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * ROWS;
  if (r0 >= R) return;
  const int64_t row = r0 + wave * 32 + q;
  const float scale2 = a.scale2;
  const bool causal = a.causal != 0;

  const __amdgpu_buffer_rsrc_t qres = rows_resource(a.op[SLOT_Q], head, batch, R);
  const __amdgpu_buffer_rsrc_t kres = rows_resource(a.op[SLOT_K], head, batch, C);
  const __amdgpu_buffer_rsrc_t vres = rows_resource(a.op[SLOT_V], head, batch, C);
  float qf[DP / 2];
  load_fragments<DP>(qf, qres, (uint32_t)row * (uint32_t)a.op[SLOT_Q].ld * 4u, row < R, hi, D);

  f32x16 o[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m = -3.402823466e+38f;          // +Caching.swift:310
  float l = 1.401298464e-45f;           // +Caching.swift:311 (denorm_min)

  // causal extension: row r sees column c iff c <= r + coff; columns past the last row's limit are never visited
  const int coff = causal_offset(R, C);
  const int Cend = causal ? (int)min((int64_t)C, min((int64_t)R, r0 + ROWS) + coff) : C;
  const int nt = (Cend + BT - 1) / BT;
  Stager<DP> ks, vs;
  ks.init(wave, lane, (uint32_t)a.op[SLOT_K].ld, D, 0);
  vs.init(wave, lane, (uint32_t)a.op[SLOT_V].ld, D, 0);
  const uint32_t lbase = lds_addr(smem);
  const uint32_t first = first_address<DP>(q, hi), second = second_address<DP>(q, hi);
  const int64_t limit = row + coff;
  if (nt > 0) {
    ks.issue(kres, smem, wave);
    vs.issue(vres, smem + G::TILE, wave);
  }
  for (int j = 0; j < nt; ++j) {
    const int c0 = j * BT;
    const uint32_t stage = (uint32_t)(j & 1) * (2 * G::TILE);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces of tile j have landed
    __syncthreads();                      // ... everybody's; and every wave has finished tile j - 1, whose stage is written next
    if (j + 1 < nt) {
      ks.issue(kres, smem + (stage ^ (2 * G::TILE)), wave);
      vs.issue(vres, smem + (stage ^ (2 * G::TILE)) + G::TILE, wave);
    }
    // S^T = K Q^T : lane holds query `row`, keys c0 + crow(r, hi)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    first_product<DP>(s, (lbase + stage) + first, qf);
    SecondRing<DP> vring;
    second_prefetch<DP>(vring, (lbase + stage + G::TILE) + second);
    if (c0 + BT > C) { // maskAttentionMatrixEdge, +Softmax.swift:228-260
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) >= C) s[r] = mask_value();
    }
    if (causal && c0 + BT - 1 > r0 + wave * 32 + coff) {    // same mechanism, applied to the columns the row may not see
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) > limit) s[r] = mask_value();
    }
    // onlineReduceMaximum / onlineCorrectO, +Softmax.swift:267-301
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    const float m_new = half_max(mx) * scale2;
    // corr = (m_new > m) ? exp2(m - m_new) : 1.  When no lane of the wave saw its maximum grow the correction is exactly 1
    // everywhere: skip the O-wide multiply (wave-uniform branch, same results)
    if (__builtin_amdgcn_ballot_w64(m_new > m) != 0) {
      float corr = 1.f;
      if (m_new > m) { corr = fast_exp2(m - m_new); m = m_new; }
      l *= corr;
#pragma unroll
      for (int db = 0; db < G::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= corr;
    }
    // softmax + onlineReduceSum, +Softmax.swift:304-324, :406-417
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(s[r] * scale2 - m); psum += s[r]; }
    l += psum;
    // O^T += V^T P^T with the key index permuted: step t uses key crow(t, hi)
    second_product<DP>(o, vring, (lbase + stage + G::TILE) + second, s);
  }

  const float l_tot = half_add(l);
  const float inv = (m > -1e37f) ? 1.0f / l_tot : 0.f;   // +Source.swift:165-171 (0: a row whose every block is masked out)
  store_rows<DP>(o, a.op[SLOT_O], head, batch, row, R, hi, D, inv);
  if (hi == 0 && row < R)  // L = m + log2(l), +Caching.swift:373-377
    reinterpret_cast<float *>(operand_base(a.op[SLOT_L], head, batch))[row] = m + log2f(l_tot);
}

// ----------------------------------------------------------------------------------------------
// backward dQ: D = rowsum(dO*O)/sqrt(D); dQ = sum_c dS K                (+Source.swift:202-242)
// same grid; one workgroup per compute unit (Q and dO fragments, the dQ accumulators: 192 + 32 registers before any buffer)
// ----------------------------------------------------------------------------------------------
// ABL (developer builds, timing only -- results are wrong): 1 no softmax arithmetic, 2 no barrier / staging in the loop,
// 4 no second product, 8 no first products
template <int DP, int ABL = 0>
__global__ __launch_bounds__(256) void attn_f32_dq(const KernelArgs a, const Fwd16Grid grid) {
  typedef Geo<DP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &rblk, &head, &batch);
  if (a.causal) rblk = grid.rowBlocks - 1 - rblk;   // later row blocks traverse more keys: start them first
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t r0 = (int64_t)rblk * ROWS;
  if (r0 >= R) return;
  const int64_t row = r0 + wave * 32 + q;
  const float scale = a.scale, scale2 = a.scale2;
  const bool causal = a.causal != 0;

  const __amdgpu_buffer_rsrc_t qres = rows_resource(a.op[SLOT_Q], head, batch, R);
  const __amdgpu_buffer_rsrc_t gres = rows_resource(a.op[SLOT_dO], head, batch, R);
  const __amdgpu_buffer_rsrc_t ores = rows_resource(a.op[SLOT_O], head, batch, R);
  const __amdgpu_buffer_rsrc_t kres = rows_resource(a.op[SLOT_K], head, batch, C);
  const __amdgpu_buffer_rsrc_t vres = rows_resource(a.op[SLOT_V], head, batch, C);
  float qf[DP / 2], gf[DP / 2];
  load_fragments<DP>(qf, qres, (uint32_t)row * (uint32_t)a.op[SLOT_Q].ld * 4u, row < R, hi, D);
  load_fragments<DP>(gf, gres, (uint32_t)row * (uint32_t)a.op[SLOT_dO].ld * 4u, row < R, hi, D);
  // computeD, +Softmax.swift:32-221: D = (sum_d dO*O) * 1/sqrt(D); the two half-waves split the head dimension
  float dterm = 0.f;
  {
    float of[DP / 2];
    load_fragments<DP>(of, ores, (uint32_t)row * (uint32_t)a.op[SLOT_O].ld * 4u, row < R, hi, D);
#pragma unroll
    for (int t = 0; t < DP / 2; ++t) dterm += of[t] * gf[t];
    dterm = half_add(dterm) * scale;
  }
  float Lrow = 0.f;
  if (row < R) Lrow = reinterpret_cast<const float *>(operand_base(a.op[SLOT_L], head, batch))[row];

  f32x16 acc[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

  const int coff = causal_offset(R, C);
  const int Cend = causal ? (int)min((int64_t)C, min((int64_t)R, r0 + ROWS) + coff) : C;
  const int nt = (Cend + BT - 1) / BT;
  Stager<DP> ks, vs;
  ks.init(wave, lane, (uint32_t)a.op[SLOT_K].ld, D, 0);
  vs.init(wave, lane, (uint32_t)a.op[SLOT_V].ld, D, 0);
  const uint32_t lbase = lds_addr(smem);
  const uint32_t first = first_address<DP>(q, hi), second = second_address<DP>(q, hi);
  const int64_t limit = row + coff;
  if (nt > 0) {
    ks.issue(kres, smem, wave);
    vs.issue(vres, smem + G::TILE, wave);
  }
  for (int j = 0; j < nt; ++j) {
    const int c0 = j * BT;
    const uint32_t stage = (uint32_t)(j & 1) * (2 * G::TILE);
    if constexpr (!(ABL & 2)) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    if (!(ABL & 2) && j + 1 < nt) {
      ks.issue(kres, smem + (stage ^ (2 * G::TILE)), wave);
      vs.issue(vres, smem + (stage ^ (2 * G::TILE)) + G::TILE, wave);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    // S^T = K Q^T, dP^T = V dO^T
    if constexpr (!(ABL & 8)) first_product_pair<DP>(s, (lbase + stage) + first, qf, dp, (lbase + stage + G::TILE) + first, gf);
    SecondRing<DP> kring;
    second_prefetch<DP>(kring, (lbase + stage) + second);
    // P = exp2(S*scale2 - L); dS = P * (dP*scale - D)     (+Softmax.swift:409-427).  Padded columns: K, V rows are zero, so
    // dS * K contributes nothing (as in the reference, where the zero padding comes from the async copy, +Accumulate.swift:330-346)
    if constexpr (!(ABL & 1)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = fast_exp2(s[r] * scale2 - Lrow);
        if (causal && c0 + crow(r, hi) > limit) p = 0.f;   // masked column: P = 0, hence dS = 0
        s[r] = p * (dp[r] * scale - dterm);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] += dp[r];
    }
    // dQ^T += K^T dS^T, key index permuted as in forward
    if constexpr (!(ABL & 4)) second_product<DP>(acc, kring, (lbase + stage) + second, s);
    else { lds_wait<0>(kring.v[0]); lds_wait<0>(kring.v[1]); lds_wait<0>(kring.v[2]); lds_wait<0>(kring.v[3]); acc[0] += s; acc[1][0] += kring.v[0][0] + kring.v[1][0] + kring.v[2][0] + kring.v[3][0]; }
  }
  store_rows<DP>(acc, a.op[SLOT_dQ], head, batch, row, R, hi, D, 1.f);
  if (hi == 0 && row < R)   // +Caching.swift:381-413
    reinterpret_cast<float *>(operand_base(a.op[SLOT_D], head, batch))[row] = dterm;
}

// ----------------------------------------------------------------------------------------------
// backward dK/dV: dV = sum_r P^T dO ; dK = sum_r dS^T Q, parallel over columns     (+Source.swift:244-293)
// grid: 1-D, ceil(C / 128) x heads x batches; one workgroup per compute unit
// ----------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void attn_f32_dkv(const KernelArgs a, const Fwd16Grid grid) {
  typedef Geo<DP> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, kc = lane & 31, hi = lane >> 5;
  uint32_t cblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &cblk, &head, &batch);
  int R = a.R, C = a.C;
  const int D = a.D;
  batch_lengths(a, batch, R, C);
  const int64_t c0 = (int64_t)cblk * ROWS;
  if (c0 >= C) return;
  const int64_t col = c0 + wave * 32 + kc;
  const float scale = a.scale, scale2 = a.scale2;
  const bool causal = a.causal != 0;

  const __amdgpu_buffer_rsrc_t kres = rows_resource(a.op[SLOT_K], head, batch, C);
  const __amdgpu_buffer_rsrc_t vres = rows_resource(a.op[SLOT_V], head, batch, C);
  const __amdgpu_buffer_rsrc_t qres = rows_resource(a.op[SLOT_Q], head, batch, R);
  const __amdgpu_buffer_rsrc_t gres = rows_resource(a.op[SLOT_dO], head, batch, R);
  const __amdgpu_buffer_rsrc_t lres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_L], head, batch), 0, (uint32_t)R * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_D], head, batch), 0, (uint32_t)R * 4u, 0x00020000);
  float kf[DP / 2], vf[DP / 2];
  load_fragments<DP>(kf, kres, (uint32_t)col * (uint32_t)a.op[SLOT_K].ld * 4u, col < C, hi, D);
  load_fragments<DP>(vf, vres, (uint32_t)col * (uint32_t)a.op[SLOT_V].ld * 4u, col < C, hi, D);

  f32x16 dk[G::NDB], dv[G::NDB];
#pragma unroll
  for (int db = 0; db < G::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  // causal extension: rows above the workgroup's first column minus the offset see none of its columns
  const int coff = causal_offset(R, C);
  const int rstart = causal ? (int)(max((int64_t)0, c0 - coff) / BT) * BT : 0;
  const int nt = (R - rstart + BT - 1) / BT;
  Stager<DP> qs, gs;
  qs.init(wave, lane, (uint32_t)a.op[SLOT_Q].ld, D, (uint32_t)rstart);
  gs.init(wave, lane, (uint32_t)a.op[SLOT_dO].ld, D, (uint32_t)rstart);
  const uint32_t lbase = lds_addr(smem);
  const uint32_t first = first_address<DP>(kc, hi), second = second_address<DP>(kc, hi);
  // L and D slices along the traversal dimension (+Softmax.swift:356-381, :472-503): 32 floats each per tile, staged like the
  // tiles (wave 0: L, wave 1: D; one dword per lane, the upper 32 lanes' rows belong to the next tile and are not read)
  char *ldst = smem + lds_bytes<DP>();
  const uint32_t ldoff = (uint32_t)(rstart + lane) * 4u;
  auto issue_ld = [&](int tile) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(lres, (lds_ptr)(ldst + (tile & 1) * 512), 4, ldoff + (uint32_t)tile * (BT * 4u), 0, 0, 0);
    if (wave == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(dres, (lds_ptr)(ldst + (tile & 1) * 512 + 256), 4, ldoff + (uint32_t)tile * (BT * 4u), 0, 0, 0);
#endif
  };
  if (nt > 0) {
    qs.issue(qres, smem, wave);
    gs.issue(gres, smem + G::TILE, wave);
    issue_ld(0);
  }
  for (int j = 0; j < nt; ++j) {
    const int rr0 = rstart + j * BT;
    const uint32_t stage = (uint32_t)(j & 1) * (2 * G::TILE);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (j + 1 < nt) {
      qs.issue(qres, smem + (stage ^ (2 * G::TILE)), wave);
      gs.issue(gres, smem + (stage ^ (2 * G::TILE)) + G::TILE, wave);
      issue_ld(j + 1);
    }
    // S = Q K^T (not swapped): lane holds key `col`, rows rr0 + crow(r, hi); dP = dO V^T
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    first_product_pair<DP>(s, (lbase + stage) + first, kf, dp, (lbase + stage + G::TILE) + first, vf);
    // rows rr0 + crow(r, hi) of L and D: four reads of four each, then the first reads of the second products (the counter of
    // pending LDS operations holds 15: the rings are requested in two halves around the wait for the slices)
    f32x4 lv[4], dvv[4];
    const uint32_t ldbase = lbase + lds_bytes<DP>() + (uint32_t)(j & 1) * 512 + hi * 16;
    static_for<4>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      lv[g] = rd128<32 * g>(ldbase);
      dvv[g] = rd128<256 + 32 * g>(ldbase);
    });
    SecondRing<DP> gring, qring;
    second_prefetch_pair<DP, 0>(gring, (lbase + stage + G::TILE) + second, qring, (lbase + stage) + second);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lv[0]), "+v"(lv[1]), "+v"(lv[2]), "+v"(lv[3]), "+v"(dvv[0]), "+v"(dvv[1]), "+v"(dvv[2]), "+v"(dvv[3]));
    second_prefetch_pair<DP, 1>(gring, (lbase + stage + G::TILE) + second, qring, (lbase + stage) + second);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float Lr = lv[r >> 2][r & 3];
      const float Dr = dvv[r >> 2][r & 3];
      float p = fast_exp2(s[r] * scale2 - Lr);
      if (causal && col > rr0 + crow(r, hi) + coff) p = 0.f;   // masked: P = 0
      s[r] = p * (dp[r] * scale - Dr);
      dp[r] = p;
    }
    // dV^T += dO^T P ; dK^T += Q^T dS   (row index permuted; padded rows of Q / dO are zero)
    // (the two rings were requested dO first: the pair below waits for them in that order)
    second_product_pair<DP>(dv, (lbase + stage + G::TILE) + second, dp, dk, (lbase + stage) + second, s, gring, qring);
  }
  store_rows<DP>(dv, a.op[SLOT_dV], head, batch, col, C, hi, D, 1.f);
  store_rows<DP>(dk, a.op[SLOT_dK], head, batch, col, C, hi, D, 1.f);
}

// ---- host side: does a launch of the general kernel's variant go to these kernels?  (the general kernels' launchers ask)
inline bool rows_ok(const OperandView &v, uint32_t rows) {
  return v.ptr && v.precision == PREC_FP32 && !v.transposed &&
         ((reinterpret_cast<uintptr_t>(v.ptr) | (uint64_t)v.ld * 4 | (uint64_t)v.headStride * 4 | (uint64_t)v.batchStride * 4) & 15) == 0 &&
         (uint64_t)rows * (uint64_t)v.ld * 4 < 0xFFFFFF00ull;   // one bounds-checked resource per (head, batch)
}
inline bool vector_ok(const OperandView &v) { return v.ptr && v.precision == PREC_FP32; }
// type: 0 forward, 1 backwardQuery, 2 backwardKeyValue (mfa.h); DP: the variant's head block
inline bool serves(int type, int DP, const KernelArgs &a) {
  if ((DP != 64 && DP != 128) || a.mask || (a.D & 3) || a.D > (uint32_t)DP || a.R == 0 || a.C == 0) return false;
  if (!rows_ok(a.op[SLOT_Q], a.R) || !rows_ok(a.op[SLOT_K], a.C) || !rows_ok(a.op[SLOT_V], a.C) || !vector_ok(a.op[SLOT_L])) return false;
  if (type == 0) return rows_ok(a.op[SLOT_O], a.R);
  if (!rows_ok(a.op[SLOT_dO], a.R) || !vector_ok(a.op[SLOT_D])) return false;
  if (type == 1) return rows_ok(a.op[SLOT_O], a.R) && rows_ok(a.op[SLOT_dQ], a.R);
  return rows_ok(a.op[SLOT_dK], a.C) && rows_ok(a.op[SLOT_dV], a.C);
}

}  // namespace f32k
}  // namespace mfa
