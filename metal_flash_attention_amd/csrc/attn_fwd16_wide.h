// attn_fwd16_wide.h -- forward attention on the 16-bit matrix cores for 256 < D <= 384 (head blocks 320 and 384), round 6.
//
//   reference: the `| 384 | ... |` rows of the mixed-precision parameter tables (AttentionDescriptor+Parameters.swift:113, :120) and the
//   D-blocked accumulate loop they drive (AttentionKernel+Accumulate.swift:403-469: what does not fit the register file is paged);
//   loop structure = loopForward, AttentionKernel+Source.swift:158-200, softmax = +Softmax.swift:228-324.
//
// Until round 5 a 16-bit problem with D > 256 ran fp32 arithmetic on 16-bit storage (attn_generic.h: 1/16 of the bf16 matrix rate).
// What the gfx950 register budget allows at this head dimension, re-derived:
//   * O^T of 64 rows x 384 would take 384 of a lane's 512 registers and Q' another 192: a wave owns 32 ROWS here (O^T 192 registers,
//     Q' 96), four waves = 128 rows per workgroup, one wave per SIMD.  The accumulators and the cached left-hand operand stay in
//     registers (nothing is paged through memory: at D <= 384 "what gets evicted first" is the second row block of a wave);
//   * the price: every K row fragment and every V^T fragment read from LDS feeds ONE matrix instruction (1 KiB of LDS reads per
//     matrix instruction, four waves: the LDS pipe's 128 bytes per clock are exactly the matrix pipe's appetite) -- the kernel is
//     bound by LDS bandwidth at about half the matrix rate, which is 8 x what the fp32 path delivered;
//   * 32-key steps, ring of three {K | V} stages (48 KiB each at 384) filled through registers (global -> VGPR -> ds_write_b128),
//     one barrier per step; S^T of step j + 1 is multiplied while the vector ALU exponentiates step j (software pipeline in
//     source order, the interleaving is hipcc's);
//   * same fragment maps as the other 16-bit kernels: S^T = K Q'^T "swapped", so that the accumulator layout of the first product
//     is the B operand of the second; K image rows of D x 2 bytes with their 16-byte chunks XOR-swizzled by (key & 7), V image
//     [D / 32][32 keys][64 bytes] read by ds_read_b64_tr_b16 pairs.
// Dense, causal (CAUSAL code object) and per-batch lengths; row-major operands with 16-byte aligned rows (the host checks);
// transposed operands and block masks keep the general kernel.  Exact-scale arithmetic (s * scale2 - m in fp32) in both
// precision modes; L in the descriptor's storage type.
#pragma once
#include "attn_fwd16_common.h"
#include <type_traits>

namespace mfa {
namespace wide {
constexpr int BK = 32, ROWS = 128, RING = 3, THR = 8;
template <int DP> constexpr int lds_bytes() { return RING * (BK * (DP * 2 + 16) + BK * DP * 2); }   // (K rows padded by one chunk, see the kernel)
// two second-product matrix instructions on accumulators that LIVE in the accumulation registers (one asm statement per pair:
// hipcc puts a wait state between asm statements; the leading s_nop covers a freshly packed P^T fragment)
template <typename T, typename V8> __device__ __forceinline__ void pv_pair(f32x16 &o0, f32x16 &o1, const V8 &v0, const V8 &v1, const V8 &p) {
  if constexpr (__is_same(T, __bf16))
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %4, %1" : "+a"(o0), "+a"(o1) : "v"(v0), "v"(v1), "v"(p));
  else
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, %3, %4, %1" : "+a"(o0), "+a"(o1) : "v"(v0), "v"(v1), "v"(p));
}
// LDS fragment reads as asm volatile statements with a counted wait (LDS returns in order; asm volatile statements keep their
// order): rings of four fragments.  Plain loads let hipcc request all 24 K row fragments of a step at once -- the live ranges that
// pushed Q' and the staging registers into the accumulation file and to scratch
template <int OFF> __device__ __forceinline__ u32x4 frag_read_b128(uint32_t addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
struct TrFrag { u32x2 lo, hi; };
template <int OFF> __device__ __forceinline__ TrFrag frag_read_tr16(uint32_t addr) {   // keys +0..3 and +8..11 of a 16-key group
  TrFrag r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.lo) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r.hi) : "v"(addr), "n"(OFF + 8 * 64));
  return r;
}
template <int N> __device__ __forceinline__ void frag_wait(u32x4 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N < 15 ? N : 15)); }
template <int N> __device__ __forceinline__ void frag_wait(TrFrag &a, TrFrag &b) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : "n"(N < 15 ? N : 15));
}
template <int N, typename Fn> __device__ __forceinline__ void unrolled(Fn &&f) {
  if constexpr (N > 0) {
    unrolled<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
// first-product matrix instruction whose B operand (a Q' fragment) lives in the accumulation registers: half of the Q' fragments are
// parked there beside O^T (192 + 48 of the 256), which leaves the architectural half room for the staging registers and the rings
template <typename T, typename V8> __device__ __forceinline__ void qk_acc_operand(f32x16 &s, const u32x4 &k, const V8 &qa) {
  if constexpr (__is_same(T, __bf16)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(qa));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(qa));
}
template <typename V8> __device__ __forceinline__ void park(V8 &dst, const V8 &src) {   // VGPR -> accumulation registers, dword by dword
  const u32x4 w = __builtin_bit_cast(u32x4, src);
  u32x4 d;
  asm volatile("v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
               : "=a"(d[0]), "=a"(d[1]), "=a"(d[2]), "=a"(d[3]) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
  dst = __builtin_bit_cast(V8, d);
}
// every issued matrix instruction has left the pipe (16 passes: 18 wait states before a VALU read of its result, gfx940 rules)
template <int N> __device__ __forceinline__ void acc_fence(f32x16 *o) {
#pragma unroll
  for (int i = 0; i < N; i += 2) asm volatile("s_nop 15\n\ts_nop 3" : "+a"(o[i]), "+a"(o[i + 1]));
}
template <int N> __device__ __forceinline__ void acc_written(f32x16 *o) {   // v_accvgpr_write -> matrix instruction SrcC
#pragma unroll
  for (int i = 0; i < N; i += 2) asm volatile("s_nop 4" : "+a"(o[i]), "+a"(o[i + 1]));
}
}  // namespace wide

template <typename T, int DP, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd16_wide(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS layout (round 6, after the counter pass of profiles/r06_fwdbwd_bf16_d384_mixed_summary_before_lds_fix.txt: a third of this
  // kernel's LDS cycles were bank conflicts): K rows of 640 / 768 bytes all start on the same banks (768 = 3 x 256; 640: two of them) and
  // an XOR of the low three chunk bits cannot spread SIXTEEN rows -- rows padded by one 16-byte chunk instead (41 / 49 chunks: sixteen
  // consecutive rows start on sixteen different slots), no swizzle, ONE read address per lane; the V image's staging writes go four rows x
  // the four chunks of one d-block per sixteen lanes (two d-blocks, 2048 bytes apart, shared a slot)
  constexpr int BK = wide::BK, RING = wide::RING, NKS = DP / 16, NDB = DP / 32, ROWB = DP * 2 + 16, CPR = DP / 8;
  constexpr int TILE = BK * ROWB, STAGE = TILE + BK * DP * 2, NCH = BK * CPR / 256;
  static_assert(NDB % 2 == 0 && BK * CPR % 256 == 0 && CPR % 8 == 0, "tile must divide evenly over the workgroup; the swizzle needs whole groups of eight chunks");
  constexpr uint32_t OOB = 0xFFFFFF00u;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, q = lane & 31, hi = lane >> 5;
  uint32_t rblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &rblk, &head, &batch);
  if constexpr (CAUSAL) rblk = grid.rowBlocks - 1 - rblk;   // later row blocks traverse more keys: start them first
  int R = a.R, C = a.C;
  const int Dr = a.D;
  batch_lengths(a, batch, R, C);
  if ((int64_t)rblk * wide::ROWS >= R) return;   // padded batch entry: the whole workgroup lies beyond its rows
  const int64_t r0 = (int64_t)rblk * wide::ROWS + wave * 32;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldk2 = (uint32_t)a.op[SLOT_K].ld * 2, ldv2 = (uint32_t)a.op[SLOT_V].ld * 2;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)R * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t kres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_K], head, batch), 0, (uint32_t)C * ldk2, 0x00020000);
  const __amdgpu_buffer_rsrc_t vres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_V], head, batch), 0, (uint32_t)C * ldv2, 0x00020000);

  // ---- Q fragments (B operand of S^T = K Q^T), cached in registers for the whole kernel
  v8 qf[NKS];
  {
    const uint32_t rowoff = (uint32_t)(r0 + q) * ldq2;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int d0 = 16 * s + 8 * hi;
      qf[s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(qres, (d0 < Dr && r0 + q < R) ? rowoff + d0 * 2 : OOB, 0, 0));
    }
  }

  constexpr int NQA = 0;   // (parking Q' fragments in the spare accumulation registers through asm operands made hipcc spill MORE: 59 against 49)

  // ---- key range: everything, or (causal: row r sees key c iff c <= r + coff) up to the step holding the last key of the block's last row
  const int coff = causal_offset(R, C);
  int nt = (C + BK - 1) / BK;
  if constexpr (CAUSAL) {
    const int64_t last_row = min((int64_t)R, ((int64_t)rblk + 1) * wide::ROWS) - 1;
    nt = (int)min((int64_t)nt, (last_row + coff) / BK + 1);
  }
  if (nt < 1) nt = 1;   // (an entry without keys: one fully masked step, O = 0)

  // ---- staging global -> VGPR -> LDS: thread -> key row tid / 8, 16-byte chunks (tid & 7) + 8 i of the step's K and V rows.  ONE
  // offset register per operand and side: chunk i lies 128 bytes further in memory and in the K image (the swizzle exchanges chunks
  // inside a group of eight) and one 32-key sub-tile pair (4096 bytes) further in the V image -- immediates, not registers (the first
  // version kept per-chunk offsets and read addresses: 125 spilled registers at the 384 head block)
  constexpr uint32_t SAT = 0xFFFFF000u;   // (past the end: stays out of range with the chunk immediates added, no 32-bit wrap)
  const int srow = tid >> 3, sc0 = tid & 7;
  // V: a wave = eight keys x a pair of d-blocks (128 contiguous bytes of a row per load), sixteen lanes = four keys x one d-block's chunks
  const int vrow = 8 * (tid >> 6) + 4 * ((tid >> 5) & 1) + ((tid >> 2) & 3), vc0 = 4 * ((tid >> 4) & 1) + (tid & 3);
  uint32_t koff = srow * ldk2 + sc0 * 16, voff = vrow * ldv2 + vc0 * 16;
  const uint32_t klds = srow * ROWB + (sc0 << 4);
  const uint32_t vlds = TILE + ((vc0 >> 2) * BK + vrow) * 64 + (vc0 & 3) * 16;
  const int nvalid = (Dr / 8 - sc0 + 7) / 8;   // chunks of this thread inside the head dimension (the others read zeros)
  const int nvalidv = (Dr / 8 - vc0 + 7) / 8;
  const uint32_t kinc = BK * ldk2, vinc = BK * ldv2;
  // ONE set of staging registers (NCH x 16 bytes per thread): the K chunks of tile j + 2 travel through it during the first half of
  // step j, the V chunks during the second (both operands at once were 48 registers the 384 head block does not have: 49 spilled)
  u32x4 sreg[NCH];
  auto load_k = [&]() {   // K chunks of the next tile in sequence (zeros past the end of the sequence: out-of-range offsets)
#pragma unroll
    for (int i = 0; i < NCH; ++i) sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(kres, i < nvalid ? koff + 128 * i : OOB, 0, 0);
    koff = min(koff + kinc, SAT);   // (slices are below 0xFF000000 bytes, include/mfa.h: the sum cannot wrap)
  };
  auto load_v = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(vres, i < nvalidv ? voff + 128 * i : OOB, 0, 0);
    voff = min(voff + vinc, SAT);
  };
  auto write_k = [&](auto ST_) {
    char *base = smem + decltype(ST_)::value * STAGE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4 *>(base + klds + 128 * i) = sreg[i];
  };
  auto write_v = [&](auto ST_) {
    char *base = smem + decltype(ST_)::value * STAGE;
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4 *>(base + vlds + 4096 * i) = sreg[i];
  };

  const int n16 = lane & 15;
  const uint32_t vtr0 = (uint32_t)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char *)smem + TILE + ((n16 >> 2) + 4 * hi) * 64 +
                       (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2;
  // K row fragment of k-step t: logical chunk 2 t + hi of row q; the swizzle touches the low three bits only -- four addresses (t & 3),
  // the group of eight chunks (t >> 2) an immediate
  const uint32_t lds0 = (uint32_t)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char *)smem;
  const uint32_t kread = lds0 + q * ROWB + (hi << 4);   // chunk 2 t + hi of row q: + 32 t

  // S^T of the 32 keys of stage ST: key = lane % 32, one K row fragment per 16 elements of the head dimension, ring of four
  auto qk = [&](auto ST_, f32x16 &s) {
    // (DS instruction offsets are 16 bits: the stage base goes into the four address registers, the k-step group is the immediate)
    const uint32_t ka = kread + decltype(ST_)::value * STAGE;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    u32x4 kr[4];
    wide::unrolled<4>([&](auto T_) { constexpr int t = decltype(T_)::value; kr[t] = wide::frag_read_b128<t * 32>(ka); });
    wide::unrolled<NKS>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      constexpr int pending = (NKS - 1 - t) < 3 ? (NKS - 1 - t) : 3;
      wide::frag_wait<pending>(kr[t & 3]);
      if constexpr (t < NQA) wide::qk_acc_operand<T>(s, kr[t & 3], qf[t]);   // (Q' fragments t < NQA are parked in the accumulation file)
      else s = F::mfma(__builtin_bit_cast(v8, kr[t & 3]), qf[t], s);
      if constexpr (t + 4 < NKS) kr[t & 3] = wide::frag_read_b128<(t + 4) * 32>(ka);
    });
  };
  // maskAttentionMatrixEdge (+Softmax.swift:228-260) and the causal limit, on the steps that need them (wave-uniform tests)
  auto mask = [&](f32x16 &s, int c0) {
    if (c0 + BK > C) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (c0 + crow(r, hi) >= C) s[r] = mask_value();
    }
    if constexpr (CAUSAL) {
      if (c0 + BK - 1 > r0 + coff) {
        const int64_t limit = r0 + q + coff;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (c0 + crow(r, hi) > limit) s[r] = mask_value();
      }
    }
  };

  f32x16 o[NDB];
  float m = -3.402823466e+38f, l = 0.f;   // +Caching.swift:310
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  wide::acc_written<NDB>(o);
  // O^T += V^T P^T for the 32 keys of stage ST: fragments (u, db) in order, ring of four, two matrix instructions per asm statement.  The
  // accumulators are pinned to the accumulation registers ("+a": left to itself hipcc moved them between the two halves of the register
  // file -- 432 v_accvgpr moves per step at the 384 head block)
  auto pv = [&](auto ST_, const v8 (&pf)[2]) {
    constexpr int VOFF = 0, NF = 2 * NDB;
    const uint32_t vtr = vtr0 + decltype(ST_)::value * STAGE;
    wide::TrFrag vr[4];
    wide::unrolled<4>([&](auto F_) { constexpr int f = decltype(F_)::value; vr[f] = wide::frag_read_tr16<VOFF + ((f % NDB) * BK + 16 * (f / NDB)) * 64>(vtr); });
    wide::unrolled<NDB>([&](auto P_) {
      constexpr int f = 2 * decltype(P_)::value, u = f / NDB, db = f % NDB;   // (NDB even: a pair never straddles the two key groups)
      constexpr int younger = (NF - 2 - f) < 2 ? (NF - 2 - f) : 2;
      wide::frag_wait<2 * younger>(vr[f & 3], vr[(f + 1) & 3]);
      const u32x4 w0 = {vr[f & 3].lo[0], vr[f & 3].lo[1], vr[f & 3].hi[0], vr[f & 3].hi[1]};
      const u32x4 w1 = {vr[(f + 1) & 3].lo[0], vr[(f + 1) & 3].lo[1], vr[(f + 1) & 3].hi[0], vr[(f + 1) & 3].hi[1]};
      wide::pv_pair<T>(o[db], o[db + 1], __builtin_bit_cast(v8, w0), __builtin_bit_cast(v8, w1), pf[u]);
      if constexpr (f + 4 < NF) vr[f & 3] = wide::frag_read_tr16<VOFF + (((f + 4) % NDB) * BK + 16 * ((f + 4) / NDB)) * 64>(vtr);
      if constexpr (f + 5 < NF) vr[(f + 1) & 3] = wide::frag_read_tr16<VOFF + (((f + 5) % NDB) * BK + 16 * ((f + 5) / NDB)) * 64>(vtr);
    });
  };

  // ---- prologue: tiles 0, 1 in LDS, tile 2 in registers, S^T of step 0
  load_k(); write_k(std::integral_constant<int, 0>{});
  load_v(); write_v(std::integral_constant<int, 0>{});
  load_k(); write_k(std::integral_constant<int, 1>{});
  load_v(); write_v(std::integral_constant<int, 1>{});
  load_k();   // K of tile 2 is in flight
  __syncthreads();

  // step j on stage ST (the stage index is a compile-time constant: every LDS offset of the step is an immediate)
  auto step = [&](auto ST_, int j) {
    constexpr int ST = decltype(ST_)::value, ST1 = (ST + 1) % RING, ST2 = (ST + 2) % RING;
    if (j > 0) __syncthreads();   // tile j + 1 (written during step j - 1) is complete; stage ST2 (tile j - 1) has no reader left
    write_k(std::integral_constant<int, ST2>{});   // K of tile j + 2 (requested at the end of step j - 1)
    load_v();                                      // V of tile j + 2: lands under the first product and the softmax
    f32x16 s_cur;
    qk(ST_, s_cur);
    mask(s_cur, j * BK);
    // onlineReduceMaximum / onlineCorrectO (deferred by THR, contract in include/mfa.h) / softmax + onlineReduceSum of step j
    float mx0 = fmaxf(s_cur[0], s_cur[1]), mx1 = fmaxf(s_cur[2], s_cur[3]);
#pragma unroll
    for (int r = 4; r < 16; r += 4) {
      mx0 = fmaxf(fmaxf(mx0, s_cur[r]), s_cur[r + 1]);
      mx1 = fmaxf(fmaxf(mx1, s_cur[r + 2]), s_cur[r + 3]);
    }
    const float m_new = half_swap_max(fmaxf(mx0, mx1)) * a.scale2;
    if (__builtin_amdgcn_ballot_w64(m_new > m + (float)wide::THR) != 0) {
      const float m_up = fmaxf(m, m_new);
      const float corr = fast_exp2(m - m_up);
      m = m_up;
      l *= corr;
      wide::acc_fence<NDB>(o);
      // one accumulator block at a time goes through the architectural registers and BACK (the empty statement pins it again): left
      // to itself hipcc read all 192 accumulator registers first -- the copies, not the loop, were what spilled
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= corr;
        asm volatile("" : "+a"(o[db]));
      }
      wide::acc_written<NDB>(o);
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(s_cur[r] * a.scale2 - m);
      s_cur[r] = p;
      ps[r & 3] += p;
    }
    l += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    v8 pf[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // matrix-instruction step u (16 keys) uses registers 8 u .. 8 u + 7
      v8 pk;
#pragma unroll
      for (int i = 0; i < 8; ++i) pk[i] = (T)s_cur[8 * u + i];
      pf[u] = pk;
    }
    write_v(std::integral_constant<int, ST2>{});
    load_k();                                      // K of tile j + 3: lands under the second product
    pv(ST_, pf);
  };
  for (int j = 0; j < nt; j += RING) {
    step(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nt) step(std::integral_constant<int, 1>{}, j + 1);
    if (j + 2 < nt) step(std::integral_constant<int, 2>{}, j + 2);
  }

  // ---- epilogue: O /= l (+Source.swift:165-171), L = m + log2(l) (+Caching.swift:373-377), straight from the registers:
  // register r of block db is element 32 db + crow(r, hi) of the lane's row, four consecutive elements per register group
  wide::acc_fence<NDB>(o);
  const float l_tot = half_swap_add(l) + 1.401298464e-45f;
  const float inv = l_tot > 1e-30f ? 1.0f / l_tot : 0.f;   // a row may see no key at all (empty batch entry)
  const int64_t row = r0 + q;
  const int po = a.op[SLOT_O].precision;
  const uint32_t osz = po == PREC_FP32 ? 4u : 2u, ldo = (uint32_t)a.op[SLOT_O].ld;
  const __amdgpu_buffer_rsrc_t ores = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_O], head, batch), 0, (uint32_t)R * ldo * osz, 0x00020000);
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 32 * db + 8 * g + 4 * hi;
      const uint32_t off = (d0 < Dr && row < R) ? ((uint32_t)row * ldo + (uint32_t)d0) * osz : OOB;
      const float v0 = o[db][4 * g] * inv, v1 = o[db][4 * g + 1] * inv, v2 = o[db][4 * g + 2] * inv, v3 = o[db][4 * g + 3] * inv;
      if (po == PREC_FP32) {
        const u32x4 w = {__builtin_bit_cast(uint32_t, v0), __builtin_bit_cast(uint32_t, v1), __builtin_bit_cast(uint32_t, v2), __builtin_bit_cast(uint32_t, v3)};
        __builtin_amdgcn_raw_buffer_store_b128(w, ores, off, 0, 0);
      } else {
        const u32x2 w = {pack16<T>(v0, v1), pack16<T>(v2, v3)};
        __builtin_amdgcn_raw_buffer_store_b64(w, ores, off, 0, 0);
      }
    }
  if (hi == 0 && row < R) store_elem(operand_base(a.op[SLOT_L], head, batch), row, a.op[SLOT_L].precision, m + log2f(l_tot));
}

} // namespace mfa
