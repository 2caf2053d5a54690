// attn_dkv16_wide.h -- backwardKeyValue for 16-bit inputs at 256 < D <= 384 on the 16-bit matrix cores (round 6).
//
// Until round 6 these launches ran fp32 arithmetic on 16-bit storage (attn_generic_dkv: 1/16 of the matrix rate).  The math, the
// fragment maps and the role split are attn_dkv16_rs.h's (a block of 32 keys goes to a PAIR of waves: the V-wave computes
// S = Q K^T -> P -> dV^T += dO^T P, the K-wave dP = dO V^T -> dS = P (dP*scale - D) -> dK^T += Q^T dS; one cached operand and one
// accumulator per wave: D / 4 + D / 2 of a lane's 512 registers, a wave per SIMD).  What does not carry over is that kernel's ring of
// FOUR {Q | dO} stages (three row blocks live in its software pipeline): 4 x 2 x 32 x 384 x 2 bytes = 192 KiB.  Here:
//   * TWO stages; row block t + 1 waits in the staging registers while block t is worked on and is written behind the step's last
//     barrier (it replaces t - 1);
//   * no pipeline across steps: first product, arithmetic, second product of the SAME row block, and the K-wave picks P up in the same
//     step -- two barriers per step (tile ready / P ready).  Each wave has a SIMD's matrix pipe to itself, so what a software pipeline
//     would buy is the softmax arithmetic (~300 of a step's ~2000 clocks), not a second wave's matrix work;
//   * the epilogue's staging rows one role at a time (2 x 32 x (D + 4) floats = 97 KiB).
// Two pairs = 64 keys per workgroup.  Dense, causal and per-batch lengths; block masks, traversal-parallel pieces and transposed
// operands keep the general kernel.  LDS-bound like attn_fwd16_wide (every fragment read feeds ONE matrix instruction).
// Reference: the `| 384 | ... |` rows of the mixed backwardKeyValue table (AttentionDescriptor+Parameters.swift:185-201),
// +Source.swift:244-293, +Softmax.swift:406-427.
#pragma once
#include "attn_bwd16.h"
#include <type_traits>

namespace mfa {

template <int D> constexpr int dkv16w_lds_bytes() {
  constexpr int ring = 2 * (2 * 32 * D * 2 + 256) + 2 * 4096;
  constexpr int epi = 2 * 32 * (D + 4) * 4;
  return ring > epi ? ring : epi;
}

template <typename T, int D, typename TG = T, bool CAUSAL = false>
__global__ __launch_bounds__(256) void attn_dkv16_wide(const KernelArgs a, const Fwd16Grid grid) {
  typedef Frag16<T> F;
  typedef typename F::v8 v8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPAIR = 2, WGCOLS = NPAIR * 32;
  constexpr int BR = 32, NT = 256, NDB = D / 32, NKS = D / 16;
  constexpr int TILE = BR * D * 2, STAGE = 2 * TILE + 256, XBUF = 2 * STAGE;
  constexpr int CPR = D / 8, NCHUNK = BR * CPR, SCH = NCHUNK / NT;
  static_assert(NCHUNK % NT == 0 && D % 64 == 0, "a tile must divide evenly over the workgroup (pairs of d-blocks)");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave % NPAIR, role = wave / NPAIR;   // role 0: V-wave (dV), role 1: K-wave (dK)
  const int lane = tid & 63, kc = lane & 31, hi = lane >> 5;
  uint32_t cblk, head, batch;
  fwd16_decode_block(grid, blockIdx.x, &cblk, &head, &batch);
  int R = a.R, C = a.C;
  const int Dr = a.D;
  batch_lengths(a, batch, R, C);
  if ((int64_t)cblk * WGCOLS >= C) return;   // padded batch entry: the whole workgroup lies beyond its keys
  const int64_t c0 = (int64_t)cblk * WGCOLS + pair * 32;
  const int64_t col = c0 + kc;
  const uint32_t ldq2 = (uint32_t)a.op[SLOT_Q].ld * 2, ldg2 = (uint32_t)a.op[SLOT_dO].ld * 2;
  constexpr uint32_t OOB = 0xFFFFFF00u;
  const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_Q], head, batch), 0, (uint32_t)R * ldq2, 0x00020000);
  const __amdgpu_buffer_rsrc_t gres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[SLOT_dO], head, batch), 0, (uint32_t)R * ldg2, 0x00020000);

  // ---- cached operand of this wave's first product: K (V-wave) or V (K-wave) fragments, B operands (+Caching.swift:316-346)
  v8 cf[NKS];
  {
    const int slot = role ? SLOT_V : SLOT_K;
    const uint32_t ld2 = (uint32_t)a.op[slot].ld * 2;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[slot], head, batch), 0, (uint32_t)C * ld2, 0x00020000);
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int d0 = 16 * s + 8 * hi;
      const bool ok = d0 < Dr && col < C;
      cf[s] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(res, ok ? (uint32_t)col * ld2 + d0 * 2 : OOB, 0, 0));
    }
  }

  // CAUSAL (extension): the traversal starts at the first row block that sees the workgroup's first key
  const int coff = causal_offset(R, C);
  const int block0 = CAUSAL ? (int)(max((int64_t)0, (int64_t)cblk * WGCOLS - coff) / 32) : 0;
  const int nblocks = (R + 31) / 32 - block0;

  // ---- Q / dO staging (registers, one row block ahead of the LDS) + the L / D slices of the row block
  uint32_t qoff[SCH], goff[SCH], wlds[SCH];
  // (sixteen consecutive lanes = four rows x the four chunks of ONE d-block: sixteen different 16-byte slots of the image -- with
  // consecutive chunks of one row on consecutive lanes, as in attn_dkv16_rs.h, they fall on four slots, the d-blocks being 2048 bytes
  // apart: 22 % of this kernel's LDS cycles were bank conflicts, profiles/r06_fwdbwd_bf16_d384_mixed_summary_before_lds_fix.txt.
  // A wave = eight rows x a PAIR of d-blocks: 128 contiguous bytes of a row per load)
#pragma unroll
  for (int i = 0; i < SCH; ++i) {
    const int id = tid + i * NT;
    const int l = id & 63, unit = id >> 6, dbp = unit % (NDB / 2), rg = unit / (NDB / 2);
    const int srow = 8 * rg + 4 * (l >> 5) + ((l >> 2) & 3), sc = 4 * (2 * dbp + ((l >> 4) & 1)) + (l & 3);
    const bool svalid = sc * 8 < Dr;
    qoff[i] = svalid ? __builtin_elementwise_add_sat((uint32_t)(srow * ldq2 + sc * 16), (uint32_t)block0 * BR * ldq2) : OOB;
    goff[i] = svalid ? __builtin_elementwise_add_sat((uint32_t)(srow * ldg2 + sc * 16), (uint32_t)block0 * BR * ldg2) : OOB;
    wlds[i] = ((sc >> 2) * BR + srow) * 64 + (((sc & 3) ^ ((srow >> 2) & 3)) * 16);   // [D/32][32 rows][32 elements], chunks swizzled; Q at +0, dO at +TILE
  }
  const uint32_t qinc = BR * ldq2, ginc = BR * ldg2;
  u32x4 qreg[SCH], greg[SCH];
  uint16_t ldlo = 0, ldhi = 0;   // L (wave 0) / D (wave 1) of the block in the staging registers, as loaded (attn_dkv16_rs.h)
  const bool ldloader = wave < 2 && lane < 32;
  const int ldslot = wave == 0 ? SLOT_L : SLOT_D;
  const int ldprec = a.op[ldslot].precision;
  const uint32_t ldesz = ldprec == PREC_FP32 ? 4u : 2u;
  const __amdgpu_buffer_rsrc_t ldres = __builtin_amdgcn_make_buffer_rsrc(operand_base(a.op[ldslot], head, batch), 0, (uint32_t)R * ldesz, 0x00020000);
  uint32_t ldoff = (uint32_t)(block0 * BR + lane) * ldesz;
  auto issue_loads = [&]() {
#pragma unroll
    for (int i = 0; i < SCH; ++i) {
      qreg[i] = __builtin_amdgcn_raw_buffer_load_b128(qres, qoff[i], 0, 0);
      greg[i] = __builtin_amdgcn_raw_buffer_load_b128(gres, goff[i], 0, 0);
      qoff[i] = __builtin_elementwise_add_sat(qoff[i], qinc);
      goff[i] = __builtin_elementwise_add_sat(goff[i], ginc);
    }
    if (ldloader) {
      ldlo = __builtin_amdgcn_raw_buffer_load_b16(ldres, ldoff, 0, 0);
      ldhi = __builtin_amdgcn_raw_buffer_load_b16(ldres, ldoff + (ldesz - 2u), 0, 0);
      ldoff += BR * ldesz;
    }
  };
  auto write_tiles = [&](int stage) {
    char *base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < SCH; ++i) {
      *reinterpret_cast<u32x4 *>(base + wlds[i]) = qreg[i];
      *reinterpret_cast<u32x4 *>(base + TILE + wlds[i]) = __builtin_bit_cast(u32x4, convert_chunk<T, TG>(greg[i]));
    }
    if (ldloader) {
      const float ldval = ldprec == PREC_FP32 ? __builtin_bit_cast(float, (uint32_t)ldlo | ((uint32_t)ldhi << 16))
                        : ldprec == PREC_FP16 ? (float)__builtin_bit_cast(_Float16, ldlo) : bf16_bits_to_f32(ldlo);
      reinterpret_cast<float *>(base + 2 * TILE)[wave * 32 + lane] = ldval;
    }
  };

  // row fragment of k-step t (this lane: row kc, elements 16 t + 8 hi .. + 7) and the transposing reads (attn_dkv16_rs.h)
  const int fr0 = kc * 64 + ((hi ^ ((kc >> 2) & 3)) * 16), fr1 = kc * 64 + (((2 + hi) ^ ((kc >> 2) & 3)) * 16);   // t even / odd, + (t >> 1) * BR * 64
  const int n16 = lane & 15;
  const int trow = (n16 >> 2) + 4 * hi, tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1), thalf = (n16 & 3) & 1;
  const int tr0 = trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8;
  const int tr1 = (trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8;
  char *xp = smem + XBUF + pair * 4096 + lane * 16;   // the pair's P tile: [4 register groups][64 lanes x 16 bytes]

  f32x16 acc[NDB];   // dV^T (V-wave) or dK^T (K-wave): lane = key, registers = head-dimension rows
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

  auto run = [&](auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr int img_row = ROLE ? TILE : 0;     // first product:  K-wave dO rows, V-wave Q rows
    constexpr int img_tr = ROLE ? 0 : TILE;      // second product: K-wave Q^T,     V-wave dO^T
    for (int t = 0; t < nblocks; ++t) {
      const char *st = smem + (t & 1) * STAGE;
      if (t + 1 < nblocks) {      // row block t + 1 (staging registers) replaces t - 1, whose last readers passed the barrier below
        write_tiles((t + 1) & 1);
        issue_loads();
      }
      // ---- first product: S = Q K^T (V-wave) / dP = dO V^T (K-wave); the row fragment of step s + 1 is requested before the
      // matrix instruction of step s
      f32x16 x;
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = 0.f;
      v8 af[2];
      af[0] = *reinterpret_cast<const v8 *>(st + img_row + fr0);
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        if (s + 1 < NKS) {
          af[(s + 1) & 1] = *reinterpret_cast<const v8 *>(st + img_row + ((s + 1) >> 1) * BR * 64 + (((s + 1) & 1) ? fr1 : fr0));
          __builtin_amdgcn_sched_barrier(0x406);
        }
        x = F::mfma(af[s & 1], cf[s], x);
      }
      // ---- P = exp2(S*scale2 - L) (+Softmax.swift:409-417), fp32 copy for the partner; dS = P (dP*scale - D) (+Softmax.swift:419-427)
      const float *Ls = reinterpret_cast<const float *>(st + 2 * TILE) + (ROLE ? 32 : 0) + 4 * hi;
      v8 frag[2];
      if constexpr (ROLE == 0) {
        const int row0 = (block0 + t) * BR;
        const bool diag = CAUSAL && (c0 + 31 > row0 + coff);   // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = *reinterpret_cast<const f32x4 *>(Ls + 8 * g);
          f32x4 p4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            float p = fast_exp2(x[r] * a.scale2 - l4[i]);
            if (CAUSAL && diag && col > row0 + crow(r, hi) + coff) p = 0.f;
            p4[i] = p;
            frag[g >> 1][4 * (g & 1) + i] = (T)p;
          }
          *reinterpret_cast<f32x4 *>(xp + g * 1024) = p4;
        }
      }
      __syncthreads();   // P of this row block is in the exchange buffer
      if constexpr (ROLE == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 d4 = *reinterpret_cast<const f32x4 *>(Ls + 8 * g);
          const f32x4 p4 = *reinterpret_cast<const f32x4 *>(xp + g * 1024);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            frag[g >> 1][4 * (g & 1) + i] = (T)(p4[i] * (x[r] * a.scale - d4[i]));
          }
        }
      }
      // ---- second product: dV^T += dO^T P / dK^T += Q^T dS (rows 16 u .. of d-block db, transposed by the LDS read)
      auto read_tr = [&](int idx) -> v8 {   // idx = u * NDB + db
        const char *p = st + img_tr + (16 * (idx / NDB)) * 64 + (idx % NDB) * BR * 64;
        const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + tr0));
        const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + tr1));
        return __builtin_bit_cast(v8, __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      v8 tf[2];
      tf[0] = read_tr(0);
#pragma unroll
      for (int idx = 0; idx < 2 * NDB; ++idx) {
        if (idx + 1 < 2 * NDB) {
          tf[(idx + 1) & 1] = read_tr(idx + 1);
          __builtin_amdgcn_sched_barrier(0x406);
        }
        acc[idx % NDB] = F::mfma(tf[idx & 1], frag[idx / NDB], acc[idx % NDB]);
      }
      __syncthreads();   // every wave is done with this stage (and with P); row block t + 1 is in the other one
    }
  };
  if (nblocks > 0) {
    issue_loads();
    write_tiles(0);
    if (nblocks > 1) issue_loads();   // row block 1 waits in the staging registers
    __syncthreads();
    if (role == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
  }

  // ---- epilogue: this wave's accumulator through LDS (whole-row stores), one role at a time: dV (V-waves), then dK (K-waves)
  constexpr int OLD = D + 4;
  float *Os = reinterpret_cast<float *>(smem) + pair * (32 * OLD);
  float *orow = Os + kc * OLD;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (role == pass) {
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4 *>(orow + 32 * db + 8 * g + 4 * hi) =
              make_float4(acc[db][4 * g], acc[db][4 * g + 1], acc[db][4 * g + 2], acc[db][4 * g + 3]);
      const int slot = role ? SLOT_dK : SLOT_dV;
      store_block_rows<T, D>(Os, operand_base(a.op[slot], head, batch), a.op[slot].precision, (uint32_t)a.op[slot].ld, c0, C, Dr, lane);
    }
  }
}

} // namespace mfa
