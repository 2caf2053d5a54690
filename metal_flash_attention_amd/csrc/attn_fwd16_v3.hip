// attn_fwd16_v3.hip -- instantiations of the one-wave-per-SIMD, 64-rows-per-wave forward kernel.
#include "attn_fwd16_v3_launch.h"

namespace mfa {

// impl 0: product schedule.
//   D = 128: 8 waves x 32 rows, 3-stage ring filled by LDS-DMA (buffer_load_dwordx4 ... lds, no staging registers, no
//     ds_write_b128), LDS reads through inline asm with counted waits, V^T fragments in groups of four MFMAs: +3 %
//     over the register-staged schedule (impl 7), bit-identical results; the causal, column-parallel and block-sparse
//     siblings use the same schedule (block-sparse: the DMA prologue restarts per run of active tiles).
//   D = 64 / 32: register-staged, fragment reads left to hipcc (two waves per SIMD hide the LDS latency; the
//     LDS-DMA schedule measured -8 % at D = 64, which is VALU-bound).
//   D = 256: 4 waves x 32 rows (one per SIMD, 512 registers), 2-stage ring, K fragments hoisted (+11 %), V^T
//     fragments double-buffered in groups of four MFMAs and the staging writes of the next tile spread between those
//     groups instead of one burst behind the barrier (+11 %: 64 KiB of ds_write_b128 per tile kept the LDS busy while
//     no wave had work).
// Developer schedules (MFA_FWD16_IMPL=v3:<n>): 1 / 2 = K / K + first V fragments hoisted (D = 256: 2 = no grouped
// reads / spread writes, 3 = grouped reads only); D = 128: 3 / 4 / 5 = grouped reads / + spread writes / writes in
// the middle of step A (all within +-1 %), 7 = register-staged previous product, 8 at D = 64 / 256 = LDS-DMA
// (-8 % / -1.7 %; D = 256 with three K and two V images and one barrier per tile); 41 = K rows
// padded instead of swizzled; 11, 12, 14, 50-52 = timing-only ablations (WRONG RESULTS) behind DESIGN.md 4.2.
// Schedules that were measured and removed (numbers in DESIGN.md 4.2, profiles/ab*.txt): 4 waves x 64 rows with
// asm-placed QK MFMAs, row sum on the matrix pipe, split QK accumulator, sched_group_barrier interleave, static wave
// priority, packed-VALU softmax, K fragments requested one step ahead on top of LDS-DMA (+0.5 %); at D = 64, 64
// rows per wave with two waves per SIMD (+-0).
bool fwd16_v3_variant(int precision, int D, int impl, VariantInfo *out) {
  if (precision == PREC_BF16) {
    if (D == 128 && impl == 0) { fill_with_split<__bf16, 128, 8, 1, 8, 1, 3, 36>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_ldsdma"); return true; }
#ifdef MFA_DEV_VARIANTS
    if (D == 128 && impl == 7) { fill<__bf16, 128, 8, 1, 8, 0>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8"); return true; }
    if (D == 128 && impl == 1) { fill<__bf16, 128, 8, 1, 8, 1>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_prek"); return true; }
    if (D == 128 && impl == 2) { fill<__bf16, 128, 8, 1, 8, 2>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_prekv"); return true; }
    if (D == 128 && impl == 41) { fill<__bf16, 128, 8, 1, 8, 0, 0, 3, 2>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_kpad"); return true; }
    if (D == 128 && impl == 11) { fill<__bf16, 128, 8, 1, 8, 0, 2>(out, "ablate_no_exp_WRONG_RESULTS"); return true; }
    if (D == 128 && impl == 12) { fill<__bf16, 128, 8, 1, 8, 0, 3>(out, "ablate_one_k_fragment_WRONG_RESULTS"); return true; }
    if (D == 128 && impl == 14) { fill<__bf16, 128, 8, 1, 8, 0, 8>(out, "ablate_no_tile_barrier_WRONG_RESULTS"); return true; }
    if (D == 128 && impl == 50) { fill<__bf16, 128, 8, 1, 8, 0, 20>(out, "ablate_no_lds_reads_WRONG_RESULTS"); return true; }
    if (D == 128 && impl == 51) { fill<__bf16, 128, 8, 1, 8, 0, 21>(out, "ablate_no_softmax_WRONG_RESULTS"); return true; }
    if (D == 128 && impl == 52) { fill<__bf16, 128, 8, 1, 8, 0, 22>(out, "ablate_no_lds_reads_no_softmax_WRONG_RESULTS"); return true; }
#endif
    if (D == 64 && impl == 0) { fill_with_split<__bf16, 64, 8, 1, 8, 0>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8"); return true; }
#ifdef MFA_DEV_VARIANTS
    if (D == 64 && impl == 2) { fill<__bf16, 64, 8, 1, 8, 2>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8_prekv"); return true; }
    if (D == 64 && impl == 41) { fill<__bf16, 64, 8, 1, 8, 0, 0, 3, 2>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8_kpad"); return true; }
#endif
    if (D == 32 && impl == 0) { fill_with_split<__bf16, 32, 4, 1, 8, 0>(out, "attn_fwd16v3_bf16_d32_w4x32_thr8"); return true; }
    if (D == 256 && impl == 0) { fill_with_split<__bf16, 256, 4, 1, 8, 1, 2, 12>(out, "attn_fwd16v3_bf16_d256_w4x32_thr8_ring2_spread"); return true; }
#ifdef MFA_DEV_VARIANTS
    if (D == 256 && impl == 8) { fill<__bf16, 256, 4, 1, 8, 1, 0, 2, 36>(out, "attn_fwd16v3_bf16_d256_w4x32_thr8_ldsdma_k3v2"); return true; }
    if (D == 256 && impl == 2) { fill<__bf16, 256, 4, 1, 8, 1, 0, 2>(out, "attn_fwd16v3_bf16_d256_w4x32_thr8_ring2_prek"); return true; }
    if (D == 256 && impl == 1) { fill<__bf16, 256, 4, 1, 8, 0, 0, 2>(out, "attn_fwd16v3_bf16_d256_w4x32_thr8_ring2"); return true; }
    if (D == 256 && impl == 3) { fill<__bf16, 256, 4, 1, 8, 1, 0, 2, 4>(out, "attn_fwd16v3_bf16_d256_w4x32_thr8_ring2_prek_vpipe"); return true; }
    if (D == 128 && impl == 4) { fill<__bf16, 128, 8, 1, 8, 1, 0, 3, 12>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_prek_vpipe_wspread"); return true; }
    if (D == 128 && impl == 5) { fill<__bf16, 128, 8, 1, 8, 0, 0, 3, 8>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_wmid"); return true; }
    if (D == 64 && impl == 6) { fill<__bf16, 64, 8, 1, 8, 0, 0, 3, 64>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8_vsplit"); return true; }
    if (D == 64 && impl == 50) { fill<__bf16, 64, 8, 1, 8, 0, 20>(out, "ablate_no_lds_reads_WRONG_RESULTS"); return true; }
    if (D == 64 && impl == 52) { fill<__bf16, 64, 8, 1, 8, 0, 22>(out, "ablate_no_lds_reads_no_softmax_WRONG_RESULTS"); return true; }
    if (D == 64 && impl == 14) { fill<__bf16, 64, 8, 1, 8, 0, 8>(out, "ablate_no_tile_barrier_WRONG_RESULTS"); return true; }
    if (D == 64 && impl == 53) { fill<__bf16, 64, 8, 1, 8, 0, 23>(out, "ablate_no_row_maximum_WRONG_RESULTS"); return true; }
    if (D == 64 && impl == 51) { fill<__bf16, 64, 8, 1, 8, 0, 21>(out, "ablate_no_softmax_WRONG_RESULTS"); return true; }
    if (D == 64 && impl == 11) { fill<__bf16, 64, 8, 1, 8, 0, 2>(out, "ablate_no_exp_WRONG_RESULTS"); return true; }
    if (D == 64 && impl == 5) { fill<__bf16, 64, 8, 1, 8, 0, 0, 3, 8>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8_wmid"); return true; }
    if (D == 64 && impl == 8) { fill<__bf16, 64, 8, 1, 8, 1, 0, 3, 36>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8_ldsdma"); return true; }
    if (D == 128 && impl == 3) { fill<__bf16, 128, 8, 1, 8, 1, 0, 3, 4>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_prek_vpipe"); return true; }
#endif
  }
  if (precision == PREC_FP16) {
    if (D == 128 && impl == 0) { fill_with_split<_Float16, 128, 8, 1, 8, 1, 3, 36>(out, "attn_fwd16v3_f16_d128_w8x32_thr8_ldsdma"); return true; }
    if (D == 64 && impl == 0) { fill_with_split<_Float16, 64, 8, 1, 8, 0>(out, "attn_fwd16v3_f16_d64_w8x32_thr8"); return true; }
    if (D == 32 && impl == 0) { fill_with_split<_Float16, 32, 4, 1, 8, 0>(out, "attn_fwd16v3_f16_d32_w4x32_thr8"); return true; }
    if (D == 256 && impl == 0) { fill_with_split<_Float16, 256, 4, 1, 8, 1, 2, 12>(out, "attn_fwd16v3_f16_d256_w4x32_thr8_ring2_spread"); return true; }
  }
  return false;
}

// dense launches of the D = 64 bucket that the persistent kernel (attn_fwd16_p6.hip) does not serve
void fwd16_v3_d64_launch(int precision, dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if (precision == PREC_BF16) launch_v3<__bf16, 64, 8, 1, 8, 0>(grid, stream, args);
  else launch_v3<_Float16, 64, 8, 1, 8, 0>(grid, stream, args);
}

void fwd16_v3_d64_launch_split(int precision, dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) {
  if (precision == PREC_BF16) launch_v3_split<__bf16, 64, 8, 1, 8, 0>(grid, splits, wsO, wsML, stream, args);
  else launch_v3_split<_Float16, 64, 8, 1, 8, 0>(grid, splits, wsO, wsML, stream, args);
}

void fwd16_v3_d64_launch_causal(int precision, dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if (precision == PREC_BF16) launch_v3_causal<__bf16, 64, 8, 1, 8, 0, 3, 0>(grid, stream, args);
  else launch_v3_causal<_Float16, 64, 8, 1, 8, 0, 3, 0>(grid, stream, args);
}

} // namespace mfa
