// attn_fwd16_v2.hip -- instantiations of the software-pipelined 16-bit-MFMA forward kernel.
#include "attn_fwd16_v2.h"
#include "launchers.h"

namespace mfa {

template <typename T, int D, int NW, int RB, int THR, bool MSUM>
static void launch_v2(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_v2<T, D, NW, RB, THR, MSUM>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64),
                     (fwd16v2_lds_bytes<D, NW, RB>()), stream, args, g);
}

template <typename T, int D, int NW, int RB, int THR, bool MSUM>
static void fill(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16_v2<T, D, NW, RB, THR, MSUM>);
  v->name = name;
  v->parallelization = NW * RB * 32;
  v->traversal = 64;
  v->headBlock = D;
  v->threads = NW * 64;
  v->ldsBytes = fwd16v2_lds_bytes<D, NW, RB>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_v2<T, D, NW, RB, THR, MSUM>;
}

// impl: 0 = default (8 waves x 32 rows, deferred rescale THR=8, row sum on the matrix pipe),
//       1 = THR=0 (the reference's rescale rule), 2 = row sum on the VALU, 3 = 4 waves x 64 rows
bool fwd16_v2_variant(int precision, int D, int impl, VariantInfo *out) {
  if (precision == PREC_BF16) {
    if (D == 128 && impl == 0) { fill<__bf16, 128, 8, 1, 8, true>(out, "attn_fwd16v2_bf16_d128_w8x32_thr8_msum"); return true; }
    if (D == 128 && impl == 1) { fill<__bf16, 128, 8, 1, 0, true>(out, "attn_fwd16v2_bf16_d128_w8x32_thr0_msum"); return true; }
    if (D == 128 && impl == 2) { fill<__bf16, 128, 8, 1, 8, false>(out, "attn_fwd16v2_bf16_d128_w8x32_thr8_vsum"); return true; }
    if (D == 128 && impl == 3) { fill<__bf16, 128, 4, 2, 8, true>(out, "attn_fwd16v2_bf16_d128_w4x64_thr8_msum"); return true; }
    if (D == 64 && impl == 0) { fill<__bf16, 64, 8, 1, 8, true>(out, "attn_fwd16v2_bf16_d64_w8x32_thr8_msum"); return true; }
    if (D == 64 && impl == 2) { fill<__bf16, 64, 8, 1, 8, false>(out, "attn_fwd16v2_bf16_d64_w8x32_thr8_vsum"); return true; }
    if (D == 64 && impl == 3) { fill<__bf16, 64, 4, 2, 8, true>(out, "attn_fwd16v2_bf16_d64_w4x64_thr8_msum"); return true; }
  }
  if (precision == PREC_FP16) {
    if (D == 128 && impl == 0) { fill<_Float16, 128, 8, 1, 8, true>(out, "attn_fwd16v2_f16_d128_w8x32_thr8_msum"); return true; }
    if (D == 64 && impl == 0) { fill<_Float16, 64, 8, 1, 8, true>(out, "attn_fwd16v2_f16_d64_w8x32_thr8_msum"); return true; }
  }
  return false;
}

} // namespace mfa
