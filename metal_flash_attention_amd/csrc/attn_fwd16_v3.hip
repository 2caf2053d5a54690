// attn_fwd16_v3.hip -- instantiations of the one-wave-per-SIMD, 64-rows-per-wave forward kernel.
#include "attn_fwd16_v3.h"
#include "launchers.h"

namespace mfa {

template <typename T, int D, int NW, int RB, int THR, int PRE>
static void launch_v3(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_v3<T, D, NW, RB, THR, PRE>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64),
                     (fwd16v2_lds_bytes<D, NW, RB>()), stream, args, g);
}

template <typename T, int D, int NW, int RB, int THR, int PRE>
static void fill(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16_v3<T, D, NW, RB, THR, PRE>);
  v->name = name;
  v->parallelization = NW * RB * 32;
  v->traversal = 64;
  v->headBlock = D;
  v->threads = NW * 64;
  v->ldsBytes = fwd16v2_lds_bytes<D, NW, RB>();
  v->cacheLeft = true;
  v->launch = &launch_v3<T, D, NW, RB, THR, PRE>;
}

// impl 0: 8 waves x 32 rows, compiler-placed LDS reads; 1: K fragments hoisted; 2: K + first V
// fragments hoisted; 3: 4 waves x 64 rows (K hoisted)
bool fwd16_v3_variant(int precision, int D, int impl, VariantInfo *out) {
  if (precision == PREC_BF16) {
    if (D == 128 && impl == 0) { fill<__bf16, 128, 8, 1, 8, 0>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8"); return true; }
    if (D == 128 && impl == 1) { fill<__bf16, 128, 8, 1, 8, 1>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_prek"); return true; }
    if (D == 128 && impl == 2) { fill<__bf16, 128, 8, 1, 8, 2>(out, "attn_fwd16v3_bf16_d128_w8x32_thr8_prekv"); return true; }
    if (D == 128 && impl == 3) { fill<__bf16, 128, 4, 2, 8, 1>(out, "attn_fwd16v3_bf16_d128_w4x64_thr8_prek"); return true; }
    if (D == 64 && impl == 0) { fill<__bf16, 64, 8, 1, 8, 0>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8"); return true; }
    if (D == 64 && impl == 1) { fill<__bf16, 64, 8, 1, 8, 1>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8_prek"); return true; }
    if (D == 64 && impl == 2) { fill<__bf16, 64, 8, 1, 8, 2>(out, "attn_fwd16v3_bf16_d64_w8x32_thr8_prekv"); return true; }
    if (D == 64 && impl == 3) { fill<__bf16, 64, 4, 2, 8, 1>(out, "attn_fwd16v3_bf16_d64_w4x64_thr8_prek"); return true; }
  }
  return false;
}

} // namespace mfa
