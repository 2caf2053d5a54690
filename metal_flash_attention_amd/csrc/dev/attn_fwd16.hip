// attn_fwd16.hip -- instantiations of the 16-bit-MFMA forward kernel for gfx950.
#include "attn_fwd16.h"
#include "launchers.h"

namespace mfa {

template <typename T, int D, int NW, int RB>
static void launch_fwd16(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  // `grid` arrives as (row blocks, heads, batches); the kernel uses a flat XCD-aware order
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16<T, D, NW, RB>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64),
                     fwd16_lds_bytes<D>(), stream, args, g);
}

template <typename T, int D, int NW, int RB>
static void fill(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16<T, D, NW, RB>);
  v->name = name;
  v->parallelization = NW * RB * 32;
  v->traversal = 64;
  v->headBlock = D;
  v->threads = NW * 64;
  v->ldsBytes = fwd16_lds_bytes<D>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_fwd16<T, D, NW, RB>;
}

// precision: PREC_FP16 or PREC_BF16; D: padded head dimension bucket
bool fwd16_variant(int precision, int D, VariantInfo *out) {
  if (precision == PREC_BF16) {
    switch (D) {
      case 32:  fill<__bf16, 32, 4, 1>(out, "attn_fwd16_bf16_d32_w4x32"); return true;
      case 64:  fill<__bf16, 64, 8, 1>(out, "attn_fwd16_bf16_d64_w8x32"); return true;
      case 128: fill<__bf16, 128, 8, 1>(out, "attn_fwd16_bf16_d128_w8x32"); return true;
      case 256: fill<__bf16, 256, 4, 1>(out, "attn_fwd16_bf16_d256_w4x32"); return true;
      default: return false;
    }
  }
  if (precision == PREC_FP16) {
    switch (D) {
      case 32:  fill<_Float16, 32, 4, 1>(out, "attn_fwd16_f16_d32_w4x32"); return true;
      case 64:  fill<_Float16, 64, 8, 1>(out, "attn_fwd16_f16_d64_w8x32"); return true;
      case 128: fill<_Float16, 128, 8, 1>(out, "attn_fwd16_f16_d128_w8x32"); return true;
      case 256: fill<_Float16, 256, 4, 1>(out, "attn_fwd16_f16_d256_w4x32"); return true;
      default: return false;
    }
  }
  return false;
}

} // namespace mfa
