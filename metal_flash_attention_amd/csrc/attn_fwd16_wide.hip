// attn_fwd16_wide.hip -- instantiations and launchers of the 256 < D <= 384 forward kernel on the 16-bit matrix cores (attn_fwd16_wide.h).
#include "attn_fwd16_wide.h"
#include "launchers.h"

namespace mfa {

template <typename T, int DP, bool CAUSAL>
static void launch_wide(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_wide<T, DP, CAUSAL>), dim3(grid.x * grid.y * grid.z), dim3(256), wide::lds_bytes<DP>(), stream, args, g);
}

template <typename T, int DP> static void fill_wide(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16_wide<T, DP, false>);
  v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_wide<T, DP, true>);
  v->name = name;
  v->parallelization = wide::ROWS;
  v->traversal = wide::BK;
  v->headBlock = DP;
  v->threads = 256;
  v->ldsBytes = wide::lds_bytes<DP>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->causal = true;
  v->launch = &launch_wide<T, DP, false>;
  v->launchCausal = &launch_wide<T, DP, true>;
}

// head blocks 320 and 384: four waves x 32 rows, 32-key steps (attn_fwd16_wide.h); no column-parallel / block-sparse siblings (such
// launches stay row-parallel / go to the general kernel)
bool fwd16_wide_variant(int precision, int D, VariantInfo *out) {
  if (D <= 256 || D > 384) return false;
  const bool d320 = D <= 320;
  if (precision == PREC_BF16) {
    if (d320) fill_wide<__bf16, 320>(out, "attn_fwd16w_bf16_d320_w4x32_thr8"); else fill_wide<__bf16, 384>(out, "attn_fwd16w_bf16_d384_w4x32_thr8");
    return true;
  }
  if (precision == PREC_FP16) {
    if (d320) fill_wide<_Float16, 320>(out, "attn_fwd16w_f16_d320_w4x32_thr8"); else fill_wide<_Float16, 384>(out, "attn_fwd16w_f16_d384_w4x32_thr8");
    return true;
  }
  return false;
}

} // namespace mfa
