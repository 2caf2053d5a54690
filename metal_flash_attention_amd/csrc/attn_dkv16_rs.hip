// attn_dkv16_rs.hip -- instantiations of the role-split backwardKeyValue kernel (attn_dkv16_rs.h).
#include "attn_dkv16_rs_launch.h"

namespace mfa {

bool dkv16_rs_variant(int precision, int gprecision, int D, int impl, VariantInfo *out) {
#ifdef MFA_DEV_VARIANTS
  if (impl >= 1 && impl <= 4 && precision == PREC_BF16 && gprecision == PREC_BF16 && D == 128) {   // timing-only ablations
    fill<__bf16, 128>(out, "ablate_dkv16rs_WRONG_RESULTS");
    out->launchCausal = nullptr; out->funcCausal = nullptr; out->causal = false;
    out->launchSparse = nullptr; out->funcSparse = nullptr; out->funcSparseCausal = nullptr;
    out->launchSplit = nullptr; out->funcSplit = nullptr; out->funcSplitCausal = nullptr;
    switch (impl) {
      case 1: out->func = reinterpret_cast<const void *>(&attn_dkv16_rs<__bf16, 128, __bf16, false, 1>); out->launch = &launch_rs<__bf16, 128, __bf16, false, 1>; break;
      case 2: out->func = reinterpret_cast<const void *>(&attn_dkv16_rs<__bf16, 128, __bf16, false, 2>); out->launch = &launch_rs<__bf16, 128, __bf16, false, 2>; break;
      case 3: out->func = reinterpret_cast<const void *>(&attn_dkv16_rs<__bf16, 128, __bf16, false, 3>); out->launch = &launch_rs<__bf16, 128, __bf16, false, 3>; break;
      default: out->func = reinterpret_cast<const void *>(&attn_dkv16_rs<__bf16, 128, __bf16, false, 4>); out->launch = &launch_rs<__bf16, 128, __bf16, false, 4>; break;
    }
    return true;
  }
#endif
  if (precision == PREC_FP16 && gprecision == PREC_BF16) {
    if (D == 128) { fill<_Float16, 128, __bf16>(out, "attn_dkv16rs_f16_dObf16_d128_p4x32"); return true; }
    if (D == 64) { fill<_Float16, 64, __bf16>(out, "attn_dkv16rs_f16_dObf16_d64_p4x32"); return true; }
    if (D == 256) { fill<_Float16, 256, __bf16>(out, "attn_dkv16rs_f16_dObf16_d256_p2x32"); return true; }
    return false;
  }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) {
    if (D == 128) { fill<__bf16, 128>(out, "attn_dkv16rs_bf16_d128_p4x32"); return true; }
    if (D == 64) { fill<__bf16, 64>(out, "attn_dkv16rs_bf16_d64_p4x32"); return true; }
    if (D == 256) { fill<__bf16, 256>(out, "attn_dkv16rs_bf16_d256_p2x32"); return true; }
  }
  if (precision == PREC_FP16) {
    if (D == 128) { fill<_Float16, 128>(out, "attn_dkv16rs_f16_d128_p4x32"); return true; }
    if (D == 64) { fill<_Float16, 64>(out, "attn_dkv16rs_f16_d64_p4x32"); return true; }
    if (D == 256) { fill<_Float16, 256>(out, "attn_dkv16rs_f16_d256_p2x32"); return true; }
  }
  return false;
}

} // namespace mfa
