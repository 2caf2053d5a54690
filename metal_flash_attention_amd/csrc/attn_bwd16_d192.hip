// attn_bwd16_d192.hip -- head-dimension bucket 192 of the 16-bit backwardQuery kernel (attn_bwd16.h): four waves, one per SIMD.
#include "attn_bwd16_launch.h"

namespace mfa {

bool dq16_variant_d192(int precision, int gprecision, VariantInfo *out) {
  if (precision == PREC_FP16 && gprecision == PREC_BF16) { fill_dq<_Float16, 192, 4, __bf16>(out, "attn_dq16_f16_dObf16_d192_w4x32"); return true; }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) { fill_dq<__bf16, 192, 4>(out, "attn_dq16_bf16_d192_w4x32"); return true; }
  if (precision == PREC_FP16) { fill_dq<_Float16, 192, 4>(out, "attn_dq16_f16_d192_w4x32"); return true; }
  return false;
}

} // namespace mfa
