// attn_bwd16_d160.hip -- head-dimension bucket 160 of the 16-bit backwardQuery kernel (attn_bwd16.h): four waves, one per SIMD.
#include "attn_bwd16_launch.h"

namespace mfa {

bool dq16_variant_d160(int precision, int gprecision, VariantInfo *out) {
  if (precision == PREC_FP16 && gprecision == PREC_BF16) { fill_dq<_Float16, 160, 4, __bf16>(out, "attn_dq16_f16_dObf16_d160_w4x32"); return true; }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) { fill_dq<__bf16, 160, 4>(out, "attn_dq16_bf16_d160_w4x32"); return true; }
  if (precision == PREC_FP16) { fill_dq<_Float16, 160, 4>(out, "attn_dq16_f16_d160_w4x32"); return true; }
  return false;
}

} // namespace mfa
