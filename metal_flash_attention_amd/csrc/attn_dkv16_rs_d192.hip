// attn_dkv16_rs_d192.hip -- head-dimension bucket 192 of the role-split backwardKeyValue kernel (attn_dkv16_rs.h).
#include "attn_dkv16_rs_launch.h"

namespace mfa {

bool dkv16_rs_variant_d192(int precision, int gprecision, VariantInfo *out) {
  if (precision == PREC_FP16 && gprecision == PREC_BF16) { fill<_Float16, 192, __bf16>(out, "attn_dkv16rs_f16_dObf16_d192_p2x32"); return true; }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) { fill<__bf16, 192>(out, "attn_dkv16rs_bf16_d192_p2x32"); return true; }
  if (precision == PREC_FP16) { fill<_Float16, 192>(out, "attn_dkv16rs_f16_d192_p2x32"); return true; }
  return false;
}

} // namespace mfa
