// attn_dkv16_rs_d160.hip -- head-dimension bucket 160 of the role-split backwardKeyValue kernel (attn_dkv16_rs.h).
#include "attn_dkv16_rs_launch.h"

namespace mfa {

bool dkv16_rs_variant_d160(int precision, int gprecision, VariantInfo *out) {
  if (precision == PREC_FP16 && gprecision == PREC_BF16) { fill<_Float16, 160, __bf16>(out, "attn_dkv16rs_f16_dObf16_d160_p2x32"); return true; }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) { fill<__bf16, 160>(out, "attn_dkv16rs_bf16_d160_p2x32"); return true; }
  if (precision == PREC_FP16) { fill<_Float16, 160>(out, "attn_dkv16rs_f16_d160_p2x32"); return true; }
  return false;
}

} // namespace mfa
