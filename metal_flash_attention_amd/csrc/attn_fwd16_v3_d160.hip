// attn_fwd16_v3_d160.hip -- head-dimension bucket 160 of the 16-bit forward kernel (attn_fwd16_v3.h): D = 136 .. 160 no longer
// runs the next power-of-two code object with up to 47 % padded matrix work.  K rows padded instead of XOR-swizzled (the
// swizzle needs a power-of-two row); three-stage ring, register-staged (the grouped V^T schedule of D = 256 needs D / 32 even).
#include "attn_fwd16_v3_launch.h"

namespace mfa {

bool fwd16_v3_variant_d160(int precision, VariantInfo *out) {
  if (precision == PREC_BF16) { fill_with_split<__bf16, 160, 4, 1, 8, 0, 3, 2>(out, "attn_fwd16v3_bf16_d160_w4x32_thr8_kpad"); return true; }
  if (precision == PREC_FP16) { fill_with_split<_Float16, 160, 4, 1, 8, 0, 3, 2>(out, "attn_fwd16v3_f16_d160_w4x32_thr8_kpad"); return true; }
  return false;
}

} // namespace mfa
