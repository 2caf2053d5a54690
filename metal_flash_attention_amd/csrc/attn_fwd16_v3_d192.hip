// attn_fwd16_v3_d192.hip -- head-dimension bucket 192 of the 16-bit forward kernel (attn_fwd16_v3.h): D = 168 .. 192 no longer
// runs the next power-of-two code object with up to 34 % padded matrix work.  K rows padded instead of XOR-swizzled (the
// swizzle needs a power-of-two row); the D = 256 schedule (two-stage ring, grouped V^T reads, spread staging writes).
#include "attn_fwd16_v3_launch.h"

namespace mfa {

bool fwd16_v3_variant_d192(int precision, VariantInfo *out) {
  if (precision == PREC_BF16) { fill_with_split<__bf16, 192, 4, 1, 8, 1, 2, 14>(out, "attn_fwd16v3_bf16_d192_w4x32_thr8_ring2_spread_kpad"); return true; }
  if (precision == PREC_FP16) { fill_with_split<_Float16, 192, 4, 1, 8, 1, 2, 14>(out, "attn_fwd16v3_f16_d192_w4x32_thr8_ring2_spread_kpad"); return true; }
  return false;
}

} // namespace mfa
