// attn_fwd16_v3_tr_d256.hip -- the 16-bit forward kernel (attn_fwd16_v3.h) for operands stored transposed, read in place: head-dimension
// bucket 256.  Register-staged ring, fragment reads left to hipcc.
#include "attn_fwd16_v3_launch.h"

namespace mfa {

bool fwd16_v3_tr_variant_d256(int precision, int D, int pattern, VariantInfo *out) {
  if (D == 256) {
    if (precision == PREC_BF16) { MFA_FWD16_V3_TR_BUCKET(__bf16, "bf16", 256, 4, 2, 0, "w4x32") }
    if (precision == PREC_FP16) { MFA_FWD16_V3_TR_BUCKET(_Float16, "f16", 256, 4, 2, 0, "w4x32") }
  }
  return false;
}

} // namespace mfa
