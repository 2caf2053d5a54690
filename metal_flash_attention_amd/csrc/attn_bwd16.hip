// attn_bwd16.hip -- instantiations of the 16-bit-MFMA backward kernels for gfx950.
#include "attn_bwd16_launch.h"

namespace mfa {

// precision: storage type of Q, K, V; gprecision: storage type of dO (equal, or BF16 next to FP16)
bool dq16_variant(int precision, int gprecision, int D, VariantInfo *out) {
  if (precision == PREC_FP16 && gprecision == PREC_BF16) {
    if (D == 128) { fill_dq<_Float16, 128, 8, __bf16>(out, "attn_dq16_f16_dObf16_d128_w8x32"); return true; }
    if (D == 64) { fill_dq<_Float16, 64, 8, __bf16>(out, "attn_dq16_f16_dObf16_d64_w8x32"); return true; }
    if (D == 256) { fill_dq<_Float16, 256, 4, __bf16>(out, "attn_dq16_f16_dObf16_d256_w4x32"); return true; }
    return false;
  }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) {
    if (D == 128) { fill_dq<__bf16, 128, 8>(out, "attn_dq16_bf16_d128_w8x32"); return true; }
    if (D == 64) { fill_dq<__bf16, 64, 8>(out, "attn_dq16_bf16_d64_w8x32"); return true; }
    if (D == 256) { fill_dq<__bf16, 256, 4>(out, "attn_dq16_bf16_d256_w4x32"); return true; }   // one wave per SIMD: dQ alone takes 128 registers
  }
  if (precision == PREC_FP16) {
    if (D == 128) { fill_dq<_Float16, 128, 8>(out, "attn_dq16_f16_d128_w8x32"); return true; }
    if (D == 64) { fill_dq<_Float16, 64, 8>(out, "attn_dq16_f16_d64_w8x32"); return true; }
    if (D == 256) { fill_dq<_Float16, 256, 4>(out, "attn_dq16_f16_d256_w4x32"); return true; }
  }
  return false;
}

bool dkv16_variant(int precision, int gprecision, int D, VariantInfo *out) {
  if (precision == PREC_FP16 && gprecision == PREC_BF16) {
    if (D == 128) { fill_dkv<_Float16, 128, 4, 1, __bf16>(out, "attn_dkv16_f16_dObf16_d128_w4x32"); return true; }
    if (D == 64) { fill_dkv<_Float16, 64, 4, 1, __bf16>(out, "attn_dkv16_f16_dObf16_d64_w4x32"); return true; }
    return false;
  }
  if (precision != gprecision) return false;
#ifdef MFA_DEV_VARIANTS
  const char *knob = std::getenv("MFA_DKV16_IMPL");   // developer A/B knob: "0" = compiler-placed LDS reads
  if (knob && knob[0] == '0' && precision == PREC_BF16 && D == 128) {
    fill_dkv<__bf16, 128, 4, 0>(out, "attn_dkv16_bf16_d128_w4x32_nopre"); return true;
  }
#endif
  if (precision == PREC_BF16) {
    if (D == 128) { fill_dkv<__bf16, 128, 4>(out, "attn_dkv16_bf16_d128_w4x32"); return true; }
    if (D == 64) { fill_dkv<__bf16, 64, 4>(out, "attn_dkv16_bf16_d64_w4x32"); return true; }
  }
  if (precision == PREC_FP16) {
    if (D == 128) { fill_dkv<_Float16, 128, 4>(out, "attn_dkv16_f16_d128_w4x32"); return true; }
    if (D == 64) { fill_dkv<_Float16, 64, 4>(out, "attn_dkv16_f16_d64_w4x32"); return true; }
  }
  return false;
}

} // namespace mfa
