// attn_bwd16_wide.hip -- the backward kernels for 16-bit inputs at 256 < D <= 384 on the 16-bit matrix cores (round 6).
// backwardKeyValue: attn_dkv16_wide.h.  backwardQuery: attn_dq16 of attn_bwd16.h at the head blocks 320 and 384 -- four waves x 32 rows (Q and dO fragments 2 x D / 4, dQ^T D / 2 registers of a lane's
// 512), 32-key tiles (two stages of {K | V} = 4 x 32 x D x 2 bytes), the epilogue's staging rows two waves at a time.  Dense, causal and
// per-batch lengths; block masks, traversal-parallel pieces and transposed operands keep the general kernel.  Until round 6 these
// launches ran fp32 arithmetic on 16-bit storage (attn_generic_dq: 1/16 of the matrix rate).
// Reference: the `| 384 | ... |` rows of the mixed backwardQuery table, AttentionDescriptor+Parameters.swift:153-170.
#include "attn_bwd16_launch.h"
#include "attn_dkv16_wide.h"

namespace mfa {

template <typename T, int D, typename TG, bool CAUSAL>
static void launch_dq16_wide(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dq16<T, D, 4, TG, CAUSAL, false, false, 32>), dim3(grid.x * grid.y * grid.z), dim3(256),
                     (dq16_lds_bytes<D, 4, 32>()), stream, args, g);
}

template <typename T, int D, typename TG = T> static void fill_dq_wide(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_dq16<T, D, 4, TG, false, false, false, 32>);
  v->name = name;
  v->parallelization = 128;
  v->traversal = 32;
  v->headBlock = D;
  v->threads = 256;
  v->ldsBytes = dq16_lds_bytes<D, 4, 32>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_dq16_wide<T, D, TG, false>;
  v->launchCausal = &launch_dq16_wide<T, D, TG, true>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dq16<T, D, 4, TG, true, false, false, 32>);
  v->causal = true;
}

// precision: storage type of Q, K, V; gprecision: of dO (equal, or BF16 next to FP16)
bool dq16_wide_variant(int precision, int gprecision, int D, VariantInfo *out) {
#define MFA_DQW(DD)                                                                                                       \
  if (D == DD) {                                                                                                          \
    if (precision == PREC_BF16 && gprecision == PREC_BF16) { fill_dq_wide<__bf16, DD>(out, "attn_dq16w_bf16_d" #DD "_w4x32"); return true; } \
    if (precision == PREC_FP16 && gprecision == PREC_FP16) { fill_dq_wide<_Float16, DD>(out, "attn_dq16w_f16_d" #DD "_w4x32"); return true; } \
    if (precision == PREC_FP16 && gprecision == PREC_BF16) { fill_dq_wide<_Float16, DD, __bf16>(out, "attn_dq16w_f16_dObf16_d" #DD "_w4x32"); return true; } \
  }
  MFA_DQW(320)
  MFA_DQW(384)
#undef MFA_DQW
  return false;
}

template <typename T, int D, typename TG, bool CAUSAL>
static void launch_dkv16_wide(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dkv16_wide<T, D, TG, CAUSAL>), dim3(grid.x * grid.y * grid.z), dim3(256), (dkv16w_lds_bytes<D>()), stream, args, g);
}

template <typename T, int D, typename TG = T> static void fill_dkv_wide(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_dkv16_wide<T, D, TG, false>);
  v->name = name;
  v->parallelization = 64;
  v->traversal = 32;
  v->headBlock = D;
  v->threads = 256;
  v->ldsBytes = dkv16w_lds_bytes<D>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_dkv16_wide<T, D, TG, false>;
  v->launchCausal = &launch_dkv16_wide<T, D, TG, true>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dkv16_wide<T, D, TG, true>);
  v->causal = true;
}

bool dkv16_wide_variant(int precision, int gprecision, int D, VariantInfo *out) {
#define MFA_DKVW(DD)                                                                                                      \
  if (D == DD) {                                                                                                          \
    if (precision == PREC_BF16 && gprecision == PREC_BF16) { fill_dkv_wide<__bf16, DD>(out, "attn_dkv16w_bf16_d" #DD "_p2x32"); return true; } \
    if (precision == PREC_FP16 && gprecision == PREC_FP16) { fill_dkv_wide<_Float16, DD>(out, "attn_dkv16w_f16_d" #DD "_p2x32"); return true; } \
    if (precision == PREC_FP16 && gprecision == PREC_BF16) { fill_dkv_wide<_Float16, DD, __bf16>(out, "attn_dkv16w_f16_dObf16_d" #DD "_p2x32"); return true; } \
  }
  MFA_DKVW(320)
  MFA_DKVW(384)
#undef MFA_DKVW
  return false;
}

} // namespace mfa
