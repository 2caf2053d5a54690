// attn_generic_dq.hip -- instantiations of the generic (fp32-MFMA) dq kernel for gfx950.
#include "attn_generic.h"
#include "launchers.h"
#include <cstdlib>

namespace mfa {

template <int DP, int NW, bool CACHE, bool X = false>
static void launch_dq(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if constexpr (NW == 4 && CACHE && (DP == 64 || DP == 128))   // the FP32 production case: attn_f32.h
    if (f32_launch(1, DP, grid, stream, args)) return;
  constexpr uint32_t lds = generic_dq_lds_floats<DP, NW, CACHE, X>() * sizeof(float);
  hipLaunchKernelGGL((attn_generic_dq<DP, NW, CACHE, false, X>), grid, dim3(NW * 64), lds, stream, args);
}

template <int DP, int NW, bool CACHE, bool X = false>
static void launch_dq_masked(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  constexpr uint32_t lds = generic_dq_lds_floats<DP, NW, CACHE, X>() * sizeof(float);
  hipLaunchKernelGGL((attn_generic_dq<DP, NW, CACHE, true, X>), grid, dim3(NW * 64), lds, stream, args);
}

template <int DP> static const char *f32_form_of(const KernelArgs &args) { return f32_form(1, DP, args); }

template <int DP, int NW, bool CACHE, bool X = false>
static void fill(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_generic_dq<DP, NW, CACHE, false, X>);
  v->name = name;
  v->parallelization = NW * 32;
  v->traversal = 32;
  v->headBlock = DP;
  v->threads = NW * 64;
  v->ldsBytes = generic_dq_lds_floats<DP, NW, CACHE, X>() * sizeof(float);
  v->cacheLeft = CACHE || X;
  v->cacheSecond = CACHE;
  v->causal = true;
  v->launchSparse = &launch_dq_masked<DP, NW, CACHE, X>;   // block mask: own code objects
  v->funcSparse = reinterpret_cast<const void *>(&attn_generic_dq<DP, NW, CACHE, true, X>);
  v->launch = &launch_dq<DP, NW, CACHE, X>;
  if constexpr (NW == 4 && CACHE && (DP == 64 || DP == 128)) v->launchForm = &f32_form_of<DP>;
}

bool generic_dq_variant(int DP, VariantInfo *out) {
  switch (DP) {
    case 32:  fill<32, 4, true>(out, "attn_generic_dq_f32mfma_d32_w4_cached"); return true;
    case 64:  fill<64, 4, true>(out, "attn_generic_dq_f32mfma_d64_w4_cached"); return true;
    case 128: fill<128, 4, true>(out, "attn_generic_dq_f32mfma_d128_w4_cached"); return true;   // (2 waves per workgroup, i.e. half the LDS and twice the workgroups per CU: 10-15 % slower, measured)
    case 256: fill<256, 4, true>(out, "attn_generic_dq_f32mfma_d256_w4_cached"); return true;
    case 384: fill<384, 1, false, true>(out, "attn_generic_dq_f32mfma_d384_w1_qcached"); return true;
    default: return false;
  }
}

} // namespace mfa
