// attn_generic_fwd.hip -- instantiations of the generic (fp32-MFMA) fwd kernel for gfx950.
#include "attn_generic.h"
#include "launchers.h"
#include <cstdlib>

namespace mfa {

template <int DP, int NW, bool CACHE>
static void launch_fwd(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if constexpr (NW == 4 && CACHE && (DP == 64 || DP == 128))   // the FP32 production case: attn_f32.h
    if (f32_launch(0, DP, grid, stream, args)) return;
  constexpr uint32_t lds = generic_fwd_lds_floats<DP, NW, CACHE>() * sizeof(float);
  hipLaunchKernelGGL((attn_generic_fwd<DP, NW, CACHE>), grid, dim3(NW * 64), lds, stream, args);
}

template <int DP, int NW, bool CACHE>
static void launch_fwd_masked(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  constexpr uint32_t lds = generic_fwd_lds_floats<DP, NW, CACHE>() * sizeof(float);
  hipLaunchKernelGGL((attn_generic_fwd<DP, NW, CACHE, true>), grid, dim3(NW * 64), lds, stream, args);
}

template <int DP> static const char *f32_form_of(const KernelArgs &args) { return f32_form(0, DP, args); }

template <int DP, int NW, bool CACHE>
static void fill(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_generic_fwd<DP, NW, CACHE>);
  v->name = name;
  v->parallelization = NW * 32;
  v->traversal = 32;
  v->headBlock = DP;
  v->threads = NW * 64;
  v->ldsBytes = generic_fwd_lds_floats<DP, NW, CACHE>() * sizeof(float);
  v->cacheLeft = CACHE;
  v->cacheSecond = CACHE;
  v->causal = true;
  v->launchSparse = &launch_fwd_masked<DP, NW, CACHE>;   // block mask: own code objects
  v->funcSparse = reinterpret_cast<const void *>(&attn_generic_fwd<DP, NW, CACHE, true>);
  v->launch = &launch_fwd<DP, NW, CACHE>;
  if constexpr (NW == 4 && CACHE && (DP == 64 || DP == 128)) v->launchForm = &f32_form_of<DP>;
}

bool generic_fwd_variant(int DP, VariantInfo *out) {
  switch (DP) {
    case 32:  fill<32, 4, true>(out, "attn_generic_fwd_f32mfma_d32_w4_cached"); return true;
    case 64:  fill<64, 4, true>(out, "attn_generic_fwd_f32mfma_d64_w4_cached"); return true;
    case 128: fill<128, 4, true>(out, "attn_generic_fwd_f32mfma_d128_w4_cached"); return true;   // (2 waves per workgroup, i.e. half the LDS and twice the workgroups per CU: 10-15 % slower, measured)
    case 256: fill<256, 4, true>(out, "attn_generic_fwd_f32mfma_d256_w4_cached"); return true;
    // D <= 384: two waves per workgroup (64 rows x 385 floats of LDS staging), 192 registers of O and 192 of Q per lane
    case 384: fill<384, 2, true>(out, "attn_generic_fwd_f32mfma_d384_w2_cached"); return true;
    default: return false;
  }
}

} // namespace mfa
