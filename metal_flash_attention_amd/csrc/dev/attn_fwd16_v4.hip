// attn_fwd16_v4.hip -- instantiations of the role-alternating forward kernel (attn_fwd16_v4.h).
#include "attn_fwd16_v4.h"
#include "launchers.h"

namespace mfa {

template <typename T, int D, int THR, int OPT, int RING>
static void launch_v4(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_v4<T, D, 8, THR, OPT, RING, false, false>), dim3(grid.x * grid.y * grid.z), dim3(512),
                     (fwd16v2_lds_bytes<D, 8, 1, RING>()), stream, args, g);
}
template <typename T, int D, int THR, int OPT, int RING>
static void launch_v4_causal(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_v4<T, D, 8, THR, OPT, RING, false, true>), dim3(grid.x * grid.y * grid.z), dim3(512),
                     (fwd16v2_lds_bytes<D, 8, 1, RING>()), stream, args, g);
}
template <typename T, int D, int THR, int OPT, int RING>
static void launch_v4_split(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, wsO, wsML};
  hipLaunchKernelGGL((attn_fwd16_v4<T, D, 8, THR, OPT, RING, true, false>), dim3(grid.x * grid.y * grid.z * splits), dim3(512),
                     (fwd16v2_lds_bytes<D, 8, 1, RING>()), stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.R;
  hipLaunchKernelGGL(attn_fwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g);
}

template <typename T, int D, int THR, int OPT, int RING, bool FULL>
static void fill(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16_v4<T, D, 8, THR, OPT, RING, false, false>);
  v->name = name;
  v->parallelization = 256;
  v->traversal = 64;
  v->headBlock = D;
  v->threads = 512;
  v->ldsBytes = fwd16v2_lds_bytes<D, 8, 1, RING>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_v4<T, D, THR, OPT, RING>;
  if constexpr (FULL) {
    v->launchSplit = &launch_v4_split<T, D, THR, OPT, RING>;
    v->launchCausal = &launch_v4_causal<T, D, THR, OPT, RING>;
    v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_v4<T, D, 8, THR, OPT, RING, false, true>);
    v->causal = true;
  }
}

// impl 0: plain role alternation, 3-stage ring.  1-15: OPT bits (attn_fwd16_v4.h).  16 + OPT: 2-stage ring.
bool fwd16_v4_variant(int precision, int D, int impl, VariantInfo *out) {
#define MFA_V4(TYPE, TAG, DD)                                                                              \
  if (D == DD && impl == 0) { fill<TYPE, DD, 8, 0, 3, true>(out, "attn_fwd16v4_" TAG "_d" #DD "_pp"); return true; } \
  if (D == DD && impl == 1) { fill<TYPE, DD, 8, 1, 3, false>(out, "attn_fwd16v4_" TAG "_d" #DD "_pp_mprio"); return true; } \
  if (D == DD && impl == 2) { fill<TYPE, DD, 8, 2, 3, false>(out, "attn_fwd16v4_" TAG "_d" #DD "_pp_young"); return true; } \
  if (D == DD && impl == 4) { fill<TYPE, DD, 8, 4, 3, false>(out, "attn_fwd16v4_" TAG "_d" #DD "_pp_pvfirst"); return true; } \
  if (D == DD && impl == 8) { fill<TYPE, DD, 8, 8, 3, false>(out, "attn_fwd16v4_" TAG "_d" #DD "_pp_latereads"); return true; } \
  if (D == DD && impl == 16) { fill<TYPE, DD, 8, 0, 2, false>(out, "attn_fwd16v4_" TAG "_d" #DD "_pp_ring2"); return true; }
  if (precision == PREC_BF16) {
    MFA_V4(__bf16, "bf16", 128)
    MFA_V4(__bf16, "bf16", 64)
  }
  if (precision == PREC_FP16) {
    if (D == 128 && impl == 0) { fill<_Float16, 128, 8, 0, 3, true>(out, "attn_fwd16v4_f16_d128_pp"); return true; }
    if (D == 64 && impl == 0) { fill<_Float16, 64, 8, 0, 3, true>(out, "attn_fwd16v4_f16_d64_pp"); return true; }
  }
#undef MFA_V4
  return false;
}

} // namespace mfa
