// attn_dkv16_rs.hip -- instantiations of the role-split backwardKeyValue kernel (attn_dkv16_rs.h).
#include "attn_dkv16_rs.h"
#include "launchers.h"

namespace mfa {

template <typename T, int D, typename TG, bool CAUSAL, int ABL = 0>
static void launch_rs(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dkv16_rs<T, D, TG, CAUSAL, ABL>), dim3(grid.x * grid.y * grid.z), dim3(dkv16rs_pairs<D>() * 128), (dkv16rs_lds_bytes<D>()), stream,
                     args, g);
}

template <typename T, int D, typename TG>
static void launch_rs_sparse(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  if (args.causal)
    hipLaunchKernelGGL((attn_dkv16_rs<T, D, TG, true, 0, true>), dim3(grid.x * grid.y * grid.z), dim3(dkv16rs_pairs<D>() * 128), (dkv16rs_lds_bytes<D>()), stream, args, g);
  else
    hipLaunchKernelGGL((attn_dkv16_rs<T, D, TG, false, 0, true>), dim3(grid.x * grid.y * grid.z), dim3(dkv16rs_pairs<D>() * 128), (dkv16rs_lds_bytes<D>()), stream, args, g);
}

template <typename T, int D, typename TG>
static void launch_rs_split(dim3 grid, uint32_t splits, float *ws, float *, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, ws, nullptr};
  const dim3 blocks(grid.x * grid.y * grid.z * splits), threads(dkv16rs_pairs<D>() * 128);
  if (args.causal)
    hipLaunchKernelGGL((attn_dkv16_rs<T, D, TG, true, 0, false, true>), blocks, threads, (dkv16rs_lds_bytes<D>()), stream, args, g);
  else
    hipLaunchKernelGGL((attn_dkv16_rs<T, D, TG, false, 0, false, true>), blocks, threads, (dkv16rs_lds_bytes<D>()), stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.C;
  const float *dk_slabs = ws + (uint64_t)splits * rows * args.D;   // dV slabs first, then dK slabs
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dV, args.C, (const float *)ws);
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dK, args.C, dk_slabs);
}

template <typename T, int D, typename TG = T>
static void fill(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_dkv16_rs<T, D, TG, false>);
  v->name = name;
  v->parallelization = dkv16rs_pairs<D>() * 32;   // key columns per workgroup: wave pairs x 32
  v->traversal = 32;
  v->headBlock = D;
  v->threads = dkv16rs_pairs<D>() * 128;
  v->ldsBytes = dkv16rs_lds_bytes<D>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_rs<T, D, TG, false>;
  v->launchCausal = &launch_rs<T, D, TG, true>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dkv16_rs<T, D, TG, true>);
  v->causal = true;
  v->launchSparse = &launch_rs_sparse<T, D, TG>;
  v->funcSparse = reinterpret_cast<const void *>(&attn_dkv16_rs<T, D, TG, false, 0, true>);
  v->funcSparseCausal = reinterpret_cast<const void *>(&attn_dkv16_rs<T, D, TG, true, 0, true>);
  v->launchSplit = &launch_rs_split<T, D, TG>;
  v->funcSplit = reinterpret_cast<const void *>(&attn_dkv16_rs<T, D, TG, false, 0, false, true>);
  v->funcSplitCausal = reinterpret_cast<const void *>(&attn_dkv16_rs<T, D, TG, true, 0, false, true>);
}

bool dkv16_rs_variant(int precision, int gprecision, int D, int impl, VariantInfo *out) {
#ifdef MFA_DEV_VARIANTS
  if (impl >= 1 && impl <= 4 && precision == PREC_BF16 && gprecision == PREC_BF16 && D == 128) {   // timing-only ablations
    fill<__bf16, 128>(out, "ablate_dkv16rs_WRONG_RESULTS");
    out->launchCausal = nullptr; out->funcCausal = nullptr; out->causal = false;
    out->launchSparse = nullptr; out->funcSparse = nullptr; out->funcSparseCausal = nullptr;
    out->launchSplit = nullptr; out->funcSplit = nullptr; out->funcSplitCausal = nullptr;
    switch (impl) {
      case 1: out->func = reinterpret_cast<const void *>(&attn_dkv16_rs<__bf16, 128, __bf16, false, 1>); out->launch = &launch_rs<__bf16, 128, __bf16, false, 1>; break;
      case 2: out->func = reinterpret_cast<const void *>(&attn_dkv16_rs<__bf16, 128, __bf16, false, 2>); out->launch = &launch_rs<__bf16, 128, __bf16, false, 2>; break;
      case 3: out->func = reinterpret_cast<const void *>(&attn_dkv16_rs<__bf16, 128, __bf16, false, 3>); out->launch = &launch_rs<__bf16, 128, __bf16, false, 3>; break;
      default: out->func = reinterpret_cast<const void *>(&attn_dkv16_rs<__bf16, 128, __bf16, false, 4>); out->launch = &launch_rs<__bf16, 128, __bf16, false, 4>; break;
    }
    return true;
  }
#endif
  if (precision == PREC_FP16 && gprecision == PREC_BF16) {
    if (D == 128) { fill<_Float16, 128, __bf16>(out, "attn_dkv16rs_f16_dObf16_d128_p4x32"); return true; }
    if (D == 64) { fill<_Float16, 64, __bf16>(out, "attn_dkv16rs_f16_dObf16_d64_p4x32"); return true; }
    if (D == 256) { fill<_Float16, 256, __bf16>(out, "attn_dkv16rs_f16_dObf16_d256_p2x32"); return true; }
    return false;
  }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) {
    if (D == 128) { fill<__bf16, 128>(out, "attn_dkv16rs_bf16_d128_p4x32"); return true; }
    if (D == 64) { fill<__bf16, 64>(out, "attn_dkv16rs_bf16_d64_p4x32"); return true; }
    if (D == 256) { fill<__bf16, 256>(out, "attn_dkv16rs_bf16_d256_p2x32"); return true; }
  }
  if (precision == PREC_FP16) {
    if (D == 128) { fill<_Float16, 128>(out, "attn_dkv16rs_f16_d128_p4x32"); return true; }
    if (D == 64) { fill<_Float16, 64>(out, "attn_dkv16rs_f16_d64_p4x32"); return true; }
    if (D == 256) { fill<_Float16, 256>(out, "attn_dkv16rs_f16_d256_p2x32"); return true; }
  }
  return false;
}

} // namespace mfa
