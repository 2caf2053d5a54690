// attn_bwd16_launch.h -- host-side launchers and VariantInfo fill templates of attn_bwd16.h
#pragma once
#include "attn_bwd16.h"
#include "launchers.h"
#include <cstdlib>

namespace mfa {


template <typename T, int D, int NW, typename TG = T>
static void launch_dq16(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dq16<T, D, NW, TG>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64),
                     (dq16_lds_bytes<D, NW>()), stream, args, g);
}
template <typename T, int D, int NW, int PRE = 1, typename TG = T>
static void launch_dkv16(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dkv16<T, D, NW, PRE, TG>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64),
                     (dkv16_lds_bytes<D, NW>()), stream, args, g);
}

template <typename T, int D, int NW, typename TG = T>
static void launch_dq16_causal(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dq16<T, D, NW, TG, true>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64),
                     (dq16_lds_bytes<D, NW>()), stream, args, g);
}
template <typename T, int D, int NW, int PRE = 1, typename TG = T>
static void launch_dkv16_causal(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dkv16<T, D, NW, PRE, TG, true>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64),
                     (dkv16_lds_bytes<D, NW>()), stream, args, g);
}

template <typename T, int D, int NW, typename TG>
static void launch_dq16_sparse(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  if (args.causal)
    hipLaunchKernelGGL((attn_dq16<T, D, NW, TG, true, true>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64), (dq16_lds_bytes<D, NW>()), stream, args, g);
  else
    hipLaunchKernelGGL((attn_dq16<T, D, NW, TG, false, true>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64), (dq16_lds_bytes<D, NW>()), stream, args, g);
}

template <typename T, int D, int NW, typename TG>
static void launch_dq16_split(dim3 grid, uint32_t splits, float *ws, float *, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, ws, nullptr};
  const dim3 blocks(grid.x * grid.y * grid.z * splits);
  if (args.causal)
    hipLaunchKernelGGL((attn_dq16<T, D, NW, TG, true, false, true>), blocks, dim3(NW * 64), (dq16_lds_bytes<D, NW>()), stream, args, g);
  else
    hipLaunchKernelGGL((attn_dq16<T, D, NW, TG, false, false, true>), blocks, dim3(NW * 64), (dq16_lds_bytes<D, NW>()), stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.R;
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dQ, args.R, (const float *)ws);
}

template <typename T, int D, int NW, typename TG = T>
static void fill_dq(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_dq16<T, D, NW, TG>);
  v->name = name;
  v->parallelization = NW * 32;
  v->traversal = 64;
  v->headBlock = D;
  v->threads = NW * 64;
  v->ldsBytes = dq16_lds_bytes<D, NW>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_dq16<T, D, NW, TG>;
  v->launchCausal = &launch_dq16_causal<T, D, NW, TG>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dq16<T, D, NW, TG, true>);
  v->causal = true;
  v->launchSparse = &launch_dq16_sparse<T, D, NW, TG>;
  v->funcSparse = reinterpret_cast<const void *>(&attn_dq16<T, D, NW, TG, false, true>);
  v->funcSparseCausal = reinterpret_cast<const void *>(&attn_dq16<T, D, NW, TG, true, true>);
  v->launchSplit = &launch_dq16_split<T, D, NW, TG>;
  v->funcSplit = reinterpret_cast<const void *>(&attn_dq16<T, D, NW, TG, false, false, true>);
  v->funcSplitCausal = reinterpret_cast<const void *>(&attn_dq16<T, D, NW, TG, true, false, true>);
}
template <typename T, int D, int NW, int PRE = 1, typename TG = T>
static void fill_dkv(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_dkv16<T, D, NW, PRE, TG>);
  v->name = name;
  v->parallelization = NW * 32;
  v->traversal = 64;
  v->headBlock = D;
  v->threads = NW * 64;
  v->ldsBytes = dkv16_lds_bytes<D, NW>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_dkv16<T, D, NW, PRE, TG>;
  v->launchCausal = &launch_dkv16_causal<T, D, NW, PRE, TG>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dkv16<T, D, NW, PRE, TG, true>);
  v->causal = true;
}

} // namespace mfa
