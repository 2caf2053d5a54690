// attn_dkv16_rs_launch.h -- host-side launchers and VariantInfo fill templates of attn_dkv16_rs.h
#pragma once
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

} // namespace mfa
