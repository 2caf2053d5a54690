// attn_fwd16_v3_launch.h -- host-side launchers and VariantInfo fill templates of attn_fwd16_v3.h (shared by attn_fwd16_v3.hip and the per-head-dimension bucket TUs)
#pragma once
#include "attn_fwd16_v3.h"
#include "launchers.h"

namespace mfa {


// LDS bytes of a schedule: the LDS-DMA schedule on the 2-stage ring keeps three K and two V images (all 160 KiB)
template <int D, int NW, int RB, int RING, int VD> constexpr int fwd16v3_lds_bytes() {
  if ((VD & 32) && RING == 2) return 5 * 64 * D * 2;
  return fwd16v2_lds_bytes<D, NW, RB, RING, (VD & 2) ? 16 : 0>();
}

template <typename T, int D, int NW, int RB, int THR, int PRE, int ABL = 0, int RING = 3, int VD = 0>
static void launch_v3(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_v3<T, D, NW, RB, THR, PRE, ABL, RING, false, false, VD>), dim3(grid.x * grid.y * grid.z), dim3(NW * 64),
                     (fwd16v3_lds_bytes<D, NW, RB, RING, VD>()), stream, args, g);
}

template <typename T, int D, int NW, int RB, int THR, int PRE, int ABL = 0, int RING = 3, int VD = 0>
static void launch_v3_split(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, wsO, wsML};
  hipLaunchKernelGGL((attn_fwd16_v3<T, D, NW, RB, THR, PRE, ABL, RING, true, false, VD>), dim3(grid.x * grid.y * grid.z * splits),
                     dim3(NW * 64), (fwd16v3_lds_bytes<D, NW, RB, RING, VD>()), stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.R;
  hipLaunchKernelGGL(attn_fwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g);
}

template <typename T, int D, int NW, int RB, int THR, int PRE, int ABL = 0, int RING = 3, int VD = 0>
static void fill(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16_v3<T, D, NW, RB, THR, PRE, ABL, RING, false, false, VD>);
  v->name = name;
  v->parallelization = NW * RB * 32;
  v->traversal = 32;   // pipeline step: half a 64-key LDS tile (attn_fwd16_v3.h)
  v->headBlock = D;
  v->threads = NW * 64;
  v->ldsBytes = fwd16v3_lds_bytes<D, NW, RB, RING, VD>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_v3<T, D, NW, RB, THR, PRE, ABL, RING, VD>;
}

template <typename T, int D, int NW, int RB, int THR, int PRE, int RING, int VD>
static void launch_v3_causal(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_v3<T, D, NW, RB, THR, PRE, 0, RING, false, true, VD>), dim3(grid.x * grid.y * grid.z),
                     dim3(NW * 64), (fwd16v3_lds_bytes<D, NW, RB, RING, VD>()), stream, args, g);
}

template <typename T, int D, int NW, int RB, int THR, int PRE, int RING, int VD>
static void launch_v3_sparse(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  if (args.causal)
    hipLaunchKernelGGL((attn_fwd16_v3<T, D, NW, RB, THR, PRE, 0, RING, false, true, VD, true>), dim3(grid.x * grid.y * grid.z),
                       dim3(NW * 64), (fwd16v3_lds_bytes<D, NW, RB, RING, VD>()), stream, args, g);
  else
    hipLaunchKernelGGL((attn_fwd16_v3<T, D, NW, RB, THR, PRE, 0, RING, false, false, VD, true>), dim3(grid.x * grid.y * grid.z),
                       dim3(NW * 64), (fwd16v3_lds_bytes<D, NW, RB, RING, VD>()), stream, args, g);
}

// product variants: the dense code object plus its causal, block-sparse and column-parallel siblings (VDS: schedule
// bits of the block-sparse pair, which restarts its pipeline per run of active tiles and keeps register staging)
template <typename T, int D, int NW, int RB, int THR, int PRE, int RING = 3, int VD = 0, int PRES = PRE, int VDS = VD>
static void fill_with_split(VariantInfo *v, const char *name) {
  fill<T, D, NW, RB, THR, PRE, 0, RING, VD>(v, name);
  v->launchSparse = &launch_v3_sparse<T, D, NW, RB, THR, PRES, RING, VDS>;
  v->funcSparse = reinterpret_cast<const void *>(&attn_fwd16_v3<T, D, NW, RB, THR, PRES, 0, RING, false, false, VDS, true>);
  v->funcSparseCausal = reinterpret_cast<const void *>(&attn_fwd16_v3<T, D, NW, RB, THR, PRES, 0, RING, false, true, VDS, true>);
  v->launchSplit = &launch_v3_split<T, D, NW, RB, THR, PRE, 0, RING, VD>;
  v->funcSplit = reinterpret_cast<const void *>(&attn_fwd16_v3<T, D, NW, RB, THR, PRE, 0, RING, true, false, VD>);
  v->launchCausal = &launch_v3_causal<T, D, NW, RB, THR, PRE, RING, VD>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_v3<T, D, NW, RB, THR, PRE, 0, RING, false, true, VD>);
  v->causal = true;
}

// transposed operands read in place (TR of attn_fwd16_v3.h): one code object per pattern of (K, V); Q / O and the causal mask
// are run-time flags of these kernels.  No column-parallel or block-sparse siblings: such launches stay row-parallel /
// go to the general kernel.
template <typename T, int D, int NW, int RING, int VD, int TR>
static void launch_v3_tr(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_v3<T, D, NW, 1, 8, 0, 0, RING, false, true, VD, false, TR>), dim3(grid.x * grid.y * grid.z),
                     dim3(NW * 64), (fwd16v3_lds_bytes<D, NW, 1, RING, VD>()), stream, args, g);
}

template <typename T, int D, int NW, int RING, int VD, int TR>
static void fill_tr(VariantInfo *v, const char *name) {
  *v = VariantInfo();
  v->func = reinterpret_cast<const void *>(&attn_fwd16_v3<T, D, NW, 1, 8, 0, 0, RING, false, true, VD, false, TR>);
  v->name = name;
  v->parallelization = NW * 32;
  v->traversal = 32;
  v->headBlock = D;
  v->threads = NW * 64;
  v->ldsBytes = fwd16v3_lds_bytes<D, NW, 1, RING, VD>();
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->causal = true;
  v->transposedInPlace = true;
  v->launch = &launch_v3_tr<T, D, NW, RING, VD, TR>;
}

// pattern: bit 0 = K transposed, bit 1 = V transposed (Q / O: any)
#define MFA_FWD16_V3_TR_BUCKET(T, TNAME, D, NW, RING, VD, GEOM)                                                                   \
  switch (pattern) {                                                                                                               \
    case 0: fill_tr<T, D, NW, RING, VD, 4>(out, "attn_fwd16v3_" TNAME "_d" #D "_" GEOM "_thr8_tr"); return true;                 \
    case 1: fill_tr<T, D, NW, RING, VD, 5>(out, "attn_fwd16v3_" TNAME "_d" #D "_" GEOM "_thr8_tr_k"); return true;               \
    case 2: fill_tr<T, D, NW, RING, VD, 6>(out, "attn_fwd16v3_" TNAME "_d" #D "_" GEOM "_thr8_tr_v"); return true;               \
    case 3: fill_tr<T, D, NW, RING, VD, 7>(out, "attn_fwd16v3_" TNAME "_d" #D "_" GEOM "_thr8_tr_kv"); return true;              \
    default: return false;                                                                                                         \
  }

} // namespace mfa
