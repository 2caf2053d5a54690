// attn_paged.hip -- instantiations of the any-head-dimension kernels (attn_paged.h): D-blocked products, accumulators paged
// through the output buffers (the reference's scheme beyond what registers hold, +Accumulate.swift:403-469).
#include "attn_paged.h"
#include "launchers.h"

namespace mfa {

template <void (*KERNEL)(const KernelArgs, const paged::Grid)>
static void launch_paged(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  paged::Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL(KERNEL, dim3(grid.x * grid.y * grid.z), dim3(256), 0, stream, args, g);
}

template <void (*KERNEL)(const KernelArgs, const paged::Grid)> static void fill_paged(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(KERNEL);
  v->name = name;
  v->parallelization = paged::BR;   // rows (forward, backwardQuery) or keys (backwardKeyValue) per workgroup
  v->traversal = paged::BC;
  v->headBlock = paged::DC;         // the head dimension is walked in chunks of 64, whatever D is
  v->threads = 256;
  v->ldsBytes = 0;                  // static LDS only
  v->cacheLeft = false;             // nothing is cached: every operand block is re-staged per traversal step
  v->cacheSecond = false;
  v->pagedAccumulators = true;
  v->causal = true;                 // causal mask, per-batch lengths and block masks are handled by the one code object
  v->sparse = true;
  v->launch = &launch_paged<KERNEL>;
}

bool paged_variant(int type, VariantInfo *out) {
  switch (type) {
    case 0: fill_paged<attn_paged_fwd>(out, "attn_paged_fwd_f32_any_d"); return true;
    case 1: fill_paged<attn_paged_dq>(out, "attn_paged_dq_f32_any_d"); return true;
    case 2: fill_paged<attn_paged_dkv>(out, "attn_paged_dkv_f32_any_d"); return true;
    default: return false;
  }
}

} // namespace mfa
