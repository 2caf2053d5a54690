// attn_dkv16_p5.hip -- instantiations of the role-split, 64-keys-per-wave backwardKeyValue kernel of the head-dimension buckets
// 160 / 192 / 256 (attn_dkv16_p5.h).
#include "attn_dkv16_p5.h"
#include "launchers.h"
#include <cstdlib>

namespace mfa {

template <typename T, int STREAM, bool CAUSAL>
static void launch_dkv_p5(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  constexpr int LDS = dkv5::lds_bytes(dkv5::stream_bucket(STREAM));
  hipLaunchKernelGGL((attn_dkv16_p5<T, STREAM, CAUSAL>), dim3(grid.x * grid.y * grid.z), dim3(256), LDS, stream, args, g);
}

// row-parallel launch (round 6): the 32-row blocks in `splits` pieces (SPLIT of attn_dkv16_p5.h), then the sums of the dV and dK slabs
template <typename T, int STREAM>
static void launch_dkv_p5_split(dim3 grid, uint32_t splits, float *ws, float *, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, ws, nullptr};
  constexpr int LDS = dkv5::lds_bytes(dkv5::stream_bucket(STREAM));
  hipLaunchKernelGGL((attn_dkv16_p5<T, STREAM, false, true>), dim3(grid.x * grid.y * grid.z * splits), dim3(256), LDS, stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.C;
  const float *dk_slabs = ws + (uint64_t)splits * rows * args.D;   // dV slabs first, then dK slabs
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dV, args.C, (const float *)ws);
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dK, args.C, dk_slabs);
}

// `v` arrives filled by dkv16_rs_variant*: block-sparse and CAUSAL row-parallel launches keep the 32-key role-split kernel's code objects
template <typename T, int STREAM> static void fill_dkv_p5(VariantInfo *v, const char *name) {
  constexpr int LDS = dkv5::lds_bytes(dkv5::stream_bucket(STREAM));
  v->func = reinterpret_cast<const void *>(&attn_dkv16_p5<T, STREAM, false>);
  if (v->name && v->name[0]) v->siblingName = v->name;   // (arrives filled by the kernel whose split / sparse launches it keeps)
  v->name = name;
  v->siblingParallelization = v->parallelization;   // split / block-sparse launches: the 32-key role-split kernel's workgroups
  v->parallelization = dkv5::WGKEYS;   // key columns per workgroup: two wave pairs x 64
  v->traversal = 32;
  v->headBlock = dkv5::stream_bucket(STREAM);
  v->threads = 256;
  v->ldsBytes = v->ldsBytes > (uint32_t)LDS ? v->ldsBytes : (uint32_t)LDS;
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_dkv_p5<T, STREAM, false>;
  v->launchCausal = &launch_dkv_p5<T, STREAM, true>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dkv16_p5<T, STREAM, true>);
  v->causal = true;
  if constexpr (!dkv5::stream_profiles(STREAM)) {
    v->launchSplitCausal = v->launchSplit;   // (the sibling's)
    v->launchSplit = &launch_dkv_p5_split<T, STREAM>;
    v->funcSplit = reinterpret_cast<const void *>(&attn_dkv16_p5<T, STREAM, false, true>);
    v->splitParallelization = dkv5::WGKEYS;
    v->splitTarget = 256;   // one workgroup per compute unit (512 registers per lane)
  }
}

// precision: Q, K, V; gprecision: dO; lprec / dprec: storage types of L and D.  The streams exist for the two combinations the
// reference's descriptors produce (+Precisions.swift:13-96): FP16 L with BF16 D (mixed-precision mode) and FP32 L, D.
bool dkv16_p5_variant(int precision, int gprecision, int lprec, int dprec, int D, VariantInfo *out) {
  const bool mixed = lprec == PREC_FP16 && dprec == PREC_BF16, f32 = lprec == PREC_FP32 && dprec == PREC_FP32;
  if (!mixed && !f32) return false;
#ifdef MFA_DEV_VARIANTS   // developer library: MFA_BWD5_PROF=1 -> the clock-stamping streams (tools/bwd5_prof.py)
  if (std::getenv("MFA_BWD5_PROF") && mixed && precision == PREC_BF16 && gprecision == PREC_BF16) {
    if (D == 256) { fill_dkv_p5<__bf16, dkv5::S_D256_BF16_MIXED_PROF>(out, "attn_dkv16p5_DEV_D256_BF16_MIXED_PROF"); return true; }
    if (D == 160) { fill_dkv_p5<__bf16, dkv5::S_D160_BF16_MIXED_PROF>(out, "attn_dkv16p5_DEV_D160_BF16_MIXED_PROF"); return true; }
  }
#endif
#define MFA_DKV5_PICK(DD)                                                                                                                      \
  if (D == DD) {                                                                                                                               \
    if (precision == PREC_FP16 && gprecision == PREC_BF16) {                                                                                   \
      if (mixed) fill_dkv_p5<_Float16, dkv5::S_D##DD##_F16_DOBF16_MIXED>(out, "attn_dkv16p5_f16_dObf16_d" #DD "_p2x64");                       \
      else fill_dkv_p5<_Float16, dkv5::S_D##DD##_F16_DOBF16_F32>(out, "attn_dkv16p5_f16_dObf16_d" #DD "_p2x64_exact");                         \
      return true;                                                                                                                             \
    }                                                                                                                                          \
    if (precision != gprecision) return false;                                                                                                 \
    if (precision == PREC_BF16) {                                                                                                              \
      if (mixed) fill_dkv_p5<__bf16, dkv5::S_D##DD##_BF16_MIXED>(out, "attn_dkv16p5_bf16_d" #DD "_p2x64");                                     \
      else fill_dkv_p5<__bf16, dkv5::S_D##DD##_BF16_F32>(out, "attn_dkv16p5_bf16_d" #DD "_p2x64_exact");                                       \
      return true;                                                                                                                             \
    }                                                                                                                                          \
    if (precision == PREC_FP16) {                                                                                                              \
      if (mixed) fill_dkv_p5<_Float16, dkv5::S_D##DD##_F16_MIXED>(out, "attn_dkv16p5_f16_d" #DD "_p2x64");                                     \
      else fill_dkv_p5<_Float16, dkv5::S_D##DD##_F16_F32>(out, "attn_dkv16p5_f16_d" #DD "_p2x64_exact");                                       \
      return true;                                                                                                                             \
    }                                                                                                                                          \
    return false;                                                                                                                              \
  }
  MFA_DKV5_PICK(160)
  MFA_DKV5_PICK(192)
  MFA_DKV5_PICK(256)
#undef MFA_DKV5_PICK
  return false;
}

} // namespace mfa
