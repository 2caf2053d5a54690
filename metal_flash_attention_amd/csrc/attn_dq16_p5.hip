// attn_dq16_p5.hip -- instantiations of the role-split, 64-rows-per-pair backwardQuery kernel of the head-dimension buckets
// 160 / 192 / 256 (attn_dq16_p5.h).
#include "attn_dq16_p5.h"
#include "launchers.h"
#include <cstdlib>
#include <cstring>

namespace mfa {

template <typename T, int STREAM, bool CAUSAL, typename TG>
static void launch_dq_p5(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  constexpr int LDS = dq5::lds_bytes(dq5::stream_bucket(STREAM));
  hipLaunchKernelGGL((attn_dq16_p5<T, STREAM, CAUSAL, TG>), dim3(grid.x * grid.y * grid.z), dim3(256), LDS, stream, args, g);
}

// column-parallel launch (round 6): the 32-key blocks in `splits` pieces (SPLIT of attn_dq16_p5.h), then the sum of the slabs
template <typename T, int STREAM, typename TG>
static void launch_dq_p5_split(dim3 grid, uint32_t splits, float *ws, float *, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, ws, nullptr};
  constexpr int LDS = dq5::lds_bytes(dq5::stream_bucket(STREAM));
  hipLaunchKernelGGL((attn_dq16_p5<T, STREAM, false, TG, true>), dim3(grid.x * grid.y * grid.z * splits), dim3(256), LDS, stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.R;
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dQ, args.R, (const float *)ws);
}

// `v` arrives filled by dq16_variant*: block-sparse and CAUSAL column-parallel launches keep the 32-row-wave kernel's code objects
template <typename T, int STREAM, typename TG = T> static void fill_dq_p5(VariantInfo *v, const char *name) {
  constexpr int LDS = dq5::lds_bytes(dq5::stream_bucket(STREAM));
  v->func = reinterpret_cast<const void *>(&attn_dq16_p5<T, STREAM, false, TG>);
  if (v->name && v->name[0]) v->siblingName = v->name;   // (arrives filled by the kernel whose split / sparse launches it keeps)
  v->name = name;
  v->siblingParallelization = v->parallelization;
  v->parallelization = dq5::WGROWS;   // rows per workgroup: two wave pairs x 64
  v->traversal = 32;
  v->headBlock = dq5::stream_bucket(STREAM);
  v->threads = 256;
  v->ldsBytes = v->ldsBytes > (uint32_t)LDS ? v->ldsBytes : (uint32_t)LDS;
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_dq_p5<T, STREAM, false, TG>;
  v->launchCausal = &launch_dq_p5<T, STREAM, true, TG>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dq16_p5<T, STREAM, true, TG>);
  v->causal = true;
  if constexpr (!dq5::stream_profiles(STREAM)) {
    v->launchSplitCausal = v->launchSplit;   // (the sibling's)
    v->launchSplit = &launch_dq_p5_split<T, STREAM, TG>;
    v->funcSplit = reinterpret_cast<const void *>(&attn_dq16_p5<T, STREAM, false, TG, true>);
    v->splitParallelization = dq5::WGROWS;
    v->splitTarget = 256;   // one workgroup per compute unit (512 registers per lane)
  }
}

// impl 0: Q as stored, softmax scale in fp32 (descriptors that keep the attention matrix in FP32 registers); impl 10: Q
// pre-multiplied by the scale in the 16-bit type (lowPrecisionIntermediates)
bool dq16_p5_variant(int precision, int gprecision, int D, int impl, VariantInfo *out) {
#ifdef MFA_DEV_VARIANTS   // developer library: MFA_BWD5_PROF=1 -> the clock-stamping streams (tools/bwd5_prof.py)
  if (std::getenv("MFA_BWD5_PROF") && impl == 10 && precision == PREC_BF16 && gprecision == PREC_BF16) {
    if (D == 256) { fill_dq_p5<__bf16, dq5::S_D256_BF16_FOLD_PROF>(out, "attn_dq16p5_DEV_D256_BF16_FOLD_PROF"); return true; }
    if (D == 160) { fill_dq_p5<__bf16, dq5::S_D160_BF16_FOLD_PROF>(out, "attn_dq16p5_DEV_D160_BF16_FOLD_PROF"); return true; }
  }
#endif
#ifdef MFA_DEV_VARIANTS   // MFA_DQ5_DEV_STREAM=<name>: a timing-only ablation of the D = 256 BF16 stream (tools/dq5gen.py DEV_ABLATIONS)
  if (const char *e = std::getenv("MFA_DQ5_DEV_STREAM")) {
    if (impl == 10 && precision == PREC_BF16 && gprecision == PREC_BF16 && D == 256) {
#define MFA_DQ5_DEV_PICK(name) \
      if (std::strcmp(e, #name) == 0) { fill_dq_p5<__bf16, dq5::S_##name>(out, "attn_dq16p5_DEV_" #name); return true; }
      MFA_DQ5_DEV_ABL_LIST(MFA_DQ5_DEV_PICK)
#undef MFA_DQ5_DEV_PICK
    }
  }
#endif
  if (impl != 0 && impl != 10) return false;
  const bool fold = impl == 10;
#define MFA_DQ5_PICK(DD)                                                                                                                       \
  if (D == DD) {                                                                                                                               \
    if (precision == PREC_FP16 && gprecision == PREC_BF16) {                                                                                   \
      if (fold) fill_dq_p5<_Float16, dq5::S_D##DD##_F16_FOLD, __bf16>(out, "attn_dq16p5_f16_dObf16_d" #DD "_p2x64");                           \
      else fill_dq_p5<_Float16, dq5::S_D##DD##_F16_EXACT, __bf16>(out, "attn_dq16p5_f16_dObf16_d" #DD "_p2x64_exact");                         \
      return true;                                                                                                                             \
    }                                                                                                                                          \
    if (precision != gprecision) return false;                                                                                                 \
    if (precision == PREC_BF16) {                                                                                                              \
      if (fold) fill_dq_p5<__bf16, dq5::S_D##DD##_BF16_FOLD>(out, "attn_dq16p5_bf16_d" #DD "_p2x64");                                          \
      else fill_dq_p5<__bf16, dq5::S_D##DD##_BF16_EXACT>(out, "attn_dq16p5_bf16_d" #DD "_p2x64_exact");                                        \
      return true;                                                                                                                             \
    }                                                                                                                                          \
    if (precision == PREC_FP16) {                                                                                                              \
      if (fold) fill_dq_p5<_Float16, dq5::S_D##DD##_F16_FOLD>(out, "attn_dq16p5_f16_d" #DD "_p2x64");                                          \
      else fill_dq_p5<_Float16, dq5::S_D##DD##_F16_EXACT>(out, "attn_dq16p5_f16_d" #DD "_p2x64_exact");                                        \
      return true;                                                                                                                             \
    }                                                                                                                                          \
    return false;                                                                                                                              \
  }
  MFA_DQ5_PICK(160)
  MFA_DQ5_PICK(192)
  MFA_DQ5_PICK(256)
#undef MFA_DQ5_PICK
  return false;
}

} // namespace mfa
