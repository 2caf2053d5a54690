// attn_dq16_p4.hip -- instantiations of the four-wave, 64-rows-per-wave backwardQuery kernel (attn_dq16_p4.h).
#include "attn_dq16_p4.h"
#include "launchers.h"

namespace mfa {

template <typename T, int STREAM, bool CAUSAL, typename TG>
static void launch_dq_p4(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dq16_p4<T, STREAM, CAUSAL, TG>), dim3(grid.x * grid.y * grid.z), dim3(256), dq4::LDS_BYTES, stream, args, g);
}

// column-parallel launch: the key tiles in `splits` pieces (SPLIT of attn_dq16_p4.h), then the sum of the slabs
template <typename T, int STREAM, typename TG>
static void launch_dq_p4_split(dim3 grid, uint32_t splits, float *ws, float *, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, ws, nullptr};
  hipLaunchKernelGGL((attn_dq16_p4<T, STREAM, false, TG, true>), dim3(grid.x * grid.y * grid.z * splits), dim3(256), dq4::LDS_BYTES, stream,
                     args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.R;
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dQ, args.R, (const float *)ws);
}

// `v` arrives filled by dq16_variant (eight waves x 32 rows, the same 256 rows per workgroup): block-sparse launches and
// causal column-parallel ones keep that kernel's code objects
template <typename T, int STREAM, typename TG = T> static void fill_dq_p4(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_dq16_p4<T, STREAM, false, TG>);
  if (v->name && v->name[0]) v->siblingName = v->name;   // (arrives filled by the kernel whose split / sparse launches it keeps)
  v->name = name;
  v->parallelization = 256;
  v->traversal = 64;
  v->headBlock = dq4::stream_bucket(STREAM);
  v->threads = 256;
  v->ldsBytes = v->ldsBytes > (uint32_t)dq4::LDS_BYTES ? v->ldsBytes : (uint32_t)dq4::LDS_BYTES;
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_dq_p4<T, STREAM, false, TG>;
  v->launchCausal = &launch_dq_p4<T, STREAM, true, TG>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dq16_p4<T, STREAM, true, TG>);
  v->causal = true;
  v->launchSplitCausal = v->launchSplit;   // (the eight-wave kernel's)
  v->launchSplit = &launch_dq_p4_split<T, STREAM, TG>;
  v->funcSplit = reinterpret_cast<const void *>(&attn_dq16_p4<T, STREAM, false, TG, true>);
  v->splitParallelization = 256;
  v->splitTarget = 256;   // one workgroup per compute unit (512 registers per lane)
}

// impl 0: Q as stored, softmax scale in fp32 (descriptors that keep the attention matrix in FP32 registers); impl 10: Q
// pre-multiplied by the scale in the 16-bit type (lowPrecisionIntermediates, like the forward FOLD stream)
bool dq16_p4_variant(int precision, int gprecision, int D, int impl, VariantInfo *out) {
  if (D != 128 && D != 64) return false;
  const bool d64 = D == 64;
  if (precision == PREC_FP16 && gprecision == PREC_BF16) {   // the reference's own mix: FP16 Q, K, V with BF16 dO
    if (impl == 0 && !d64) { fill_dq_p4<_Float16, dq4::S_F16_EXACT, __bf16>(out, "attn_dq16p4_f16_dObf16_d128_w4x64_exact"); return true; }
    if (impl == 10 && !d64) { fill_dq_p4<_Float16, dq4::S_F16_FOLD, __bf16>(out, "attn_dq16p4_f16_dObf16_d128_w4x64"); return true; }
    if (impl == 0 && d64) { fill_dq_p4<_Float16, dq4::S_D64_F16_EXACT, __bf16>(out, "attn_dq16p4_f16_dObf16_d64_w4x64_exact"); return true; }
    if (impl == 10 && d64) { fill_dq_p4<_Float16, dq4::S_D64_F16_FOLD, __bf16>(out, "attn_dq16p4_f16_dObf16_d64_w4x64"); return true; }
    return false;
  }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) {
    if (impl == 0 && !d64) { fill_dq_p4<__bf16, dq4::S_BF16_EXACT>(out, "attn_dq16p4_bf16_d128_w4x64_exact"); return true; }
    if (impl == 10 && !d64) { fill_dq_p4<__bf16, dq4::S_BF16_FOLD>(out, "attn_dq16p4_bf16_d128_w4x64"); return true; }
    if (impl == 0 && d64) { fill_dq_p4<__bf16, dq4::S_D64_BF16_EXACT>(out, "attn_dq16p4_bf16_d64_w4x64_exact"); return true; }
    if (impl == 10 && d64) { fill_dq_p4<__bf16, dq4::S_D64_BF16_FOLD>(out, "attn_dq16p4_bf16_d64_w4x64"); return true; }
#ifdef MFA_DEV_VARIANTS
#define MFA_DQ4_DEV(name) if (impl == 1000 + dq4::S_##name && D == dq4::stream_bucket(dq4::S_##name)) { fill_dq_p4<__bf16, dq4::S_##name>(out, "attn_dq16p4_DEV_" #name); return true; }
    MFA_DQ4_DEV_STREAM_LIST(MFA_DQ4_DEV)
#undef MFA_DQ4_DEV
#endif
  }
  if (precision == PREC_FP16) {
    if (impl == 0 && !d64) { fill_dq_p4<_Float16, dq4::S_F16_EXACT>(out, "attn_dq16p4_f16_d128_w4x64_exact"); return true; }
    if (impl == 10 && !d64) { fill_dq_p4<_Float16, dq4::S_F16_FOLD>(out, "attn_dq16p4_f16_d128_w4x64"); return true; }
    if (impl == 0 && d64) { fill_dq_p4<_Float16, dq4::S_D64_F16_EXACT>(out, "attn_dq16p4_f16_d64_w4x64_exact"); return true; }
    if (impl == 10 && d64) { fill_dq_p4<_Float16, dq4::S_D64_F16_FOLD>(out, "attn_dq16p4_f16_d64_w4x64"); return true; }
  }
  return false;
}

} // namespace mfa
