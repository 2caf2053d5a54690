// attn_dkv16_p4.hip -- instantiations of the four-wave, 64-keys-per-wave backwardKeyValue kernel (attn_dkv16_p4.h).
#include "attn_dkv16_p4.h"
#include "launchers.h"

namespace mfa {

template <typename T, int STREAM, bool CAUSAL>
static void launch_dkv_p4(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_dkv16_p4<T, STREAM, CAUSAL>), dim3(grid.x * grid.y * grid.z), dim3(256), dkv4::LDS_BYTES, stream, args, g);
}

// row-parallel launch: the 32-row steps in `splits` pieces (SPLIT of attn_dkv16_p4.h), then the sums of the dV and dK slabs
template <typename T, int STREAM>
static void launch_dkv_p4_split(dim3 grid, uint32_t splits, float *ws, float *, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, ws, nullptr};
  hipLaunchKernelGGL((attn_dkv16_p4<T, STREAM, false, true>), dim3(grid.x * grid.y * grid.z * splits), dim3(256), dkv4::LDS_BYTES, stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.C;
  const float *dk_slabs = ws + (uint64_t)splits * rows * args.D;   // dV slabs first, then dK slabs
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dV, args.C, (const float *)ws);
  hipLaunchKernelGGL(attn_bwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g, (int)SLOT_dK, args.C, dk_slabs);
}

// `v` arrives filled by dkv16_rs_variant: block-sparse launches and causal row-parallel ones keep the role-split kernel's code objects
template <typename T, int STREAM> static void fill_dkv_p4(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_dkv16_p4<T, STREAM, false>);
  if (v->name && v->name[0]) v->siblingName = v->name;   // (arrives filled by the kernel whose split / sparse launches it keeps)
  v->name = name;
  v->siblingParallelization = v->parallelization;   // split / block-sparse launches: the role-split kernel's workgroups
  v->parallelization = 256;   // key columns per workgroup: four waves x 64
  v->traversal = 32;
  v->headBlock = dkv4::stream_bucket(STREAM);
  v->threads = 256;
  v->ldsBytes = v->ldsBytes > (uint32_t)dkv4::LDS_BYTES ? v->ldsBytes : (uint32_t)dkv4::LDS_BYTES;
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_dkv_p4<T, STREAM, false>;
  v->launchCausal = &launch_dkv_p4<T, STREAM, true>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_dkv16_p4<T, STREAM, true>);
  v->causal = true;
  v->launchSplitCausal = v->launchSplit;   // (the role-split kernel's)
  v->launchSplit = &launch_dkv_p4_split<T, STREAM>;
  v->funcSplit = reinterpret_cast<const void *>(&attn_dkv16_p4<T, STREAM, false, true>);
  v->splitParallelization = 256;
  v->splitTarget = 256;   // one workgroup per compute unit (512 registers per lane)
}

// precision: Q, K, V and dO (one 16-bit type); lprec / dprec: storage types of L and D.  The streams exist for the two
// combinations the reference's descriptors produce (+Precisions.swift:13-96): FP16 L with BF16 D (mixed-precision mode)
// and FP32 L, D.  impl >= 1000 (developer build): stream index.
bool dkv16_p4_variant(int precision, int gprecision, int lprec, int dprec, int D, int impl, VariantInfo *out) {
  if (D != 128 && D != 64) return false;
  const bool mixed = lprec == PREC_FP16 && dprec == PREC_BF16, f32 = lprec == PREC_FP32 && dprec == PREC_FP32;
  const bool d64 = D == 64;
  if (precision == PREC_FP16 && gprecision == PREC_BF16) {   // the reference's own mix: FP16 Q, K, V with BF16 dO
    if (impl == 0 && mixed && !d64) { fill_dkv_p4<_Float16, dkv4::S_F16_DOBF16_MIXED>(out, "attn_dkv16p4_f16_dObf16_d128_w4x64"); return true; }
    if (impl == 0 && f32 && !d64) { fill_dkv_p4<_Float16, dkv4::S_F16_DOBF16_F32>(out, "attn_dkv16p4_f16_dObf16_d128_w4x64_exact"); return true; }
    if (impl == 0 && mixed && d64) { fill_dkv_p4<_Float16, dkv4::S_D64_F16_DOBF16_MIXED>(out, "attn_dkv16p4_f16_dObf16_d64_w4x64"); return true; }
    if (impl == 0 && f32 && d64) { fill_dkv_p4<_Float16, dkv4::S_D64_F16_DOBF16_F32>(out, "attn_dkv16p4_f16_dObf16_d64_w4x64_exact"); return true; }
    return false;
  }
  if (precision != gprecision) return false;
  if (precision == PREC_BF16) {
    if (impl == 0 && mixed && !d64) { fill_dkv_p4<__bf16, dkv4::S_BF16_MIXED>(out, "attn_dkv16p4_bf16_d128_w4x64"); return true; }
    if (impl == 0 && f32 && !d64) { fill_dkv_p4<__bf16, dkv4::S_BF16_F32>(out, "attn_dkv16p4_bf16_d128_w4x64_exact"); return true; }
    if (impl == 0 && mixed && d64) { fill_dkv_p4<__bf16, dkv4::S_D64_BF16_MIXED>(out, "attn_dkv16p4_bf16_d64_w4x64"); return true; }
    if (impl == 0 && f32 && d64) { fill_dkv_p4<__bf16, dkv4::S_D64_BF16_F32>(out, "attn_dkv16p4_bf16_d64_w4x64_exact"); return true; }
#ifdef MFA_DEV_VARIANTS
    if (impl == 1000 + dkv4::S_BF16_MIXED_PROF && mixed && !d64) { fill_dkv_p4<__bf16, dkv4::S_BF16_MIXED_PROF>(out, "attn_dkv16p4_DEV_BF16_MIXED_PROF"); return true; }
#endif
  }
  if (precision == PREC_FP16 && impl == 0) {
    if (mixed && !d64) { fill_dkv_p4<_Float16, dkv4::S_F16_MIXED>(out, "attn_dkv16p4_f16_d128_w4x64"); return true; }
    if (f32 && !d64) { fill_dkv_p4<_Float16, dkv4::S_F16_F32>(out, "attn_dkv16p4_f16_d128_w4x64_exact"); return true; }
    if (mixed && d64) { fill_dkv_p4<_Float16, dkv4::S_D64_F16_MIXED>(out, "attn_dkv16p4_f16_d64_w4x64"); return true; }
    if (f32 && d64) { fill_dkv_p4<_Float16, dkv4::S_D64_F16_F32>(out, "attn_dkv16p4_f16_d64_w4x64_exact"); return true; }
  }
  return false;
}

} // namespace mfa
