// attn_fwd16_p5.hip -- instantiations of the four-wave, 64-rows-per-wave forward kernel for 128 < D <= 256 (attn_fwd16_p5.h).
#include "attn_fwd16_p5.h"
#include "launchers.h"

namespace mfa {

template <typename T, int STREAM, bool CAUSAL>
static void launch_p5(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  const uint32_t groups = CAUSAL ? (grid.x + 1) / 2 : grid.x;   // causal: one workgroup per pair of row blocks (last - i, i)
  hipLaunchKernelGGL((attn_fwd16_p5<T, STREAM, CAUSAL>), dim3(groups * grid.y * grid.z), dim3(256), p5::LDS_BYTES, stream, args, g);
}

// column-parallel launch (few-workgroup problems: one head, long sequences): pieces of the key range, then the combine pass
template <typename T, int STREAM>
static void launch_p5_split(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, wsO, wsML};
  hipLaunchKernelGGL((attn_fwd16_p5<T, STREAM, false, true>), dim3(grid.x * grid.y * grid.z * splits), dim3(256), p5::LDS_BYTES, stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.R;
  hipLaunchKernelGGL(attn_fwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g);
}

// `v` arrives filled by fwd16_v3_variant (D = 256: four waves x 32 rows): block-sparse launches keep its code objects
template <typename T, int STREAM> static void fill_p5(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16_p5<T, STREAM, false>);
  if (v->name && v->name[0]) v->siblingName = v->name;   // (arrives filled by the kernel whose split / sparse launches it keeps)
  v->name = name;
  v->siblingParallelization = v->parallelization;
  v->parallelization = 256;
  v->traversal = 32;
  v->headBlock = p5::stream_bucket(STREAM);
  v->threads = 256;
  v->ldsBytes = v->ldsBytes > (uint32_t)p5::LDS_BYTES ? v->ldsBytes : (uint32_t)p5::LDS_BYTES;
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_p5<T, STREAM, false>;
  v->launchCausal = &launch_p5<T, STREAM, true>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_p5<T, STREAM, true>);
  v->causal = true;
  v->launchSplit = &launch_p5_split<T, STREAM>;
  v->funcSplit = reinterpret_cast<const void *>(&attn_fwd16_p5<T, STREAM, false, true>);
  v->splitParallelization = 256;
  v->splitTarget = 256;   // one workgroup per compute unit
}

// impl 0 = scale applied in fp32; impl 10 = FOLD (see attn_fwd16_p4.hip); 1000 + stream index: developer streams
bool fwd16_p5_variant(int precision, int D, int impl, VariantInfo *out) {
  if (precision == PREC_BF16) {
    if (D == 256 && impl == 0) { fill_p5<__bf16, p5::S_BF16_THR8>(out, "attn_fwd16p5_bf16_d256_w4x64_thr8"); return true; }
    if (D == 256 && impl == 10) { fill_p5<__bf16, p5::S_BF16_FOLD>(out, "attn_fwd16p5_bf16_d256_w4x64_thr8_fold"); return true; }
    if (D == 192 && impl == 0) { fill_p5<__bf16, p5::S_D192_BF16_THR8>(out, "attn_fwd16p5_bf16_d192_w4x64_thr8"); return true; }
    if (D == 192 && impl == 10) { fill_p5<__bf16, p5::S_D192_BF16_FOLD>(out, "attn_fwd16p5_bf16_d192_w4x64_thr8_fold"); return true; }
    if (D == 160 && impl == 0) { fill_p5<__bf16, p5::S_D160_BF16_THR8>(out, "attn_fwd16p5_bf16_d160_w4x64_thr8"); return true; }
    if (D == 160 && impl == 10) { fill_p5<__bf16, p5::S_D160_BF16_FOLD>(out, "attn_fwd16p5_bf16_d160_w4x64_thr8_fold"); return true; }
#ifdef MFA_DEV_VARIANTS
    if (D == 128 && impl == 10) { fill_p5<__bf16, p5::S_D128_BF16_FOLD>(out, "attn_fwd16p5_DEV_bf16_d128_w4x64_thr8_fold"); return true; }
    if (D == 256 && impl == 1000 + p5::S_BF16_FOLD_PROF) { fill_p5<__bf16, p5::S_BF16_FOLD_PROF>(out, "attn_fwd16p5_DEV_BF16_FOLD_PROF"); return true; }
#endif
  }
  if (precision == PREC_FP16) {
    if (D == 256 && impl == 0) { fill_p5<_Float16, p5::S_F16_THR8>(out, "attn_fwd16p5_f16_d256_w4x64_thr8"); return true; }
    if (D == 256 && impl == 10) { fill_p5<_Float16, p5::S_F16_FOLD>(out, "attn_fwd16p5_f16_d256_w4x64_thr8_fold"); return true; }
    if (D == 192 && impl == 0) { fill_p5<_Float16, p5::S_D192_F16_THR8>(out, "attn_fwd16p5_f16_d192_w4x64_thr8"); return true; }
    if (D == 192 && impl == 10) { fill_p5<_Float16, p5::S_D192_F16_FOLD>(out, "attn_fwd16p5_f16_d192_w4x64_thr8_fold"); return true; }
    if (D == 160 && impl == 0) { fill_p5<_Float16, p5::S_D160_F16_THR8>(out, "attn_fwd16p5_f16_d160_w4x64_thr8"); return true; }
    if (D == 160 && impl == 10) { fill_p5<_Float16, p5::S_D160_F16_FOLD>(out, "attn_fwd16p5_f16_d160_w4x64_thr8_fold"); return true; }
  }
  return false;
}

} // namespace mfa
