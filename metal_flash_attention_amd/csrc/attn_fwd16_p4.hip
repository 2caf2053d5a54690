// attn_fwd16_p4.hip -- instantiations of the four-wave, 64-rows-per-wave forward kernel (attn_fwd16_p4.h).
#include "attn_fwd16_p4.h"
#include "launchers.h"

namespace mfa {

// persistent form (attn_fwd16_p4p.hip): dense launches without per-batch lengths; false = not served, launch this kernel
template <typename T, bool FOLD> bool launch_p4p(dim3 grid, hipStream_t stream, const KernelArgs &args);
template <typename T, bool FOLD> const char *p4p_form(const KernelArgs &args);
template <typename T, bool FOLD> bool launch_p4p_split(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args);
template <typename T, bool FOLD> bool p4p_split_serves(const KernelArgs &args, uint32_t splits);

template <typename T, int STREAM, bool CAUSAL>
static void launch_p4(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if constexpr (STREAM == p4::S_BF16_THR8 || STREAM == p4::S_F16_THR8 || STREAM == p4::S_BF16_FOLD || STREAM == p4::S_F16_FOLD) {
    if (launch_p4p<T, p4::stream_folds(STREAM)>(grid, stream, args)) return;   // (dense and causal)
  }
  Fwd16Grid g{grid.x, grid.y, grid.z};
  const uint32_t groups = CAUSAL ? (grid.x + 1) / 2 : grid.x;   // causal: one workgroup per pair of row blocks (last - i, i)
  hipLaunchKernelGGL((attn_fwd16_p4<T, STREAM, CAUSAL>), dim3(groups * grid.y * grid.z), dim3(256), p4::LDS_BYTES, stream, args, g);
}

// column-parallel launch (few-workgroup problems: one head, long sequences): pieces of the key range, then the combine pass
template <typename T, int STREAM>
static void launch_p4_split(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, wsO, wsML};
  bool persistent = false;
  if constexpr (STREAM == p4::S_BF16_THR8 || STREAM == p4::S_F16_THR8 || STREAM == p4::S_BF16_FOLD || STREAM == p4::S_F16_FOLD)
    persistent = launch_p4p_split<T, p4::stream_folds(STREAM)>(grid, splits, wsO, wsML, stream, args);   // (round 6: the pieces on the persistent kernel)
  if (!persistent)
    hipLaunchKernelGGL((attn_fwd16_p4<T, STREAM, false, true>), dim3(grid.x * grid.y * grid.z * splits), dim3(256), p4::LDS_BYTES, stream, args, g);
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.R;
  hipLaunchKernelGGL(attn_fwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g);
}

template <typename T, int STREAM> static const char *p4_split_form(const KernelArgs &args, uint32_t splits) {
  if constexpr (STREAM == p4::S_BF16_THR8 || STREAM == p4::S_F16_THR8 || STREAM == p4::S_BF16_FOLD || STREAM == p4::S_F16_FOLD) {
    if (p4p_split_serves<T, p4::stream_folds(STREAM)>(args, splits)) return "pieces by attn_fwd16_p4p, persistent";
  }
  return "pieces by attn_fwd16_p4, one block per workgroup";
}

template <typename T, int STREAM> static void fill_p4(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16_p4<T, STREAM, false>);
  if (v->name && v->name[0]) v->siblingName = v->name;   // (arrives filled by the kernel whose split / sparse launches it keeps)
  v->name = name;
  v->parallelization = 256;
  v->traversal = 64;
  v->headBlock = 128;
  v->threads = 256;
  v->ldsBytes = v->ldsBytes > (uint32_t)p4::LDS_BYTES ? v->ldsBytes : (uint32_t)p4::LDS_BYTES;   // (siblings of the 8 x 32 kernel keep theirs)
  v->cacheLeft = true;
  v->cacheSecond = true;
  v->launch = &launch_p4<T, STREAM, false>;
  v->launchCausal = &launch_p4<T, STREAM, true>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_p4<T, STREAM, true>);
  v->causal = true;
  v->launchSplit = &launch_p4_split<T, STREAM>;   // (block-sparse launches keep the sibling of the 8 x 32 kernel)
  v->funcSplit = reinterpret_cast<const void *>(&attn_fwd16_p4<T, STREAM, false, true>);
  v->splitTarget = 256;   // one workgroup per compute unit
  v->splitParallelization = 256;   // (the pieces of a column-parallel launch are THIS kernel's: attn_fwd16_p4<..., split>, not the sibling's)
  v->splitForm = &p4_split_form<T, STREAM>;
  if constexpr (STREAM == p4::S_BF16_THR8 || STREAM == p4::S_F16_THR8 || STREAM == p4::S_BF16_FOLD || STREAM == p4::S_F16_FOLD)
    v->launchForm = &p4p_form<T, p4::stream_folds(STREAM)>;
}

template <typename T, int STREAM> static void fill_p4_dev(VariantInfo *v, const char *name) {   // dense launches only
  fill_p4<T, p4::S_BF16_THR8>(v, name);
  v->func = reinterpret_cast<const void *>(&attn_fwd16_p4<T, STREAM, false>);
  v->launch = &launch_p4<T, STREAM, false>;
  if constexpr (p4::stream_profiles(STREAM)) {   // phase clocks of causal launches too (tools/p4_prof.py --causal)
    v->launchCausal = &launch_p4<T, STREAM, true>;
    v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_p4<T, STREAM, true>);
  } else {
    v->launchCausal = nullptr; v->funcCausal = nullptr; v->causal = false;
  }
}

// Product streams: impl 0 = scale applied in fp32 (exact S; selected when the descriptor keeps the attention matrix in
// FP32 registers), impl 10 = FOLD (Q pre-multiplied by the softmax scale in the 16-bit type, the running maximum
// subtracted inside the matrix pipe; selected with lowPrecisionIntermediates, where the reference itself holds P -- and
// for FP16 also S -- in 16 bits), impl 1 = THR = 0 (the reference's rescale rule, +Softmax.swift:290-301, instead of the
// deferred one).  Everything else is a developer stream (libmfa_hip_dev.so only): MFA_FWD16_IMPL=p4:<1000 + stream index>
// for A/B placements, phase-clock stamps and timing-only ablations (tools/p4_prof.py lists the indices).
bool fwd16_p4_variant(int precision, int D, int impl, VariantInfo *out) {
  if (D != 128) return false;
  if (precision == PREC_BF16) {
    if (impl == 0) { fill_p4<__bf16, p4::S_BF16_THR8>(out, "attn_fwd16p4_bf16_d128_w4x64_thr8"); return true; }
#ifdef MFA_DEV_VARIANTS   // (no descriptor selects the THR = 0 stream: developer library only, MFA_FWD16_IMPL=p4:1)
    if (impl == 1) { fill_p4<__bf16, p4::S_BF16_THR0>(out, "attn_fwd16p4_bf16_d128_w4x64_thr0"); return true; }
#endif
    if (impl == 10) { fill_p4<__bf16, p4::S_BF16_FOLD>(out, "attn_fwd16p4_bf16_d128_w4x64_thr8_fold"); return true; }
#ifdef MFA_DEV_VARIANTS
#define MFA_P4_DEV(name) if (impl == 1000 + p4::S_##name) { fill_p4_dev<__bf16, p4::S_##name>(out, "attn_fwd16p4_DEV_" #name); return true; }
    MFA_P4_DEV_STREAM_LIST(MFA_P4_DEV)
#undef MFA_P4_DEV
#endif
  }
  if (precision == PREC_FP16) {
    if (impl == 0) { fill_p4<_Float16, p4::S_F16_THR8>(out, "attn_fwd16p4_f16_d128_w4x64_thr8"); return true; }
    if (impl == 10) { fill_p4<_Float16, p4::S_F16_FOLD>(out, "attn_fwd16p4_f16_d128_w4x64_thr8_fold"); return true; }
  }
  return false;
}

} // namespace mfa
