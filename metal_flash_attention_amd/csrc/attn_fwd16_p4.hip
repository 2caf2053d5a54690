// attn_fwd16_p4.hip -- instantiations of the four-wave, 64-rows-per-wave forward kernel (attn_fwd16_p4.h).
#include "attn_fwd16_p4.h"
#include "launchers.h"

namespace mfa {

template <typename T, int STREAM, bool CAUSAL>
static void launch_p4(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  hipLaunchKernelGGL((attn_fwd16_p4<T, STREAM, CAUSAL>), dim3(grid.x * grid.y * grid.z), dim3(256), p4::LDS_BYTES, stream, args, g);
}

template <typename T, int STREAM> static void fill_p4(VariantInfo *v, const char *name) {
  v->func = reinterpret_cast<const void *>(&attn_fwd16_p4<T, STREAM, false>);
  v->name = name;
  v->parallelization = 256;
  v->traversal = 64;
  v->headBlock = 128;
  v->threads = 256;
  v->ldsBytes = v->ldsBytes > (uint32_t)p4::LDS_BYTES ? v->ldsBytes : (uint32_t)p4::LDS_BYTES;   // (siblings of the 8 x 32 kernel keep theirs)
  v->cacheLeft = true;
  v->launch = &launch_p4<T, STREAM, false>;
  v->launchCausal = &launch_p4<T, STREAM, true>;
  v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_p4<T, STREAM, true>);
  v->causal = true;
}

// impl 0: product stream (deferred rescale THR = 8).  1: THR = 0, the reference's rule (+Softmax.swift:290-301).
// Developer streams (only reachable with MFA_DEV_VARIANTS builds): 2 = 16 exp2 per tile moved into phase B, 3 = QK
// accumulators in rotation, 4 = code placement pad.
bool fwd16_p4_variant(int precision, int D, int impl, VariantInfo *out) {
  if (D != 128) return false;
  if (precision == PREC_BF16) {
    if (impl == 0) { fill_p4<__bf16, p4::S_BF16_THR8>(out, "attn_fwd16p4_bf16_d128_w4x64_thr8"); return true; }
    if (impl == 1) { fill_p4<__bf16, p4::S_BF16_THR0>(out, "attn_fwd16p4_bf16_d128_w4x64_thr0"); return true; }
    if (impl == 2) { fill_p4<__bf16, p4::S_BF16_THR8_XE16>(out, "attn_fwd16p4_bf16_d128_w4x64_thr8_xe16"); return true; }
    if (impl == 3) { fill_p4<__bf16, p4::S_BF16_THR8_ROT>(out, "attn_fwd16p4_bf16_d128_w4x64_thr8_rot4"); return true; }
    if (impl == 4) { fill_p4<__bf16, p4::S_BF16_THR8_PAD>(out, "attn_fwd16p4_bf16_d128_w4x64_thr8_pad"); return true; }
    if (impl == 5) { fill_p4<__bf16, p4::S_BF16_THR8_PROF>(out, "attn_fwd16p4_bf16_d128_w4x64_thr8_PROF_CLOBBERS_O"); return true; }
  }
  if (precision == PREC_FP16) {
    if (impl == 0) { fill_p4<_Float16, p4::S_F16_THR8>(out, "attn_fwd16p4_f16_d128_w4x64_thr8"); return true; }
  }
  return false;
}

} // namespace mfa
