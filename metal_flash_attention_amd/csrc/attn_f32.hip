// attn_f32.hip -- instantiations and launchers of the FP32 production kernels (attn_f32.h).  The general kernels' launchers
// (attn_generic_{fwd,dq,dkv}.hip) hand a launch over when f32k::serves() says its operands qualify; the launch form names the
// code object that ran (mfa_attention_kernel_launch_form).
#include "attn_f32.h"
#include "launchers.h"

#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

namespace mfa {

namespace {

// the large-LDS attribute of a code object, once per (kernel, device): launches stay free of driver calls after the first one
template <typename Kernel> bool raise_lds(Kernel kernel, int bytes) {
  static std::mutex guard;
  static std::set<std::pair<const void *, int>> done;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return false;
  const std::pair<const void *, int> key(reinterpret_cast<const void *>(kernel), device);
  std::lock_guard<std::mutex> lock(guard);
  if (done.count(key)) return true;
  if (hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  done.insert(key);
  return true;
}

template <int DP> bool launch(int type, dim3 grid, hipStream_t stream, const KernelArgs &args) {
  Fwd16Grid g{grid.x, grid.y, grid.z};
  const dim3 flat(grid.x * grid.y * grid.z);
  switch (type) {
    case 0:
      if (!raise_lds(&f32k::attn_f32_fwd<DP>, f32k::lds_bytes<DP>())) return false;
      hipLaunchKernelGGL((f32k::attn_f32_fwd<DP>), flat, dim3(256), f32k::lds_bytes<DP>(), stream, args, g);
      return true;
    case 1:
#ifdef MFA_DEV_VARIANTS   // MFA_F32_PROF=1: phase clocks instead of dQ (attn_f32.h, PROF)
      if constexpr (DP == 128) {
        if (std::getenv("MFA_F32_PROF")) {
          if (!raise_lds(&f32k::attn_f32_dq<DP, true>, f32k::lds_bytes_dq<DP>())) return false;
          hipLaunchKernelGGL((f32k::attn_f32_dq<DP, true>), flat, dim3(256), f32k::lds_bytes_dq<DP>(), stream, args, g);
          return true;
        }
      }
#endif
      if (!raise_lds(&f32k::attn_f32_dq<DP>, f32k::lds_bytes_dq<DP>())) return false;
      hipLaunchKernelGGL((f32k::attn_f32_dq<DP>), flat, dim3(256), f32k::lds_bytes_dq<DP>(), stream, args, g);
      return true;
    default:
      if (!raise_lds(&f32k::attn_f32_dkv<DP>, f32k::lds_bytes_dkv<DP>())) return false;
      hipLaunchKernelGGL((f32k::attn_f32_dkv<DP>), flat, dim3(256), f32k::lds_bytes_dkv<DP>(), stream, args, g);
      return true;
  }
}

bool taken(int type, int DP, const KernelArgs &args) {
#ifdef MFA_DEV_VARIANTS   // developer builds: MFA_F32_GENERAL=1 keeps the general kernels on these launches (A/B runs)
  if (std::getenv("MFA_F32_GENERAL")) return false;
#endif
  return f32k::serves(type, DP, args);
}

}  // namespace

bool f32_launch(int type, int DP, dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if (!taken(type, DP, args)) return false;
  return DP == 64 ? launch<64>(type, grid, stream, args) : launch<128>(type, grid, stream, args);
}

const char *f32_form(int type, int DP, const KernelArgs &args) {
  if (!taken(type, DP, args)) return nullptr;
  static const char *const names[3][2] = {{"attn_f32_fwd_d64_w4x32", "attn_f32_fwd_d128_w4x32"},
                                          {"attn_f32_dq_d64_w4x32", "attn_f32_dq_d128_w4x32"},
                                          {"attn_f32_dkv_d64_w4x32", "attn_f32_dkv_d128_w4x32"}};
  return names[type][DP == 128];
}

namespace {
template <int TYPE, int DP> const char *form_or_general(const KernelArgs &args) {
  if (taken(TYPE, DP, args)) return nullptr;   // (the variant's own name: the FP32 production kernel runs)
  static const char *const general[3][2] = {
      {"attn_generic_fwd_f32mfma_d64_w4_cached (general kernel: an operand's rows are not 16-byte aligned)",
       "attn_generic_fwd_f32mfma_d128_w4_cached (general kernel: an operand's rows are not 16-byte aligned)"},
      {"attn_generic_dq_f32mfma_d64_w4_cached (general kernel: an operand's rows are not 16-byte aligned)",
       "attn_generic_dq_f32mfma_d128_w4_cached (general kernel: an operand's rows are not 16-byte aligned)"},
      {"attn_generic_dkv_f32mfma_d64_w4_cached (general kernel: an operand's rows are not 16-byte aligned)",
       "attn_generic_dkv_f32mfma_d128_w4_cached (general kernel: an operand's rows are not 16-byte aligned)"}};
  return general[TYPE][DP == 128];
}
template <int TYPE, int DP> void fill_f32(VariantInfo *v) {
  static const char *const names[3][2] = {{"attn_f32_fwd_d64_w4x32", "attn_f32_fwd_d128_w4x32"},
                                          {"attn_f32_dq_d64_w4x32", "attn_f32_dq_d128_w4x32"},
                                          {"attn_f32_dkv_d64_w4x32", "attn_f32_dkv_d128_w4x32"}};
  v->siblingName = v->name;
  v->name = names[TYPE][DP == 128];
  v->attrLdsBytes = v->ldsBytes;
  v->ldsBytes = TYPE == 0 ? f32k::lds_bytes<DP>() : TYPE == 1 ? f32k::lds_bytes_dq<DP>() : f32k::lds_bytes_dkv<DP>();
  v->launchForm = &form_or_general<TYPE, DP>;
}
}  // namespace

bool f32_variant(int type, int DP, VariantInfo *out) {
  if (DP != 64 && DP != 128) return false;
  switch (type) {
    case 0: if (DP == 64) fill_f32<0, 64>(out); else fill_f32<0, 128>(out); break;
    case 1: if (DP == 64) fill_f32<1, 64>(out); else fill_f32<1, 128>(out); break;
    default: if (DP == 64) fill_f32<2, 64>(out); else fill_f32<2, 128>(out); break;
  }
  return true;
}

}  // namespace mfa
