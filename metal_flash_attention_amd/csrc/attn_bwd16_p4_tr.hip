// attn_bwd16_p4_tr.hip -- launchers of the backward kernels that read transposed operands in place (attn_dq16_p4_tr.h: K^T / V^T;
// attn_dkv16_p4_tr.h: Q^T / dO^T).  Reached from mfa_attention_kernel_launch / _time / _launch_form when a transposed backward
// launch carries no workspace (with one, the re-layout path runs: the caller chose it); MFA_BWD16_TR=0 is the developer
// library's A/B knob; false / nullptr = the launch is not one these kernels take (the general kernel serves it).
#include <mutex>
#include <set>
#include <utility>
#include "attn_dq16_p4_tr.h"
#include "attn_dkv16_p4_tr.h"
#include "launchers.h"

namespace mfa {

static bool rows_aligned(const OperandView &v) {
  return ((reinterpret_cast<uintptr_t>(v.ptr) | (uint64_t)v.ld * 2 | (uint64_t)v.headStride * 2 | (uint64_t)v.batchStride * 2) & 15) == 0;
}

// the large-LDS attribute of a code object, once per (kernel, device): launches stay free of driver calls after the first one
// (what a hipGraph capture needs; the product library keeps such a mask per kernel object, ensure_lds_attribute in mfa_kernel.hip)
template <typename Kernel> static bool raise_lds(Kernel kernel, int bytes) {
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

template <typename T, int STREAM, typename TG = T> static bool launch_dq_tr(const KernelArgs &a, uint32_t heads, uint32_t batches, hipStream_t stream) {
  const uint32_t blocks = (a.R + 255) / 256;
  Fwd16Grid g{blocks, heads, batches};
  if (a.causal) {
    if (!raise_lds(&attn_dq16_p4_tr<T, STREAM, true, TG>, dq4::LDS_BYTES)) return false;
    hipLaunchKernelGGL((attn_dq16_p4_tr<T, STREAM, true, TG>), dim3(blocks * heads * batches), dim3(256), dq4::LDS_BYTES, stream, a, g);
  } else {
    if (!raise_lds(&attn_dq16_p4_tr<T, STREAM, false, TG>, dq4::LDS_BYTES)) return false;
    hipLaunchKernelGGL((attn_dq16_p4_tr<T, STREAM, false, TG>), dim3(blocks * heads * batches), dim3(256), dq4::LDS_BYTES, stream, a, g);
  }
  return true;
}

template <typename T, int STREAM> static bool launch_dkv_tr(const KernelArgs &a, uint32_t heads, uint32_t batches, hipStream_t stream) {
  const uint32_t blocks = (a.C + 255) / 256;
  Fwd16Grid g{blocks, heads, batches};
  if (a.causal) {
    if (!raise_lds(&attn_dkv16_p4_tr<T, STREAM, true>, dkv4::LDS_BYTES)) return false;
    hipLaunchKernelGGL((attn_dkv16_p4_tr<T, STREAM, true>), dim3(blocks * heads * batches), dim3(256), dkv4::LDS_BYTES, stream, a, g);
  } else {
    if (!raise_lds(&attn_dkv16_p4_tr<T, STREAM, false>, dkv4::LDS_BYTES)) return false;
    hipLaunchKernelGGL((attn_dkv16_p4_tr<T, STREAM, false>), dim3(blocks * heads * batches), dim3(256), dkv4::LDS_BYTES, stream, a, g);
  }
  return true;
}

// what the in-place kernels take (their headers): 16-bit operands of one type (dO may be BF16 next to FP16), 64 < D <= 128, no
// per-batch lengths, no block mask; backwardQuery: K^T and V^T in whole 64-key tiles of aligned rows; backwardKeyValue: Q^T and dO^T
// in whole 32-row steps of aligned rows, L / D stored as the reference stores them (FP16 + BF16, or both FP32)
static bool takes(int type, const KernelArgs &a) {
  const int p = a.op[SLOT_Q].precision, pg = a.op[SLOT_dO].precision;
  const bool gmix = p == PREC_FP16 && pg == PREC_BF16;
  if (p == PREC_FP32 || a.op[SLOT_K].precision != p || a.op[SLOT_V].precision != p || (pg != p && !gmix)) return false;
  if (a.rowLen || a.colLen || a.mask || a.D <= 64 || a.D > 128 || a.D % 8) return false;
  if (a.causal && a.C < a.R) return false;
  if (type == 1) {
    if (!a.op[SLOT_K].transposed || !a.op[SLOT_V].transposed || a.C % 64 != 0) return false;
    return rows_aligned(a.op[SLOT_K]) && rows_aligned(a.op[SLOT_V]);
  }
  if (!a.op[SLOT_Q].transposed || !a.op[SLOT_dO].transposed || a.R % 32 != 0) return false;
  if (!rows_aligned(a.op[SLOT_Q]) || !rows_aligned(a.op[SLOT_dO])) return false;
  const int lp = a.op[SLOT_L].precision, dp = a.op[SLOT_D].precision;
  return (lp == PREC_FP16 && dp == PREC_BF16) || (lp == PREC_FP32 && dp == PREC_FP32);
}

// the name a launch reports (mfa_attention_kernel_launch_form), nullptr = not a launch these kernels take
const char *bwd16_p4_tr_form(int type, const KernelArgs &a) {
  if (!takes(type, a)) return nullptr;
  return type == 1 ? "attn_dq16_p4_tr (four waves x 64 rows, hand-placed stream on transposed K / V in place, no workspace)"
                   : "attn_dkv16_p4_tr (four waves x 64 keys, hand-placed stream on transposed Q / dO in place, no workspace)";
}

// type: 1 = backwardQuery, 2 = backwardKeyValue (mfa_kernel_type); fold: the descriptor keeps the attention matrix in 16-bit registers
bool bwd16_p4_tr_launch(int type, const KernelArgs &a, uint32_t heads, uint32_t batches, hipStream_t stream, bool fold) {
  if (!takes(type, a)) return false;
  const int p = a.op[SLOT_Q].precision;
  const bool gmix = p == PREC_FP16 && a.op[SLOT_dO].precision == PREC_BF16;   // the reference's own mix: FP16 Q, K, V with BF16 dO (+Precisions.swift:13-17)
  if (type == 1) {
    if (p == PREC_BF16) return fold ? launch_dq_tr<__bf16, dq4tr::S_BF16_FOLD_TR>(a, heads, batches, stream) : launch_dq_tr<__bf16, dq4tr::S_BF16_EXACT_TR>(a, heads, batches, stream);
    if (gmix) return fold ? launch_dq_tr<_Float16, dq4tr::S_F16_FOLD_TR, __bf16>(a, heads, batches, stream) : launch_dq_tr<_Float16, dq4tr::S_F16_EXACT_TR, __bf16>(a, heads, batches, stream);
    return fold ? launch_dq_tr<_Float16, dq4tr::S_F16_FOLD_TR>(a, heads, batches, stream) : launch_dq_tr<_Float16, dq4tr::S_F16_EXACT_TR>(a, heads, batches, stream);
  }
  const bool mixed = a.op[SLOT_L].precision == PREC_FP16;
  if (p == PREC_BF16) return mixed ? launch_dkv_tr<__bf16, dkv4tr::S_BF16_MIXED_TR>(a, heads, batches, stream) : launch_dkv_tr<__bf16, dkv4tr::S_BF16_F32_TR>(a, heads, batches, stream);
  if (gmix) return mixed ? launch_dkv_tr<_Float16, dkv4tr::S_F16_DOBF16_MIXED_TR>(a, heads, batches, stream) : launch_dkv_tr<_Float16, dkv4tr::S_F16_DOBF16_F32_TR>(a, heads, batches, stream);
  return mixed ? launch_dkv_tr<_Float16, dkv4tr::S_F16_MIXED_TR>(a, heads, batches, stream) : launch_dkv_tr<_Float16, dkv4tr::S_F16_F32_TR>(a, heads, batches, stream);
}

} // namespace mfa
