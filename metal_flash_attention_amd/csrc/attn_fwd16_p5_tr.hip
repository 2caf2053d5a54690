// attn_fwd16_p5_tr.hip -- DEVELOPER BUILD ONLY: launcher of the hand-placed forward kernel for transposed K and V at the
// head-dimension buckets 160 / 192 / 256 (attn_fwd16_p5_tr.h).  Reached from mfa_attention_kernel_launch when the developer
// library runs with MFA_FWD16_P5_TR=1 (the product library launches the 8 x 32 kernel's transposed code object there);
// false = not a launch this kernel takes.
#include "attn_fwd16_p5_tr.h"
#include "launchers.h"

namespace mfa {

static bool rows_aligned16(const OperandView &v) {
  return ((reinterpret_cast<uintptr_t>(v.ptr) | (uint64_t)v.ld * 2 | (uint64_t)v.headStride * 2 | (uint64_t)v.batchStride * 2) & 15) == 0;
}

template <typename T, int STREAM> static bool launch_p5_tr(const KernelArgs &a, uint32_t heads, uint32_t batches, hipStream_t stream) {
  const uint32_t blocks = (a.R + 255) / 256;
  Fwd16Grid g{blocks, heads, batches};
  auto raise = [](const void *f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, p5::LDS_BYTES) == hipSuccess; };
  if (a.causal) {
    if (!raise(reinterpret_cast<const void *>(&attn_fwd16_p5_tr<T, STREAM, true>))) return false;
    hipLaunchKernelGGL((attn_fwd16_p5_tr<T, STREAM, true>), dim3((blocks + 1) / 2 * heads * batches), dim3(256), p5::LDS_BYTES, stream, a, g);
  } else {
    if (!raise(reinterpret_cast<const void *>(&attn_fwd16_p5_tr<T, STREAM, false>))) return false;
    hipLaunchKernelGGL((attn_fwd16_p5_tr<T, STREAM, false>), dim3(blocks * heads * batches), dim3(256), p5::LDS_BYTES, stream, a, g);
  }
  return true;
}

// fold: the descriptor keeps the attention matrix in 16-bit registers (Q pre-multiplied by the softmax scale in the 16-bit type)
bool fwd16_p5_tr_launch(const KernelArgs &a, uint32_t heads, uint32_t batches, hipStream_t stream, bool fold) {
  const int p = a.op[SLOT_Q].precision;
  if (p == PREC_FP32 || a.op[SLOT_K].precision != p || a.op[SLOT_V].precision != p) return false;
  if (a.rowLen || a.colLen || a.mask || a.D <= 128 || a.D > 256 || a.D % 8 || a.C % 32 || a.C == 0) return false;
  if (a.causal && a.C < a.R) return false;
  if (!a.op[SLOT_K].transposed || !a.op[SLOT_V].transposed) return false;
  if (!rows_aligned16(a.op[SLOT_K]) || !rows_aligned16(a.op[SLOT_V])) return false;
  if (!a.op[SLOT_Q].transposed && !rows_aligned16(a.op[SLOT_Q])) return false;
  const int bucket = a.D <= 160 ? 160 : a.D <= 192 ? 192 : 256;
#define MFA_P5TR_PICK(T, TN)                                                                                                  \
  switch (bucket) {                                                                                                           \
    case 160: return fold ? launch_p5_tr<T, p5tr::S_D160_##TN##_FOLD_TR>(a, heads, batches, stream)                            \
                          : launch_p5_tr<T, p5tr::S_D160_##TN##_THR8_TR>(a, heads, batches, stream);                           \
    case 192: return fold ? launch_p5_tr<T, p5tr::S_D192_##TN##_FOLD_TR>(a, heads, batches, stream)                            \
                          : launch_p5_tr<T, p5tr::S_D192_##TN##_THR8_TR>(a, heads, batches, stream);                           \
    default: return fold ? launch_p5_tr<T, p5tr::S_D256_##TN##_FOLD_TR>(a, heads, batches, stream)                             \
                         : launch_p5_tr<T, p5tr::S_D256_##TN##_THR8_TR>(a, heads, batches, stream);                            \
  }
  if (p == PREC_BF16) { MFA_P5TR_PICK(__bf16, BF16) }
  MFA_P5TR_PICK(_Float16, F16)
#undef MFA_P5TR_PICK
}

} // namespace mfa
