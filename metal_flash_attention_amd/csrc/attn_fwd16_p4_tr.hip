// attn_fwd16_p4_tr.hip -- instantiations of the hand-placed forward kernel for transposed K and / or V (attn_fwd16_p4_tr.h) and the
// launcher that prefers it over the 8 x 32 kernel's transposed code object when the launch is whole chunks of aligned rows.
#include "attn_fwd16_p4_tr.h"
#include "attn_fwd16_v3_launch.h"

namespace mfa {

// what the tile walk of attn_fwd16_p4_tr needs (its header): no per-batch lengths, no block mask, 16-byte aligned rows of the
// transposed operands (K^T, V^T and, if transposed, Q^T), whole 16-byte chunks of keys in V^T (the last tile's chunks at or beyond
// key C are left out, a chunk cannot be cut); row-major operands as the row-major kernel wants them (checked before)
static bool p4_tr_takes(const KernelArgs &a, int pattern) {
  if (a.rowLen || a.colLen || a.mask) return false;
  auto aligned = [](const OperandView &v) {
    return ((reinterpret_cast<uintptr_t>(v.ptr) | (uint64_t)v.ld * 2 | (uint64_t)v.headStride * 2 | (uint64_t)v.batchStride * 2) & 15) == 0;
  };
  if ((a.op[SLOT_K].transposed != 0) != ((pattern & 1) != 0) || (a.op[SLOT_V].transposed != 0) != ((pattern & 2) != 0)) return false;
  if ((pattern & 2) && a.C % 8 != 0) return false;
  return aligned(a.op[SLOT_K]) && aligned(a.op[SLOT_V]) && aligned(a.op[SLOT_Q]);
}

template <typename T, int STREAM>
static void launch_p4_tr(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  constexpr int PATTERN = p4tr::stream_pattern(STREAM);
  // grid arrives in the 8 x 32 kernel's 256-row workgroups: the same row blocks
  if (!p4_tr_takes(args, PATTERN)) { launch_v3_tr<T, 128, 8, 3, 0, 4 | PATTERN>(grid, stream, args); return; }
  Fwd16Grid g{grid.x, grid.y, grid.z};
  if (args.causal) {
    const uint32_t groups = (grid.x + 1) / 2;   // one workgroup per pair of row blocks (last - i, i)
    hipLaunchKernelGGL((attn_fwd16_p4_tr<T, STREAM, true>), dim3(groups * grid.y * grid.z), dim3(256), p4::LDS_BYTES, stream, args, g);
  } else {
    hipLaunchKernelGGL((attn_fwd16_p4_tr<T, STREAM, false>), dim3(grid.x * grid.y * grid.z), dim3(256), p4::LDS_BYTES, stream, args, g);
  }
}

template <typename T, int STREAM> static const char *p4_tr_form(const KernelArgs &args) {
  constexpr int PATTERN = p4tr::stream_pattern(STREAM);
  if (!p4_tr_takes(args, PATTERN)) return nullptr;
  if (PATTERN == 3)
    return p4tr::stream_folds(STREAM) ? "attn_fwd16_p4_tr (four waves x 64 rows, hand-placed stream on transposed K / V; scale folded into Q)"
                                       : "attn_fwd16_p4_tr (four waves x 64 rows, hand-placed stream on transposed K / V)";
  if (PATTERN == 1)
    return p4tr::stream_folds(STREAM) ? "attn_fwd16_p4_tr (four waves x 64 rows, hand-placed stream on transposed K; scale folded into Q)"
                                       : "attn_fwd16_p4_tr (four waves x 64 rows, hand-placed stream on transposed K)";
  return p4tr::stream_folds(STREAM) ? "attn_fwd16_p4_tr (four waves x 64 rows, hand-placed stream on transposed V; scale folded into Q)"
                                     : "attn_fwd16_p4_tr (four waves x 64 rows, hand-placed stream on transposed V)";
}

template <typename T, int STREAM> static void attach(VariantInfo *v) {
  v->launch = &launch_p4_tr<T, STREAM>;
  v->launchForm = &p4_tr_form<T, STREAM>;
  // (fields that only name code objects whose LDS limit must be raised before the first launch)
  v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_p4_tr<T, STREAM, true>);
  v->funcSplit = reinterpret_cast<const void *>(&attn_fwd16_p4_tr<T, STREAM, false>);
  v->ldsBytes = v->ldsBytes > (uint32_t)p4::LDS_BYTES ? v->ldsBytes : (uint32_t)p4::LDS_BYTES;
}

// `out` arrives filled by fwd16_v3_tr_variant_d128 for `pattern` (bit 0 = K, bit 1 = V transposed; 0 = only Q / O: nothing to do):
// launches the stream can take go to it, the others stay.  fold: Q pre-multiplied by the softmax scale in the 16-bit type
// (mixed-precision descriptors)
bool fwd16_p4_tr_variant(int precision, int pattern, bool fold, VariantInfo *out) {
#define MFA_P4TR_ATTACH(T, TN)                                                                        \
  switch (pattern) {                                                                                   \
    case 1: if (fold) attach<T, p4tr::S_##TN##_FOLD_TRK>(out); else attach<T, p4tr::S_##TN##_THR8_TRK>(out); return true; \
    case 2: if (fold) attach<T, p4tr::S_##TN##_FOLD_TRV>(out); else attach<T, p4tr::S_##TN##_THR8_TRV>(out); return true; \
    case 3: if (fold) attach<T, p4tr::S_##TN##_FOLD_TR>(out); else attach<T, p4tr::S_##TN##_THR8_TR>(out); return true;   \
    default: return false;                                                                             \
  }
  if (precision == PREC_BF16) { MFA_P4TR_ATTACH(__bf16, BF16) }
  if (precision == PREC_FP16) { MFA_P4TR_ATTACH(_Float16, F16) }
#undef MFA_P4TR_ATTACH
  return false;
}

} // namespace mfa
