// attn_fwd16_p4_tr.hip -- instantiations of the hand-placed forward kernel for transposed K and V (attn_fwd16_p4_tr.h) and the
// launcher that prefers it over the 8 x 32 kernel's transposed code object when the launch is whole tiles of aligned rows.
#include "attn_fwd16_p4_tr.h"
#include "attn_fwd16_v3_launch.h"

namespace mfa {

// what the tile walk of attn_fwd16_p4_tr needs (its header): whole 16-byte chunks of keys, no per-batch lengths, no block mask,
// 16-byte aligned rows of K^T, V^T and (if transposed) Q^T, row-major Q as the row-major kernel wants it
static bool p4_tr_takes(const KernelArgs &a) {
  if (a.C % 8 != 0 || a.rowLen || a.colLen || a.mask) return false;
  auto aligned = [](const OperandView &v) {
    return ((reinterpret_cast<uintptr_t>(v.ptr) | (uint64_t)v.ld * 2 | (uint64_t)v.headStride * 2 | (uint64_t)v.batchStride * 2) & 15) == 0;
  };
  if (!a.op[SLOT_K].transposed || !a.op[SLOT_V].transposed) return false;
  if (!aligned(a.op[SLOT_K]) || !aligned(a.op[SLOT_V]) || !aligned(a.op[SLOT_Q])) return false;
  return true;
}

template <typename T, int STREAM>
static void launch_p4_tr(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  // grid arrives in the 8 x 32 kernel's 256-row workgroups: the same row blocks
  if (!p4_tr_takes(args)) { launch_v3_tr<T, 128, 8, 3, 0, 7>(grid, stream, args); return; }
  Fwd16Grid g{grid.x, grid.y, grid.z};
  if (args.causal) {
    const uint32_t groups = (grid.x + 1) / 2;   // one workgroup per pair of row blocks (last - i, i)
    hipLaunchKernelGGL((attn_fwd16_p4_tr<T, STREAM, true>), dim3(groups * grid.y * grid.z), dim3(256), p4::LDS_BYTES, stream, args, g);
  } else {
    hipLaunchKernelGGL((attn_fwd16_p4_tr<T, STREAM, false>), dim3(grid.x * grid.y * grid.z), dim3(256), p4::LDS_BYTES, stream, args, g);
  }
}

template <typename T, int STREAM> static const char *p4_tr_form(const KernelArgs &args) {
  if (!p4_tr_takes(args)) return nullptr;
  return p4tr::stream_folds(STREAM) ? "attn_fwd16_p4_tr (four waves x 64 rows, hand-placed stream on transposed K / V; scale folded into Q)"
                                     : "attn_fwd16_p4_tr (four waves x 64 rows, hand-placed stream on transposed K / V)";
}

template <typename T, int STREAM> static void attach(VariantInfo *v) {
  v->launch = &launch_p4_tr<T, STREAM>;
  v->launchForm = &p4_tr_form<T, STREAM>;
  // (fields that only name code objects whose LDS limit must be raised before the first launch)
  v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_p4_tr<T, STREAM, true>);
  v->funcSplit = reinterpret_cast<const void *>(&attn_fwd16_p4_tr<T, STREAM, false>);
  v->ldsBytes = v->ldsBytes > (uint32_t)p4::LDS_BYTES ? v->ldsBytes : (uint32_t)p4::LDS_BYTES;
}

// `out` arrives filled by fwd16_v3_tr_variant_d128 for the pattern (K, V) = (transposed, transposed): launches the stream can
// take go to it, the others stay.  fold: Q pre-multiplied by the softmax scale in the 16-bit type (mixed-precision descriptors)
bool fwd16_p4_tr_variant(int precision, bool fold, VariantInfo *out) {
  if (precision == PREC_BF16) {
    if (fold) attach<__bf16, p4tr::S_BF16_FOLD_TR>(out); else attach<__bf16, p4tr::S_BF16_THR8_TR>(out);
    return true;
  }
  if (precision == PREC_FP16) {
    if (fold) attach<_Float16, p4tr::S_F16_FOLD_TR>(out); else attach<_Float16, p4tr::S_F16_THR8_TR>(out);
    return true;
  }
  return false;
}

} // namespace mfa
