// attn_fwd16_p5_tr.hip -- instantiations of the hand-placed forward kernel for transposed K and / or V at the head-dimension
// buckets 160 / 192 / 256 (attn_fwd16_p5_tr.h) and the launcher that prefers it over the 8 x 32 kernel's transposed code object
// when the launch is whole 32-key steps of aligned rows -- the same arrangement as attn_fwd16_p4_tr.hip at D <= 128.
// GPU evidence (round 4, all patterns): tests/test_attention_gpu.py::test_forward_stream_reads_transposed_keys_and_values_at_large_head_dimensions,
// profiles/r04_candidate/time_p5_tr_32heads.txt.
#include <cstdlib>
#include <cstring>
#include "attn_fwd16_p5_tr.h"
#include "launchers.h"

namespace mfa {

typedef void (*LaunchFn)(dim3 grid, hipStream_t stream, const KernelArgs &args);
// the launcher of the code object `out` arrived with (fwd16_v3_tr_variant_dNN of the same pattern): one per (type, stream)
template <typename T, int STREAM> struct P5TrFallback { static LaunchFn launch; };
template <typename T, int STREAM> LaunchFn P5TrFallback<T, STREAM>::launch = nullptr;

// what the step walk of attn_fwd16_p5_tr needs (its header): K / V transposed as the stream's pattern says (bit 0 = K, bit 1 = V),
// no per-batch lengths, no block mask, whole 32-key steps, 16-byte aligned rows of K and V in either orientation (and of Q when
// it is row-major: its fragments are 8- or 16-byte loads; Q^T is gathered)
static bool p5_tr_takes(const KernelArgs &a, int pattern) {
#ifdef MFA_DEV_VARIANTS
  const char *knob = std::getenv("MFA_FWD16_P5_TR");   // developer library: MFA_FWD16_P5_TR=0 keeps the 8 x 32 code object (A/B runs)
  if (knob && std::strcmp(knob, "0") == 0) return false;
#endif
  auto aligned = [](const OperandView &v) {
    return ((reinterpret_cast<uintptr_t>(v.ptr) | (uint64_t)v.ld * 2 | (uint64_t)v.headStride * 2 | (uint64_t)v.batchStride * 2) & 15) == 0;
  };
  if (a.rowLen || a.colLen || a.mask || a.D <= 128 || a.D > 256 || a.D % 8 || a.C % 32 || a.C == 0) return false;
  if (a.causal && a.C < a.R) return false;
  if ((a.op[SLOT_K].transposed != 0) != ((pattern & 1) != 0) || (a.op[SLOT_V].transposed != 0) != ((pattern & 2) != 0)) return false;
  if (!aligned(a.op[SLOT_K]) || !aligned(a.op[SLOT_V])) return false;   // (16-byte chunks of either orientation)
  return a.op[SLOT_Q].transposed || aligned(a.op[SLOT_Q]);
}

template <typename T, int STREAM>
static void launch_p5_tr(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if (!p5_tr_takes(args, p5tr::stream_pattern(STREAM))) { P5TrFallback<T, STREAM>::launch(grid, stream, args); return; }
  // grid arrives in the 8 x 32 kernel's row blocks (VariantInfo.parallelization); this kernel's are 256 rows
  const uint32_t blocks = ((uint32_t)args.R + 255) / 256;
  Fwd16Grid g{blocks, grid.y, grid.z};
  if (args.causal) {
    const uint32_t groups = (blocks + 1) / 2;   // one workgroup per pair of row blocks (last - i, i)
    hipLaunchKernelGGL((attn_fwd16_p5_tr<T, STREAM, true>), dim3(groups * grid.y * grid.z), dim3(256), p5::LDS_BYTES, stream, args, g);
  } else {
    hipLaunchKernelGGL((attn_fwd16_p5_tr<T, STREAM, false>), dim3(blocks * grid.y * grid.z), dim3(256), p5::LDS_BYTES, stream, args, g);
  }
}

template <typename T, int STREAM> static const char *p5_tr_form(const KernelArgs &args) {
  constexpr int PATTERN = p5tr::stream_pattern(STREAM);
  if (!p5_tr_takes(args, PATTERN)) return nullptr;
  if (PATTERN == 1)
    return p5tr::stream_folds(STREAM) ? "attn_fwd16_p5_tr (four waves x 64 rows, 32-key steps, hand-placed stream on transposed K; scale folded into Q)"
                                       : "attn_fwd16_p5_tr (four waves x 64 rows, 32-key steps, hand-placed stream on transposed K)";
  if (PATTERN == 2)
    return p5tr::stream_folds(STREAM) ? "attn_fwd16_p5_tr (four waves x 64 rows, 32-key steps, hand-placed stream on transposed V; scale folded into Q)"
                                       : "attn_fwd16_p5_tr (four waves x 64 rows, 32-key steps, hand-placed stream on transposed V)";
  return p5tr::stream_folds(STREAM) ? "attn_fwd16_p5_tr (four waves x 64 rows, 32-key steps, hand-placed stream on transposed K / V; scale folded into Q)"
                                     : "attn_fwd16_p5_tr (four waves x 64 rows, 32-key steps, hand-placed stream on transposed K / V)";
}

template <typename T, int STREAM> static void attach(VariantInfo *v) {
  P5TrFallback<T, STREAM>::launch = v->launch;
  v->launch = &launch_p5_tr<T, STREAM>;
  v->launchForm = &p5_tr_form<T, STREAM>;
  // (fields that only name code objects whose LDS limit must be raised before the first launch)
  v->funcCausal = reinterpret_cast<const void *>(&attn_fwd16_p5_tr<T, STREAM, true>);
  v->funcSplit = reinterpret_cast<const void *>(&attn_fwd16_p5_tr<T, STREAM, false>);
  v->ldsBytes = v->ldsBytes > (uint32_t)p5::LDS_BYTES ? v->ldsBytes : (uint32_t)p5::LDS_BYTES;
}

// `out` arrives filled by fwd16_v3_tr_variant_d160 / _d192 / _d256 for `pattern` (bit 0 = K, bit 1 = V transposed; 0 = only Q / O:
// nothing to do): launches the stream can take go to it, the others stay.  fold: Q pre-multiplied by the softmax scale in the 16-bit
// type (mixed-precision descriptors)
bool fwd16_p5_tr_variant(int precision, int bucket, int pattern, bool fold, VariantInfo *out) {
  if (!out->launch || out->launchCausal || out->launchSplit) return false;   // (the transposed code objects take the causal flag at run time and are never split)
#define MFA_P5TR_ATTACH1(T, TN, B, SFX) { if (fold) attach<T, p5tr::S_D##B##_##TN##_FOLD_##SFX>(out); else attach<T, p5tr::S_D##B##_##TN##_THR8_##SFX>(out); return true; }
#define MFA_P5TR_ATTACH(T, TN, SFX)                \
  switch (bucket) {                                \
    case 160: MFA_P5TR_ATTACH1(T, TN, 160, SFX)    \
    case 192: MFA_P5TR_ATTACH1(T, TN, 192, SFX)    \
    case 256: MFA_P5TR_ATTACH1(T, TN, 256, SFX)    \
    default: return false;                         \
  }
#ifdef MFA_P5_TR1_STREAMS   // (the one-operand streams are generated at build time, attn_fwd16_p5_tr.h)
#define MFA_P5TR_ONE_OPERAND(T, TN) case 1: MFA_P5TR_ATTACH(T, TN, TRK) case 2: MFA_P5TR_ATTACH(T, TN, TRV)
#else
#define MFA_P5TR_ONE_OPERAND(T, TN)
#endif
#define MFA_P5TR_PATTERNS(T, TN)                   \
  switch (pattern) {                               \
    MFA_P5TR_ONE_OPERAND(T, TN)                    \
    case 3: MFA_P5TR_ATTACH(T, TN, TR)             \
    default: return false;                         \
  }
  if (precision == PREC_BF16) { MFA_P5TR_PATTERNS(__bf16, BF16) }
  if (precision == PREC_FP16) { MFA_P5TR_PATTERNS(_Float16, F16) }
#undef MFA_P5TR_PATTERNS
#undef MFA_P5TR_ONE_OPERAND
#undef MFA_P5TR_ATTACH
#undef MFA_P5TR_ATTACH1
  return false;
}

} // namespace mfa
