// launchers.h -- host-side entry points of each kernel translation unit (internal, C++).
#pragma once
#include "attn_common.h"

namespace mfa {

struct VariantInfo {
  const void *func = nullptr;   // __global__ function address (for hipFuncSetAttribute)
  const char *name = "";
  // the variant whose launchSplit / launchSplitCausal / launchSparse this one inherited (a hand-placed stream laid over the 8 x 32 /
  // role-split kernel of the same block dimensions): named in the launch form of such launches; nullptr = they are its own
  const char *siblingName = nullptr;
  uint16_t parallelization = 0; // rows (fwd, dQ) or columns (dK/dV) per workgroup
  uint16_t siblingParallelization = 0;   // the same for launchSparse / launchSplit when they belong to another kernel (0: equal)
  uint16_t splitTarget = 0;     // workgroups a traversal-parallel launch aims at (0: 512 = two per compute unit)
  uint16_t splitParallelization = 0;   // rows / columns per workgroup of launchSplit when it is the variant's own kernel again and
                                       // only launchSplitCausal / launchSparse belong to the sibling (0: as the sibling)
  uint16_t traversal = 0;       // columns (fwd, dQ) or rows (dK/dV) per main-loop step
  uint16_t headBlock = 0;       // padded head dimension the code object is unrolled for
  uint32_t threads = 0;         // work-items per workgroup
  uint32_t ldsBytes = 0;        // dynamic LDS of the variant's code object (mfa_attention_kernel_threadgroup_memory_allocation)
  uint32_t attrLdsBytes = 0;    // what `func` (+ siblings) is raised to when that differs (0: ldsBytes): the FP32 production variants keep
                                // the general kernel in `func` for the launches it still serves (attn_f32.hip raises its own kernels)
  bool cacheLeft = false;       // left-hand operands cached in VGPRs (Q / Q,dO / K,V)
  bool cacheSecond = false;     // the second of them alone (dO / V); fill code sets it = cacheLeft unless a variant splits the pair
  bool pagedAccumulators = false;   // accumulators paged through the output buffers (any-D kernels, attn_paged.h); else in registers
  bool causal = false;          // the code object implements the causal mask itself (general kernels: always)
  bool transposedInPlace = false;   // reads / writes transposed operands where they lie, whatever their alignment (attn_fwd16_v3.h, TR)
  void (*launch)(dim3 grid, hipStream_t stream, const KernelArgs &args) = nullptr;
  // dense / causal launches that `launch` / `launchCausal` hand to another code object (the persistent form of the D <= 128
  // forward kernel): its name for such a launch, nullptr when the variant's own kernel runs (mfa_attention_kernel_launch_form)
  const char *(*launchForm)(const KernelArgs &args) = nullptr;
  // the same for a column-parallel launch whose pieces `launchSplit` hands to another code object than the sibling's (the persistent
  // D <= 64 forward kernel cuts launches of whole four-tile pieces itself): text for the launch form, nullptr = the sibling's pieces
  const char *(*splitForm)(const KernelArgs &args, uint32_t splits) = nullptr;
  // forward only: column-parallel launch (key range cut into `splits` pieces, partial results in the
  // caller's workspace, then the combine kernel); nullptr if the variant has none
  void (*launchSplit)(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream,
                      const KernelArgs &args) = nullptr;
  // causal traversal-parallel launches, when they belong to another kernel than launchSplit (nullptr: launchSplit takes both)
  void (*launchSplitCausal)(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) = nullptr;
  // separate code object implementing the causal mask (the unmasked loop bodies stay branch-free);
  // nullptr when `launch` handles the flag itself (general kernels) or the variant has no mask
  void (*launchCausal)(dim3 grid, hipStream_t stream, const KernelArgs &args) = nullptr;
  const void *funcCausal = nullptr;
  // code objects that honour KernelArgs.mask (block-sparse extension; the launcher picks the causal or the
  // unmasked one from args.causal); nullptr = the variant has none and the general kernels serve the launch
  void (*launchSparse)(dim3 grid, hipStream_t stream, const KernelArgs &args) = nullptr;
  const void *funcSparse = nullptr, *funcSparseCausal = nullptr;
  bool sparse = false;   // `launch` itself honours the mask (general kernels)
  // code objects of launchSplit that need the large-LDS attribute (backward kernels; the forward split variant
  // is registered by its launcher's first use of the same attribute path)
  const void *funcSplit = nullptr, *funcSplitCausal = nullptr;
};

// any head dimension (D > 384): D-blocked products, accumulators paged through the FP32 output buffers (attn_paged.h); type = kernel type
bool paged_variant(int type, VariantInfo *out);

// generic (fp32-MFMA) family: returns false if (DP) is not compiled
// FP32 production kernels (attn_f32.h): the general kernels' launchers of the 64 / 128 head blocks hand over the launches whose
// operands qualify (all FP32, row-major, 16-byte aligned rows, D % 4 == 0, no block mask).  type: 0 forward, 1 backwardQuery,
// 2 backwardKeyValue; grid as the general kernel's (blocks of 128, heads, batches).  false / nullptr: not one of theirs
bool f32_launch(int type, int DP, dim3 grid, hipStream_t stream, const KernelArgs &args);
const char *f32_form(int type, int DP, const KernelArgs &args);
// FP32 descriptors with row-major operands and D % 4 == 0 at the 64 / 128 head blocks: the variant IS the FP32 production kernel
// (own name, own LDS bytes); `out` arrives filled by generic_*_variant(DP), whose kernel becomes the sibling that keeps block-sparse
// launches and launches whose operands miss the 16-byte row alignment
bool f32_variant(int type, int DP, VariantInfo *out);
bool generic_fwd_variant(int DP, VariantInfo *out);
bool generic_dq_variant(int DP, VariantInfo *out);
bool generic_dkv_variant(int DP, VariantInfo *out);

// 16-bit MFMA forward family (Q, K, V in one 16-bit type, row-major, D % 8 == 0)
bool fwd16_variant(int precision, int D, VariantInfo *out);
// software-pipelined version; impl selects an experimental schedule (see attn_fwd16_v2.hip)
bool fwd16_v2_variant(int precision, int D, int impl, VariantInfo *out);
// one wave per SIMD, 64 query rows per wave, half-tile pipeline (see attn_fwd16_v3.h)
bool fwd16_v3_variant(int precision, int D, int impl, VariantInfo *out);
// four waves x 64 rows, one wave per SIMD, hand-placed instruction stream (attn_fwd16_p4.h); D <= 128 only
bool fwd16_p4_variant(int precision, int D, int impl, VariantInfo *out);
// D <= 64: four waves x 64 rows, persistent (attn_fwd16_p6.h; fold = the descriptor holds the attention matrix in 16-bit registers:
// scale folded into Q, row sums in the matrix pipe); `out` arrives filled by fwd16_v3_variant(precision, 64, 0), whose kernel keeps
// the launches this one does not serve
bool fwd16_p6_variant(int precision, bool fold, VariantInfo *out);
bool launch_p6(int precision, bool fold, dim3 grid, hipStream_t stream, const KernelArgs &args);
const char *p6_form(int precision, bool fold, const KernelArgs &args);
bool launch_p6_split(int precision, bool fold, dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args);
// 256 < D <= 384 (head blocks 320, 384): four waves x 32 rows, 32-key steps, compiler-scheduled (attn_fwd16_wide.h, round 6)
bool fwd16_wide_variant(int precision, int D, VariantInfo *out);
// the backward kernels of the same head blocks (attn_bwd16_wide.hip: attn_dq16 with 32-key tiles, attn_dkv16_wide.h; round 6)
bool dq16_wide_variant(int precision, int gprecision, int D, VariantInfo *out);
bool dkv16_wide_variant(int precision, int gprecision, int D, VariantInfo *out);
// 128 < D <= 256: four waves x 64 rows, 32-key steps (attn_fwd16_p5.h); `out` arrives filled by fwd16_v3_variant
bool fwd16_p5_variant(int precision, int D, int impl, VariantInfo *out);
// backwardKeyValue counterpart: four waves x 64 keys (attn_dkv16_p4.h); `out` arrives filled by dkv16_rs_variant, whose
// split / block-sparse launchers it keeps.  lprec / dprec: storage types of L and D (fixed per instruction stream)
bool dkv16_p4_variant(int precision, int gprecision, int lprec, int dprec, int D, int impl, VariantInfo *out);
// buckets 160 / 192 / 256: role-split wave pairs x 64 keys, hand-placed stream (attn_dkv16_p5.h); `out` arrives filled by the 32-key
// role-split kernel of the bucket (attn_dkv16_rs.h), which keeps the block-sparse and row-parallel launches
bool dkv16_p5_variant(int precision, int gprecision, int lprec, int dprec, int D, VariantInfo *out);
// buckets 160 / 192 / 256: role-split wave pairs x 64 rows, hand-placed stream (attn_dq16_p5.h); `out` arrives filled by the 32-row-wave
// kernel of the bucket (attn_bwd16.h attn_dq16), which keeps the block-sparse and column-parallel launches
bool dq16_p5_variant(int precision, int gprecision, int D, int impl, VariantInfo *out);
// backwardQuery counterpart: four waves x 64 rows (attn_dq16_p4.h); `out` arrives filled by dq16_variant
bool dq16_p4_variant(int precision, int gprecision, int D, int impl, VariantInfo *out);
// 8 waves x 32 rows, SIMD partners alternate matrix / vector segments (see attn_fwd16_v4.h)
bool fwd16_v4_variant(int precision, int D, int impl, VariantInfo *out);

// 16-bit MFMA backward kernels (Q, K, V, dO in one 16-bit type, row-major, D in {64, 128, 256})
// (gprecision = storage type of dO: the same 16-bit type, or BF16 next to FP16 Q/K/V)
bool dq16_variant(int precision, int gprecision, int D, VariantInfo *out);
bool dkv16_variant(int precision, int gprecision, int D, VariantInfo *out);
// role-split wave pairs: one wave of a SIMD accumulates dV, its partner dK (see attn_dkv16_rs.h)
bool dkv16_rs_variant(int precision, int gprecision, int D, int impl, VariantInfo *out);

// head-dimension buckets 160 / 192 of the 16-bit trio and 96 of the dK/dV kernel (one translation unit per kernel type and
// bucket).  Measured at N = 4096, 64 heads (profiles/r02_bucket_perf.txt): the 96-wide forward and dQ objects LOSE to the 128
// objects run on zero-padded chunks (0.576 vs 0.499 ms, 0.885 vs 0.781 ms) and are not built; dK/dV wins at 96 (1.02 vs 1.18 ms)
bool fwd16_v3_variant_d160(int precision, VariantInfo *out);
bool fwd16_v3_variant_d192(int precision, VariantInfo *out);
// operands stored transposed, read in place (TR kernels of attn_fwd16_v3.h): pattern bit 0 = K, bit 1 = V transposed (Q / O: any)
bool fwd16_v3_tr_variant_d64(int precision, int D, int pattern, VariantInfo *out);   // buckets 32, 64
bool fwd16_v3_tr_variant_d128(int precision, int D, int pattern, VariantInfo *out);
bool fwd16_v3_tr_variant_d160(int precision, int D, int pattern, VariantInfo *out);
bool fwd16_v3_tr_variant_d192(int precision, int D, int pattern, VariantInfo *out);
bool fwd16_v3_tr_variant_d256(int precision, int D, int pattern, VariantInfo *out);
// K and / or V transposed at D <= 128 (pattern: bit 0 = K, bit 1 = V): launches of whole chunks of aligned rows run the hand-placed
// stream (attn_fwd16_p4_tr.h); `out` arrives filled by fwd16_v3_tr_variant_d128, whose kernel keeps the others
bool fwd16_p4_tr_variant(int precision, int pattern, bool fold, VariantInfo *out);
bool dq16_variant_d160(int precision, int gprecision, VariantInfo *out);
bool dq16_variant_d192(int precision, int gprecision, VariantInfo *out);
bool dkv16_rs_variant_d96(int precision, int gprecision, VariantInfo *out);
bool dkv16_rs_variant_d160(int precision, int gprecision, VariantInfo *out);
bool dkv16_rs_variant_d192(int precision, int gprecision, VariantInfo *out);

// backward kernels that read transposed operands in place (attn_bwd16_p4_tr.hip); false = not such a launch
bool bwd16_p4_tr_launch(int type, const KernelArgs &args, uint32_t heads, uint32_t batches, hipStream_t stream, bool fold);
const char *bwd16_p4_tr_form(int type, const KernelArgs &args);
// launches with K^T and / or V^T at the buckets 160 / 192 / 256 that are whole 32-key steps of aligned rows go
// to the hand-placed stream (attn_fwd16_p5_tr.h); `out` arrives filled by fwd16_v3_tr_variant_dNN, whose kernel keeps the others
bool fwd16_p5_tr_variant(int precision, int bucket, int pattern, bool fold, VariantInfo *out);

} // namespace mfa
