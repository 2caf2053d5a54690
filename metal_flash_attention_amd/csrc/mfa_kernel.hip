// mfa_kernel.hip -- AttentionKernel object, variant selection and launch (C ABI of include/mfa.h).
//
// Replaces, for gfx950, the reference's AttentionKernel(descriptor:) + createSource() and the
// Metal calls its callers make (Sources/FlashAttention/Attention/AttentionKernel/
// AttentionKernel.swift:27-50, :268-363; AttentionKernel+Source.swift:11-55;
// Tests/FlashAttentionTests/Attention/SquareAttentionTest.swift:244-260, :319-368): instead of
// emitting shader source for a JIT, the descriptor selects one of the pre-compiled code objects.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "launchers.h"
#include "mfa_internal.h"

using namespace mfa;

struct mfa_attention_kernel {
  mfa_attention_kernel_descriptor desc;  // as requested
  mfa_attention_kernel_descriptor effective; // what the selected code object really does
  VariantInfo variant;            // preferred code object
  VariantInfo fallback;           // general code object, used when a launch does not meet the
  bool hasFallback = false;       // preferred variant's alignment requirements
  bool relayout = false;          // transposed operands: the preferred variant runs on row-major copies in the caller's workspace
  std::mutex attrMutex;
  uint64_t attrDeviceMask = 0;    // devices on which the LDS attribute has been raised (variant)
  uint64_t attrDeviceMaskFallback = 0;
};

static mfa_status hip_fail(hipError_t err, const char *what) {
  return fail(MFA_ERR_HIP, std::string(what) + ": " + hipGetErrorName(err) + " (" + hipGetErrorString(err) + ")");
}

static int slot_operand(int slot) {
  static const int ops[MFA_BUFFER_SLOTS] = {MFA_Q, MFA_K, MFA_V, MFA_O, MFA_L, MFA_D, MFA_dO, MFA_dV, MFA_dK, MFA_dQ};
  return ops[slot];
}

// operands each kernel type touches (+Source.swift:72-103)
static bool slot_used(int type, int slot) {
  switch (type) {
    case MFA_FORWARD: return slot <= 4;
    case MFA_BACKWARD_QUERY: return slot <= 6 || slot == 9;
    default: return slot <= 2 || (slot >= 4 && slot <= 8);
  }
}

// head-dimension buckets of the 16-bit matrix-core code objects (the forward has a D = 32 object, the backward pair starts
// at 64 and runs smaller heads zero-padded); 0 = none (D > 256: fp32-arithmetic kernels)
enum { B16_FORWARD, B16_DQ, B16_DKV };
static int bucket16(int D, int kind) {
  static const int buckets[] = {32, 64, 96, 128, 160, 192, 256};
  for (int b : buckets) {
    if (b == 32 && kind != B16_FORWARD) continue;
    if (b == 96 && kind != B16_DKV) continue;   // forward and dQ: the 128 objects are faster on D <= 96 than 96-wide ones
    if (D <= b) return b;
  }
  return 0;
}

static int generic_bucket(int D) {
  static const int buckets[] = {32, 64, 128, 256, 384};
  for (int b : buckets)
    if (D <= b) return b;
  return -1;
}

extern "C" {

mfa_status mfa_attention_kernel_create(const mfa_attention_kernel_descriptor *kdesc, mfa_attention_kernel **out) {
  if (!kdesc || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  // AttentionKernel.init guard (AttentionKernel.swift:28-34)
  if (!kdesc->hasBlockDimensions || !kdesc->hasHeadDimension || kdesc->preferAsyncCache < 0 ||
      kdesc->preferAsyncLoad < 0 || kdesc->type < 0)
    return fail(MFA_ERR_INCOMPLETE_DESCRIPTOR, "Descriptor was incomplete.");
  const int type = kdesc->type;
  if (type > 2) return fail(MFA_ERR_INVALID_ARGUMENT, "unknown kernel type");
  if (kdesc->headDimension == 0) return fail(MFA_ERR_INVALID_ARGUMENT, "headDimension must be non-zero");
  for (int slot = 0; slot < MFA_BUFFER_SLOTS; ++slot) {
    if (!slot_used(type, slot)) continue;
    const int op = slot_operand(slot);
    const int prec = kdesc->memoryPrecisions[op];
    if (prec < 0)  // AttentionKernel.swift:56-58 "Memory precision of X was not specified."
      return fail(MFA_ERR_INCOMPLETE_DESCRIPTOR, std::string("Memory precision of ") + mfa_operand_name(op) + " was not specified.");
    if (prec > MFA_BF16) return fail(MFA_ERR_INVALID_ARGUMENT, "unknown precision");
    if (op != MFA_L && op != MFA_D && kdesc->transposeState[op] < 0)
      return fail(MFA_ERR_INCOMPLETE_DESCRIPTOR, std::string("Transpose state of ") + mfa_operand_name(op) + " was not specified.");
  }

  // ---- candidates: every compiled code object that can serve this descriptor.  The general (fp32-arithmetic) kernel
  // of the head-dimension bucket always can; the matrix-core kernels need Q, K, V (and dO) in ONE 16-bit type, nothing
  // transposed, outputs in FP32 or the inputs' type, and a head dimension that is a multiple of 8 (16-byte chunks).
  VariantInfo general;
  std::vector<VariantInfo> candidates;   // matrix-core candidates, the product default first
  bool found = false;
  const int D = kdesc->headDimension;
  const int bucket = generic_bucket(D);
  if (bucket > 0) {
    switch (type) {
      case MFA_FORWARD: found = generic_fwd_variant(bucket, &general); break;
      case MFA_BACKWARD_QUERY: found = generic_dq_variant(bucket, &general); break;
      default: found = generic_dkv_variant(bucket, &general); break;
    }
  }
  if (!found) {
    // D > 384: the reference falls through to its tables' last row and pages the accumulators through the output buffers
    // (+Parameters.swift:60-65, +Accumulate.swift:403-469); so do the any-D kernels -- which therefore need those buffers in FP32,
    // as the reference always has them (+Precisions.swift:140-143)
    static const int outs[3][2] = {{MFA_O, MFA_O}, {MFA_dQ, MFA_dQ}, {MFA_dK, MFA_dV}};
    for (int i = 0; i < 2; ++i)
      if (kdesc->memoryPrecisions[outs[type][i]] != MFA_FP32)
        return fail(MFA_ERR_UNSUPPORTED, "head dimension " + std::to_string(D) + " > 384 pages the accumulators through the output buffer: " +
                                             mfa_operand_name(outs[type][i]) + " must be FP32 (lowPrecisionOutputs is not available there)");
    found = paged_variant(type, &general);
  }
  if (!found) return fail(MFA_ERR_UNSUPPORTED, "no gfx950 code object for head dimension " + std::to_string(D));
  {
    // FP32 descriptors (every operand FP32, nothing transposed, D % 4 == 0) at the 64 / 128 head blocks ARE the FP32 production
    // kernels of attn_f32.h: own variant name, own LDS bytes, same block dimensions as the general kernel that stays their sibling
    bool allF32 = (D % 4) == 0;
    for (int slot = 0; slot < MFA_BUFFER_SLOTS && allF32; ++slot) {
      if (!slot_used(type, slot)) continue;
      const int op = slot_operand(slot);
      allF32 = kdesc->memoryPrecisions[op] == MFA_FP32 && (op == MFA_L || op == MFA_D || kdesc->transposeState[op] == 0);
    }
#ifdef MFA_DEV_VARIANTS
    if (std::getenv("MFA_F32_GENERAL")) allF32 = false;
#endif
    if (allF32) f32_variant(type, bucket, &general);
  }
  const int pq = kdesc->memoryPrecisions[MFA_Q];
  const bool same16 = pq != MFA_FP32 && pq == kdesc->memoryPrecisions[MFA_K] && pq == kdesc->memoryPrecisions[MFA_V];
  auto f32_or_inputs = [&](int op) { return kdesc->memoryPrecisions[op] == MFA_FP32 || kdesc->memoryPrecisions[op] == pq; };
  auto add = [&](bool ok, const VariantInfo &v) { if (ok) candidates.push_back(v); };
  VariantInfo v;
  bool relayout = false;
  if (type == MFA_FORWARD) {
    // transposed operands (transposeState, AttentionKernelDescriptor.swift:28-42): the forward kernel has code objects that
    // read and write them in place, like the reference (AttentionKernel.swift:189-204) -- one per pattern of (K, V), Q and O
    // are run-time flags of those (attn_fwd16_v3.h, TR).  No workspace, no re-layout pass.
    const bool transposedOperands = kdesc->transposeState[MFA_Q] || kdesc->transposeState[MFA_K] || kdesc->transposeState[MFA_V] || kdesc->transposeState[MFA_O];
    const int b16 = bucket16(D, B16_FORWARD);
    if (transposedOperands && same16 && f32_or_inputs(MFA_O) && (D % 8) == 0 && b16 > 0) {
      const int pattern = (kdesc->transposeState[MFA_K] ? 1 : 0) | (kdesc->transposeState[MFA_V] ? 2 : 0);
      switch (b16) {
        case 32: case 64: add(fwd16_v3_tr_variant_d64(pq, b16, pattern, &v), v); break;
        case 128: {
          bool have = fwd16_v3_tr_variant_d128(pq, b16, pattern, &v);
          if (have && pattern != 0) fwd16_p4_tr_variant(pq, pattern, kdesc->registerPrecisions[MFA_P] > MFA_FP32, &v);
          add(have, v);
          break;
        }
        default: {
          bool have = b16 == 160 ? fwd16_v3_tr_variant_d160(pq, b16, pattern, &v)
                    : b16 == 192 ? fwd16_v3_tr_variant_d192(pq, b16, pattern, &v) : fwd16_v3_tr_variant_d256(pq, b16, pattern, &v);
          if (have && pattern != 0) fwd16_p5_tr_variant(pq, b16, pattern, kdesc->registerPrecisions[MFA_P] > MFA_FP32, &v);
          add(have, v);
          break;
        }
      }
    } else if (same16 && f32_or_inputs(MFA_O) && (D % 8) == 0 && b16 > 0) {
      VariantInfo v3;
      bool have3 = false;
      switch (b16) {
        case 160: have3 = fwd16_v3_variant_d160(pq, &v3); break;
        case 192: have3 = fwd16_v3_variant_d192(pq, &v3); break;
        default: have3 = fwd16_v3_variant(pq, b16, 0, &v3); break;
      }
      if (have3 && b16 == 128) {
        // four waves x 64 rows, hand-placed stream (attn_fwd16_p4.h); split / block-sparse launches keep the siblings of
        // the 8 x 32 kernel.  A descriptor that holds the attention matrix in 16-bit registers (the reference's
        // lowPrecisionIntermediates: P, and for FP16 also S, +Precisions.swift:149-215) selects the stream that
        // pre-multiplies Q by the softmax scale in the 16-bit type; otherwise the scale is applied in fp32 per score
        const bool lowS = kdesc->registerPrecisions[MFA_P] > MFA_FP32;
        v = v3;
        add(fwd16_p4_variant(pq, 128, lowS ? 10 : 0, &v), v);
      }
      // (D <= 32: the same kernel on zero-padded chunks, selected by a | 32 | 256 | 64 | 64 | row -- the FOLD streams from D = 16 on: with
      // fewer terms per score the rounding of Q' = Q log2(e)/sqrt(D) to BF16 no longer averages out and L leaves the reference's 7e-3)
      if (have3 && (b16 == 64 || (b16 == 32 && (D >= 16 || kdesc->registerPrecisions[MFA_P] <= MFA_FP32)))) {
        // D <= 64 (buckets 32 and 64 of the eight-wave kernel): four waves x 64 rows, persistent, 64-key steps (attn_fwd16_p6.h, round 5);
        // mixed-precision descriptors get the streams with the row sums in the matrix pipe.  | 64 | 256 | 32 | 64 | selects the eight
        // 32-row waves of attn_fwd16_v3.h, which also keep this kernel's causal / block-sparse / column-parallel launches
        v = v3;
        // (D <= 32 on this kernel: the launches it does not serve -- per-batch lengths, an L of the other storage type, pieces that
        // are not whole multiples of four tiles -- go to the D = 64 eight-wave kernels, so the base must be THEIR variant: functions
        // whose dynamic LDS attribute ensure_lds_attribute raises, 256-row blocks for split grids and choose_splits)
        if (b16 == 32 && !fwd16_v3_variant(pq, 64, 0, &v)) v = v3;
        add(fwd16_p6_variant(pq, kdesc->registerPrecisions[MFA_P] > MFA_FP32, &v), v);
      }
      if (have3 && (b16 == 160 || b16 == 192 || b16 == 256)) {   // four waves x 64 rows, 32-key steps (attn_fwd16_p5.h)
        const bool lowS = kdesc->registerPrecisions[MFA_P] > MFA_FP32;
        v = v3;
        add(fwd16_p5_variant(pq, b16, lowS ? 10 : 0, &v), v);
      }
      add(have3, v3);
    } else if (same16 && f32_or_inputs(MFA_O) && (D % 8) == 0 && D > 256 && D <= 384) {
      // 256 < D <= 384: the `| 384 | ... |` rows of the reference's mixed tables (AttentionDescriptor+Parameters.swift:113, :120) on the
      // 16-bit matrix cores (attn_fwd16_wide.h, round 6; until then fp32 arithmetic on 16-bit storage, 1/16 of the rate).  Transposed
      // operands: these head blocks have no in-place code objects -- row-major copies in the caller's workspace, like the backward
      // kernels' re-layout pass (without a workspace: the general kernel in place)
      add(fwd16_wide_variant(pq, D, &v), v);
      relayout = transposedOperands;
    }
  } else {
    const int pg = kdesc->memoryPrecisions[MFA_dO];
    relayout = kdesc->transposeState[MFA_Q] || kdesc->transposeState[MFA_K] || kdesc->transposeState[MFA_V] || kdesc->transposeState[MFA_dO];
    if (type == MFA_BACKWARD_QUERY) relayout = relayout || kdesc->transposeState[MFA_O] || kdesc->transposeState[MFA_dQ];
    else relayout = relayout || kdesc->transposeState[MFA_dK] || kdesc->transposeState[MFA_dV];
    const int b16 = bucket16(D, type == MFA_BACKWARD_QUERY ? B16_DQ : B16_DKV);
    if (same16 && pg != MFA_FP32 && (D % 8) == 0 && b16 > 0) {
      if (type == MFA_BACKWARD_QUERY && f32_or_inputs(MFA_O) && f32_or_inputs(MFA_dQ)) {
        // buckets 160, 192, 256: two wave pairs x 64 rows, hand-placed role-split stream (attn_dq16_p5.h), in front of the four
        // 32-row waves of the same bucket (| D | 128 | 64 | D | selects those)
        auto add_dq5 = [&](bool w4, int b) {
          if (w4) {
            VariantInfo v5 = v;
            add(dq16_p5_variant(pq, pg, b, kdesc->registerPrecisions[MFA_P] > MFA_FP32 ? 10 : 0, &v5), v5);
          }
          add(w4, v);
        };
        switch (b16) {
          case 160: add_dq5(dq16_variant_d160(pq, pg, &v), 160); break;
          case 192: add_dq5(dq16_variant_d192(pq, pg, &v), 192); break;
          case 256: add_dq5(dq16_variant(pq, pg, 256, &v), 256); break;
          default: {
            const bool w8 = dq16_variant(pq, pg, b16, &v);
            if (w8 && (b16 == 128 || b16 == 64)) {   // four waves x 64 rows, hand-placed stream (attn_dq16_p4.h): the same block
              VariantInfo v4 = v;                    // dimensions as the 8 x 32 kernel, which keeps the launches this one lacks
              add(dq16_p4_variant(pq, pg, b16, kdesc->registerPrecisions[MFA_P] > MFA_FP32 ? 10 : 0, &v4), v4);
            }
            add(w8, v);
            break;
          }
        }
      }
      if (type == MFA_BACKWARD_KEY_VALUE && f32_or_inputs(MFA_dK) && f32_or_inputs(MFA_dV) &&
          kdesc->memoryPrecisions[MFA_dK] == kdesc->memoryPrecisions[MFA_dV]) {
        // buckets 160, 192, 256: two wave pairs x 64 keys, hand-placed role-split stream (attn_dkv16_p5.h), in front of the
        // 32-key pairs of the same bucket (| D | 64 | 32 | D | selects those)
        auto add_p5 = [&](bool rs, int b) {
          if (rs) {
            VariantInfo v5 = v;
            add(dkv16_p5_variant(pq, pg, kdesc->memoryPrecisions[MFA_L], kdesc->memoryPrecisions[MFA_D], b, &v5), v5);
          }
          add(rs, v);
        };
        auto add_bucket = [&](int b) {   // buckets 64, 128, 256
          const bool rs = dkv16_rs_variant(pq, pg, b, 0, &v);
          if (b == 256) { add_p5(rs, 256); return; }
          if (rs && (b == 128 || b == 64)) {   // four waves x 64 keys, hand-placed stream (attn_dkv16_p4.h)
            VariantInfo v4 = v;
            add(dkv16_p4_variant(pq, pg, kdesc->memoryPrecisions[MFA_L], kdesc->memoryPrecisions[MFA_D], b, 0, &v4), v4);
          }
          add(rs, v);
        };
        switch (b16) {   // role-split wave pairs (attn_dkv16_rs.h)
          // 64 < D <= 96: the 128 bucket's stream is 1.4 x faster than the 96-wide role-split pairs
          // (profiles/r02_bucket96_dkv.txt) and is what the default table row asks for; a | 96 | 128 | 32 | 96 | row selects these
          case 96: add(dkv16_rs_variant_d96(pq, pg, &v), v); add_bucket(128); break;
          case 160: add_p5(dkv16_rs_variant_d160(pq, pg, &v), 160); break;
          case 192: add_p5(dkv16_rs_variant_d192(pq, pg, &v), 192); break;
          default: add_bucket(b16); break;
        }
        add(dkv16_variant(pq, pg, b16 == 96 ? 128 : b16, &v), v);   // one wave per key block (attn_bwd16.h; D = 64, 128 only)
      }
    } else if (same16 && pg != MFA_FP32 && (D % 8) == 0 && D > 256 && D <= 384) {
      // 256 < D <= 384 (round 6): the backward kernels of the head blocks 320 / 384 on the 16-bit matrix cores (attn_bwd16_wide.hip;
      // until then fp32 arithmetic on 16-bit storage, 1/16 of the rate).  Dense, causal, per-batch lengths; transposed operands through
      // the re-layout pass into the caller's workspace like every other bucket; block masks keep the general kernel (the fallback)
      const int hb = D <= 320 ? 320 : 384;
      if (type == MFA_BACKWARD_QUERY && f32_or_inputs(MFA_O) && f32_or_inputs(MFA_dQ)) add(dq16_wide_variant(pq, pg, hb, &v), v);
      if (type == MFA_BACKWARD_KEY_VALUE && f32_or_inputs(MFA_dK) && f32_or_inputs(MFA_dV) &&
          kdesc->memoryPrecisions[MFA_dK] == kdesc->memoryPrecisions[MFA_dV])
        add(dkv16_wide_variant(pq, pg, hb, &v), v);
    }
  }
#ifdef MFA_DEV_VARIANTS
  // Developer builds only (make DEV=1 -> libmfa_hip_dev.so): environment knobs for A/B runs and timing-only ablations.
  // The product library contains neither this code nor the code objects it selects.
  {
    VariantInfo dev;
    bool have = false;
    const char *knob = std::getenv("MFA_FWD16_IMPL");
    if (type == MFA_FORWARD && knob && !candidates.empty()) {
      if (std::strcmp(knob, "v1") == 0) have = fwd16_variant(pq, bucket, &dev);
      else if (std::strncmp(knob, "v2:", 3) == 0) have = fwd16_v2_variant(pq, bucket, std::atoi(knob + 3), &dev);
      else if (std::strncmp(knob, "v3:", 3) == 0) have = fwd16_v3_variant(pq, bucket, std::atoi(knob + 3), &dev);
      else if (std::strncmp(knob, "v4:", 3) == 0) have = fwd16_v4_variant(pq, bucket, std::atoi(knob + 3), &dev);
      else if (std::strncmp(knob, "p4:", 3) == 0) have = fwd16_v3_variant(pq, bucket, 0, &dev) && fwd16_p4_variant(pq, bucket, std::atoi(knob + 3), &dev);
      else if (std::strncmp(knob, "p5:", 3) == 0) have = fwd16_v3_variant(pq, bucket, 0, &dev) && fwd16_p5_variant(pq, bucket, std::atoi(knob + 3), &dev);
    }
    knob = std::getenv("MFA_DKV16_IMPL");
    if (type == MFA_BACKWARD_KEY_VALUE && knob && !candidates.empty()) {
      const int pg = kdesc->memoryPrecisions[MFA_dO], bk = bucket < 64 ? 64 : bucket;
      if (std::strcmp(knob, "w4") == 0) have = dkv16_variant(pq, pg, bk, &dev);
      else if (std::strncmp(knob, "rs:", 3) == 0) have = dkv16_rs_variant(pq, pg, bk, std::atoi(knob + 3), &dev);
      else if (std::strncmp(knob, "p4:", 3) == 0)
        have = dkv16_rs_variant(pq, pg, bk, 0, &dev) &&
               dkv16_p4_variant(pq, pg, kdesc->memoryPrecisions[MFA_L], kdesc->memoryPrecisions[MFA_D], bk, std::atoi(knob + 3), &dev);
    }
    knob = std::getenv("MFA_DQ16_IMPL");
    if (type == MFA_BACKWARD_QUERY && knob && !candidates.empty() && std::strncmp(knob, "p4:", 3) == 0) {
      const int pg = kdesc->memoryPrecisions[MFA_dO];
      have = dq16_variant(pq, pg, bucket, &dev) && dq16_p4_variant(pq, pg, bucket, std::atoi(knob + 3), &dev);
    }
    if (type != MFA_FORWARD && std::getenv("MFA_BWD16_DISABLE")) candidates.clear();
    if (have) { candidates.clear(); candidates.push_back(dev); }
  }
#endif

  // ---- the parameter-table row decides among the candidates (AttentionDescriptor.swift:37-54 ->
  // AttentionKernel.swift:27-50: in the reference blockDimensions and cacheState ARE the kernel).  Exact match on
  // (parallelization, traversal, head block, cached left-hand operands) wins; otherwise the nearest candidate serves
  // the launch and mfa_attention_kernel_effective_descriptor reports what it really does -- unless the descriptor
  // asks for strictBlockDimensions, in which case an unmatched row is an error.
  // (the general kernel is not a candidate next to matrix-core variants: its block dimensions coincide with some of theirs,
  // and a table edit must not silently move a 16-bit problem onto fp32 arithmetic; it serves the launches they cannot)
  const bool fast = !candidates.empty();
  if (!fast) candidates.push_back(general);
  // left-hand operands of the kernel type: (first, second) = (Q, -) / (Q, dO) / (K, V)
  const int firstLeft = type == MFA_BACKWARD_KEY_VALUE ? MFA_K : MFA_Q;
  const int secondLeft = type == MFA_FORWARD ? MFA_Q : type == MFA_BACKWARD_QUERY ? MFA_dO : MFA_V;
  auto accumulators_cached_requested = [&]() {
    switch (type) {
      case MFA_FORWARD: return kdesc->cacheState[MFA_O] != 0;
      case MFA_BACKWARD_QUERY: return kdesc->cacheState[MFA_dQ] != 0;
      default: return kdesc->cacheState[MFA_dK] != 0 && kdesc->cacheState[MFA_dV] != 0;
    }
  };
  auto distance = [&](const VariantInfo &c) {
    int d = 0;
    if (c.headBlock < kdesc->headBlock) d += 8;   // (every candidate's head block holds D; among equals the smaller one wins below)
    if (c.parallelization != kdesc->parallelization) d += 4;
    if (c.traversal != kdesc->traversal) d += 2;
    if (c.cacheLeft != (kdesc->cacheState[firstLeft] != 0)) d += 1;
    if (type != MFA_FORWARD && c.cacheSecond != (kdesc->cacheState[secondLeft] != 0)) d += 1;
    // accumulators stay in registers in every code object except the paged one (D > 384: paged through the output buffers,
    // +Accumulate.swift:403-469): a row that asks for what the candidate does is an exact match either way
    if (accumulators_cached_requested() == c.pagedAccumulators) d += 1;
    return d;
  };
  size_t best = 0;
  for (size_t i = 1; i < candidates.size(); ++i) {
    const int di = distance(candidates[i]), db = distance(candidates[best]);
    if (di < db || (di == db && candidates[i].headBlock < candidates[best].headBlock)) best = i;
  }
  if (kdesc->strictBlockDimensions && distance(candidates[best]) != 0) {
    std::string have;
    for (const VariantInfo &c : candidates)
      have += " (" + std::to_string(c.parallelization) + ", " + std::to_string(c.traversal) + ", " + std::to_string(c.headBlock) +
              (c.cacheLeft ? ", left operands cached)" : ", left operands streamed)");
    return fail(MFA_ERR_UNSUPPORTED, "no code object implements block dimensions (" + std::to_string(kdesc->parallelization) + ", " +
                                         std::to_string(kdesc->traversal) + ", " + std::to_string(kdesc->headBlock) +
                                         ") with the requested cache state; compiled (parallelization, traversal, head):" + have);
  }
  const VariantInfo variant = candidates[best];

  mfa_attention_kernel *kernel = new mfa_attention_kernel();
  kernel->desc = *kdesc;
  kernel->variant = variant;
  kernel->fallback = general;
  kernel->hasFallback = fast;
  kernel->relayout = fast && relayout;
  kernel->effective = *kdesc;
  // register precisions the code object REALLY uses (AttentionDescriptor+Precisions.swift:149-215 describes Apple's choices):
  // the matrix-core kernels feed P (forward, dK/dV) and dS (backward) to the MFMA in the inputs' 16-bit type whatever
  // lowPrecisionIntermediates says; S, dP, the accumulators, L and D terms are fp32 registers in every kernel
  if (fast) {
    kernel->effective.registerPrecisions[MFA_P] = (int8_t)pq;
    if (type != MFA_FORWARD) kernel->effective.registerPrecisions[MFA_dS] = (int8_t)pq;
    kernel->effective.registerPrecisions[MFA_S] = MFA_FP32;
    if (type != MFA_FORWARD) kernel->effective.registerPrecisions[MFA_dP] = MFA_FP32;
  }
  kernel->effective.parallelization = variant.parallelization;
  kernel->effective.traversal = variant.traversal;
  kernel->effective.headBlock = variant.headBlock;
  // accumulators always live in registers on gfx950; left-hand operands per variant
  const int8_t accCached = variant.pagedAccumulators ? 0 : 1;
  switch (type) {
    case MFA_FORWARD:
      kernel->effective.cacheState[MFA_Q] = variant.cacheLeft;
      kernel->effective.cacheState[MFA_O] = accCached;
      break;
    case MFA_BACKWARD_QUERY:
      kernel->effective.cacheState[MFA_Q] = variant.cacheLeft;
      kernel->effective.cacheState[MFA_dO] = variant.cacheSecond;
      kernel->effective.cacheState[MFA_dQ] = accCached;
      break;
    default:
      kernel->effective.cacheState[MFA_K] = variant.cacheLeft;
      kernel->effective.cacheState[MFA_V] = variant.cacheSecond;
      kernel->effective.cacheState[MFA_dK] = kernel->effective.cacheState[MFA_dV] = accCached;
      break;
  }
  *out = kernel;
  return MFA_OK;
}

void mfa_attention_kernel_destroy(mfa_attention_kernel *kernel) { delete kernel; }

mfa_status mfa_attention_kernel_block_dimensions(const mfa_attention_kernel *kernel, uint16_t *parallelization,
                                                 uint16_t *traversal, uint16_t *headBlock) {
  if (!kernel) return fail(MFA_ERR_INVALID_ARGUMENT, "null kernel");
  if (parallelization) *parallelization = kernel->variant.parallelization;
  if (traversal) *traversal = kernel->variant.traversal;
  if (headBlock) *headBlock = kernel->variant.headBlock;
  return MFA_OK;
}

uint32_t mfa_attention_kernel_threadgroup_size(const mfa_attention_kernel *kernel) {
  return kernel ? kernel->variant.threads : 0;
}
uint32_t mfa_attention_kernel_threadgroup_memory_allocation(const mfa_attention_kernel *kernel) {
  return kernel ? kernel->variant.ldsBytes : 0;
}
const char *mfa_attention_kernel_variant(const mfa_attention_kernel *kernel) {
  return kernel ? kernel->variant.name : "";
}
const char *mfa_attention_kernel_fallback_variant(const mfa_attention_kernel *kernel) {
  return kernel && kernel->hasFallback ? kernel->fallback.name : "";
}
int mfa_attention_kernel_needs_workspace_for_fast_path(const mfa_attention_kernel *kernel) {
  return kernel && kernel->relayout ? 1 : 0;
}
mfa_status mfa_attention_kernel_effective_descriptor(const mfa_attention_kernel *kernel,
                                                     mfa_attention_kernel_descriptor *out) {
  if (!kernel || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  *out = kernel->effective;
  return MFA_OK;
}

// true if this launch satisfies the 16-byte-chunk requirements of the 16-bit MFMA kernels
static bool meets_fast_requirements(const mfa_attention_kernel *kernel, const KernelArgs &args) {
  const int type = kernel->desc.type;
  for (int slot = 0; slot < MFA_BUFFER_SLOTS; ++slot) {
    if (!slot_used(type, slot) || slot == SLOT_L || slot == SLOT_D) continue;
    const OperandView &v = args.op[slot];
    const int64_t per16 = 16 / (v.precision == PREC_FP32 ? 4 : 2);  // elements per 16 bytes
    // (transposed views: the kernels that read them in place gather what is not 16-byte aligned)
    const bool anyAlignment = v.transposed && kernel->variant.transposedInPlace;
    if (!anyAlignment && (reinterpret_cast<uintptr_t>(v.ptr) & 15) != 0) return false;
    if (!anyAlignment && (v.ld % per16 || v.headStride % per16 || v.batchStride % per16)) return false;
    // these kernels address one (head, batch) slice through a buffer descriptor with 32-bit byte
    // offsets (prefetch may run two tiles past the end): larger slices use the general kernels
    const bool rowOperand = (slot == SLOT_Q || slot == SLOT_O || slot == SLOT_dO || slot == SLOT_dQ);
    const uint64_t seq = rowOperand ? args.R : args.C;
    // (transposed views, forward only: D rows of `ld` elements; the prefetch runs two tiles = 128 elements along the last row)
    const uint64_t bytes = (v.transposed ? ((uint64_t)args.D * (uint64_t)v.ld + 192) : (seq + 192) * (uint64_t)v.ld) * (v.precision == PREC_FP32 ? 4u : 2u);
    if (bytes >= 0xFF000000ull) return false;
  }
  return true;
}

struct LaunchPlan {
  KernelArgs args;
  dim3 grid;
  const VariantInfo *variant;
  bool useFallback;
  uint32_t splits = 1;          // > 1: column-parallel forward through the caller's workspace
  float *wsO = nullptr, *wsML = nullptr;
  uint64_t workspaceNeeded = 0;
  // transposed operands served through row-major copies in the caller's workspace
  struct Relayout { int slot; OperandView user; void *copy; uint32_t seq; bool output; };
  Relayout relayouts[MFA_BUFFER_SLOTS];
  int nRelayouts = 0;
  uint32_t heads = 1, batches = 1;
  // the missing workspace is the ONLY reason this launch left the matrix-core kernel (every operand meets the alignment and
  // 32-bit slice-size requirements of the buffer descriptors): the condition under which the in-place backward kernels may take it
  bool onlyWorkspaceMissing = false;
};

// ---- re-layout pass: element (r, d) of a [seq][D] matrix between a transposed view ([D][seq], leading dimension ld) and a
// compact row-major copy ([seq][D]); 64 x 64 tiles through LDS, 16-byte accesses on both sides.  HBM-bound: 2 x seq x D x size bytes.
}  // extern "C"
template <typename E>
static __global__ __launch_bounds__(256) void attn_relayout(const char *tptr, char *cptr, uint32_t seq, uint32_t D, int64_t tld,
                                                            int64_t theadStride, int64_t tbatchStride, uint32_t heads, int toTransposed) {
  // 64 x 64 tile, 16-byte global accesses on both sides (V elements each), element-wise through LDS in between
  constexpr int V = 16 / sizeof(E), T = 64, CH = T / V;      // chunks of V elements per tile row
  __shared__ E tile[T][T + 2];
  const uint32_t tilesD = (D + T - 1) / T;
  const uint32_t r0 = (blockIdx.x / tilesD) * T, d0 = (blockIdx.x % tilesD) * T;
  const uint32_t head = blockIdx.y, batch = blockIdx.z;
  E *tview = const_cast<E *>(reinterpret_cast<const E *>(tptr)) + (int64_t)head * theadStride + (int64_t)batch * tbatchStride;
  E *copy = reinterpret_cast<E *>(cptr) + ((int64_t)batch * heads + head) * (int64_t)seq * D;
  typedef E vec __attribute__((ext_vector_type(V)));
  // the 16-byte path needs whole chunks inside the matrix and 16-byte aligned rows on both sides
  const bool wide = (seq % V) == 0 && (D % V) == 0 && (tld % V) == 0 && ((reinterpret_cast<uintptr_t>(tview) | reinterpret_cast<uintptr_t>(copy)) & 15) == 0;
  // transposed view: rows = d, contiguous along the sequence; copy: rows = sequence position, contiguous along d
  for (int c = threadIdx.x; c < T * CH; c += 256) {          // load: tile[d][r] always
    const uint32_t a = c / CH, b = (c % CH) * V;              // row a of the SOURCE tile, elements b .. b+V-1
    if (!toTransposed) {                                      // source = transposed view: row a = d, elements = sequence
      const uint32_t d = d0 + a, r = r0 + b;
      if (d < D && wide && r + V <= seq) {
        const vec x = *reinterpret_cast<const vec *>(tview + (int64_t)d * tld + r);
#pragma unroll
        for (int k = 0; k < V; ++k) tile[a][b + k] = x[k];
      } else if (d < D) {
        for (int k = 0; k < V; ++k) if (r + k < seq) tile[a][b + k] = tview[(int64_t)d * tld + r + k];
      }
    } else {                                                  // source = row-major copy: row a = sequence, elements = d
      const uint32_t r = r0 + a, d = d0 + b;
      if (r < seq && wide && d + V <= D) {
        const vec x = *reinterpret_cast<const vec *>(copy + (int64_t)r * D + d);
#pragma unroll
        for (int k = 0; k < V; ++k) tile[b + k][a] = x[k];
      } else if (r < seq) {
        for (int k = 0; k < V; ++k) if (d + k < D) tile[b + k][a] = copy[(int64_t)r * D + d + k];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < T * CH; c += 256) {          // store
    const uint32_t a = c / CH, b = (c % CH) * V;
    if (!toTransposed) {                                      // destination = copy: row a = sequence, elements = d
      const uint32_t r = r0 + a, d = d0 + b;
      if (r < seq && wide && d + V <= D) {
        vec x;
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] = tile[b + k][a];
        *reinterpret_cast<vec *>(copy + (int64_t)r * D + d) = x;
      } else if (r < seq) {
        for (int k = 0; k < V; ++k) if (d + k < D) copy[(int64_t)r * D + d + k] = tile[b + k][a];
      }
    } else {                                                  // destination = transposed view: row a = d, elements = sequence
      const uint32_t d = d0 + a, r = r0 + b;
      if (d < D && wide && r + V <= seq) {
        vec x;
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] = tile[a][b + k];
        *reinterpret_cast<vec *>(tview + (int64_t)d * tld + r) = x;
      } else if (d < D) {
        for (int k = 0; k < V; ++k) if (r + k < seq) tview[(int64_t)d * tld + r + k] = tile[a][b + k];
      }
    }
  }
}

extern "C" {

static void launch_relayout(const LaunchPlan &plan, const LaunchPlan::Relayout &r, hipStream_t stream) {
  const uint32_t D = plan.args.D;
  const dim3 grid(((r.seq + 63) / 64) * ((D + 63) / 64), plan.heads, plan.batches);
  const char *t = static_cast<const char *>(r.user.ptr);
  char *c = static_cast<char *>(r.copy);
  if (r.user.precision == PREC_FP32)
    hipLaunchKernelGGL(attn_relayout<uint32_t>, grid, dim3(256), 0, stream, t, c, r.seq, D, r.user.ld, r.user.headStride, r.user.batchStride, plan.heads, r.output ? 1 : 0);
  else
    hipLaunchKernelGGL(attn_relayout<uint16_t>, grid, dim3(256), 0, stream, t, c, r.seq, D, r.user.ld, r.user.headStride, r.user.batchStride, plan.heads, r.output ? 1 : 0);
}

static bool is_output_slot(int type, int slot) {
  switch (type) {
    case MFA_FORWARD: return slot == SLOT_O;
    case MFA_BACKWARD_QUERY: return slot == SLOT_dQ;
    default: return slot == SLOT_dK || slot == SLOT_dV;
  }
}

// bytes of workspace the row-major copies of this launch's transposed operands take (256-byte aligned each)
static uint64_t relayout_workspace_bytes(const mfa_attention_kernel *kernel, uint32_t row, uint32_t column, uint32_t heads, uint32_t batches) {
  const int type = kernel->desc.type;
  uint64_t total = 0;
  for (int slot = 0; slot < MFA_BUFFER_SLOTS; ++slot) {
    if (!slot_used(type, slot) || slot == SLOT_L || slot == SLOT_D) continue;
    const int op = slot_operand(slot);
    if (!kernel->desc.transposeState[op]) continue;
    const bool rowOperand = (op == MFA_Q || op == MFA_O || op == MFA_dO || op == MFA_dQ);
    const uint64_t seq = rowOperand ? row : column;
    const uint64_t esz = kernel->desc.memoryPrecisions[op] == MFA_FP32 ? 4 : 2;
    total += ((uint64_t)heads * batches * seq * kernel->desc.headDimension * esz + 255) & ~255ull;
  }
  return total;
}

// Column-parallel heuristic: split only when the row-parallel grid cannot fill the 256 CUs and the traversal is long enough to
// amortise the combine pass; keep >= 4 key tiles (256 keys) per piece.
// `target`: workgroups the variant wants in flight (512 = two per compute unit; 256 for the kernels that own a compute unit's whole
// register file and run one workgroup per compute unit -- a second round of half-length pieces would pay the per-block cost twice).
// Round 6 (tools/sweep_splits.py, profiles/r06_sweep_splits.txt): (1) the count is rounded DOWN to the target: 24 blocks x 11 pieces
// = 264 workgroups ran a second round of eight (N = 6144, one head: 33.8 us against 26.4 with 8 pieces); (2) more pieces shorten a
// piece's traversal by t_tile / s but every piece adds a slab to the combine pass (s x parallel x (D + 2) floats read back): the sum has
// its minimum at s^2 = K x traversal / parallel with K ~ 117 for every head dimension (a tile's time and a slab's bytes both grow
// with D), i.e. ~11 pieces for a square problem -- N = 4096 D = 64 one head forward: 21.0 us with 8 pieces, 23.5 with 16.
static uint32_t choose_splits(uint64_t blocks, uint32_t traversal, uint32_t parallel, uint32_t target = 512) {
  const uint32_t tiles = (traversal + 63) / 64;
  if (blocks >= 192 || tiles < 8) return 1;
  uint64_t s = target / blocks;
  uint64_t best = 1;
  while ((best + 1) * (best + 1) * (uint64_t)parallel <= 117ull * traversal + (uint64_t)parallel * (best + 1)) ++best;   // ~ round(sqrt(117 t / p))
  if (s > best) s = best;
#ifdef MFA_DEV_VARIANTS
  if (const char *knob = std::getenv("MFA_SPLITS")) s = (uint64_t)std::atoi(knob);   // developer library: sweep of the piece count (tools/sweep_splits.py)
#endif
  if (s > tiles / 4) s = tiles / 4;
  if (s > 64) s = 64;
  // equal pieces of whole 256-key blocks when a count between s / 2 and s gives them (what the persistent forward kernels' split
  // streams serve: attn_fwd16_p4p.hip, attn_fwd16_p6.hip)
  for (uint64_t c = s; c >= 2 && 2 * c > s; --c)
    if (traversal % (256 * c) == 0) { s = c; break; }
  return s < 2 ? 1 : (uint32_t)s;
}

// bytes of workspace a launch cut into `s` pieces needs
static uint64_t split_workspace_bytes(int type, uint32_t s, uint32_t heads, uint32_t batches, uint32_t row, uint32_t column, uint32_t D) {
  const uint64_t hb = (uint64_t)heads * batches;
  if (type == MFA_FORWARD) return (uint64_t)s * hb * row * (D + 2) * sizeof(float);          // O slabs + (m, l)
  if (type == MFA_BACKWARD_QUERY) return (uint64_t)s * hb * row * D * sizeof(float);         // dQ slabs
  return 2ull * s * hb * column * D * sizeof(float);                                         // dV slabs, then dK slabs
}

static mfa_status prepare_launch(const mfa_attention_kernel *kernel, void *const buffers[MFA_BUFFER_SLOTS],
                                 const mfa_launch_params *p, LaunchPlan *plan) {
  if (!kernel || !buffers || !p) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  if (p->row == 0 || p->column == 0) return fail(MFA_ERR_INVALID_ARGUMENT, "row and column must be non-zero");
  KernelArgs *args = &plan->args;
  const int type = kernel->desc.type;
  const uint32_t D = kernel->desc.headDimension;
  std::memset(args, 0, sizeof(*args));
  for (int slot = 0; slot < MFA_BUFFER_SLOTS; ++slot) {
    if (!slot_used(type, slot)) continue;
    const int op = slot_operand(slot);
    if (!buffers[slot])
      return fail(MFA_ERR_INVALID_ARGUMENT, std::string("buffer for operand ") + mfa_operand_name(op) + " is null");
    OperandView &v = args->op[slot];
    v.ptr = buffers[slot];
    v.precision = kernel->desc.memoryPrecisions[op];
    const bool vector = (op == MFA_L || op == MFA_D);
    v.transposed = vector ? 0 : kernel->desc.transposeState[op];
    // sequence length of the operand (AttentionKernel.swift:157-187)
    const bool rowOperand = (op == MFA_Q || op == MFA_O || op == MFA_dO || op == MFA_dQ || vector);
    const int64_t seq = rowOperand ? p->row : p->column;
    int64_t ld = p->leadingDimension[slot];
    if (ld == 0) ld = v.transposed ? seq : (int64_t)D;  // AttentionKernel.swift:189-204
    if (!vector) {
      if (!v.transposed && ld < (int64_t)D) return fail(MFA_ERR_INVALID_ARGUMENT, "leading dimension smaller than head dimension");
      if (v.transposed && ld < seq) return fail(MFA_ERR_INVALID_ARGUMENT, "leading dimension smaller than sequence length");
    }
    v.ld = vector ? 1 : ld;
    v.headStride = p->headStride[slot];
    v.batchStride = p->batchStride[slot];
  }
  args->R = p->row;
  args->C = p->column;
  args->D = D;
  args->scale = 1.0f / std::sqrt((float)D);
  args->scale2 = 1.44269504089f / std::sqrt((float)D);
  args->causal = p->causal ? 1 : 0;
  args->rowLen = p->rowLengths;
  args->colLen = p->columnLengths;
  args->mask = p->blockMask;
  args->maskWords = p->blockMaskWords;
  args->maskHeadStride = p->blockMaskHeadStride;
  args->maskBatchStride = p->blockMaskBatchStride;
  if (p->blockMask && p->blockMaskWords * 32ull * MASK_BLOCK_COLUMNS < p->column)
    return fail(MFA_ERR_INVALID_ARGUMENT, "blockMaskWords does not cover `column`");
  if (p->causal && p->column < p->row)
    return fail(MFA_ERR_INVALID_ARGUMENT, "causal masking requires column >= row");
  const uint32_t heads = p->heads ? p->heads : 1, batches = p->batches ? p->batches : 1;
  if (heads > 65535 || batches > 65535) return fail(MFA_ERR_INVALID_ARGUMENT, "heads and batches must be <= 65535");
  plan->heads = heads;
  plan->batches = batches;
  plan->nRelayouts = 0;
  bool relayoutMissing = false;
  if (kernel->relayout) {
    const uint64_t need = relayout_workspace_bytes(kernel, p->row, p->column, heads, batches);
    plan->workspaceNeeded = need;
    // per-batch lengths: the matrix-core kernels never write the padding rows of an output, so the write-back of a row-major
    // output copy (uninitialised workspace) would overwrite the caller's padding region -- such launches take the general
    // kernel, which reads and writes the transposed views in place
    const bool lengths = args->rowLen || args->colLen;
    if (!lengths && p->workspace && p->workspaceBytes >= need && (reinterpret_cast<uintptr_t>(p->workspace) & 255) == 0) {
      char *cursor = static_cast<char *>(p->workspace);
      for (int slot = 0; slot < MFA_BUFFER_SLOTS; ++slot) {
        if (!slot_used(type, slot) || slot == SLOT_L || slot == SLOT_D || !args->op[slot].transposed) continue;
        OperandView &v = args->op[slot];
        const int op = slot_operand(slot);
        const bool rowOperand = (op == MFA_Q || op == MFA_O || op == MFA_dO || op == MFA_dQ);
        const uint32_t seq = rowOperand ? p->row : p->column;
        LaunchPlan::Relayout &r = plan->relayouts[plan->nRelayouts++];
        r.slot = slot; r.user = v; r.copy = cursor; r.seq = seq; r.output = is_output_slot(type, slot);
        const uint64_t esz = v.precision == PREC_FP32 ? 4 : 2;
        cursor += ((uint64_t)heads * batches * seq * D * esz + 255) & ~255ull;
        v.ptr = r.copy; v.transposed = 0; v.ld = D;
        v.headStride = (int64_t)seq * D; v.batchStride = (int64_t)heads * seq * D;
      }
    } else {
      relayoutMissing = true;   // no (or too small a) workspace: the general kernel reads the transposed operands in place
    }
  }
  const bool otherReasons = !meets_fast_requirements(kernel, *args) || (args->causal && !kernel->variant.causal) ||
                            (args->mask && !kernel->variant.launchSparse && !kernel->variant.sparse) ||
                            // attn_dkv16_rs lists at most 4096 active 256-row blocks in LDS
                            (args->mask && type == MFA_BACKWARD_KEY_VALUE && p->row > 4096u * 256u);
  plan->useFallback = kernel->hasFallback && (relayoutMissing || otherReasons);
  plan->onlyWorkspaceMissing = kernel->hasFallback && relayoutMissing && !otherReasons;
  // strictBlockDimensions: a backward launch on 16-bit transposed operands that has no workspace for the re-layout path and is not
  // one the in-place kernels take would run the general fp32 kernel -- 20-50 x slower than the code object the descriptor selected
  // (the reference reads transposed operands in place at every head dimension, AttentionKernel.swift:189-204).  A strict caller
  // gets an error that names the remedy instead of the silent fallback
  if (kernel->desc.strictBlockDimensions && kernel->hasFallback && relayoutMissing && type != MFA_FORWARD &&
      !(plan->onlyWorkspaceMissing && bwd16_p4_tr_form(type, *args) != nullptr)) {
    return fail(MFA_ERR_UNSUPPORTED,
                std::string("strictBlockDimensions: this launch of ") + kernel->variant.name + " on transposed operands has no (or too small / "
                "misaligned) workspace and would run the general kernel " + kernel->fallback.name + "; pass a 256-byte aligned workspace of " +
                std::to_string(plan->workspaceNeeded) + " bytes (mfa_attention_kernel_workspace_size) for the re-layout path" +
                ((args->rowLen || args->colLen) ? " -- not available with per-batch lengths: store the operands row-major" : ""));
  }
  if (plan->useFallback && plan->nRelayouts) {   // (alignment, mask limits ...): the general kernel takes the user's views
    for (int i = 0; i < plan->nRelayouts; ++i) args->op[plan->relayouts[i].slot] = plan->relayouts[i].user;
    plan->nRelayouts = 0;
  }
  plan->variant = plan->useFallback ? &kernel->fallback : &kernel->variant;
  // parallelization dimension: rows for forward / backwardQuery, columns for backwardKeyValue
  // (SquareAttentionTest.swift:355-367)
  const uint32_t par = (type == MFA_BACKWARD_KEY_VALUE) ? p->column : p->row;
  uint32_t blocks = (par + plan->variant->parallelization - 1) / plan->variant->parallelization;
  // block-sparse and split launches may belong to a sibling kernel with its own workgroup shape (attn_dkv16_p4 keeps those of
  // attn_dkv16_rs)
  const uint32_t sibPar = plan->variant->siblingParallelization ? plan->variant->siblingParallelization : plan->variant->parallelization;
  const uint32_t siblingBlocks = (par + sibPar - 1) / sibPar;
  if (args->mask && plan->variant->launchSparse && !plan->useFallback) blocks = siblingBlocks;
  if ((uint64_t)blocks * heads * batches > 0x7FFFFFFFull) return fail(MFA_ERR_INVALID_ARGUMENT, "grid too large");
  plan->grid = dim3(blocks, heads, batches);
  plan->splits = 1;
  // Traversal-parallel launches through the caller's workspace, for grids that cannot fill the GPU: forward cuts
  // the key range (partial (O, m, l), online-softmax merge), backwardQuery the key range and backwardKeyValue
  // the row range (partial dQ / dK, dV in fp32 slabs, summed by attn_bwd_combine).
  const bool splittable = !plan->useFallback && !kernel->relayout && plan->variant->launchSplit && !args->rowLen && !args->colLen && !args->mask &&
                          (type != MFA_FORWARD || !args->causal);
  if (splittable) {
    // (the hand-placed backward kernels split dense launches themselves; causal ones stay with their siblings)
    const bool ownSplit = plan->variant->splitParallelization && !(args->causal && plan->variant->launchSplitCausal);
    const uint32_t sibBlocks = ownSplit ? (par + plan->variant->splitParallelization - 1) / plan->variant->splitParallelization : siblingBlocks;
    const uint32_t s = choose_splits((uint64_t)sibBlocks * heads * batches, type == MFA_BACKWARD_KEY_VALUE ? p->row : p->column,
                                     type == MFA_BACKWARD_KEY_VALUE ? p->column : p->row, plan->variant->splitTarget ? plan->variant->splitTarget : 512);
    if (s > 1) {
      plan->workspaceNeeded = split_workspace_bytes(type, s, heads, batches, p->row, p->column, D);
      if (p->workspace && p->workspaceBytes >= plan->workspaceNeeded &&
          (reinterpret_cast<uintptr_t>(p->workspace) & 15) == 0 && (D % 4) == 0 &&
          (uint64_t)sibBlocks * heads * batches * s <= 0x7FFFFFFFull) {
        plan->grid = dim3(sibBlocks, heads, batches);
        plan->splits = s;
        plan->wsO = static_cast<float *>(p->workspace);
        plan->wsML = plan->wsO + (uint64_t)s * heads * batches * p->row * D;   // forward only
      }
    }
  }
  return MFA_OK;
}

static auto split_launcher(const LaunchPlan &plan) -> decltype(plan.variant->launchSplit) {
  return (plan.args.causal && plan.variant->launchSplitCausal) ? plan.variant->launchSplitCausal : plan.variant->launchSplit;
}

// does this launch go to the in-place backward kernels? (transposed operands, no workspace)
static bool dev_in_place_backward(const mfa_attention_kernel *kernel, const LaunchPlan &plan) {
  // (a launch that ALSO misses the 16-byte alignment or the 32-bit slice size of the buffer descriptors stays with the general
  // kernel and its 64-bit addressing: the in-place kernels address every operand through such descriptors)
  if (!plan.useFallback || !plan.onlyWorkspaceMissing || !kernel->relayout || kernel->desc.type == MFA_FORWARD) return false;
#ifdef MFA_DEV_VARIANTS
  const char *knob = std::getenv("MFA_BWD16_TR");   // developer library: MFA_BWD16_TR=0 -- never (A/B runs against the general kernel)
  if (knob && std::strcmp(knob, "0") == 0) return false;
#endif
  return bwd16_p4_tr_form(kernel->desc.type, plan.args) != nullptr;
}

static mfa_status ensure_lds_attribute(mfa_attention_kernel *kernel, const LaunchPlan &plan) {
  const uint32_t attrBytes = plan.variant->attrLdsBytes > plan.variant->ldsBytes ? plan.variant->attrLdsBytes : plan.variant->ldsBytes;
  if (attrBytes <= 64 * 1024) return MFA_OK;
  int device = 0;
  hipError_t err = hipGetDevice(&device);
  if (err != hipSuccess) return hip_fail(err, "hipGetDevice");
  std::lock_guard<std::mutex> lock(kernel->attrMutex);
  uint64_t &mask = plan.useFallback ? kernel->attrDeviceMaskFallback : kernel->attrDeviceMask;
  if (device < 64 && (mask >> device) & 1ull) return MFA_OK;
  err = hipFuncSetAttribute(plan.variant->func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attrBytes);
  if (err == hipSuccess && plan.variant->funcCausal)
    err = hipFuncSetAttribute(plan.variant->funcCausal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attrBytes);
  if (err == hipSuccess && plan.variant->funcSparse)
    err = hipFuncSetAttribute(plan.variant->funcSparse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attrBytes);
  if (err == hipSuccess && plan.variant->funcSparseCausal)
    err = hipFuncSetAttribute(plan.variant->funcSparseCausal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attrBytes);
  if (err == hipSuccess && plan.variant->funcSplit)
    err = hipFuncSetAttribute(plan.variant->funcSplit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attrBytes);
  if (err == hipSuccess && plan.variant->funcSplitCausal)
    err = hipFuncSetAttribute(plan.variant->funcSplitCausal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attrBytes);
  if (err != hipSuccess) return hip_fail(err, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  if (device < 64) mask |= 1ull << device;
  return MFA_OK;
}

mfa_status mfa_attention_kernel_launch(const mfa_attention_kernel *kernel, void *const buffers[MFA_BUFFER_SLOTS],
                                       const mfa_launch_params *params, void *stream) {
  LaunchPlan plan;
  mfa_status st = prepare_launch(kernel, buffers, params, &plan);
  if (st != MFA_OK) return st;
  // a transposed backward launch without a workspace goes to the kernels that read the operands in place when they take it
  // (attn_bwd16_p4_tr.hip; AttentionKernel.swift:189-204: the reference reads transposed operands in place in every kernel)
  if (dev_in_place_backward(kernel, plan)) {
    bwd16_p4_tr_launch(kernel->desc.type, plan.args, plan.heads, plan.batches, (hipStream_t)stream, kernel->desc.registerPrecisions[MFA_P] > MFA_FP32);
    hipError_t derr = hipGetLastError();
#ifdef MFA_DEV_VARIANTS
    const char *knob = std::getenv("MFA_BWD16_TR");
    if (knob && std::strcmp(knob, "verbose") == 0)
      std::fprintf(stderr, "mfa: %s on transposed operands in place\n", kernel->desc.type == MFA_BACKWARD_QUERY ? "attn_dq16_p4_tr" : "attn_dkv16_p4_tr");
#endif
    return derr == hipSuccess ? MFA_OK : hip_fail(derr, "attn_bwd16_p4_tr");
  }
  st = ensure_lds_attribute(const_cast<mfa_attention_kernel *>(kernel), plan);
  if (st != MFA_OK) return st;
  for (int i = 0; i < plan.nRelayouts; ++i)
    if (!plan.relayouts[i].output) launch_relayout(plan, plan.relayouts[i], (hipStream_t)stream);
  if (plan.splits > 1) split_launcher(plan)(plan.grid, plan.splits, plan.wsO, plan.wsML, (hipStream_t)stream, plan.args);
  else if (plan.args.mask && plan.variant->launchSparse) plan.variant->launchSparse(plan.grid, (hipStream_t)stream, plan.args);
  else if (plan.args.causal && plan.variant->launchCausal) plan.variant->launchCausal(plan.grid, (hipStream_t)stream, plan.args);
  else plan.variant->launch(plan.grid, (hipStream_t)stream, plan.args);
  for (int i = 0; i < plan.nRelayouts; ++i)
    if (plan.relayouts[i].output) launch_relayout(plan, plan.relayouts[i], (hipStream_t)stream);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return hip_fail(err, plan.variant->name);
  return MFA_OK;
}

mfa_status mfa_attention_kernel_launch_form(const mfa_attention_kernel *kernel, void *const buffers[MFA_BUFFER_SLOTS],
                                            const mfa_launch_params *params, char *out, size_t capacity) {
  if (!out || capacity == 0) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  LaunchPlan plan;
  mfa_status st = prepare_launch(kernel, buffers, params, &plan);
  if (st != MFA_OK) return st;
  std::string text;
  if (plan.nRelayouts) text += "attn_relayout x" + std::to_string(plan.nRelayouts) + " + ";
  if (dev_in_place_backward(kernel, plan)) {
    std::snprintf(out, capacity, "%s", bwd16_p4_tr_form(kernel->desc.type, plan.args));
    return MFA_OK;
  }
  if (plan.useFallback) {
    text += std::string(plan.variant->name) + " (general kernel: the launch does not meet the requirements of " + kernel->variant.name + ")";
  } else if (plan.splits > 1) {
    text += std::string(plan.variant->name) + " column-parallel x" + std::to_string(plan.splits) + " + combine";
    const bool own = plan.variant->splitParallelization && !(plan.args.causal && plan.variant->launchSplitCausal);
    const char *pieces = plan.variant->splitForm ? plan.variant->splitForm(plan.args, plan.splits) : nullptr;
    if (pieces) text += std::string(" (") + pieces + ")";
    else if (!own && plan.variant->siblingName) text += std::string(" (pieces by the sibling kernel ") + plan.variant->siblingName + ")";
  } else {
    const bool sparse = plan.args.mask && plan.variant->launchSparse;
    const char *form = (!sparse && plan.variant->launchForm) ? plan.variant->launchForm(plan.args) : nullptr;
    text += form ? form : plan.variant->name;
    if (sparse) text += plan.variant->siblingName ? std::string(" (block-sparse sibling ") + plan.variant->siblingName + ")" : std::string(" (block-sparse code object)");
  }
  std::snprintf(out, capacity, "%s", text.c_str());
  return MFA_OK;
}

mfa_status mfa_attention_kernel_workspace_size(const mfa_attention_kernel *kernel, const mfa_launch_params *params,
                                               uint64_t *bytes) {
  if (!kernel || !params || !bytes) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  *bytes = 0;
  if (params->row == 0 || params->column == 0) return fail(MFA_ERR_INVALID_ARGUMENT, "row and column must be non-zero");
  if (kernel->relayout) {   // row-major copies of the transposed operands (without them the launch runs the general kernel)
    *bytes = relayout_workspace_bytes(kernel, params->row, params->column, params->heads ? params->heads : 1, params->batches ? params->batches : 1);
    return MFA_OK;
  }
  if (!kernel->variant.launchSplit) return MFA_OK;
  const int type = kernel->desc.type;
  if (params->rowLengths || params->columnLengths || params->blockMask || (type == MFA_FORWARD && params->causal)) return MFA_OK;
  const uint32_t heads = params->heads ? params->heads : 1, batches = params->batches ? params->batches : 1;
  const uint32_t par = (type == MFA_BACKWARD_KEY_VALUE) ? params->column : params->row;
  const bool ownSplit = kernel->variant.splitParallelization && !(params->causal && kernel->variant.launchSplitCausal);
  const uint32_t wgPar = ownSplit ? kernel->variant.splitParallelization
                                  : kernel->variant.siblingParallelization ? kernel->variant.siblingParallelization : kernel->variant.parallelization;
  const uint32_t blocks = (par + wgPar - 1) / wgPar;
  const uint32_t s = choose_splits((uint64_t)blocks * heads * batches, type == MFA_BACKWARD_KEY_VALUE ? params->row : params->column,
                                   type == MFA_BACKWARD_KEY_VALUE ? params->column : params->row, kernel->variant.splitTarget ? kernel->variant.splitTarget : 512);
  if (s > 1) *bytes = split_workspace_bytes(type, s, heads, batches, params->row, params->column, kernel->desc.headDimension);
  return MFA_OK;
}

mfa_status mfa_attention_kernel_time(const mfa_attention_kernel *kernel, void *const buffers[MFA_BUFFER_SLOTS],
                                     const mfa_launch_params *params, void *stream, int warmup, int iterations,
                                     float *milliseconds) {
  if (!milliseconds || iterations <= 0 || warmup < 0) return fail(MFA_ERR_INVALID_ARGUMENT, "bad timing arguments");
  LaunchPlan plan;
  mfa_status st = prepare_launch(kernel, buffers, params, &plan);
  if (st != MFA_OK) return st;
  st = ensure_lds_attribute(const_cast<mfa_attention_kernel *>(kernel), plan);
  if (st != MFA_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t start, stop;
  hipError_t err = hipEventCreate(&start);
  if (err != hipSuccess) return hip_fail(err, "hipEventCreate");
  err = hipEventCreate(&stop);
  if (err != hipSuccess) { (void)hipEventDestroy(start); return hip_fail(err, "hipEventCreate"); }
  auto go = [&]() {
    if (dev_in_place_backward(kernel, plan)) {
      bwd16_p4_tr_launch(kernel->desc.type, plan.args, plan.heads, plan.batches, s, kernel->desc.registerPrecisions[MFA_P] > MFA_FP32);
      return;
    }
    for (int i = 0; i < plan.nRelayouts; ++i)
      if (!plan.relayouts[i].output) launch_relayout(plan, plan.relayouts[i], s);
    if (plan.splits > 1) split_launcher(plan)(plan.grid, plan.splits, plan.wsO, plan.wsML, s, plan.args);
    else if (plan.args.mask && plan.variant->launchSparse) plan.variant->launchSparse(plan.grid, s, plan.args);
    else if (plan.args.causal && plan.variant->launchCausal) plan.variant->launchCausal(plan.grid, s, plan.args);
    else plan.variant->launch(plan.grid, s, plan.args);
    for (int i = 0; i < plan.nRelayouts; ++i)
      if (plan.relayouts[i].output) launch_relayout(plan, plan.relayouts[i], s);
  };
  for (int i = 0; i < warmup; ++i) go();
  (void)hipEventRecord(start, s);
  for (int i = 0; i < iterations; ++i) go();
  (void)hipEventRecord(stop, s);
  err = hipEventSynchronize(stop);
  if (err == hipSuccess) err = hipGetLastError();
  if (err == hipSuccess) err = hipEventElapsedTime(milliseconds, start, stop);
  (void)hipEventDestroy(start);
  (void)hipEventDestroy(stop);
  if (err != hipSuccess) return hip_fail(err, plan.variant->name);
  return MFA_OK;
}

mfa_status mfa_device_count(int *count) {
  if (!count) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  hipError_t err = hipGetDeviceCount(count);
  if (err != hipSuccess) { *count = 0; return hip_fail(err, "hipGetDeviceCount"); }
  return MFA_OK;
}

mfa_status mfa_device_name(int device, char *out, size_t capacity) {
  if (!out || capacity == 0) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  hipDeviceProp_t prop;
  hipError_t err = hipGetDeviceProperties(&prop, device);
  if (err != hipSuccess) return hip_fail(err, "hipGetDeviceProperties");
  std::strncpy(out, prop.gcnArchName, capacity - 1);
  out[capacity - 1] = '\0';
  return MFA_OK;
}

} // extern "C"
