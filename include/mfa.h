/*
 * mfa.h -- C ABI of the MI355X-native FlashAttention kernel suite.
 *
 * This is the drop-in boundary for ONE path of philipturner/metal-flash-attention:
 *     AttentionDescriptor -> AttentionKernelDescriptor -> AttentionKernel -> dispatch
 * The reference has no FFI: its boundary is the public Swift API plus the Metal calls its
 * callers make (compile source, bind buffers 0-9, set threadgroup memory, dispatch).  Each
 * entry point below names the reference interface it replaces (paths relative to
 * /root/reference).  The reference-side binding a maintainer would add is in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only (no torch / HIP types in signatures; a stream is
 * passed as void* = hipStream_t).  Every function returns an mfa_status; nothing aborts the
 * process (the reference uses fatalError).  The library never allocates device memory and
 * keeps no hidden workspace: the caller owns all ten operand buffers, as in the reference.
 *
 * Numerics contract (where this target departs from the reference's arithmetic):
 *   - Inputs must be finite.  The 16-bit matrix-core forward kernels are compiled with -ffinite-math-only (the softmax
 *     path itself never creates Inf / NaN: finite mask value, finite running-maximum start); an Inf or NaN in Q, K or V
 *     is undefined behaviour there, not propagated.  The fp32-arithmetic (general) kernels propagate them.
 *   - Deferred rescale.  The 16-bit forward kernels raise the running row maximum m -- and rescale O and l -- only
 *     when a block maximum exceeds m by more than 8 (log2 units); until then P = 2^(S - m) may reach 2^8 before it is
 *     rounded to 16 bits.  The reference corrects whenever the maximum grows (+Softmax.swift:290-301).  L = m + log2 l
 *     is unaffected; measured |dO| against the reference's rule: 7e-4 at N = 4096.
 *   - P and dS are rounded to the inputs' 16-bit type in every 16-bit matrix-core kernel (they are MFMA operands),
 *     also when lowPrecisionIntermediates is 0 and the reference would keep them in FP32 registers
 *     (+Precisions.swift:201-205).  S, the accumulators, L and D arithmetic are fp32.
 *   - lowPrecisionIntermediates = 1 (the reference then holds P, and with FP16 also S, in 16-bit registers) lets the
 *     hand-placed kernels (forward D <= 256, backward D <= 128) multiply one operand of S = Q K^T
 *     by log2(e)/sqrt(D) once, rounded to the inputs' type (forward and backwardQuery: Q; backwardKeyValue: K), instead
 *     of scaling every score in fp32: L moves by up to ~2e-3 (BF16) / 2e-4 (FP16) natural-log units, P by the same
 *     relative amount.  With the flag clear the scale is applied in fp32 per score.
 *   - lowPrecisionIntermediates = 1, forward, D <= 64 (attn_fwd16_p6): the softmax denominator l is the sum of the P values
 *     AFTER their rounding to the inputs' 16-bit type, accumulated in fp32 by the matrix pipe (L^T += ONES P^T) -- the
 *     reference's mixed mode also sums its 16-bit P (+Precisions.swift:149-215); O = (sum P v) / (sum P) then uses ONE set of
 *     P values.  With the flag clear l is the fp32 sum of the unrounded P.
 *   - backwardKeyValue, D <= 128: the per-row terms L and D enter S and dP through the matrix pipe as the sum of two
 *     16-bit values (16 / 22 bits of mantissa for BF16 / FP16 inputs): an absolute error of ~2^-16 |L| in the exponent of P.
 *   - FP16 Q, K, V with BF16 dO (the reference's own low-precision mix), backwardKeyValue, D <= 128: the two products that
 *     read dO run in BF16 -- V is rounded to BF16 once per workgroup and P is packed to BF16 for dV -- while S and dK stay
 *     FP16; the other kernels convert dO to FP16 instead (exact in FP16's range).
 *   - Transposed operands (transposeState): the forward kernel reads and writes them in place on the 16-bit matrix cores
 *     (any leading dimension).  The backward kernels of the 128 bucket (64 < D <= 128) read them in place too when the launch
 *     is whole tiles of 16-byte aligned rows and carries no workspace (attn_dq16_p4_tr / attn_dkv16_p4_tr); every other
 *     transposed backward launch runs on the matrix cores when it is given a workspace
 *     (mfa_attention_kernel_needs_workspace_for_fast_path: re-layout pass), and on the fp32-arithmetic kernels without one --
 *     20-50 x slower; a kernel created with strictBlockDimensions returns MFA_ERR_UNSUPPORTED for such a launch instead, the
 *     message naming the workspace size.
 *   - Head dimensions: any.  16-bit matrix-core code objects exist up to D = 256 for all three kernel types since round 4 and -- round 6 --
 *     up to D = 384 (head blocks 320 / 384: the `| 384 | ... |` rows of the reference's mixed tables,
 *     AttentionDescriptor+Parameters.swift:113, :120, :153-201): attn_fwd16w_*, attn_dq16w_*, attn_dkv16w_* (dense, causal, per-batch
 *     lengths, fused 16-bit outputs; transposed operands through the re-layout pass -- at these head blocks the forward kernel too:
 *     mfa_attention_kernel_needs_workspace_for_fast_path is 1 for it).  Kernels with FP32
 *     operands at D > 128, and block-masked launches at D > 256, run on the fp32-arithmetic kernels, accumulators in registers; D > 384 (beyond the reference's tables, which fall through
 *     to their last row, +Parameters.swift:60-65) runs D-blocked kernels that page the accumulators through the output
 *     buffers like the reference does (+Accumulate.swift:403-469) -- O, dQ, dK, dV must then be FP32 (the fused 16-bit
 *     output cast is MFA_ERR_UNSUPPORTED there).
 */
#ifndef MFA_H
#define MFA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFA_ABI_VERSION 6

/* ---- status codes (replace fatalError, e.g. AttentionKernel.swift:33,
 *      AttentionDescriptor.swift:45/72/90/97, AttentionParameterRow.swift:50/57/100) -------- */
typedef enum mfa_status {
  MFA_OK = 0,
  MFA_ERR_INCOMPLETE_DESCRIPTOR = 1, /* "Descriptor was incomplete." */
  MFA_ERR_INVALID_ARGUMENT = 2,      /* bad enum, null pointer, unexpected cached operand ... */
  MFA_ERR_UNSUPPORTED = 3,           /* valid request this build has no kernel for */
  MFA_ERR_HIP = 4,                   /* HIP runtime error; see mfa_last_error_string() */
  MFA_ERR_PARSE = 5                  /* malformed parameter table text */
} mfa_status;

/* ---- GEMMOperandPrecision (Sources/FlashAttention/GEMM/GEMMOperandPrecision.swift:33-60) */
typedef enum mfa_precision { MFA_FP32 = 0, MFA_FP16 = 1, MFA_BF16 = 2 } mfa_precision;
const char *mfa_precision_name(int precision); /* "float" / "half" / "bfloat" (:39-48) */
int mfa_precision_size(int precision);          /* 4 / 2 / 2               (:51-59) */

/* ---- AttentionKernelType (Sources/FlashAttention/Attention/AttentionKernelType.swift:10-23) */
typedef enum mfa_kernel_type {
  MFA_FORWARD = 0,           /* computes O and L */
  MFA_BACKWARD_QUERY = 1,    /* computes D and dQ; depends on L */
  MFA_BACKWARD_KEY_VALUE = 2 /* computes dK and dV; depends on L and D */
} mfa_kernel_type;

/* ---- AttentionOperand (Sources/FlashAttention/Attention/AttentionOperand.swift:9-71) ---- */
typedef enum mfa_operand {
  MFA_Q = 0, MFA_K, MFA_S, MFA_P, MFA_V, MFA_O, MFA_L, MFA_D,
  MFA_dO, MFA_dV, MFA_dP, MFA_dS, MFA_dK, MFA_dQ,
  MFA_OPERAND_COUNT
} mfa_operand;
const char *mfa_operand_name(int operand);   /* AttentionOperand.description (:30-50) */
/* AttentionOperand.bufferBinding (:52-71): Q0 K1 V2 O3 L4 D5 dO6 dV7 dK8 dQ9, -1 if none */
int mfa_operand_buffer_binding(int operand);
#define MFA_BUFFER_SLOTS 10

/* ---- AttentionDescriptor (Attention/AttentionDescriptor/AttentionDescriptor.swift:10-27) -- */
typedef struct mfa_attention_descriptor {
  uint8_t lowPrecisionInputs;        /* Q, K, V, dO          (:12) */
  uint8_t lowPrecisionIntermediates; /* S, P, L, D, dP, dS   (:15) */
  uint8_t hasMatrixDimensions;       /* Swift optional tuple (:20) */
  uint8_t hasTransposeState;         /* Swift optional tuple (:22) */
  uint32_t row, column;              /* matrixDimensions.row / .column */
  uint16_t head;                     /* matrixDimensions.head */
  uint8_t transposeQ, transposeK, transposeV, transposeO;
  /* Extension (not in the reference): 16-bit storage type used when lowPrecisionInputs is set.
   * MFA_FP16 reproduces +Precisions.swift:13-17 exactly (Q,K,V FP16; dO BF16).
   * MFA_BF16 stores Q,K,V,dO all as BF16 (what BASELINE.json's bf16 configs use). */
  uint8_t lowPrecisionInputType;
  /* Extension (the reference keeps O, dV, dK, dQ in FP32 and leaves the cast to
   * clients, +Precisions.swift:119-143): non-zero = the kernels store O, dQ, dK, dV directly in
   * lowPrecisionInputType (BF16 by truncation, FP16 round-to-nearest) and backwardQuery reads that O. */
  uint8_t lowPrecisionOutputs;
  uint8_t reserved[2];
} mfa_attention_descriptor;
void mfa_attention_descriptor_init(mfa_attention_descriptor *desc); /* AttentionDescriptor() */

/* ---- AttentionKernelDescriptor (Attention/AttentionKernelDescriptor.swift:8-49) ----------
 * Dictionaries keyed by AttentionOperand become arrays indexed by mfa_operand; -1 = absent. */
typedef struct mfa_attention_kernel_descriptor {
  uint8_t hasBlockDimensions;
  uint8_t hasHeadDimension;
  uint16_t parallelization, traversal, headBlock; /* blockDimensions (:9-10) */
  uint16_t headDimension;                         /* (:16) */
  int8_t cacheState[MFA_OPERAND_COUNT];           /* (:13)  -1 / 0 / 1 */
  int8_t memoryPrecisions[MFA_OPERAND_COUNT];     /* (:18)  -1 or mfa_precision */
  int8_t registerPrecisions[MFA_OPERAND_COUNT];   /* (:26) */
  int8_t transposeState[MFA_OPERAND_COUNT];       /* (:42)  -1 / 0 / 1 */
  int8_t preferAsyncCache;                        /* (:21)  -1 / 0 / 1 */
  int8_t preferAsyncLoad;                         /* (:24) */
  int8_t type;                                    /* (:44)  -1 or mfa_kernel_type */
  /* Extension.  In the reference blockDimensions and cacheState ARE the kernel (AttentionKernel.swift:27-50); a
   * pre-compiled suite can only honour tuples that exist.  0 (default): mfa_attention_kernel_create takes the nearest
   * compiled variant and mfa_attention_kernel_effective_descriptor reports it.  Non-zero: a tuple no code object
   * implements is MFA_ERR_UNSUPPORTED (the message lists the compiled tuples). */
  int8_t strictBlockDimensions;
} mfa_attention_kernel_descriptor;
void mfa_attention_kernel_descriptor_init(mfa_attention_kernel_descriptor *kdesc);

/* AttentionDescriptor.memoryPrecisions / .registerPrecisions
 * (AttentionDescriptor+Precisions.swift:10-146, :149-215).  out[MFA_OPERAND_COUNT], -1 = absent. */
mfa_status mfa_attention_descriptor_memory_precisions(const mfa_attention_descriptor *desc, int8_t *out);
mfa_status mfa_attention_descriptor_register_precisions(const mfa_attention_descriptor *desc, int8_t *out);

/* AttentionDescriptor.kernelDescriptor(type:) (AttentionDescriptor.swift:33-130): parameter
 * table lookup by head dimension, head-block clamp to pad8(D), cache-state validation,
 * gradient transposes inherited from the primal operands, precision maps. */
mfa_status mfa_attention_descriptor_kernel_descriptor(const mfa_attention_descriptor *desc,
                                                      int type,
                                                      mfa_attention_kernel_descriptor *out);

/* ---- parameter tables (AttentionDescriptor+Parameters.swift:13-66, :77-285;
 *      text format of AttentionParameterRow.parseTable, AttentionParameterRow.swift:22-74):
 *      "| max D | parallelization | traversal | head | cached operands |" one row per line.
 *      `mixed` = 1 selects the table consulted whenever `lowPrecisionInputs` is set -- Q, K, V in a 16-bit type, i.e. the
 *      16-bit matrix-core code objects -- whatever `lowPrecisionIntermediates` says; `mixed` = 0 the table of FP32 inputs.
 *      DEPARTURE from the reference, which takes its mixed tables only when BOTH flags are set (+Parameters.swift:16):
 *      there the tables follow the register footprint that FP16 intermediates halve; on gfx950 S, P and the accumulators are
 *      fp32 registers in every kernel and what changes the code object is the storage type of the inputs. */
mfa_status mfa_parameter_table_get(int type, int mixed, char *out, size_t capacity);
mfa_status mfa_parameter_table_set(int type, int mixed, const char *text); /* validates, then installs */
mfa_status mfa_parameter_table_reset(void);                                /* back to built-in gfx950 tables */
typedef struct mfa_parameter_row {
  uint16_t maximumHeadDimension;
  uint16_t parallelization, traversal, head;
  int8_t cached[MFA_OPERAND_COUNT]; /* 1 if listed */
} mfa_parameter_row;
/* parseTable + row(table:) in one call (AttentionParameterRow.swift:22-74, +Parameters.swift:41-66) */
mfa_status mfa_parameter_table_select(const char *text, uint16_t headDimension, mfa_parameter_row *out);

/* ---- AttentionKernel (Attention/AttentionKernel/AttentionKernel.swift:10-51) --------------
 * In the reference this object generates Metal source (createSource, +Source.swift:11-55)
 * which the CALLER compiles and dispatches.  Here it resolves to a pre-compiled gfx950 code
 * object; mfa_attention_kernel_launch replaces createSource + makeLibrary + pipeline +
 * setBuffer x10 + setThreadgroupMemoryLength + dispatchThreadgroups
 * (Tests/.../SquareAttentionTest.swift:244-260, :319-368). */
typedef struct mfa_attention_kernel mfa_attention_kernel;
mfa_status mfa_attention_kernel_create(const mfa_attention_kernel_descriptor *kdesc,
                                       mfa_attention_kernel **out);
void mfa_attention_kernel_destroy(mfa_attention_kernel *kernel);
/* AttentionKernel.blockDimensions (:22-23) -- the dimensions the selected code object really uses */
mfa_status mfa_attention_kernel_block_dimensions(const mfa_attention_kernel *kernel,
                                                 uint16_t *parallelization, uint16_t *traversal,
                                                 uint16_t *headBlock);
/* AttentionKernel.threadgroupSize (:268-270): work-items per workgroup (64 x waves) */
uint32_t mfa_attention_kernel_threadgroup_size(const mfa_attention_kernel *kernel);
/* AttentionKernel.threadgroupMemoryAllocation (:25, :272-363): LDS bytes per workgroup */
uint32_t mfa_attention_kernel_threadgroup_memory_allocation(const mfa_attention_kernel *kernel);
/* name of the selected HIP kernel variant, e.g. "attn_fwd_bf16_d128_r256" (diagnostics) */
const char *mfa_attention_kernel_variant(const mfa_attention_kernel *kernel);
/* name of the general (fp32-arithmetic) code object that serves the launches the selected variant cannot take (misaligned
 * pointers or strides, transposed operands without a workspace ...); "" if the selected variant is the general one */
const char *mfa_attention_kernel_fallback_variant(const mfa_attention_kernel *kernel);
/* Transposed operands (transposeState, AttentionKernelDescriptor.swift:30-41) and the 16-bit matrix-core kernels.
 * FORWARD: code objects that read Q^T, K^T, V^T and write O^T where they lie (AttentionKernel.swift:189-204: no scratch),
 * one per pattern of (K, V) and head-dimension bucket; rows that are not 16-byte aligned (an odd sequence length as the
 * leading dimension) are gathered element-wise -- slower, never the fp32-arithmetic kernel.  This function returns 0 (256 < D <= 384,
 * round 6: no in-place code objects at these head blocks -- it returns 1 and the forward launch takes the re-layout path below).
 * BACKWARD (dQ, dK/dV): those kernels read row-major tiles, so a launch with `workspace` first re-lays every transposed
 * operand out into the workspace (one HBM-bound pass per operand, 2 x sequence x D x size bytes; transposed outputs are written
 * back the same way) and then runs the matrix-core code object; without a workspace the in-place backward kernels take the
 * launches they can (128 bucket, whole aligned tiles) and the general kernel reads the transposed operands in place otherwise.  Non-zero = this kernel is in that situation (size the workspace with mfa_attention_kernel_workspace_size). */
int mfa_attention_kernel_needs_workspace_for_fast_path(const mfa_attention_kernel *kernel);
/* The descriptor as the selected code object really executes it: the Swift struct lets callers
 * request any block dimensions / cache state (AttentionKernelDescriptor.swift:9-13); a
 * pre-compiled suite honours the nearest compiled variant and reports it here. */
mfa_status mfa_attention_kernel_effective_descriptor(const mfa_attention_kernel *kernel,
                                                     mfa_attention_kernel_descriptor *out);

/* ---- launch ---------------------------------------------------------------------------------
 * row/column replace AttentionDescriptor.setFunctionConstants (AttentionDescriptor.swift:139-148).
 * heads/batches + strides are this library's multi-head extension of the recipe in
 * AttentionKernelDescriptor.swift:37-41 (leading dimension D -> D*H): grid = (blocks, heads,
 * batches); operand X of head h, batch b starts at X + h*headStride[slot] + b*batchStride[slot]
 * (element units).  leadingDimension[slot] == 0 means the reference default
 * (AttentionKernel.swift:189-204): D if not transposed, the sequence length if transposed.
 * L and D are vectors: leadingDimension is ignored for slots 4 and 5. */
typedef struct mfa_launch_params {
  uint32_t row, column;
  uint32_t heads, batches;                    /* 0 is treated as 1 */
  int64_t leadingDimension[MFA_BUFFER_SLOTS];
  int64_t headStride[MFA_BUFFER_SLOTS];
  int64_t batchStride[MFA_BUFFER_SLOTS];
  /* Optional caller-owned device scratch (extension).  A launch with too few workgroups to fill the
   * GPU (e.g. the reference's single-head benchmark, SquareAttentionTest.swift:159-165) is then run
   * traversal-parallel: forward and backwardQuery cut the key range, backwardKeyValue the row range;
   * the pieces' partial results (forward: un-normalised O, m, l; backward: fp32 gradient slabs) go to
   * this scratch and a second small kernel merges them -- no atomics.  NULL (default) = never split.
   * Second use: row-major copies of transposed operands for the backward matrix-core kernels (see
   * mfa_attention_kernel_needs_workspace_for_fast_path; such launches are not split; 256-byte aligned).
   * Size it with mfa_attention_kernel_workspace_size.  Contents need no initialisation. */
  void *workspace;
  uint64_t workspaceBytes;
  /* Causal mask (extension; the reference is unmasked and names masks as its first extension,
   * README.md:7): non-zero = row r attends column c iff c <= r + (column - row), i.e. lower-triangular
   * for square problems.  Requires column >= row.  Applies to all three kernel types. */
  uint32_t causal;
  uint32_t reserved;
  /* Variable sequence lengths (extension): device arrays of `batches` uint32
   * entries, or NULL.  Batch entry b uses the first rowLengths[b] rows and columnLengths[b] columns of its
   * row x column problem (entries are clamped to row / column); the rest of its buffers is padding that is
   * neither read into the result nor written.  With `causal` the diagonal of entry b is placed by ITS lengths: row r
   * sees column c iff c <= r + max(columnLengths[b] - rowLengths[b], 0) -- an entry with fewer columns than rows (which the
   * host cannot see: the arrays live on the device) gets offset 0, so every row keeps at least one visible key.
   * All three kernel types; launches with lengths are never column-split. */
  const uint32_t *rowLengths;
  const uint32_t *columnLengths;
  /* Block-sparse mask (extension; the reference names block sparsity next to masks, README.md:7, :210):
   * device bitmap with one bit per block of MFA_MASK_BLOCK_ROWS (256) rows x MFA_MASK_BLOCK_COLUMNS (128)
   * columns, row blocks major, `blockMaskWords` 32-bit words per row block (bit b of word w = column block
   * 32 w + b); set = the block is attended (subject to `causal` and the lengths), clear = the block is never
   * loaded.  NULL = dense.  Strides in words select a mask per head / batch entry (0 = shared).  Rows none of
   * whose blocks are set get O = 0, dQ = 0 and a hugely negative L.  All three kernel types. */
  const uint32_t *blockMask;
  uint32_t blockMaskWords;
  uint32_t reserved2;
  int64_t blockMaskHeadStride;
  int64_t blockMaskBatchStride;
} mfa_launch_params;
#define MFA_MASK_BLOCK_ROWS 256
#define MFA_MASK_BLOCK_COLUMNS 128
void mfa_launch_params_init(mfa_launch_params *params);

/* buffers[slot] = device pointer bound at AttentionOperand.bufferBinding `slot`; slots the
 * kernel type does not use may be NULL (forward: 0-4; backwardQuery: 0-6,9;
 * backwardKeyValue: 0-2,4-8; +Source.swift:72-103).  Asynchronous on `stream`. */
mfa_status mfa_attention_kernel_launch(const mfa_attention_kernel *kernel,
                                       void *const buffers[MFA_BUFFER_SLOTS],
                                       const mfa_launch_params *params, void *stream);

/* Bytes of mfa_launch_params.workspace this launch would use (0 if it would not be split). */
mfa_status mfa_attention_kernel_workspace_size(const mfa_attention_kernel *kernel,
                                               const mfa_launch_params *params, uint64_t *bytes);
/* What a launch with these buffers and parameters would run, as text -- nothing is launched.  The code object is a property of
 * the kernel object (mfa_attention_kernel_variant); the FORM is a property of the launch: the general kernel when the launch
 * does not meet the matrix-core kernels' requirements, a re-layout pass in front, column-parallel pieces + combine through
 * the workspace, or (dense and causal forward launches at D <= 128 without per-batch lengths) the persistent form
 * `attn_fwd16_p4p`, which is the name rocprofv3 shows for such launches.  FP32 descriptors (head blocks 64 and 128): the general
 * kernel's variant launches `attn_f32_{fwd,dq,dkv}_d{64,128}_w4x32` when every operand of the launch is FP32, row-major with
 * 16-byte aligned rows (pointer, leading dimension, head / batch strides multiples of 4 elements), D % 4 == 0 and there is no
 * block mask; the general kernel itself otherwise.  `buffers` as for mfa_attention_kernel_launch. */
mfa_status mfa_attention_kernel_launch_form(const mfa_attention_kernel *kernel, void *const buffers[MFA_BUFFER_SLOTS],
                                            const mfa_launch_params *params, char *out, size_t capacity);

/* Timing helper: `warmup` untimed launches, then `iterations` back-to-back launches bracketed by
 * HIP events recorded on `stream` (the harness of SquareAttentionTest.swift:733-761 uses
 * gpuEndTime-gpuStartTime around 5 dispatches).  *milliseconds = total for all iterations. */
mfa_status mfa_attention_kernel_time(const mfa_attention_kernel *kernel,
                                     void *const buffers[MFA_BUFFER_SLOTS],
                                     const mfa_launch_params *params, void *stream,
                                     int warmup, int iterations, float *milliseconds);

/* ---- device context (replaces MTLContext.global, Utilities/MTLContext.swift:10-20) -------- */
mfa_status mfa_device_count(int *count);
mfa_status mfa_device_name(int device, char *out, size_t capacity); /* gcnArchName, e.g. gfx950 */

/* thread-local description of the last non-OK status returned on this thread */
const char *mfa_last_error_string(void);
int mfa_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MFA_H */
