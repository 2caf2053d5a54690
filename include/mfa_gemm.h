/*
 * mfa_gemm.h -- C ABI of the GEMM operator: the sibling operator of the
 * attention path in philipturner/metal-flash-attention (Sources/FlashAttention/GEMM/).  Same rules as
 * mfa.h: plain pointers and sizes, caller-owned device memory, status codes instead of fatalError,
 * asynchronous launches on the caller's HIP stream.
 *
 *   C[m][n] = sum_k A(m, k) B(k, n)  (+ previous C[m][n] when loadPreviousC)
 *   A is M x K row-major (K x M when transposed), B is K x N (N x K when transposed), C is M x N.
 *
 * Reference interface each entry replaces:
 *   mfa_gemm_descriptor                       GEMMDescriptor            GEMM/GEMMDescriptor/GEMMDescriptor.swift:11-47
 *   mfa_gemm_kernel_descriptor                GEMMKernelDescriptor      GEMM/GEMMKernelDescriptor.swift (struct fields)
 *   mfa_gemm_descriptor_kernel_descriptor     GEMMKernelDescriptor.init(descriptor:)   GEMMDescriptor.swift:98-322
 *   mfa_gemm_kernel_create / _destroy         GEMMKernel.init(descriptor:) + makeLibrary + makeComputePipelineState
 *                                             (GEMMKernel/GEMMKernel.swift, GEMMDescriptor+PipelineCache.swift:16-125)
 *   mfa_gemm_kernel_block_dimensions, _threadgroup_size, _threadgroup_memory_allocation
 *                                             GEMMKernel.blockDimensions / .threadgroupSize / .threadgroupMemoryAllocation
 *   mfa_gemm_kernel_launch                    setFunctionConstants (GEMMDescriptor.swift:325-381: M, N, K, leading
 *                                             dimensions, loadPreviousC) + setBuffer x3 + dispatchThreadgroups
 *                                             (Tests/FlashAttentionTests/GEMM/LaplacianTest.swift:186-215)
 */
#ifndef MFA_GEMM_H
#define MFA_GEMM_H

#include "mfa.h"

#ifdef __cplusplus
extern "C" {
#endif

/* GEMMDescriptor (GEMMDescriptor.swift:11-47).  has* = 0 mirrors a nil optional. */
typedef struct mfa_gemm_descriptor {
  uint32_t batchDimension;          /* default 1 */
  uint8_t hasLeadingDimensions;
  uint8_t loadPreviousC;            /* default 0 */
  uint8_t hasMatrixDimensions;
  uint8_t hasMemoryPrecisions;
  uint8_t hasTransposeState;
  uint8_t transposeA, transposeB;
  uint8_t reserved;
  uint32_t leadingDimensionA, leadingDimensionB, leadingDimensionC;   /* elements */
  uint32_t M, N, K;
  int32_t precisionA, precisionB, precisionC;                          /* mfa_precision */
} mfa_gemm_descriptor;

/* GEMMKernelDescriptor (GEMMKernelDescriptor.swift).  The block dimensions are a request: the library picks
 * among pre-compiled code objects; mfa_gemm_kernel_block_dimensions reports what the object really uses. */
typedef struct mfa_gemm_kernel_descriptor {
  uint16_t blockM, blockN, blockK;                  /* 0 = unset */
  uint16_t leadingBlockA, leadingBlockB, leadingBlockC;   /* 0 = unset (LDS row pitch, elements) */
  int32_t memoryPrecisionA, memoryPrecisionB, memoryPrecisionC;
  int32_t registerPrecisionA, registerPrecisionB, registerPrecisionC;
  uint16_t splitsM, splitsN;                        /* waves per workgroup along M and N */
  uint8_t preferAsyncLoad, preferAsyncStore;
  uint8_t transposeA, transposeB;
  uint8_t complete;                                 /* set by mfa_gemm_descriptor_kernel_descriptor */
  uint8_t reserved[3];
} mfa_gemm_kernel_descriptor;

/* Launch-time values: the reference's function constants (GEMMDescriptor.swift:325-381) plus a batch
 * extension (the reference leaves batching to clients, GEMMDescriptor.swift:12-18). */
typedef struct mfa_gemm_launch_params {
  uint32_t M, N, K;
  uint32_t leadingDimensionA, leadingDimensionB, leadingDimensionC;   /* 0 = tightly packed */
  uint32_t loadPreviousC;
  uint32_t batchDimension;                                             /* 0 or 1 = single problem */
  uint64_t batchStrideA, batchStrideB, batchStrideC;                   /* elements */
} mfa_gemm_launch_params;

typedef struct mfa_gemm_kernel mfa_gemm_kernel;

void mfa_gemm_descriptor_init(mfa_gemm_descriptor *descriptor);
void mfa_gemm_launch_params_init(mfa_gemm_launch_params *params);

/* GEMMKernelDescriptor.init(descriptor:): register precisions (GEMMDescriptor.swift:185-205), block
 * dimensions re-derived for gfx950 (the reference sizes blocks by Apple core count, :240-322).
 * MFA_ERR_INCOMPLETE_DESCRIPTOR where the reference aborts with "Descriptor was incomplete." */
mfa_status mfa_gemm_descriptor_kernel_descriptor(const mfa_gemm_descriptor *descriptor,
                                                 mfa_gemm_kernel_descriptor *out);

mfa_status mfa_gemm_kernel_create(const mfa_gemm_kernel_descriptor *descriptor, mfa_gemm_kernel **out);
void mfa_gemm_kernel_destroy(mfa_gemm_kernel *kernel);
mfa_status mfa_gemm_kernel_block_dimensions(const mfa_gemm_kernel *kernel, uint16_t *M, uint16_t *N, uint16_t *K);
uint32_t mfa_gemm_kernel_threadgroup_size(const mfa_gemm_kernel *kernel);
uint32_t mfa_gemm_kernel_threadgroup_memory_allocation(const mfa_gemm_kernel *kernel);
const char *mfa_gemm_kernel_variant(const mfa_gemm_kernel *kernel);

/* A, B, C: device pointers (buffer indices 0, 1, 2 of the reference's encoder).  MFA_ERR_INVALID_ARGUMENT
 * if a leading dimension is smaller than the row it must hold ("Leading block dimension was too small."). */
mfa_status mfa_gemm_kernel_launch(const mfa_gemm_kernel *kernel, const void *A, const void *B, void *C,
                                  const mfa_gemm_launch_params *params, void *stream);
/* `iterations` back-to-back launches between two HIP events on `stream` (LaplacianTest.swift:181-228) */
mfa_status mfa_gemm_kernel_time(const mfa_gemm_kernel *kernel, const void *A, const void *B, void *C,
                                const mfa_gemm_launch_params *params, void *stream, int warmup, int iterations,
                                float *milliseconds);

#ifdef __cplusplus
}
#endif
#endif /* MFA_GEMM_H */
