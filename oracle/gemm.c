/*
 * oracle/gemm.c -- CPU restatement of the reference's GEMM checks.  TEST INFRASTRUCTURE ONLY: nothing under
 * metal_flash_attention_amd/ may call this; only tests/ and bench tooling do, as the checker.
 *
 * Two independent oracles, both taken from the reference's own tests:
 *  1. oracle_gemm_naive: the triple loop of Tests/FlashAttentionTests/GEMM/AdversarialShapeTest.swift:205-243
 *     (fp32, sequential accumulation over k, optional previous C added after the dot product), with the
 *     addressing of :214-228 (transposes, leading dimensions).
 *  2. oracle_laplacian_expected: the CLOSED-FORM answer of Tests/FlashAttentionTests/GEMM/LaplacianTest.swift:
 *     A = 2nd-order periodic Laplacian (:137-149), so (A B)[m][n] = B[m-1][n] - 2 B[m][n] + B[m+1][n]
 *     (:286-318, including the A^T / B^T role swaps of :166-175 and :299-311).  This is a known answer the
 *     reference pins itself to, so for the GEMM operator parity is PINNED: the kernel is checked against
 *     the same closed form, not only against a restated loop.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* AdversarialShapeTest.swift:205-243.  A, B, previousC, C: float arrays with the given leading dimensions. */
void oracle_gemm_naive(uint32_t M, uint32_t N, uint32_t K, const float *A, const float *B, const float *previousC,
                       float *C, uint32_t ldA, uint32_t ldB, uint32_t ldC, int transA, int transB, int loadPreviousC) {
#pragma omp parallel for schedule(static)
  for (uint32_t m = 0; m < M; ++m)
    for (uint32_t n = 0; n < N; ++n) {
      float dot = 0.f;
      for (uint32_t k = 0; k < K; ++k) {
        const size_t a = transA ? (size_t)k * ldA + m : (size_t)m * ldA + k;   /* :214-219 */
        const size_t b = transB ? (size_t)n * ldB + k : (size_t)k * ldB + n;   /* :220-224 */
        dot += A[a] * B[b];
      }
      const size_t c = (size_t)m * ldC + n;
      if (loadPreviousC) dot += previousC[c];                                  /* :232-235 */
      C[c] = dot;
    }
}

/* same in double precision (validation of the restatement itself) */
void oracle_gemm_f64(uint32_t M, uint32_t N, uint32_t K, const float *A, const float *B, const float *previousC,
                     double *C, uint32_t ldA, uint32_t ldB, uint32_t ldC, int transA, int transB, int loadPreviousC) {
#pragma omp parallel for schedule(static)
  for (uint32_t m = 0; m < M; ++m)
    for (uint32_t n = 0; n < N; ++n) {
      double dot = 0.0;
      for (uint32_t k = 0; k < K; ++k) {
        const size_t a = transA ? (size_t)k * ldA + m : (size_t)m * ldA + k;
        const size_t b = transB ? (size_t)n * ldB + k : (size_t)k * ldB + n;
        dot += (double)A[a] * (double)B[b];
      }
      const size_t c = (size_t)m * ldC + n;
      if (loadPreviousC) dot += (double)previousC[c];
      C[c] = dot;
    }
}

/* LaplacianTest.swift:137-149: the n x n 2nd-order periodic Laplacian, row-major */
void oracle_laplacian_matrix(uint32_t n, float *A) {
  for (size_t i = 0; i < (size_t)n * n; ++i) A[i] = 0.f;
  for (uint32_t d = 0; d < n; ++d) {
    A[(size_t)d * n + d] = -2.f;
    A[(size_t)d * n + (d + n - 1) % n] = 1.f;   /* order of the three writes as in the reference: for n <= 2 */
    A[(size_t)d * n + (d + n + 1) % n] = 1.f;   /* later writes overwrite earlier ones */
  }
}

/* LaplacianTest.swift:286-332: expected value and where the test reads the actual value.
 * `source` is the random operand: B when A is not transposed, otherwise the array the test passes as A after
 * its swap (:166-175).  Returns expected; *actual_index = index into C the test compares with. */
float oracle_laplacian_expected(uint32_t n, const float *source, const float *previousC, uint32_t m, uint32_t col,
                                int transA, int transB, int loadPreviousC, size_t *actual_index) {
  const uint32_t left = (m + n - 1) % n, center = m, right = (m + n + 1) % n;
  float l, c, r;
  if (transA) {                       /* :299-302 */
    l = source[(size_t)left * n + col]; c = source[(size_t)center * n + col]; r = source[(size_t)right * n + col];
  } else if (transB) {                /* :303-306 */
    l = source[(size_t)col * n + left]; c = source[(size_t)col * n + center]; r = source[(size_t)col * n + right];
  } else {                            /* :307-310 */
    l = source[(size_t)left * n + col]; c = source[(size_t)center * n + col]; r = source[(size_t)right * n + col];
  }
  float expected = l - 2 * c + r;     /* :314 */
  if (loadPreviousC) expected += transA ? previousC[(size_t)col * n + m] : previousC[(size_t)m * n + col];   /* :315-321 */
  *actual_index = transA ? (size_t)col * n + m : (size_t)m * n + col;                                         /* :325-329 */
  return expected;
}
