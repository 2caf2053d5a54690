// tests/c_abi/cpp_mirror_check.cpp -- the reference's validateProblemSize flow
// (Tests/FlashAttentionTests/Attention/SquareAttentionTest.swift:214-555) and one LaplacianTest case
// (Tests/FlashAttentionTests/GEMM/LaplacianTest.swift:118-332) written against include/mfa.hpp, the C++ mirror of
// the Swift types.  The oracle is linked as the checker (test infrastructure).  Exit code 0 = all within tolerance.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mfa.hpp"

extern "C" {
void oracle_network_init(unsigned long long seed, int R, int C, int D, float *Q, float *K, float *V, float *dO);
int oracle_network_run(int R, int C, int D, const float *Q, const float *K, const float *V, const float *dO, float *O,
                       float *L, float *Dt, float *dV, float *dK, float *dQ, int num_threads);
void oracle_laplacian_matrix(uint32_t n, float *A);
float oracle_laplacian_expected(uint32_t n, const float *source, const float *previousC, uint32_t m, uint32_t col, int transA,
                                int transB, int loadPreviousC, size_t *actual_index);
}

#define HIP(call)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 3; } \
  } while (0)

static double max_err(const std::vector<float> &a, const std::vector<float> &b, double scale_b) {
  double m = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    const double d = std::fabs((double)a[i] - (double)b[i] * scale_b);
    if (d > m || d != d) m = d;
  }
  return m;
}

int main(int argc, char **argv) {
  const int R = argc > 1 ? std::atoi(argv[1]) : 93, C = argc > 2 ? std::atoi(argv[2]) : 130, D = argc > 3 ? std::atoi(argv[3]) : 32;
  try {
    // ---- attention: SquareAttentionTest.swift:235-263
    mfa::AttentionDescriptor attentionDesc;
    attentionDesc.lowPrecisionInputs = false;
    attentionDesc.lowPrecisionIntermediates = false;
    attentionDesc.matrixDimensions = mfa::AttentionDescriptor::Dimensions{(uint32_t)R, (uint32_t)C, (uint16_t)D};
    attentionDesc.transposeState = mfa::AttentionDescriptor::Transposes{false, false, false, false};
    mfa::AttentionKernel kernelForward(attentionDesc.kernelDescriptor(mfa::AttentionKernelType::forward));
    mfa::AttentionKernel kernelBackwardQuery(attentionDesc.kernelDescriptor(mfa::AttentionKernelType::backwardQuery));
    mfa::AttentionKernel kernelBackwardKeyValue(attentionDesc.kernelDescriptor(mfa::AttentionKernelType::backwardKeyValue));
    std::printf("forward: %s  block (%u, %u, %u)\n", kernelForward.variant.c_str(), kernelForward.blockDimensions.parallelization,
                kernelForward.blockDimensions.traversal, kernelForward.blockDimensions.head);

    const size_t nq = (size_t)R * D, nk = (size_t)C * D;
    std::vector<float> Q(nq), K(nk), V(nk), dO(nq), O(nq), L(R), Dt(R), dV(nk), dK(nk), dQ(nq);
    oracle_network_init(7, R, C, D, Q.data(), K.data(), V.data(), dO.data());
    if (oracle_network_run(R, C, D, Q.data(), K.data(), V.data(), dO.data(), O.data(), L.data(), Dt.data(), dV.data(), dK.data(),
                           dQ.data(), 0) != 0)
      return 4;
    const size_t bytes[MFA_BUFFER_SLOTS] = {nq * 4, nk * 4, nk * 4, nq * 4, (size_t)R * 4, (size_t)R * 4, nq * 4, nk * 4, nk * 4, nq * 4};
    const float *init[MFA_BUFFER_SLOTS] = {Q.data(), K.data(), V.data(), nullptr, nullptr, nullptr, dO.data(), nullptr, nullptr, nullptr};
    void *buf[MFA_BUFFER_SLOTS];
    for (int s = 0; s < MFA_BUFFER_SLOTS; ++s) {
      HIP(hipMalloc(&buf[s], bytes[s]));
      if (init[s]) HIP(hipMemcpy(buf[s], init[s], bytes[s], hipMemcpyHostToDevice));
      else HIP(hipMemset(buf[s], 0xFF, bytes[s]));
    }
    kernelForward.dispatch(buf, R, C);            // SquareAttentionTest.swift:355-368
    kernelBackwardQuery.dispatch(buf, R, C);
    kernelBackwardKeyValue.dispatch(buf, R, C);
    HIP(hipDeviceSynchronize());
    std::vector<float> gO(nq), gL(R), gD(R), gdV(nk), gdK(nk), gdQ(nq);
    HIP(hipMemcpy(gO.data(), buf[3], nq * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(gL.data(), buf[4], R * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(gD.data(), buf[5], R * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(gdV.data(), buf[7], nk * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(gdK.data(), buf[8], nk * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(gdQ.data(), buf[9], nq * 4, hipMemcpyDeviceToHost));
    const double e[6] = {max_err(gO, O, 1.0), max_err(gL, L, 1.44269504089), max_err(gD, Dt, 1.0 / std::sqrt((double)D)),
                         max_err(gdV, dV, 1.0), max_err(gdK, dK, 1.0), max_err(gdQ, dQ, 1.0)};
    const char *names[6] = {"O", "L", "D", "dV", "dK", "dQ"};
    bool bad = false;
    for (int i = 0; i < 6; ++i) {
      std::printf("%-3s max |err| = %.3e\n", names[i], e[i]);
      if (!(e[i] <= 2e-5 * (i == 1 ? 1.44269504089 : 1.0))) bad = true;   // SquareAttentionTest.swift:539-554
    }

    // ---- GEMM: LaplacianTest.swift:25-41, known answer :286-332
    const uint32_t n = 49;
    mfa::GEMMDescriptor gemmDesc;
    gemmDesc.loadPreviousC = false;
    gemmDesc.matrixDimensions = mfa::GEMMDescriptor::Dimensions{n, n, n};
    gemmDesc.memoryPrecisions = mfa::GEMMDescriptor::Precisions{mfa::GEMMOperandPrecision::FP32, mfa::GEMMOperandPrecision::FP32,
                                                               mfa::GEMMOperandPrecision::FP32};
    gemmDesc.transposeState = mfa::GEMMDescriptor::Transposes{false, false};
    mfa::GEMMKernel gemm(gemmDesc.kernelDescriptor());
    std::vector<float> A(n * n), B(n * n), Cm(n * n);
    oracle_laplacian_matrix(n, A.data());
    for (size_t i = 0; i < B.size(); ++i) B[i] = (float)((i * 2654435761u) % 1000) / 1000.0f;
    void *dA, *dB, *dC;
    HIP(hipMalloc(&dA, n * n * 4)); HIP(hipMalloc(&dB, n * n * 4)); HIP(hipMalloc(&dC, n * n * 4));
    HIP(hipMemcpy(dA, A.data(), n * n * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(dB, B.data(), n * n * 4, hipMemcpyHostToDevice));
    gemm.dispatch(dA, dB, dC, gemmDesc);
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(Cm.data(), dC, n * n * 4, hipMemcpyDeviceToHost));
    double ge = 0;
    for (uint32_t m = 0; m < n; ++m)
      for (uint32_t c = 0; c < n; ++c) {
        size_t idx;
        const float want = oracle_laplacian_expected(n, B.data(), B.data(), m, c, 0, 0, 0, &idx);
        ge = std::fmax(ge, std::fabs((double)Cm[idx] - want));
      }
    std::printf("GEMM %s laplacian n=%u max |err| = %.3e\n", gemm.variant.c_str(), n, ge);
    if (!(ge <= 1e-5)) bad = true;                                            // LaplacianTest.swift:264-283

    // fatalError -> exception
    bool threw = false;
    try { mfa::AttentionDescriptor().kernelDescriptor(mfa::AttentionKernelType::forward); } catch (const mfa::Error &err) { threw = err.status == MFA_ERR_INCOMPLETE_DESCRIPTOR; }
    if (!threw) bad = true;
    std::printf(bad ? "FAILED\n" : "C++ MIRROR OK (%d x %d x %d)\n", R, C, D);
    return bad ? 1 : 0;
  } catch (const mfa::Error &err) {
    std::fprintf(stderr, "mfa::Error %d: %s\n", (int)err.status, err.what());
    return 2;
  }
}
