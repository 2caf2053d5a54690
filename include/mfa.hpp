// mfa.hpp -- C++17 host-side mirror of the reference's Swift operator interface, over the C ABI of mfa.h and
// mfa_gemm.h (header-only; link with -lmfa_hip).  Same type names, property names and argument meaning as the
// reference so that a test written against it reads like the reference's own
// (Tests/FlashAttentionTests/Attention/SquareAttentionTest.swift:214-380, GEMM/LaplacianTest.swift:25-41):
//
//   mfa::AttentionDescriptor attentionDesc;
//   attentionDesc.lowPrecisionInputs = false;
//   attentionDesc.lowPrecisionIntermediates = false;
//   attentionDesc.matrixDimensions = {{row, column, head}};
//   attentionDesc.transposeState = {{false, false, false, false}};
//   auto kernelDesc = attentionDesc.kernelDescriptor(mfa::AttentionKernelType::forward);
//   mfa::AttentionKernel kernel(kernelDesc);
//   kernel.blockDimensions, kernel.threadgroupSize, kernel.threadgroupMemoryAllocation
//   kernel.dispatch(buffers, params, stream);   // createSource + makeLibrary + pipeline + setBuffer x10 + dispatch
//
// Mirrored types (reference file:line):
//   AttentionDescriptor        Sources/FlashAttention/Attention/AttentionDescriptor/AttentionDescriptor.swift:10-148
//   AttentionKernelDescriptor  Sources/FlashAttention/Attention/AttentionKernelDescriptor.swift:8-49
//   AttentionKernelType        Sources/FlashAttention/Attention/AttentionKernelType.swift:10-23
//   AttentionOperand           Sources/FlashAttention/Attention/AttentionOperand.swift:9-71
//   AttentionKernel            Sources/FlashAttention/Attention/AttentionKernel/AttentionKernel.swift:10-51
//   GEMMOperandPrecision       Sources/FlashAttention/GEMM/GEMMOperandPrecision.swift:33-60
//   GEMMDescriptor, GEMMKernelDescriptor, GEMMKernel   Sources/FlashAttention/GEMM/...
// Where the reference calls fatalError this throws mfa::Error (status code + mfa_last_error_string()).
#pragma once
#include <array>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>

#include "mfa.h"
#include "mfa_gemm.h"

namespace mfa {

struct Error : std::runtime_error {
  mfa_status status;
  Error(mfa_status s, const char *message) : std::runtime_error(message), status(s) {}
};
inline void check(mfa_status s) {
  if (s != MFA_OK) throw Error(s, mfa_last_error_string());
}

enum class GEMMOperandPrecision : int { FP32 = MFA_FP32, FP16 = MFA_FP16, BF16 = MFA_BF16 };
inline const char *name(GEMMOperandPrecision p) { return mfa_precision_name((int)p); }   // GEMMOperandPrecision.swift:39-48
inline int size(GEMMOperandPrecision p) { return mfa_precision_size((int)p); }          // :51-59

enum class AttentionKernelType : int { forward = MFA_FORWARD, backwardQuery = MFA_BACKWARD_QUERY, backwardKeyValue = MFA_BACKWARD_KEY_VALUE };

enum class AttentionOperand : int {
  Q = MFA_Q, K = MFA_K, S = MFA_S, P = MFA_P, V = MFA_V, O = MFA_O, L = MFA_L, D = MFA_D,
  dO = MFA_dO, dV = MFA_dV, dP = MFA_dP, dS = MFA_dS, dK = MFA_dK, dQ = MFA_dQ
};
inline const char *description(AttentionOperand op) { return mfa_operand_name((int)op); }
inline std::optional<int> bufferBinding(AttentionOperand op) {   // AttentionOperand.swift:52-71
  const int b = mfa_operand_buffer_binding((int)op);
  return b < 0 ? std::nullopt : std::optional<int>(b);
}

struct AttentionKernelDescriptor {
  mfa_attention_kernel_descriptor c;
  AttentionKernelDescriptor() { mfa_attention_kernel_descriptor_init(&c); }
};

struct AttentionDescriptor {
  bool lowPrecisionInputs = false;
  bool lowPrecisionIntermediates = false;
  struct Dimensions { uint32_t row, column; uint16_t head; };
  std::optional<Dimensions> matrixDimensions;
  struct Transposes { bool Q, K, V, O; };
  std::optional<Transposes> transposeState;
  // extensions (include/mfa.h): storage type of low-precision inputs, fused 16-bit output cast
  GEMMOperandPrecision lowPrecisionInputType = GEMMOperandPrecision::FP16;
  bool lowPrecisionOutputs = false;

  mfa_attention_descriptor cDescriptor() const {
    mfa_attention_descriptor d;
    mfa_attention_descriptor_init(&d);
    d.lowPrecisionInputs = lowPrecisionInputs;
    d.lowPrecisionIntermediates = lowPrecisionIntermediates;
    d.lowPrecisionInputType = (uint8_t)lowPrecisionInputType;
    d.lowPrecisionOutputs = lowPrecisionOutputs;
    if (matrixDimensions) {
      d.hasMatrixDimensions = 1;
      d.row = matrixDimensions->row;
      d.column = matrixDimensions->column;
      d.head = matrixDimensions->head;
    }
    if (transposeState) {
      d.hasTransposeState = 1;
      d.transposeQ = transposeState->Q;
      d.transposeK = transposeState->K;
      d.transposeV = transposeState->V;
      d.transposeO = transposeState->O;
    }
    return d;
  }
  // AttentionDescriptor.kernelDescriptor(type:), AttentionDescriptor.swift:33-130
  AttentionKernelDescriptor kernelDescriptor(AttentionKernelType type) const {
    const mfa_attention_descriptor d = cDescriptor();
    AttentionKernelDescriptor out;
    check(mfa_attention_descriptor_kernel_descriptor(&d, (int)type, &out.c));
    return out;
  }
  // .memoryPrecisions / .registerPrecisions (+Precisions.swift:10-215), indexed by AttentionOperand
  std::array<int8_t, MFA_OPERAND_COUNT> memoryPrecisions() const {
    const mfa_attention_descriptor d = cDescriptor();
    std::array<int8_t, MFA_OPERAND_COUNT> out{};
    check(mfa_attention_descriptor_memory_precisions(&d, out.data()));
    return out;
  }
  std::array<int8_t, MFA_OPERAND_COUNT> registerPrecisions() const {
    const mfa_attention_descriptor d = cDescriptor();
    std::array<int8_t, MFA_OPERAND_COUNT> out{};
    check(mfa_attention_descriptor_register_precisions(&d, out.data()));
    return out;
  }
};

class AttentionKernel {
 public:
  struct BlockDimensions { uint16_t parallelization, traversal, head; };
  BlockDimensions blockDimensions{};
  uint32_t threadgroupSize = 0;
  uint32_t threadgroupMemoryAllocation = 0;
  std::string variant;

  explicit AttentionKernel(const AttentionKernelDescriptor &descriptor) {   // AttentionKernel.swift:27-50
    check(mfa_attention_kernel_create(&descriptor.c, &handle_));
    mfa_attention_kernel_block_dimensions(handle_, &blockDimensions.parallelization, &blockDimensions.traversal, &blockDimensions.head);
    threadgroupSize = mfa_attention_kernel_threadgroup_size(handle_);
    threadgroupMemoryAllocation = mfa_attention_kernel_threadgroup_memory_allocation(handle_);
    variant = mfa_attention_kernel_variant(handle_);
  }
  ~AttentionKernel() { mfa_attention_kernel_destroy(handle_); }
  AttentionKernel(const AttentionKernel &) = delete;
  AttentionKernel &operator=(const AttentionKernel &) = delete;
  AttentionKernel(AttentionKernel &&other) noexcept { *this = std::move(other); }
  AttentionKernel &operator=(AttentionKernel &&other) noexcept {
    std::swap(handle_, other.handle_);
    blockDimensions = other.blockDimensions;
    threadgroupSize = other.threadgroupSize;
    threadgroupMemoryAllocation = other.threadgroupMemoryAllocation;
    variant = std::move(other.variant);
    return *this;
  }

  // buffers[i] = device pointer bound at bufferBinding i (Q0 K1 V2 O3 L4 D5 dO6 dV7 dK8 dQ9); asynchronous on `stream`
  void dispatch(void *const buffers[MFA_BUFFER_SLOTS], const mfa_launch_params &params, void *stream = nullptr) const {
    check(mfa_attention_kernel_launch(handle_, buffers, &params, stream));
  }
  void dispatch(void *const buffers[MFA_BUFFER_SLOTS], uint32_t row, uint32_t column, void *stream = nullptr) const {
    mfa_launch_params p;
    mfa_launch_params_init(&p);
    p.row = row;
    p.column = column;
    dispatch(buffers, p, stream);
  }
  uint64_t workspaceSize(const mfa_launch_params &params) const {
    uint64_t bytes = 0;
    check(mfa_attention_kernel_workspace_size(handle_, &params, &bytes));
    return bytes;
  }
  float time(void *const buffers[MFA_BUFFER_SLOTS], const mfa_launch_params &params, void *stream, int warmup, int iterations) const {
    float ms = 0;
    check(mfa_attention_kernel_time(handle_, buffers, &params, stream, warmup, iterations, &ms));
    return ms;
  }

 private:
  mfa_attention_kernel *handle_ = nullptr;
};

// ---- GEMM operator ---------------------------------------------------------------------------------------------
struct GEMMKernelDescriptor {
  mfa_gemm_kernel_descriptor c{};
};

struct GEMMDescriptor {   // GEMMDescriptor.swift:11-47
  int batchDimension = 1;
  struct Leading { uint32_t A, B, C; };
  std::optional<Leading> leadingDimensions;
  bool loadPreviousC = false;
  struct Dimensions { uint32_t M, N, K; };
  std::optional<Dimensions> matrixDimensions;
  struct Precisions { GEMMOperandPrecision A, B, C; };
  std::optional<Precisions> memoryPrecisions;
  struct Transposes { bool A, B; };
  std::optional<Transposes> transposeState;

  mfa_gemm_descriptor cDescriptor() const {
    mfa_gemm_descriptor d;
    mfa_gemm_descriptor_init(&d);
    d.batchDimension = (uint32_t)batchDimension;
    d.loadPreviousC = loadPreviousC;
    if (leadingDimensions) { d.hasLeadingDimensions = 1; d.leadingDimensionA = leadingDimensions->A; d.leadingDimensionB = leadingDimensions->B; d.leadingDimensionC = leadingDimensions->C; }
    if (matrixDimensions) { d.hasMatrixDimensions = 1; d.M = matrixDimensions->M; d.N = matrixDimensions->N; d.K = matrixDimensions->K; }
    if (memoryPrecisions) { d.hasMemoryPrecisions = 1; d.precisionA = (int)memoryPrecisions->A; d.precisionB = (int)memoryPrecisions->B; d.precisionC = (int)memoryPrecisions->C; }
    if (transposeState) { d.hasTransposeState = 1; d.transposeA = transposeState->A; d.transposeB = transposeState->B; }
    return d;
  }
  GEMMKernelDescriptor kernelDescriptor() const {   // GEMMKernelDescriptor(descriptor:), GEMMDescriptor.swift:98-322
    const mfa_gemm_descriptor d = cDescriptor();
    GEMMKernelDescriptor out;
    check(mfa_gemm_descriptor_kernel_descriptor(&d, &out.c));
    return out;
  }
  mfa_gemm_launch_params launchParams() const {      // setFunctionConstants, GEMMDescriptor.swift:325-381
    if (!matrixDimensions) throw Error(MFA_ERR_INCOMPLETE_DESCRIPTOR, "Descriptor was incomplete.");
    mfa_gemm_launch_params p;
    mfa_gemm_launch_params_init(&p);
    p.M = matrixDimensions->M; p.N = matrixDimensions->N; p.K = matrixDimensions->K;
    if (leadingDimensions) { p.leadingDimensionA = leadingDimensions->A; p.leadingDimensionB = leadingDimensions->B; p.leadingDimensionC = leadingDimensions->C; }
    p.loadPreviousC = loadPreviousC;
    p.batchDimension = (uint32_t)batchDimension;
    return p;
  }
};

class GEMMKernel {
 public:
  struct BlockDimensions { uint16_t M, N, K; };
  BlockDimensions blockDimensions{};
  uint32_t threadgroupSize = 0;
  uint32_t threadgroupMemoryAllocation = 0;
  std::string variant;

  explicit GEMMKernel(const GEMMKernelDescriptor &descriptor) {
    check(mfa_gemm_kernel_create(&descriptor.c, &handle_));
    mfa_gemm_kernel_block_dimensions(handle_, &blockDimensions.M, &blockDimensions.N, &blockDimensions.K);
    threadgroupSize = mfa_gemm_kernel_threadgroup_size(handle_);
    threadgroupMemoryAllocation = mfa_gemm_kernel_threadgroup_memory_allocation(handle_);
    variant = mfa_gemm_kernel_variant(handle_);
  }
  ~GEMMKernel() { mfa_gemm_kernel_destroy(handle_); }
  GEMMKernel(const GEMMKernel &) = delete;
  GEMMKernel &operator=(const GEMMKernel &) = delete;

  void dispatch(const void *A, const void *B, void *C, const GEMMDescriptor &descriptor, void *stream = nullptr) const {
    const mfa_gemm_launch_params p = descriptor.launchParams();
    check(mfa_gemm_kernel_launch(handle_, A, B, C, &p, stream));
  }

 private:
  mfa_gemm_kernel *handle_ = nullptr;
};

}  // namespace mfa
