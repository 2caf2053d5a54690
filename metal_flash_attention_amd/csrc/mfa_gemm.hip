// mfa_gemm.hip -- host side of the GEMM operator (include/mfa_gemm.h): descriptor -> kernel descriptor,
// code-object selection, launch.  Reference: Sources/FlashAttention/GEMM/GEMMDescriptor/GEMMDescriptor.swift,
// GEMMKernelDescriptor.swift, GEMMKernel/GEMMKernel.swift.
#include "../../include/mfa_gemm.h"
#include "gemm_kernels.h"
#include "mfa_internal.h"

#include <cstdlib>
#include <cstring>
#include <string>

using namespace mfa;

struct mfa_gemm_kernel {
  mfa_gemm_kernel_descriptor desc;
  bool fast16 = false;          // A and B in one 16-bit type: gemm_16 when the launch is 16-byte aligned
  bool big = false;             // 256 x 256 block (8 waves) instead of 128 x 128 (4 waves)
  std::string name;
};

// number of compute units the block-size heuristic fills (MI355X: 256; queried once)
static int compute_units() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}

static bool valid_precision(int p) { return p == MFA_FP32 || p == MFA_FP16 || p == MFA_BF16; }

extern "C" void mfa_gemm_descriptor_init(mfa_gemm_descriptor *d) {
  if (!d) return;
  std::memset(d, 0, sizeof(*d));
  d->batchDimension = 1;
}

extern "C" void mfa_gemm_launch_params_init(mfa_gemm_launch_params *p) {
  if (!p) return;
  std::memset(p, 0, sizeof(*p));
  p->batchDimension = 1;
}

// GEMMKernelDescriptor.init(descriptor:) (GEMMDescriptor.swift:98-322), with the device-dependent parts
// re-derived for gfx950: one block shape per arithmetic (the reference picks 32x32 / 48x48 blocks by Apple
// core count), 2 x 2 waves, operands multiplied in their own 16-bit type only when A and B share it
// (one MFMA needs both operands in one type), otherwise promoted to FP32 as the reference does for BF16 on
// pre-Apple9 GPUs (:195-202).
extern "C" mfa_status mfa_gemm_descriptor_kernel_descriptor(const mfa_gemm_descriptor *d, mfa_gemm_kernel_descriptor *out) {
  if (!d || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "null pointer");
  if (!d->hasMatrixDimensions || !d->hasMemoryPrecisions || !d->hasTransposeState)
    return fail(MFA_ERR_INCOMPLETE_DESCRIPTOR, "Descriptor was incomplete.");   // GEMMDescriptor.swift:110
  if (!valid_precision(d->precisionA) || !valid_precision(d->precisionB) || !valid_precision(d->precisionC))
    return fail(MFA_ERR_INVALID_ARGUMENT, "unknown GEMMOperandPrecision");
  std::memset(out, 0, sizeof(*out));
  out->memoryPrecisionA = d->precisionA;
  out->memoryPrecisionB = d->precisionB;
  out->memoryPrecisionC = d->precisionC;
  const bool fast16 = d->precisionA != MFA_FP32 && d->precisionA == d->precisionB;
  out->registerPrecisionA = fast16 ? d->precisionA : MFA_FP32;
  out->registerPrecisionB = fast16 ? d->precisionB : MFA_FP32;
  out->registerPrecisionC = MFA_FP32;
  // block size by how many workgroups the problem yields (the reference: 32 x 32 vs 48 x 48 blocks by
  // "actualGroups <= idealGroups", GEMMDescriptor.swift:262-318): the 256 x 256 block of the 16-bit kernel
  // halves LDS staging per MFMA but needs >= 3/4 of the compute units' worth of workgroups to pay
  const uint64_t bigGroups = (uint64_t)((d->M + 255) / 256) * ((d->N + 255) / 256) * (d->batchDimension ? d->batchDimension : 1);
  const bool big = fast16 && bigGroups * 4 >= (uint64_t)compute_units() * 3;
  out->blockM = big ? 256 : GEMM_BM;
  out->blockN = big ? 256 : GEMM_BN;
  out->blockK = fast16 ? GEMM16_BK : GEMM_F32_BK;
  out->leadingBlockA = fast16 ? GEMM16_BK : GEMM_F32_BKP;
  out->leadingBlockB = fast16 ? GEMM16_BK : GEMM_F32_BKP;
  out->leadingBlockC = 0;              // C never passes through LDS
  out->splitsM = 2;
  out->splitsN = big ? 4 : 2;
  out->preferAsyncLoad = 0;            // no async-copy engine is used: plain global -> register -> LDS staging
  out->preferAsyncStore = 0;
  out->transposeA = d->transposeA ? 1 : 0;
  out->transposeB = d->transposeB ? 1 : 0;
  out->complete = 1;
  return MFA_OK;
}

extern "C" mfa_status mfa_gemm_kernel_create(const mfa_gemm_kernel_descriptor *kd, mfa_gemm_kernel **out) {
  if (!kd || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "null pointer");
  if (!kd->complete) return fail(MFA_ERR_INCOMPLETE_DESCRIPTOR, "Descriptor was incomplete.");
  if (!valid_precision(kd->memoryPrecisionA) || !valid_precision(kd->memoryPrecisionB) || !valid_precision(kd->memoryPrecisionC))
    return fail(MFA_ERR_INVALID_ARGUMENT, "unknown GEMMOperandPrecision");
  auto *k = new mfa_gemm_kernel();
  k->desc = *kd;
  k->fast16 = kd->memoryPrecisionA != MFA_FP32 && kd->memoryPrecisionA == kd->memoryPrecisionB &&
              kd->registerPrecisionA == kd->memoryPrecisionA;
  k->big = k->fast16 && kd->blockM >= 256 && kd->blockN >= 256;
  // the object reports what it really uses
  k->desc.blockM = k->big ? 256 : GEMM_BM;
  k->desc.blockN = k->big ? 256 : GEMM_BN;
  k->desc.blockK = k->fast16 ? GEMM16_BK : GEMM_F32_BK;
  k->desc.splitsM = 2;
  k->desc.splitsN = k->big ? 4 : 2;
  const char *type = kd->memoryPrecisionA == MFA_BF16 ? "bf16" : "f16";
  k->name = !k->fast16 ? "gemm_f32mfma_128x128x16_w2x2"
                       : std::string("gemm_16_") + type + (k->big ? "_256x256x64_w2x4" : "_128x128x64_w2x2");
  *out = k;
  return MFA_OK;
}

extern "C" void mfa_gemm_kernel_destroy(mfa_gemm_kernel *k) { delete k; }

extern "C" mfa_status mfa_gemm_kernel_block_dimensions(const mfa_gemm_kernel *k, uint16_t *M, uint16_t *N, uint16_t *K) {
  if (!k) return fail(MFA_ERR_INVALID_ARGUMENT, "null kernel");
  if (M) *M = k->desc.blockM;
  if (N) *N = k->desc.blockN;
  if (K) *K = k->desc.blockK;
  return MFA_OK;
}
extern "C" uint32_t mfa_gemm_kernel_threadgroup_size(const mfa_gemm_kernel *k) { return !k ? 0 : (k->big ? 512 : 256); }
extern "C" uint32_t mfa_gemm_kernel_threadgroup_memory_allocation(const mfa_gemm_kernel *k) {
  if (!k) return 0;
  if (!k->fast16) return 2 * 2 * GEMM_BM * GEMM_F32_BKP * 4;
  return k->big ? gemm16_lds_bytes<2, 4, 4, 2>() : gemm16_lds_bytes<2, 2, 2, 2>();
}
extern "C" const char *mfa_gemm_kernel_variant(const mfa_gemm_kernel *k) { return k ? k->name.c_str() : ""; }

static mfa_status prepare(const mfa_gemm_kernel *k, const void *A, const void *B, void *C, const mfa_gemm_launch_params *p,
                          GemmArgs *g, dim3 *grid, bool *use16) {
  if (!k || !p) return fail(MFA_ERR_INVALID_ARGUMENT, "null pointer");
  if (p->M == 0 || p->N == 0) return fail(MFA_ERR_INVALID_ARGUMENT, "matrix dimensions must be positive");
  if (!A || !B || !C) return fail(MFA_ERR_INVALID_ARGUMENT, "null operand buffer");
  const bool tA = k->desc.transposeA, tB = k->desc.transposeB;
  // chooseLeadingDimension (GEMMDescriptor.swift:340-363)
  const uint32_t expA = tA ? p->M : p->K, expB = tB ? p->K : p->N, expC = p->N;
  const uint32_t ldA = p->leadingDimensionA ? p->leadingDimensionA : expA;
  const uint32_t ldB = p->leadingDimensionB ? p->leadingDimensionB : expB;
  const uint32_t ldC = p->leadingDimensionC ? p->leadingDimensionC : expC;
  if (ldA < expA || ldB < expB || ldC < expC) return fail(MFA_ERR_INVALID_ARGUMENT, "Leading block dimension was too small.");
  const uint32_t batch = p->batchDimension ? p->batchDimension : 1;
  *g = GemmArgs{A, B, C, p->M, p->N, p->K, ldA, ldB, ldC, k->desc.memoryPrecisionA, k->desc.memoryPrecisionB,
                k->desc.memoryPrecisionC, tA, tB, p->loadPreviousC ? 1 : 0, p->batchStrideA, p->batchStrideB, p->batchStrideC};
  *grid = dim3((p->N + GEMM_BN - 1) / GEMM_BN, (p->M + GEMM_BM - 1) / GEMM_BM, batch);   // general / small block
  // 32-bit byte offsets must cover the operands (buffer addressing); alignment is not required
  const uint64_t bytesA = (uint64_t)(tA ? p->K : p->M) * ldA * 2, bytesB = (uint64_t)(tB ? p->N : p->K) * ldB * 2;
  *use16 = k->fast16 && bytesA < 0xFFFFFF00ull && bytesB < 0xFFFFFF00ull;
  return MFA_OK;
}

// dynamic LDS above 64 KiB has to be enabled per function (once per process and device is enough: the
// attribute is sticky)
template <typename K> static hipError_t enable_lds(K kernel, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <typename T, bool AKM, bool BKM>
static hipError_t launch16(bool big, const GemmArgs &g, dim3 grid, hipStream_t s) {
  hipError_t e;
  // LDS-DMA staging for launches made of whole, 16-byte aligned chunks (MFA_GEMM_IMPL=vgpr: developer knob that
  // forces the register-staged kernel, which also serves every other launch)
#ifdef MFA_DEV_VARIANTS
  static const bool dmaKnob = !(std::getenv("MFA_GEMM_IMPL") && std::strcmp(std::getenv("MFA_GEMM_IMPL"), "vgpr") == 0);
#else
  constexpr bool dmaKnob = true;
#endif
  const bool aligned = (g.K & 7) == 0 && (g.ldA & 7) == 0 && (g.ldB & 7) == 0 && (g.bsA & 7) == 0 && (g.bsB & 7) == 0 &&
                       ((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0;
  if (big && dmaKnob && aligned) {
    constexpr int LDS = gemm16_lds_bytes<2, 4, 4, 2>();
    e = enable_lds(gemm_16<T, 2, 4, 4, 2, AKM, BKM, true>, LDS);
    if (e == hipSuccess) hipLaunchKernelGGL((gemm_16<T, 2, 4, 4, 2, AKM, BKM, true>), grid, dim3(512), LDS, s, g);
  } else if (big) {
    constexpr int LDS = gemm16_lds_bytes<2, 4, 4, 2>();
    e = enable_lds(gemm_16<T, 2, 4, 4, 2, AKM, BKM>, LDS);
    if (e == hipSuccess) hipLaunchKernelGGL((gemm_16<T, 2, 4, 4, 2, AKM, BKM>), grid, dim3(512), LDS, s, g);
  } else {
    constexpr int LDS = gemm16_lds_bytes<2, 2, 2, 2>();
    e = enable_lds(gemm_16<T, 2, 2, 2, 2, AKM, BKM>, LDS);
    if (e == hipSuccess) hipLaunchKernelGGL((gemm_16<T, 2, 2, 2, 2, AKM, BKM>), grid, dim3(256), LDS, s, g);
  }
  return e;
}

// image kinds of the two operands: k-major rows when the memory order has k contiguous (A not transposed, B transposed)
template <typename T>
static hipError_t launch16_images(bool big, bool akm, bool bkm, const GemmArgs &g, dim3 grid, hipStream_t s) {
  if (akm) return bkm ? launch16<T, true, true>(big, g, grid, s) : launch16<T, true, false>(big, g, grid, s);
  return bkm ? launch16<T, false, true>(big, g, grid, s) : launch16<T, false, false>(big, g, grid, s);
}

static hipError_t launch_one(const mfa_gemm_kernel *k, const GemmArgs &g, dim3 grid, bool use16, hipStream_t s) {
  if (!use16) {
    hipLaunchKernelGGL(gemm_f32mfma, grid, dim3(256), 0, s, g);
    return hipSuccess;
  }
  const dim3 gb((g.N + 255) / 256, (g.M + 255) / 256, grid.z);
  return (g.precA == PREC_BF16) ? launch16_images<__bf16>(k->big, !g.transA, g.transB, g, k->big ? gb : grid, s)
                                : launch16_images<_Float16>(k->big, !g.transA, g.transB, g, k->big ? gb : grid, s);
}

extern "C" mfa_status mfa_gemm_kernel_launch(const mfa_gemm_kernel *k, const void *A, const void *B, void *C,
                                             const mfa_gemm_launch_params *p, void *stream) {
  GemmArgs g;
  dim3 grid;
  bool use16;
  const mfa_status st = prepare(k, A, B, C, p, &g, &grid, &use16);
  if (st != MFA_OK) return st;
  hipError_t e = launch_one(k, g, grid, use16, (hipStream_t)stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) return fail(MFA_ERR_HIP, std::string("gemm launch: ") + hipGetErrorString(e));
  return MFA_OK;
}

extern "C" mfa_status mfa_gemm_kernel_time(const mfa_gemm_kernel *k, const void *A, const void *B, void *C,
                                           const mfa_gemm_launch_params *p, void *stream, int warmup, int iterations, float *ms) {
  if (!ms || iterations <= 0) return fail(MFA_ERR_INVALID_ARGUMENT, "bad timing arguments");
  GemmArgs g;
  dim3 grid;
  bool use16;
  const mfa_status st = prepare(k, A, B, C, p, &g, &grid, &use16);
  if (st != MFA_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(MFA_ERR_HIP, "hipEventCreate failed");
  for (int i = 0; i < warmup; ++i) (void)launch_one(k, g, grid, use16, s);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iterations; ++i) (void)launch_one(k, g, grid, use16, s);
  (void)hipEventRecord(e1, s);
  hipError_t e = hipEventSynchronize(e1);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return fail(MFA_ERR_HIP, std::string("gemm timing: ") + hipGetErrorString(e));
  return MFA_OK;
}
