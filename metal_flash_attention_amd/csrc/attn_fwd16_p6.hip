// attn_fwd16_p6.hip -- instantiations and launcher of the persistent D <= 64 forward kernel (attn_fwd16_p6.h).
#include "attn_fwd16_p6.h"
#include "launchers.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace mfa {

namespace {

struct DeviceInfo { int cus = 0; uint64_t attrMask[p6::S_COUNT + 1] = {}; };
std::mutex g_mutex;
DeviceInfo g_devices[64];

template <typename T, int STREAM>
bool launch_stream(dim3 grid, hipStream_t stream, const KernelArgs &args, uint32_t splits = 1, float *wsO = nullptr, float *wsML = nullptr) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return false;
  int cus;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceInfo &d = g_devices[device];
    if (d.cus == 0) {
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) return false;
      d.cus = n;
    }
    cus = d.cus;
    if (!d.attrMask[STREAM]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_fwd16_p6<T, STREAM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              p6::LDS_BYTES) != hipSuccess)
        return false;
      d.attrMask[STREAM] = 1;
    }
  }
  // units: row blocks, or (causal) pairs of row blocks -- two table entries each
  constexpr bool CAUSAL = p6::traits(STREAM).causal;
  constexpr uint64_t PER_UNIT = CAUSAL ? 2 : 1;
  const uint64_t total = (uint64_t)(CAUSAL ? (grid.x + 1) / 2 : grid.x) * grid.y * grid.z * splits;
  // one workgroup per compute unit; more only when a workgroup's share would not fit the block table.  A multiple of 8 keeps
  // fwd16_decode_block's head -> XCD affinity for every block of a workgroup
  uint64_t groups = total < (uint64_t)cus ? total : (uint64_t)cus;
  constexpr uint64_t MAX_UNITS = (p6::TABLE_ENTRIES - 1) / PER_UNIT;   // (the table's last word holds the block count)
  if ((total + groups - 1) / groups > MAX_UNITS) groups = (total + MAX_UNITS - 1) / MAX_UNITS;
  if (groups >= 8) groups = (groups + 7) / 8 * 8;
  if (groups > total) groups = total;
  if ((total + groups - 1) / groups > MAX_UNITS) return false;
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, wsO, wsML};
  hipLaunchKernelGGL((attn_fwd16_p6<T, STREAM>), dim3((uint32_t)groups), dim3(256), p6::LDS_BYTES, stream, args, g, (uint32_t)total);
  return true;
}

bool serves(int precision, bool fold, const KernelArgs &args) {
  if (precision != PREC_BF16 && precision != PREC_FP16) return false;
  if (args.mask) return false;   // (per-batch lengths: served since round 6 -- the causal streams read rows / keys per block-table entry)
  if (args.causal && args.C < args.R) return false;
  if (args.D > 64 || args.D % 8) return false;
  const int po = args.op[SLOT_O].precision, pl = args.op[SLOT_L].precision;
  if (po != precision && po != PREC_FP32) return false;
  if (pl != (fold ? PREC_FP16 : PREC_FP32)) return false;   // (FOLD streams store FP16 L, the mixed-precision mode's type; EXACT ones FP32)
  for (int slot : {SLOT_Q, SLOT_K, SLOT_V, SLOT_O})
    if (args.op[slot].transposed) return false;
  return true;
}

}  // namespace

// Dense launch of a D <= 64 forward problem on the persistent kernel; false = not one it serves (the caller launches attn_fwd16_v3)
bool launch_p6(int precision, bool fold, dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if (!serves(precision, fold, args)) return false;
#ifdef MFA_DEV_VARIANTS
  if (std::getenv("MFA_P6_OFF")) return false;
  if (const char *want = std::getenv("MFA_P6_DEV_STREAM")) {
    if (*want && precision == PREC_BF16 && args.op[SLOT_O].precision == PREC_FP32) {
#define MFA_P6_BYNAME(name, f16, sfold, o16, l16, scausal, ssplit) \
      if constexpr (!f16 && !o16 && !ssplit) { if ((sfold != 0) == fold && (scausal != 0) == (args.causal != 0 || args.rowLen || args.colLen) && std::strcmp(want, #name) == 0) return launch_stream<__bf16, p6::S_##name>(grid, stream, args); }
      MFA_P6_DEV_STREAM_LIST(MFA_P6_BYNAME)
#undef MFA_P6_BYNAME
      return false;
    }
  }
#endif
  const bool o16 = args.op[SLOT_O].precision != PREC_FP32;
#define MFA_P6_PICK(T, PFX)                                                                                                          \
  if (args.causal || args.rowLen || args.colLen) {   /* the "geometry" streams; KernelArgs.causal is their run-time flag */           \
    if (fold) return o16 ? launch_stream<T, p6::S_##PFX##_FOLD_O16_L16_CAUSAL>(grid, stream, args) : launch_stream<T, p6::S_##PFX##_FOLD_L16_CAUSAL>(grid, stream, args); \
    return o16 ? launch_stream<T, p6::S_##PFX##_EXACT_O16_CAUSAL>(grid, stream, args) : launch_stream<T, p6::S_##PFX##_EXACT_CAUSAL>(grid, stream, args); \
  }                                                                                                                                  \
  if (fold) return o16 ? launch_stream<T, p6::S_##PFX##_FOLD_O16_L16>(grid, stream, args) : launch_stream<T, p6::S_##PFX##_FOLD_L16>(grid, stream, args); \
  return o16 ? launch_stream<T, p6::S_##PFX##_EXACT_O16>(grid, stream, args) : launch_stream<T, p6::S_##PFX##_EXACT>(grid, stream, args);
  if (precision == PREC_BF16) { MFA_P6_PICK(__bf16, BF16) }
  MFA_P6_PICK(_Float16, F16)
#undef MFA_P6_PICK
}

// Column-parallel launch (few-workgroup problems: one head, BASELINE config 2 as written): pieces of the key range on the persistent
// kernel, then the combine pass.  false = not one it serves (pieces that are not whole multiples of four tiles, ...): the caller
// launches the eight-wave kernel's split sibling
bool launch_p6_split(int precision, bool fold, dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) {
  if (!serves(precision, fold, args) || args.causal || args.rowLen || args.colLen || splits < 2) return false;
  if (args.C % (256u * splits) != 0) return false;
#ifdef MFA_DEV_VARIANTS
  if (std::getenv("MFA_P6_OFF") || std::getenv("MFA_P6_NO_SPLIT")) return false;
#endif
  bool ok;
  if (precision == PREC_BF16) ok = fold ? launch_stream<__bf16, p6::S_BF16_FOLD_SPLIT>(grid, stream, args, splits, wsO, wsML)
                                        : launch_stream<__bf16, p6::S_BF16_EXACT_SPLIT>(grid, stream, args, splits, wsO, wsML);
  else ok = fold ? launch_stream<_Float16, p6::S_F16_FOLD_SPLIT>(grid, stream, args, splits, wsO, wsML)
                 : launch_stream<_Float16, p6::S_F16_EXACT_SPLIT>(grid, stream, args, splits, wsO, wsML);
  if (!ok) return false;
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, wsO, wsML};
  const uint64_t rows = (uint64_t)grid.y * grid.z * args.R;
  hipLaunchKernelGGL(attn_fwd_combine, dim3((uint32_t)((rows + 3) / 4)), dim3(256), 0, stream, args, g);
  return true;
}

// what runs instead when this kernel does not serve a launch (launch_p6_or_v3*): the name rocprofv3 shows is the eight-wave kernel's
static const char *unserved_form(int precision, const KernelArgs &args) {
  if (precision == PREC_BF16)
    return args.causal ? "attn_fwd16v3_bf16_d64_w8x32_thr8 causal (eight-wave kernel: the launch is not one attn_fwd16_p6 serves)"
                       : "attn_fwd16v3_bf16_d64_w8x32_thr8 (eight-wave kernel: the launch is not one attn_fwd16_p6 serves)";
  return args.causal ? "attn_fwd16v3_f16_d64_w8x32_thr8 causal (eight-wave kernel: the launch is not one attn_fwd16_p6 serves)"
                     : "attn_fwd16v3_f16_d64_w8x32_thr8 (eight-wave kernel: the launch is not one attn_fwd16_p6 serves)";
}

const char *p6_form(int precision, bool fold, const KernelArgs &args) {
  if (!serves(precision, fold, args)) return nullptr;
#ifdef MFA_DEV_VARIANTS
  if (std::getenv("MFA_P6_OFF")) return nullptr;
#endif
  if (args.rowLen || args.colLen)
    return fold ? "attn_fwd16_p6 (persistent: one workgroup per compute unit walks the row-block pairs; per-batch lengths in the block table; row sums in the matrix pipe)"
                : "attn_fwd16_p6 (persistent: one workgroup per compute unit walks the row-block pairs; per-batch lengths in the block table)";
  if (args.causal)
    return fold ? "attn_fwd16_p6 (persistent: one workgroup per compute unit walks the row-block pairs; row sums in the matrix pipe)"
                : "attn_fwd16_p6 (persistent: one workgroup per compute unit walks the row-block pairs)";
  return fold ? "attn_fwd16_p6 (persistent: one workgroup per compute unit walks the row blocks; row sums in the matrix pipe)"
              : "attn_fwd16_p6 (persistent: one workgroup per compute unit walks the row blocks)";
}

void fwd16_v3_d64_launch(int precision, dim3 grid, hipStream_t stream, const KernelArgs &args);   // attn_fwd16_v3.hip

template <int PREC, bool FOLD> static void launch_p6_or_v3(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if (launch_p6(PREC, FOLD, grid, stream, args)) return;
  fwd16_v3_d64_launch(PREC, grid, stream, args);
}
void fwd16_v3_d64_launch_causal(int precision, dim3 grid, hipStream_t stream, const KernelArgs &args);   // attn_fwd16_v3.hip
void fwd16_v3_d64_launch_split(int precision, dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args);
template <int PREC, bool FOLD> static void launch_p6_or_v3_split(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) {
  if (launch_p6_split(PREC, FOLD, grid, splits, wsO, wsML, stream, args)) return;
  fwd16_v3_d64_launch_split(PREC, grid, splits, wsO, wsML, stream, args);
}
template <int PREC, bool FOLD> static void launch_p6_or_v3_causal(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if (launch_p6(PREC, FOLD, grid, stream, args)) return;
  fwd16_v3_d64_launch_causal(PREC, grid, stream, args);
}
template <int PREC, bool FOLD> static const char *p6_split_form_of(const KernelArgs &args, uint32_t splits) {
  static const char *const sibling = PREC == PREC_BF16 ? "pieces by the eight-wave kernel attn_fwd16v3_bf16_d64_w8x32_thr8 (not a split attn_fwd16_p6 serves)"
                                                       : "pieces by the eight-wave kernel attn_fwd16v3_f16_d64_w8x32_thr8 (not a split attn_fwd16_p6 serves)";
  if (!p6_form(PREC, FOLD, args) || args.causal || splits < 2 || args.C % (256u * splits) != 0) return sibling;
#ifdef MFA_DEV_VARIANTS
  if (std::getenv("MFA_P6_NO_SPLIT")) return sibling;
#endif
  return "pieces by attn_fwd16_p6, persistent";
}
template <int PREC, bool FOLD> static const char *p6_form_of(const KernelArgs &args) {
  const char *form = p6_form(PREC, FOLD, args);
  return form ? form : unserved_form(PREC, args);
}

// `out` arrives filled by fwd16_v3_variant(precision, 64, 0): its causal, block-sparse and column-parallel launches (and the dense
// launches this kernel does not serve: transposed operands, other storage types of L) stay with that kernel, which becomes the sibling
bool fwd16_p6_variant(int precision, bool fold, VariantInfo *out) {
  if (precision != PREC_BF16 && precision != PREC_FP16) return false;
  if (out->name && out->name[0]) out->siblingName = out->name;
  out->name = precision == PREC_BF16 ? (fold ? "attn_fwd16p6_bf16_d64_w4x64_thr8_fold" : "attn_fwd16p6_bf16_d64_w4x64_thr8")
                                     : (fold ? "attn_fwd16p6_f16_d64_w4x64_thr8_fold" : "attn_fwd16p6_f16_d64_w4x64_thr8");
  out->siblingParallelization = out->parallelization;
  out->parallelization = 256;
  out->traversal = 64;
  out->headBlock = 64;
  out->threads = 256;
  out->ldsBytes = out->ldsBytes > (uint32_t)p6::LDS_BYTES ? out->ldsBytes : (uint32_t)p6::LDS_BYTES;
  out->splitTarget = 256;   // one workgroup per compute unit
  if (precision == PREC_BF16) {
    if (fold) { out->launch = &launch_p6_or_v3<PREC_BF16, true>; out->launchCausal = &launch_p6_or_v3_causal<PREC_BF16, true>; out->launchSplit = &launch_p6_or_v3_split<PREC_BF16, true>; out->splitForm = &p6_split_form_of<PREC_BF16, true>; out->launchForm = &p6_form_of<PREC_BF16, true>; }
    else { out->launch = &launch_p6_or_v3<PREC_BF16, false>; out->launchCausal = &launch_p6_or_v3_causal<PREC_BF16, false>; out->launchSplit = &launch_p6_or_v3_split<PREC_BF16, false>; out->splitForm = &p6_split_form_of<PREC_BF16, false>; out->launchForm = &p6_form_of<PREC_BF16, false>; }
  } else {
    if (fold) { out->launch = &launch_p6_or_v3<PREC_FP16, true>; out->launchCausal = &launch_p6_or_v3_causal<PREC_FP16, true>; out->launchSplit = &launch_p6_or_v3_split<PREC_FP16, true>; out->splitForm = &p6_split_form_of<PREC_FP16, true>; out->launchForm = &p6_form_of<PREC_FP16, true>; }
    else { out->launch = &launch_p6_or_v3<PREC_FP16, false>; out->launchCausal = &launch_p6_or_v3_causal<PREC_FP16, false>; out->launchSplit = &launch_p6_or_v3_split<PREC_FP16, false>; out->splitForm = &p6_split_form_of<PREC_FP16, false>; out->launchForm = &p6_form_of<PREC_FP16, false>; }
  }
  return true;
}

} // namespace mfa
