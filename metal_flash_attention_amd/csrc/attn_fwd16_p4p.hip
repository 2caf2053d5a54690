// attn_fwd16_p4p.hip -- instantiations and launcher of the persistent four-wave forward kernel (attn_fwd16_p4p.h).
#include "attn_fwd16_p4p.h"
#include "launchers.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace mfa {

// sleep steps (x 512 clocks) per unit of (workgroup >> 3) & 31 in front of a workgroup's first block
#ifndef P4P_STAGGER
#define P4P_STAGGER 0
#endif

namespace {

struct DeviceInfo { int cus = 0; uint64_t attrMask[p4p::S_COUNT * 2 + 2] = {}; };
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
    constexpr int slot = STREAM * 2 + (__is_same(T, _Float16) ? 1 : 0);
    if (!d.attrMask[slot]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_fwd16_p4p<T, STREAM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              p4p::LDS_BYTES) != hipSuccess)
        return false;
      d.attrMask[slot] = 1;
    }
  }
  // units: row blocks, or (causal) pairs of row blocks -- two table entries each
  constexpr bool CAUSAL = p4p::traits(STREAM).causal;
  constexpr uint64_t PER_UNIT = CAUSAL ? 2 : 1;
  const uint64_t total = (uint64_t)(CAUSAL ? (grid.x + 1) / 2 : grid.x) * grid.y * grid.z * splits;
  // one workgroup per compute unit; more only when a workgroup's share would not fit the block table.  A multiple of 8 keeps
  // fwd16_decode_block's head -> XCD affinity for every block of a workgroup
  uint64_t groups = total < (uint64_t)cus ? total : (uint64_t)cus;
  constexpr uint64_t MAX_UNITS = (p4p::TABLE_ENTRIES - 1) / PER_UNIT;   // (the table's last word holds the block count)
  if ((total + groups - 1) / groups > MAX_UNITS) groups = (total + MAX_UNITS - 1) / MAX_UNITS;
  if (groups >= 8) groups = (groups + 7) / 8 * 8;
  if (groups > total) groups = total;
  if ((total + groups - 1) / groups > MAX_UNITS) return false;
  Fwd16Grid g{grid.x, grid.y, grid.z, splits, wsO, wsML};
  uint32_t stagger = P4P_STAGGER;
#ifdef MFA_DEV_VARIANTS
  if (const char *e = std::getenv("MFA_P4P_STAGGER")) stagger = (uint32_t)std::atoi(e);
#endif
  hipLaunchKernelGGL((attn_fwd16_p4p<T, STREAM>), dim3((uint32_t)groups), dim3(256), p4p::LDS_BYTES, stream, args, g, (uint32_t)total, stagger);
  return true;
}

}  // namespace

// Dense launch of a D <= 128 forward problem on the persistent kernel.  Returns false when the launch is not one it serves
// (the caller then launches attn_fwd16_p4): block masks, a storage type of O / L no stream was generated for.
template <typename T, bool FOLD> bool launch_p4p(dim3 grid, hipStream_t stream, const KernelArgs &args) {
  if (args.mask) return false;
  if (args.causal && args.C < args.R) return false;
  // per-batch lengths (round 6): the causal ("geometry") streams carry the rows and keys of a block's batch entry in its table entry and
  // serve such launches with or without the causal mask (KernelArgs.causal is the stream's flag).  Without the mask they win (+4 % on
  // full-length batches, +1 % on mixed lengths against the one-block-per-workgroup kernel; interleaved rounds, profiles/r06_final/
  // time_varlen_d128.txt); WITH it a workgroup's fixed share of (long, short) row-block pairs is 6 % slower on mixed lengths than what
  // the dispatcher balances block by block (+4 % on full-length batches, which the host cannot tell apart: the lengths are device
  // arrays) -- causal launches with lengths stay with attn_fwd16_p4; the developer library routes them here with MFA_P4P_LENGTHS=1
  bool lengths_here = !args.causal;
#ifdef MFA_DEV_VARIANTS
  lengths_here = lengths_here || std::getenv("MFA_P4P_LENGTHS") != nullptr;
#endif
  if ((args.rowLen || args.colLen) && !lengths_here) return false;
  const bool geometry = args.causal || args.rowLen || args.colLen;
#ifdef MFA_DEV_VARIANTS   // developer builds: A/B against the one-block-per-workgroup kernel, phase clocks (tools/p4p_prof.py)
  if (std::getenv("MFA_P4_NO_PERSISTENT")) return false;
  if constexpr (!FOLD && __is_same(T, __bf16)) {
    if (std::getenv("MFA_P4P_PROF") && args.op[SLOT_O].precision == PREC_FP32 && args.op[SLOT_L].precision == PREC_FP32)
      return launch_stream<T, p4p::S_BF16_EXACT_PROF>(grid, stream, args);
  }
  // MFA_P4P_DEV_STREAM=<name of a developer stream of tools/p4pgen.py>: dense bf16 launches with FP32 O whose mode (mixed: FP16 L;
  // fp32 intermediates: FP32 L) the stream was generated for run that stream (schedule experiments and timing-only ablations,
  // tools/p4p_streams_ab.py)
  if constexpr (__is_same(T, __bf16)) {
    const char *want = std::getenv("MFA_P4P_DEV_STREAM");
    if (want && *want && args.op[SLOT_O].precision == PREC_FP32 && args.op[SLOT_L].precision == (FOLD ? PREC_FP16 : PREC_FP32)) {
#define MFA_P4P_BYNAME(name, f16, fold, o16, l16, scausal) \
      if constexpr (!f16 && fold == FOLD && !o16 && l16 == FOLD) { if (((scausal) & 3) != 2 && (((scausal) & 3) != 0) == geometry && std::strcmp(want, #name) == 0) return launch_stream<T, p4p::S_##name>(grid, stream, args); }
      MFA_P4P_DEV_STREAM_LIST(MFA_P4P_BYNAME)
#undef MFA_P4P_BYNAME
      return false;   // (an unknown name must not silently time the product stream)
    }
  }
#endif
  const int po = args.op[SLOT_O].precision, pl = args.op[SLOT_L].precision;
  constexpr int PT = __is_same(T, _Float16) ? PREC_FP16 : PREC_BF16;
  const bool o16 = po == PT, l16 = pl == PREC_FP16;
  if (!o16 && po != PREC_FP32) return false;
  if (!l16 && pl != PREC_FP32) return false;
  if constexpr (FOLD) {
    if (!l16) return false;   // (FOLD streams exist with FP16 L: the mixed-precision mode's storage type)
    if constexpr (__is_same(T, _Float16)) {
      if (geometry) return o16 ? launch_stream<T, p4p::S_F16_FOLD_O16_L16_CAUSAL>(grid, stream, args) : launch_stream<T, p4p::S_F16_FOLD_L16_CAUSAL>(grid, stream, args);
      return o16 ? launch_stream<T, p4p::S_F16_FOLD_O16_L16>(grid, stream, args) : launch_stream<T, p4p::S_F16_FOLD_L16>(grid, stream, args);
    } else {
      if (geometry) return o16 ? launch_stream<T, p4p::S_BF16_FOLD_O16_L16_CAUSAL>(grid, stream, args) : launch_stream<T, p4p::S_BF16_FOLD_L16_CAUSAL>(grid, stream, args);
      return o16 ? launch_stream<T, p4p::S_BF16_FOLD_O16_L16>(grid, stream, args) : launch_stream<T, p4p::S_BF16_FOLD_L16>(grid, stream, args);
    }
  } else {
    if (l16) return false;
    if constexpr (__is_same(T, _Float16)) {
      if (geometry) return o16 ? launch_stream<T, p4p::S_F16_EXACT_O16_CAUSAL>(grid, stream, args) : launch_stream<T, p4p::S_F16_EXACT_CAUSAL>(grid, stream, args);
      return o16 ? launch_stream<T, p4p::S_F16_EXACT_O16>(grid, stream, args) : launch_stream<T, p4p::S_F16_EXACT>(grid, stream, args);
    } else {
      if (geometry) return o16 ? launch_stream<T, p4p::S_BF16_EXACT_O16_CAUSAL>(grid, stream, args) : launch_stream<T, p4p::S_BF16_EXACT_CAUSAL>(grid, stream, args);
      return o16 ? launch_stream<T, p4p::S_BF16_EXACT_O16>(grid, stream, args) : launch_stream<T, p4p::S_BF16_EXACT>(grid, stream, args);
    }
  }
}

// Column-parallel launch (few-workgroup problems: one head -- the reference's own benchmark shape): the pieces of the key range on the
// persistent kernel's split streams (round 6); the caller launches attn_fwd_combine behind it.  false = not one it serves (pieces that
// are not whole multiples of two tiles): the caller launches the one-block-per-workgroup kernel's pieces
template <typename T, bool FOLD> bool p4p_split_serves(const KernelArgs &args, uint32_t splits) {
  if (args.rowLen || args.colLen || args.mask || args.causal || splits < 2) return false;
  if (args.C % (128u * splits) != 0) return false;
#ifdef MFA_DEV_VARIANTS
  if (std::getenv("MFA_P4_NO_PERSISTENT") || std::getenv("MFA_P4P_NO_SPLIT")) return false;
#endif
  return true;
}
template <typename T, bool FOLD> bool launch_p4p_split(dim3 grid, uint32_t splits, float *wsO, float *wsML, hipStream_t stream, const KernelArgs &args) {
  if (!p4p_split_serves<T, FOLD>(args, splits)) return false;
  if constexpr (__is_same(T, _Float16))
    return FOLD ? launch_stream<T, p4p::S_F16_FOLD_SPLIT>(grid, stream, args, splits, wsO, wsML) : launch_stream<T, p4p::S_F16_EXACT_SPLIT>(grid, stream, args, splits, wsO, wsML);
  else
    return FOLD ? launch_stream<T, p4p::S_BF16_FOLD_SPLIT>(grid, stream, args, splits, wsO, wsML) : launch_stream<T, p4p::S_BF16_EXACT_SPLIT>(grid, stream, args, splits, wsO, wsML);
}
template bool launch_p4p_split<__bf16, true>(dim3, uint32_t, float *, float *, hipStream_t, const KernelArgs &);
template bool launch_p4p_split<__bf16, false>(dim3, uint32_t, float *, float *, hipStream_t, const KernelArgs &);
template bool launch_p4p_split<_Float16, true>(dim3, uint32_t, float *, float *, hipStream_t, const KernelArgs &);
template bool launch_p4p_split<_Float16, false>(dim3, uint32_t, float *, float *, hipStream_t, const KernelArgs &);
template bool p4p_split_serves<__bf16, true>(const KernelArgs &, uint32_t);
template bool p4p_split_serves<__bf16, false>(const KernelArgs &, uint32_t);
template bool p4p_split_serves<_Float16, true>(const KernelArgs &, uint32_t);
template bool p4p_split_serves<_Float16, false>(const KernelArgs &, uint32_t);

// the launches launch_p4p serves (the same conditions, nothing launched)
template <typename T, bool FOLD> const char *p4p_form(const KernelArgs &args) {
  if (args.mask) return nullptr;
  if (args.causal && args.C < args.R) return nullptr;
  bool lengths_here = !args.causal;   // (causal launches with per-batch lengths: attn_fwd16_p4, see launch_p4p)
#ifdef MFA_DEV_VARIANTS
  lengths_here = lengths_here || std::getenv("MFA_P4P_LENGTHS") != nullptr;
#endif
  if ((args.rowLen || args.colLen) && !lengths_here) return nullptr;
#ifdef MFA_DEV_VARIANTS
  if (std::getenv("MFA_P4_NO_PERSISTENT")) return nullptr;
#endif
  const int po = args.op[SLOT_O].precision, pl = args.op[SLOT_L].precision;
  constexpr int PT = __is_same(T, _Float16) ? PREC_FP16 : PREC_BF16;
  if (po != PT && po != PREC_FP32) return nullptr;
  if (FOLD ? pl != PREC_FP16 : pl != PREC_FP32) return nullptr;
  if (args.rowLen || args.colLen)
    return args.causal ? "attn_fwd16_p4p (persistent: one workgroup per compute unit walks the row-block pairs; per-batch lengths in the block table)"
                       : "attn_fwd16_p4p (persistent: one workgroup per compute unit walks the row-block pairs; per-batch lengths in the block table, no causal mask)";
  return args.causal ? "attn_fwd16_p4p (persistent: one workgroup per compute unit walks the row-block pairs)"
                     : "attn_fwd16_p4p (persistent: one workgroup per compute unit walks the row blocks)";
}
template const char *p4p_form<__bf16, true>(const KernelArgs &);
template const char *p4p_form<__bf16, false>(const KernelArgs &);
template const char *p4p_form<_Float16, true>(const KernelArgs &);
template const char *p4p_form<_Float16, false>(const KernelArgs &);

template bool launch_p4p<__bf16, true>(dim3, hipStream_t, const KernelArgs &);
template bool launch_p4p<__bf16, false>(dim3, hipStream_t, const KernelArgs &);
template bool launch_p4p<_Float16, true>(dim3, hipStream_t, const KernelArgs &);
template bool launch_p4p<_Float16, false>(dim3, hipStream_t, const KernelArgs &);

} // namespace mfa
