// mfa_descriptor.cpp -- host-side plugin surface: enums, precision policy, gfx950 parameter
// tables and AttentionDescriptor.kernelDescriptor(type:).  Pure C++ (no HIP calls).
//
// Mirrors, behind the C ABI of include/mfa.h:
//   Sources/FlashAttention/Attention/AttentionDescriptor/AttentionDescriptor.swift:10-148
//   Sources/FlashAttention/Attention/AttentionDescriptor/AttentionDescriptor+Parameters.swift:13-66
//   Sources/FlashAttention/Attention/AttentionDescriptor/AttentionDescriptor+Precisions.swift:10-215
//   Sources/FlashAttention/Attention/AttentionDescriptor/AttentionParameterRow.swift:8-106
//   Sources/FlashAttention/Attention/AttentionOperand.swift:9-71
//   Sources/FlashAttention/GEMM/GEMMOperandPrecision.swift:33-60
// The table VALUES are re-derived for gfx950; the 5-column text
// format, the row-selection rule and the validation rules are the reference's.
#include "mfa_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

namespace mfa {

static thread_local std::string g_last_error;

mfa_status fail(mfa_status status, const std::string &message) {
  g_last_error = message;
  return status;
}

// ---------------------------------------------------------------------------------------------
// gfx950 parameter tables.  Columns (AttentionDescriptor+Parameters.swift:106-148):
//   | max D | parallelization | traversal | head | cached operands |
// parallelization = rows (fwd, dQ) / columns (dK/dV) per workgroup; a multiple of 32 because one
// wave owns one 32x32 MFMA tile.  traversal = columns (rows) consumed per main-loop (pipeline) step.
// head = head-dimension block the code object is unrolled over.  "FP32" tables list the fp32-arithmetic
// kernels (attn_generic.h), "mixed" tables the 16-bit matrix-core kernels (attn_fwd16_p4.h, attn_fwd16_v3.h,
// attn_bwd16.h, attn_dkv16_rs.h).  Every row below IS a compiled code object: mfa_attention_kernel_create matches
// the row against the variants (tests/test_host_api.py::test_default_table_rows_are_compiled_variants keeps the two
// from drifting apart); a row no variant implements falls back to the nearest one, or fails with
// strictBlockDimensions.  Accumulators stay in registers at every supported D (512 registers per lane); the 384 rows are
// the gfx950 form of the reference's large-D rows (+Parameters.swift:77-285): what gets evicted first there -- the
// left-hand operands -- is what leaves the registers here too (dO, then K and V), the accumulators never do.
// ---------------------------------------------------------------------------------------------
static const char *kDefaultTables[3][2] = {
    {// forward, FP32
     "| 32  | 128 | 32 | 32  | Q, O |\n"
     "| 64  | 128 | 32 | 64  | Q, O |\n"
     "| 128 | 128 | 32 | 128 | Q, O |\n"
     "| 256 | 128 | 32 | 256 | Q, O |\n"
     "| 384 | 64  | 32 | 384 | Q, O |\n",
     // forward, mixed: D <= 128 -> 4 waves x 64 rows, 64-key steps (attn_fwd16_p4.h); | 128 | 256 | 32 | 128 | selects the
     // 8 waves x 32 rows kernel with 32-key pipeline steps (attn_fwd16_v3.h), which also serves the other buckets;
     // D in (128, 256] -> 4 waves x 64 rows, 32-key steps (attn_fwd16_p5.h; buckets 160, 192, 256); | D | 128 | 32 | D | selects
     // the 4 waves x 32 rows objects; D in (256, 384] -> 4 waves x 32 rows, 32-key steps, Q' and O^T in registers (attn_fwd16_wide.h, round 6:
     // the reference's `| 384 | ... |` rows on the 16-bit matrix cores; with FP32 inputs the row is the general kernel's | 384 | 64 | 32 | 384 |)
     "| 32  | 128 | 32 | 32  | Q, O |\n"
     "| 64  | 256 | 64 | 64  | Q, O |\n"
     "| 128 | 256 | 64 | 128 | Q, O |\n"
     "| 160 | 256 | 32 | 160 | Q, O |\n"
     "| 192 | 256 | 32 | 192 | Q, O |\n"
     "| 256 | 256 | 32 | 256 | Q, O |\n"
     "| 320 | 128 | 32 | 320 | Q, O |\n"
     "| 384 | 128 | 32 | 384 | Q, O |\n"},
    {// backwardQuery, FP32
     "| 32  | 128 | 32 | 32  | Q, dO, dQ |\n"
     "| 64  | 128 | 32 | 64  | Q, dO, dQ |\n"
     "| 128 | 128 | 32 | 128 | Q, dO, dQ |\n"
     "| 256 | 128 | 32 | 256 | Q, dO, dQ |\n"
     "| 384 | 32  | 32 | 384 | Q, dQ     |\n",
     // backwardQuery, mixed: D <= 128 -> 4 waves x 64 rows, 64-key tiles (attn_dq16_p4.h); D > 128 -> two role-split pairs x 64 rows,
     // 32-key blocks (attn_dq16_p5.h, round 4); | D | 128 | 64 | D | selects the four 32-row waves of attn_bwd16.h there
     "| 64  | 256 | 64 | 64  | Q, dO, dQ |\n"
     "| 128 | 256 | 64 | 128 | Q, dO, dQ |\n"
     "| 160 | 128 | 32 | 160 | Q, dO, dQ |\n"
     "| 192 | 128 | 32 | 192 | Q, dO, dQ |\n"
     "| 256 | 128 | 32 | 256 | Q, dO, dQ |\n"
     // D in (256, 384] (round 6): four waves x 32 rows, 32-key tiles, Q / dO fragments and dQ^T in registers (attn_bwd16_wide.hip)
     "| 320 | 128 | 32 | 320 | Q, dO, dQ |\n"
     "| 384 | 128 | 32 | 384 | Q, dO, dQ |\n"},
    {// backwardKeyValue, FP32
     "| 32  | 128 | 32 | 32  | K, V, dV, dK |\n"
     "| 64  | 128 | 32 | 64  | K, V, dV, dK |\n"
     "| 128 | 128 | 32 | 128 | K, V, dV, dK |\n"
     "| 256 | 128 | 32 | 256 | K, V, dV, dK |\n"
     "| 384 | 32  | 32 | 384 | dV, dK       |\n",
     // backwardKeyValue, mixed: D <= 128 -> 4 waves x 64 keys (attn_dkv16_p4.h); | 128 | 128 | 32 | 128 | selects the role-split wave pairs
     // (attn_dkv16_rs.h; a 96-wide object too: | 96 | 128 | 32 | 96 |); | 128 | 128 | 64 | 128 | the one-wave-per-key-block kernel
     // (attn_bwd16.h); D > 128 -> two role-split pairs x 64 keys (attn_dkv16_p5.h, round 4), | D | 64 | 32 | D | the 32-key pairs
     "| 64  | 256 | 32 | 64  | K, V, dV, dK |\n"
     "| 128 | 256 | 32 | 128 | K, V, dV, dK |\n"
     "| 160 | 128 | 32 | 160 | K, V, dV, dK |\n"
     "| 192 | 128 | 32 | 192 | K, V, dV, dK |\n"
     "| 256 | 128 | 32 | 256 | K, V, dV, dK |\n"
     // D in (256, 384] (round 6): two role-split pairs x 32 keys, 32-row steps, two LDS stages (attn_dkv16_wide.h)
     "| 320 | 64  | 32 | 320 | K, V, dV, dK |\n"
     "| 384 | 64  | 32 | 384 | K, V, dV, dK |\n"}};

static std::mutex g_table_mutex;
static std::string g_tables[3][2];
static bool g_tables_initialised = false;

static void ensure_tables() {
  if (g_tables_initialised) return;
  for (int t = 0; t < 3; ++t)
    for (int m = 0; m < 2; ++m) g_tables[t][m] = kDefaultTables[t][m];
  g_tables_initialised = true;
}

struct ParsedRow {
  uint16_t maximumHeadDimension = 0;
  std::string parallelization, traversal, head, cachedOperands;
};

// AttentionParameterRow.parseTable (AttentionParameterRow.swift:22-74)
static mfa_status parse_table(const std::string &file, std::vector<ParsedRow> *rows) {
  std::istringstream stream(file);
  std::string line;
  while (std::getline(stream, line, '\n')) {
    // Swift's split(separator:) drops empty subsequences: skip blank lines, and empty cells
    // between adjacent bars disappear -- but a cell of spaces survives as an empty string.
    std::vector<std::string> segments;
    size_t pos = 0;
    while (pos <= line.size()) {
      size_t bar = line.find('|', pos);
      if (bar == std::string::npos) bar = line.size();
      if (bar > pos) {
        std::string cell = line.substr(pos, bar - pos);
        cell.erase(std::remove(cell.begin(), cell.end(), ' '), cell.end()); // strip 0x20 (:38-40)
        segments.push_back(cell);
      }
      pos = bar + 1;
    }
    if (segments.empty()) continue;
    if (segments.size() != 5)  // (:50-52)
      return fail(MFA_ERR_PARSE, "Number of segments was invalid: " + std::to_string(segments.size()));
    ParsedRow row;
    char *end = nullptr;
    const unsigned long maxD = std::strtoul(segments[0].c_str(), &end, 10);
    if (segments[0].empty() || *end != '\0' || maxD > 65535)  // (:55-58)
      return fail(MFA_ERR_PARSE, "Could not extract maximum head dimension.");
    row.maximumHeadDimension = (uint16_t)maxD;
    row.parallelization = segments[1];
    row.traversal = segments[2];
    row.head = segments[3];
    row.cachedOperands = segments[4];
    rows->push_back(row);
  }
  if (rows->empty()) return fail(MFA_ERR_PARSE, "Parameter table has no rows.");
  return MFA_OK;
}

// AttentionParameterRow.parseOperands (AttentionParameterRow.swift:76-106)
static mfa_status parse_operands(const std::string &text, std::vector<int> *operands) {
  static const int accepted[] = {MFA_Q, MFA_K, MFA_V, MFA_O, MFA_dO, MFA_dV, MFA_dK, MFA_dQ};
  size_t pos = 0;
  while (pos <= text.size()) {
    size_t comma = text.find(',', pos);
    if (comma == std::string::npos) comma = text.size();
    if (comma > pos) {
      const std::string name = text.substr(pos, comma - pos);
      int matched = -1;
      for (int op : accepted)
        if (name == mfa_operand_name(op)) matched = op;
      if (matched < 0) return fail(MFA_ERR_PARSE, "Could not find match for " + name + ".");
      operands->push_back(matched);
    }
    pos = comma + 1;
  }
  return MFA_OK;
}

static bool parse_u16(const std::string &s, uint16_t *out) {
  if (s.empty()) return false;
  char *end = nullptr;
  const unsigned long v = std::strtoul(s.c_str(), &end, 10);
  if (*end != '\0' || v > 65535) return false;
  *out = (uint16_t)v;
  return true;
}

// parseTable + row(table:) (AttentionDescriptor+Parameters.swift:41-66): first row whose
// maximum head dimension is >= D, else the last row.
static mfa_status select_row(const std::string &text, uint16_t headDimension, mfa_parameter_row *out) {
  std::vector<ParsedRow> rows;
  mfa_status st = parse_table(text, &rows);
  if (st != MFA_OK) return st;
  const ParsedRow *matched = &rows.back();
  for (const ParsedRow &row : rows)
    if (headDimension <= row.maximumHeadDimension) { matched = &row; break; }
  std::memset(out, 0, sizeof(*out));
  out->maximumHeadDimension = matched->maximumHeadDimension;
  if (!parse_u16(matched->parallelization, &out->parallelization) ||
      !parse_u16(matched->traversal, &out->traversal) || !parse_u16(matched->head, &out->head))
    return fail(MFA_ERR_PARSE, "Could not decode block dimensions.");  // AttentionDescriptor.swift:45
  std::vector<int> operands;
  st = parse_operands(matched->cachedOperands, &operands);
  if (st != MFA_OK) return st;
  for (int op : operands) out->cached[op] = 1;
  return MFA_OK;
}

static mfa_status validate_table(const std::string &text) {
  std::vector<ParsedRow> rows;
  mfa_status st = parse_table(text, &rows);
  if (st != MFA_OK) return st;
  for (const ParsedRow &row : rows) {
    uint16_t tmp;
    if (!parse_u16(row.parallelization, &tmp) || !parse_u16(row.traversal, &tmp) || !parse_u16(row.head, &tmp))
      return fail(MFA_ERR_PARSE, "Could not decode block dimensions.");
    std::vector<int> operands;
    st = parse_operands(row.cachedOperands, &operands);
    if (st != MFA_OK) return st;
  }
  return MFA_OK;
}

// AttentionDescriptor.memoryPrecisions (+Precisions.swift:10-146)
static void memory_precisions(const mfa_attention_descriptor &d, int8_t *out) {
  for (int i = 0; i < MFA_OPERAND_COUNT; ++i) out[i] = -1;
  const bool bf16Inputs = d.lowPrecisionInputType == MFA_BF16;
  if (d.lowPrecisionInputs) {
    out[MFA_Q] = out[MFA_K] = out[MFA_V] = bf16Inputs ? MFA_BF16 : MFA_FP16;  // (:13-16)
    out[MFA_dO] = MFA_BF16;                                                  // (:17)
  } else {
    out[MFA_Q] = out[MFA_K] = out[MFA_V] = out[MFA_dO] = MFA_FP32;            // (:19-22)
  }
  if (d.lowPrecisionIntermediates) {
    out[MFA_L] = MFA_FP16;  // (:82)
    out[MFA_D] = MFA_BF16;  // (:83)
  } else {
    out[MFA_L] = out[MFA_D] = MFA_FP32;  // (:85-86)
  }
  out[MFA_O] = out[MFA_dV] = out[MFA_dK] = out[MFA_dQ] = MFA_FP32;  // (:140-143)
  if (d.lowPrecisionOutputs)   // extension: fused output cast
    out[MFA_O] = out[MFA_dV] = out[MFA_dK] = out[MFA_dQ] = bf16Inputs ? MFA_BF16 : MFA_FP16;
}

// AttentionDescriptor.registerPrecisions (+Precisions.swift:149-215).  gfx950 converts BF16 in
// hardware (v_cvt_pk_bf16_f32), i.e. the "hasNativeBF16Casting" branch.  On this target the map
// is descriptive: 16-bit operands feed the MFMA directly, everything else is held in fp32.
static void register_precisions(const mfa_attention_descriptor &d, int8_t *out) {
  for (int i = 0; i < MFA_OPERAND_COUNT; ++i) out[i] = -1;
  const bool bf16Inputs = d.lowPrecisionInputType == MFA_BF16;
  if (d.lowPrecisionInputs) {
    out[MFA_Q] = out[MFA_K] = out[MFA_V] = bf16Inputs ? MFA_BF16 : MFA_FP16;
    out[MFA_dO] = MFA_BF16;
  } else {
    out[MFA_Q] = out[MFA_K] = out[MFA_V] = out[MFA_dO] = MFA_FP32;
  }
  if (d.lowPrecisionIntermediates) {
    out[MFA_L] = MFA_FP16;
    out[MFA_D] = MFA_BF16;
    out[MFA_S] = d.lowPrecisionInputs ? (bf16Inputs ? MFA_FP32 : MFA_FP16) : MFA_FP32;  // (:197)
    out[MFA_P] = bf16Inputs ? MFA_BF16 : MFA_FP16;                                      // (:198)
    out[MFA_dP] = MFA_FP32;
    out[MFA_dS] = MFA_BF16;
  } else {
    out[MFA_L] = out[MFA_D] = MFA_FP32;
    out[MFA_S] = out[MFA_P] = out[MFA_dP] = out[MFA_dS] = MFA_FP32;
  }
  out[MFA_O] = out[MFA_dV] = out[MFA_dK] = out[MFA_dQ] = MFA_FP32;  // (:209-212)
}

} // namespace mfa

using namespace mfa;

extern "C" {

const char *mfa_last_error_string(void) { return g_last_error.c_str(); }
int mfa_abi_version(void) { return MFA_ABI_VERSION; }

const char *mfa_precision_name(int precision) {
  switch (precision) {
    case MFA_FP32: return "float";
    case MFA_FP16: return "half";
    case MFA_BF16: return "bfloat";
    default: return "";
  }
}
int mfa_precision_size(int precision) {
  switch (precision) {
    case MFA_FP32: return 4;
    case MFA_FP16: return 2;
    case MFA_BF16: return 2;
    default: return 0;
  }
}

const char *mfa_operand_name(int operand) {
  static const char *names[MFA_OPERAND_COUNT] = {"Q", "K", "S", "P", "V", "O", "L", "D",
                                                 "dO", "dV", "dP", "dS", "dK", "dQ"};
  return (operand >= 0 && operand < MFA_OPERAND_COUNT) ? names[operand] : "";
}
int mfa_operand_buffer_binding(int operand) {
  switch (operand) {
    case MFA_Q: return 0;
    case MFA_K: return 1;
    case MFA_V: return 2;
    case MFA_O: return 3;
    case MFA_L: return 4;
    case MFA_D: return 5;
    case MFA_dO: return 6;
    case MFA_dV: return 7;
    case MFA_dK: return 8;
    case MFA_dQ: return 9;
    default: return -1;
  }
}

void mfa_attention_descriptor_init(mfa_attention_descriptor *desc) {
  if (!desc) return;
  std::memset(desc, 0, sizeof(*desc));
  desc->lowPrecisionInputType = MFA_FP16;
}

void mfa_attention_kernel_descriptor_init(mfa_attention_kernel_descriptor *k) {
  if (!k) return;
  std::memset(k, 0, sizeof(*k));
  for (int i = 0; i < MFA_OPERAND_COUNT; ++i)
    k->cacheState[i] = k->memoryPrecisions[i] = k->registerPrecisions[i] = k->transposeState[i] = -1;
  k->preferAsyncCache = k->preferAsyncLoad = k->type = -1;
}

void mfa_launch_params_init(mfa_launch_params *params) {
  if (!params) return;
  std::memset(params, 0, sizeof(*params));
  params->heads = params->batches = 1;
}

mfa_status mfa_attention_descriptor_memory_precisions(const mfa_attention_descriptor *desc, int8_t *out) {
  if (!desc || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  memory_precisions(*desc, out);
  return MFA_OK;
}
mfa_status mfa_attention_descriptor_register_precisions(const mfa_attention_descriptor *desc, int8_t *out) {
  if (!desc || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  register_precisions(*desc, out);
  return MFA_OK;
}

mfa_status mfa_parameter_table_get(int type, int mixed, char *out, size_t capacity) {
  if (type < 0 || type > 2 || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "bad kernel type or null buffer");
  std::lock_guard<std::mutex> lock(g_table_mutex);
  ensure_tables();
  const std::string &text = g_tables[type][mixed ? 1 : 0];
  if (text.size() + 1 > capacity) return fail(MFA_ERR_INVALID_ARGUMENT, "buffer too small");
  std::memcpy(out, text.c_str(), text.size() + 1);
  return MFA_OK;
}

mfa_status mfa_parameter_table_set(int type, int mixed, const char *text) {
  if (type < 0 || type > 2 || !text) return fail(MFA_ERR_INVALID_ARGUMENT, "bad kernel type or null text");
  mfa_status st = validate_table(text);
  if (st != MFA_OK) return st;
  std::lock_guard<std::mutex> lock(g_table_mutex);
  ensure_tables();
  g_tables[type][mixed ? 1 : 0] = text;
  return MFA_OK;
}

mfa_status mfa_parameter_table_reset(void) {
  std::lock_guard<std::mutex> lock(g_table_mutex);
  g_tables_initialised = false;
  ensure_tables();
  return MFA_OK;
}

mfa_status mfa_parameter_table_select(const char *text, uint16_t headDimension, mfa_parameter_row *out) {
  if (!text || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  return select_row(text, headDimension, out);
}

// AttentionDescriptor.kernelDescriptor(type:) -- AttentionDescriptor.swift:33-130
mfa_status mfa_attention_descriptor_kernel_descriptor(const mfa_attention_descriptor *desc, int type,
                                                      mfa_attention_kernel_descriptor *out) {
  if (!desc || !out) return fail(MFA_ERR_INVALID_ARGUMENT, "null argument");
  if (type < 0 || type > 2) return fail(MFA_ERR_INVALID_ARGUMENT, "unknown kernel type");
  if (!desc->hasMatrixDimensions || !desc->hasTransposeState)  // (:88-98)
    return fail(MFA_ERR_INCOMPLETE_DESCRIPTOR, "Descriptor was incomplete.");
  if (desc->row == 0 || desc->column == 0 || desc->head == 0)
    return fail(MFA_ERR_INVALID_ARGUMENT, "matrixDimensions must be non-zero");

  // parameterFile(type:).  The reference takes its mixed tables only when BOTH flags are set (+Parameters.swift:16):
  // there the tables follow the REGISTER footprint, which FP16 intermediates halve.  On gfx950 S, P and every
  // accumulator are fp32 registers whatever the flags say (MFMA results), and what changes the kernel -- 16-bit
  // matrix-core code objects instead of the fp32-arithmetic ones -- is the storage type of Q, K, V: the mixed tables
  // list the block dimensions of the 16-bit code objects, so they are taken whenever the inputs are 16-bit.
  std::string file;
  {
    std::lock_guard<std::mutex> lock(g_table_mutex);
    ensure_tables();
    const int mixed = desc->lowPrecisionInputs ? 1 : 0;
    file = g_tables[type][mixed];
  }
  mfa_parameter_row row;
  mfa_status st = select_row(file, desc->head, &row);
  if (st != MFA_OK) return st;

  mfa_attention_kernel_descriptor_init(out);
  // createBlockDimensions (:41-54): head block clamped to the head dimension padded to 8
  const uint16_t paddedHeadDimension = (uint16_t)((desc->head + 7) / 8 * 8);
  out->hasBlockDimensions = 1;
  out->parallelization = row.parallelization;
  out->traversal = row.traversal;
  out->headBlock = std::min(row.head, paddedHeadDimension);

  // createCacheState (:56-86)
  int expected[4];
  int expectedCount = 0;
  switch (type) {
    case MFA_FORWARD: expected[0] = MFA_Q; expected[1] = MFA_O; expectedCount = 2; break;
    case MFA_BACKWARD_QUERY: expected[0] = MFA_Q; expected[1] = MFA_dO; expected[2] = MFA_dQ; expectedCount = 3; break;
    default: expected[0] = MFA_K; expected[1] = MFA_V; expected[2] = MFA_dV; expected[3] = MFA_dK; expectedCount = 4; break;
  }
  for (int op = 0; op < MFA_OPERAND_COUNT; ++op) {
    if (!row.cached[op]) continue;
    bool ok = false;
    for (int i = 0; i < expectedCount; ++i) ok |= (expected[i] == op);
    if (!ok) return fail(MFA_ERR_INVALID_ARGUMENT, std::string("Unexpected operand: ") + mfa_operand_name(op));
  }
  for (int i = 0; i < expectedCount; ++i) out->cacheState[expected[i]] = row.cached[expected[i]] ? 1 : 0;

  out->hasHeadDimension = 1;
  out->headDimension = desc->head;
  memory_precisions(*desc, out->memoryPrecisions);
  register_precisions(*desc, out->registerPrecisions);
  // preferAsyncCache / preferAsyncLoad (:118-124) select, in the reference, between direct
  // device access and the threadgroup-memory async copy.  On gfx950 operands shared by the waves
  // of a workgroup are always staged through LDS (the "async load" role) and per-lane operands
  // never are, i.e. the Apple7/8 branch.
  out->preferAsyncCache = 0;
  out->preferAsyncLoad = 1;

  // createTransposeState (:88-113): gradients inherit the transpose state of their primal
  out->transposeState[MFA_Q] = desc->transposeQ ? 1 : 0;
  out->transposeState[MFA_K] = desc->transposeK ? 1 : 0;
  out->transposeState[MFA_V] = desc->transposeV ? 1 : 0;
  out->transposeState[MFA_O] = desc->transposeO ? 1 : 0;
  out->transposeState[MFA_dO] = out->transposeState[MFA_O];
  out->transposeState[MFA_dV] = out->transposeState[MFA_V];
  out->transposeState[MFA_dK] = out->transposeState[MFA_K];
  out->transposeState[MFA_dQ] = out->transposeState[MFA_Q];
  out->type = (int8_t)type;
  return MFA_OK;
}

} // extern "C"
