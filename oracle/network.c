/*
 * oracle/network.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99) of the reference's naive attention "Network":
 *   /root/reference/Tests/FlashAttentionTests/Utilities/Network.swift:70-402
 * plus the dtype round-trips of the reference's buffer helpers:
 *   /root/reference/Tests/FlashAttentionTests/Utilities/MTLContext+Buffers.swift:5-78
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this file.  Nothing under metal_flash_attention_amd/ links or imports it.
 *
 * PARITY STATUS: "parity unpinned" against reference OUTPUTS.  The reference
 * ships no golden vectors for this path (its inputs are unseeded random,
 * SURVEY.md 8c) and no Swift toolchain exists in this image, so this
 * restatement cannot be checked against vectors or outputs of the reference.
 * What anchors it instead follows from the reference's FORMULAS alone
 * (Network.swift:134-402), not from any code of this repository:
 * (i) closed forms (tests/golden/closed_form.py: uniform and one-hot P, C = 1,
 * the C = 2 logistic form with its analytic dS), (ii) metamorphic relations on
 * RANDOM inputs (key shift K + 1 u^T, value shift V + 1 w^T, (aQ, K/a)
 * rescaling, key permutation; tests/test_closed_form.py), (iii) an independent
 * numpy fp64 twin (oracle/network_np.py), central finite differences of the
 * loss (idea: Documentation/Archive/FiniteDifferencingTest.swift:86-131) and
 * torch fp64 autograd (tests/test_oracle.py).  DESIGN.md 8 says the same.
 *
 * Arithmetic contract: every scalar sum below is accumulated in fp32 in the
 * same ORDER as the Swift loops (sequential over d for dot products,
 * sequential over columns for row sums, sequential over rows for dV/dK).  Loops
 * are re-nested only where that keeps each output element's operation sequence
 * unchanged (so the compiler can vectorise across independent outputs).  Build
 * with -ffp-contract=off and without -ffast-math: Swift does not fuse a*b+c.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* Deterministic inputs.  Network.swift:96-129 draws pairs from an unseeded   */
/* system RNG; we keep the Box-Muller structure (Q/dO share a pair, K/V share */
/* a pair) and replace the RNG with a seeded splitmix64 stream.              */
/* ------------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t *state) {
  uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* uniform in [0,1) with 24 random bits, like Float.random(in: 0..<1) */
static float uniform01(uint64_t *state) {
  return (float)(splitmix64(state) >> 40) * (1.0f / 16777216.0f);
}

/* Network.swift:115-129 */
static void box_muller(uint64_t *state, float out[2]) {
  float u0 = uniform01(state);
  float u1 = uniform01(state);
  if (u0 == 0.0f) u0 = 1.0f / 16777216.0f; /* log(0) guard; Swift would give inf */
  float logPart = logf(u0);
  float magnitudePart = sqrtf(-2.0f * logPart);
  float anglePart = 2.0f * 3.14159265358979323846f * u1;
  out[0] = magnitudePart * cosf(anglePart);
  out[1] = magnitudePart * sinf(anglePart);
}

/* Network.swift:96-113: fills Q,dO [R*D] and K,V [C*D], row-major. */
void oracle_network_init(uint64_t seed, int R, int C, int D,
                         float *Q, float *K, float *V, float *dO) {
  uint64_t state = seed * 0xD1342543DE82EF95ull + 0x1234567ull;
  float pair[2];
  for (int r = 0; r < R; ++r)
    for (int d = 0; d < D; ++d) {
      box_muller(&state, pair);
      Q[(size_t)r * D + d] = pair[0];
      dO[(size_t)r * D + d] = pair[1];
    }
  for (int c = 0; c < C; ++c)
    for (int d = 0; d < D; ++d) {
      box_muller(&state, pair);
      K[(size_t)c * D + d] = pair[0];
      V[(size_t)c * D + d] = pair[1];
    }
}

/* ------------------------------------------------------------------------ */
/* dtype round trips, MTLContext+Buffers.swift:29-42 (pack) and :60-75 (copy) */
/* ------------------------------------------------------------------------ */
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* BF16 = upper 16 bits, TRUNCATION (MTLContext+Buffers.swift:36-42). */
uint16_t oracle_f32_to_bf16_trunc(float f) { return (uint16_t)(f2u(f) >> 16); }
float oracle_bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }

/* FP16 = Float16(x): IEEE round-to-nearest-even (MTLContext+Buffers.swift:31-34). */
uint16_t oracle_f32_to_f16_rne(float f) {
  uint32_t x = f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t absx = x & 0x7FFFFFFFu;
  if (absx >= 0x7F800000u) /* inf / nan */
    return (uint16_t)(sign | 0x7C00u | ((absx > 0x7F800000u) ? 0x200u : 0));
  if (absx >= 0x477FF000u) /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7C00u);
  if (absx < 0x38800000u) { /* subnormal half or zero (< 2^-14) */
    if (absx < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
    int e = (int)(absx >> 23);                       /* biased exponent */
    uint32_t mant = (absx & 0x7FFFFFu) | 0x800000u;
    int shift = 126 - e;                             /* 14..24 */
    uint32_t half = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half & 1u))) half++;
    return (uint16_t)(sign | half);
  }
  uint32_t e = ((absx >> 23) - 112u) << 10;
  uint32_t m = (absx >> 13) & 0x3FFu;
  uint32_t rem = absx & 0x1FFFu;
  uint32_t h = e | m;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}

float oracle_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1Fu;
  uint32_t m = h & 0x3FFu;
  if (e == 0) {
    if (m == 0) return u2f(sign);
    float v = (float)m * (1.0f / 16777216.0f); /* m * 2^-24 */
    return (sign ? -v : v);
  }
  if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
  return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* precision codes follow GEMMOperandPrecision.swift:33-36: FP32=0 FP16=1 BF16=2 */
void oracle_round_trip(float *x, size_t n, int precision) {
  if (precision == 1)
    for (size_t i = 0; i < n; ++i) x[i] = oracle_f16_to_f32(oracle_f32_to_f16_rne(x[i]));
  else if (precision == 2)
    for (size_t i = 0; i < n; ++i) x[i] = oracle_bf16_to_f32(oracle_f32_to_bf16_trunc(x[i]));
}

/* ------------------------------------------------------------------------ */
/* The Network.  All outputs optional (NULL = skip).                         */
/*   O  [R*D]  inferenceAttention           Network.swift:286-311             */
/*   L  [R]    createLTerm (natural log)    Network.swift:181-203             */
/*   Dt [R]    createDTerm (unscaled)       Network.swift:259-281             */
/*   dV [C*D]  derivativeV                  Network.swift:329-349             */
/*   dK [C*D]  derivativeK                  Network.swift:352-372             */
/*   dQ [R*D]  derivativeQ                  Network.swift:375-402             */
/* ------------------------------------------------------------------------ */
#define ROW_CHUNK 128

/* `causal` is this project's extension (the reference is unmasked, README.md:7 names masks as the first
   extension): row r may attend column c iff c <= r + max(C - R, 0).  Masked columns are simply left out of
   every sum (maximum, exp-sum, P V, dS), which is what P = 0 there means.  causal = 0 reproduces
   Network.swift exactly. */
int oracle_network_run_masked(int R, int C, int D,
                       const float *Q, const float *K, const float *V, const float *dO,
                       float *O, float *L, float *Dt, float *dV, float *dK, float *dQ,
                       int num_threads, int causal);

int oracle_network_run(int R, int C, int D,
                       const float *Q, const float *K, const float *V, const float *dO,
                       float *O, float *L, float *Dt, float *dV, float *dK, float *dQ,
                       int num_threads) {
  return oracle_network_run_masked(R, C, D, Q, K, V, dO, O, L, Dt, dV, dK, dQ, num_threads, 0);
}

int oracle_network_run_masked(int R, int C, int D,
                       const float *Q, const float *K, const float *V, const float *dO,
                       float *O, float *L, float *Dt, float *dV, float *dK, float *dQ,
                       int num_threads, int causal) {
  const int need_bwd = (Dt || dV || dK || dQ) ? 1 : 0;
  if (need_bwd && !dO) return -1;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
  const float scaleFactor = 1.0f / sqrtf((float)D); /* 1 / Float(headDimension).squareRoot() */

  /* transposed copies so that the per-column dot products (sequential over d)
     vectorise across columns without changing any element's sum order */
  float *Kt = (float *)malloc((size_t)C * D * sizeof(float));
  float *Vt = need_bwd ? (float *)malloc((size_t)C * D * sizeof(float)) : NULL;
  float *Pc = (float *)malloc((size_t)ROW_CHUNK * C * sizeof(float));
  float *dSc = need_bwd ? (float *)malloc((size_t)ROW_CHUNK * C * sizeof(float)) : NULL;
  if (!Kt || !Pc || (need_bwd && (!Vt || !dSc))) { free(Kt); free(Vt); free(Pc); free(dSc); return -2; }
  for (int c = 0; c < C; ++c)
    for (int d = 0; d < D; ++d) {
      Kt[(size_t)d * C + c] = K[(size_t)c * D + d];
      if (Vt) Vt[(size_t)d * C + c] = V[(size_t)c * D + d];
    }
  if (dV) memset(dV, 0, (size_t)C * D * sizeof(float));
  if (dK) memset(dK, 0, (size_t)C * D * sizeof(float));

  for (int r0 = 0; r0 < R; r0 += ROW_CHUNK) {
    const int rows = (R - r0 < ROW_CHUNK) ? (R - r0) : ROW_CHUNK;
#pragma omp parallel
    {
      float *orow = (float *)malloc((size_t)D * sizeof(float));
      float *dprow = need_bwd ? (float *)malloc((size_t)C * sizeof(float)) : NULL;
#pragma omp for schedule(dynamic, 4)
      for (int rr = 0; rr < rows; ++rr) {
        const int rowID = r0 + rr;
        float *p = Pc + (size_t)rr * C;
        /* number of visible columns of this row: all of them, or c <= rowID + max(C - R, 0) (the offset is clamped at 0 when
           C < R -- only per-batch lengths can produce that, include/mfa.h rowLengths -- so every row sees at least one column) */
        int CV = C;
        if (causal) { CV = rowID + (C > R ? C - R : 0) + 1; if (CV > C) CV = C; }
        /* createMatrixSRow, Network.swift:134-149 : dot over d, sequential */
        for (int c = 0; c < C; ++c) p[c] = 0.0f;
        for (int d = 0; d < D; ++d) {
          const float q = Q[(size_t)rowID * D + d];
          const float *kt = Kt + (size_t)d * C;
          for (int c = 0; c < C; ++c) p[c] += q * kt[c];
        }
        /* createMatrixPRow, Network.swift:151-179 */
        float maximum = -FLT_MAX;
        for (int c = 0; c < CV; ++c) {
          float value = scaleFactor * p[c];
          maximum = fmaxf(maximum, value);
        }
        float sum = 0.0f;
        for (int c = 0; c < CV; ++c) {
          float value = scaleFactor * p[c];
          sum += expf(value - maximum);
        }
        const float lse = maximum + logf(sum);
        for (int c = 0; c < CV; ++c) {
          float value = scaleFactor * p[c];
          p[c] = expf(value - lse);
        }
        for (int c = CV; c < C; ++c) p[c] = 0.0f;   /* masked: P = 0 */
        if (L) L[rowID] = lse; /* createLTerm, Network.swift:181-203 (same arithmetic) */

        /* P * V, Network.swift:292-303 : sum over columns, sequential per d */
        for (int d = 0; d < D; ++d) orow[d] = 0.0f;
        for (int c = 0; c < CV; ++c) {
          const float valueP = p[c];
          const float *v = V + (size_t)c * D;
          for (int d = 0; d < D; ++d) orow[d] += valueP * v[d];
        }
        if (O) for (int d = 0; d < D; ++d) O[(size_t)rowID * D + d] = orow[d];
        if (!need_bwd) continue;

        /* createDTerm, Network.swift:259-281 */
        float termD = 0.0f;
        for (int d = 0; d < D; ++d) termD += orow[d] * dO[(size_t)rowID * D + d];
        if (Dt) Dt[rowID] = termD;
        if (!(dK || dQ)) continue;

        /* createDerivativePRow, Network.swift:205-218 */
        for (int c = 0; c < C; ++c) dprow[c] = 0.0f;
        for (int d = 0; d < D; ++d) {
          const float g = dO[(size_t)rowID * D + d];
          const float *vt = Vt + (size_t)d * C;
          for (int c = 0; c < C; ++c) dprow[c] += g * vt[c];
        }
        /* createDerivativeSRow, Network.swift:245-255 */
        float *ds = dSc + (size_t)rr * C;
        for (int c = 0; c < C; ++c) {
          float valueS = p[c] * (dprow[c] - termD);
          valueS *= scaleFactor;
          ds[c] = valueS;
        }
        /* derivativeQ, Network.swift:383-392 : sum over columns, sequential per d */
        if (dQ) {
          for (int d = 0; d < D; ++d) orow[d] = 0.0f;
          for (int c = 0; c < CV; ++c) {
            const float s = ds[c];
            const float *k = K + (size_t)c * D;
            for (int d = 0; d < D; ++d) orow[d] += s * k[d];
          }
          for (int d = 0; d < D; ++d) dQ[(size_t)rowID * D + d] = orow[d];
        }
      }
      free(orow);
      free(dprow);
    }
    /* derivativeV / derivativeK, Network.swift:334-347, :357-370: accumulate over
       rows in increasing rowID order for every (column, d) */
    if (dV || dK) {
#pragma omp parallel for schedule(static)
      for (int c = 0; c < C; ++c) {
        for (int rr = 0; rr < rows; ++rr) {
          const int rowID = r0 + rr;
          if (dV) {
            const float pv = Pc[(size_t)rr * C + c];
            float *out = dV + (size_t)c * D;
            const float *g = dO + (size_t)rowID * D;
            for (int d = 0; d < D; ++d) out[d] += pv * g[d];
          }
          if (dK) {
            const float sv = dSc[(size_t)rr * C + c];
            float *out = dK + (size_t)c * D;
            const float *q = Q + (size_t)rowID * D;
            for (int d = 0; d < D; ++d) out[d] += sv * q[d];
          }
        }
      }
    }
  }
  free(Kt); free(Vt); free(Pc); free(dSc);
  return 0;
}

/* loss, Network.swift:314-326:  sum_n sum_d dO[n][d] * O[n][d]  (fp32, sequential) */
float oracle_network_loss(int R, int C, int D, const float *Q, const float *K,
                          const float *V, const float *dO) {
  float *O = (float *)malloc((size_t)R * D * sizeof(float));
  oracle_network_run(R, C, D, Q, K, V, NULL, O, NULL, NULL, NULL, NULL, NULL, 0);
  float output = 0.0f;
  for (int r = 0; r < R; ++r)
    for (int d = 0; d < D; ++d) output += dO[(size_t)r * D + d] * O[(size_t)r * D + d];
  free(O);
  return output;
}

/* ------------------------------------------------------------------------ */
/* fp64 twin of the same formulas (error budgeting, finite differences).     */
/* ------------------------------------------------------------------------ */
int oracle_network_run_f64(int R, int C, int D,
                           const double *Q, const double *K, const double *V, const double *dO,
                           double *O, double *L, double *Dt, double *dV, double *dK, double *dQ) {
  const double scaleFactor = 1.0 / sqrt((double)D);
  double *p = (double *)malloc((size_t)C * sizeof(double));
  double *ds = (double *)malloc((size_t)C * sizeof(double));
  double *orow = (double *)malloc((size_t)D * sizeof(double));
  if (dV) memset(dV, 0, (size_t)C * D * sizeof(double));
  if (dK) memset(dK, 0, (size_t)C * D * sizeof(double));
  for (int r = 0; r < R; ++r) {
    double maximum = -DBL_MAX;
    for (int c = 0; c < C; ++c) {
      double dot = 0;
      for (int d = 0; d < D; ++d) dot += Q[(size_t)r * D + d] * K[(size_t)c * D + d];
      p[c] = scaleFactor * dot;
      if (p[c] > maximum) maximum = p[c];
    }
    double sum = 0;
    for (int c = 0; c < C; ++c) sum += exp(p[c] - maximum);
    const double lse = maximum + log(sum);
    for (int c = 0; c < C; ++c) p[c] = exp(p[c] - lse);
    if (L) L[r] = lse;
    for (int d = 0; d < D; ++d) {
      double dot = 0;
      for (int c = 0; c < C; ++c) dot += p[c] * V[(size_t)c * D + d];
      orow[d] = dot;
      if (O) O[(size_t)r * D + d] = dot;
    }
    if (!dO) continue;
    double termD = 0;
    for (int d = 0; d < D; ++d) termD += orow[d] * dO[(size_t)r * D + d];
    if (Dt) Dt[r] = termD;
    for (int c = 0; c < C; ++c) {
      double dp = 0;
      for (int d = 0; d < D; ++d) dp += dO[(size_t)r * D + d] * V[(size_t)c * D + d];
      ds[c] = p[c] * (dp - termD) * scaleFactor;
    }
    for (int c = 0; c < C; ++c)
      for (int d = 0; d < D; ++d) {
        if (dV) dV[(size_t)c * D + d] += p[c] * dO[(size_t)r * D + d];
        if (dK) dK[(size_t)c * D + d] += ds[c] * Q[(size_t)r * D + d];
      }
    if (dQ)
      for (int d = 0; d < D; ++d) {
        double dot = 0;
        for (int c = 0; c < C; ++c) dot += ds[c] * K[(size_t)c * D + d];
        dQ[(size_t)r * D + d] = dot;
      }
  }
  free(p); free(ds); free(orow);
  return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
