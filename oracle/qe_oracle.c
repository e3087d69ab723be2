/*
 * CPU oracle — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the arithmetic the reference's physical plan performs
 * through arrow2 0.10.1 (third-party crate, not vendored in /root/reference:
 * query-engine/Cargo.toml:9-16).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this file; the product path
 * (query-engine_amd/) never does.
 *
 * PARITY STATUS: pinned by the reference's own fixture for the filter path
 * (tests/mod.rs:138-188: Utf8 == scalar predicate, stable all-column
 * compaction) and column pass-through (tests/mod.rs:86-136).  Arithmetic,
 * numeric comparison, hashing/grouping and min/max values are
 * "parity unpinned": no reference test touches them and the reference cannot
 * be built here (no Rust toolchain, no arrow2 sources); they follow the Arrow
 * columnar specification and the published arrow2 semantics.
 *
 * Layout: Arrow.  Bitmaps are LSB-first, 1 = valid / true.  A NULL validity
 * pointer means "all valid".
 *
 * Each function names the reference call site it restates
 * (paths relative to /root/reference/query-engine/src/).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GET_BIT(bm, i) (((bm)[(i) >> 3] >> ((i) & 7)) & 1)
#define SET_BIT(bm, i) ((bm)[(i) >> 3] |= (uint8_t)(1u << ((i) & 7)))

enum { CMP_EQ = 1, CMP_NEQ = 2, CMP_GT = 3, CMP_GTEQ = 4, CMP_LT = 5, CMP_LTEQ = 6 };
enum { AR_ADD = 9, AR_SUB = 10, AR_MUL = 11, AR_DIV = 12, AR_MOD = 13 };

/* ---- bitmap helpers (arrow2 Bitmap::bitand, physical_expressions.rs:381) */
void orc_bitmap_and(const uint8_t *a, const uint8_t *b, int64_t n, uint8_t *out) {
  int64_t nb = (n + 7) / 8;
  for (int64_t i = 0; i < nb; ++i) out[i] = a[i] & b[i];
}
void orc_bitmap_or(const uint8_t *a, const uint8_t *b, int64_t n, uint8_t *out) {
  int64_t nb = (n + 7) / 8;
  for (int64_t i = 0; i < nb; ++i) out[i] = a[i] | b[i];
}
int64_t orc_bitmap_count(const uint8_t *a, int64_t n) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i) c += GET_BIT(a, i);
  return c;
}

/* ---- K1 / K2: compute::comparison::{eq,neq}[_scalar] (physical_expressions.rs:184-202)
 * plus the ordered comparisons the reference only has logically.  The values
 * bitmap is written for every row (null rows included, like arrow2, which
 * compares the value buffers and attaches the combined validity separately). */
#define DEF_CMP(NAME, T)                                                                        \
  void NAME(int op, const T *a, const T *b, int64_t n, int b_is_scalar, uint8_t *out) {         \
    memset(out, 0, (size_t)((n + 7) / 8));                                                      \
    for (int64_t i = 0; i < n; ++i) {                                                           \
      T x = a[i], y = b_is_scalar ? b[0] : b[i];                                                \
      int r = 0;                                                                                \
      switch (op) {                                                                             \
        case CMP_EQ: r = x == y; break;                                                         \
        case CMP_NEQ: r = x != y; break;                                                        \
        case CMP_GT: r = x > y; break;                                                          \
        case CMP_GTEQ: r = x >= y; break;                                                       \
        case CMP_LT: r = x < y; break;                                                          \
        default: r = x <= y; break;                                                             \
      }                                                                                         \
      if (r) SET_BIT(out, i);                                                                   \
    }                                                                                           \
  }
DEF_CMP(orc_cmp_i32, int32_t)
DEF_CMP(orc_cmp_i64, int64_t)
DEF_CMP(orc_cmp_f64, double)

/* Utf8 eq / neq against a scalar or another array (test_filter: tests/mod.rs:141). */
void orc_cmp_utf8(int op, const int32_t *ao, const uint8_t *ab, const int32_t *bo, const uint8_t *bb, int64_t n,
                  int b_is_scalar, uint8_t *out) {
  memset(out, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; ++i) {
    int32_t la = ao[i + 1] - ao[i];
    int64_t j = b_is_scalar ? 0 : i;
    int32_t lb = bo[j + 1] - bo[j];
    int eq = la == lb && memcmp(ab + ao[i], bb + bo[j], (size_t)la) == 0;
    if ((op == CMP_EQ) ? eq : !eq) SET_BIT(out, i);
  }
}

/* ---- K3 / K4: compute::arithmetics::{add,sub,mul,div}[_scalar] (physical_expressions.rs:249-268).
 * Integers wrap (Rust release arithmetic).  Division / remainder by zero and MIN / -1 panic in
 * the reference; here they clear out_valid[i] (which the caller pre-fills with the AND of the
 * operand validities) and store 0. */
#define DEF_ARITH_INT(NAME, T, UT, TMIN)                                                        \
  void NAME(int op, const T *a, const T *b, int64_t n, int b_is_scalar, T *out, uint8_t *out_valid) { \
    for (int64_t i = 0; i < n; ++i) {                                                           \
      T x = a[i], y = b_is_scalar ? b[0] : b[i];                                                \
      T r = 0;                                                                                  \
      switch (op) {                                                                             \
        case AR_ADD: r = (T)((UT)x + (UT)y); break;                                             \
        case AR_SUB: r = (T)((UT)x - (UT)y); break;                                             \
        case AR_MUL: r = (T)((UT)x * (UT)y); break;                                             \
        default:                                                                                \
          if (y == 0 || (x == TMIN && y == -1)) {                                               \
            if (out_valid) out_valid[i >> 3] &= (uint8_t) ~(1u << (i & 7));                     \
            r = 0;                                                                              \
          } else {                                                                              \
            r = op == AR_DIV ? x / y : x % y;                                                   \
          }                                                                                     \
      }                                                                                         \
      out[i] = r;                                                                               \
    }                                                                                           \
  }
DEF_ARITH_INT(orc_arith_i32, int32_t, uint32_t, INT32_MIN)
DEF_ARITH_INT(orc_arith_i64, int64_t, uint64_t, INT64_MIN)

/* float64: one IEEE round-to-nearest operation per node, never fused (the reference runs mul
 * and add as two separate array passes).  Built with -ffp-contract=off. */
void orc_arith_f64(int op, const double *a, const double *b, int64_t n, int b_is_scalar, double *out) {
  for (int64_t i = 0; i < n; ++i) {
    double x = a[i], y = b_is_scalar ? b[0] : b[i];
    switch (op) {
      case AR_ADD: out[i] = x + y; break;
      case AR_SUB: out[i] = x - y; break;
      case AR_MUL: out[i] = x * y; break;
      default: out[i] = x / y; break;
    }
  }
}

/* ---- K5: compute::filter::filter_chunk (physical_plan/mod.rs:213-220): stable compaction of
 * one column by a boolean mask.  A row is kept iff its mask value is true AND valid (a null
 * predicate drops the row).  Returns the number of rows kept. */
int64_t orc_filter_count(const uint8_t *mask_values, const uint8_t *mask_valid, int64_t n) {
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) m += GET_BIT(mask_values, i) && (!mask_valid || GET_BIT(mask_valid, i));
  return m;
}
int64_t orc_filter_fixed(int width, const void *values, const uint8_t *valid, int64_t n, const uint8_t *mask_values,
                         const uint8_t *mask_valid, void *out, uint8_t *out_valid) {
  int64_t m = 0;
  const uint8_t *src = (const uint8_t *)values;
  uint8_t *dst = (uint8_t *)out;
  for (int64_t i = 0; i < n; ++i) {
    if (GET_BIT(mask_values, i) && (!mask_valid || GET_BIT(mask_valid, i))) {
      memcpy(dst + m * width, src + i * width, (size_t)width);
      if (out_valid && (!valid || GET_BIT(valid, i))) SET_BIT(out_valid, m);
      ++m;
    }
  }
  return m;
}
int64_t orc_filter_bitmap(const uint8_t *values, const uint8_t *valid, int64_t n, const uint8_t *mask_values,
                          const uint8_t *mask_valid, uint8_t *out, uint8_t *out_valid) {
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (GET_BIT(mask_values, i) && (!mask_valid || GET_BIT(mask_valid, i))) {
      if (GET_BIT(values, i)) SET_BIT(out, m);
      if (out_valid && (!valid || GET_BIT(valid, i))) SET_BIT(out_valid, m);
      ++m;
    }
  }
  return m;
}
int64_t orc_filter_utf8(const int32_t *offsets, const uint8_t *bytes, const uint8_t *valid, int64_t n,
                        const uint8_t *mask_values, const uint8_t *mask_valid, int32_t *out_offsets, uint8_t *out_bytes,
                        uint8_t *out_valid) {
  int64_t m = 0;
  int32_t pos = 0;
  out_offsets[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (GET_BIT(mask_values, i) && (!mask_valid || GET_BIT(mask_valid, i))) {
      int32_t len = offsets[i + 1] - offsets[i];
      memcpy(out_bytes + pos, bytes + offsets[i], (size_t)len);
      pos += len;
      if (out_valid && (!valid || GET_BIT(valid, i))) SET_BIT(out_valid, m);
      ++m;
      out_offsets[m] = pos;
    }
  }
  return m;
}

/* ---- hashing (K6: compute::hash::hash + ArrayAdd::add, physical_plan/mod.rs:310-320).
 * arrow2 hashes with ahash under fixed seeds; those values cannot be reproduced here, and
 * only the equality structure of the hash sums matters to the result, so a 64-bit mixer
 * stands in: row hash = H(0) + sum_j H(key_j[i]) with wrapping adds, exactly the reference's
 * combine (commutative across key columns). */
static uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}
void orc_hash_init(int64_t n, uint64_t *hashes) {
  uint64_t h0 = mix64(0);
  for (int64_t i = 0; i < n; ++i) hashes[i] = h0;
}
void orc_hash_add_i64(const int64_t *keys, int64_t n, uint64_t *hashes) {
  for (int64_t i = 0; i < n; ++i) hashes[i] += mix64((uint64_t)keys[i]);
}

/* Order-preserving image used to state min / max once for every numeric type:
 * -0.0 < +0.0, NaN is skipped by the callers. */
static uint64_t ord_image_f64(double d) {
  uint64_t b;
  memcpy(&b, &d, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static double ord_inverse_f64(uint64_t img) {
  uint64_t b = (img >> 63) ? (img & 0x7fffffffffffffffull) : ~img;
  double d;
  memcpy(&d, &b, 8);
  return d;
}

enum { AGG_MAX = 1, AGG_MIN = 2, AGG_SUM = 3, AGG_COUNT = 4, AGG_AVG = 5 };
enum { VT_I32 = 1, VT_I64 = 2, VT_F64 = 3 };

/* ---- AggregateExec::execute restated literally (physical_plan/mod.rs:298-374):
 * per batch, for every row whose hash is first seen IN THIS BATCH, build the full-batch
 * membership mask `hashes == key` (K7, :333-342) and fold the masked batch into the group's
 * accumulator (K8, physical_expressions.rs:380-389, :411-463): state replaced when
 * `new > state` (max) / `new < state` (min) or when the state is still Null.
 * Group identity is the 64-bit hash only (:343).  Rows whose hash is null (hash_valid bit 0)
 * are skipped (:367).  Cost is O(rows x groups-in-batch): small inputs only.
 *
 * One call = one batch; state persists in the caller's arrays between calls:
 *   g_hash[G_max], g_first_row[G_max] (row index inside its batch of the key slice, :350-354),
 *   g_first_batch[G_max], state[G_max] (ord image), state_set[G_max].
 * Returns the new group count, or -1 if G_max is exceeded. */
int64_t orc_aggregate_reference_batch(const uint64_t *hashes, const uint8_t *hash_valid, int64_t n, int agg_kind,
                                      const double *values, const uint8_t *values_valid, int64_t batch_index,
                                      uint64_t *g_hash, int64_t *g_first_row, int64_t *g_first_batch, uint64_t *state,
                                      uint8_t *state_set, int64_t n_groups, int64_t g_max) {
  uint8_t *seen_in_batch = (uint8_t *)calloc((size_t)(n ? n : 1), 1);
  for (int64_t i = 0; i < n; ++i) {
    if (hash_valid && !GET_BIT(hash_valid, i)) continue;
    if (seen_in_batch[i]) continue;
    uint64_t key = hashes[i];
    /* find / create the global group */
    int64_t g = -1;
    for (int64_t k = 0; k < n_groups; ++k)
      if (g_hash[k] == key) { g = k; break; }
    if (g < 0) {
      if (n_groups >= g_max) { free(seen_in_batch); return -1; }
      g = n_groups++;
      g_hash[g] = key;
      g_first_row[g] = i;
      g_first_batch[g] = batch_index;
      state_set[g] = 0;
      state[g] = 0;
    }
    /* masked full-batch reduction */
    int have = 0;
    uint64_t best = 0;
    for (int64_t r = i; r < n; ++r) {
      if (hashes[r] != key) continue;  /* eq_scalar values, validity ignored (values_iter) */
      seen_in_batch[r] = 1;
      if (values_valid && !GET_BIT(values_valid, r)) continue;
      double v = values[r];
      if (v != v) continue; /* arrow2 min/max skip NaN */
      uint64_t img = ord_image_f64(v);
      if (!have || (agg_kind == AGG_MAX ? img > best : img < best)) { best = img; have = 1; }
    }
    if (have) {
      if (!state_set[g] || (agg_kind == AGG_MAX ? best > state[g] : best < state[g])) {
        state[g] = best;
        state_set[g] = 1;
      }
    }
  }
  free(seen_in_batch);
  return n_groups;
}
double orc_ord_inverse_f64(uint64_t img) { return ord_inverse_f64(img); }

/* ---- conventional single-pass hash aggregate (the CPU baseline for large inputs and the
 * checker for SUM / COUNT / AVG, which the reference only has logically:
 * logical_expression.rs:386-423).  Keys are compared in full.  Groups come out in
 * first-seen order.  `values` is float64 or int64 (`value_type`); rows with key_valid == 0
 * are dropped; rows with value_valid == 0 do not contribute to sum/min/max/count(x).
 * SUM over float64 accumulates in long double so the checker itself carries no
 * order-dependent rounding at the 1e-12 level. */
typedef struct {
  uint64_t *slot_key;
  int64_t *slot_group;
  uint64_t mask;
} orc_table;

static int64_t table_find_or_insert(orc_table *t, uint64_t key, int64_t *n_groups) {
  uint64_t s = mix64(key) & t->mask;
  for (;;) {
    if (t->slot_group[s] < 0) {
      t->slot_group[s] = (*n_groups)++;
      t->slot_key[s] = key;
      return t->slot_group[s];
    }
    if (t->slot_key[s] == key) return t->slot_group[s];
    s = (s + 1) & t->mask;
  }
}

int64_t orc_hash_aggregate(const uint64_t *keys, const uint8_t *key_valid, const uint8_t *row_mask, int64_t n,
                           int value_type, const void *values, const uint8_t *value_valid, int64_t g_max,
                           uint64_t *out_keys, int64_t *out_rows, int64_t *out_nn, double *out_sum_f64,
                           int64_t *out_sum_i64, double *out_min_f64, double *out_max_f64, int64_t *out_min_i64,
                           int64_t *out_max_i64) {
  uint64_t cap = 1024;
  while (cap < (uint64_t)g_max * 2) cap <<= 1;
  orc_table t;
  t.slot_key = (uint64_t *)malloc(cap * 8);
  t.slot_group = (int64_t *)malloc(cap * 8);
  t.mask = cap - 1;
  for (uint64_t i = 0; i < cap; ++i) t.slot_group[i] = -1;
  long double *sums = (long double *)calloc((size_t)g_max, sizeof(long double));
  int64_t n_groups = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (row_mask && !GET_BIT(row_mask, i)) continue;
    if (key_valid && !GET_BIT(key_valid, i)) continue;
    if (n_groups >= g_max) {
      /* only an existing key may still be folded */
      uint64_t s = mix64(keys[i]) & t.mask;
      int found = 0;
      while (t.slot_group[s] >= 0) {
        if (t.slot_key[s] == keys[i]) { found = 1; break; }
        s = (s + 1) & t.mask;
      }
      if (!found) { n_groups = -1; break; }
    }
    int64_t before = n_groups;
    int64_t g = table_find_or_insert(&t, keys[i], &n_groups);
    if (g == before) {
      out_keys[g] = keys[i];
      out_rows[g] = 0;
      out_nn[g] = 0;
      if (out_sum_i64) out_sum_i64[g] = 0;
    }
    out_rows[g] += 1;
    if (!values) continue;
    if (value_valid && !GET_BIT(value_valid, i)) continue;
    if (value_type == VT_F64) {
      double v = ((const double *)values)[i];
      sums[g] += v;
      if (v == v) {
        /* min / max ignore NaN; -0.0 orders below +0.0 */
        if (out_min_f64 && (out_nn[g] == 0 || out_min_f64[g] != out_min_f64[g] ||
                            ord_image_f64(v) < ord_image_f64(out_min_f64[g])))
          out_min_f64[g] = v;
        if (out_max_f64 && (out_nn[g] == 0 || out_max_f64[g] != out_max_f64[g] ||
                            ord_image_f64(v) > ord_image_f64(out_max_f64[g])))
          out_max_f64[g] = v;
      } else if (out_nn[g] == 0) {
        if (out_min_f64) out_min_f64[g] = v;
        if (out_max_f64) out_max_f64[g] = v;
      }
    } else {
      int64_t v = ((const int64_t *)values)[i];
      if (out_sum_i64) out_sum_i64[g] = (int64_t)((uint64_t)out_sum_i64[g] + (uint64_t)v);
      sums[g] += (long double)v;
      if (out_min_i64 && (out_nn[g] == 0 || v < out_min_i64[g])) out_min_i64[g] = v;
      if (out_max_i64 && (out_nn[g] == 0 || v > out_max_i64[g])) out_max_i64[g] = v;
    }
    out_nn[g] += 1;
  }
  if (n_groups >= 0 && out_sum_f64)
    for (int64_t g = 0; g < n_groups; ++g) out_sum_f64[g] = (double)sums[g];
  free(sums);
  free(t.slot_key);
  free(t.slot_group);
  return n_groups;
}

/* ---- synthetic column generator: the same counter-based law as the device generator, so a
 * shard can be reproduced on the host (kind 0 / 1 / 3 are bit-identical; kind 2 goes through
 * pow() and may differ in rare boundary rows). */
static uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
double pow(double, double);
void orc_generate(int kind, uint64_t seed, int64_t row0, int64_t n, double lo, double hi, void *out) {
  for (int64_t i = 0; i < n; ++i) {
    uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)(row0 + i)));
    double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
    if (kind == 0) {
      ((double *)out)[i] = lo + u * (hi - lo);
    } else if (kind == 1) {
      int64_t a = (int64_t)lo, b = (int64_t)hi;
      uint64_t range = (uint64_t)(b - a);
      ((int64_t *)out)[i] = a + (int64_t)(range ? r % range : 0);
    } else if (kind == 2) {
      double N = lo, s = hi, e = 1.0 - s;
      double x = pow(1.0 + u * (pow(N + 1.0, e) - 1.0), 1.0 / e);
      int64_t rank = (int64_t)x - 1;
      if (rank < 0) rank = 0;
      if (rank >= (int64_t)N) rank = (int64_t)N - 1;
      ((int64_t *)out)[i] = (int64_t)splitmix64((uint64_t)rank ^ 0x5851f42d4c957f2dull);
    } else {
      int64_t k;
      if (u < 0.70) k = 1;
      else if (u < 0.84) k = 2;
      else if (u < 0.88) k = 5;
      else if (u < 0.92) k = 3;
      else if (u < 0.95) k = 6;
      else if (u < 0.97) k = 4;
      else if (u < 0.99) k = 0;
      else if (u < 0.994) k = 7;
      else if (u < 0.997) k = 8;
      else k = 9;
      ((int64_t *)out)[i] = k;
    }
  }
}
