// Kernels specialised at build time for one lowered launch shape ("Spec": interpreter code, cache
// layout and sink descriptors fixed as compile-time constants; addresses, literal values and row
// counts stay run-time arguments).  The interpreter loop, its dispatch and the stack-slot
// selection all fold away: what is left is the straight-line code a hand-written kernel for
// that one query would have.  The interpreted kernels in qe_kernels.hip serve every shape that
// has no Spec; the results are identical by construction (same helpers, same evaluation order).
#pragma once
#include "qe_device.h"

namespace qe {

template <uint32_t KIND, bool NULLABLE>
__device__ __forceinline__ void cache_load_static(const VmCol &col, const RowCtx &c, uint64_t (&v)[VM_R], uint32_t &valid) {
  const int64_t r0 = c.row0 + 2 * (int64_t)c.lane;
  const int64_t r1 = r0 + 128;
  const bool a0 = (c.active & 3u) != 0, a1 = (c.active & 12u) != 0;
  v[0] = v[1] = v[2] = v[3] = 0;
  if constexpr (KIND == CK_64) {
    const uint64_t *p = (const uint64_t *)col.values;
    if (a0) {
      U64x2 t = stream_load_pair64(p + r0);
      v[0] = t.x;
      v[1] = t.y;
    }
    if (a1) {
      U64x2 t = stream_load_pair64(p + r1);
      v[2] = t.x;
      v[3] = t.y;
    }
  } else if constexpr (KIND == CK_32) {
    const uint32_t *p = (const uint32_t *)col.values;
    if (a0) {
      U32x2 t = stream_load_pair32(p + r0);
      v[0] = (uint64_t)(int64_t)(int32_t)t.x;
      v[1] = (uint64_t)(int64_t)(int32_t)t.y;
    }
    if (a1) {
      U32x2 t = stream_load_pair32(p + r1);
      v[2] = (uint64_t)(int64_t)(int32_t)t.x;
      v[3] = (uint64_t)(int64_t)(int32_t)t.y;
    }
  } else {
    uint32_t bits = load_bits4((const uint8_t *)col.values, c);
#pragma unroll
    for (int j = 0; j < VM_R; ++j) v[j] = (bits >> j) & 1u;
  }
  valid = 0xFu;
  if constexpr (NULLABLE) valid = load_bits4(col.validity, c);
}

// Loads the cache slots named by MASK (a compile-time constant), decoded, all loads in flight.
template <class Spec, uint32_t MASK, int I>
__device__ __forceinline__ void cache_fill_static(const VmProgram &P, const RowCtx &c, ColCache &K) {
  if constexpr (I < Spec::cache_n) {
    if constexpr ((MASK >> I) & 1u)
      cache_load_static<Spec::cache_kind[I], Spec::cache_nullable[I] != 0>(P.cols[P.cache_col[I]], c, K.v[I], K.valid[I]);
    cache_fill_static<Spec, MASK, I + 1>(P, c, K);
  }
}

// The same for batch `b` of a launch over several batches (AggArgs::n_more): batch 0's columns are the program's.
template <class Spec, uint32_t MASK, int I>
__device__ __forceinline__ void cache_fill_batch(const AggArgs &A, int b, const RowCtx &c, ColCache &K) {
  if constexpr (I < Spec::cache_n) {
    if constexpr ((MASK >> I) & 1u) {
      const VmCol &col = b == 0 ? A.prog.cols[A.prog.cache_col[I]] : A.more_cols[b - 1][I];
      cache_load_static<Spec::cache_kind[I], Spec::cache_nullable[I] != 0>(col, c, K.v[I], K.valid[I]);
    }
    cache_fill_batch<Spec, MASK, I + 1>(A, b, c, K);
  }
}

template <class Spec, uint32_t MASK, int I>
__device__ __forceinline__ void cache_copy_static(ColCache &dst, const ColCache &src) {
  if constexpr (I < Spec::cache_n) {
    if constexpr ((MASK >> I) & 1u) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) dst.v[I][j] = src.v[I][j];
      dst.valid[I] = src.valid[I];
    }
    cache_copy_static<Spec, MASK, I + 1>(dst, src);
  }
}

// Integer division / remainder with the null rules of the interpreter (x/0, x%0 and MIN/-1 give
// null).  Rolled over the four rows: a 64-bit division is long, four copies of it are not worth
// their instruction-cache space.
template <bool W32, bool WANT_MOD>
__device__ __forceinline__ void static_int_divide(Slot &a, Slot &b) {
  const int64_t tmin = W32 ? (int64_t)INT32_MIN : INT64_MIN;
  uint32_t valid = a.valid & b.valid;
#pragma unroll 1
  for (int it = 0; it < VM_R; ++it) {
    int64_t x = (int64_t)a.v[0], y = (int64_t)b.v[0];
    int64_t r = 0;
    if (y == 0 || (x == tmin && y == -1)) {
      valid &= ~1u;
    } else {
      int64_t q = x / y;
      r = WANT_MOD ? x - q * y : q;
    }
    a.v[0] = (uint64_t)r;
    QE_ROT4(a.v);
    QE_ROT4(b.v);
    valid = rot4(valid);
  }
  a.valid = valid;
}

// The code of one expression, unrolled at compile time: instruction PC, then PC + 1, ... up to
// OP_END.  Every stack slot and cache entry is named by a constant.
template <class Spec, int PC>
__device__ __forceinline__ void eval_static(const VmProgram &P, const RowCtx &c, const ColCache &K, Slot (&S)[VM_MAX_DEPTH]) {
  constexpr uint32_t ins = Spec::code[PC];
  constexpr uint32_t op = ins & 0xffu, d = (ins >> 8) & 0xfu, arg = ins >> 12;
  static_assert(vm_op_specialisable(op), "utf8 comparisons stay with the interpreter kernels");
  if constexpr (op != OP_END) {
    if constexpr (op == OP_PUSH_LIT) {
      const uint64_t l = P.lits[arg];
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d].v[j] = l;
      S[d].valid = 0xFu;
    } else if constexpr (vm_op_is_cached_push(op)) {
      // the cache already holds decoded values (cache_load_static)
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d].v[j] = K.v[arg][j];
      S[d].valid = K.valid[arg];
    } else if constexpr (op == OP_PUSH_COL64 || op == OP_PUSH_COL32 || op == OP_PUSH_COLBOOL) {
      // a column beyond the cache: loaded where it is used
      const VmCol &col = P.cols[arg];
      constexpr uint32_t kind = op == OP_PUSH_COL64 ? CK_64 : op == OP_PUSH_COL32 ? CK_32 : CK_BOOL;
      cache_load_static<kind, false>(col, c, S[d].v, S[d].valid);
      if (col.validity) S[d].valid = load_bits4(col.validity, c);
    } else if constexpr (op == OP_PUSH_ROWID) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d].v[j] = (uint64_t)(c.row0 + (j >> 1) * 128 + 2 * (int64_t)c.lane + (j & 1));
      S[d].valid = 0xFu;
    } else if constexpr (op == OP_SWAP) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) {
        const uint64_t t = S[d - 2].v[j];
        S[d - 2].v[j] = S[d - 1].v[j];
        S[d - 1].v[j] = t;
      }
      const uint32_t tv = S[d - 2].valid;
      S[d - 2].valid = S[d - 1].valid;
      S[d - 1].valid = tv;
    } else if constexpr (op == OP_I64_TO_F64) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d - 1].v[j] = d2u((double)(int64_t)S[d - 1].v[j]);
    } else if constexpr (op == OP_UTF8_EQ_LIT16 || op == OP_UTF8_NE_LIT16) {
      const VmCol &col = P.cols[arg & 0xffu];
      constexpr uint32_t li = (arg >> 8) & 0xffu;
      const uint32_t eq = utf8_equals_lit16(col.offsets, (const uint8_t *)col.values, P.lits[li], P.lits[li + 1], (uint32_t)P.lits[li + 2], c.row0, c.lane, c.active);
      const uint32_t bits = op == OP_UTF8_EQ_LIT16 ? eq : ~eq;
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d].v[j] = (bits >> j) & 1u;
      S[d].valid = col.validity ? load_bits4(col.validity, c) : 0xFu;
    } else if constexpr (op == OP_MIX64) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d - 1].v[j] = mix64(S[d - 1].v[j]);
    } else if constexpr (op == OP_FIX_EXP) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d - 1].v[j] = fix_exponent(S[d - 1].v[j]);
    } else if constexpr (op == OP_FIX_LIMB0 || op == OP_FIX_LIMB1 || op == OP_FIX_LIMB2) {
      const int scale = (int)(int64_t)P.lits[arg];
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d - 1].v[j] = fix_limb(S[d - 1].v[j], scale, (int)(op - OP_FIX_LIMB0));
    } else if constexpr (op == OP_FIX_SPECIAL) {
      const int scale = (int)(int64_t)P.lits[arg & 0xffu];
      unsigned long long *overflow = (unsigned long long *)(uintptr_t)P.lits[(arg >> 8) & 0xffu];
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d - 1].v[j] = fix_special(S[d - 1].v[j], scale, overflow, ((c.active & S[d - 1].valid) >> j) & 1u);
    } else if constexpr (op == OP_DIV_F64) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[d - 2].v[j] = d2u(u2d(S[d - 2].v[j]) / u2d(S[d - 1].v[j]));
      S[d - 2].valid &= S[d - 1].valid;
    } else if constexpr (op == OP_DIV_I32 || op == OP_MOD_I32 || op == OP_DIV_I64 || op == OP_MOD_I64) {
      static_int_divide<op == OP_DIV_I32 || op == OP_MOD_I32, op == OP_MOD_I32 || op == OP_MOD_I64>(S[d - 2], S[d - 1]);
    } else {
      static_assert(op_is_fast_binary(op), "unhandled instruction");
      vm_binary_fast(op, S[d - 2], S[d - 1]);
    }
    eval_static<Spec, PC + 1>(P, c, K, S);
  }
}

// --------------------------------------------------------------- projection
template <class Spec, int E>
__device__ __forceinline__ void project_outs_static(const ProjectArgs &A, const RowCtx &c, const ColCache &K) {
  if constexpr (E < Spec::n_outs) {
    Slot S[VM_MAX_DEPTH];
    eval_static<Spec, Spec::out_pc[E]>(A.prog, c, K, S);
    VmOut o = A.outs[E];
    o.type = (uint8_t)Spec::out_type[E];
    if constexpr (Spec::out_nullable[E] == 0) o.validity = nullptr;
    store_dense(o, c, S[0]);
    project_outs_static<Spec, E + 1>(A, c, K);
  }
}

template <class Spec>
__global__ __launch_bounds__(VM_BLOCK_THREADS, 4) void qe_project_static(const ProjectArgs A) {
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const int64_t n = A.n_rows;
  const int64_t nsteps = (n + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  for (int64_t ws = (int64_t)blockIdx.x * VM_BLOCK_WAVES + wave; ws < nsteps;
       ws += (int64_t)gridDim.x * VM_BLOCK_WAVES) {
    RowCtx c;
    c.row0 = ws * VM_WAVE_ROWS;
    c.n = n;
    c.lane = lane;
    c.active = in_range_bits(c.row0, lane, n);
    ColCache K;
    cache_fill_static<Spec, (1u << Spec::cache_n) - 1u, 0>(A.prog, c, K);
    project_outs_static<Spec, 0>(A, c, K);
  }
}

// ------------------------------------------------------------------- filter
// A wave-step's selected values of an 8-byte column that cannot be null go out through the wave's LDS staging row:
// every lane drops its values at their ranks, then takes two neighbours back and writes 16 bytes - one full store
// instruction per 128 values instead of four instructions that each fill a fraction of the same cache lines.
struct alignas(8) U64Pair8 {
  uint64_t x, y;
};
typedef uint64_t u64v2a8 __attribute__((ext_vector_type(2), aligned(8)));

template <class Spec, int E>
__device__ __forceinline__ void filter_outs_static(const FilterArgs &A, const RowCtx &c, const ColCache &K, unsigned long long obase,
                                                   uint32_t sel, const uint32_t (&rank)[VM_R], uint64_t *stage, uint32_t cnt) {
  if constexpr (E < Spec::n_outs) {
    Slot S[VM_MAX_DEPTH];
    eval_static<Spec, Spec::out_pc[E]>(A.prog, c, K, S);
    VmOut o = A.outs[E];
    o.type = (uint8_t)Spec::out_type[E];
    if constexpr (Spec::out_nullable[E] == 0) o.validity = nullptr;
    if constexpr (Spec::out_type[E] == VT_I64 || Spec::out_type[E] == VT_F64) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j)
        if ((sel >> j) & 1u) stage[rank[j]] = S[0].v[j];
      uint64_t *out = (uint64_t *)o.values + obase;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t i = (uint32_t)h * 128u + 2u * c.lane;
        if (i + 1 < cnt) {
          const U64Pair8 two = *(const U64Pair8 *)(stage + i);
          u64v2a8 t;
          t.x = two.x, t.y = two.y;
          __builtin_nontemporal_store(t, (u64v2a8 *)(out + i));
        } else if (i < cnt) {
          out[i] = stage[i];
        }
      }
      if constexpr (Spec::out_nullable[E] != 0) store_validity_ranked(o.validity, obase, sel, rank, S[0].valid, (uint8_t *)stage, cnt, c.lane);
    } else {
      store_compact_ranked(o, obase, sel, rank, S[0], (uint8_t *)stage, cnt, c.lane);
    }
    filter_outs_static<Spec, E + 1>(A, c, K, obase, sel, rank, stage, cnt);
  }
}

// Tile geometry of the specialised filter kernel.  What a tile costs beyond its bytes is fixed:
// one ticket, two barriers, one look-back whose walk is as long as the number of tiles in flight
// on the chip.  Measured on MI355X (c > 0.5 -> a + b*c, 256 M rows): 4096-row tiles with three
// 256-thread workgroups per compute unit 1.72 ms; 8192-row tiles with ONE 1024-thread workgroup per
// compute unit 1.34 ms (the same kernel without any look-back: 1.24 ms).  So: big tiles, few of
// them in flight.  A 1024-thread workgroup leaves 128 registers per lane, enough for two steps of
// up to three cached columns; wider shapes use 512 threads (256 registers) and more steps.
template <class Spec>
struct FilterGeometry {
  static constexpr int WAVES = filter_static_waves(Spec::cache_n);
  static constexpr int STEPS = filter_static_steps(Spec::cache_n);
  static constexpr int STEP_ROWS = WAVES * VM_WAVE_ROWS;
  static constexpr int TILE_ROWS = STEPS * STEP_ROWS;
};

// Same ticket / look-back protocol as qe_filter_kernel.  The loads of all steps of a phase are
// issued back to back: a tile costs three dependent trips to memory (ticket, predicate columns,
// look-back with the remaining output columns in flight) however many steps it has.  Columns the
// predicate loaded stay in registers for the output expressions.
template <class Spec>
__global__ __launch_bounds__(FilterGeometry<Spec>::WAVES * 64, 1) void qe_filter_static(const FilterArgs A) {
  using G = FilterGeometry<Spec>;
  constexpr int STEPS = G::STEPS, WAVES = G::WAVES;
  __shared__ unsigned long long s_tile;
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_cnt[STEPS * WAVES];
  __shared__ __attribute__((aligned(16))) uint64_t s_stage[WAVES][VM_WAVE_ROWS];  // a wave's selected values on their way out
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const int64_t n = A.n_rows;
  unsigned long long *status = (unsigned long long *)(A.scratch + 1);
  constexpr uint32_t PRED_MASK = Spec::pred_cache_mask;
  constexpr uint32_t LATE_MASK = Spec::out_cache_mask & ~Spec::pred_cache_mask;

  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&A.scratch->ticket, 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)s_tile;
    if (tile >= A.n_tiles) break;

    ColCache K[STEPS];
    RowCtx c[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      c[s].row0 = tile * G::TILE_ROWS + s * G::STEP_ROWS + wave * VM_WAVE_ROWS;
      c[s].n = n;
      c[s].lane = lane;
      c[s].active = in_range_bits(c[s].row0, lane, n);
      cache_fill_static<Spec, PRED_MASK, 0>(A.prog, c[s], K[s]);
    }
    uint32_t sel[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      Slot S[VM_MAX_DEPTH];
      eval_static<Spec, Spec::pred_pc>(A.prog, c[s], K[s], S);
      sel[s] = c[s].active & S[0].valid & bool_bits(S[0]);
      const uint32_t cnt = popc_sel(sel[s]);
      if (lane == 0) s_cnt[s * WAVES + wave] = cnt;
      // the remaining output columns, for selected rows only; in flight during the look-back
      c[s].active = sel[s];
      cache_fill_static<Spec, LATE_MASK, 0>(A.prog, c[s], K[s]);
    }
    __syncthreads();
    uint32_t pref[STEPS];
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < STEPS * WAVES; ++i) {
      if ((uint32_t)(i % WAVES) == wave) pref[i / WAVES] = run;
      run += s_cnt[i];
    }
    if (wave == 0) {
      unsigned long long excl = lookback_exclusive(status, tile, run, lane, &A.scratch->error);
      if (lane == 0) {
        s_base = excl;
        if (tile == A.n_tiles - 1) A.scratch->total = excl + run;
      }
    }
    __syncthreads();
    const unsigned long long base = s_base;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (__ballot(sel[s] != 0) == 0) continue;
      uint32_t rank[VM_R];
      rank_rows(sel[s], rank);
      filter_outs_static<Spec, 0>(A, c[s], K[s], base + pref[s], sel[s], rank, s_stage[wave], s_cnt[s * WAVES + wave]);
    }
  }
}

// ---------------------------------------------------------------- aggregate
// Hot-group cache.  In a low-cardinality or skewed GROUP BY most rows belong to a handful of keys,
// and sending each of them through an LDS atomic makes the LDS pipeline the limit long before
// HBM is.  Each wave therefore keeps up to HOT_KEYS keys of its own (wave-uniform) together with
// PER-LANE partial accumulators for them in registers: a row of a hot key costs a compare and a
// masked add, no lookup, no atomic, no cross-lane traffic.  The partials are reduced across the
// wave and added to the global table when the key is replaced and at the end of the kernel.
// A key becomes hot when it holds at least HOT_MIN_ROWS of the 256 rows of a wave-step; a hot key
// that gets no row in a wave-step gives way to a new candidate (runs of equal keys).  When several
// candidates in a row fail the test (high-cardinality uniform keys) the wave stops testing for a
// while, so the cost there is a few instructions per wave-step.
constexpr int HOT_KEYS = 2;
constexpr uint32_t HOT_MIN_ROWS = 12;
constexpr uint32_t HOT_FAIL_LIMIT = 4;   // failed candidates in a row before backing off
constexpr uint32_t HOT_BACKOFF_STEPS = 32;

template <int NA>
struct HotCache {
  uint64_t key[HOT_KEYS];  // wave-uniform
  uint32_t valid;          // wave-uniform; bit h: slot h holds a key
  uint32_t fails;          // wave-uniform; candidates that failed in a row
  uint32_t pause;          // wave-uniform; wave-steps left without candidate tests
  uint32_t rows[HOT_KEYS];                  // per lane
  uint64_t acc[HOT_KEYS][NA > 0 ? NA : 1];  // per lane partial accumulators
  uint32_t nn[HOT_KEYS][NA > 0 ? NA : 1];   // per lane count of valid inputs
};

// Clears slot HSLOT's partials when `on` (selects, not stores under a branch: the optimiser merges
// such stores of different slots into one store with a computed index, and the arrays end up in
// scratch memory).
template <class Spec, int HSLOT>
__device__ __forceinline__ void hot_reset(HotCache<Spec::n_accs> &C, bool on) {
  C.rows[HSLOT] = on ? 0u : C.rows[HSLOT];
#pragma unroll
  for (int a = 0; a < Spec::n_accs; ++a) {
    C.acc[HSLOT][a] = on ? acc_identity(Spec::acc_op[a]) : C.acc[HSLOT][a];
    C.nn[HSLOT][a] = on ? 0u : C.nn[HSLOT][a];
  }
}

// Wave-wide, `on` wave-uniform: adds slot HSLOT's partials to the global table and clears them.
template <class Spec, int HSLOT>
__device__ __forceinline__ void hot_flush(HotCache<Spec::n_accs> &C, const AggTable &T, uint32_t lane, bool on) {
  if (on) {
    const uint64_t key = C.key[HSLOT];
    const unsigned long long rows = wave_sum_u64(C.rows[HSLOT]);
    long long g = -1;
    if (lane == 0 && rows != 0) g = global_find_or_claim(T, key, mix64(key));  // full table: flagged for the host
    if (lane == 0 && g >= 0) atomicAdd(&T.rowcnt[g], rows);
#pragma unroll
    for (int a = 0; a < Spec::n_accs; ++a) {
      const uint32_t op = Spec::acc_op[a];  // constant after unrolling
      const uint64_t part = wave_reduce_acc(op, C.acc[HSLOT][a]);
      const unsigned long long nn = wave_sum_u64(C.nn[HSLOT][a]);
      if (lane == 0 && g >= 0) {
        if (part != acc_identity(op) || op == ACC_SUM_F64) acc_apply(&T.acc[(uint64_t)a * T.stride + g], op, part);
        if (Spec::acc_track[a] != 0 && nn != 0) atomicAdd(&T.nn[(uint64_t)a * T.stride + g], nn);
      }
    }
  }
  hot_reset<Spec, HSLOT>(C, on);
}

// Rows of the hot keys among `active` (bits per lane) -> hm[]; the other rows are returned.
// Also runs the install / replace policy.
template <class Spec>
__device__ __forceinline__ uint32_t hot_match(HotCache<Spec::n_accs> &C, const AggTable &T, const uint64_t (&key)[VM_R], uint32_t active,
                                              uint32_t lane, uint32_t (&hm)[HOT_KEYS]) {
  uint32_t rest = active;
#pragma unroll
  for (int h = 0; h < HOT_KEYS; ++h) {
    hm[h] = 0;
    if ((C.valid >> h) & 1u) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j)
        if (((active >> j) & 1u) && key[j] == C.key[h]) hm[h] |= 1u << j;
    }
    rest &= ~hm[h];
  }
  if (C.pause != 0) {
    --C.pause;
    return rest;
  }
  const uint64_t ball = __ballot(rest != 0);
  if (ball == 0) return rest;
  // a slot that is free, or whose key got no row in this wave-step
  int target = -1;
#pragma unroll
  for (int h = HOT_KEYS - 1; h >= 0; --h)
    if (!((C.valid >> h) & 1u)) target = h;
  if (target < 0) {
#pragma unroll
    for (int h = HOT_KEYS - 1; h >= 0; --h)
      if (__ballot(hm[h] != 0) == 0) target = h;
  }
  if (target < 0) return rest;
  // candidate: the first remaining row of the first lane that has one
  const int src = __builtin_ctzll(ball);
  uint64_t kl = opaque64(key[3]);
  if (rest & 4u) kl = opaque64(key[2]);
  if (rest & 2u) kl = opaque64(key[1]);
  if (rest & 1u) kl = opaque64(key[0]);
  const uint64_t k0 = readlane64(kl, src);
  uint32_t mb = 0;
#pragma unroll
  for (int j = 0; j < VM_R; ++j)
    if (((rest >> j) & 1u) && key[j] == k0) mb |= 1u << j;
  if (popc_sel(mb) < HOT_MIN_ROWS) {
    if (++C.fails >= HOT_FAIL_LIMIT) {
      C.fails = 0;
      C.pause = HOT_BACKOFF_STEPS;
    }
    return rest;
  }
  C.fails = 0;
  static_assert(HOT_KEYS == 2, "one flush call per slot below");
  hot_flush<Spec, 0>(C, T, lane, target == 0 && (C.valid & 1u) != 0);
  hot_flush<Spec, 1>(C, T, lane, target == 1 && (C.valid & 2u) != 0);
  // selects, not stores under a branch: those get merged into one store with a computed index,
  // which moves the array to scratch memory
#pragma unroll
  for (int h = 0; h < HOT_KEYS; ++h) {
    C.key[h] = target == h ? k0 : C.key[h];
    hm[h] = target == h ? mb : hm[h];
  }
  C.valid |= 1u << target;
  return rest & ~mb;
}

// Folds one evaluated input into the hot keys' per-lane partials (rows hm[h]).
template <class Spec, int ACC>
__device__ __forceinline__ void hot_fold(HotCache<Spec::n_accs> &C, const Slot &in, const uint32_t (&hm)[HOT_KEYS]) {
  constexpr uint32_t op = Spec::acc_op[ACC], in_type = Spec::acc_type[ACC];
  uint32_t valid = in.valid;
  uint64_t v[VM_R];
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    uint64_t x = in.v[j];
    if (op == ACC_MIN || op == ACC_MAX) {
      if (in_type == VT_F64 && u2d(x) != u2d(x)) valid &= ~(1u << j);  // NaN is ignored by min / max
      x = ord_image(x, in_type);
    } else if (op == ACC_COUNT) {
      x = 1;
    }
    v[j] = x;
  }
#pragma unroll
  for (int h = 0; h < HOT_KEYS; ++h) {
    const uint32_t m = hm[h] & valid;
#pragma unroll
    for (int j = 0; j < VM_R; ++j)
      if ((m >> j) & 1u) C.acc[h][ACC] = acc_combine(op, C.acc[h][ACC], v[j]);
    if (Spec::acc_track[ACC] != 0) C.nn[h][ACC] += (uint32_t)__popc(m);
  }
}

// Bytes of column values a row brings in through the column cache.
template <class Spec>
constexpr int spec_row_bytes() {
  int b = 0;
  for (uint32_t i = 0; i < Spec::cache_n; ++i) b += Spec::cache_kind[i] == CK_64 ? 8 : Spec::cache_kind[i] == CK_32 ? 4 : 1;
  return b;
}

// Dense-key bins.  When the packed key of a row is a small number (a boolean, a category code, a
// passenger count) nothing has to be looked up: the key addresses a bin directly.  Every lane has
// bins of its own - rows[key][lane], acc[a][key][lane], shared by the workgroup's waves - so that
// LDS atomics on them never meet on an address within a wave and never conflict on a bank
// (consecutive lanes, consecutive words), whatever the key distribution: a row costs one LDS add
// for the count and one LDS operation per accumulator, no hashing, no probing, no compare chains.
// Rows with other keys take the hot-key / hash-table path.  The host switches the bins on (a byte
// offset into dynamic LDS) for aggregates with few groups and inputs that cannot be null.
struct DenseBins {
  uint32_t *rows;
  unsigned long long *acc;
  bool on;
};

template <class Spec>
__device__ __forceinline__ void dense_clear(const DenseBins &D) {
  if (!D.on) return;
  for (uint32_t i = threadIdx.x; i < DENSE_KEYS * 64u; i += blockDim.x) {
    D.rows[i] = 0;
#pragma unroll
    for (int a = 0; a < Spec::n_accs; ++a) D.acc[(uint32_t)a * DENSE_KEYS * 64u + i] = acc_identity(Spec::acc_op[a]);
  }
}

template <class Spec, int ACC>
__device__ __forceinline__ void dense_fold(const DenseBins &D, const Slot &in, const uint64_t (&key)[VM_R], uint32_t dm, uint32_t lane) {
  constexpr uint32_t op = Spec::acc_op[ACC], in_type = Spec::acc_type[ACC];
  const uint32_t m = dm & in.valid;
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    if ((m >> j) & 1u) {
      uint64_t x = in.v[j];
      bool take = true;
      if (op == ACC_MIN || op == ACC_MAX) {
        if (in_type == VT_F64 && u2d(x) != u2d(x)) take = false;  // NaN is ignored by min / max
        x = ord_image(x, in_type);
      } else if (op == ACC_COUNT) {
        x = 1;
      }
      if (take) acc_apply(&D.acc[((uint32_t)ACC * DENSE_KEYS + (uint32_t)key[j]) * 64u + lane], op, x);
    }
  }
}

// Wave `wave` of the workgroup adds the bins of its share of the keys to the global table.
template <class Spec>
__device__ __forceinline__ void dense_flush(const DenseBins &D, const AggTable &T, uint32_t wave, uint32_t n_waves, uint32_t lane) {
  if (!D.on) return;
  for (uint32_t key = wave; key < DENSE_KEYS; key += n_waves) {
    const unsigned long long rows = wave_sum_u64(D.rows[key * 64u + lane]);
    if (rows == 0) continue;
    long long g = -1;
    if (lane == 0) g = global_find_or_claim(T, key, mix64(key));  // full table: flagged for the host
    if (lane == 0 && g >= 0) atomicAdd(&T.rowcnt[g], rows);
#pragma unroll
    for (int a = 0; a < Spec::n_accs; ++a) {
      const uint32_t op = Spec::acc_op[a];
      const uint64_t part = wave_reduce_acc(op, D.acc[((uint32_t)a * DENSE_KEYS + key) * 64u + lane]);
      if (lane == 0 && g >= 0 && (part != acc_identity(op) || op == ACC_SUM_F64)) acc_apply(&T.acc[(uint64_t)a * T.stride + g], op, part);
    }
  }
}

template <class Spec, int T>
__device__ __forceinline__ void agg_tasks_static(const AggArgs &A, const LdsTable &L, const DenseBins &D, uint32_t &dm, RowCtx &c, const ColCache &K,
                                                 AggRows &G, HotCache<Spec::n_accs> &C, uint32_t (&hm)[HOT_KEYS], SpillState &sp, uint32_t lane) {
  if constexpr (T < Spec::n_tasks) {
    constexpr uint32_t role = Spec::task_role[T], idx = Spec::task_idx[T];
    Slot S[VM_MAX_DEPTH];
    S[0].valid = 0xFu;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) S[0].v[j] = 1;
    if constexpr (Spec::task_eval[T] != 0) eval_static<Spec, Spec::task_pc[T]>(A.prog, c, K, S);
    if constexpr (role == ROLE_PRED) {
      c.active &= S[0].valid & bool_bits(S[0]);
      if (__ballot(c.active != 0) == 0) return;  // nothing of this wave-step survives the filter
    } else if constexpr (role == ROLE_KEY) {
      c.active &= S[0].valid;
      constexpr uint32_t kbits = Spec::key_bits[idx], sh = Spec::key_shift[idx];
      constexpr uint64_t fmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
#pragma unroll
      for (int j = 0; j < VM_R; ++j) G.key[j] |= (S[0].v[j] & fmask) << sh;
    } else if constexpr (role == ROLE_LOOKUP) {
      // The bins take the rows with small keys (an LDS add per row instead of the hot keys' compare-and-select
      // chains); the hot keys in registers and then the hash tables see the rest.
      uint32_t rest = c.active;
      dm = 0;
      if (D.on) {
#pragma unroll
        for (int j = 0; j < VM_R; ++j)
          if (((rest >> j) & 1u) && G.key[j] < DENSE_KEYS) {
            dm |= 1u << j;
            atomicAdd(&D.rows[(uint32_t)G.key[j] * 64u + lane], 1u);
          }
        rest &= ~dm;
      }
      if (__ballot(rest != 0) != 0) {
        rest = hot_match<Spec>(C, A.table, G.key, rest, lane, hm);
#pragma unroll
        for (int h = 0; h < HOT_KEYS; ++h) C.rows[h] += (uint32_t)__popc(hm[h]);
      }
      // rows of other keys: LDS / global table (no dominant-key search: the hot keys took those)
      // (with the wave-wide reduction of a key that dominates what is left: skewed keys beyond the two hot ones
      // would otherwise meet on one LDS address row by row - zipf s = 2 over 100 keys 2.1 ms per 64 M rows, with it
      // 0.9 ms; the narrow-row kernels, whose body exists twice, have no registers to spare for it: 0.50 -> 0.72 ms)
      if (__ballot(rest != 0) != 0)  // (else the bins and the hot keys took every row: the common step of a low-cardinality aggregate)
        agg_lookup<(spec_row_bytes<Spec>() >= 16), (spec_row_bytes<Spec>() < 16)>(L, A.table, rest, lane, G, sp);
    } else {
      if (__ballot((hm[0] | hm[1]) != 0) != 0) hot_fold<Spec, (int)idx>(C, S[0], hm);
      if (dm != 0) dense_fold<Spec, (int)idx>(D, S[0], G.key, dm, lane);
      if (G.spill_mb != 0 && A.spill_col[idx] != 0xffu) spill_values(sp, A.spill_col[idx], S[0], G);
      if (G.lead_slot != SLOT_NONE || __ballot(G.rem != 0) != 0)  // rows folded one by one, or a dominant key's through the wave
        agg_fold(L, A.table, idx, Spec::acc_op[idx], Spec::acc_type[idx], Spec::acc_track[idx] != 0, S[0], G, lane);
    }
    agg_tasks_static<Spec, T + 1>(A, L, D, dm, c, K, G, C, hm, sp, lane);
  }
}

// Does any accumulator of the shape keep a count of its valid inputs?  (Only then does the LDS table carry them.)
template <class Spec>
constexpr bool spec_tracks_validity() {
  for (int a = 0; a < Spec::n_accs; ++a)
    if (Spec::acc_track[a] != 0) return true;
  return false;
}

// Launched with 256 threads per workgroup, several workgroups to a compute unit - or, when rows are expected to miss
// the LDS table in numbers (high-cardinality mode), with one workgroup of 1024 threads per compute unit, whose sixteen
// waves share one table four times the size: the same waves in flight, a quarter fewer rows that miss.  Either way a
// lane has 128 registers.
// Narrow rows with few accumulators stream fastest with five waves per SIMD (96 registers a lane); the others take the 128
// that four waves - a 1024-thread workgroup - leave.
template <class Spec>
constexpr int spec_min_waves_per_simd() {
  return spec_row_bytes<Spec>() <= 16 && Spec::n_accs <= 2 ? 5 : 4;
}

template <class Spec>
__global__ __launch_bounds__(AGG_WIDE_THREADS, spec_min_waves_per_simd<Spec>()) void qe_aggregate_static(const AggArgs A) {
  extern __shared__ unsigned long long lds_raw[];
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t n_waves = blockDim.x >> 6;
  const int64_t n = A.n_rows;
  LdsTable L;
  agg_lds_setup(L, lds_raw, A.lds_cap, A.lds_limit, Spec::n_accs, spec_tracks_validity<Spec>());
  DenseBins D;
  D.on = A.dense_off != 0;
  D.rows = (uint32_t *)((char *)lds_raw + A.dense_off);
  D.acc = (unsigned long long *)(D.rows + DENSE_KEYS * 64u);
  if (A.door_off != 0) L.door = (uint16_t *)((char *)lds_raw + A.door_off);
  dense_clear<Spec>(D);
  agg_lds_clear(L, A.accs, Spec::n_accs);  // ends with a barrier

  HotCache<Spec::n_accs> C;
  C.valid = 0;
  C.fails = 0;
  C.pause = (A.tune & 1u) ? 0x7fffffffu : 0u;
  C.key[0] = C.key[1] = 0;
  hot_reset<Spec, 0>(C, true);
  hot_reset<Spec, 1>(C, true);
  SpillState sp = spill_begin(A.spill, wave);

  int64_t nsteps = (n + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  auto step = [&](int64_t row0, uint32_t active) __attribute__((always_inline)) {
    RowCtx c;
    c.row0 = row0;
    c.n = n;
    c.lane = lane;
    c.active = active;
    ColCache K;
    cache_fill_static<Spec, (1u << Spec::cache_n) - 1u, 0>(A.prog, c, K);
    AggRows G;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) G.key[j] = 0, G.slot[j] = SLOT_NONE;
    G.rem = 0;
    G.lead_mb = 0;
    G.lead_slot = SLOT_NONE;
    G.spill_mb = 0;
    G.spill_pos = 0;
    uint32_t hm[HOT_KEYS] = {0, 0};
    uint32_t dm = 0;
    agg_tasks_static<Spec, 0>(A, L, D, dm, c, K, G, C, hm, sp, lane);
  };
  // Kernels that read few bytes per row are bound by instruction issue, not by HBM (measured: a 12-byte row at
  // 5.4 TB/s against 6.4 TB/s for a 16-byte one).  For them the body exists twice: the copy for whole steps - all
  // but the table's last - starts from "every row in range" as a constant, which takes the per-row range tests,
  // the guards around the loads and, in plans without a predicate, the per-row activity tests out of the code
  // (12 % less time for the 12-byte row); the wave's step counter then lives in scalar registers so that the
  // choice is a scalar branch.  Kernels with wider rows are bound by HBM and lose more to the registers the
  // second copy costs (occupancy) than they gain: one copy.
  GiveUpWatch steps_done;
  steps_done.on = A.spill.keys == nullptr;
  if constexpr (spec_row_bytes<Spec>() < 16) {
    const int64_t ws0 = (int64_t)blockIdx.x * n_waves + __builtin_amdgcn_readfirstlane(wave);
    for (int64_t ws = ws0; ws < nsteps; ws += (int64_t)gridDim.x * n_waves) {
      if (table_given_up(A.table, steps_done)) nsteps = 0;  // (a shrinking bound, not a second exit: the loop keeps its shape)
      const int64_t row0 = ws * VM_WAVE_ROWS;
      if (row0 + VM_WAVE_ROWS <= n) step(row0, 0xFu);
      else step(row0, in_range_bits(row0, lane, n));
    }
  } else {
    for (int64_t ws = (int64_t)blockIdx.x * n_waves + wave; ws < nsteps; ws += (int64_t)gridDim.x * n_waves) {
      if (table_given_up(A.table, steps_done)) nsteps = 0;
      step(ws * VM_WAVE_ROWS, in_range_bits(ws * VM_WAVE_ROWS, lane, n));
    }
  }
  spill_end(A.spill, sp, wave, lane);
  __syncthreads();  // every wave's bin updates are in
  dense_flush<Spec>(D, A.table, wave, n_waves, lane);
  hot_flush<Spec, 0>(C, A.table, lane, (C.valid & 1u) != 0);
  hot_flush<Spec, 1>(C, A.table, lane, (C.valid & 2u) != 0);
  agg_flush(L, A.table, A.accs, Spec::n_accs);
}


// ------------------------------------------------- first pass of the high-cardinality mode
// With very many groups the first pass keeps the frequent keys in the workgroup's LDS table and logs every other row
// for the partition passes.  Per row that is a hash, a look at the table, and either a handful of LDS atomics or two
// stores - but the general kernel above spends four times as many instructions as that on paths this mode never takes
// (hot keys in registers, dense bins, a dominant key's wave reduction, the global table) and is bound by instruction
// issue, not by LDS or HBM.  This kernel is the same plan with only what the mode needs:
//  * the table is probed a pair of slots at a time (one 16-byte LDS read: buckets of two, two buckets), and a key
//    that is not in the table when its probe meets an empty slot claims it (admission by second sighting, as above)
//    or is logged - there is no third home;
//  * with at most four accumulators the inputs are evaluated first and folded row by row, so that a row's LDS
//    atomics sit under one test of its lane mask.
struct SpillRows {
  uint64_t key[VM_R];
  uint32_t slot8[VM_R];  // byte offset (slot x 8) into the key array of rows whose group is in the LDS table
  uint32_t hit;          // those rows
  uint32_t miss;         // rows logged instead
  uint64_t pos;          // log position of the wave-step's first logged row
  uint32_t off[VM_R];    // + this: where the lane's row j goes
};

__device__ __forceinline__ void spill_lookup(const LdsTable &L, uint32_t active, SpillRows &G, SpillState &sp, uint32_t *hist, int hist_bits,
                                             uint32_t ablate = 0) {
#ifdef QE_ABLATE  // timing experiments (results are wrong): 32 = no look at the table, 16 = no stores into the log, 64 = no partition histogram
  const uint32_t hit = (ablate & 32u) ? 0u : lds_two_pairs_find_or_claim4(L, G.key, active, G.slot8);
  if (ablate & 64u) hist = nullptr;
#else
  const uint32_t hit = lds_two_pairs_find_or_claim4(L, G.key, active, G.slot8);
#endif
  const uint32_t miss = active & ~hit;
  G.hit = hit;
  G.miss = miss;
  G.pos = 0;
  const uint64_t any = __ballot(miss != 0);
  if (any != 0) {
    // The rows without a slot go to the wave's log slot by slot: the lanes' rows j lie next to each other, so each of
    // the four store instructions writes one dense run and a cache line of the log is written by one instruction (in
    // lane order - a lane's rows next to each other - every instruction touched every line of the step's stretch:
    // 125 M write requests reached the L2 for 35 M lines; first pass of C4 1.34 -> 1.30 ms, a million uniform keys
    // 2.28 -> 2.25).
    G.pos = sp.base + sp.cursor;
#ifdef QE_ABLATE
    if (ablate & 0x100u) G.pos = sp.base;  // every step over the region's first rows: the stores without their traffic to memory
#endif
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      const bool mine = ((miss >> j) & 1u) != 0;
      const uint64_t m = __ballot(mine);
      G.off[j] = run + mbcnt64(m);
      run += (uint32_t)__popcll(m);
#ifdef QE_ABLATE
      if (ablate & 16u) {
        if (mine && hist) atomicAdd(&hist[hash32(G.key[j]) >> (32 - hist_bits)], 1u);
        continue;
      }
#endif
      if (mine) {
        // (a log of pairs: the key is stored together with its value, below)
        if (!sp.pairs) sp.keys[G.pos + G.off[j]] = G.key[j];  // (a streaming store here measured slower: the partial lines of a wave no longer merge)
        if (hist) atomicAdd(&hist[hash32(G.key[j]) >> (32 - hist_bits)], 1u);  // (= partition_of: what the partition passes go by)
      }
    }
    sp.cursor += run;
  }
}

// One evaluated input of row `J` into accumulator `ACC` of the row's slot.
template <class Spec, int ACC, int J>
__device__ __forceinline__ void spill_fold_one(const LdsTable &L, const Slot &in, const SpillRows &G) {
  constexpr uint32_t op = Spec::acc_op[ACC], in_type = Spec::acc_type[ACC];
  if (!((in.valid >> J) & 1u)) return;
  uint64_t x = in.v[J];
  unsigned long long *p = (unsigned long long *)((char *)(L.acc + (uint32_t)ACC * L.stride) + G.slot8[J]);
  if constexpr (op == ACC_MIN || op == ACC_MAX) {
    if (in_type == VT_F64 && u2d(x) != u2d(x)) return;  // NaN is ignored by min / max
    x = ord_image(x, in_type);
    // (atomics straight away: reading the stored value first to skip those that cannot win measured slower here,
    // 1.56 -> 1.61 ms per 256 M rows - the read is a dependent LDS round trip per row)
    if constexpr (op == ACC_MIN) atomicMin(p, (unsigned long long)x);
    else atomicMax(p, (unsigned long long)x);
  } else if constexpr (op == ACC_SUM_F64) {
    atomicAdd((double *)p, u2d(x));
  } else if constexpr (op == ACC_COUNT) {
    atomicAdd(p, 1ull);
  } else {
    atomicAdd(p, (unsigned long long)x);
  }
  if constexpr (Spec::acc_track[ACC] != 0) atomicAdd((uint32_t *)((char *)(L.nn + (uint32_t)ACC * L.stride) + (G.slot8[J] >> 1)), 1u);
}

template <class Spec, int ACC, int J>
__device__ __forceinline__ void spill_fold_row(const LdsTable &L, const Slot *in, const SpillRows &G) {
  if constexpr (ACC < Spec::n_accs) {
    spill_fold_one<Spec, ACC, J>(L, in[ACC], G);
    spill_fold_row<Spec, ACC + 1, J>(L, in, G);
  }
}

template <class Spec, int J>
__device__ __forceinline__ void spill_fold_rows(const LdsTable &L, const Slot *in, const SpillRows &G) {
  if constexpr (J < VM_R) {
    if ((G.hit >> J) & 1u) {
      atomicAdd((uint32_t *)((char *)L.rowcnt + (G.slot8[J] >> 1)), 1u);
      spill_fold_row<Spec, 0, J>(L, in, G);
    }
    spill_fold_rows<Spec, J + 1>(L, in, G);
  }
}

template <class Spec, int ACC, int J>
__device__ __forceinline__ void spill_fold_acc(const LdsTable &L, const Slot &in, const SpillRows &G) {
  if constexpr (J < VM_R) {
    if ((G.hit >> J) & 1u) spill_fold_one<Spec, ACC, J>(L, in, G);
    spill_fold_acc<Spec, ACC, J + 1>(L, in, G);
  }
}

constexpr int SPILL_ROW_MAJOR_ACCS = 4;  // up to this many accumulators are folded row by row

template <class Spec, int T>
__device__ __forceinline__ void spill_tasks_static(const AggArgs &A, const LdsTable &L, RowCtx &c, const ColCache &K, SpillRows &G, Slot *in,
                                                   SpillState &sp, uint32_t *hist) {
  if constexpr (T < Spec::n_tasks) {
    constexpr uint32_t role = Spec::task_role[T], idx = Spec::task_idx[T];
    Slot S[VM_MAX_DEPTH];
    S[0].valid = 0xFu;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) S[0].v[j] = 1;
    if constexpr (Spec::task_eval[T] != 0) eval_static<Spec, Spec::task_pc[T]>(A.prog, c, K, S);
    if constexpr (role == ROLE_PRED) {
      c.active &= S[0].valid & bool_bits(S[0]);
      if (__ballot(c.active != 0) == 0) return;  // nothing of this wave-step survives the filter
    } else if constexpr (role == ROLE_KEY) {
      c.active &= S[0].valid;
      constexpr uint32_t kbits = Spec::key_bits[idx], sh = Spec::key_shift[idx];
      constexpr uint64_t fmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
#pragma unroll
      for (int j = 0; j < VM_R; ++j) G.key[j] |= (S[0].v[j] & fmask) << sh;
    } else if constexpr (role == ROLE_LOOKUP) {
      if (__ballot(c.active != 0) == 0) return;
#ifdef QE_ABLATE
      spill_lookup(L, c.active, G, sp, hist, A.cell_bits, A.tune);
#else
      spill_lookup(L, c.active, G, sp, hist, A.cell_bits);
#endif
      if constexpr (Spec::n_accs > SPILL_ROW_MAJOR_ACCS || Spec::n_accs == 0) {
#pragma unroll
        for (int j = 0; j < VM_R; ++j)
          if ((G.hit >> j) & 1u) atomicAdd((uint32_t *)((char *)L.rowcnt + (G.slot8[j] >> 1)), 1u);
      }
    } else {
#ifdef QE_ABLATE
      if (A.tune & 0x90u) {  // (128: the keys are logged, the values are not)
      } else
#endif
      if (G.miss != 0 && A.spill_col[idx] != 0xffu) {
        if (sp.pairs) {
          // one 16-byte store a row: half the store instructions of a key column and a value column
          ulonglong2 *dst = (ulonglong2 *)sp.keys + G.pos;
#pragma unroll
          for (int j = 0; j < VM_R; ++j)
            if ((G.miss >> j) & 1u)
              dst[G.off[j]] = make_ulonglong2((unsigned long long)G.key[j], spill_logged_value(sp.null_cols, sp.null_f64, 0, S[0].v[j], ((S[0].valid >> j) & 1u) != 0));
        } else {
          unsigned long long *dst = sp.vals + (uint64_t)A.spill_col[idx] * sp.capacity + G.pos;
#pragma unroll
          for (int j = 0; j < VM_R; ++j)
            if ((G.miss >> j) & 1u) dst[G.off[j]] = spill_logged_value(sp.null_cols, sp.null_f64, A.spill_col[idx], S[0].v[j], ((S[0].valid >> j) & 1u) != 0);
        }
      }
      if constexpr (Spec::n_accs <= SPILL_ROW_MAJOR_ACCS) {
        in[idx] = S[0];
      } else {
        if (__ballot(G.hit != 0) != 0) spill_fold_acc<Spec, (int)idx, 0>(L, S[0], G);
      }
    }
    spill_tasks_static<Spec, T + 1>(A, L, c, K, G, in, sp, hist);
  } else {
    if constexpr (Spec::n_accs <= SPILL_ROW_MAJOR_ACCS && Spec::n_accs > 0) {
      if (__ballot(G.hit != 0) != 0) spill_fold_rows<Spec, 0>(L, in, G);
    }
  }
}

template <class Spec>
__global__ __launch_bounds__(AGG_WIDE_THREADS, 4) void qe_aggregate_spill_static(const AggArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds_raw[];
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t n_waves = blockDim.x >> 6;
  LdsTable L;
  agg_lds_setup(L, lds_raw, A.lds_cap, A.lds_limit, Spec::n_accs, spec_tracks_validity<Spec>());
  if (A.door_off != 0) L.door = (uint16_t *)((char *)lds_raw + A.door_off);
  uint32_t *hist = A.cells != nullptr ? (uint32_t *)((char *)lds_raw + A.hist_off) : nullptr;
  if (hist)
    for (uint32_t p = threadIdx.x; p < (1u << A.cell_bits); p += blockDim.x) hist[p] = 0;
  agg_lds_clear(L, A.accs, Spec::n_accs);  // ends with a barrier
  SpillState sp = spill_begin(A.spill, wave);
  uint32_t turn = wave >> 2;  // (waves 4 apart share a SIMD)
  uint32_t selected = 0;      // rows of this wave that passed the predicate with a valid key (wave-uniform)
  for (int batch = 0; batch <= A.n_more; ++batch) {  // (the launch's batches one after the other: the table stays)
  const int64_t n = batch == 0 ? A.n_rows : A.more_rows[batch - 1];
  const int64_t nsteps = (n + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  for (int64_t ws = (int64_t)blockIdx.x * n_waves + wave; ws < nsteps; ws += (int64_t)gridDim.x * n_waves) {
    if (A.tune & 8u) {
      // The four waves of a SIMD take the high priority in turn, a step each.  Left alone the oldest wave is served
      // first whenever several are ready, the waves drift apart, and the last ones finish alone (first pass of C4:
      // 1.44 -> 1.41 ms; a fixed order the other way round, youngest first: 1.50).
      switch (turn++ & 3u) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
      }
    }
    RowCtx c;
    c.row0 = ws * VM_WAVE_ROWS;
    c.n = n;
    c.lane = lane;
    c.active = in_range_bits(c.row0, lane, n);
    ColCache K;
    cache_fill_batch<Spec, (1u << Spec::cache_n) - 1u, 0>(A, batch, c, K);
    SpillRows G;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) G.key[j] = 0, G.slot8[j] = 0;
    G.hit = 0;
    G.miss = 0;
    G.pos = 0;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) G.off[j] = 0;
    Slot in[Spec::n_accs > 0 && Spec::n_accs <= SPILL_ROW_MAJOR_ACCS ? Spec::n_accs : 1];
    spill_tasks_static<Spec, 0>(A, L, c, K, G, in, sp, hist);
#pragma unroll
    for (int j = 0; j < VM_R; ++j) selected += (uint32_t)__popcll(__ballot((((G.hit | G.miss) >> j) & 1u) != 0));
  }
  }
  spill_end(A.spill, sp, wave, lane);
  if (hist) {
    // this workgroup's column of the partition passes' cell table, and its share of the log's length
    // (one atomic a workgroup: a wave each are four thousand operations on one word of memory as the launch ends,
    // a tenth of a millisecond in which nothing else happens)
    __shared__ unsigned long long s_logged, s_selected;
    if (threadIdx.x == 0) s_logged = 0, s_selected = 0;
    __syncthreads();
    const uint32_t cells_stride = A.cells_stride ? A.cells_stride : gridDim.x;
    for (uint32_t p = threadIdx.x; p < (1u << A.cell_bits); p += blockDim.x) A.cells[(size_t)p * cells_stride + blockIdx.x] = (int32_t)hist[p];
#ifdef QE_ABLATE
    if (A.tune & 0x150u) sp.cursor = 0;  // (the log is not what the histogram says: the partition passes see none)
#endif
    if (lane == 0 && sp.cursor != 0) atomicAdd(&s_logged, (unsigned long long)sp.cursor);
    // (a launch with a block of columns in a shared cell table also says how many rows it saw: logged / selected is what
    // the host chooses between this kernel and the one that partitions directly by)
    if (lane == 0 && selected != 0) atomicAdd(&s_selected, (unsigned long long)selected);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (s_logged != 0) atomicAdd(A.cells_total, s_logged);
      if (s_selected != 0 && A.cells_stride != 0) atomicAdd(A.cells_total + 2, s_selected);
    }
  }
  if (!(A.tune & 4u)) agg_flush(L, A.table, A.accs, Spec::n_accs);  // (bit 2: a launch whose results are thrown away - the host's look at how many rows miss the tables)
}


// ------------------------------------------------- first pass of the many-groups mode that partitions as it goes
// When nearly every selected row misses the workgroup's table (very many keys, none of them frequent) the table is of
// no use, and logging the rows only to read them back and partition them costs a write and a read of every one.  This
// first pass has no table and no log: a workgroup evaluates predicate, key and value, and collects each partition's
// rows as (key, value) pairs in a ring of 16 rows - two 128-byte lines - in LDS (512 partitions x 256 bytes), writing
// a line the moment its eight rows are there.  No count pass comes first, so a workgroup cannot know where its rows of
// a partition will lie among everybody's: each (partition, workgroup) has a stretch of its own, sized for its expected
// share with room to spare, and the partition-wise aggregate walks a partition's stretches.  A stretch that turns out
// too small drops the rows beyond it and says so (AggSpill::counts[0]): the caller then runs the batch through the
// table-and-log first pass instead - skewed keys, for which that one is the right choice anyway.
// The waves of a workgroup move in rounds (a barrier between the rows' arrival and the writing of complete lines), so
// a wave's next step's columns are fetched before it works on the current one: nobody else is there to hide that trip.

template <class Spec, int T>
__device__ __forceinline__ void direct_tasks_static(const AggArgs &A, RowCtx &c, const ColCache &K, uint64_t (&key)[VM_R], uint64_t (&val)[VM_R]) {
  if constexpr (T < Spec::n_tasks) {
    constexpr uint32_t role = Spec::task_role[T], idx = Spec::task_idx[T];
    if constexpr (role == ROLE_LOOKUP) {
      if (__ballot(c.active != 0) == 0) return;
    } else if constexpr (role == ROLE_ACC) {
      if (A.spill_col[idx] != 0xffu) {  // (uniform: the accumulator whose input is the row's value)
        Slot S[VM_MAX_DEPTH];
        S[0].valid = 0xFu;
#pragma unroll
        for (int j = 0; j < VM_R; ++j) S[0].v[j] = 1;
        if constexpr (Spec::task_eval[T] != 0) eval_static<Spec, Spec::task_pc[T]>(A.prog, c, K, S);
#pragma unroll
        for (int j = 0; j < VM_R; ++j) val[j] = spill_logged_value(A.spill.null_cols, A.spill.null_f64, A.spill_col[idx], S[0].v[j], ((S[0].valid >> j) & 1u) != 0);
      }
    } else {
      Slot S[VM_MAX_DEPTH];
      S[0].valid = 0xFu;
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[0].v[j] = 1;
      if constexpr (Spec::task_eval[T] != 0) eval_static<Spec, Spec::task_pc[T]>(A.prog, c, K, S);
      if constexpr (role == ROLE_PRED) {
        c.active &= S[0].valid & bool_bits(S[0]);
        if (__ballot(c.active != 0) == 0) return;  // nothing of this wave-step survives the filter
      } else {
        c.active &= S[0].valid;
        constexpr uint32_t kbits = Spec::key_bits[idx], sh = Spec::key_shift[idx];
        constexpr uint64_t fmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
#pragma unroll
        for (int j = 0; j < VM_R; ++j) key[j] |= (S[0].v[j] & fmask) << sh;
      }
    }
    direct_tasks_static<Spec, T + 1>(A, c, K, key, val);
  }
}

template <class Spec>
__global__ __launch_bounds__(AGG_WIDE_THREADS, 4) void qe_aggregate_direct_static(const AggArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds_raw[];
  const uint32_t P = 1u << A.cell_bits;
  // A ring's stride is 17 rows: with 16 (256 bytes) every partition's slot i would sit in the same LDS banks.
  KeyVal *ring = (KeyVal *)lds_raw;                                              // [P][DIRECT_RING_STRIDE]
  unsigned long long *s_state = (unsigned long long *)(ring + (size_t)P * DIRECT_RING_STRIDE);  // [P] low word: rows arrived; high word: first row not yet written (a multiple of 8)
  uint32_t *s_valid = (uint32_t *)(s_state + P);                                 // [P] bit i: ring slot i holds a row
  uint32_t *s_ready = s_valid + P;                                               // [2 P] lines to write this round: partition | line number << 9
  uint32_t *s_nready = s_ready + 2 * P;                                          // [2] their number, rounds taking turns
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const uint32_t cap = A.spill.direct_cap;
  KeyVal *dst = (KeyVal *)A.spill.keys;
  auto stretch = [&](uint32_t p) { return ((uint64_t)p * gridDim.x + blockIdx.x) * cap; };

  for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) {
    s_state[p] = 0;
    s_valid[p] = 0;
  }
  if (threadIdx.x < 2) s_nready[threadIdx.x] = 0;
  lds_barrier();

  // Writes the lines that are complete (every row of them has arrived); `all`: whatever has arrived, at the end.
  // Three barriers a round: after the rows' arrival, after the list of lines, after the writing.
  uint32_t turn = 0;
  auto flush = [&](bool all) {
    uint32_t *nready_at = s_nready + (turn & 1u);
    for (uint32_t t = threadIdx.x; t < 2 * P; t += blockDim.x) {  // (2 P is a multiple of 64: a wave's lanes stay together)
      const uint32_t p = t >> 1, j = t & 1u;
      const unsigned long long st = s_state[p];
      const uint32_t uw = (uint32_t)st, line0 = (uint32_t)(st >> 32) + 8u * j;  // rows [line0, line0 + 8)
      const bool ready = uw >= line0 + 8u || (all && uw > line0);
      const uint64_t m = __ballot(ready);
      if (m != 0) {  // (one atomic a wave)
        uint32_t at = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) at = atomicAdd(nready_at, (uint32_t)__popcll(m));
        at = (uint32_t)__builtin_amdgcn_readlane((int)at, __builtin_ctzll(m));
        if (ready) s_ready[at + mbcnt64(m)] = p | ((line0 >> 3) << 9);
      }
    }
    lds_barrier();
    const uint32_t nready = *nready_at;
    if (threadIdx.x == 0) s_nready[(turn + 1u) & 1u] = 0;  // (last read before the barrier that ended the previous round)
    for (uint32_t q = wave * 8u; q < nready; q += n_waves * 8u) {
      const uint32_t e = q + (lane >> 3);
      if (e < nready) {
        const uint32_t entry = s_ready[e], p = entry & 511u, line = entry >> 9, r = lane & 7u;
        const uint32_t slot = (line & 1u) * 8u + r;
#ifdef QE_ABLATE  // timing experiments (results are wrong): 0x400 = the lines are not written to memory
        if (A.tune & 0x400u) {
          if (ring[(size_t)p * DIRECT_RING_STRIDE + slot].key == 0x123456789ull) dst[0].key = 0;
        } else
#endif
        if ((s_valid[p] >> slot) & 1u) {
          const KeyVal row = ring[(size_t)p * DIRECT_RING_STRIDE + slot];
          qe_ull2 t;
          t.x = row.key;
          t.y = row.val;
          if (A.tune & 0x10u) __builtin_nontemporal_store(t, (qe_ull2 *)&dst[stretch(p) + line * 8u + r]);
          else dst[stretch(p) + line * 8u + r] = row;
        }
        // the written line leaves the window (a partition's two lines are listed in order, and both or the first)
        if (r == 0) {
          atomicAnd(&s_valid[p], ~(0xffu << (8u * (line & 1u))));
          atomicAdd(&s_state[p], 8ull << 32);
        }
      }
    }
    lds_barrier();
    ++turn;
  };

  uint32_t dropped = 0;
  for (int batch = 0; batch <= A.n_more; ++batch) {  // (the launch's batches one after the other: the rings stay)
  const int64_t n = batch == 0 ? A.n_rows : A.more_rows[batch - 1];
  const int64_t nsteps = (n + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  const int64_t steps_a_round = (int64_t)gridDim.x * n_waves;
  const int64_t rounds = (nsteps + steps_a_round - 1) / steps_a_round;  // the same for every wave: barriers inside
  auto fetch = [&](int64_t ws, RowCtx &c, ColCache &K) __attribute__((always_inline)) {
    c.row0 = ws * VM_WAVE_ROWS;
    c.n = n;
    c.lane = lane;
    c.active = in_range_bits(c.row0, lane, n);  // (nothing beyond the last step)
    cache_fill_batch<Spec, (1u << Spec::cache_n) - 1u, 0>(A, batch, c, K);
  };
  int64_t ws = (int64_t)blockIdx.x * n_waves + wave;
  RowCtx cn;
  ColCache Kn;
  fetch(ws, cn, Kn);
  for (int64_t r = 0; r < rounds; ++r) {
    RowCtx c;
    c.row0 = cn.row0;
    c.n = n;
    c.lane = lane;
    c.active = cn.active;
    ColCache K;
    cache_copy_static<Spec, (1u << Spec::cache_n) - 1u, 0>(K, Kn);
    ws += steps_a_round;
    fetch(ws, cn, Kn);  // a round ahead: the trip to memory overlaps this round's work and barriers
    uint64_t key[VM_R], val[VM_R];
#pragma unroll
    for (int j = 0; j < VM_R; ++j) key[j] = 0, val[j] = 0;
    direct_tasks_static<Spec, 0>(A, c, K, key, val);
#ifdef QE_ABLATE  // 0x800 = the rows go nowhere and the waves do not wait for each other; 0x200 = the rows go nowhere, the rounds stay
    if (A.tune & 0x800u) continue;
    if (A.tune & 0x200u) c.active = 0;
#endif
    // the rows take their places: one LDS operation numbers a row and tells it where the ring's window starts
    uint32_t part[VM_R];
    unsigned long long st[VM_R];
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      part[j] = partition_of(key[j], A.cell_bits);
      st[j] = 0;
      if ((c.active >> j) & 1u) st[j] = atomicAdd(&s_state[part[j]], 1ull);
    }
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      if ((c.active >> j) & 1u) {
        const uint32_t u = (uint32_t)st[j], ur = (uint32_t)(st[j] >> 32);
        KeyVal row;
        row.key = key[j];
        row.val = val[j];
        if (u >= cap) {
          dropped = 1;  // the stretch is full
        } else if (u < ur + (uint32_t)DIRECT_RING) {
          ring[(size_t)part[j] * DIRECT_RING_STRIDE + (u & (DIRECT_RING - 1))] = row;
          atomicOr(&s_valid[part[j]], 1u << (u & (DIRECT_RING - 1)));
        } else {
          dst[stretch(part[j]) + u] = row;  // the ring is full: straight to its place (the line is completed around it)
        }
      }
    }
    lds_barrier();
    flush(false);
  }
  }
  flush(true);
  // what this workgroup wrote, per partition
  const uint32_t cells_stride = A.cells_stride ? A.cells_stride : gridDim.x;
  uint32_t mine = 0;
  for (uint32_t p = threadIdx.x; p < P; p += blockDim.x) {
    uint32_t got = (uint32_t)s_state[p];
#ifdef QE_ABLATE
    if (A.tune & 0xe00u) got = 0;  // (what was written is not what the counts would say: the partition pass sees nothing)
#endif
    if (got > cap) dropped = 1;
    A.cells[(size_t)p * cells_stride + blockIdx.x] = (int32_t)(got < cap ? got : cap);
    mine += got < cap ? got : cap;
  }
  if (__ballot(dropped != 0) != 0 && lane == 0) atomicOr(A.spill.counts, 1u);
  // (the total of the rows written: sizes nothing, but says how selective the batch was; one atomic a workgroup)
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if (threadIdx.x == 0) s_nready[0] = 0;  // (the list's counter: free by now - the last flush ended with a barrier)
  lds_barrier();
  if (lane == 0 && mine != 0) atomicAdd(&s_nready[0], mine);
  lds_barrier();
  if (threadIdx.x == 0 && s_nready[0] != 0) atomicAdd(A.cells_total, (unsigned long long)s_nready[0]);
}

}  // namespace qe
