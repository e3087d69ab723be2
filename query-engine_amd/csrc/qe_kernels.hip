// Hand-written gfx950 (CDNA4, wave64) kernels of the physical-plan hot path:
//   * qe_project_kernel   — fused expression evaluation (ProjectionIterator::next,
//                           physical_plan/mod.rs:138-156 in the reference)
//   * qe_filter_kernel    — predicate -> stable compaction -> projection
//                           (SelectionIterator::next + filter_chunk, physical_plan/mod.rs:205-229)
//   * qe_aggregate_kernel — predicate/projection fused into a hash aggregate with a per-workgroup
//                           LDS table and a global open-addressing table
//                           (AggregateExec::execute, physical_plan/mod.rs:292-374)
//   * qe_finalize_kernel  — group table -> output batch (physical_plan/mod.rs:375-488)
// All of them are HBM-bandwidth bound byte/integer/f64 streaming work: wide coalesced loads,
// wave64 ballot/popcount ranking, LDS scans and LDS atomics; there is no dense contraction here.
//
// Code size is a first-order cost for the interpreter kernels: the instruction cache holds
// 64 KiB and a wave that jumps around a larger body stalls on instruction fetch.  Hence every
// kernel instantiates the interpreter ONCE, rare operations share one out-of-line-style copy
// that works on temporaries, and per-row work that is more than a few instructions long runs in
// a rolled loop that rotates the four row registers instead of being unrolled four times.
#include "qe_kernels.h"

namespace qe {

// ------------------------------------------------------------------ helpers
// Column value buffers are 16-byte aligned (the host re-aligns imported buffers that are not),
// so a lane's row pair is one global_load_dwordx4 / global_store_dwordx4.  A pair that starts on
// a valid row never leaves the 16-byte granule of that row, so it cannot cross into an unmapped
// page even when the second row is past the end of the column.
struct alignas(16) U64x2 { uint64_t x, y; };
struct alignas(8) U32x2 { uint32_t x, y; };

struct Slot {
  uint64_t v[VM_R];
  uint32_t valid;  // bit j: row j holds a non-null value
};

// Rows of one wave-step.  Lane l, slot j  <->  row0 + (j>>1)*128 + 2*l + (j&1):
// every 16-byte load of an 8-byte column is contiguous across the 64 lanes.
struct RowCtx {
  int64_t row0;
  int64_t n;
  uint32_t lane;
  uint32_t active;  // bit j: row j takes part
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint32_t in_range_bits(int64_t row0, uint32_t lane, int64_t n) {
  uint32_t a = 0;
  int64_t r = row0 + 2 * (int64_t)lane;
  if (r < n) a |= 1u;
  if (r + 1 < n) a |= 2u;
  if (r + 128 < n) a |= 4u;
  if (r + 129 < n) a |= 8u;
  return a;
}

// Hides a lane-varying value from loop-invariant code motion: without it everything the
// interpreter might derive from loop-invariant registers (addresses from the lane id, decoded
// forms of every cached column) is precomputed ahead of the instruction loop and kept live
// across it, which costs far more registers than the work saved.
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ uint64_t opaque64(uint64_t x) {
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }

__device__ __forceinline__ uint32_t rot4(uint32_t m) { return ((m >> 1) | (m << 3)) & 0xFu; }
#define QE_ROT4(a)   \
  {                  \
    auto t_ = (a)[0]; \
    (a)[0] = (a)[1]; \
    (a)[1] = (a)[2]; \
    (a)[2] = (a)[3]; \
    (a)[3] = t_;     \
  }

// Two adjacent bits (rows 2l, 2l+1 of each chunk) out of an LSB-first bitmap.
__device__ __forceinline__ uint32_t load_bits4(const uint8_t *bm, const RowCtx &c) {
  uint32_t out = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t a2 = (c.active >> (2 * h)) & 3u;
    if (a2) {
      int64_t r = c.row0 + h * 128 + 2 * (int64_t)c.lane;
      uint32_t byte = bm[r >> 3];
      out |= ((byte >> (r & 7)) & 3u) << (2 * h);
    }
  }
  return out;
}

__device__ __forceinline__ bool utf8_equal(const uint8_t *a, const uint8_t *b, int32_t len) {
  for (int32_t i = 0; i < len; ++i)
    if (a[i] != b[i]) return false;
  return true;
}

// ------------------------------------------------------- per-step column cache
// Raw bits of up to VM_NCACHE columns for the 4 rows of the lane.  Filled by cache_load with
// every load in flight at once; only ever indexed with compile-time constants.
struct ColCache {
  uint64_t v[VM_NCACHE][VM_R];
  uint32_t valid[VM_NCACHE];  // validity bits (4 per lane), 0xF when the column has no bitmap
};

__device__ __forceinline__ void cache_load_one(const VmProgram &P, int i, const RowCtx &c, uint64_t (&v)[VM_R],
                                               uint32_t &valid) {
  const VmCol &col = P.cols[P.cache_col[i]];
  const uint32_t kind = P.cache_kind[i];
  // every path produces the four words in locals and the cache entry is written once, below:
  // storing from inside the branches makes the compiler merge them into one store with a
  // computed index, which pushes the whole cache out to scratch memory
  uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
  const int64_t r0 = c.row0 + 2 * (int64_t)c.lane;
  const int64_t r1 = r0 + 128;
  const bool a0 = (c.active & 3u) != 0, a1 = (c.active & 12u) != 0;
  if (kind == CK_64) {
    const uint64_t *p = (const uint64_t *)col.values;
    if (a0) {
      U64x2 t = *(const U64x2 *)(p + r0);
      w0 = t.x;
      w1 = t.y;
    }
    if (a1) {
      U64x2 t = *(const U64x2 *)(p + r1);
      w2 = t.x;
      w3 = t.y;
    }
  } else if (kind == CK_32) {
    const uint32_t *p = (const uint32_t *)col.values;  // raw pairs; sign-extended when pushed
    if (a0) {
      U32x2 t = *(const U32x2 *)(p + r0);
      w0 = (uint64_t)t.x | ((uint64_t)t.y << 32);
    }
    if (a1) {
      U32x2 t = *(const U32x2 *)(p + r1);
      w1 = (uint64_t)t.x | ((uint64_t)t.y << 32);
    }
  } else {
    w0 = load_bits4((const uint8_t *)col.values, c);
  }
  uint32_t vb = 0xFu;
  if (col.validity) vb = load_bits4(col.validity, c);
  v[0] = w0;
  v[1] = w1;
  v[2] = w2;
  v[3] = w3;
  valid = vb;
}

template <int I>
__device__ __forceinline__ void cache_load_from(const VmProgram &P, uint32_t mask, const RowCtx &c, ColCache &K) {
  if constexpr (I < VM_NCACHE) {
    if ((mask >> I) & 1u) cache_load_one(P, I, c, K.v[I], K.valid[I]);
    cache_load_from<I + 1>(P, mask, c, K);
  }
}
__device__ __forceinline__ void cache_load(const VmProgram &P, uint32_t mask, const RowCtx &c, ColCache &K) {
  cache_load_from<0>(P, mask, c, K);
}

// ------------------------------------------------------------ the VM proper
// The program (instruction words, literals) is wave-uniform data that the interpreter reads at a
// run-time index.  Reading it from the kernel-argument segment costs one scalar-memory round
// trip per fetch, in a dependent chain; instead lane i of a few VGPRs holds word i, loaded once
// per kernel, and a fetch is a v_readlane_b32.
struct ProgRegs {
  uint32_t code0, code1;    // insn[lane], insn[64 + lane]
  uint32_t lit_lo, lit_hi;  // lits[lane & 31]
};
static_assert(VM_MAX_INSN == 128 && VM_MAX_LITS <= 64, "ProgRegs layout");

__device__ __forceinline__ ProgRegs prog_regs(const VmProgram &P, uint32_t lane) {
  ProgRegs R;
  R.code0 = P.insn[lane];
  R.code1 = P.insn[64 + lane];
  uint64_t l = P.lits[lane & (VM_MAX_LITS - 1)];
  R.lit_lo = (uint32_t)l;
  R.lit_hi = (uint32_t)(l >> 32);
  return R;
}
__device__ __forceinline__ uint32_t fetch_insn(const ProgRegs &R, uint32_t pc) {
  return pc < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)R.code0, (int)pc)
                 : (uint32_t)__builtin_amdgcn_readlane((int)R.code1, (int)(pc - 64));
}
__device__ __forceinline__ uint64_t fetch_lit(const ProgRegs &R, uint32_t i) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)R.lit_lo, (int)i);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)R.lit_hi, (int)i);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// Selecting a stack slot / cache entry by a run-time (wave-uniform) number.  Each case passes
// its values through `opaque`: otherwise the optimiser folds the cases into one access with a
// computed index, and an array indexed that way is placed in scratch memory instead of registers.
__device__ __forceinline__ void slot_copy(Slot &r, const Slot &s) {
  // element by element: a whole-struct copy is a memcpy to the optimiser and leaves the
  // slots in scratch memory
  r.v[0] = opaque64(s.v[0]);
  r.v[1] = opaque64(s.v[1]);
  r.v[2] = opaque64(s.v[2]);
  r.v[3] = opaque64(s.v[3]);
  r.valid = opaque(s.valid);
}
__device__ __forceinline__ void slot_get(const Slot (&S)[VM_MAX_DEPTH], uint32_t i, Slot &out) {
  switch (i) {
    case 0: slot_copy(out, S[0]); break;
    case 1: slot_copy(out, S[1]); break;
    case 2: slot_copy(out, S[2]); break;
    default: slot_copy(out, S[3]); break;
  }
}
__device__ __forceinline__ void slot_put(Slot (&S)[VM_MAX_DEPTH], uint32_t i, const Slot &x) {
  switch (i) {
    case 0: slot_copy(S[0], x); break;
    case 1: slot_copy(S[1], x); break;
    case 2: slot_copy(S[2], x); break;
    default: slot_copy(S[3], x); break;
  }
}

__device__ __forceinline__ void cached_raw(const ColCache &K, uint32_t slot, uint64_t (&v)[VM_R], uint32_t &valid) {
#define QE_TAKE(i)                                                               \
  {                                                                              \
    _Pragma("unroll") for (int j = 0; j < VM_R; ++j) v[j] = opaque64(K.v[i][j]); \
    valid = opaque(K.valid[i]);                                                  \
  }
  switch (slot) {
    case 0: QE_TAKE(0) break;
    case 1: QE_TAKE(1) break;
    case 2: QE_TAKE(2) break;
    case 3: QE_TAKE(3) break;
    case 4: QE_TAKE(4) break;
    default: QE_TAKE(5) break;
  }
#undef QE_TAKE
}

// Hot pushes, instantiated per stack depth: a cached column or a literal.
__device__ __forceinline__ void vm_push_fast(uint32_t op, uint32_t arg, const ProgRegs &R, const ColCache &K, Slot &s) {
  if (op == OP_PUSH_LIT) {
    uint64_t l = fetch_lit(R, arg);
#pragma unroll
    for (int j = 0; j < VM_R; ++j) s.v[j] = l;
    s.valid = 0xFu;
    return;
  }
  cached_raw(K, arg, s.v, s.valid);
  if (op == OP_PUSH_CACHED32) {
    // raw int32 pairs -> sign-extended values.  `opaque`: decoded here, per instruction; left
    // visible, the decoded form of every cache slot is precomputed ahead of the instruction loop
    uint32_t w0 = opaque((uint32_t)s.v[0]), w1 = opaque((uint32_t)(s.v[0] >> 32));
    uint32_t w2 = opaque((uint32_t)s.v[1]), w3 = opaque((uint32_t)(s.v[1] >> 32));
    s.v[0] = (uint64_t)(int64_t)(int32_t)w0;
    s.v[1] = (uint64_t)(int64_t)(int32_t)w1;
    s.v[2] = (uint64_t)(int64_t)(int32_t)w2;
    s.v[3] = (uint64_t)(int64_t)(int32_t)w3;
  } else if (op == OP_PUSH_CACHEDBOOL) {
    uint32_t bits = opaque((uint32_t)s.v[0]);
#pragma unroll
    for (int j = 0; j < VM_R; ++j) s.v[j] = (bits >> j) & 1u;
  }
}

#define QE_BIN_EACH(expr)                            \
  _Pragma("unroll") for (int j = 0; j < VM_R; ++j) { \
    uint64_t x = a.v[j], y = b.v[j];                 \
    (void)x;                                         \
    (void)y;                                         \
    a.v[j] = (expr);                                 \
  }

// Hot binaries, instantiated per stack depth: a = a OP b.  Result validity = AND of the
// operands (arrow2 binary kernels combine validity bitmaps the same way).
__device__ __forceinline__ void vm_binary_fast(uint32_t op, Slot &a, const Slot &b) {
  switch (op) {
    case OP_ADD_I32: QE_BIN_EACH((uint64_t)(int64_t)(int32_t)((uint32_t)x + (uint32_t)y)); break;
    case OP_SUB_I32: QE_BIN_EACH((uint64_t)(int64_t)(int32_t)((uint32_t)x - (uint32_t)y)); break;
    case OP_MUL_I32: QE_BIN_EACH((uint64_t)(int64_t)(int32_t)((uint32_t)x * (uint32_t)y)); break;
    case OP_ADD_I64: QE_BIN_EACH(x + y); break;
    case OP_SUB_I64: QE_BIN_EACH(x - y); break;
    case OP_MUL_I64: QE_BIN_EACH(x * y); break;
    case OP_ADD_F64: QE_BIN_EACH(d2u(u2d(x) + u2d(y))); break;
    case OP_SUB_F64: QE_BIN_EACH(d2u(u2d(x) - u2d(y))); break;
    case OP_MUL_F64: QE_BIN_EACH(d2u(u2d(x) * u2d(y))); break;
    case OP_EQ_I64: QE_BIN_EACH((uint64_t)(x == y)); break;
    case OP_NE_I64: QE_BIN_EACH((uint64_t)(x != y)); break;
    case OP_LT_I64: QE_BIN_EACH((uint64_t)((int64_t)x < (int64_t)y)); break;
    case OP_LE_I64: QE_BIN_EACH((uint64_t)((int64_t)x <= (int64_t)y)); break;
    case OP_GT_I64: QE_BIN_EACH((uint64_t)((int64_t)x > (int64_t)y)); break;
    case OP_GE_I64: QE_BIN_EACH((uint64_t)((int64_t)x >= (int64_t)y)); break;
    case OP_EQ_F64: QE_BIN_EACH((uint64_t)(u2d(x) == u2d(y))); break;
    case OP_NE_F64: QE_BIN_EACH((uint64_t)(u2d(x) != u2d(y))); break;
    case OP_LT_F64: QE_BIN_EACH((uint64_t)(u2d(x) < u2d(y))); break;
    case OP_LE_F64: QE_BIN_EACH((uint64_t)(u2d(x) <= u2d(y))); break;
    case OP_GT_F64: QE_BIN_EACH((uint64_t)(u2d(x) > u2d(y))); break;
    case OP_GE_F64: QE_BIN_EACH((uint64_t)(u2d(x) >= u2d(y))); break;
    case OP_AND: QE_BIN_EACH(x & y); break;
    default: QE_BIN_EACH(x | y); break;  // OP_OR
  }
  a.valid &= b.valid;
}

__device__ __forceinline__ bool op_is_fast_push(uint32_t op) { return op >= OP_PUSH_CACHED64 || op == OP_PUSH_LIT; }
__device__ __forceinline__ bool op_is_fast_binary(uint32_t op) {
  return (op >= OP_ADD_I32 && op <= OP_MUL_I32) || (op >= OP_ADD_I64 && op <= OP_MUL_I64) ||
         (op >= OP_ADD_F64 && op <= OP_MUL_F64) || (op >= OP_EQ_I64 && op <= OP_OR);
}

// Everything else: one copy for all stack depths.  Operands are copied into the temporaries
// a / b, the result goes back through slot_put.  `d` is the stack depth before the instruction.
__device__ __forceinline__ void vm_slow(uint32_t op, uint32_t d, uint32_t arg, const VmProgram &P, const RowCtx &c0,
                                        const ColCache &K, Slot (&S)[VM_MAX_DEPTH]) {
  RowCtx c;
  c.row0 = c0.row0;
  c.n = c0.n;
  c.active = c0.active;
  c.lane = opaque(c0.lane);
  const bool is_push = (op <= OP_PUSH_ROWID) || (op >= OP_UTF8_EQ_LIT && op <= OP_UTF8_NE_COL);
  const bool is_unary = op == OP_I64_TO_F64;
  Slot a, b;
  uint32_t dst;
  if (is_push) {
    dst = d;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) a.v[j] = 0, b.v[j] = 0;
    a.valid = b.valid = 0xFu;
  } else if (is_unary) {
    dst = d - 1;
    slot_get(S, d - 1, a);
    slot_copy(b, a);
  } else {
    dst = d - 2;
    slot_get(S, d - 2, a);
    slot_get(S, d - 1, b);
  }
  switch (op) {
    case OP_PUSH_COL64:
    case OP_PUSH_COL32: {
      // columns beyond the cache: one round trip per use
      const VmCol &col = P.cols[arg];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool on = ((c.active >> (2 * h)) & 3u) != 0;
        const int64_t r = c.row0 + h * 128 + 2 * (int64_t)c.lane;
        uint64_t x = 0, y = 0;
        if (on) {
          if (op == OP_PUSH_COL64) {
            U64x2 t = *(const U64x2 *)((const uint64_t *)col.values + r);
            x = t.x;
            y = t.y;
          } else {
            U32x2 t = *(const U32x2 *)((const uint32_t *)col.values + r);
            x = (uint64_t)(int64_t)(int32_t)t.x;
            y = (uint64_t)(int64_t)(int32_t)t.y;
          }
        }
        a.v[2 * h] = x;
        a.v[2 * h + 1] = y;
      }
      a.valid = col.validity ? load_bits4(col.validity, c) : 0xFu;
      break;
    }
    case OP_PUSH_COLBOOL: {
      const VmCol &col = P.cols[arg];
      uint32_t bits = load_bits4((const uint8_t *)col.values, c);
#pragma unroll
      for (int j = 0; j < VM_R; ++j) a.v[j] = (bits >> j) & 1u;
      a.valid = col.validity ? load_bits4(col.validity, c) : 0xFu;
      break;
    }
    case OP_PUSH_ROWID: {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) a.v[j] = (uint64_t)(c.row0 + (j >> 1) * 128 + 2 * (int64_t)c.lane + (j & 1));
      break;
    }
    case OP_UTF8_EQ_LIT:
    case OP_UTF8_NE_LIT:
    case OP_UTF8_EQ_COL:
    case OP_UTF8_NE_COL: {
      const bool vs_lit = op == OP_UTF8_EQ_LIT || op == OP_UTF8_NE_LIT;
      const bool want_eq = op == OP_UTF8_EQ_LIT || op == OP_UTF8_EQ_COL;
      const VmCol &ca = P.cols[arg & 0xffu];
      const uint32_t second = (arg >> 8) & 0xffu;
      const uint8_t *lit = vs_lit ? (const uint8_t *)P.lits[second] : nullptr;
      const int32_t llen = vs_lit ? (int32_t)P.lits[second + 1] : 0;
      const VmCol &cb = P.cols[vs_lit ? 0 : second];
      uint32_t act = c.active;
      // rolled over the four rows: row j = (it>>1)*128 + (it&1) ahead of the lane's first row
#pragma unroll 1
      for (int it = 0; it < VM_R; ++it) {
        uint64_t res = 0;
        if (act & 1u) {
          int64_t r = c.row0 + (it >> 1) * 128 + 2 * (int64_t)c.lane + (it & 1);
          int32_t a0 = ca.offsets[r], a1 = ca.offsets[r + 1];
          bool eq;
          if (vs_lit) {
            eq = (a1 - a0 == llen) && utf8_equal((const uint8_t *)ca.values + a0, lit, llen);
          } else {
            int32_t b0 = cb.offsets[r], b1 = cb.offsets[r + 1];
            eq = (a1 - a0 == b1 - b0) && utf8_equal((const uint8_t *)ca.values + a0, (const uint8_t *)cb.values + b0, a1 - a0);
          }
          res = (eq == want_eq) ? 1u : 0u;
        }
        a.v[0] = res;
        QE_ROT4(a.v);
        act = rot4(act);
      }
      uint32_t va = ca.validity ? load_bits4(ca.validity, c) : 0xFu;
      uint32_t vb = (!vs_lit && cb.validity) ? load_bits4(cb.validity, c) : 0xFu;
      a.valid = va & vb;
      break;
    }
    case OP_SWAP: {
      slot_put(S, d - 1, a);
      slot_copy(a, b);
      break;
    }
    case OP_I64_TO_F64: {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) a.v[j] = d2u((double)(int64_t)a.v[j]);
      break;
    }
    case OP_DIV_F64: {
#pragma unroll 1
      for (int it = 0; it < VM_R; ++it) {
        a.v[0] = d2u(u2d(a.v[0]) / u2d(b.v[0]));
        QE_ROT4(a.v);
        QE_ROT4(b.v);
      }
      a.valid &= b.valid;
      break;
    }
    case OP_DIV_I32:
    case OP_MOD_I32:
    case OP_DIV_I64:
    case OP_MOD_I64: {
      // int32 operands are sign-extended, so one 64-bit division serves both widths; only the
      // overflow case (MIN / -1) depends on the width.  x/0, x%0 and MIN/-1 panic in the
      // reference; here they give null.
      const bool w32 = op == OP_DIV_I32 || op == OP_MOD_I32;
      const bool want_mod = op == OP_MOD_I32 || op == OP_MOD_I64;
      const int64_t tmin = w32 ? (int64_t)INT32_MIN : INT64_MIN;
      uint32_t valid = a.valid & b.valid;
#pragma unroll 1
      for (int it = 0; it < VM_R; ++it) {
        int64_t x = (int64_t)a.v[0], y = (int64_t)b.v[0];
        int64_t r = 0;
        if (y == 0 || (x == tmin && y == -1)) {
          valid &= ~1u;
        } else {
          int64_t q = x / y;
          r = want_mod ? x - q * y : q;
        }
        a.v[0] = (uint64_t)r;
        QE_ROT4(a.v);
        QE_ROT4(b.v);
        valid = rot4(valid);
      }
      a.valid = valid;
      break;
    }
    default:
      break;
  }
  slot_put(S, dst, a);
}

// Runs the postfix code starting at `pc` until OP_END; the value is left in S[0].
// Control flow depends only on the (wave-uniform) instruction words.  FULL = false leaves out
// vm_slow (division, utf8, uncached columns, casts): the host picks that variant when a program
// has none of those, and it needs far fewer registers.
template <bool FULL>
__device__ __forceinline__ void vm_eval(const VmProgram &P, const ProgRegs &R, uint32_t pc, const RowCtx &c,
                                        const ColCache &K, Slot (&S)[VM_MAX_DEPTH]) {
  for (;;) {
    uint32_t ins = fetch_insn(R, pc++);
    uint32_t op = ins & 0xffu;
    if (op == OP_END) break;
    uint32_t d = (ins >> 8) & 0xfu;
    uint32_t arg = ins >> 12;
    if (op_is_fast_binary(op)) {
      switch (d) {
        case 2: vm_binary_fast(op, S[0], S[1]); break;
        case 3: vm_binary_fast(op, S[1], S[2]); break;
        default: vm_binary_fast(op, S[2], S[3]); break;
      }
    } else if (op_is_fast_push(op)) {
      switch (d) {
        case 0: vm_push_fast(op, arg, R, K, S[0]); break;
        case 1: vm_push_fast(op, arg, R, K, S[1]); break;
        case 2: vm_push_fast(op, arg, R, K, S[2]); break;
        default: vm_push_fast(op, arg, R, K, S[3]); break;
      }
    } else if constexpr (FULL) {
      vm_slow(op, d, arg, P, c, K, S);
    }
  }
}

__device__ __forceinline__ uint32_t bool_bits(const Slot &s) {
  uint32_t b = 0;
#pragma unroll
  for (int j = 0; j < VM_R; ++j) b |= ((uint32_t)s.v[j] & 1u) << j;
  return b;
}

// ------------------------------------------------------------ dense stores
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
  x &= 0xffffu;
  x = (x | (x << 8)) & 0x00ff00ffu;
  x = (x | (x << 4)) & 0x0f0f0f0fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

// Writes the 256 bits of a wave-step (bit j of `bits4` = row j of the lane) as 8 dwords.
__device__ __forceinline__ void store_bitmap256(uint32_t *words, const RowCtx &c, uint32_t bits4) {
  uint64_t b0 = __ballot((bits4 & 1u) != 0), b1 = __ballot((bits4 & 2u) != 0);
  uint64_t b2 = __ballot((bits4 & 4u) != 0), b3 = __ballot((bits4 & 8u) != 0);
  if (c.lane < 8) {
    uint32_t h = c.lane >> 2, q = c.lane & 3u;
    uint64_t ev = h ? b2 : b0, od = h ? b3 : b1;
    uint32_t w = spread16((uint32_t)(ev >> (16 * q))) | (spread16((uint32_t)(od >> (16 * q))) << 1);
    int64_t idx = (c.row0 >> 5) + c.lane;
    if ((idx << 5) < c.n) words[idx] = w;
  }
}

// Output buffers are allocated by the library with slack, so a row pair is always one store.
__device__ __forceinline__ void store_dense(const VmOut &o, const RowCtx &c, const Slot &s) {
  uint32_t inr = in_range_bits(c.row0, c.lane, c.n);
  if (o.type == VT_BOOL) {
    store_bitmap256((uint32_t *)o.values, c, bool_bits(s) & inr);
  } else if (o.type == VT_I32) {
    uint32_t *p = (uint32_t *)o.values;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t r = c.row0 + h * 128 + 2 * (int64_t)c.lane;
      if (r < c.n) {
        U32x2 t;
        t.x = (uint32_t)s.v[2 * h];
        t.y = (uint32_t)s.v[2 * h + 1];
        *(U32x2 *)(p + r) = t;
      }
    }
  } else {
    uint64_t *p = (uint64_t *)o.values;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t r = c.row0 + h * 128 + 2 * (int64_t)c.lane;
      if (r < c.n) {
        U64x2 t;
        t.x = s.v[2 * h];
        t.y = s.v[2 * h + 1];
        *(U64x2 *)(p + r) = t;
      }
    }
  }
  if (o.validity) store_bitmap256(o.validity, c, s.valid & inr);
}

// --------------------------------------------------------- projection kernel
template <bool FULL>
__global__ __launch_bounds__(VM_BLOCK_THREADS, FULL ? 2 : 3) void qe_project_kernel(const ProjectArgs A) {
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const int64_t n = A.n_rows;
  const int64_t nsteps = (n + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  const ProgRegs R = prog_regs(A.prog, lane);
  for (int64_t ws = (int64_t)blockIdx.x * VM_BLOCK_WAVES + wave; ws < nsteps;
       ws += (int64_t)gridDim.x * VM_BLOCK_WAVES) {
    RowCtx c;
    c.row0 = ws * VM_WAVE_ROWS;
    c.n = n;
    c.lane = lane;
    c.active = in_range_bits(c.row0, lane, n);
    ColCache K;
    cache_load(A.prog, (1u << A.prog.cache_n) - 1u, c, K);
#pragma unroll 1
    for (int e = 0; e < A.n_outs; ++e) {
      Slot S[VM_MAX_DEPTH];
      vm_eval<FULL>(A.prog, R, A.outs[e].pc, c, K, S);
      store_dense(A.outs[e], c, S[0]);
    }
  }
}

// ------------------------------------------------------------ filter kernel
constexpr unsigned long long ST_FLAG_AGG = 1ull << 62;
constexpr unsigned long long ST_FLAG_PREFIX = 2ull << 62;
constexpr unsigned long long ST_VALUE_MASK = (1ull << 62) - 1;
constexpr int LOOKBACK_SPIN_LIMIT = 1 << 22;
static_assert(FILTER_STEPS == 4, "the filter kernel selects among four per-step prefixes");

__device__ __forceinline__ unsigned long long ld_status(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Decoupled look-back over the per-tile status words, run by one full wave.
// Each status word is a single 8-byte {flag, value} granule written by one
// agent-scope store, so no further ordering between tiles is needed.  Tiles are
// handed out by a ticket counter, so every predecessor of a running tile is
// itself running or finished: the wait below always ends.
__device__ unsigned long long lookback_exclusive(unsigned long long *status, int64_t tile, unsigned long long total,
                                                 uint32_t lane, unsigned long long *err) {
  if (tile == 0) {
    if (lane == 0) st_status(&status[0], ST_FLAG_PREFIX | total);
    return 0;
  }
  if (lane == 0) st_status(&status[tile], ST_FLAG_AGG | total);
  unsigned long long excl = 0;
  int64_t idx = tile - 1;
  int spins = 0;
  for (;;) {
    int64_t t = idx - (int64_t)lane;
    unsigned long long v = (t >= 0) ? ld_status(&status[t]) : ST_FLAG_PREFIX;
    uint32_t flag = (uint32_t)(v >> 62);
    uint64_t inv = __ballot(flag == 0u);
    uint64_t pre = __ballot(flag == 2u);
    if (pre != 0) {
      int p = __builtin_ctzll(pre);
      uint64_t need = (p == 63) ? ~0ull : ((1ull << (p + 1)) - 1ull);
      if ((inv & need) == 0) {
        unsigned long long contrib = ((int)lane <= p) ? (v & ST_VALUE_MASK) : 0ull;
        excl += wave_sum_u64(contrib);
        break;
      }
    } else if (inv == 0) {
      excl += wave_sum_u64(v & ST_VALUE_MASK);
      idx -= 64;
      continue;
    }
    if (++spins > LOOKBACK_SPIN_LIMIT) {
      if (lane == 0) *err = 1ull;
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) st_status(&status[tile], ST_FLAG_PREFIX | ((excl + total) & ST_VALUE_MASK));
  return excl;
}

__device__ __forceinline__ uint32_t popc_sel(uint32_t sel) {
  return (uint32_t)(__popcll(__ballot((sel & 1u) != 0)) + __popcll(__ballot((sel & 2u) != 0)) +
                    __popcll(__ballot((sel & 4u) != 0)) + __popcll(__ballot((sel & 8u) != 0)));
}

// Output position (relative to the wave-step's base) of each selected row, in row order.
__device__ __forceinline__ void rank_rows(uint32_t sel, uint32_t (&rank)[VM_R]) {
  uint64_t b0 = __ballot((sel & 1u) != 0), b1 = __ballot((sel & 2u) != 0);
  uint64_t b2 = __ballot((sel & 4u) != 0), b3 = __ballot((sel & 8u) != 0);
  uint32_t c0 = (uint32_t)(__popcll(b0) + __popcll(b1));
  uint32_t lt0 = mbcnt64(b0) + mbcnt64(b1);
  uint32_t lt1 = mbcnt64(b2) + mbcnt64(b3);
  rank[0] = lt0;
  rank[1] = lt0 + (sel & 1u);
  rank[2] = c0 + lt1;
  rank[3] = c0 + lt1 + ((sel >> 2) & 1u);
}

__device__ __forceinline__ void store_compact(const VmOut &o, unsigned long long base, uint32_t sel,
                                              const uint32_t (&rank)[VM_R], const Slot &s) {
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    if ((sel >> j) & 1u) {
      unsigned long long pos = base + rank[j];
      if (o.type == VT_BOOL) {
        if (s.v[j] & 1u) atomicOr((uint32_t *)o.values + (pos >> 5), 1u << (pos & 31u));
      } else if (o.type == VT_I32) {
        ((uint32_t *)o.values)[pos] = (uint32_t)s.v[j];
      } else {
        ((uint64_t *)o.values)[pos] = s.v[j];
      }
      if (o.validity && ((s.valid >> j) & 1u)) atomicOr(o.validity + (pos >> 5), 1u << (pos & 31u));
    }
  }
}

// One ticket = one tile of 4096 rows = 4 steps of 1024.  Sub-steps 0..3 evaluate the predicate
// of each step and keep the 16 selection bits of each lane in a register; then the tile's count
// goes through the look-back; sub-steps 4..7 evaluate every output expression for the selected
// rows only and write it at its final position.  Rows keep their input order (the reference's
// filter_chunk is order preserving, tests/mod.rs:138-188); a row whose predicate is null is
// dropped.  Both phases run through one loop body so that the interpreter exists once.
template <bool FULL>
__global__ __launch_bounds__(VM_BLOCK_THREADS, FULL ? 2 : 3) void qe_filter_kernel(const FilterArgs A) {
  __shared__ unsigned long long s_tile;
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_cnt[FILTER_STEPS * VM_BLOCK_WAVES];
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const int64_t n = A.n_rows;
  unsigned long long *status = (unsigned long long *)(A.scratch + 1);
  const ProgRegs R = prog_regs(A.prog, lane);

  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&A.scratch->ticket, 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)s_tile;
    if (tile >= A.n_tiles) break;
    const int64_t tile_row0 = tile * FILTER_TILE_ROWS;

    uint32_t mask = 0;
    uint32_t pref[FILTER_STEPS] = {0, 0, 0, 0};
    unsigned long long base = 0;
#pragma unroll 1
    for (int u = 0; u < 2 * FILTER_STEPS; ++u) {
      const bool phase2 = u >= FILTER_STEPS;
      const int s = u & (FILTER_STEPS - 1);
      if (u == FILTER_STEPS) {
        // all four counts of every wave are in: prefix inside the tile, then across tiles
        __syncthreads();
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < FILTER_STEPS * VM_BLOCK_WAVES; ++i) {
          if ((uint32_t)(i & (VM_BLOCK_WAVES - 1)) == wave) pref[i / VM_BLOCK_WAVES] = run;
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
        base = s_base;
      }
      RowCtx c;
      c.row0 = tile_row0 + s * VM_BLOCK_ROWS + wave * VM_WAVE_ROWS;
      c.n = n;
      c.lane = lane;
      c.active = phase2 ? ((mask >> (4 * s)) & 0xFu) : in_range_bits(c.row0, lane, n);
      const bool any = __ballot(c.active != 0) != 0;
      uint32_t sel = 0;
      if (any) {
        // phase 2 loads for the selected rows only: unselected lanes issue no load
        ColCache K;
        cache_load(A.prog, phase2 ? A.out_cache_mask : A.pred_cache_mask, c, K);
        uint32_t rank[VM_R];
        if (phase2) rank_rows(c.active, rank);
        const unsigned long long obase = base + (s == 0 ? pref[0] : s == 1 ? pref[1] : s == 2 ? pref[2] : pref[3]);
        const int ntask = phase2 ? A.n_outs : 1;
#pragma unroll 1
        for (int e = 0; e < ntask; ++e) {
          Slot S[VM_MAX_DEPTH];
          vm_eval<FULL>(A.prog, R, phase2 ? A.outs[e].pc : A.pred_pc, c, K, S);
          if (phase2)
            store_compact(A.outs[e], obase, c.active, rank, S[0]);
          else
            sel = c.active & S[0].valid & bool_bits(S[0]);
        }
      }
      if (!phase2) {
        mask |= sel << (4 * s);
        uint32_t cnt = popc_sel(sel);
        if (lane == 0) s_cnt[s * VM_BLOCK_WAVES + wave] = cnt;
      }
    }
  }
}

// --------------------------------------------------------- aggregate kernel
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

// Order-preserving unsigned image: min/max of the images = image of the min/max.
// float64: -0.0 < +0.0 and NaN never reaches here (NaN inputs are skipped).
__device__ __forceinline__ uint64_t ord_image(uint64_t bits, uint32_t type) {
  if (type == VT_F64) return (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
  return bits ^ 0x8000000000000000ull;
}
__device__ __forceinline__ uint64_t ord_inverse(uint64_t img, uint32_t type) {
  if (type == VT_F64) return (img >> 63) ? (img & 0x7fffffffffffffffull) : ~img;
  return img ^ 0x8000000000000000ull;
}

__device__ __forceinline__ uint64_t acc_identity(uint32_t op) { return op == ACC_MIN ? ~0ull : 0ull; }

constexpr int32_t SLOT_NONE = INT32_MIN;
constexpr int LDS_MAX_PROBE = 8;

struct LdsTable {
  unsigned long long *keys;  // [cap + 1]
  unsigned long long *acc;   // [n_accs][cap + 1]
  uint32_t *rowcnt;          // [cap + 1]
  uint32_t *nn;              // [n_accs][cap + 1]
  uint32_t *used;            // claimed slots
  uint32_t cap;
  uint32_t stride;           // cap + 1
  uint32_t limit;
};

__device__ __forceinline__ int64_t global_find_or_claim(const AggTable &T, uint64_t key, uint64_t h) {
  if (key == AGG_EMPTY_KEY) {
    // the one key that collides with the empty marker lives in the reserved slot
    unsigned long long old = atomicCAS(&T.keys[T.cap], AGG_EMPTY_KEY, 0ull);
    if (old == AGG_EMPTY_KEY) atomicAdd(&T.ctrl[0], 1ull);
    return (int64_t)T.cap;
  }
  uint64_t mask = T.cap - 1;
  uint64_t s = (h >> 20) & mask;
  for (uint64_t p = 0; p <= mask; ++p) {
    unsigned long long cur = __hip_atomic_load(&T.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return (int64_t)s;
    if (cur == AGG_EMPTY_KEY) {
      unsigned long long old = atomicCAS(&T.keys[s], AGG_EMPTY_KEY, (unsigned long long)key);
      if (old == AGG_EMPTY_KEY) {
        atomicAdd(&T.ctrl[0], 1ull);
        return (int64_t)s;
      }
      if (old == key) return (int64_t)s;
    }
    s = (s + 1) & mask;
  }
  atomicExch(&T.ctrl[1], 1ull);
  return -1;
}

// Cheap 32-bit hash for the LDS table (the global table uses the full mix64).
__device__ __forceinline__ uint32_t hash32(uint64_t key) {
  uint32_t h = (uint32_t)key * 0x9E3779B1u ^ (uint32_t)(key >> 32) * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 13;
  return h;
}

// LDS slot of `key` (claiming one if new), or -1 when the key does not fit the LDS table.
__device__ __forceinline__ int32_t lds_find_or_claim(const LdsTable &L, uint64_t key) {
  if (key == AGG_EMPTY_KEY) {
    (void)atomicCAS(&L.keys[L.cap], AGG_EMPTY_KEY, 0ull);
    return (int32_t)L.cap;
  }
  uint32_t s = hash32(key) & (L.cap - 1);
  for (int p = 0; p < LDS_MAX_PROBE; ++p) {
    unsigned long long cur = *(volatile unsigned long long *)&L.keys[s];
    if (cur == key) return (int32_t)s;
    if (cur == AGG_EMPTY_KEY) {
      if (*(volatile uint32_t *)L.used >= L.limit) break;
      unsigned long long old = atomicCAS(&L.keys[s], AGG_EMPTY_KEY, (unsigned long long)key);
      if (old == AGG_EMPTY_KEY) {
        atomicAdd(L.used, 1u);
        return (int32_t)s;
      }
      if (old == key) return (int32_t)s;
    }
    s = (s + 1) & (L.cap - 1);
  }
  return -1;
}

__device__ __forceinline__ void acc_apply(unsigned long long *p, uint32_t op, uint64_t v) {
  switch (op) {
    case ACC_SUM_F64: atomicAdd((double *)p, u2d(v)); break;
    case ACC_MIN: atomicMin(p, (unsigned long long)v); break;
    case ACC_MAX: atomicMax(p, (unsigned long long)v); break;
    default: atomicAdd(p, (unsigned long long)v); break;  // ACC_SUM_I64, ACC_COUNT
  }
}

__device__ __forceinline__ uint64_t acc_combine(uint32_t op, uint64_t a, uint64_t b) {
  switch (op) {
    case ACC_SUM_F64: return d2u(u2d(a) + u2d(b));
    case ACC_MIN: return a < b ? a : b;
    case ACC_MAX: return a > b ? a : b;
    default: return a + b;
  }
}

__device__ __forceinline__ uint64_t wave_reduce_acc(uint32_t op, uint64_t v) {
#pragma unroll 1
  for (int o = 32; o > 0; o >>= 1) v = acc_combine(op, v, (uint64_t)__shfl_xor((unsigned long long)v, o));
  return v;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

constexpr uint32_t LEADER_MIN_ROWS = 24;
constexpr int LEADER_TRIES = 2;
constexpr int32_t SLOT_LDS_MISS = -1;  // transient: the row still needs a global slot

// Each workgroup folds its rows into an LDS hash table (keys claimed by 64-bit LDS compare-swap,
// accumulators updated by LDS atomics) and flushes the table into the global open-addressing
// table once, at the end.  Keys that do not fit the LDS table go straight to the global table.
// A key that dominates a wave-step (low cardinality, skew) is reduced across the wave first, so
// that it costs one atomic per wave-step instead of one per row.
//
// Slot encoding per row: >= 0 LDS slot; <= -2: global slot g stored as -2 - g; SLOT_NONE: dropped.
template <bool FULL>
__global__ __launch_bounds__(VM_BLOCK_THREADS, FULL ? 2 : 3) void qe_aggregate_kernel(const AggArgs A) {
  extern __shared__ unsigned long long lds_raw[];
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const int64_t n = A.n_rows;
  const int n_accs = A.n_accs;

  LdsTable L;
  L.cap = A.lds_cap;
  L.stride = A.lds_cap + 1;
  L.limit = A.lds_limit;
  L.keys = lds_raw;
  L.acc = L.keys + L.stride;
  L.rowcnt = (uint32_t *)(L.acc + (size_t)n_accs * L.stride);
  L.nn = L.rowcnt + L.stride;
  L.used = L.nn + (size_t)n_accs * L.stride;

  for (uint32_t i = threadIdx.x; i < L.stride; i += blockDim.x) {
    L.keys[i] = AGG_EMPTY_KEY;
    L.rowcnt[i] = 0;
    for (int a = 0; a < n_accs; ++a) {
      L.acc[a * L.stride + i] = acc_identity(A.accs[a].op);
      L.nn[a * L.stride + i] = 0;
    }
  }
  if (threadIdx.x == 0) *L.used = 0;
  __syncthreads();

  // wave-uniform descriptors, one per lane (see ProgRegs): task list, key fields, accumulators
  const ProgRegs R = prog_regs(A.prog, lane);
  const uint32_t tl = lane < VM_MAX_TASKS ? lane : 0;
  const uint32_t task_pc = A.tasks[tl].pc;
  const uint32_t task_w = (uint32_t)A.tasks[tl].role | ((uint32_t)A.tasks[tl].idx << 8) | ((uint32_t)A.tasks[tl].has_eval << 16);
  const uint32_t key_w = (uint32_t)A.keys[lane & (VM_MAX_KEYS - 1)].shift | ((uint32_t)A.keys[lane & (VM_MAX_KEYS - 1)].bits << 8);
  const uint32_t acc_w = (uint32_t)A.accs[lane & (VM_MAX_ACCS - 1)].op | ((uint32_t)A.accs[lane & (VM_MAX_ACCS - 1)].in_type << 8) |
                         ((uint32_t)A.accs[lane & (VM_MAX_ACCS - 1)].track_nn << 16);

  const int64_t nsteps = (n + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  for (int64_t ws = (int64_t)blockIdx.x * VM_BLOCK_WAVES + wave; ws < nsteps;
       ws += (int64_t)gridDim.x * VM_BLOCK_WAVES) {
    RowCtx c;
    c.row0 = ws * VM_WAVE_ROWS;
    c.n = n;
    c.lane = lane;
    c.active = in_range_bits(c.row0, lane, n);
    ColCache K;
    cache_load(A.prog, (1u << A.prog.cache_n) - 1u, c, K);

    uint64_t key[VM_R] = {0, 0, 0, 0};
    int32_t slot[VM_R] = {SLOT_NONE, SLOT_NONE, SLOT_NONE, SLOT_NONE};
    uint32_t rem = 0;      // rows folded one by one
    uint32_t lead_mb = 0;  // rows folded through the wave reduction
    int32_t lead_slot = SLOT_NONE;

#pragma unroll 1
    for (int t = 0; t < A.n_tasks; ++t) {
      const uint32_t tw = (uint32_t)__builtin_amdgcn_readlane((int)task_w, t);
      const uint32_t role = tw & 0xffu;
      const uint32_t idx = (tw >> 8) & 0xffu;
      Slot S[VM_MAX_DEPTH];
      S[0].valid = 0xFu;
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[0].v[j] = 1;
      if ((tw >> 16) & 1u) vm_eval<FULL>(A.prog, R, (uint32_t)__builtin_amdgcn_readlane((int)task_pc, t), c, K, S);

      if (role == ROLE_PRED) {
        c.active &= S[0].valid & bool_bits(S[0]);
        if (__ballot(c.active != 0) == 0) break;  // nothing of this wave-step survives the filter
      } else if (role == ROLE_KEY) {
        c.active &= S[0].valid;  // a null key drops the row (physical_plan/mod.rs:367)
        const uint32_t kw = (uint32_t)__builtin_amdgcn_readlane((int)key_w, (int)idx);
        const uint32_t kbits = (kw >> 8) & 0xffu;
        const uint64_t fmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
        const uint32_t sh = kw & 0xffu;
#pragma unroll
        for (int j = 0; j < VM_R; ++j) key[j] |= (S[0].v[j] & fmask) << sh;
      } else if (role == ROLE_LOOKUP) {
        rem = c.active;
        // ---- a key that dominates the wave-step?  Try the first row of up to two lanes.
        uint64_t ball = __ballot(rem != 0);
        uint64_t k0 = 0;
        uint32_t lead_cnt = 0;
        int src = 0;
#pragma unroll 1
        for (int tr = 0; tr < LEADER_TRIES && ball != 0; ++tr) {
          src = __builtin_ctzll(ball);
          const uint64_t kl = (rem & 1u) ? key[0] : (rem & 2u) ? key[1] : (rem & 4u) ? key[2] : key[3];
          k0 = readlane64(kl, src);
          uint32_t mb = 0;
#pragma unroll
          for (int j = 0; j < VM_R; ++j)
            if (((rem >> j) & 1u) && key[j] == k0) mb |= 1u << j;
          const uint32_t cnt = popc_sel(mb);
          if (cnt >= LEADER_MIN_ROWS) {
            lead_mb = mb;
            lead_cnt = cnt;
            break;
          }
          ball &= ~__ballot(mb != 0);  // next candidate: a lane none of whose rows matched
        }
        rem &= ~lead_mb;
        // ---- LDS slots: the dominant key (one lane) and the remaining rows, back to back
        int32_t lsl = SLOT_NONE;
        if (lead_cnt != 0 && (int)lane == src) lsl = lds_find_or_claim(L, k0);
#pragma unroll
        for (int j = 0; j < VM_R; ++j) {
          slot[j] = SLOT_NONE;
          if ((rem >> j) & 1u) slot[j] = lds_find_or_claim(L, key[j]);
        }
        // ---- keys the LDS table has no room for go to the global table (rare: rolled loop)
        uint32_t miss = 0;
#pragma unroll
        for (int j = 0; j < VM_R; ++j)
          if (slot[j] == SLOT_LDS_MISS) miss |= 1u << j;
        const bool lead_miss = lsl == SLOT_LDS_MISS;
        if (__ballot(miss != 0 || lead_miss) != 0) {
#pragma unroll 1
          for (int it = 0; it <= VM_R; ++it) {
            const bool lead_it = it == 0;
            const bool act = lead_it ? lead_miss : ((miss & 1u) != 0);
            const uint64_t kcur = lead_it ? k0 : key[0];
            int32_t sl = lead_it ? lsl : slot[0];
            if (act) {
              int64_t g = global_find_or_claim(A.table, kcur, mix64(kcur));
              sl = g < 0 ? SLOT_NONE : (int32_t)(-2 - (int32_t)g);
            }
            if (lead_it) {
              lsl = sl;
            } else {
              slot[0] = sl;
              miss = rot4(miss);
              QE_ROT4(key);
              QE_ROT4(slot);
            }
          }
        }
        // ---- row counts
        if (lead_cnt != 0 && (int)lane == src && lsl != SLOT_NONE) {
          if (lsl >= 0)
            atomicAdd(&L.rowcnt[lsl], lead_cnt);
          else
            atomicAdd(&A.table.rowcnt[(uint32_t)(-2 - lsl)], (unsigned long long)lead_cnt);
        }
        lead_slot = lead_cnt != 0 ? __builtin_amdgcn_readlane(lsl, src) : SLOT_NONE;
        if (lead_slot == SLOT_NONE) lead_mb = 0;
#pragma unroll
        for (int j = 0; j < VM_R; ++j) {
          if ((rem >> j) & 1u) {
            if (slot[j] >= 0)
              atomicAdd(&L.rowcnt[slot[j]], 1u);
            else if (slot[j] != SLOT_NONE)
              atomicAdd(&A.table.rowcnt[(uint32_t)(-2 - slot[j])], 1ull);
            else
              rem &= ~(1u << j);  // the global table is full: the host grows it and runs again
          }
        }
      } else {  // ROLE_ACC
        const uint32_t aw = (uint32_t)__builtin_amdgcn_readlane((int)acc_w, (int)idx);
        const uint32_t op = aw & 0xffu;
        const uint32_t in_type = (aw >> 8) & 0xffu;
        const bool track = ((aw >> 16) & 0xffu) != 0;
        uint32_t valid = S[0].valid;
        uint64_t v[VM_R];
#pragma unroll
        for (int j = 0; j < VM_R; ++j) {
          uint64_t x = S[0].v[j];
          if (op == ACC_MIN || op == ACC_MAX) {
            if (in_type == VT_F64 && u2d(x) != u2d(x)) valid &= ~(1u << j);  // NaN is ignored by min / max
            x = ord_image(x, in_type);
          } else if (op == ACC_COUNT) {
            x = 1;
          }
          v[j] = x;
        }
        unsigned long long *lacc = L.acc + idx * L.stride;
        unsigned long long *gacc = A.table.acc + (uint64_t)idx * A.table.stride;
        // wave-reduced contribution of the dominant key
        if (lead_slot != SLOT_NONE) {
          const uint32_t m = lead_mb & valid;
          uint64_t part = acc_identity(op);
#pragma unroll
          for (int j = 0; j < VM_R; ++j)
            if ((m >> j) & 1u) part = acc_combine(op, part, v[j]);
          part = wave_reduce_acc(op, part);
          const uint32_t part_cnt = popc_sel(m);
          if (lane == 0 && part_cnt != 0) {
            if (lead_slot >= 0) {
              acc_apply(lacc + lead_slot, op, part);
              if (track) atomicAdd(&L.nn[idx * L.stride + lead_slot], part_cnt);
            } else {
              const uint32_t g = (uint32_t)(-2 - lead_slot);
              acc_apply(gacc + g, op, part);
              if (track) atomicAdd(&A.table.nn[(uint64_t)idx * A.table.stride + g], (unsigned long long)part_cnt);
            }
          }
        }
        const uint32_t m = rem & valid;
#pragma unroll
        for (int j = 0; j < VM_R; ++j) {
          if ((m >> j) & 1u) {
            if (slot[j] >= 0) {
              acc_apply(lacc + slot[j], op, v[j]);
              if (track) atomicAdd(&L.nn[idx * L.stride + slot[j]], 1u);
            } else {
              const uint32_t g = (uint32_t)(-2 - slot[j]);
              acc_apply(gacc + g, op, v[j]);
              if (track) atomicAdd(&A.table.nn[(uint64_t)idx * A.table.stride + g], 1ull);
            }
          }
        }
      }
    }
  }

  // ---- flush the workgroup's table into the global table
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < L.stride; i += blockDim.x) {
    unsigned long long k = L.keys[i];
    uint32_t rc = L.rowcnt[i];
    if (i == L.cap) {
      if (k == AGG_EMPTY_KEY) continue;  // reserved slot never claimed
      k = AGG_EMPTY_KEY;
    } else if (k == AGG_EMPTY_KEY) {
      continue;
    }
    int64_t g = global_find_or_claim(A.table, k, mix64(k));
    if (g < 0) continue;
    if (rc) atomicAdd(&A.table.rowcnt[g], (unsigned long long)rc);
    for (int a = 0; a < n_accs; ++a) {
      uint32_t op = A.accs[a].op;
      unsigned long long val = L.acc[a * L.stride + i];
      if (val != acc_identity(op) || op == ACC_SUM_F64) acc_apply(&A.table.acc[(uint64_t)a * A.table.stride + g], op, val);
      uint32_t nn = L.nn[a * L.stride + i];
      if (nn) atomicAdd(&A.table.nn[(uint64_t)a * A.table.stride + g], (unsigned long long)nn);
    }
  }
}

// ------------------------------------------------------ table init/finalize
__global__ void qe_table_init_kernel(AggTable T, uint64_t packed_lo, uint64_t packed_hi, int n_accs) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < T.stride;
       i += (uint64_t)gridDim.x * blockDim.x) {
    T.keys[i] = AGG_EMPTY_KEY;
    T.rowcnt[i] = 0;
    for (int a = 0; a < n_accs; ++a) {
      uint64_t packed = a < 8 ? packed_lo : packed_hi;
      T.acc[(uint64_t)a * T.stride + i] = acc_identity((uint32_t)(packed >> (8 * (a & 7))) & 0xffu);
      T.nn[(uint64_t)a * T.stride + i] = 0;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) T.ctrl[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void qe_finalize_kernel(const FinalizeArgs A) {
  const AggTable &T = A.table;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= T.cap;
       i += (uint64_t)gridDim.x * blockDim.x) {
    unsigned long long k = T.keys[i];
    if (k == AGG_EMPTY_KEY) continue;
    if (i == T.cap) k = AGG_EMPTY_KEY;  // reserved slot holds the marker-valued key
    unsigned long long pos = atomicAdd(A.out_count, 1ull);
    for (int o = 0; o < A.n_outs; ++o) {
      const FinOut &f = A.outs[o];
      uint64_t val = 0;
      bool valid = true;
      unsigned long long cnt = f.cnt_src == CNT_ROWS  ? T.rowcnt[i]
                               : f.cnt_src == CNT_NN ? T.nn[(uint64_t)f.b * T.stride + i]
                                                     : T.acc[(uint64_t)f.b * T.stride + i];
      if (f.null_mode == NULL_IF_COUNT_ZERO && cnt == 0) valid = false;
      if (f.null_mode == NULL_IF_ACC_ALL_ONES && T.acc[(uint64_t)f.a * T.stride + i] == ~0ull) valid = false;
      if (f.null_mode == NULL_IF_ACC_ZERO && T.acc[(uint64_t)f.a * T.stride + i] == 0ull) valid = false;
      switch (f.kind) {
        case FIN_KEY: {
          uint64_t fm = f.bits >= 64 ? ~0ull : ((1ull << f.bits) - 1ull);
          val = (k >> f.shift) & fm;
          if (f.type == VT_I32) val = (uint64_t)(int64_t)(int32_t)(uint32_t)val;
          break;
        }
        case FIN_ACC_I64:
        case FIN_ACC_F64:
          val = T.acc[(uint64_t)f.a * T.stride + i];
          break;
        case FIN_ACC_ORD:
          val = valid ? ord_inverse(T.acc[(uint64_t)f.a * T.stride + i], f.type) : 0;
          break;
        case FIN_COUNT:
          val = cnt;
          break;
        case FIN_AVG:
          val = valid ? d2u(u2d(T.acc[(uint64_t)f.a * T.stride + i]) / (double)cnt) : 0;
          break;
        default:
          break;
      }
      if (f.type == VT_I32)
        ((uint32_t *)f.values)[pos] = (uint32_t)val;
      else if (f.type == VT_BOOL) {
        if (val & 1u) atomicOr((uint32_t *)f.values + (pos >> 5), 1u << (pos & 31u));
      } else
        ((uint64_t *)f.values)[pos] = val;
      if (f.validity && valid) atomicOr(f.validity + (pos >> 5), 1u << (pos & 31u));
    }
  }
}

// ------------------------------------------------------------- utf8 helpers
__global__ void qe_utf8_gather_lengths_kernel(const int32_t *offsets, const uint8_t *validity, const int64_t *idx,
                                              int64_t m, int32_t *out_lengths, uint32_t *out_validity) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx[i];
    out_lengths[i] = offsets[r + 1] - offsets[r];
    if (out_validity && ((validity[r >> 3] >> (r & 7)) & 1u)) atomicOr(out_validity + (i >> 5), 1u << (i & 31));
  }
}

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// In-place exclusive scan of one tile per block; tile sums go to block_sums.
__global__ __launch_bounds__(SCAN_BLOCK) void qe_scan_tiles_kernel(int32_t *data, int64_t n, int32_t *block_sums) {
  __shared__ int32_t s_wave[SCAN_BLOCK / 64];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int32_t sum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? data[base + i] : 0;
    sum += v[i];
  }
  int32_t incl = sum;
  const uint32_t lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int32_t t = __shfl_up(incl, o);
    if ((int)lane >= o) incl += t;
  }
  uint32_t wave = threadIdx.x >> 6;
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int32_t woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
    if ((uint32_t)w < wave) woff += s_wave[w];
    total += s_wave[w];
  }
  int32_t run = woff + incl - sum;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) data[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 0 && block_sums) block_sums[blockIdx.x] = total;
}

__global__ void qe_scan_add_kernel(int32_t *data, int64_t n, const int32_t *block_offsets) {
  int64_t i = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x;
  int32_t off = block_offsets[blockIdx.x];
  for (int k = 0; k < SCAN_ITEMS; ++k, i += SCAN_BLOCK)
    if (i < n) data[i] += off;
}

__global__ void qe_utf8_gather_bytes_kernel(const int32_t *src_offsets, const uint8_t *src_bytes, const int64_t *idx,
                                            int64_t m, const int32_t *dst_offsets, uint8_t *dst_bytes) {
  // one wave per 64 rows; each row's bytes copied by its lane
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx[i];
    int32_t s0 = src_offsets[r], len = src_offsets[r + 1] - s0;
    int32_t d0 = dst_offsets[i];
    for (int32_t b = 0; b < len; ++b) dst_bytes[d0 + b] = src_bytes[s0 + b];
  }
}

__global__ void qe_utf8_broadcast_kernel(const uint8_t *lit, int32_t len, int64_t n, int32_t *out_offsets,
                                         uint8_t *out_bytes) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
    out_offsets[i] = (int32_t)(i * len);
    if (i < n)
      for (int32_t b = 0; b < len; ++b) out_bytes[i * len + b] = lit[b];
  }
}

__global__ void qe_count_zero_bits_kernel(const uint32_t *bitmap, int64_t n, unsigned long long *out) {
  int64_t nwords = (n + 31) / 32;
  unsigned long long local = 0;
  for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = bitmap[w];
    int64_t rows = n - w * 32;
    uint32_t live = rows >= 32 ? 0xffffffffu : ((1u << rows) - 1u);
    local += __popc(~x & live);
  }
  local = wave_sum_u64(local);
  if (lane_id() == 0 && local) atomicAdd(out, local);
}

// ---------------------------------------------------------- synthetic data
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

__global__ void qe_generate_kernel(int kind, uint64_t seed, int64_t row0, int64_t n, double lo, double hi, void *out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)(row0 + i)));
    double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
    if (kind == 0) {
      ((double *)out)[i] = lo + u * (hi - lo);
    } else if (kind == 1) {
      int64_t a = (int64_t)lo, b = (int64_t)hi;
      uint64_t range = (uint64_t)(b - a);
      ((int64_t *)out)[i] = a + (int64_t)(range ? r % range : 0);
    } else if (kind == 2) {
      // power-law rank over {0..N-1} with exponent s = hi (> 1): inverse CDF of the continuous law
      double N = lo, s = hi;
      double e = 1.0 - s;
      double x = pow(1.0 + u * (pow(N + 1.0, e) - 1.0), 1.0 / e);
      int64_t rank = (int64_t)x - 1;
      if (rank < 0) rank = 0;
      if (rank >= (int64_t)N) rank = (int64_t)N - 1;
      // spread the ranks over the int64 domain with a bijection so hot keys are not neighbours
      ((int64_t *)out)[i] = (int64_t)splitmix64((uint64_t)rank ^ 0x5851f42d4c957f2dull);
    } else {
      // passenger_count style key over {0..9}: mode 1 (~70 %), then 2, 5, 3, 6, 4, 0, 7, 8, 9
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

// ---------------------------------------------------------------- launchers
// Does the program use an instruction only the FULL interpreter variant carries?
static bool program_needs_full(const VmProgram &P) {
  for (int i = 0; i < VM_MAX_INSN; ++i) {
    uint32_t op = P.insn[i] & 0xffu;
    bool fast = op == OP_END || op == OP_PUSH_LIT || op >= OP_PUSH_CACHED64 || (op >= OP_ADD_I32 && op <= OP_MUL_I32) ||
                (op >= OP_ADD_I64 && op <= OP_MUL_I64) || (op >= OP_ADD_F64 && op <= OP_MUL_F64) ||
                (op >= OP_EQ_I64 && op <= OP_OR);
    if (!fast) return true;
  }
  return false;
}

static int grid_for(int64_t work_items, int per_block, int max_blocks) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

hipError_t launch_project(const ProjectArgs &args, const DeviceInfo &dev, hipStream_t stream) {
  if (args.n_rows <= 0 || args.n_outs <= 0) return hipSuccess;
  int grid = grid_for(args.n_rows, VM_BLOCK_ROWS, dev.cus * 8);
  if (program_needs_full(args.prog))
    hipLaunchKernelGGL(qe_project_kernel<true>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, stream, args);
  else
    hipLaunchKernelGGL(qe_project_kernel<false>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, stream, args);
  return hipGetLastError();
}

int64_t filter_num_tiles(int64_t n_rows) { return (n_rows + FILTER_TILE_ROWS - 1) / FILTER_TILE_ROWS; }
size_t filter_scratch_bytes(int64_t n_rows) {
  return sizeof(FilterScratch) + (size_t)filter_num_tiles(n_rows) * sizeof(unsigned long long);
}

hipError_t launch_filter_reset(FilterScratch *scratch, int64_t n_rows, hipStream_t stream) {
  return hipMemsetAsync(scratch, 0, filter_scratch_bytes(n_rows), stream);
}

hipError_t launch_filter(const FilterArgs &args, const DeviceInfo &dev, hipStream_t stream) {
  if (args.n_rows <= 0) return hipSuccess;
  int grid = grid_for(args.n_tiles, 1, dev.cus * 8);
  if (program_needs_full(args.prog))
    hipLaunchKernelGGL(qe_filter_kernel<true>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, stream, args);
  else
    hipLaunchKernelGGL(qe_filter_kernel<false>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, stream, args);
  return hipGetLastError();
}

static size_t agg_lds_bytes(int n_accs, uint32_t slots) {
  size_t stride = (size_t)slots + 1;
  return stride * 8 + stride * 8 * n_accs + stride * 4 + stride * 4 * n_accs + 16;
}

uint32_t aggregate_lds_slots(int n_accs, const DeviceInfo &dev) {
  size_t budget = 64 * 1024;  // two workgroups per CU keep 128 of the 160 KiB
  if ((size_t)dev.max_lds < budget) budget = dev.max_lds;
  uint32_t slots = 4096;
  while (slots > 16 && agg_lds_bytes(n_accs, slots) > budget) slots >>= 1;
  return slots;
}

hipError_t launch_aggregate(const AggArgs &args, const DeviceInfo &dev, hipStream_t stream) {
  if (args.n_rows <= 0) return hipSuccess;
  size_t lds = agg_lds_bytes(args.n_accs, args.lds_cap);
  int grid = grid_for(args.n_rows, VM_BLOCK_ROWS, dev.cus * 2);
  if (program_needs_full(args.prog))
    hipLaunchKernelGGL(qe_aggregate_kernel<true>, dim3(grid), dim3(VM_BLOCK_THREADS), lds, stream, args);
  else
    hipLaunchKernelGGL(qe_aggregate_kernel<false>, dim3(grid), dim3(VM_BLOCK_THREADS), lds, stream, args);
  return hipGetLastError();
}

hipError_t launch_table_init(const AggTable &table, const VmAcc *accs, int n_accs, hipStream_t stream) {
  static_assert(VM_MAX_ACCS <= 16, "two packed words hold 16 accumulator ops");
  uint64_t packed[2] = {0, 0};
  for (int i = 0; i < n_accs && i < VM_MAX_ACCS; ++i) packed[i >> 3] |= (uint64_t)accs[i].op << (8 * (i & 7));
  int grid = grid_for((int64_t)table.stride, 256, 4096);
  hipLaunchKernelGGL(qe_table_init_kernel, dim3(grid), dim3(256), 0, stream, table, packed[0], packed[1], n_accs);
  return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeArgs &args, hipStream_t stream) {
  int grid = grid_for((int64_t)args.table.cap + 1, 256, 4096);
  hipLaunchKernelGGL(qe_finalize_kernel, dim3(grid), dim3(256), 0, stream, args);
  return hipGetLastError();
}

hipError_t launch_utf8_gather_lengths(const int32_t *offsets, const uint8_t *validity, const int64_t *idx, int64_t m,
                                      int32_t *out_lengths, uint32_t *out_validity, hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_utf8_gather_lengths_kernel, dim3(grid_for(m, 256, 4096)), dim3(256), 0, stream, offsets,
                     validity, idx, m, out_lengths, validity ? out_validity : nullptr);
  return hipGetLastError();
}

size_t scan_scratch_elems(int64_t n) {
  // block sums of every level, laid end to end
  size_t total = 0;
  int64_t m = n;
  while (m > SCAN_TILE) {
    m = (m + SCAN_TILE - 1) / SCAN_TILE;
    total += (size_t)m;
  }
  return total + 1;
}

hipError_t launch_exclusive_scan_i32(int32_t *data, int64_t n, int32_t *block_sums, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (blocks == 1) {
    hipLaunchKernelGGL(qe_scan_tiles_kernel, dim3(1), dim3(SCAN_BLOCK), 0, stream, data, n, (int32_t *)nullptr);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(qe_scan_tiles_kernel, dim3((unsigned)blocks), dim3(SCAN_BLOCK), 0, stream, data, n, block_sums);
  hipError_t e = launch_exclusive_scan_i32(block_sums, blocks, block_sums + blocks, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(qe_scan_add_kernel, dim3((unsigned)blocks), dim3(SCAN_BLOCK), 0, stream, data, n, block_sums);
  return hipGetLastError();
}

hipError_t launch_utf8_gather_bytes(const int32_t *src_offsets, const uint8_t *src_bytes, const int64_t *idx, int64_t m,
                                    const int32_t *dst_offsets, uint8_t *dst_bytes, hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_utf8_gather_bytes_kernel, dim3(grid_for(m, 256, 4096)), dim3(256), 0, stream, src_offsets,
                     src_bytes, idx, m, dst_offsets, dst_bytes);
  return hipGetLastError();
}

hipError_t launch_utf8_broadcast(const uint8_t *lit, int32_t len, int64_t n, int32_t *out_offsets, uint8_t *out_bytes,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(qe_utf8_broadcast_kernel, dim3(grid_for(n + 1, 256, 4096)), dim3(256), 0, stream, lit, len, n,
                     out_offsets, out_bytes);
  return hipGetLastError();
}

hipError_t launch_count_zero_bits(const uint32_t *bitmap, int64_t n, unsigned long long *out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_count_zero_bits_kernel, dim3(grid_for((n + 31) / 32, 256, 1024)), dim3(256), 0, stream, bitmap,
                     n, out);
  return hipGetLastError();
}

hipError_t launch_generate(int kind, uint64_t seed, int64_t row0, int64_t n, double lo, double hi, void *out,
                           hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_generate_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream, kind, seed, row0, n, lo,
                     hi, out);
  return hipGetLastError();
}

}  // namespace qe
