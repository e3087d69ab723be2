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
#pragma once
// Device code only, written to compile both with hipcc and with the run-time compiler (which
// provides the HIP device declarations itself).
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include "qe_vm.h"

namespace qe {

// ------------------------------------------------------------------ helpers
// Column value buffers are 16-byte aligned (the host re-aligns imported buffers that are not),
// so a lane's row pair is one global_load_dwordx4 / global_store_dwordx4.  A pair that starts on
// a valid row never leaves the 16-byte granule of that row, so it cannot cross into an unmapped
// page even when the second row is past the end of the column.
struct alignas(16) U64x2 { uint64_t x, y; };
struct alignas(8) U32x2 { uint32_t x, y; };

// Column values are read exactly once per kernel: nontemporal loads keep the stream from
// displacing useful lines (group tables, status words) in L2 / Infinity Cache, and measure
// ~10 % faster than plain loads on a pure read stream of this shape on MI355X.
typedef uint64_t u64v2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ U64x2 stream_load_pair64(const uint64_t *p) {
  u64v2 t = __builtin_nontemporal_load((const u64v2 *)p);
  U64x2 r;
  r.x = t.x;
  r.y = t.y;
  return r;
}
__device__ __forceinline__ U32x2 stream_load_pair32(const uint32_t *p) {
  u32v2 t = __builtin_nontemporal_load((const u32v2 *)p);
  U32x2 r;
  r.x = t.x;
  r.y = t.y;
  return r;
}

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

// Rows of the lane (bit j of the result) whose string in `col` equals a literal of `len` <= 16 bytes held in two
// registers (`lo`: bytes 0..7, `hi`: bytes 8..15).  The length decides most rows; the others read their bytes once,
// nothing else (a literal in memory costs a second load per byte compared).
__device__ __forceinline__ uint32_t utf8_equals_lit16(const int32_t *offsets, const uint8_t *bytes, uint64_t lo, uint64_t hi, uint32_t len,
                                                    int64_t row0, uint32_t lane, uint32_t active) {
  // rows whose length is the literal's, and where they start
  uint32_t cand = 0;
  int32_t at[4] = {0, 0, 0, 0};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if ((active >> (2 * h)) & 3u) {
      const int64_t r = row0 + h * 128 + 2 * (int64_t)lane;
      // offsets r, r + 1, r + 2: the two rows of the pair (the last one is only read when the second row exists)
      const int32_t o0 = offsets[r], o1 = offsets[r + 1];
      const int32_t o2 = ((active >> (2 * h + 1)) & 1u) ? offsets[r + 2] : o1;
      if (((active >> (2 * h)) & 1u) && (uint32_t)(o1 - o0) == len) {
        cand |= 1u << (2 * h);
        at[2 * h] = o0;
      }
      if (((active >> (2 * h + 1)) & 1u) && (uint32_t)(o2 - o1) == len) {
        cand |= 2u << (2 * h);
        at[2 * h + 1] = o1;
      }
    }
  }
  if (len == 0 || __ballot(cand != 0) == 0) return cand;  // wave-uniform
  // The candidates' bytes, all fetches of the lane's four rows in flight together.  A row that is no candidate reads
  // the column's first bytes instead (at = 0): some row of the wave holds `len` bytes, so those exist.  From four
  // bytes on a string is read as unaligned dwords at 0, 4, ... with the last one ending at its last byte - two
  // fetches for "Female", never a byte outside the row.
  uint32_t eq = cand;
  if (len >= 4) {
    uint32_t w[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((uint32_t)(4 * i) < len) {  // uniform
        const uint32_t p = min((uint32_t)(4 * i), len - 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_memcpy(&w[j][i], bytes + at[j] + (int32_t)p, 4);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((uint32_t)(4 * i) < len) {
        const uint32_t p = min((uint32_t)(4 * i), len - 4u);  // bytes p .. p + 3 of the literal
        uint64_t v = p < 8 ? lo >> (8 * p) : hi >> (8 * (p - 8));
        if (p > 4 && p < 8) v |= hi << (64 - 8 * p);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (w[j][i] != (uint32_t)v) eq &= ~(1u << j);
      }
    }
  } else {
    uint32_t b[4][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if ((uint32_t)i < len) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j][i] = bytes[at[j] + i];
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if ((uint32_t)i < len) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (b[j][i] != (uint32_t)((lo >> (8 * i)) & 0xffu)) eq &= ~(1u << j);
      }
    }
  }
  return eq;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
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
      U64x2 t = stream_load_pair64(p + r0);
      w0 = t.x;
      w1 = t.y;
    }
    if (a1) {
      U64x2 t = stream_load_pair64(p + r1);
      w2 = t.x;
      w3 = t.y;
    }
  } else if (kind == CK_32) {
    const uint32_t *p = (const uint32_t *)col.values;  // raw pairs; sign-extended when pushed
    if (a0) {
      U32x2 t = stream_load_pair32(p + r0);
      w0 = (uint64_t)t.x | ((uint64_t)t.y << 32);
    }
    if (a1) {
      U32x2 t = stream_load_pair32(p + r1);
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

// ---- order-independent float64 sums (OP_FIX_*, FIN_FIX_*)
// Limb `k` of trunc(|x| * 2^(96 - S)), with the sign of x.  |x| < 2^S is the caller's business (S comes from a pass
// over the same rows); a value that breaks it loses its top bits instead of spilling into another limb's range.
__device__ __forceinline__ uint64_t fix_limb(uint64_t bits, int S, int k) {
  const uint32_t e = (uint32_t)(bits >> 52) & 0x7ffu;
  if (e == 0x7ffu) return 0;
  uint64_t m = bits & 0xfffffffffffffull;
  if (e != 0) m |= 1ull << 52;
  const int ee = e != 0 ? (int)e : 1;  // |x| = m * 2^(ee - 1075)
  const int s = ee - (1075 - FIX_FRACTION_BITS) - S - 32 * k;
  uint64_t v = s >= 0 ? (s < 64 ? m << s : 0) : (s > -64 ? m >> (-s) : 0);
  v &= 0xffffffffull;
  return (bits >> 63) ? 0 - v : v;
}
__device__ __forceinline__ uint64_t fix_exponent(uint64_t bits) {
  const uint32_t e = (uint32_t)(bits >> 52) & 0x7ffu;
  return e == 0x7ffu ? 0 : e;
}
// The infinity / NaN marks of x.  A finite x of a row that takes part (`live`) with |x| >= 2^S does not fit the limbs at
// scale S: `overflow` (when given) is set, and the host, which chose S without having seen every value, starts over.
__device__ __forceinline__ uint64_t fix_special(uint64_t bits, int S, unsigned long long *overflow, bool live) {
  const uint32_t e = (uint32_t)(bits >> 52) & 0x7ffu;
  if (e != 0x7ffu) {
    if (overflow != nullptr && live && (int)e - 1023 >= S) __hip_atomic_store(overflow, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  if (bits & 0xfffffffffffffull) return 1ull | (1ull << 32);
  return (bits >> 63) ? 1ull << 32 : 1ull;
}
// The float64 nearest to (l0 + l1 * 2^32 + l2 * 2^64) * 2^(S - 96); `special` as OP_FIX_SPECIAL counts it.
__device__ __forceinline__ double fix_combine(uint64_t l0, uint64_t l1, uint64_t l2, uint64_t special, int S) {
  const uint32_t pos = (uint32_t)special, neg = (uint32_t)(special >> 32);
  if (pos && neg) return u2d(0x7ff8000000000000ull);
  if (pos) return u2d(0x7ff0000000000000ull);
  if (neg) return u2d(0xfff0000000000000ull);
  __int128 t = (__int128)(int64_t)l0 + (((__int128)(int64_t)l1) << 32) + (((__int128)(int64_t)l2) << 64);
  const bool negative = t < 0;
  unsigned __int128 u = negative ? (unsigned __int128)(-t) : (unsigned __int128)t;
  const uint64_t hi = (uint64_t)(u >> 64), lo = (uint64_t)u;
  if (hi == 0 && lo == 0) return 0.0;
  int shift = 0;
  uint64_t top = lo;
  if (hi != 0) {
    shift = 64 - __builtin_clzll(hi);  // bits above the low word
    top = (uint64_t)(u >> shift);
    if ((u & ((((unsigned __int128)1) << shift) - 1)) != 0) top |= 1ull;  // sticky: the one rounding happens in the conversion below
  }
  const double d = ldexp((double)top, shift + S - FIX_FRACTION_BITS);
  return negative ? -d : d;
}
// The limbs of a sum moved to a scale `delta` bits up (the sum's low bits beyond the new unit are cut off).
__device__ __forceinline__ void fix_rescale(uint64_t &l0, uint64_t &l1, uint64_t &l2, int delta) {
  __int128 t = (__int128)(int64_t)l0 + (((__int128)(int64_t)l1) << 32) + (((__int128)(int64_t)l2) << 64);
  const bool negative = t < 0;
  unsigned __int128 u = negative ? (unsigned __int128)(-t) : (unsigned __int128)t;
  u = delta >= 128 ? 0 : u >> delta;
  const uint64_t a = (uint64_t)u & 0xffffffffull, b = (uint64_t)(u >> 32) & 0xffffffffull, c = (uint64_t)(u >> 64);
  l0 = negative ? 0 - a : a;
  l1 = negative ? 0 - b : b;
  l2 = negative ? 0 - c : c;
}

__host__ __device__ constexpr bool op_is_fast_push(uint32_t op) { return vm_op_is_cached_push(op) || op == OP_PUSH_LIT; }
__host__ __device__ constexpr bool op_is_fast_binary(uint32_t op) {
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
  const bool is_push = (op <= OP_PUSH_ROWID) || (op >= OP_UTF8_EQ_LIT && op <= OP_HASH_UTF8) || op == OP_UTF8_EQ_LIT16 || op == OP_UTF8_NE_LIT16;
  const bool is_unary = op == OP_I64_TO_F64 || op == OP_MIX64 || vm_op_is_fix(op);
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
    case OP_UTF8_EQ_LIT16:
    case OP_UTF8_NE_LIT16: {
      const VmCol &ca = P.cols[arg & 0xffu];
      const uint32_t li = (arg >> 8) & 0xffu;
      const uint32_t eq = utf8_equals_lit16(ca.offsets, (const uint8_t *)ca.values, P.lits[li], P.lits[li + 1], (uint32_t)P.lits[li + 2], c.row0, c.lane, c.active);
      const uint32_t bits = op == OP_UTF8_EQ_LIT16 ? eq : ~eq;
#pragma unroll
      for (int j = 0; j < VM_R; ++j) a.v[j] = (bits >> j) & 1u;
      a.valid = ca.validity ? load_bits4(ca.validity, c) : 0xFu;
      break;
    }
    case OP_HASH_UTF8: {
      // FNV-1a over the bytes, finished with a 64-bit mixer; rolled over the four rows
      const VmCol &ca = P.cols[arg];
      uint32_t act = c.active;
#pragma unroll 1
      for (int it = 0; it < VM_R; ++it) {
        uint64_t h = 0xcbf29ce484222325ull;
        if (act & 1u) {
          int64_t r = c.row0 + (it >> 1) * 128 + 2 * (int64_t)c.lane + (it & 1);
          const uint8_t *p = (const uint8_t *)ca.values;
          for (int32_t i = ca.offsets[r], e = ca.offsets[r + 1]; i < e; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
        }
        a.v[0] = mix64(h);
        QE_ROT4(a.v);
        act = rot4(act);
      }
      a.valid = ca.validity ? load_bits4(ca.validity, c) : 0xFu;
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
    case OP_MIX64: {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) a.v[j] = mix64(a.v[j]);
      break;
    }
    case OP_FIX_EXP: {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) a.v[j] = fix_exponent(a.v[j]);
      break;
    }
    case OP_FIX_LIMB0:
    case OP_FIX_LIMB1:
    case OP_FIX_LIMB2: {
      const int S = (int)(int64_t)P.lits[arg];
#pragma unroll
      for (int j = 0; j < VM_R; ++j) a.v[j] = fix_limb(a.v[j], S, (int)(op - OP_FIX_LIMB0));
      break;
    }
    case OP_FIX_SPECIAL: {
      const int S = (int)(int64_t)P.lits[arg & 0xffu];
      unsigned long long *overflow = (unsigned long long *)(uintptr_t)P.lits[(arg >> 8) & 0xffu];
#pragma unroll
      for (int j = 0; j < VM_R; ++j) a.v[j] = fix_special(a.v[j], S, overflow, ((c.active & a.valid) >> j) & 1u);
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
        u32v2 t;
        t.x = (uint32_t)s.v[2 * h];
        t.y = (uint32_t)s.v[2 * h + 1];
        __builtin_nontemporal_store(t, (u32v2 *)(p + r));
      }
    }
  } else {
    uint64_t *p = (uint64_t *)o.values;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int64_t r = c.row0 + h * 128 + 2 * (int64_t)c.lane;
      if (r < c.n) {
        u64v2 t;  // (written once, read by another kernel if at all: a streaming store)
        t.x = s.v[2 * h];
        t.y = s.v[2 * h + 1];
        __builtin_nontemporal_store(t, (u64v2 *)(p + r));
      }
    }
  }
  if (o.validity) store_bitmap256(o.validity, c, s.valid & inr);
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
// Every tile in flight has published only its own count, so the walk back to an
// inclusive prefix is about as long as the number of tiles in flight; one status
// word per lane per round (reading four per lane measured slower: the extra polling
// traffic costs more than the shorter walk saves).
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

// Validity bits of a wave-step's selected rows, which go to the consecutive output rows base .. base + cnt - 1 in rank
// order.  The lanes drop their rows' bits as bytes at their ranks in `flags` (256 bytes of LDS owned by the wave), four
// ballots over the bytes give the step's bits in output order, and nine lanes put them into the bitmap's words (zeroed
// beforehand; neighbouring waves share the words at either end) - nine word-wide atomics a wave-step at most instead of
// one per selected row.  Called by the whole wave.
__device__ __forceinline__ void store_validity_ranked(uint32_t *validity, unsigned long long base, uint32_t sel, const uint32_t (&rank)[VM_R],
                                                      uint32_t valid, uint8_t *flags, uint32_t cnt, uint32_t lane) {
#pragma unroll
  for (int j = 0; j < VM_R; ++j)
    if ((sel >> j) & 1u) flags[rank[j]] = (uint8_t)((valid >> j) & 1u);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave's own LDS writes, before its reads below
  __builtin_amdgcn_wave_barrier();
  uint64_t m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t p = (uint32_t)k * 64u + lane;
    m[k] = __ballot(p < cnt && flags[p] != 0);
  }
  if ((m[0] | m[1] | m[2] | m[3]) == 0) return;
  uint32_t cur = 0, prev = 0;  // words `lane` and `lane - 1` of the 256 bits
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t w = (uint32_t)(m[i >> 1] >> (32 * (i & 1)));
    cur = lane == (uint32_t)i ? w : cur;
    prev = lane == (uint32_t)i + 1u ? w : prev;
  }
  const uint32_t sh = (uint32_t)(base & 31u);
  const uint32_t word = sh ? (cur << sh) | (prev >> (32u - sh)) : cur;
  if (lane < 9 && word != 0) atomicOr(validity + (base >> 5) + lane, word);
}

// store_compact with the bit-packed parts - validity, boolean values - assembled by the wave (store_validity_ranked)
// instead of one atomic per row.  Called by the whole wave; `cnt` = the wave-step's selected rows.
__device__ __forceinline__ void store_compact_ranked(const VmOut &o, unsigned long long base, uint32_t sel, const uint32_t (&rank)[VM_R],
                                                     const Slot &s, uint8_t *flags, uint32_t cnt, uint32_t lane) {
  if (o.type == VT_BOOL) {
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) bits |= (uint32_t)(s.v[j] & 1u) << j;
    store_validity_ranked((uint32_t *)o.values, base, sel, rank, bits, flags, cnt, lane);
  } else {
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      if ((sel >> j) & 1u) {
        if (o.type == VT_I32) ((uint32_t *)o.values)[base + rank[j]] = (uint32_t)s.v[j];
        else ((uint64_t *)o.values)[base + rank[j]] = s.v[j];
      }
    }
  }
  if (o.validity) store_validity_ranked(o.validity, base, sel, rank, s.valid, flags, cnt, lane);
}

// --------------------------------------------------------- aggregate kernel
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
constexpr int32_t SLOT_LDS_MISS = -1;  // transient: no room in the LDS table, the row needs another home
constexpr int LDS_MAX_PROBE = 64;

struct LdsTable {
  unsigned long long *keys;  // [cap + 1]
  unsigned long long *acc;   // [n_accs][cap + 1]
  uint32_t *rowcnt;          // [cap + 1]
  uint32_t *nn;              // [n_accs][cap + 1]
  uint32_t *used;            // claimed slots
  uint16_t *door;            // high-cardinality mode: tags of keys seen once (see lds_find_or_claim4); nullptr = none
  uint32_t cap;
  uint32_t stride;           // cap + 1
  uint32_t limit;
  uint32_t home_mask;        // hash -> first slot of a key's probe sequence: cap - 1, or (cap - 1) & ~1 in a table that is
                             // also probed a pair of slots at a time (both walks then meet the slots in the same order)
};

// Cheap 32-bit hash: its low bits name a key's slot in an LDS table, its top bits the key's partition in the many-groups
// mode and its home slot in the global table.
__device__ __forceinline__ uint32_t hash32(uint64_t key) {
  uint32_t h = (uint32_t)key * 0x9E3779B1u ^ (uint32_t)(key >> 32) * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 13;
  return h;
}

// A logged row with one value column, and its streaming load as one 16-byte access (two 8-byte loads of the halves
// each touch every line of a wave's stretch).
struct alignas(16) KeyVal { unsigned long long key, val; };
typedef unsigned long long qe_ull2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ KeyVal load_pair_streaming(const KeyVal *p) {
  const qe_ull2 v = __builtin_nontemporal_load((const qe_ull2 *)p);
  KeyVal r;
  r.key = v.x;
  r.val = v.y;
  return r;
}

// Partition of a key: the top bits of the 32-bit hash whose low bits pick the slot in the LDS tables (three 32-bit
// multiplies; a 64-bit mixer costs several times that, per row and pass).
__device__ __forceinline__ uint32_t partition_of(unsigned long long key, int bits) { return hash32(key) >> (32 - bits); }

// Workgroup barrier for code whose waves talk through LDS only.  __syncthreads() also waits for every global store the
// wave has in flight (it is a fence for all memory): after a burst of stores that is the whole trip to HBM, per round.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt and expcnt left alone
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Home slot of a key in the global table = the top bits of the 32-bit hash whose top bits also name the key's partition
// in the many-groups mode (below them, bits of `h` in a table of more than 2^32 slots): the keys of one partition live
// in one stretch of the table, which a partition's flush then walks instead of touching lines all over it.
__device__ __forceinline__ uint64_t global_home(const AggTable &T, uint64_t key, uint64_t h) {
  const int lg = 63 - __builtin_clzll(T.cap);
  const uint64_t s = lg <= 32 ? (uint64_t)(hash32(key) >> 1 >> (31 - lg)) : (((uint64_t)hash32(key) << (lg - 32)) | ((h >> 20) & ((1ull << (lg - 32)) - 1ull)));
  return s & (T.cap - 1);
}

// COUNT_HERE: a claimed slot is counted into the table's group counter on the spot.  Callers that claim many slots at
// once (a workgroup flushing its LDS table) pass false, take `claimed` and add the wave's total with one atomic: a
// million claims are otherwise a million atomics on one word.
template <bool COUNT_HERE = true>
__device__ __forceinline__ int64_t global_find_or_claim(const AggTable &T, uint64_t key, uint64_t h, bool *claimed = nullptr) {
  if (claimed) *claimed = false;
  if (key == AGG_EMPTY_KEY) {
    // the one key that collides with the empty marker lives in the reserved slot
    unsigned long long old = atomicCAS(&T.keys[T.cap], AGG_EMPTY_KEY, 0ull);
    if (old == AGG_EMPTY_KEY) {
      if (COUNT_HERE) atomicAdd(&T.ctrl[0], 1ull);
      else if (claimed) *claimed = true;
    }
    return (int64_t)T.cap;
  }
  const uint64_t mask = T.cap - 1;
  uint64_t s = global_home(T, key, h);
#pragma unroll 1
  for (uint64_t p = 0; p <= mask; ++p) {
    unsigned long long cur = __hip_atomic_load(&T.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return (int64_t)s;
    if (cur == AGG_EMPTY_KEY) {
      unsigned long long old = atomicCAS(&T.keys[s], AGG_EMPTY_KEY, (unsigned long long)key);
      if (old == AGG_EMPTY_KEY) {
        if (COUNT_HERE) atomicAdd(&T.ctrl[0], 1ull);
        else if (claimed) *claimed = true;
        return (int64_t)s;
      }
      if (old == key) return (int64_t)s;
    }
    // A long probe sequence means a crowded table.  Once 7/8 of the slots are taken the table counts as full:
    // the host starts over with a larger one, and walking a full table slot by slot for every further key
    // (minutes, for a million keys against 65 536 slots) serves nobody; a table already given up is left at once.
    if ((p & 31) == 8) {
      if (__hip_atomic_load(&T.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return -1;
      if (__hip_atomic_load(&T.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= T.cap - T.cap / 8) break;
    }
    s = (s + 1) & mask;
  }
  atomicExch(&T.ctrl[1], 1ull);
  return -1;
}

// A group table that has been given up (full: the host starts over with a larger one) makes the rest of the
// launch pointless; the waves look every 64th wave-step and leave.  A look is a load whose result is only
// read at the start of the next step - by then the step's own column loads have been waited for, and this
// one with them - so that the wave never waits for it.
struct GiveUpWatch {
  bool on = true;  // off in high-cardinality mode: its first phase logs rows and leaves the table alone
  uint32_t steps = 0;
  unsigned long long seen = 0;
};
__device__ __forceinline__ bool table_given_up(const AggTable &T, GiveUpWatch &w) {
  if (!w.on) return false;
  const bool lost = __builtin_amdgcn_readfirstlane((int)(w.seen != 0)) != 0;
  if ((w.steps++ & 63u) == 63u) w.seen = __hip_atomic_load(&T.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return lost;
}

// LDS slot of `key` (claiming one if new), or -1 when the key does not fit the LDS table.
// The general case: linear probing with a claim by compare-swap.  Kept as a rolled loop: it runs
// for the first rows that meet a key and for home-slot collisions, the hot case is
// lds_probe_home below.
__device__ __forceinline__ int32_t lds_find_or_claim(const LdsTable &L, uint64_t key) {
  if (key == AGG_EMPTY_KEY) {
    (void)atomicCAS(&L.keys[L.cap], AGG_EMPTY_KEY, 0ull);
    return (int32_t)L.cap;
  }
  uint32_t s = hash32(key) & L.home_mask;
#pragma unroll 1
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

// The lookup of the lane's four rows, all at once: every probe round has the four LDS reads in flight together (a
// serial chain per row costs four LDS round trips per probe), and a key that meets an empty slot claims it there and
// then (compare-swap) unless the table has stopped taking keys.  Rows whose key is not found within `max_probes`
// slots, or not in a table that takes no more keys, come back as SLOT_LDS_MISS: their rows need another home (the
// global table, or the row log in high-cardinality mode - where giving up early is always allowed, because a logged row
// is aggregated all the same).
constexpr int SPILL_MODE_PROBES = 4;
constexpr uint32_t DOOR_ENTRIES = 4096;  // 16-bit tags
__device__ __forceinline__ void lds_find_or_claim4(const LdsTable &L, const uint64_t (&key)[VM_R], uint32_t rows, int max_probes,
                                                   int32_t (&slot)[VM_R]) {
  const uint32_t mask = L.cap - 1;
  uint32_t h[VM_R];
  uint32_t open = rows;  // rows still looking
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    h[j] = hash32(key[j]) & L.home_mask;
    slot[j] = ((rows >> j) & 1u) ? SLOT_LDS_MISS : SLOT_NONE;
    if (((rows >> j) & 1u) && key[j] == AGG_EMPTY_KEY) {
      // the one key equal to the empty marker lives in the reserved slot
      (void)atomicCAS(&L.keys[L.cap], AGG_EMPTY_KEY, 0ull);
      slot[j] = (int32_t)L.cap;
      open &= ~(1u << j);
    }
  }
#pragma unroll 1
  for (int p = 0; p < max_probes; ++p) {
    if (__ballot(open != 0) == 0) break;
    unsigned long long cur[VM_R];
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      cur[j] = AGG_EMPTY_KEY;
      if ((open >> j) & 1u) cur[j] = *(volatile unsigned long long *)&L.keys[h[j]];
    }
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      // (every array element is written once, by a select at this level: stores to slot[] / h[] from inside the branches
      // get merged into one store with a computed index, and the arrays end up in scratch memory)
      const bool is_open = ((open >> j) & 1u) != 0;
      const bool hit = is_open && cur[j] == key[j];
      bool got = hit, closed = hit;
      if (is_open && cur[j] == AGG_EMPTY_KEY) {
        // an empty slot ends the probe sequence: the key is not in the table.  Claim the slot if the table still takes keys.
        closed = true;
        bool admit = *(volatile uint32_t *)L.used < L.limit;
        if (admit && L.door) {
          // Admission by second sighting.  When rows that find no slot are logged anyway, a slot is worth having only
          // for a key that comes often, and with many distinct keys most first sightings are of keys that never
          // come again: a key is let in when its tag is still in a small direct-mapped table of recent sightings.
          const uint32_t hf = hash32(key[j]);  // (again: rare, and registers held across the loop cost more)
          const uint32_t d = (hf >> 12) & (DOOR_ENTRIES - 1);
          const uint16_t tag = (uint16_t)((hf >> 16) | 1u);
          if (*(volatile uint16_t *)&L.door[d] != tag) {
            L.door[d] = tag;
            admit = false;
          }
        }
        if (admit) {
          const unsigned long long old = atomicCAS(&L.keys[h[j]], AGG_EMPTY_KEY, (unsigned long long)key[j]);
          if (old == AGG_EMPTY_KEY) atomicAdd(L.used, 1u);
          got = old == AGG_EMPTY_KEY || old == key[j];
          closed = got;  // else another key took the slot in the meantime: the sequence goes on
        }
      }
      slot[j] = got ? (int32_t)h[j] : slot[j];
      h[j] = (is_open && !closed) ? ((h[j] + 1) & mask) : h[j];
      open = closed ? (open & ~(1u << j)) : open;
    }
  }
}

// The same lookup, a pair of slots at a time: a key's probe sequence is the aligned pairs (buckets) from the one its
// hash names onwards, each read with one 16-byte LDS load, and a key that meets an empty slot before itself claims the
// bucket's first empty slot.  `rounds` buckets are looked at; rows whose key was neither found nor placed by then - or
// that lost a claim to another key, or were not admitted - are simply not in the returned mask (the caller has another
// home for them).  slot8[j] = byte offset of row j's slot in an array of 8-byte entries (slot x 8).
// ADMISSION: claims obey the table's limit and, when it has one, the second-sighting filter (lds_find_or_claim4).
template <bool ADMISSION>
__device__ __forceinline__ uint32_t lds_pair_find_or_claim4(const LdsTable &L, const uint64_t (&key)[VM_R], uint32_t rows, int rounds,
                                                            uint32_t (&slot8)[VM_R]) {
  const uint32_t mask = L.cap - 1;
  uint32_t b[VM_R];
  uint32_t open = rows, hit = 0;
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    b[j] = hash32(key[j]) & mask & ~1u;
    slot8[j] = 0;
  }
  {
    // the one key equal to the empty marker lives in the reserved slot
    uint32_t marker = 0;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) marker |= (((rows >> j) & 1u) && key[j] == AGG_EMPTY_KEY) ? 1u << j : 0u;
    if (__ballot(marker != 0) != 0) {
      if (marker != 0) (void)atomicCAS(&L.keys[L.cap], AGG_EMPTY_KEY, 0ull);
#pragma unroll
      for (int j = 0; j < VM_R; ++j) slot8[j] = ((marker >> j) & 1u) ? L.cap * 8u : slot8[j];
      hit |= marker;
      open &= ~marker;
    }
  }
#pragma unroll 1
  for (int r = 0; r < rounds; ++r) {
    if (__ballot(open != 0) == 0) break;
    uint64_t k0[VM_R], k1[VM_R];
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      // (b is even and the key array 16-byte aligned: one 16-byte read.  Every row reads - a row that is not open the
      // table's first bucket, one address for all such lanes, which costs the LDS next to nothing: behind a test of its
      // lane mask each read would be waited for before the next one is issued, four LDS round trips a round instead of one)
      const U64x2 two = *(const U64x2 *)__builtin_assume_aligned(&L.keys[((open >> j) & 1u) ? b[j] : 0u], 16);
      k0[j] = two.x;
      k1[j] = two.y;
    }
    uint32_t want = 0;  // rows whose probe ended at an empty slot without finding the key
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      const bool o = ((open >> j) & 1u) != 0;
      const bool h0 = o && k0[j] == key[j], h1 = o && k1[j] == key[j];
      const bool e0 = k0[j] == AGG_EMPTY_KEY, e1 = k1[j] == AGG_EMPTY_KEY;
      const bool found = h0 || h1;
      const bool ended = o && !found && (e0 || e1);
      slot8[j] = found ? (b[j] + (h0 ? 0u : 1u)) * 8u : slot8[j];
      hit |= found ? 1u << j : 0u;
      want |= ended ? 1u << j : 0u;
      b[j] = ended ? b[j] + (e0 ? 0u : 1u) : ((b[j] + 2u) & mask);  // the empty slot itself, or the next bucket
      open &= (found || ended) ? ~(1u << j) : ~0u;
    }
    // (claims happen while the table fills and when a key is let in later.  A table that takes no more keys - the steady
    // state of the many-groups first pass - is found out with one read for the whole step: the rows' own checks below,
    // each behind its lane test, are four LDS round trips one after the other)
    if (__ballot(want != 0) != 0 && *(volatile uint32_t *)L.used < L.limit) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) {
        bool got = false, again = false;
        if ((want >> j) & 1u) {
          bool admit = *(volatile uint32_t *)L.used < L.limit;
          if (ADMISSION && admit && L.door) {
            const uint32_t hf = hash32(key[j]);
            const uint32_t d = (hf >> 12) & (DOOR_ENTRIES - 1);
            const uint16_t tag = (uint16_t)((hf >> 16) | 1u);
            if (*(volatile uint16_t *)&L.door[d] != tag) {
              L.door[d] = tag;
              admit = false;
            }
          }
          if (admit) {
            const unsigned long long old = atomicCAS(&L.keys[b[j]], AGG_EMPTY_KEY, (unsigned long long)key[j]);
            if (old == AGG_EMPTY_KEY) atomicAdd(L.used, 1u);
            got = old == AGG_EMPTY_KEY || old == key[j];
            again = !got;  // lost the slot to another key: look at this bucket again (its other slot may be free)
          }
        }
        slot8[j] = got ? b[j] * 8u : slot8[j];
        hit |= got ? 1u << j : 0u;
        if (!ADMISSION) {
          // every key must find a home here: the row goes on probing from its bucket
          open |= again ? 1u << j : 0u;
          b[j] = again ? (b[j] & ~1u) : b[j];
        }
      }
    }
  }
  return hit;
}

// The lookup of the partition-wise aggregate, where every key must find a slot: the home buckets of a lane's four rows
// are read together (one LDS round trip), and whatever that look has not settled - a key further down its probe
// sequence, a new key that claims an empty slot - goes on ONE row per lane at a time.  An LDS instruction costs the
// unit its dozen cycles whatever the number of lanes that want something, and after the first look few lanes do: four
// instructions a round for the stragglers (lds_pair_find_or_claim4) were half the partition pass's LDS time.
// Returns the rows that have a slot (slot8[j] = slot x 8); a row is left out when the table has reached its limit
// or `rounds` looks per row were not enough (the caller has another home for it).
__device__ __forceinline__ uint32_t lds_pair_find_or_claim_rows(const LdsTable &L, const uint64_t (&key)[VM_R], uint32_t rows, int rounds,
                                                                uint32_t (&slot8)[VM_R]) {
  const uint32_t mask = L.cap - 1;
  uint32_t b[VM_R];
  uint32_t todo = rows, hit = 0;
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    b[j] = hash32(key[j]) & mask & ~1u;
    slot8[j] = 0;
  }
  {
    // the one key equal to the empty marker lives in the reserved slot
    uint32_t marker = 0;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) marker |= (((rows >> j) & 1u) && key[j] == AGG_EMPTY_KEY) ? 1u << j : 0u;
    if (__ballot(marker != 0) != 0) {
      if (marker != 0) (void)atomicCAS(&L.keys[L.cap], AGG_EMPTY_KEY, 0ull);
#pragma unroll
      for (int j = 0; j < VM_R; ++j) slot8[j] = ((marker >> j) & 1u) ? L.cap * 8u : slot8[j];
      hit |= marker;
      todo &= ~marker;
    }
  }
  {
    // the first look (a row that is not looked for reads the table's first bucket: one address for all such lanes)
    uint64_t k0[VM_R], k1[VM_R];
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      const U64x2 two = *(const U64x2 *)__builtin_assume_aligned(&L.keys[((todo >> j) & 1u) ? b[j] : 0u], 16);
      k0[j] = two.x;
      k1[j] = two.y;
    }
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      const bool o = ((todo >> j) & 1u) != 0;
      const bool h0 = o && k0[j] == key[j], h1 = o && k1[j] == key[j];
      const bool found = h0 || h1;
      const bool has_empty = k0[j] == AGG_EMPTY_KEY || k1[j] == AGG_EMPTY_KEY;
      slot8[j] = found ? (b[j] + (h0 ? 0u : 1u)) * 8u : slot8[j];
      hit |= found ? 1u << j : 0u;
      todo &= found ? ~(1u << j) : ~0u;
      b[j] = (o && !found && !has_empty) ? ((b[j] + 2u) & mask) : b[j];  // (a bucket with an empty slot is looked at again below, to claim it)
    }
  }
  // the stragglers, one row per lane and look
#pragma unroll 1
  for (int r = 0; r < rounds * VM_R; ++r) {
    if (__ballot(todo != 0) == 0) break;
    const bool any = todo != 0;
    const uint32_t jj = any ? (uint32_t)__builtin_ctz(todo) : 0u;
    uint64_t kk = opaque64(key[3]);
    uint32_t bb = opaque(b[3]);
    if (jj == 2u) kk = opaque64(key[2]), bb = opaque(b[2]);
    if (jj == 1u) kk = opaque64(key[1]), bb = opaque(b[1]);
    if (jj == 0u) kk = opaque64(key[0]), bb = opaque(b[0]);
    const U64x2 two = *(const U64x2 *)__builtin_assume_aligned(&L.keys[any ? bb : 0u], 16);
    const bool h0 = any && two.x == kk, h1 = any && two.y == kk;
    const bool e0 = two.x == AGG_EMPTY_KEY, e1 = two.y == AGG_EMPTY_KEY;
    bool found = h0 || h1, gone = false, fresh = false;
    uint32_t at = bb + (h0 ? 0u : 1u);
    if (any && !found && (e0 || e1)) {
      // a new key: the bucket's first empty slot, if the table still takes keys
      at = bb + (e0 ? 0u : 1u);
      if (*(volatile uint32_t *)L.used < L.limit) {
        const unsigned long long old = atomicCAS(&L.keys[at], AGG_EMPTY_KEY, (unsigned long long)kk);
        fresh = old == AGG_EMPTY_KEY;
        found = fresh || old == kk;  // (lost to another key: the same bucket once more - its other slot may be free)
      } else {
        gone = true;
      }
    } else if (any && !found) {
      bb = (bb + 2u) & mask;
    }
    {
      // (the wave's claims counted with one atomic, not lane by lane on one LDS address)
      const uint64_t claims = __ballot(fresh);
      if (claims != 0 && lane_id() == (uint32_t)__builtin_ctzll(claims)) atomicAdd(L.used, (uint32_t)__popcll(claims));
    }
    const uint32_t bit = any ? 1u << jj : 0u;
    hit |= found ? bit : 0u;
    todo &= (found || gone) ? ~bit : ~0u;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      const bool me = any && jj == (uint32_t)j;
      slot8[j] = (me && found) ? at * 8u : slot8[j];
      b[j] = me ? bb : b[j];
    }
  }
  return hit;
}

// The two-bucket lookup of the many-groups first pass (lds_pair_find_or_claim4<true> with two rounds) with both buckets
// of every row read at once: at three quarters full a bucket is full without the key more often than not, the second
// look is needed by some lane of nearly every step, and one after the other the two looks are two LDS round trips.
__device__ __forceinline__ uint32_t lds_two_pairs_find_or_claim4(const LdsTable &L, const uint64_t (&key)[VM_R], uint32_t rows, uint32_t (&slot8)[VM_R]) {
  const uint32_t mask = L.cap - 1;
  uint32_t b[VM_R];
  uint32_t hit = 0, open = rows;
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    b[j] = hash32(key[j]) & mask & ~1u;
    slot8[j] = 0;
  }
  {
    // the one key equal to the empty marker lives in the reserved slot
    uint32_t marker = 0;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) marker |= (((rows >> j) & 1u) && key[j] == AGG_EMPTY_KEY) ? 1u << j : 0u;
    if (__ballot(marker != 0) != 0) {
      if (marker != 0) (void)atomicCAS(&L.keys[L.cap], AGG_EMPTY_KEY, 0ull);
#pragma unroll
      for (int j = 0; j < VM_R; ++j) slot8[j] = ((marker >> j) & 1u) ? L.cap * 8u : slot8[j];
      hit |= marker;
      open &= ~marker;
    }
  }
  uint64_t k0[VM_R], k1[VM_R], k2[VM_R], k3[VM_R];
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    k0[j] = k1[j] = k2[j] = k3[j] = 0;
    if ((open >> j) & 1u) {
      const U64x2 first = *(const U64x2 *)__builtin_assume_aligned(&L.keys[b[j]], 16);
      const U64x2 second = *(const U64x2 *)__builtin_assume_aligned(&L.keys[(b[j] + 2u) & mask], 16);
      k0[j] = first.x, k1[j] = first.y, k2[j] = second.x, k3[j] = second.y;
    }
  }
  uint32_t want = 0;  // rows whose probe ended at an empty slot without finding the key
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    const bool o = ((open >> j) & 1u) != 0;
    const bool h0 = k0[j] == key[j], h1 = k1[j] == key[j], h2 = k2[j] == key[j], h3 = k3[j] == key[j];
    const bool e0 = k0[j] == AGG_EMPTY_KEY, e1 = k1[j] == AGG_EMPTY_KEY, e2 = k2[j] == AGG_EMPTY_KEY, e3 = k3[j] == AGG_EMPTY_KEY;
    // the first bucket decides if it holds the key or an empty slot; only a full bucket without the key sends the row on
    const bool first_decides = h0 || h1 || e0 || e1;
    const bool found = o && (first_decides ? (h0 || h1) : (h2 || h3));
    const bool ended = o && !found && (first_decides || e2 || e3);
    const uint32_t at = first_decides ? b[j] + ((h0 || (!h1 && e0)) ? 0u : 1u) : ((b[j] + 2u) & mask) + ((h2 || (!h3 && e2)) ? 0u : 1u);
    slot8[j] = found ? at * 8u : slot8[j];
    hit |= found ? 1u << j : 0u;
    want |= ended ? 1u << j : 0u;
    b[j] = at;  // (for a row that wants a slot: the empty one)
  }
  if (__ballot(want != 0) != 0 && *(volatile uint32_t *)L.used < L.limit) {
#pragma unroll
    for (int j = 0; j < VM_R; ++j) {
      bool got = false, fresh = false;
      if ((want >> j) & 1u) {
        bool admit = *(volatile uint32_t *)L.used < L.limit;
        if (admit && L.door) {
          const uint32_t hf = hash32(key[j]);
          const uint32_t d = (hf >> 12) & (DOOR_ENTRIES - 1);
          const uint16_t tag = (uint16_t)((hf >> 16) | 1u);
          if (*(volatile uint16_t *)&L.door[d] != tag) {
            L.door[d] = tag;
            admit = false;
          }
        }
        if (admit) {
          const unsigned long long old = atomicCAS(&L.keys[b[j]], AGG_EMPTY_KEY, (unsigned long long)key[j]);
          fresh = old == AGG_EMPTY_KEY;
          got = fresh || old == key[j];
        }
      }
      // (the wave's claims counted with one atomic: lane by lane they are as many operations on one LDS address, one
      // after the other - while a table fills, most of what its workgroup does)
      const uint64_t claims = __ballot(fresh);
      if (claims != 0 && lane_id() == (uint32_t)__builtin_ctzll(claims)) atomicAdd(L.used, (uint32_t)__popcll(claims));
      slot8[j] = got ? b[j] * 8u : slot8[j];
      hit |= got ? 1u << j : 0u;
    }
  }
  return hit;
}

__device__ __forceinline__ void acc_apply(unsigned long long *p, uint32_t op, uint64_t v) {
  switch (op) {
    case ACC_SUM_F64: atomicAdd((double *)p, u2d(v)); break;
    // min / max: a plain read first.  The stored value only ever moves one way, so a value that
    // does not beat what a (possibly stale) read returns cannot beat the current one either, and
    // the atomic is skipped: after the first few rows of a group that is nearly every row.
    case ACC_MIN:
      if ((unsigned long long)v < *(volatile unsigned long long *)p) atomicMin(p, (unsigned long long)v);
      break;
    case ACC_MAX:
      if ((unsigned long long)v > *(volatile unsigned long long *)p) atomicMax(p, (unsigned long long)v);
      break;
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

// ------------------------------------------------------------ aggregate sink
// Shared by the interpreted and the statically specialised aggregate kernels.
// `with_nn` = false leaves out the per-accumulator counts of valid inputs (nobody tracks them in
// the partition-wise aggregate), which makes room for a third more slots.
__device__ __forceinline__ void agg_lds_setup(LdsTable &L, unsigned long long *lds_raw, uint32_t lds_cap, uint32_t lds_limit,
                                              int n_accs, bool with_nn = true) {
  L.cap = lds_cap;
  L.stride = lds_cap + 1;
  L.limit = lds_limit;
  L.home_mask = lds_cap - 1;
  L.keys = lds_raw;
  L.acc = L.keys + L.stride;
  L.rowcnt = (uint32_t *)(L.acc + (size_t)n_accs * L.stride);
  L.nn = with_nn ? L.rowcnt + L.stride : nullptr;
  L.used = L.rowcnt + L.stride + (with_nn ? (size_t)n_accs * L.stride : 0);
  L.door = nullptr;
}

struct AggRows {
  uint64_t key[VM_R];
  int32_t slot[VM_R];  // >= 0 LDS slot; <= -2: global slot g as -2 - g; SLOT_NONE: dropped
  uint32_t rem;        // rows folded one by one
  uint32_t lead_mb;    // rows folded through the wave reduction
  int32_t lead_slot;   // wave-uniform; SLOT_NONE = no dominant key in this wave-step
  uint32_t spill_mb;   // rows logged for the second phase instead (high-cardinality mode)
  uint64_t spill_pos;  // log position of the lane's first such row; the others follow
};

// A wave's cursor into its region of the spill log.
struct SpillState {
  unsigned long long *keys;  // nullptr: high-cardinality mode off
  unsigned long long *vals;
  uint64_t capacity;
  uint64_t base;     // first row of this wave's region
  uint32_t cursor;   // rows logged so far (wave-uniform)
  uint32_t pairs;    // AggSpill::pairs
  uint32_t null_cols, null_f64;  // AggSpill::null_cols / null_f64
};

// What the log holds for value `v` of value column `col` (AggSpill::null_cols).
__device__ __forceinline__ unsigned long long spill_logged_value(uint32_t null_cols, uint32_t null_f64, uint32_t col, unsigned long long v, bool valid) {
  if (!((null_cols >> col) & 1u)) return v;  // (uniform)
  if ((null_f64 >> col) & 1u) return valid ? (v == SPILL_NULL_F64 ? SPILL_PLAIN_NAN : v) : SPILL_NULL_F64;
  return valid ? v : SPILL_NULL_INT;
}
// Rows (bits of `rows`) whose logged value of column `col` is not the null mark.
__device__ __forceinline__ uint32_t spill_valid_rows(uint32_t null_cols, uint32_t null_f64, uint32_t col, const uint64_t (&v)[VM_R], uint32_t rows) {
  if (!((null_cols >> col) & 1u)) return rows;  // (uniform)
  const uint64_t mark = ((null_f64 >> col) & 1u) ? (uint64_t)SPILL_NULL_F64 : (uint64_t)SPILL_NULL_INT;
#pragma unroll
  for (int j = 0; j < VM_R; ++j)
    if (v[j] == mark) rows &= ~(1u << j);
  return rows;
}

__device__ __forceinline__ SpillState spill_begin(const AggSpill &sp, uint32_t wave_in_block) {
  SpillState s;
  s.keys = sp.keys;
  s.vals = sp.vals;
  s.capacity = sp.capacity;
  s.base = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block) * sp.rows_per_wave;
  s.cursor = 0;
  s.pairs = sp.pairs;
  s.null_cols = sp.null_cols;
  s.null_f64 = sp.null_f64;
  return s;
}
__device__ __forceinline__ void spill_end(const AggSpill &sp, const SpillState &s, uint32_t wave_in_block, uint32_t lane) {
  if (sp.keys && lane == 0) sp.counts[blockIdx.x * (blockDim.x >> 6) + wave_in_block] = s.cursor;
}

// Logs one evaluated accumulator input of the spilled rows to value column `col`.
__device__ __forceinline__ void spill_values(const SpillState &sp, uint32_t col, const Slot &in, const AggRows &G) {
  // (a log of pairs has one value column: the value sits behind its key)
  unsigned long long *dst = sp.pairs ? sp.keys + 2 * G.spill_pos + 1 : sp.vals + (uint64_t)col * sp.capacity + G.spill_pos;
  uint32_t k = 0;
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    if ((G.spill_mb >> j) & 1u) dst[k << sp.pairs] = spill_logged_value(sp.null_cols, sp.null_f64, col, in.v[j], ((in.valid >> j) & 1u) != 0);
    k += (G.spill_mb >> j) & 1u;
  }
}

// Finds / claims the group slot of every active row and counts the rows in.
// ROLLED: the rows are looked up one after the other by a rolled loop - far fewer registers than the four-at-once
// lookup, for kernels in which lookups are the exception (narrow rows, where the registers decide how many waves fit).
template <bool TRY_LEADER = true, bool ROLLED = false>
__device__ __forceinline__ void agg_lookup(const LdsTable &L, const AggTable &T, uint32_t active, uint32_t lane, AggRows &G,
                                           SpillState &sp) {
  uint32_t rem = active;
  G.spill_mb = 0;
  G.spill_pos = 0;
  // ---- a key that dominates the wave-step?  Try the first row of up to two lanes.
  uint64_t ball = (TRY_LEADER && !sp.keys) ? __ballot(rem != 0) : 0ull;
  uint64_t k0 = 0;
  uint32_t lead_cnt = 0, lead_mb = 0;
  int src = 0;
#pragma unroll 1
  for (int tr = 0; tr < LEADER_TRIES && ball != 0; ++tr) {
    src = __builtin_ctzll(ball);
    // `opaque64`: a plain select chain is folded into one load with a computed index, which
    // moves the key registers to scratch memory
    uint64_t kl = opaque64(G.key[3]);
    if (rem & 4u) kl = opaque64(G.key[2]);
    if (rem & 2u) kl = opaque64(G.key[1]);
    if (rem & 1u) kl = opaque64(G.key[0]);
    k0 = readlane64(kl, src);
    uint32_t mb = 0;
#pragma unroll
    for (int j = 0; j < VM_R; ++j)
      if (((rem >> j) & 1u) && G.key[j] == k0) mb |= 1u << j;
    const uint32_t cnt = popc_sel(mb);
    if (cnt >= LEADER_MIN_ROWS) {
      lead_mb = mb;
      lead_cnt = cnt;
      break;
    }
    ball &= ~__ballot(mb != 0);  // next candidate: a lane none of whose rows matched
  }
  rem &= ~lead_mb;
  // ---- LDS slots of the rows (claiming as they go), and of the dominant key (one lane looks it up for the wave)
  int32_t lsl = SLOT_NONE;
  if constexpr (ROLLED) {
    uint32_t pend = rem;
#pragma unroll 1
    for (int it = 0; it < VM_R; ++it) {
      G.slot[0] = (pend & 1u) ? lds_find_or_claim(L, G.key[0]) : SLOT_NONE;  // -1 = SLOT_LDS_MISS
      pend = rot4(pend);
      QE_ROT4(G.key);
      QE_ROT4(G.slot);
    }
  } else {
    lds_find_or_claim4(L, G.key, rem, sp.keys != nullptr ? SPILL_MODE_PROBES : LDS_MAX_PROBE, G.slot);
  }
  if (lead_cnt != 0 && (int)lane == src) lsl = lds_find_or_claim(L, k0);
  // ---- keys the LDS table has no room for go to the global table (rare: rolled loop)
  uint32_t miss = 0;
#pragma unroll
  for (int j = 0; j < VM_R; ++j)
    if (G.slot[j] == SLOT_LDS_MISS) miss |= 1u << j;
  const bool lead_miss = lsl == SLOT_LDS_MISS;
  if (sp.keys) {
    // high-cardinality mode: these rows go to the wave's log, in lane order
    const uint64_t any = __ballot(miss != 0);
    if (any != 0) {
      const uint32_t mine = (uint32_t)__popc(miss);
      // exclusive prefix of `mine` over the lanes: two ballots per bit of a 3-bit count
      uint32_t before = 0, total = 0;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const uint64_t m = __ballot(((mine >> b) & 1u) != 0);
        before += mbcnt64(m) << b;
        total += (uint32_t)__popcll(m) << b;
      }
      G.spill_mb = miss;
      G.spill_pos = sp.base + sp.cursor + before;
      uint32_t k = 0;
#pragma unroll
      for (int j = 0; j < VM_R; ++j) {
        if ((miss >> j) & 1u) {
          sp.keys[(G.spill_pos + k) << sp.pairs] = G.key[j];
          G.slot[j] = SLOT_NONE;
        }
        k += (miss >> j) & 1u;
      }
      sp.cursor += total;
      rem &= ~miss;
    }
  } else if (__ballot(miss != 0 || lead_miss) != 0) {
#pragma unroll 1
    for (int it = 0; it <= VM_R; ++it) {
      const bool lead_it = it == 0;
      const bool act = lead_it ? lead_miss : ((miss & 1u) != 0);
      const uint64_t kcur = lead_it ? k0 : G.key[0];
      int32_t sl = lead_it ? lsl : G.slot[0];
      if (act) {
        int64_t g = global_find_or_claim(T, kcur, mix64(kcur));
        sl = g < 0 ? SLOT_NONE : (int32_t)(-2 - (int32_t)g);
      }
      if (lead_it) {
        lsl = sl;
      } else {
        G.slot[0] = sl;
        miss = rot4(miss);
        QE_ROT4(G.key);
        QE_ROT4(G.slot);
      }
    }
  }
  // ---- row counts
  if (lead_cnt != 0 && (int)lane == src && lsl != SLOT_NONE) {
    if (lsl >= 0)
      atomicAdd(&L.rowcnt[lsl], lead_cnt);
    else
      atomicAdd(&T.rowcnt[(uint32_t)(-2 - lsl)], (unsigned long long)lead_cnt);
  }
  G.lead_slot = lead_cnt != 0 ? __builtin_amdgcn_readlane(lsl, src) : SLOT_NONE;
  G.lead_mb = G.lead_slot == SLOT_NONE ? 0u : lead_mb;
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    if ((rem >> j) & 1u) {
      if (G.slot[j] >= 0)
        atomicAdd(&L.rowcnt[G.slot[j]], 1u);
      else if (G.slot[j] != SLOT_NONE)
        atomicAdd(&T.rowcnt[(uint32_t)(-2 - G.slot[j])], 1ull);
      else
        rem &= ~(1u << j);  // the global table is full: the host grows it and runs again
    }
  }
  G.rem = rem;
}

// Folds one evaluated input (values + validity of the 4 rows) into accumulator `idx`.
__device__ __forceinline__ void agg_fold(const LdsTable &L, const AggTable &T, uint32_t idx, uint32_t op, uint32_t in_type,
                                         bool track, const Slot &in, const AggRows &G, uint32_t lane) {
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
  unsigned long long *lacc = L.acc + idx * L.stride;
  unsigned long long *gacc = T.acc + (uint64_t)idx * T.stride;
  // wave-reduced contribution of the dominant key
  if (G.lead_slot != SLOT_NONE) {
    const uint32_t m = G.lead_mb & valid;
    uint64_t part = acc_identity(op);
#pragma unroll
    for (int j = 0; j < VM_R; ++j)
      if ((m >> j) & 1u) part = acc_combine(op, part, v[j]);
    part = wave_reduce_acc(op, part);
    const uint32_t part_cnt = popc_sel(m);
    if (lane == 0 && part_cnt != 0) {
      if (G.lead_slot >= 0) {
        acc_apply(lacc + G.lead_slot, op, part);
        if (track) atomicAdd(&L.nn[idx * L.stride + G.lead_slot], part_cnt);
      } else {
        const uint32_t g = (uint32_t)(-2 - G.lead_slot);
        acc_apply(gacc + g, op, part);
        if (track) atomicAdd(&T.nn[(uint64_t)idx * T.stride + g], (unsigned long long)part_cnt);
      }
    }
  }
  const uint32_t m = G.rem & valid;
#pragma unroll
  for (int j = 0; j < VM_R; ++j) {
    if ((m >> j) & 1u) {
      if (G.slot[j] >= 0) {
        acc_apply(lacc + G.slot[j], op, v[j]);
        if (track) atomicAdd(&L.nn[idx * L.stride + G.slot[j]], 1u);
      } else {
        const uint32_t g = (uint32_t)(-2 - G.slot[j]);
        acc_apply(gacc + g, op, v[j]);
        if (track) atomicAdd(&T.nn[(uint64_t)idx * T.stride + g], 1ull);
      }
    }
  }
}

// Workgroup's LDS table -> global table, once, at the end of the kernel.  Every wave walks its slots in step (the
// loop's trip count is the same for all lanes) so that the slots the wave claims in the global table are counted with
// one atomic; min / max go out as atomics straight away (a read first would be a dependent trip to HBM per group).
// EXCLUSIVE: no other workgroup of the launch holds any of this table's keys (the partition-wise aggregate: partitions
// share no key) - a group's row in the global table is then read, combined and stored plainly, and a group claimed
// just now is only stored; slots are still claimed by compare-and-swap (keys of different partitions meet in the table).
// Up to FLUSH_BATCH slots a thread at once when the table is at most that many times the workgroup wide and has at most
// FLUSH_BATCH_ACCS accumulators without validity counts (the wide workgroups of the many-groups mode): the key probes of
// all the thread's slots are in flight together, then the reads of every group's row, then the stores.  One slot after
// the other, accumulator by accumulator, a flush is a chain of some twenty-five dependent trips to memory - it measured
// 40 to 130 microseconds a partition, a fifth of the partition pass.
constexpr int FLUSH_BATCH = 5, FLUSH_BATCH_ACCS = 4, FLUSH_BATCH_PROBES = 6;

template <bool EXCLUSIVE>
__device__ __forceinline__ void agg_flush_batched(const LdsTable &L, const AggTable &T, const VmAcc *accs, int n_accs) {
  unsigned long long k[FLUSH_BATCH];
  uint64_t at[FLUSH_BATCH];
  long long g[FLUSH_BATCH];
  uint32_t live = 0, open = 0, claimed = 0;
#pragma unroll
  for (int r = 0; r < FLUSH_BATCH; ++r) {
    const uint32_t i = (uint32_t)r * blockDim.x + threadIdx.x;
    bool on = i < L.stride;
    k[r] = on ? L.keys[i] : AGG_EMPTY_KEY;
    if (on && i == L.cap) {
      // the reserved slot: the key equal to the empty marker, if it came (its home is the global table's reserved slot)
      on = k[r] != AGG_EMPTY_KEY;
      k[r] = AGG_EMPTY_KEY;
      g[r] = -1;
      if (on) {
        bool c = false;
        g[r] = global_find_or_claim<false>(T, AGG_EMPTY_KEY, 0, &c);
        claimed |= c ? 1u << r : 0u;
        live |= g[r] >= 0 ? 1u << r : 0u;
      }
      at[r] = 0;
      continue;
    }
    on = on && k[r] != AGG_EMPTY_KEY;
    g[r] = -1;
    at[r] = on ? global_home(T, k[r], mix64(k[r])) : 0;
    open |= on ? 1u << r : 0u;
  }
  const uint64_t mask = T.cap - 1;
#pragma unroll 1
  for (int probe = 0; probe < FLUSH_BATCH_PROBES && open != 0; ++probe) {
    unsigned long long cur[FLUSH_BATCH];
#pragma unroll
    for (int r = 0; r < FLUSH_BATCH; ++r) cur[r] = ((open >> r) & 1u) ? __hip_atomic_load(&T.keys[at[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    uint32_t want = 0;
#pragma unroll
    for (int r = 0; r < FLUSH_BATCH; ++r) want |= (((open >> r) & 1u) && cur[r] == AGG_EMPTY_KEY) ? 1u << r : 0u;
#pragma unroll
    for (int r = 0; r < FLUSH_BATCH; ++r)
      if ((want >> r) & 1u) cur[r] = atomicCAS(&T.keys[at[r]], AGG_EMPTY_KEY, k[r]);  // (the old value: empty = the slot is ours)
#pragma unroll
    for (int r = 0; r < FLUSH_BATCH; ++r) {
      const bool o = ((open >> r) & 1u) != 0, w = ((want >> r) & 1u) != 0;
      const bool ours = o && w && cur[r] == AGG_EMPTY_KEY;
      const bool found = o && (ours || cur[r] == k[r]);
      claimed |= ours ? 1u << r : 0u;
      live |= found ? 1u << r : 0u;
      g[r] = found ? (long long)at[r] : g[r];
      open &= found ? ~(1u << r) : ~0u;
      at[r] = (o && !found) ? ((at[r] + 1) & mask) : at[r];
    }
  }
  if (open != 0) {
    // a long probe sequence: the general walk, which also knows when to give the table up
#pragma unroll
    for (int r = 0; r < FLUSH_BATCH; ++r) {
      if ((open >> r) & 1u) {
        bool c = false;
        g[r] = global_find_or_claim<false>(T, k[r], mix64(k[r]), &c);
        claimed |= c ? 1u << r : 0u;
        live |= g[r] >= 0 ? 1u << r : 0u;
      }
    }
  }
  {
    // The slots claimed, counted into the table's group counter with ONE atomic a workgroup: the waves' totals meet in
    // the upper half of the LDS table's fill word first (nobody reads that any more).  A wave each is sixteen operations
    // a workgroup on one word of memory, some ten nanoseconds apiece and one after the other: four thousand of them as a
    // first pass ends, eight thousand in a partition pass - a tenth of a millisecond either way.
    const uint32_t fresh = (uint32_t)__popc(claimed);
    const unsigned long long total = wave_sum_u64(fresh);
    if (lane_id() == 0 && total != 0) atomicAdd(L.used, (uint32_t)total << 16);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t all = *(volatile uint32_t *)L.used >> 16;
      if (all != 0) atomicAdd(&T.ctrl[0], (unsigned long long)all);
    }
  }
  if (live == 0) return;
  if (EXCLUSIVE) {
    // nobody else touches these groups: read the rows of all of them, combine, store
    unsigned long long row[FLUSH_BATCH], val[FLUSH_BATCH][FLUSH_BATCH_ACCS];
#pragma unroll
    for (int r = 0; r < FLUSH_BATCH; ++r) {
      const bool old = ((live >> r) & 1u) && !((claimed >> r) & 1u);
      row[r] = old ? T.rowcnt[g[r]] : 0ull;
#pragma unroll
      for (int a = 0; a < FLUSH_BATCH_ACCS; ++a) val[r][a] = (a < n_accs && old) ? T.acc[(uint64_t)a * T.stride + g[r]] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < FLUSH_BATCH; ++r) {
      if (!((live >> r) & 1u)) continue;
      const uint32_t i = (uint32_t)r * blockDim.x + threadIdx.x;
      const bool fresh = ((claimed >> r) & 1u) != 0;
      T.rowcnt[g[r]] = row[r] + L.rowcnt[i];
#pragma unroll
      for (int a = 0; a < FLUSH_BATCH_ACCS; ++a) {
        if (a < n_accs) {
          const uint32_t op = accs[a].op;
          T.acc[(uint64_t)a * T.stride + g[r]] = acc_combine(op, fresh ? acc_identity(op) : val[r][a], L.acc[a * L.stride + i]);
        }
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < FLUSH_BATCH; ++r) {
      if (!((live >> r) & 1u)) continue;
      const uint32_t i = (uint32_t)r * blockDim.x + threadIdx.x;
      const uint32_t rc = L.rowcnt[i];
      if (rc) atomicAdd(&T.rowcnt[g[r]], (unsigned long long)rc);
#pragma unroll
      for (int a = 0; a < FLUSH_BATCH_ACCS; ++a) {
        if (a < n_accs) {
          const uint32_t op = accs[a].op;
          const unsigned long long v = L.acc[a * L.stride + i];
          unsigned long long *dst = &T.acc[(uint64_t)a * T.stride + g[r]];
          if (v != acc_identity(op) || op == ACC_SUM_F64) {
            if (op == ACC_MIN) atomicMin(dst, v);
            else if (op == ACC_MAX) atomicMax(dst, v);
            else acc_apply(dst, op, v);
          }
        }
      }
    }
  }
}

template <bool EXCLUSIVE = false>
__device__ __forceinline__ void agg_flush(const LdsTable &L, const AggTable &T, const VmAcc *accs, int n_accs) {
  __syncthreads();
  if (L.nn == nullptr && n_accs <= FLUSH_BATCH_ACCS && L.stride <= (uint32_t)FLUSH_BATCH * blockDim.x) {
    agg_flush_batched<EXCLUSIVE>(L, T, accs, n_accs);
    return;
  }
  const uint32_t rounds = (L.stride + blockDim.x - 1) / blockDim.x;
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t i = r * blockDim.x + threadIdx.x;
    bool live = i < L.stride;
    unsigned long long k = live ? L.keys[i] : AGG_EMPTY_KEY;
    if (live && i == L.cap) {
      live = k != AGG_EMPTY_KEY;  // reserved slot: claimed or not
      k = AGG_EMPTY_KEY;
    } else if (k == AGG_EMPTY_KEY) {
      live = false;
    }
    bool claimed = false;
    int64_t g = -1;
    if (live) g = global_find_or_claim<false>(T, k, mix64(k), &claimed);
    const uint64_t fresh = __ballot(claimed);
    if (fresh != 0 && lane_id() == (uint32_t)__builtin_ctzll(fresh)) atomicAdd(&T.ctrl[0], (unsigned long long)__popcll(fresh));
    if (g < 0) continue;
    const uint32_t rc = L.rowcnt[i];
    if (EXCLUSIVE) {
      T.rowcnt[g] = (claimed ? 0ull : T.rowcnt[g]) + rc;
      for (int a = 0; a < n_accs; ++a) {
        const uint32_t op = accs[a].op;
        unsigned long long *dst = &T.acc[(uint64_t)a * T.stride + g];
        *dst = acc_combine(op, claimed ? acc_identity(op) : *dst, L.acc[a * L.stride + i]);
        if (L.nn) {
          unsigned long long *dn = &T.nn[(uint64_t)a * T.stride + g];
          *dn = (claimed ? 0ull : *dn) + L.nn[a * L.stride + i];
        }
      }
      continue;
    }
    if (rc) atomicAdd(&T.rowcnt[g], (unsigned long long)rc);
    for (int a = 0; a < n_accs; ++a) {
      const uint32_t op = accs[a].op;
      const unsigned long long val = L.acc[a * L.stride + i];
      unsigned long long *dst = &T.acc[(uint64_t)a * T.stride + g];
      if (val != acc_identity(op) || op == ACC_SUM_F64) {
        if (op == ACC_MIN) atomicMin(dst, val);
        else if (op == ACC_MAX) atomicMax(dst, val);
        else acc_apply(dst, op, val);
      }
      const uint32_t nn = L.nn ? L.nn[a * L.stride + i] : 0u;
      if (nn) atomicAdd(&T.nn[(uint64_t)a * T.stride + g], (unsigned long long)nn);
    }
  }
}

__device__ __forceinline__ void agg_lds_clear(const LdsTable &L, const VmAcc *accs, int n_accs) {
  for (uint32_t i = threadIdx.x; i < L.stride; i += blockDim.x) {
    L.keys[i] = AGG_EMPTY_KEY;
    L.rowcnt[i] = 0;
    for (int a = 0; a < n_accs; ++a) {
      L.acc[a * L.stride + i] = acc_identity(accs[a].op);
      if (L.nn) L.nn[a * L.stride + i] = 0;
    }
  }
  if (L.door)
    for (uint32_t i = threadIdx.x; i < DOOR_ENTRIES; i += blockDim.x) L.door[i] = 0;
  if (threadIdx.x == 0) *L.used = 0;
  __syncthreads();
}

}  // namespace qe
