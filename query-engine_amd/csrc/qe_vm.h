// Encoding shared by the host-side expression compiler and the device kernels.
//
// A physical expression tree (physical_expressions.rs:22-24 in the reference)
// is lowered to postfix code for a small stack machine.  One wavefront
// evaluates one instruction for 256 rows at a time (4 rows per lane), so an
// instruction's decode cost is shared by 256 rows and every intermediate value
// stays in vector registers: no per-node array is materialised (the reference
// allocates one per node, physical_expressions.rs:243-268).
#pragma once
#include <stdint.h>

namespace qe {

constexpr int VM_R = 4;              // rows per lane per wave-step
constexpr int VM_WAVE_ROWS = 256;    // rows per wave-step (64 lanes x 4)
constexpr int VM_BLOCK_THREADS = 256;
constexpr int VM_BLOCK_WAVES = 4;
constexpr int VM_BLOCK_ROWS = VM_WAVE_ROWS * VM_BLOCK_WAVES;  // 1024
constexpr int VM_MAX_COLS = 24;
constexpr int VM_MAX_LITS = 32;
constexpr int VM_MAX_INSN = 128;
constexpr int VM_MAX_DEPTH = 4;
constexpr int VM_MAX_OUTS = 12;
constexpr int VM_MAX_KEYS = 4;
constexpr int VM_MAX_ACCS = 16;
constexpr int AGG_WIDE_THREADS = 1024;  // the specialised aggregate kernels' wide launch: one workgroup of 16 waves per compute unit
constexpr int VM_NCACHE = 6;          // columns whose loads are issued together at the start of a wave-step

// Filter tiles: 4 steps of 1024 rows = 4096 rows per ticket.
constexpr int FILTER_STEPS = 4;
constexpr int FILTER_TILE_ROWS = VM_BLOCK_ROWS * FILTER_STEPS;

// Instruction word: op[7:0] | depth[11:8] | arg[31:12].
// `depth` is the stack depth before the instruction executes; the host
// computes it, so the kernel addresses stack slots with compile-time indices.
enum VmOp : uint32_t {
  OP_END = 0,
  OP_PUSH_COL32,    // arg = column: int32 values, kept sign-extended in 64 bits
  OP_PUSH_COL64,    // arg = column: int64 / float64 raw bits
  OP_PUSH_COLBOOL,  // arg = column: bit-packed boolean -> 0/1
  OP_PUSH_LIT,      // arg = literal index: raw 64 bits, always valid
  OP_PUSH_ROWID,    // global row index as int64 (arg unused)
  OP_SWAP,          // exchange the two top slots
  // arithmetic (wrapping two's complement for integers; IEEE, no FMA, for f64)
  OP_ADD_I32, OP_SUB_I32, OP_MUL_I32, OP_DIV_I32, OP_MOD_I32,
  OP_ADD_I64, OP_SUB_I64, OP_MUL_I64, OP_DIV_I64, OP_MOD_I64,
  OP_ADD_F64, OP_SUB_F64, OP_MUL_F64, OP_DIV_F64,
  // comparisons -> 0/1.  The I64 forms also serve int32 (sign-extended) and boolean.
  OP_EQ_I64, OP_NE_I64, OP_LT_I64, OP_LE_I64, OP_GT_I64, OP_GE_I64,
  OP_EQ_F64, OP_NE_F64, OP_LT_F64, OP_LE_F64, OP_GT_F64, OP_GE_F64,
  OP_AND, OP_OR,
  // utf8 comparisons push their boolean result directly.
  OP_UTF8_EQ_LIT, OP_UTF8_NE_LIT,  // arg = col | lit<<8 (lits[lit] = device ptr, lits[lit+1] = length)
  OP_UTF8_EQ_COL, OP_UTF8_NE_COL,  // arg = colA | colB<<8
  OP_HASH_UTF8,                    // arg = column: 64-bit hash of the string (group key), null string -> null
  OP_I64_TO_F64,                   // top slot: int64 -> float64 (AVG over integers)
  // pushes out of the per-step column cache (arg = cache slot)
  OP_PUSH_CACHED64, OP_PUSH_CACHED32, OP_PUSH_CACHEDBOOL,
  OP_MIX64,                        // top slot: 64-bit finaliser of a hash (composite group keys)
  // utf8 column against a literal of at most 16 bytes, which travels in the program's literals instead of device memory:
  // arg = col | lit<<8, lits[lit] = bytes 0..7, lits[lit+1] = bytes 8..15 (little endian), lits[lit+2] = length
  OP_UTF8_EQ_LIT16, OP_UTF8_NE_LIT16,
  // Order-independent float64 sums: a value becomes a 96-bit fixed-point integer in units of 2^(S - 96), S = lits[arg],
  // whose three 32-bit limbs (signed like the value) are summed as integers by three accumulators; integer addition is
  // associative, so the result does not depend on the order the rows meet the accumulators in.
  OP_FIX_EXP,                      // top slot: float64 -> its biased exponent field as int64 (0 for infinities and NaN)
  OP_FIX_LIMB0, OP_FIX_LIMB1, OP_FIX_LIMB2,  // top slot: float64 -> limb (0 for infinities and NaN); arg = literal index of S
  OP_FIX_SPECIAL,                  // top slot: float64 -> 1 for +inf, 2^32 for -inf, both for NaN, 0 otherwise
  OP_COUNT_
};

constexpr bool vm_op_is_cached_push(uint32_t op) { return op >= OP_PUSH_CACHED64 && op <= OP_PUSH_CACHEDBOOL; }
// Instructions the shape-specialised kernels implement: everything except the comparisons of a utf8 column with
// another column or a long literal, and the utf8 hash.
constexpr bool vm_op_specialisable(uint32_t op) { return op < OP_COUNT_ && !(op >= OP_UTF8_EQ_LIT && op <= OP_HASH_UTF8); }

inline uint32_t vm_insn(uint32_t op, uint32_t depth, uint32_t arg) { return op | (depth << 8) | (arg << 12); }

// Element types of a stack slot / an output column.
enum VmType : uint8_t { VT_I32 = 1, VT_I64 = 2, VT_F64 = 3, VT_BOOL = 4, VT_UTF8 = 5 };

struct VmCol {
  const void *values;
  const uint8_t *validity;  // nullptr = all valid
  const int32_t *offsets;   // utf8 only
  uint64_t reserved;
};

// Kind of a cached column's raw bits.
enum VmCacheKind : uint8_t { CK_64 = 0, CK_32 = 1, CK_BOOL = 2 };

struct VmProgram {
  VmCol cols[VM_MAX_COLS];
  uint64_t lits[VM_MAX_LITS];
  uint32_t insn[VM_MAX_INSN];
  // The first `cache_n` distinct fixed-width columns a program touches are loaded into
  // registers up front, all loads in flight together, instead of one round trip per use.
  uint8_t cache_col[8];   // program column of each cache slot
  uint8_t cache_kind[8];  // VmCacheKind
  uint32_t cache_n;
  uint32_t cache_pad;
};

struct VmOut {
  void *values;        // int32/int64/float64 packed values, or boolean bitmap words
  uint32_t *validity;  // bitmap words or nullptr (output cannot be null)
  uint32_t pc;         // first instruction of the expression
  uint8_t type;        // VmType
  uint8_t pad[3];
};

struct ProjectArgs {
  VmProgram prog;
  VmOut outs[VM_MAX_OUTS];
  int32_t n_outs;
  int32_t pad;
  int64_t n_rows;
};

// Layout of the per-launch scratch words used by the filter kernel.
struct FilterScratch {
  unsigned long long ticket;     // next tile to hand out
  unsigned long long total;      // number of selected rows (written by the last tile)
  unsigned long long error;      // non-zero: a bounded spin ran out
  unsigned long long pad[5];
  // followed by unsigned long long status[n_tiles]
};

// Tile geometry of the shape-specialised filter kernels, a function of the number of cached
// columns (registers per lane): waves per workgroup, steps per tile.  Host and device use this
// one definition.
constexpr int filter_static_waves(int cache_n) { return cache_n <= 3 ? 16 : 8; }
constexpr int filter_static_steps(int cache_n) { return cache_n <= 3 ? 2 : cache_n <= 5 ? 4 : 3; }
constexpr int filter_static_tile_rows(int cache_n) { return filter_static_steps(cache_n) * filter_static_waves(cache_n) * VM_WAVE_ROWS; }

struct FilterArgs {
  VmProgram prog;
  VmOut outs[VM_MAX_OUTS];
  int32_t n_outs;
  uint32_t pred_pc;
  uint32_t pred_cache_mask;  // cache slots the predicate reads
  uint32_t out_cache_mask;   // cache slots the output expressions read
  int64_t n_rows;
  int64_t n_tiles;
  FilterScratch *scratch;
};

// Accumulator operations of the hash aggregate.
enum AccOp : uint8_t {
  ACC_SUM_I64 = 1,  // wrapping int64 sum of sign-extended integers
  ACC_SUM_F64 = 2,  // float64 sum (atomic order: not bit-reproducible)
  ACC_MIN = 3,      // min / max on an order-preserving unsigned image of the value
  ACC_MAX = 4,
  ACC_COUNT = 5     // number of valid inputs
};

struct VmKey {
  uint32_t pc;
  uint8_t type;   // VmType (VT_I32, VT_I64, VT_BOOL)
  uint8_t shift;  // bit position inside the packed 64-bit key
  uint8_t bits;   // field width
  uint8_t pad;
};

struct VmAcc {
  uint32_t pc;       // input expression (unused when has_input == 0)
  uint8_t op;        // AccOp
  uint8_t in_type;   // VmType of the input
  uint8_t track_nn;  // keep a count of valid inputs next to the accumulator
  uint8_t has_input; // 0: COUNT(*) style, every row counts
};

// Global open-addressing group table (structure of arrays, `stride` slots per
// array; slot `cap` is reserved for the one key equal to the empty marker).
struct AggTable {
  unsigned long long *keys;
  unsigned long long *rowcnt;
  unsigned long long *acc;  // acc[a * stride + slot]
  unsigned long long *nn;   // nn[a * stride + slot]
  unsigned long long *ctrl; // [0] = groups claimed, [1] = overflow flag, [2] = output rows drawn by a deferred finalize
  uint64_t cap;             // power of two
  uint64_t stride;          // cap + 1 rounded up to 64
};

constexpr unsigned long long AGG_EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;

// Dense-key bins of the specialised aggregate kernels: packed keys below DENSE_KEYS are counted and
// accumulated in per-lane LDS bins addressed by the key itself (rows[key][lane], acc[a][key][lane]).
constexpr uint32_t DENSE_KEYS = 16;
constexpr uint32_t dense_bins_bytes(int n_accs) { return DENSE_KEYS * 64u * (4u + 8u * (uint32_t)n_accs); }

// The aggregate kernel runs one list of tasks per wave-step, so that the expression
// interpreter is instantiated once in the kernel (instruction-cache footprint).
enum AggRole : uint8_t {
  ROLE_PRED = 1,    // evaluate, AND into the active rows
  ROLE_KEY = 2,     // evaluate, pack into the 64-bit key (idx = key field)
  ROLE_LOOKUP = 3,  // no expression: find / claim the group slot of every active row
  ROLE_ACC = 4      // evaluate (if has_eval), fold into accumulator idx
};
struct VmTask {
  uint32_t pc;
  uint8_t role;
  uint8_t idx;
  uint8_t has_eval;
  uint8_t pad;
};
constexpr int VM_MAX_TASKS = 2 + VM_MAX_KEYS + VM_MAX_ACCS;

// High-cardinality mode.  Rows whose key finds no room in the workgroup's LDS table are not sent
// to the global table one atomic at a time (scattered global atomics run at a few per cent of the
// memory rate); each wave appends them (packed key + the distinct accumulator inputs) to a
// region of its own, and a second phase partitions that log by key hash and aggregates it
// partition by partition in LDS.  keys == nullptr: off.
struct AggSpill {
  unsigned long long *keys;  // [capacity]
  unsigned long long *vals;  // [n_inputs][capacity] raw input values
  uint32_t *counts;          // [waves of the launch] rows each wave spilled
  uint64_t capacity;
  uint32_t rows_per_wave;    // size of one wave's region
  uint32_t n_inputs;
  // How the partition passes deal the log's segments (one per first-pass wave) to their workgroups: 0 = round-robin
  // (workgroup b takes b, b + blocks, ...), n = runs of n (workgroup b takes [b n, (b + 1) n): the segments of first-pass
  // workgroup b, which is what lets that workgroup count its own rows per partition).
  uint32_t seg_chunk;
  // 1: a log with one value column holds (key, value) pairs of 16 bytes in `keys` (row i at keys[2 i], keys[2 i + 1];
  // `vals` is not used): the first pass writes a row with one 16-byte store, the scatter reads it with one load.
  uint32_t pairs;
  // Not 0: no log at all - the first pass partitions its rows itself (qe_aggregate_direct_static).  `keys` then holds
  // pairs in one stretch of `direct_cap` rows (a multiple of 8) per partition and workgroup, stretch (p, b) at row
  // (p * workgroups + b) * direct_cap; AggArgs::cells[p * cells_stride + b] receives the rows written there and
  // counts[0] is set when some stretch was too small for its rows (which are then missing: the caller runs the batch
  // again, through the log).
  uint32_t direct_cap;
  // Value columns of the log whose input can be null (bit c: column c): a null is logged as a value no input holds -
  // SPILL_NULL_F64 for a Float64 column (bit c of null_f64 set; an input NaN with that very payload is logged as the
  // plain quiet NaN), SPILL_NULL_INT for an Int32 one (kept sign-extended in 64 bits, so INT64_MIN is free) - and the
  // partition-wise aggregate leaves such a row out of the accumulators that read the column (it still counts as a row
  // of its group).
  uint16_t null_cols;
  uint16_t null_f64;
};
constexpr unsigned long long SPILL_NULL_F64 = 0x7FF8000000006E75ull, SPILL_NULL_INT = 0x8000000000000000ull, SPILL_PLAIN_NAN = 0x7FF8000000000000ull;

constexpr int AGG_MORE_BATCHES = 3;  // batches a first pass of the many-groups mode takes on top of its first

struct AggArgs {
  VmProgram prog;
  VmKey keys[VM_MAX_KEYS];
  VmAcc accs[VM_MAX_ACCS];
  VmTask tasks[VM_MAX_TASKS];
  AggTable table;
  int32_t n_keys;
  int32_t n_accs;
  int32_t n_tasks;
  int32_t dense_off;   // byte offset of the dense-key bins in dynamic LDS (specialised kernels); 0 = none
  int64_t n_rows;
  uint32_t lds_cap;    // LDS table slots (power of two)
  uint32_t lds_limit;  // stop claiming LDS slots past this many
  AggSpill spill;
  uint8_t spill_col[VM_MAX_ACCS];  // value column an accumulator's input is logged to; 0xff: not this one
  uint32_t block_threads;          // host only: threads per workgroup of the launch (0 = VM_BLOCK_THREADS)
  uint32_t door_off;               // byte offset of the admission tags in dynamic LDS (high-cardinality mode); 0 = none
  // High-cardinality first pass counting its logged rows per partition as it goes (saves the count pass over the log):
  // cells[p * gridDim.x + blockIdx.x] = rows of partition p this workgroup logged, *cells_total += rows logged;
  // the histogram lives at hist_off in dynamic LDS.  cells == nullptr: not asked for.
  int32_t *cells;
  unsigned long long *cells_total;
  int32_t cell_bits;
  uint32_t hist_off;
  uint32_t tune;                   // bit 0: no hot keys in registers for this launch; bit 1 (host): the caller promised many groups;
                                   // bit 2: the many-groups first pass leaves its tables unflushed (a trial launch); bit 3: its waves take the high priority in turn
  // Columns in a row of the cell table (0 = the launch's workgroups).  Several launches whose logs are partitioned
  // together share one table, each with a block of columns of its own: `cells` then points at this launch's first.
  uint32_t cells_stride;
  // Further batches a many-groups first pass takes in the same launch - its workgroups' tables, filled and flushed
  // once, serve them all - when the program reads cached columns only: batch b (1 .. n_more) has more_rows[b - 1]
  // rows and its columns, by cache slot, in more_cols[b - 1].  The log's regions and the stretches are sized for
  // the rows of all of them.
  int32_t n_more;
  int32_t more_pad;
  int64_t more_rows[AGG_MORE_BATCHES];
  VmCol more_cols[AGG_MORE_BATCHES][VM_NCACHE];
};

// Does the workgroup's LDS table keep per-accumulator counts of valid inputs?  Host and device size the table alike.
inline __host__ __device__ bool agg_tracks_validity(const VmAcc *accs, int n_accs) {
  for (int a = 0; a < n_accs; ++a)
    if (accs[a].track_nn) return true;
  return false;
}

// The first pass that partitions its rows itself: rows a partition's ring in LDS holds (two 128-byte lines of eight
// pairs), the most partitions it serves, and the LDS a workgroup takes (rings, three words a partition, the list of
// lines to write).
constexpr int DIRECT_RING = 16;
constexpr int DIRECT_RING_STRIDE = 17;  // rows from one ring to the next
constexpr int DIRECT_MAX_PARTS = 512;
constexpr size_t direct_lds_bytes(int bits) { return ((size_t)1 << bits) * (DIRECT_RING_STRIDE * 16 + 8 + 4 + 2 * 4) + 64; }

// Rows a first pass partitioned itself (AggSpill::direct_cap): stretch (p, b) of `cap` pairs at rows[(p * blocks + b) * cap],
// counts[p * counts_stride + b] of them filled.
struct PartDirect {
  const unsigned long long *rows;
  const int32_t *counts;
  uint32_t cap;
  uint32_t counts_stride;
};
constexpr int PART_MAX_DIRECT = 8;         // such sources per launch of the partition-wise aggregate
constexpr int PART_MAX_DIRECT_UNITS = 2048;  // ... and stretches per partition over all of them

// Second phase of the high-cardinality mode: the spilled rows, partitioned by key hash.
struct PartArgs {
  const unsigned long long *keys;
  const unsigned long long *vals;  // [n_inputs][capacity]
  uint64_t capacity;
  const int32_t *offsets;          // offsets[p * stride] = first row of partition p
  int32_t n_parts;
  int32_t stride;
  int64_t total;                   // rows in all partitions
  VmAcc accs[VM_MAX_ACCS];
  uint8_t acc_col[VM_MAX_ACCS];    // value column of each accumulator
  int32_t n_accs;
  int32_t n_inputs;
  AggTable table;
  uint32_t lds_cap;
  uint32_t lds_limit;
  unsigned long long *ticket;      // zeroed before the launch
  int32_t ablate;                  // timing experiments only (QE_PART_ABLATE): 1 no flush, 2 no fold, 4 no lookup
  int32_t interleaved;             // 1: `keys` holds (key, value) pairs of 16 bytes, one value column (the one-pass scatter's output)
  // On top of (or, with keys == nullptr, instead of) the partitioned copy: rows that first passes partitioned themselves.
  int32_t n_direct;
  int32_t direct_blocks;           // workgroups of those first passes = stretches per partition and source
  PartDirect direct[PART_MAX_DIRECT];
  // More groups than n_parts tables hold: a partition is folded in 1 << sub_bits passes over its rows, each keeping
  // the rows whose next sub_bits hash bits (below the part_bits that name the partition) are the pass's number.
  int32_t part_bits, sub_bits;
  uint16_t null_cols, null_f64;  // AggSpill::null_cols / null_f64 of the log the rows come from
};

constexpr int FIX_FRACTION_BITS = 96;  // fixed-point digits below 2^S
constexpr int FIX_ACCS = 4;            // accumulators of one order-independent sum: three limbs and the infinity / NaN counts
constexpr bool vm_op_is_fix(uint32_t op) { return op >= OP_FIX_EXP && op <= OP_FIX_SPECIAL; }

// One output column of the finalize pass.
enum FinKind : uint8_t {
  FIN_KEY = 1,      // unpack key field -> values (type in `type`)
  FIN_ACC_I64 = 2,  // raw int64 accumulator (truncated to int32 when type == VT_I32)
  FIN_ACC_F64 = 3,  // raw float64 accumulator
  FIN_ACC_ORD = 4,  // inverse of the order-preserving image (min / max)
  FIN_COUNT = 5,    // the count source itself
  FIN_AVG = 6,      // acc[a] (float64) / count source
  FIN_FIX_SUM = 7,  // order-independent sum: limbs acc[a .. a+2] in units of 2^(scale - 96), acc[a+3] = infinity / NaN counts
  FIN_FIX_AVG = 8   // that sum / count source
};
// Where a column's count comes from.
enum FinCnt : uint8_t { CNT_ROWS = 0, CNT_NN = 1, CNT_ACC = 2 };
// When a result is null.
enum FinNull : uint8_t { NULL_NEVER = 0, NULL_IF_COUNT_ZERO = 1, NULL_IF_ACC_ALL_ONES = 2, NULL_IF_ACC_ZERO = 3 };

struct FinOut {
  void *values;
  uint32_t *validity;  // nullptr or bitmap words (pre-zeroed)
  uint8_t kind;        // FinKind
  uint8_t type;        // VmType of the output
  int8_t a;            // accumulator index (FIN_KEY: key field index)
  int8_t b;            // accumulator index of the count source (CNT_NN / CNT_ACC)
  uint8_t shift;       // FIN_KEY
  uint8_t bits;        // FIN_KEY
  uint8_t cnt_src;     // FinCnt
  uint8_t null_mode;   // FinNull
  int16_t scale;       // FIN_FIX_*: S
  uint8_t pad[6];
};

struct FinalizeArgs {
  AggTable table;
  FinOut outs[VM_MAX_KEYS + 2 * VM_MAX_ACCS];
  int32_t n_outs;
  int32_t pad;
  unsigned long long *out_count;  // rows written
  unsigned long long *overflow_sink;  // when set: incremented if the table's overflow flag is up (deferred checks)
};

}  // namespace qe
