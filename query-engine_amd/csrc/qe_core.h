// Host-side object model of the backend: context + device memory pool, Arrow
// column/batch containers, expression trees.  Mirrors the reference's value
// model (columnar_value.rs) and expression structs (physical_expressions.rs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/qe_gpu.h"
#include "qe_kernels.h"

namespace qe {

struct Failure {
  qe_status code;
  std::string msg;
};
[[noreturn]] void fail(qe_status code, const std::string &msg);
void hip_check(hipError_t e, const char *what);
#define QE_HIP(expr) ::qe::hip_check((expr), #expr)

struct Context : std::enable_shared_from_this<Context> {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  uint32_t flags = QE_FLAG_REFERENCE_SCALAR_ARRAY_SWAP;
  DeviceInfo dev{};
  // size-bucketed free lists: hipMalloc/hipFree are far too slow to call per batch
  std::map<size_t, std::vector<void *>> pool;
  size_t pool_bytes = 0;
  void *filter_scratch = nullptr;
  size_t filter_scratch_cap = 0;
  unsigned long long *counters = nullptr;  // small device scratch for counts
  // A ring of pinned buffers for bulk host -> device copies (file scans), used in turn: reader threads fill
  // the next ones while the copies out of the earlier ones are still in flight.  They outlive the files that use them.
  struct Staging {
    void *ptr = nullptr;
    size_t cap = 0;
    hipEvent_t copied = nullptr;  // recorded after the copy out of the buffer
    bool pending = false;
    // for a chunk that reader threads copy to the device slice by slice, on the copy stream: `fence` orders those
    // copies behind whatever the context's stream still does with the device block they go to, `images` is recorded
    // behind the last of them
    hipEvent_t fence = nullptr, images = nullptr;
  };
  hipStream_t copy_stream = nullptr;  // host -> device copies of file scans, beside the context's stream (made on first use)
  static constexpr int STAGING_BUFFERS = 8;
  Staging staging[STAGING_BUFFERS];
  int next_staging = 0;
  // Plain host memory that scans reuse for compressed bytes (a fresh allocation per column chunk
  // costs a page fault per 4 KiB on its first touch).
  std::vector<unsigned char> host_scratch;
  // Aggregates that returned before the device was done (QE_FLAG_DEFERRED_GROUP_COUNT) leave their
  // overflow flag in one device word (their finalize kernel adds to it); the next synchronise reads it.
  bool deferred_pending = false;
  unsigned long long *deferred_sink();  // the device word
  void check_deferred();                // with the stream idle: fails if any such aggregate overflowed
  unsigned long long *host_counters = nullptr;  // pinned mirror
  // Selections that returned before the device was done (QE_FLAG_DEFERRED_ROW_COUNT): each output batch owns a block of
  // pinned words its row count (and its string columns' byte totals) travel to, and an event recorded behind that copy.
  // Blocks and events are handed out from free lists (pinned memory comes in slabs); a batch gives them back when the
  // last copy of it goes.
  static constexpr int PENDING_WORDS = 128;
  std::vector<unsigned long long *> pending_free, pending_slabs;
  std::vector<hipEvent_t> pending_events;
  unsigned long long *take_pending_words();
  hipEvent_t take_pending_event();
  // profiling
  bool profiling = false;
  int64_t profile_min_rows = 0;  // > 0: only launches over at least this many rows are timed
  bool prof_open = false;        // the launch between prof_begin and prof_end is being timed
  struct LaunchRec {
    const char *name;
    hipEvent_t start, stop;
  };
  std::vector<LaunchRec> launches;

  void *alloc(size_t bytes, size_t *actual);
  void release(void *p, size_t actual);
  void bind() const;
  void sync() const;
  void prof_begin(const char *name, int64_t rows = -1);  // rows: what the launch covers, when known
  void prof_end(const char *actual_name = nullptr);  // the kernel variant the launcher picked
  ~Context();
};

struct Buffer {
  // Owned device blocks keep their context alive: qe_context_free only drops the caller's
  // reference, the pool is torn down when the last buffer has gone back to it.
  std::shared_ptr<Context> ctx_ref;
  Context *ctx = nullptr;
  void *ptr = nullptr;
  size_t bytes = 0;   // usable bytes
  size_t actual = 0;  // pool bucket size (device, owned)
  int space = QE_MEM_HOST;
  bool owned = false;
  std::shared_ptr<Buffer> parent;  // a view into another buffer keeps it alive
  ~Buffer();
};
using BufferPtr = std::shared_ptr<Buffer>;

BufferPtr device_buffer(Context *ctx, size_t bytes);
BufferPtr host_buffer(size_t bytes);
BufferPtr borrowed_buffer(const void *p, size_t bytes, int space);
BufferPtr sub_buffer(const BufferPtr &parent, size_t offset, size_t bytes);

struct Column {
  int type = 0;
  int space = QE_MEM_DEVICE;
  int64_t length = 0;
  int64_t null_count = 0;  // -1 = not counted yet
  // Set by a producer that has seen every row valid although the column carries a validity buffer (an
  // optional Parquet column without nulls): kernels may then leave the bitmap alone.
  bool known_all_valid = false;
  BufferPtr validity, offsets, values;
  int64_t values_bytes = 0;
  const void *values_ptr() const { return values ? values->ptr : nullptr; }
  const uint8_t *validity_ptr() const { return validity ? (const uint8_t *)validity->ptr : nullptr; }
  const int32_t *offsets_ptr() const { return offsets ? (const int32_t *)offsets->ptr : nullptr; }
};

// The row count of a Selection's output that the host has not read yet (QE_FLAG_DEFERRED_ROW_COUNT): the batch's
// columns are sized for every input row, `rows` is that bound, and the words the device copies to `host` behind the
// kernels say how many rows (and, per string column, bytes) there really are.
struct PendingRows {
  std::shared_ptr<Context> ctx_ref;
  unsigned long long *host = nullptr;  // four words per string column (ticket, bytes, error, -), then at `filter_at` the filter's ticket / rows / look-back error
  int filter_at = 0;
  hipEvent_t ready = nullptr;          // recorded behind the last copy into `host`
  std::vector<int> utf8_slot;          // per output column: its block of four words, or -1
  bool waited = false;
  void wait();  // returns once the words have arrived
  ~PendingRows();
};

struct Batch {
  Context *ctx = nullptr;
  std::vector<Column> cols;
  int64_t rows = 0;
  std::shared_ptr<PendingRows> pending;  // set: `rows` and the columns' lengths are upper bounds until batch_settle()
};
// Waits for a pending row count and gives the batch its final shape (no-op for any other batch).
void batch_settle(Batch &b);

size_t bitmap_bytes(int64_t n);   // padded to whole 32-byte groups
int type_width(int type);         // bytes per value of a fixed-width type (0 otherwise)
const char *type_name(int type);

// ------------------------------------------------------------- expressions
enum ExprKind {
  EK_COLUMN = 0,
  EK_LITERAL = 1,
  EK_BINARY = 2,
  // internal, made by the aggregate lowering only:
  EK_HASH_UTF8 = 3,  // 64-bit hash of the utf8 column `l` (a string group key is aggregated by its hash)
  EK_ROWID = 4,      // global row number: `ival` (rows of the earlier batches) + row in the batch
  // 64-bit hash of a tuple of key columns, built column by column: `l` = hash of the columns before
  // (null for the first), `r` = the column (any key type, utf8 included)
  EK_KEY_HASH = 5
};

struct Expr;
using ExprPtr = std::shared_ptr<const Expr>;

struct Expr {
  ExprKind kind = EK_COLUMN;
  int index = 0;       // EK_COLUMN
  int type = 0;        // EK_LITERAL: QE_TYPE_*
  bool valid = true;   // EK_LITERAL: false = null scalar
  int64_t ival = 0;    // INT32 / INT64 / BOOLEAN payload
  double fval = 0.0;   // FLOAT64 payload
  std::string sval;    // UTF8 payload
  int op = 0;          // EK_BINARY: QE_OP_*
  bool to_f64 = false; // EK_BINARY unused; internal cast marker (AVG over integers)
  // internal, made by the aggregate lowering for order-independent sums: a unary node over the float64 `l`
  // (FIX_EXPONENT: its exponent field; FIX_LIMB0 + k: limb k of its fixed-point image at scale `ival`; FIX_SPECIAL:
  // infinity / NaN marks), of type Int64
  int fix = 0;
  // FIX_SPECIAL: device address of the word the kernels set when a finite value does not fit the limbs at the scale
  // (0 = nobody is listening).  An address, not part of the expression's identity.
  uint64_t fix_overflow_word = 0;
  ExprPtr l, r;
};

struct AggExpr {
  int kind = 0;   // QE_AGG_*
  ExprPtr input;  // may be null for COUNT(*)
};

ExprPtr make_column(int index);
ExprPtr make_lit_i32(int32_t v);
ExprPtr make_lit_i64(int64_t v);
ExprPtr make_lit_f64(double v);
ExprPtr make_lit_bool(bool v);
ExprPtr make_lit_utf8(const std::string &s);
ExprPtr make_binary(int op, ExprPtr l, ExprPtr r);
ExprPtr make_hash_utf8(ExprPtr column);
ExprPtr make_key_hash(ExprPtr before, ExprPtr column);
ExprPtr make_rowid(int64_t base);
enum FixPart { FIX_NONE = 0, FIX_EXPONENT = 1, FIX_LIMB0 = 2, FIX_LIMB1 = 3, FIX_LIMB2 = 4, FIX_SPECIAL = 5 };
ExprPtr make_fix(int part, ExprPtr value, int scale, uint64_t overflow_word = 0);
std::string expr_to_string(const ExprPtr &e);
std::string expr_identity(const ExprPtr &e);  // equal exactly for equal trees (expr_to_string is the reference's ambiguous Display)
bool expr_is_scalar(const ExprPtr &e);

struct ColInfo {
  int type;
  bool nullable;
};
using Schema = std::vector<ColInfo>;
Schema schema_of(const Batch &b);
// The batch with the validity buffers dropped from columns known to hold no nulls.
Batch without_redundant_validity(const Batch &b);

// Type of the value an expression evaluates to over `schema`; throws the
// reference's error for out-of-range columns / unsupported operand types.
int expr_type(const ExprPtr &e, const Schema &schema);
bool expr_nullable(const ExprPtr &e, const Schema &schema);
// Folds every column-free subtree into a literal with the semantics of the
// reference's (Scalar, Scalar) arms (physical_expressions.rs:204-208,269-321)
// and applies the (Scalar, Array) operand swap when `flags` asks for it.
ExprPtr expr_normalize(const ExprPtr &e, uint32_t flags);
// Replace Column(i) by outputs[i].
ExprPtr expr_substitute(const ExprPtr &e, const std::vector<ExprPtr> &outputs);
// What a (normalised) expression takes of one kernel's program: instructions without the closing
// OP_END, literal slots (an upper bound: equal literals share one), stack slots when the deeper
// operand of every node is evaluated first.
struct ExprCost {
  int insns = 0, lits = 0, slots = 0;
};
ExprCost expr_cost(const ExprPtr &e, const Schema &schema);
// Batch columns the expression reads.
void expr_columns(const ExprPtr &e, std::map<int, int> &seen);

// ----------------------------------------------------- expression compiler
struct Compiler {
  Context *ctx;
  const Batch *batch;
  Schema schema;
  VmProgram prog{};
  int n_insn = 0, n_lits = 0, n_cols = 0;
  std::map<int, int> colmap;           // batch column -> program column
  std::map<int, int> cachemap;         // batch column -> cache slot
  uint32_t last_cache_mask = 0;        // cache slots read by the expression added last
  uint32_t all_cache_mask = 0;         // ... by every expression added so far
  std::vector<BufferPtr> keepalive;    // device copies of utf8 literals
  Compiler(Context *c, const Batch *b);
  // Space check before adding an expression: instruction / literal / column needs.
  bool fits(const ExprPtr &e) const;
  // Emits code ending in OP_END, returns the start pc.
  uint32_t add(const ExprPtr &e);
  uint32_t add_rowid();
 private:
  int need(const ExprPtr &e) const;
  void emit(const ExprPtr &e, int depth);
  int col_slot(int index);
  int lit_slot(uint64_t bits);
  void push_insn(uint32_t op, uint32_t depth, uint32_t arg);
};

}  // namespace qe
