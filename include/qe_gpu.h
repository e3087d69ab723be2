/*
 * qe_gpu.h — C ABI of the MI355X (gfx950) execution backend for the
 * query-engine physical plan (Scan -> Selection -> Projection -> Aggregate).
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes / opaque
 * handles and returns a `qe_status` (0 = OK).  Nothing aborts or throws across
 * this boundary: where the reference panics (`unwrap()`), the matching call
 * returns an error code and `qe_last_error_message()` holds the text.
 *
 * The reference (Rust) has no FFI layer; the interface replaced here is the
 * set of Rust traits / constructors cited on each declaration
 * (paths relative to /root/reference/query-engine/src/).
 *
 * Threading: handles are not thread-safe (the reference has no Send/Sync
 * bounds: physical_plan/mod.rs:58).  One context = one HIP device + one
 * stream; use one context per host thread / per GPU.
 *
 * Data layout: Arrow columnar.  Primitive = values buffer + optional validity
 * bitmap (LSB first, 1 = valid); Boolean = bit-packed values + validity;
 * Utf8 = int32 offsets[n+1] + bytes + validity (columnar_value.rs:1-5).
 */
#ifndef QE_GPU_H
#define QE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t qe_status;

/* Status codes.  10..24 map 1:1 onto the variants of the reference's
 * `enum Error` (error.rs:3-35), in declaration order. */
enum {
  QE_OK = 0,
  QE_END = 1, /* iterator exhausted (the reference's `None` from next()) */
  QE_ERR_EXCEEDING_BOUNDS = 10,               /* error.rs:5  */
  QE_ERR_NO_FIELD_IN_LOGICAL_PLAN = 11,       /* error.rs:7  */
  QE_ERR_DIFFERENT_SIZES = 12,                /* error.rs:9  */
  QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED = 13, /* error.rs:11 */
  QE_ERR_PHYSICAL_PLAN_NOT_SUPPORTED = 14,    /* error.rs:13 */
  QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED = 15,    /* error.rs:15 */
  QE_ERR_PRIMITIVE_TYPE_NOT_SUPPORTED = 16,   /* error.rs:17 */
  QE_ERR_MISSING_CHILDREN = 17,               /* error.rs:19 */
  QE_ERR_MISSING_INPUT_PHYSICAL_PLAN = 18,    /* error.rs:21 */
  QE_ERR_EMPTY_HASHMAP_FOR_AGGREGATE = 19,    /* error.rs:23 */
  QE_ERR_DOWNCAST = 20,                       /* error.rs:25 */
  QE_ERR_SCALAR_TO_ARRAY = 21,                /* error.rs:27 */
  QE_ERR_NO_BOOLEAN_ARRAY_FOR_FILTER = 22,    /* error.rs:29 */
  QE_ERR_IO = 23,                             /* error.rs:31 */
  QE_ERR_ARROW = 24,                          /* error.rs:33 */
  /* Backend-only conditions (no reference counterpart). */
  QE_ERR_INVALID_ARGUMENT = 40,
  QE_ERR_DEVICE = 41,        /* a HIP call failed / no gfx950 device */
  QE_ERR_OUT_OF_MEMORY = 42,
  QE_ERR_AGG_CAPACITY = 43,  /* group table full even after growth */
  QE_ERR_INTERNAL = 44
};

/* Column element types.  INT32/FLOAT64/UTF8/BOOLEAN are the four the
 * reference's ColumnExpression accepts (physical_expressions.rs:37-60);
 * INT64 is the extension the north-star configs use.  TIMESTAMP_NS (int64
 * nanoseconds since the epoch, arrow's Timestamp(Nanosecond, None)) is what
 * the Parquet scan yields for INT96 columns (tests/mod.rs:31): such a column
 * passes through scans and selections, but no expression computes on it (the
 * reference's ColumnExpression rejects it the same way). */
enum {
  QE_TYPE_INT32 = 1,
  QE_TYPE_INT64 = 2,
  QE_TYPE_FLOAT64 = 3,
  QE_TYPE_BOOLEAN = 4,
  QE_TYPE_UTF8 = 5,
  QE_TYPE_TIMESTAMP_NS = 6
};

enum { QE_MEM_HOST = 0, QE_MEM_DEVICE = 1 };

/* Binary operators.  EQ/NEQ: physical_expressions.rs:232-233; ADD..DIV:
 * physical_expressions.rs:346-349.  GT..OR and MOD exist only logically in the
 * reference (logical_expression.rs:284-294,342) and are rejected by its
 * planner (query_planner.rs:82); they are lowered here. */
enum {
  QE_OP_EQ = 1, QE_OP_NEQ = 2, QE_OP_GT = 3, QE_OP_GTEQ = 4, QE_OP_LT = 5,
  QE_OP_LTEQ = 6, QE_OP_AND = 7, QE_OP_OR = 8,
  QE_OP_ADD = 9, QE_OP_SUB = 10, QE_OP_MUL = 11, QE_OP_DIV = 12, QE_OP_MOD = 13
};

/* Aggregate functions.  MAX/MIN: physical_expressions.rs:503-504;
 * SUM/COUNT/AVG are logical-only in the reference
 * (logical_expression.rs:386-423, stub at physical_expressions.rs:505). */
enum { QE_AGG_MAX = 1, QE_AGG_MIN = 2, QE_AGG_SUM = 3, QE_AGG_COUNT = 4, QE_AGG_AVG = 5 };

/* Aggregation mode (no reference counterpart: the reference is single
 * process).  PARTIAL emits mergeable state (one row per local group), FINAL
 * consumes PARTIAL batches from any number of shards. */
enum { QE_AGG_MODE_SINGLE = 0, QE_AGG_MODE_PARTIAL = 1, QE_AGG_MODE_FINAL = 2 };

/* Context flags. */
enum {
  /* Reproduce physical_expressions.rs:264-268: for Sub/Div/Mod with a scalar
   * left operand and an array right operand the reference computes
   * `array OP scalar`.  On by default (drop-in parity). */
  QE_FLAG_REFERENCE_SCALAR_ARRAY_SWAP = 1,
  /* Always run the general interpreter kernels, never a kernel that was built
   * ahead of time for the exact shape of a launch.  Results are the same
   * either way; the flag exists for testing and for timing the two paths. */
  QE_FLAG_NO_SPECIALIZED_KERNELS = 2,
  /* Kernels for launch shapes that were not built ahead of time are compiled at
   * run time.  By default that happens on a background thread, and only for
   * batches large enough to matter, while the general kernels serve the shape;
   * with this flag the first launch of a shape compiles and waits, whatever the
   * batch size (tests, prepared workloads).  The environment variable QE_JIT =
   * off | async | sync sets the default. */
  QE_FLAG_JIT_SYNC = 4,
  /* Aggregates whose group table is small (at most 65536 slots, no utf8 key)
   * hand back their output without waiting for the device to report how many
   * groups there are: the batch has one row per table slot, the rows past the
   * number of groups are all zero (a COUNT(*) column tells real rows apart:
   * count > 0), and nothing in the call synchronises with the device, so a
   * caller can queue step after step - partial aggregate, exchange, final
   * merge - and let the host run ahead.  A table that overflows its
   * `group_capacity_hint` cannot be regrown in this mode: the failure is
   * reported by the next qe_context_synchronize (QE_ERR_AGG_CAPACITY).  No
   * counterpart in the reference. */
  QE_FLAG_DEFERRED_GROUP_COUNT = 8,
  /* Float64 SUM and AVG give the same bits for the same batches whatever order
   * the device meets the rows in (by default the sums are folded with atomic
   * adds in an order that changes from run to run; the reference folds them
   * row by row in input order, physical_plan/mod.rs:340-372).  Every
   * value is turned into a 96-bit fixed-point integer below a power of two that
   * bounds the batch's values (found by an extra pass over the input columns),
   * the integers are summed exactly, and one rounding happens at the end: the
   * result is within rows * 2^-87 * max|value| of the exact sum.  Costs the
   * extra pass, four integer accumulators per sum and a wait per batch; at most
   * 2^31 rows per aggregate. */
  QE_FLAG_DETERMINISTIC_SUM = 16,
  /* A Selection (with the Projection fused onto it) hands its output batch back
   * without waiting for the device to report how many rows were selected: the
   * kernels are queued, the batch's buffers are sized for every input row (a
   * string column's bytes by its source column's), and the row count travels
   * to pinned host memory behind them.  The batch takes its final shape - row
   * count, column lengths, string byte totals - the first time anything asks
   * for it: qe_batch_num_rows / qe_batch_column / qe_batch_to_host, or any call
   * that consumes the batch, wait for that one copy (not for the stream) if it
   * has not arrived.  qe_iter_next then never synchronises, so the host can
   * queue batch after batch the way the reference's SelectionIterator::next
   * streams them (physical_plan/mod.rs:205-229); device-side consumers on the
   * context's stream need no wait at all.  A look-back that timed out is
   * reported (QE_ERR_INTERNAL) by that first use instead of by qe_iter_next.
   * Selections that need the count on the host to go on - a string literal as
   * an output column, more outputs than one launch takes - wait as before. */
  QE_FLAG_DEFERRED_ROW_COUNT = 32,
  /* Group keys that do not fit the 64-bit packed key (a string, a string next to
   * other keys, more than 64 bits of integers) are identified by a 64-bit hash
   * of the tuple.  By default the aggregate then checks, once every batch is
   * in, that every row's key columns equal those of its group's representative
   * row, and fails with QE_ERR_INTERNAL if two distinct keys were merged.  With
   * this flag the check is left out and such groups are identified by their
   * hash alone, as the reference identifies every group
   * (physical_plan/mod.rs:310-358). */
  QE_FLAG_HASH_ONLY_GROUP_IDENTITY = 64,
  /* Testing only: the tuple hash above is taken modulo 5, so that distinct keys
   * do collide and the check has something to find. */
  QE_FLAG_DEBUG_WEAK_KEY_HASH = 128
};

typedef struct qe_context qe_context;
typedef struct qe_batch qe_batch;
typedef struct qe_expr qe_expr;
typedef struct qe_agg_expr qe_agg_expr;
typedef struct qe_plan qe_plan;
typedef struct qe_iter qe_iter;
typedef struct qe_agg_state qe_agg_state;

/* One Arrow array, described by raw buffers.  Buffer order follows the Arrow
 * C data interface (validity, offsets, values). */
typedef struct qe_column {
  int32_t type;         /* QE_TYPE_* */
  int32_t memory_space; /* QE_MEM_* : where the three pointers live */
  int64_t length;       /* rows */
  int64_t null_count;   /* 0 when validity == NULL */
  const void *validity; /* LSB-first bitmap, 1 = valid; NULL = all valid */
  const void *offsets;  /* UTF8: int32[length+1]; else NULL */
  const void *values;   /* INT32/INT64/FLOAT64: packed values; BOOLEAN: bitmap; UTF8: bytes */
  int64_t values_bytes; /* size of the values buffer in bytes */
} qe_column;

/* ColumnarValue (columnar_value.rs:12-15): Array or Scalar. */
typedef struct qe_value {
  int32_t is_scalar;    /* 1: `scalar_*` fields hold the value; 0: `array` */
  int32_t type;         /* QE_TYPE_* */
  int32_t scalar_valid; /* 0 = null scalar */
  int32_t _pad;
  int64_t scalar_i64;   /* INT32 / INT64 / BOOLEAN (0/1) payload */
  double scalar_f64;    /* FLOAT64 payload */
  const char *scalar_utf8; /* UTF8 payload (owned by the value, NUL terminated) */
  int64_t scalar_utf8_len;
  qe_batch *array;      /* one-column batch holding the array result */
} qe_value;

/* ---------------------------------------------------------------- library */
const char *qe_version(void);
/* Thread-local text of the last failure on this thread ("" if none). */
const char *qe_last_error_message(void);
/* Static name of a status code ("DifferentSizes", ...). */
const char *qe_status_name(qe_status status);
/* Number of usable gfx950 devices (0 when none / no driver). */
int32_t qe_device_count(void);

/* --------------------------------------------------------------- context */
/* `stream` is a hipStream_t to launch on (NULL = a stream the context makes).
 * Fails with QE_ERR_DEVICE when `device` is not a usable GPU: there is no CPU
 * fallback behind this ABI. */
qe_status qe_context_create(int32_t device, void *stream, qe_context **out);
qe_status qe_context_set_flags(qe_context *ctx, uint32_t flags);
/* Waits for everything queued on the context's stream; reports a deferred
 * aggregate (QE_FLAG_DEFERRED_GROUP_COUNT) whose table overflowed since the last call. */
qe_status qe_context_synchronize(qe_context *ctx);
void qe_context_free(qe_context *ctx);

/* ---------------------------------------------------------------- batches */
/* RecordBatch = arrow2 `Chunk<Arc<dyn Array>>` (physical_plan/mod.rs:32).
 * The buffers are BORROWED: the caller keeps them alive until the batch and
 * everything derived zero-copy from it is freed. */
qe_status qe_batch_import(qe_context *ctx, const qe_column *columns, int32_t n_columns, qe_batch **out);
qe_status qe_batch_num_rows(const qe_batch *batch, int64_t *out);
qe_status qe_batch_num_columns(const qe_batch *batch, int32_t *out);
/* Fills `out` with pointers into the batch's buffers (valid until qe_batch_free). */
qe_status qe_batch_column(const qe_batch *batch, int32_t index, qe_column *out);
/* Deep copies across the host link; the result owns its buffers. */
qe_status qe_batch_to_device(qe_context *ctx, const qe_batch *batch, qe_batch **out);
qe_status qe_batch_to_host(qe_context *ctx, const qe_batch *batch, qe_batch **out);
void qe_batch_free(qe_batch *batch);

/* ------------------------------------------------------------ expressions */
/* ColumnExpression { index } (physical_expressions.rs:26-29). */
qe_status qe_expr_column(int32_t index, qe_expr **out);
/* LiteralIntegerExpression::new(i32) (physical_expressions.rs:125-131). */
qe_status qe_expr_lit_i32(int32_t value, qe_expr **out);
/* Int64 literal: extension (the reference's integer literal is i32 only). */
qe_status qe_expr_lit_i64(int64_t value, qe_expr **out);
/* LiteralFloatExpression::new(f64) (physical_expressions.rs:150-156). */
qe_status qe_expr_lit_f64(double value, qe_expr **out);
/* LiteralBoolExpression::new(bool) (physical_expressions.rs:75-81). */
qe_status qe_expr_lit_bool(int32_t value, qe_expr **out);
/* LiteralStringExpression::new(String) (physical_expressions.rs:100-106). */
qe_status qe_expr_lit_utf8(const char *bytes, int64_t len, qe_expr **out);
/* EqExpression::new .. DivExpression::new (physical_expressions.rs:214-222,
 * 327-335).  Consumes `left` and `right` (Box ownership in the reference). */
qe_status qe_expr_binary(int32_t op, qe_expr *left, qe_expr *right, qe_expr **out);
/* Deep copy (the reference's expression structs are Clone/owned). */
qe_status qe_expr_clone(const qe_expr *expr, qe_expr **out);
/* `impl Display` (physical_expressions.rs:64-68,225-229,338-342): writes a NUL
 * terminated string, returns the untruncated length via `needed`. */
qe_status qe_expr_to_string(const qe_expr *expr, char *buf, int64_t buf_len, int64_t *needed);
/* PhysicalExpression::evaluate(&Chunk) -> Result<ColumnarValue, Error>
 * (physical_expressions.rs:22-24).  Borrows expr and batch. */
qe_status qe_expr_evaluate(qe_context *ctx, const qe_expr *expr, const qe_batch *batch, qe_value *out);
void qe_value_release(qe_value *value);
void qe_expr_free(qe_expr *expr);

/* MaxExpression::new / MinExpression::new (physical_expressions.rs:489-493)
 * plus SUM/COUNT/AVG.  Consumes `input`.  COUNT with input == NULL counts rows
 * (COUNT(*)). */
qe_status qe_agg_expr_new(int32_t kind, qe_expr *input, qe_agg_expr **out);
void qe_agg_expr_free(qe_agg_expr *agg);

/* Accumulator protocol (physical_expressions.rs:351-362: create_accumulator,
 * Accumulator::accumulate(&mut self, batch values, mask), final_value(self)) for
 * callers that drive the batches themselves instead of handing over a plan.
 * The state is the device group table, not one scalar per group and
 * aggregate: qe_agg_update folds one device-resident batch in (one kernel
 * launch, no synchronisation), qe_agg_finalize assembles the output batch the
 * way AggregateExec does (physical_plan/mod.rs:375-488); it may be called more
 * than once (running results).  Expression column indices refer to the
 * batches' columns.  qe_agg_state_new consumes the expressions.  The batches'
 * buffers stay borrowed until the state is freed: should the group table
 * outgrow `group_capacity_hint`, finalize folds them in again into a larger
 * one.  `mode` as for qe_plan_aggregate. */
qe_status qe_agg_state_new(qe_context *ctx, qe_expr *const *group_exprs, int32_t n_group_exprs,
                           qe_agg_expr *const *agg_exprs, int32_t n_agg_exprs, int32_t mode,
                           int64_t group_capacity_hint, qe_agg_state **out);
qe_status qe_agg_update(qe_agg_state *state, const qe_batch *batch);
qe_status qe_agg_finalize(qe_agg_state *state, qe_batch **out);
void qe_agg_state_free(qe_agg_state *state);

/* ------------------------------------------------------------------ plans */
/* In-memory scan leaf: stands where ScanExec::new(DataSource, projection,
 * schema) stands (physical_plan/mod.rs:68-110); yields the given batches in
 * order.  The plan shares (does not copy) the batches' buffers. */
qe_status qe_plan_scan_memory(qe_context *ctx, qe_batch *const *batches, int32_t n_batches, qe_plan **out);
/* Parquet scan leaf: ScanExec::new(DataSource::Parquet(ParquetDataSource::new(path)),
 * projection, schema) (physical_plan/mod.rs:68-110, data_source.rs:24-72).  Yields
 * one device-resident batch per row group, holding the columns whose names
 * `projection` lists, in file order (n_projection < 0: every column, the
 * reference's `None`).  The file's bytes are decoded on the GPU; the host reads
 * the footer and the page headers only.  QE_ERR_IO when the file cannot be
 * opened (Error::IoError), QE_ERR_ARROW when it is not a Parquet file or uses a
 * feature this reader lacks (nested schemas, delta encodings, LZO / Brotli),
 * QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED at execution when a scanned column has a
 * type outside QE_TYPE_* (FLOAT columns are widened to QE_TYPE_FLOAT64). */
qe_status qe_plan_scan_parquet(qe_context *ctx, const char *path, const char *const *projection, int32_t n_projection, qe_plan **out);
/* The same scan restricted to row groups [first_row_group, first_row_group +
 * n_row_groups) of the file (clamped to what the file has): the unit by which a
 * file is split over GPUs, one contiguous run of row groups per rank (no
 * counterpart in the reference, which is single process). */
qe_status qe_plan_scan_parquet_row_groups(qe_context *ctx, const char *path, const char *const *projection, int32_t n_projection,
                                          int32_t first_row_group, int32_t n_row_groups, qe_plan **out);
/* ParquetDataSource::schema (data_source.rs:45-47) as JSON, in the format of
 * qe_dataframe_schema, plus "rows" and "row_groups":
 * {"fields":[...],"rows":N,"row_groups":G}.  Needs no GPU. */
qe_status qe_parquet_schema(const char *path, char *buf, int64_t buf_len, int64_t *needed);
/* ProjectionExec::new(input, exprs, schema) (physical_plan/mod.rs:118-130).
 * Consumes `child` and the expressions. */
qe_status qe_plan_projection(qe_plan *child, qe_expr *const *exprs, int32_t n_exprs, qe_plan **out);
/* SelectionExec::new(input, expr, schema) (physical_plan/mod.rs:184-196). */
qe_status qe_plan_selection(qe_plan *child, qe_expr *predicate, qe_plan **out);
/* AggregateExec::new(input, group_exprs, agg_exprs, schema)
 * (physical_plan/mod.rs:258-272).  `mode` is QE_AGG_MODE_*;
 * `group_capacity_hint` (0 = default) sizes the global group table; the table
 * grows and the aggregation is re-run when the hint proves too small.  Group
 * keys: up to four Int32 / Int64 / Boolean expressions packed into 64 bits, or
 * one Utf8 column; tuples beyond that (strings next to other keys, several
 * strings, more than 64 bits, up to twelve expressions) are identified by a
 * 64-bit hash of the tuple - the reference identifies every group by hash
 * (physical_plan/mod.rs:317-358) - and computed keys among them are evaluated
 * into columns first.  Any number of aggregates: lists beyond one kernel's
 * sixteen accumulators are split over several kernels and joined by group. */
qe_status qe_plan_aggregate(qe_plan *child, qe_expr *const *group_exprs, int32_t n_group_exprs,
                            qe_agg_expr *const *agg_exprs, int32_t n_agg_exprs, int32_t mode,
                            int64_t group_capacity_hint, qe_plan **out);
/* `format!("{}", plan)`-style description of the physical plan and of how it
 * was fused into kernels. */
qe_status qe_plan_to_string(const qe_plan *plan, char *buf, int64_t buf_len, int64_t *needed);
/* PhysicalPlan::children (physical_plan/mod.rs:50-57): 0 for a scan, else 1. */
qe_status qe_plan_num_children(const qe_plan *plan, int32_t *out);
void qe_plan_free(qe_plan *plan);

/* PhysicalPlan::execute(self) -> Box<dyn Iterator<Item = Batch>>
 * (physical_plan/mod.rs:58-65).  Consumes the plan. */
qe_status qe_plan_execute(qe_plan *plan, qe_iter **out);
/* Iterator::next: QE_OK with a batch, QE_END when exhausted, or an error.
 * Output batches are device resident and owned by the caller.  Their row
 * count is final on return; their buffers are complete in the order of the
 * context's stream (a projection's values, a selection's string bytes may
 * still be on their way): work queued on that stream afterwards, the next
 * operator and qe_batch_to_host see them whole, a caller that reads them from
 * another stream or maps them to the host calls qe_context_synchronize first. */
qe_status qe_iter_next(qe_iter *iter, qe_batch **out);
void qe_iter_free(qe_iter *iter);

/* ------------------------------------------------ multi-GPU exchange (RCCL) */
/* No counterpart in the reference (a single-threaded pull iterator,
 * physical_plan/mod.rs:58).  One process per GPU: every rank runs the plan over
 * its row range with QE_AGG_MODE_PARTIAL, the partial tables meet in ONE
 * exchange, a QE_AGG_MODE_FINAL aggregate merges what arrives - the merge the
 * reference performs batch by batch (physical_plan/mod.rs:298-374), across
 * devices.  RCCL (librccl.so) is bound when the first of these calls is made;
 * a process that never calls them does not need it. */
typedef struct qe_comm qe_comm;
#define QE_COMM_ID_BYTES 128
/* The communicator's rendezvous id (ncclGetUniqueId): made by ONE rank and
 * handed to every rank by the host program's own means (a file, a socket, MPI,
 * torch.distributed ...). */
qe_status qe_comm_unique_id(uint8_t *id /* [QE_COMM_ID_BYTES] */);
/* Collective over all `world` ranks (ncclCommInitRank) on the context's device
 * and stream; every exchange of this communicator is queued on that stream. */
qe_status qe_comm_init(qe_context *ctx, int32_t rank, int32_t world, const uint8_t *id, qe_comm **out);
/* A communicator with no peers behind it, for running the multi-rank layouts on
 * one GPU and for host programs that move the blocks between ranks by their own
 * means.  Every exchange is laid out exactly as rank `rank` of `world` ranks lays
 * it out - the owner hash and scatter of qe_exchange_by_key run over `world`
 * owners, qe_exchange_all_gather puts this rank's table at rank * padded_rows -
 * and the transfers are device copies on the context's stream:
 *   qe_exchange_all_gather: the other ranks' blocks of the output are all zero
 *     (what ranks with empty tables would have contributed);
 *   qe_exchange_by_key: block p of the output holds what this rank sends to
 *     rank p (over a real communicator block p holds what rank p sent here), so
 *     block q of every rank's output, put together, is what rank q receives.
 * Never binds RCCL.  No counterpart in the reference. */
qe_status qe_comm_init_loopback(qe_context *ctx, int32_t rank, int32_t world, qe_comm **out);
/* RCCL's version as ncclGetVersion reports it, binding the library if this is
 * the first call that needs it; QE_ERR_DEVICE when no RCCL can be loaded.  Lets
 * every rank of a job confirm that it can take part before any of them enters
 * the collective qe_comm_init (a rank that failed there would leave the others
 * waiting). */
qe_status qe_comm_rccl_version(int32_t *version);
/* Any of the outputs may be NULL.  rccl_version as ncclGetVersion reports it
 * (0 for a loopback communicator). */
qe_status qe_comm_info(const qe_comm *comm, int32_t *rank, int32_t *world, int32_t *rccl_version);
void qe_comm_free(qe_comm *comm);
/* Few groups: every rank receives every rank's partial table.  `partial`: the
 * device-resident output of a PARTIAL aggregate (Int32 / Int64 / Float64
 * columns).  `padded_rows` (the same on every rank, at least every rank's row
 * count; a multiple of 32 when a column can be null): the output has
 * world x padded_rows rows, rank r's table at rows [r * padded_rows, ...), rows
 * nobody filled all zero - a COUNT(*) column tells them apart (count > 0), so a
 * Selection on it under the FINAL aggregate merges everything in one launch.
 * One grouped ncclAllGather; nothing is read back to the host, nothing waits. */
qe_status qe_exchange_all_gather(qe_comm *comm, const qe_batch *partial, int64_t padded_rows, qe_batch **out);
/* Many groups: every partial row goes to the rank that owns its key (hash of the
 * first `n_key_columns` columns modulo the ranks), so that each rank merges
 * only its share of the groups and the result is the concatenation over the
 * ranks.  Device-side owner hash and scatter into per-peer blocks of a fixed
 * size (derived from `max_rows_per_rank`, a bound on any rank's row count, the
 * same on every rank), then grouped ncclSend / ncclRecv: W - 1 direct transfers
 * per rank at once.  The output has world x block rows, unfilled ones all zero
 * (as above).  Nothing waits for the device: a block that overflows is reported
 * by the next qe_context_synchronize (QE_ERR_AGG_CAPACITY). */
qe_status qe_exchange_by_key(qe_comm *comm, const qe_batch *partial, int32_t n_key_columns, int64_t max_rows_per_rank, qe_batch **out);

/* -------------------------------------------- selection without allocation */
/* No counterpart in the reference.  For callers that cannot allocate device
 * memory while they run - tight serving loops, harnesses that forbid hipMalloc,
 * processes whose device memory is spoken for: Selection(predicate) ->
 * Projection(exprs) over one device-resident batch, the fused launch
 * qe_plan_selection / qe_plan_projection / qe_iter_next make, with every output
 * written into buffers the CALLER provides and the kernels' scratch (ticket
 * counter, one look-back status word per tile) in a fixed array inside the
 * library instead of a buffer from hipMalloc.  The call allocates no device
 * memory, takes nothing from the context's pool and compiles nothing.
 *   outs[i]: type = the expression's type (INT32 / INT64 / FLOAT64 / BOOLEAN),
 *            memory_space = QE_MEM_DEVICE, values = the buffer, length = its
 *            capacity in rows (at least the batch's rows: the number selected
 *            is not known beforehand), validity = a bitmap buffer when the
 *            expression can be null.  Bitmap buffers (boolean values, validity)
 *            span whole 4-byte words: ((length + 31) / 32) * 4 bytes.
 *   rows_out: the number of rows selected; rows keep their input order.
 * Limits: at most QE_NOALLOC_MAX_ROWS rows per batch and 12 expressions that fit
 * one kernel launch; no Utf8 / Timestamp columns or literals (their gathers and
 * literal copies take buffers) - anything else is refused with an error, never
 * served by allocating.  Every call on one GPU shares that GPU's array (1 MiB,
 * held until the process ends).  Calls from several threads of one process take
 * turns on it (a lock per device, held until the call's result has arrived);
 * calls from different PROCESSES on one GPU each have their own copy of the
 * array.  The call copies the row count to the host and waits for it
 * (hipMemcpyAsync + hipStreamSynchronize on the context's stream), so it must
 * NOT be made on a stream that is being captured into a HIP graph: both calls
 * are illegal there and would invalidate the capture. */
#define QE_NOALLOC_MAX_ROWS (1ll << 29)
qe_status qe_selection_noalloc(qe_context *ctx, const qe_batch *batch, const qe_expr *predicate, const qe_expr *const *exprs,
                               int32_t n_exprs, const qe_column *outs, int64_t *rows_out);

/* ------------------------------------------------- synthetic data (bench) */
/* Counter-based generator (splitmix64 of row index and seed): fills
 * device buffers of `n` rows starting at global row `row0`, so any shard is
 * reproducible on host and device.  kind: 0 = f64 uniform [lo,hi),
 * 1 = i64 uniform {lo..hi-1} (lo/hi rounded), 2 = i64 zipf(s=hi) over
 * {0..lo-1}, 3 = i64 skewed passenger-count style key over {0..9}. */
qe_status qe_generate_column(qe_context *ctx, int32_t kind, uint64_t seed, int64_t row0, int64_t n,
                             double lo, double hi, void *device_out);

/* Device timing of the kernels the last qe_iter_next / qe_expr_evaluate call
 * launched on this context: name, launches and total milliseconds measured
 * with HIP events on the context's stream (enabled by qe_context_profile:
 * 1 = every launch, 2 = only the main kernels of launches over at least 2^20
 * rows, leaving the small helper launches of a step free of event records). */
qe_status qe_context_profile(qe_context *ctx, int32_t enable);
qe_status qe_context_profile_read(qe_context *ctx, char *buf, int64_t buf_len, int64_t *needed);

/* ---------------------------------------------------------------------------
 * The callers of the physical path: logical expressions, the DataFrame builder
 * and the planner that lowers a logical plan onto the operators above.
 * ------------------------------------------------------------------------- */
typedef struct qe_logical_expr qe_logical_expr;
typedef struct qe_dataframe qe_dataframe;

/* prelude.rs:7-25: col, lit_string, lit_int (i32), lit_float, lit_bool.
 * qe_lit_int64 is an extension. */
qe_status qe_col(const char *name, qe_logical_expr **out);
qe_status qe_lit_string(const char *value, qe_logical_expr **out);
qe_status qe_lit_int(int32_t value, qe_logical_expr **out);
qe_status qe_lit_int64(int64_t value, qe_logical_expr **out);
qe_status qe_lit_float(double value, qe_logical_expr **out);
qe_status qe_lit_bool(int32_t value, qe_logical_expr **out);
/* LogicalExpression::{Eq..Or, Add..Mod} (logical_expression.rs:284-342) and the
 * LogicalExpressionMethods (:425-472); op is a QE_OP_*.  Consumes both operands. */
qe_status qe_logical_binary(int32_t op, qe_logical_expr *left, qe_logical_expr *right, qe_logical_expr **out);
/* LogicalExpression::{Max, Min, Sum, Count, Avg} (logical_expression.rs:386-423,
 * prelude.rs:27-33); kind is a QE_AGG_*.  Consumes expr. */
qe_status qe_logical_aggregate(int32_t kind, qe_logical_expr *expr, qe_logical_expr **out);
/* impl Display for LogicalExpression (logical_expression.rs:68-96). */
qe_status qe_logical_expr_to_string(const qe_logical_expr *expr, char *buf, int64_t buf_len, int64_t *needed);
void qe_logical_expr_free(qe_logical_expr *expr);

/* DataFrame::parquet(path) (dataframe.rs:38-41): a scan of a Parquet file (see
 * qe_plan_scan_parquet).  Projection push-down (qe_dataframe_optimize) narrows
 * what is read from the file. */
qe_status qe_dataframe_parquet(qe_context *ctx, const char *path, qe_dataframe **out);
/* DataFrame::parquet (dataframe.rs:38-41) with the file already decoded: a scan
 * over device-resident batches whose columns carry the given names.  `label`
 * stands where the path does in the plan's Display. */
qe_status qe_dataframe_from_batches(qe_context *ctx, const char *label, const char *const *column_names, int32_t n_columns,
                                    qe_batch *const *batches, int32_t n_batches, qe_dataframe **out);
/* DataFrameTrait::{project, filter, aggregate} (dataframe.rs:44-69).  Each consumes
 * `df` (also on failure) and the expressions.  An unknown column name fails with
 * QE_ERR_NO_FIELD_IN_LOGICAL_PLAN (the reference panics while deriving the schema). */
qe_status qe_dataframe_project(qe_dataframe *df, qe_logical_expr *const *exprs, int32_t n_exprs, qe_dataframe **out);
qe_status qe_dataframe_filter(qe_dataframe *df, qe_logical_expr *expr, qe_dataframe **out);
qe_status qe_dataframe_aggregate(qe_dataframe *df, qe_logical_expr *const *group_by, int32_t n_group_by,
                                 qe_logical_expr *const *aggregate_expr, int32_t n_aggregate_expr, qe_dataframe **out);
/* LogicalPlan::optimize = projection push-down (logical_plan/optimizer.rs:7-58).  Consumes df. */
qe_status qe_dataframe_optimize(qe_dataframe *df, qe_dataframe **out);
/* DataFrameTrait::schema (dataframe.rs:71-73) as JSON:
 * [{"name":..,"data_type":"Int32"|..,"type":QE_TYPE_*,"is_nullable":bool}, ...] */
qe_status qe_dataframe_schema(const qe_dataframe *df, char *buf, int64_t buf_len, int64_t *needed);
/* format_logical_plan (logical_plan/mod.rs:48-58). */
qe_status qe_dataframe_logical_plan(const qe_dataframe *df, char *buf, int64_t buf_len, int64_t *needed);
/* LogicalPlan::to_physical_plan (query_planner.rs:104-166).  Consumes df. */
qe_status qe_dataframe_to_physical_plan(qe_dataframe *df, qe_plan **out);
/* DataFrame::execute (dataframe.rs:80-83).  Consumes df. */
qe_status qe_dataframe_execute(qe_dataframe *df, qe_iter **out);
void qe_dataframe_free(qe_dataframe *df);

/* Introspection (no counterpart in the reference; needs no GPU): lowers
 *   Scan(schema) -> [Selection predicate] -> [Projection exprs] -> [Aggregate group_exprs, agg_exprs]
 * exactly as qe_plan_execute would and writes a JSON description of every
 * kernel launch (interpreter code, column-cache layout, sink descriptors and
 * the launch signature that selects a shape-specialised kernel) to `buf`.
 * Expressions are borrowed, not consumed.  The build uses it to generate the
 * shape-specialised kernels; it is also the way to see what a plan compiles to. */
qe_status qe_lowering_describe(const int32_t *col_types, const uint8_t *col_nullable, int32_t n_cols, uint32_t flags,
                               const qe_expr *predicate, const qe_expr *const *exprs, int32_t n_exprs,
                               const qe_expr *const *group_exprs, int32_t n_group_exprs,
                               const qe_agg_expr *const *agg_exprs, int32_t n_agg_exprs, int32_t mode,
                               char *buf, int64_t buf_len, int64_t *needed);

/* Same lowering as qe_lowering_describe; compiles the shape-specialised kernel
 * of every launch with the HIP run-time compiler (no GPU needed) and reports
 * how many kernels were built and their total code size.  Launches whose
 * expressions need the general kernels (division, utf8, casts) are skipped.
 * No counterpart in the reference. */
qe_status qe_lowering_compile(const int32_t *col_types, const uint8_t *col_nullable, int32_t n_cols, uint32_t flags,
                              const qe_expr *predicate, const qe_expr *const *exprs, int32_t n_exprs,
                              const qe_expr *const *group_exprs, int32_t n_group_exprs,
                              const qe_agg_expr *const *agg_exprs, int32_t n_agg_exprs, int32_t mode,
                              int32_t *n_kernels, int64_t *code_bytes);

#ifdef __cplusplus
}
#endif
#endif /* QE_GPU_H */
