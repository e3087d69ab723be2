// Physical operators (Scan / Projection / Selection / Aggregate) as pull
// iterators over device-resident batches, the fusion of operator chains into
// single kernel launches, and the extern "C" boundary.
//
// Reference shape being mirrored: PhysicalPlan::execute(self) returns a boxed
// iterator of batches (physical_plan/mod.rs:58-65); each *Exec pops its single
// child (:167-170, :240-243, :293-296); Aggregate is eager and yields exactly
// one batch (:274-283).
#include <algorithm>
#include <climits>
#include <cstring>
#include <functional>
#include <mutex>
#include <sstream>
#include <unordered_map>

#include "qe_core.h"
#include "qe_jit.h"
#include "qe_plan.h"

namespace qe {

// ---------------------------------------------------------------- iterators
struct Iter {
  virtual ~Iter() {}
  virtual bool next(Batch &out) = 0;  // false = exhausted
  virtual void reset() = 0;           // start over (used when a group table has to grow)
  virtual int64_t batches_ahead() const { return 0; }  // how many batches next() will hand out in all; 0 = not known
};

struct ScanIter : Iter {
  Context *ctx = nullptr;
  std::vector<Batch> batches;
  std::shared_ptr<ParquetFile> parquet;  // when set: one batch per row group, decoded on demand
  std::vector<int> parquet_columns;
  size_t first = 0, end = 0;  // row groups of the file this scan covers
  size_t cur = 0;
  ParquetReadAhead ahead;     // the next chunks' bytes, already on their way into pinned memory
  bool next(Batch &out) override {
    if (parquet) {
      // ScanIterator::next (physical_plan/mod.rs:99-109) over the file reader's row groups
      if (first + cur >= end) return false;
      out = parquet_read_row_group(ctx, *parquet, (int)(first + cur++), parquet_columns, &ahead, (int)end);
      return true;
    }
    if (cur >= batches.size()) return false;
    out = batches[cur++];
    return true;
  }
  void reset() override { cur = 0; }
  int64_t batches_ahead() const override { return parquet ? (int64_t)(end - first) : (int64_t)batches.size(); }
};

// A run of Projection / Selection operators over one source, expressed over the
// source's columns so that the whole run is one kernel per batch.
struct Stream {
  std::unique_ptr<Iter> src;
  bool identity = true;          // outputs == source columns
  std::vector<ExprPtr> outputs;  // valid when !identity
  ExprPtr pred;                  // conjunction of every Selection below
};

static std::vector<ExprPtr> stream_outputs(const Stream &s, int n_source_cols) {
  if (!s.identity) return s.outputs;
  std::vector<ExprPtr> o;
  for (int i = 0; i < n_source_cols; ++i) o.push_back(make_column(i));
  return o;
}

// `cleared_validity`: a zeroed bitmap of bitmap_bytes(capacity_rows) bytes to use instead of a block of the column's own
// (several columns' bitmaps in one block take one fill instead of a small launch each).
static Column make_fixed_column(Context *ctx, int type, int64_t capacity_rows, bool with_validity, bool zero_values, BufferPtr cleared_validity = nullptr) {
  Column c;
  c.type = type;
  c.space = QE_MEM_DEVICE;
  size_t vb = type == QE_TYPE_BOOLEAN ? bitmap_bytes(capacity_rows) : (size_t)capacity_rows * type_width(type) + 16;
  c.values = device_buffer(ctx, vb);
  c.values_bytes = (int64_t)vb;
  if (type == QE_TYPE_BOOLEAN || zero_values) QE_HIP(hipMemsetAsync(c.values->ptr, 0, vb, ctx->stream));
  if (with_validity) {
    if (cleared_validity) {
      c.validity = cleared_validity;
    } else {
      size_t bb = bitmap_bytes(capacity_rows);
      c.validity = device_buffer(ctx, bb);
      QE_HIP(hipMemsetAsync(c.validity->ptr, 0, bb, ctx->stream));
    }
    c.null_count = -1;
  }
  return c;
}

static void finish_fixed_column(Column &c, int64_t rows) {
  c.length = rows;
  if (c.type != QE_TYPE_BOOLEAN) c.values_bytes = rows * type_width(c.type);
  else c.values_bytes = (rows + 7) / 8;
}

static uint8_t vm_type_of(int t) { return (uint8_t)t; }  // QE_TYPE_* and VmType share values

static void ensure_filter_scratch(Context *ctx, int64_t n_rows) {
  size_t need = filter_scratch_bytes(n_rows);
  if (need > ctx->filter_scratch_cap) {
    ctx->sync();
    if (ctx->filter_scratch) QE_HIP(hipFree(ctx->filter_scratch));
    size_t cap = need * 2;
    QE_HIP(hipMalloc(&ctx->filter_scratch, cap));
    ctx->filter_scratch_cap = cap;
  }
}

static void ensure_counters(Context *ctx) {
  if (!ctx->counters) {
    QE_HIP(hipMalloc((void **)&ctx->counters, 64 * sizeof(unsigned long long)));
    QE_HIP(hipHostMalloc((void **)&ctx->host_counters, 128 * sizeof(unsigned long long), hipHostMallocDefault));
  }
}

static Column utf8_broadcast_column(Context *ctx, const std::string &s, int64_t rows) {
  if ((double)rows * (double)s.size() > 2147483647.0)
    fail(QE_ERR_ARROW, "utf8 broadcast exceeds int32 offsets");
  Column c;
  c.type = QE_TYPE_UTF8;
  c.space = QE_MEM_DEVICE;
  c.length = rows;
  c.offsets = device_buffer(ctx, (size_t)(rows + 1) * 4);
  c.values_bytes = rows * (int64_t)s.size();
  c.values = device_buffer(ctx, (size_t)c.values_bytes + 16);
  BufferPtr lit = device_buffer(ctx, s.size() + 1);
  QE_HIP(hipMemcpyAsync(lit->ptr, s.data(), s.size(), hipMemcpyHostToDevice, ctx->stream));
  ctx->prof_begin("qe_utf8_broadcast_kernel");
  QE_HIP(launch_utf8_broadcast((const uint8_t *)lit->ptr, (int32_t)s.size(), rows, (int32_t *)c.offsets->ptr,
                               (uint8_t *)c.values->ptr, ctx->stream));
  ctx->prof_end();
  ctx->sync();  // `lit` is released on return
  return c;
}

// Rows idx[0..m) of several utf8 columns.  The offsets of a group of (up to sixteen) columns are computed first, their byte
// totals come back with one synchronisation, then the bytes are copied: the host waits once per group, not once per column.
// With `pending` the host does not wait at all: `m` is then only a bound (the batch's rows), the kernels read the
// number of rows from `m_dev` (the word the filter kernel left it in), every output is sized for the bound - the bytes by
// the source column's - and the byte totals travel to the pending batch's pinned words (from block `first_slot` on).
// A block the caller has cleared with one fill and that holds what a deferred selection's kernels need zeroed: every
// string column's control block (in column order, the filter's own scratch right behind them, so that one copy takes
// all result words to the host), their status arrays and their validity bitmaps.  Without it the gather takes and
// clears blocks of its own - a fill and a copy are a small launch each, one after the other on the stream, and a
// selection of ten string columns made six of them a batch.
struct GatherArena {
  BufferPtr block;
  size_t control_at = 0, status_at = 0, bitmaps_at = 0;
};

static std::vector<Column> utf8_gather_columns(Context *ctx, const std::vector<const Column *> &srcs, const int32_t *idx, int64_t m,
                                               const unsigned long long *m_dev = nullptr, PendingRows *pending = nullptr, int first_slot = 0,
                                               const GatherArena *arena = nullptr) {
  // a column's control block: ticket, byte total, look-back error, one word of padding - the head of a FilterScratch,
  // of which the offsets kernel touches nothing else
  constexpr size_t GROUP = UTF8_GATHER_MAX_COLS, FIRST_WORD = 64, CONTROL_WORDS = 4, CONTROL_BYTES = CONTROL_WORDS * sizeof(unsigned long long);
  static_assert(offsetof(FilterScratch, error) + sizeof(unsigned long long) <= CONTROL_BYTES, "control block");
  std::vector<Column> out(srcs.size());
  ensure_counters(ctx);
  const size_t status_bytes = m > 0 ? (utf8_gather_offsets_status_bytes(m) + 63) / 64 * 64 : 0;
  for (size_t g0 = 0; g0 < srcs.size(); g0 += GROUP) {
    const size_t g1 = std::min(srcs.size(), g0 + GROUP), cols = g1 - g0;
    BufferPtr scratch;  // the group's control blocks, then its status arrays
    if (m > 0 && !arena) {
      const size_t sb = (cols * CONTROL_BYTES + 63) / 64 * 64 + cols * status_bytes + sizeof(FilterScratch);
      scratch = device_buffer(ctx, sb);
      QE_HIP(hipMemsetAsync(scratch->ptr, 0, sb, ctx->stream));
    }
    Utf8GatherCols launch{};
    launch.n = (int)cols;
    // the group's validity bitmaps lie in one block, cleared by one fill (a fill per column is a small launch each,
    // one after the other on the stream)
    const size_t bb = bitmap_bytes(m);
    size_t nullable = 0;
    for (size_t j = g0; j < g1; ++j) nullable += srcs[j]->validity ? 1 : 0;
    BufferPtr bitmaps;
    size_t bitmaps_at = 0;
    if (nullable && arena) {
      bitmaps = arena->block;
      bitmaps_at = arena->bitmaps_at;
      for (size_t j = 0; j < g0; ++j) bitmaps_at += srcs[j]->validity ? bb : 0;
    } else if (nullable) {
      bitmaps = device_buffer(ctx, nullable * bb);
      QE_HIP(hipMemsetAsync(bitmaps->ptr, 0, nullable * bb, ctx->stream));
    }
    size_t next_bitmap = 0;
    // Offsets and bytes in one kernel when the output bytes' buffers can be had before their sizes are known: always for
    // a selection that does not wait (they are sized by the source's bytes), and for one that does when it keeps a
    // quarter of the rows or more - the buffers are then at most about four times what they need to be, and the pass
    // saves a kernel, a second reading of row ids and offsets, and the wait between the two.
    bool fuse = utf8_gather_fused_available() && m > 0;
    if (fuse && !pending)
      for (size_t j = g0; j < g1; ++j) fuse = fuse && m * 4 >= srcs[j]->length;
    for (size_t j = g0; j < g1; ++j) {
      const Column &src = *srcs[j];
      Column &c = out[j];
      c.type = QE_TYPE_UTF8;
      c.space = QE_MEM_DEVICE;
      c.length = m;
      c.offsets = device_buffer(ctx, (size_t)(m + 1) * 4 + 16);
      if (src.validity) {
        c.validity = sub_buffer(bitmaps, bitmaps_at + next_bitmap++ * bb, bb);
        c.null_count = -1;
      }
      if (m <= 0) {
        QE_HIP(hipMemsetAsync(c.offsets->ptr, 0, 4, ctx->stream));
        continue;
      }
      Utf8GatherCol &g = launch.col[j - g0];
      g.src_offsets = src.offsets_ptr();
      g.src_validity = src.validity_ptr();
      g.src_bytes = (const uint8_t *)src.values_ptr();
      // (the source's bytes as its buffer bounds them: a 16-byte fetch of a short row must end inside)
      g.src_total = src.values ? (int64_t)std::max<size_t>(src.values->bytes, (size_t)std::max<int64_t>(src.values_bytes, 0)) : 0;
      g.out_offsets = (int32_t *)c.offsets->ptr;
      g.out_validity = c.validity ? (uint32_t *)c.validity->ptr : nullptr;
      if (arena) {
        g.control = (char *)arena->block->ptr + arena->control_at + j * CONTROL_BYTES;
        g.status = (char *)arena->block->ptr + arena->status_at + j * status_bytes;
      } else {
        g.control = (char *)scratch->ptr + (j - g0) * CONTROL_BYTES;
        g.status = (char *)scratch->ptr + (cols * CONTROL_BYTES + 63) / 64 * 64 + (j - g0) * status_bytes;
      }
      if (pending || fuse) {
        // no selection keeps more bytes than the column has
        c.values_bytes = std::max<int64_t>(src.values_bytes, 0);
        c.values = device_buffer(ctx, (size_t)c.values_bytes + 16);
        g.out_bytes = (uint8_t *)c.values->ptr;
      }
    }
    if (m <= 0) continue;
    if (fuse) {
      // the output bytes' buffers exist already: offsets and bytes in one pass (the tile that places its rows copies them)
      ctx->prof_begin("qe_utf8_gather_fused_kernel");
      QE_HIP(launch_utf8_gather_fused(launch, idx, m, m_dev, ctx->dev.cus, ctx->stream));
      ctx->prof_end();
      if (pending) {
        // (with an arena the caller copies every column's words, and the filter's, in one go)
        if (!arena) QE_HIP(hipMemcpyAsync(pending->host + CONTROL_WORDS * (first_slot + g0), scratch->ptr, cols * CONTROL_BYTES, hipMemcpyDeviceToHost, ctx->stream));
        continue;
      }
      QE_HIP(hipMemcpyAsync(ctx->host_counters + FIRST_WORD, scratch->ptr, cols * CONTROL_BYTES, hipMemcpyDeviceToHost, ctx->stream));
      ctx->sync();
      for (size_t j = g0; j < g1; ++j) {
        const unsigned long long *w = ctx->host_counters + FIRST_WORD + CONTROL_WORDS * (j - g0);
        if (w[2] != 0) fail(QE_ERR_INTERNAL, "string gather look-back timed out");
        if (w[1] > 2147483647ull) fail(QE_ERR_INVALID_ARGUMENT, "the selected strings of one column exceed 2^31 - 1 bytes: scan smaller batches");
        out[j].values_bytes = (int64_t)w[1];
      }
      continue;
    }
    // lengths by row id, scanned into offsets, in one pass over all columns of the group; offs[m] = a column's byte total
    ctx->prof_begin("qe_utf8_gather_offsets_kernel");
    QE_HIP(launch_utf8_gather_offsets(launch, idx, m, m_dev, ctx->dev.cus, ctx->stream));
    ctx->prof_end();
    if (pending) {
      if (!arena) QE_HIP(hipMemcpyAsync(pending->host + CONTROL_WORDS * (first_slot + g0), scratch->ptr, cols * CONTROL_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    } else {
      QE_HIP(hipMemcpyAsync(ctx->host_counters + FIRST_WORD, scratch->ptr, cols * CONTROL_BYTES, hipMemcpyDeviceToHost, ctx->stream));
      ctx->sync();  // (the scratch block is done with here too)
      for (size_t j = g0; j < g1; ++j) {
        Column &c = out[j];
        const unsigned long long *w = ctx->host_counters + FIRST_WORD + CONTROL_WORDS * (j - g0);
        if (w[2] != 0) fail(QE_ERR_INTERNAL, "string gather look-back timed out");
        if (w[1] > 2147483647ull) fail(QE_ERR_INVALID_ARGUMENT, "the selected strings of one column exceed 2^31 - 1 bytes: scan smaller batches");
        const int32_t total = (int32_t)w[1];
        c.values_bytes = total;
        c.values = device_buffer(ctx, (size_t)total + 16);
        launch.col[j - g0].out_bytes = (uint8_t *)c.values->ptr;
      }
    }
    ctx->prof_begin("qe_utf8_gather_bytes_kernel");
    QE_HIP(launch_utf8_gather_bytes(launch, idx, m, m_dev, ctx->stream));
    ctx->prof_end();
  }
  return out;
}

// A utf8 column of one scanned batch and the global row number of its first row.
struct Utf8KeySource {
  Column col;
  int64_t first_row;
};

// Strings of the rows `ids` (global row numbers over the batches of `sources`), as a new utf8 column.
static Column utf8_gather_rows(Context *ctx, const std::vector<Utf8KeySource> &sources, const int64_t *ids, int64_t m) {
  std::vector<Utf8Source> table;
  for (auto &s : sources) table.push_back(Utf8Source{s.col.offsets_ptr(), (const uint8_t *)s.col.values_ptr(), s.first_row});
  BufferPtr dev_table = device_buffer(ctx, table.size() * sizeof(Utf8Source) + 16);
  QE_HIP(hipMemcpyAsync(dev_table->ptr, table.data(), table.size() * sizeof(Utf8Source), hipMemcpyHostToDevice, ctx->stream));
  Column c;
  c.type = QE_TYPE_UTF8;
  c.space = QE_MEM_DEVICE;
  c.length = m;
  c.offsets = device_buffer(ctx, (size_t)(m + 1) * 4 + 16);
  int32_t *offs = (int32_t *)c.offsets->ptr;
  QE_HIP(hipMemsetAsync(offs + m, 0, 4, ctx->stream));
  ctx->prof_begin("qe_utf8_gather_rows_kernel");
  QE_HIP(launch_utf8_gather_rows((const Utf8Source *)dev_table->ptr, (int)table.size(), ids, m, offs, nullptr, nullptr, ctx->stream));
  ctx->prof_end();
  BufferPtr scan_tmp = device_buffer(ctx, scan_scratch_elems(m + 1) * 4 + 64);
  QE_HIP(launch_exclusive_scan_i32(offs, m + 1, (int32_t *)scan_tmp->ptr, ctx->stream));
  ensure_counters(ctx);
  QE_HIP(hipMemcpyAsync(ctx->host_counters + 8, offs + m, 4, hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync();  // also: `table` (host) has been read
  int32_t total = *(int32_t *)(ctx->host_counters + 8);
  c.values_bytes = total;
  c.values = device_buffer(ctx, (size_t)total + 16);
  ctx->prof_begin("qe_utf8_gather_rows_kernel");
  QE_HIP(launch_utf8_gather_rows((const Utf8Source *)dev_table->ptr, (int)table.size(), ids, m, nullptr, offs, (uint8_t *)c.values->ptr,
                                 ctx->stream));
  ctx->prof_end();
  ctx->sync();  // `dev_table` goes back to the pool on return
  return c;
}

// The same for a fixed-width key column (Int32 / Int64 / Boolean; keys of surviving groups are never null).
static Column fixed_gather_rows(Context *ctx, int type, const std::vector<Utf8KeySource> &sources, const int64_t *ids, int64_t m) {
  std::vector<FixedSource> table;
  for (auto &s : sources) table.push_back(FixedSource{s.col.values_ptr(), s.first_row});
  BufferPtr dev_table = device_buffer(ctx, table.size() * sizeof(FixedSource) + 16);
  QE_HIP(hipMemcpyAsync(dev_table->ptr, table.data(), table.size() * sizeof(FixedSource), hipMemcpyHostToDevice, ctx->stream));
  Column c = make_fixed_column(ctx, type, m, false, false);
  finish_fixed_column(c, m);
  ctx->prof_begin("qe_gather_rows_fixed_kernel");
  QE_HIP(launch_gather_rows_fixed((const FixedSource *)dev_table->ptr, (int)table.size(), ids, m, type_width(type), c.values->ptr, ctx->stream));
  ctx->prof_end();
  ctx->sync();  // `table` (host) has been read; `dev_table` goes back to the pool on return
  return c;
}

// ---------------------------------------------------------------- lowering
// Expression trees -> kernel arguments, apart from output buffers: those come from `sink`, so
// that the same code serves execution (buffers from the pool) and qe_lowering_describe (none).
using OutSink = std::function<void(int index, bool nullable, VmOut &o)>;

// One projection launch for outs[vm_list[k...]]; returns the position after the last one placed.
static size_t lower_project(Compiler &comp, const Schema &schema, const std::vector<ExprPtr> &outs, const std::vector<int> &types,
                            const std::vector<int> &vm_list, size_t k, int64_t n, ProjectArgs &args, const OutSink &sink) {
  int cnt = 0;
  while (k < vm_list.size() && cnt < VM_MAX_OUTS && (cnt == 0 || comp.fits(outs[vm_list[k]]))) {
    int i = vm_list[k];
    VmOut &o = args.outs[cnt];
    o.pc = comp.add(outs[i]);
    o.type = vm_type_of(types[i]);
    sink(i, expr_nullable(outs[i], schema), o);
    ++cnt;
    ++k;
  }
  args.prog = comp.prog;
  args.n_outs = cnt;
  args.n_rows = n;
  return k;
}

// One filter launch: the predicate, optionally the row-id output (utf8 gathers), then as many
// of outs[vm_list[k...]] as fit.
static size_t lower_filter(Compiler &comp, const Schema &schema, const ExprPtr &pred, const std::vector<ExprPtr> &outs,
                           const std::vector<int> &types, const std::vector<int> &vm_list, size_t k, int64_t n, void *rowid_values,
                           FilterArgs &args, const OutSink &sink) {
  args.pred_pc = comp.add(pred);
  args.pred_cache_mask = comp.last_cache_mask;
  int cnt = 0;
  if (rowid_values) {
    VmOut &o = args.outs[cnt++];
    o.pc = comp.add_rowid();
    o.type = VT_I32;  // (a batch with string columns has fewer than 2^31 rows: four bytes a row id, read by every gather)
    o.values = rowid_values;
    o.validity = nullptr;
  }
  while (k < vm_list.size() && cnt < VM_MAX_OUTS && comp.fits(outs[vm_list[k]])) {
    int i = vm_list[k];
    VmOut &o = args.outs[cnt];
    o.pc = comp.add(outs[i]);
    args.out_cache_mask |= comp.last_cache_mask;
    o.type = vm_type_of(types[i]);
    sink(i, expr_nullable(outs[i], schema), o);
    ++cnt;
    ++k;
  }
  args.prog = comp.prog;
  args.n_outs = cnt;
  args.n_rows = n;
  args.n_tiles = filter_num_tiles(n);
  return k;
}

// Evaluates `outputs` (expressions over `in`'s columns) for the rows of `in`
// that satisfy `pred` (all rows when null).  ProjectionIterator::next
// (physical_plan/mod.rs:138-156) and SelectionIterator::next (:205-229) fused.
static bool mentions_column_of_type(const ExprPtr &e, const Batch &in, int type) {
  if (!e) return false;
  if (e->kind == EK_COLUMN) return e->index >= 0 && e->index < (int)in.cols.size() && in.cols[e->index].type == type;
  return mentions_column_of_type(e->l, in, type) || mentions_column_of_type(e->r, in, type);
}

static Batch run_pipeline_typed(Context *ctx, const Batch &in, const std::vector<ExprPtr> &outputs_raw, const ExprPtr &pred_raw);

// ---- expressions larger than one kernel's program
// A kernel evaluates its expressions on a stack of VM_MAX_DEPTH slots, out of a program of VM_MAX_INSN
// instructions, VM_MAX_LITS literals and VM_MAX_COLS columns.  The reference evaluates a tree of any
// size, one array per node (physical_expressions.rs:243-268).  A tree that exceeds a kernel is cut:
// subtrees are evaluated by a projection launch of their own into temporary columns appended to the
// batch, and the tree reads those columns instead.  The limits per expression leave room for a
// predicate and one output in the same program.
constexpr int FIT_INSNS = 40, FIT_LITS = 10, FIT_COLS = 8;

struct ExprFitter {
  Context *ctx;
  Batch &in;      // grows by the temporary columns
  Schema schema;  // of `in`
  int temporaries = 0;

  static bool leaf(const ExprPtr &e) { return e->kind != EK_BINARY; }
  bool over(const ExprPtr &e) const {
    const ExprCost c = expr_cost(e, schema);
    if (c.insns > FIT_INSNS || c.lits > FIT_LITS || c.slots > VM_MAX_DEPTH) return true;
    std::map<int, int> seen;
    expr_columns(e, seen);
    return (int)seen.size() > FIT_COLS;
  }
  ExprPtr materialise(const ExprPtr &e) {
    Batch t = run_pipeline_typed(ctx, in, {e}, nullptr);
    in.cols.push_back(t.cols[0]);
    schema = schema_of(in);
    ++temporaries;
    return make_column((int)in.cols.size() - 1);
  }
  // The expression with every subtree small enough for one kernel (expects a normalised, type-checked tree).
  ExprPtr shrink(const ExprPtr &e) {
    if (leaf(e)) return e;
    if (e->to_f64 || !e->r) {
      auto n = std::make_shared<Expr>(*e);
      n->l = shrink(e->l);
      return n;
    }
    if (expr_type(e->l, schema) == QE_TYPE_UTF8) return e;  // one comparison instruction
    ExprPtr l = shrink(e->l), r = shrink(e->r);
    auto node = [&] {
      auto n = std::make_shared<Expr>(*e);
      n->l = l;
      n->r = r;
      return ExprPtr(n);
    };
    ExprPtr cur = node();
    while (over(cur) && !(leaf(l) && leaf(r))) {
      // the costlier operand becomes a column
      const bool left = leaf(r) || (!leaf(l) && expr_cost(l, schema).insns >= expr_cost(r, schema).insns);
      if (left) l = materialise(l);
      else r = materialise(r);
      cur = node();
    }
    return cur;
  }
  // Expressions that share one program (an aggregate's predicate, keys and accumulator inputs): the largest
  // become columns until all of them fit.  An expression listed twice is compiled twice.
  void fit_together(const std::vector<ExprPtr *> &exprs, int reserve_insns, int reserve_lits) {
    for (;;) {
      int insns = reserve_insns, lits = reserve_lits, worst_insns = 0;
      std::map<int, int> seen;
      ExprPtr *worst = nullptr;
      for (ExprPtr *p : exprs) {
        if (!*p) continue;
        const ExprCost c = expr_cost(*p, schema);
        insns += c.insns + 2;
        lits += c.lits;
        expr_columns(*p, seen);
        if (!leaf(*p) && c.insns > worst_insns) worst = p, worst_insns = c.insns;
      }
      if (!worst || (insns <= VM_MAX_INSN && lits <= VM_MAX_LITS && (int)seen.size() <= VM_MAX_COLS)) return;
      *worst = materialise(*worst);
    }
  }
};

// Timestamp columns only ever pass through (the reference's ColumnExpression would not even
// hand them out, physical_expressions.rs:37-60): they travel as the 64-bit integers they are
// and get their type back on the way out.
static Batch run_pipeline(Context *ctx, const Batch &in, const std::vector<ExprPtr> &outputs_raw, const ExprPtr &pred_raw) {
  bool any = false;
  for (auto &c : in.cols) any = any || c.type == QE_TYPE_TIMESTAMP_NS;
  if (!any) return run_pipeline_typed(ctx, in, outputs_raw, pred_raw);
  if (mentions_column_of_type(pred_raw, in, QE_TYPE_TIMESTAMP_NS))
    fail(QE_ERR_PRIMITIVE_TYPE_NOT_SUPPORTED, "Primitive type `Timestamp(Nanosecond, None)` is not supported.");
  std::vector<bool> restore(outputs_raw.size(), false);
  for (size_t i = 0; i < outputs_raw.size(); ++i) {
    const ExprPtr &e = outputs_raw[i];
    if (e->kind == EK_COLUMN) restore[i] = mentions_column_of_type(e, in, QE_TYPE_TIMESTAMP_NS);
    else if (mentions_column_of_type(e, in, QE_TYPE_TIMESTAMP_NS))
      fail(QE_ERR_PRIMITIVE_TYPE_NOT_SUPPORTED, "Primitive type `Timestamp(Nanosecond, None)` is not supported.");
  }
  Batch plain = in;
  for (auto &c : plain.cols)
    if (c.type == QE_TYPE_TIMESTAMP_NS) c.type = QE_TYPE_INT64;
  Batch out = run_pipeline_typed(ctx, plain, outputs_raw, pred_raw);
  for (size_t i = 0; i < restore.size(); ++i)
    if (restore[i]) out.cols[i].type = QE_TYPE_TIMESTAMP_NS;
  return out;
}

static Batch run_pipeline_typed(Context *ctx, const Batch &in_raw, const std::vector<ExprPtr> &outputs_raw, const ExprPtr &pred_raw) {
  Batch in = without_redundant_validity(in_raw);
  ctx->bind();
  const int64_t n = in.rows;
  Schema schema = schema_of(in);
  std::vector<ExprPtr> outs;
  for (auto &e : outputs_raw) outs.push_back(expr_normalize(e, ctx->flags));
  ExprPtr pred = pred_raw ? expr_normalize(pred_raw, ctx->flags) : nullptr;
  std::vector<int> types;
  for (auto &e : outs) types.push_back(expr_type(e, schema));
  if (pred) {
    // Selection needs a BooleanArray (physical_plan/mod.rs:215-218)
    if (expr_type(pred, schema) != QE_TYPE_BOOLEAN)
      fail(QE_ERR_NO_BOOLEAN_ARRAY_FOR_FILTER, "Expr doesn't evaluate to a boolean array, which is needed to filter.");
    if (pred->kind == EK_LITERAL && !pred->valid)
      fail(QE_ERR_SCALAR_TO_ARRAY, "Couldn't convert Scalar value `null` to array.");
  }
  for (auto &e : outs)
    if (e->kind == EK_LITERAL && !e->valid)
      fail(QE_ERR_SCALAR_TO_ARRAY, "Couldn't convert Scalar value `null` to array.");  // columnar_value.rs:33
  {
    ExprFitter fitter{ctx, in, schema};
    for (auto &e : outs) e = fitter.shrink(e);
    if (pred) pred = fitter.shrink(pred);
    if (fitter.temporaries) schema = fitter.schema;
  }

  Batch out;
  out.ctx = ctx;
  out.cols.resize(outs.size());

  if (!pred) {
    out.rows = n;
    std::vector<int> vm_list;
    for (size_t i = 0; i < outs.size(); ++i) {
      const ExprPtr &e = outs[i];
      if (e->kind == EK_COLUMN) {
        out.cols[i] = in.cols[e->index];  // Arc clone in the reference (physical_expressions.rs:41)
      } else if (e->kind == EK_LITERAL && e->type == QE_TYPE_UTF8) {
        out.cols[i] = utf8_broadcast_column(ctx, e->sval, n);
      } else {
        if (types[i] == QE_TYPE_UTF8) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Physical type `Utf8` is not supported in a computed expression.");
        vm_list.push_back((int)i);
      }
    }
    size_t k = 0;
    while (k < vm_list.size()) {
      Compiler comp(ctx, &in);
      ProjectArgs args{};
      k = lower_project(comp, schema, outs, types, vm_list, k, n, args, [&](int i, bool nullable, VmOut &o) {
        Column c = make_fixed_column(ctx, types[i], n, nullable, false);
        finish_fixed_column(c, n);
        o.values = c.values->ptr;
        o.validity = c.validity ? (uint32_t *)c.validity->ptr : nullptr;
        out.cols[i] = c;
      });
      const char *kname = "qe_project_kernel";
      ctx->prof_begin(kname, n);
      QE_HIP(launch_project(args, ctx->dev, ctx->stream, &kname));
      ctx->prof_end(kname);
      if (!comp.keepalive.empty()) ctx->sync();
    }
    return out;
  }

  // ---- filtered
  std::vector<int> vm_list, utf8_cols, utf8_lits;
  for (size_t i = 0; i < outs.size(); ++i) {
    const ExprPtr &e = outs[i];
    if (types[i] == QE_TYPE_UTF8) {
      if (e->kind == EK_COLUMN) utf8_cols.push_back((int)i);
      else if (e->kind == EK_LITERAL) utf8_lits.push_back((int)i);
      else fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Physical type `Utf8` is not supported in a computed expression.");
    } else {
      vm_list.push_back((int)i);
    }
  }
  ensure_counters(ctx);
  const size_t out_bb = bitmap_bytes(n);
  size_t nullable_outs = 0, next_bitmap = 0;
  for (int i : vm_list) nullable_outs += expr_nullable(outs[i], schema) ? 1 : 0;
  int64_t m = 0;
  Column rowid;
  bool need_rowid = !utf8_cols.empty();
  size_t k = 0;
  bool first = true;
  // QE_FLAG_DEFERRED_ROW_COUNT: the batch is handed back with the kernels queued and its row count on its way to pinned
  // words of its own; nothing here waits for the device.  Not for what needs the count on the host to go on: string
  // literals as outputs (their broadcast is sized by it), more string columns than the pinned block has room for,
  // more outputs than one filter launch takes.
  std::shared_ptr<PendingRows> pending;
  if ((ctx->flags & QE_FLAG_DEFERRED_ROW_COUNT) && n > 0 && utf8_lits.empty() && 8 + 4 * utf8_cols.size() <= (size_t)Context::PENDING_WORDS) {
    pending = std::make_shared<PendingRows>();
    pending->ctx_ref = ctx->shared_from_this();
    pending->host = ctx->take_pending_words();
    pending->ready = ctx->take_pending_event();
    pending->utf8_slot.assign(outs.size(), -1);
    pending->filter_at = 4 * (int)utf8_cols.size();
  }
  // Everything the kernels of a deferred selection need zeroed, in one block under one fill: the string columns' control
  // blocks, the filter's scratch right behind them, the fixed-width outputs' validity bitmaps, the string columns'
  // status arrays and bitmaps.
  GatherArena arena;
  size_t arena_scratch_at = 0, arena_out_bitmaps_at = 0;
  FilterScratch *scratch;
  BufferPtr out_bitmaps;
  size_t out_bitmaps_at = 0;
  if (pending) {
    auto up64 = [](size_t v) { return (v + 63) / 64 * 64; };
    const size_t n_u = utf8_cols.size();
    const size_t control = up64(n_u * 32);
    arena.control_at = control - n_u * 32;  // (the last control block ends where the filter's scratch begins)
    arena_scratch_at = control;
    arena_out_bitmaps_at = arena_scratch_at + up64(filter_scratch_bytes(n));
    arena.status_at = arena_out_bitmaps_at + up64(nullable_outs * out_bb);
    const size_t status_bytes = n_u ? (utf8_gather_offsets_status_bytes(n) + 63) / 64 * 64 : 0;
    arena.bitmaps_at = arena.status_at + n_u * status_bytes + sizeof(FilterScratch);
    size_t nullable_strings = 0;
    for (int i : utf8_cols) nullable_strings += in.cols[outs[i]->index].validity ? 1 : 0;
    const size_t total = arena.bitmaps_at + up64(nullable_strings * out_bb);
    arena.block = device_buffer(ctx, total);
    QE_HIP(hipMemsetAsync(arena.block->ptr, 0, total, ctx->stream));
    scratch = (FilterScratch *)((char *)arena.block->ptr + arena_scratch_at);
    if (nullable_outs) {
      out_bitmaps = arena.block;
      out_bitmaps_at = arena_out_bitmaps_at;
    }
  } else {
    ensure_filter_scratch(ctx, n);
    scratch = (FilterScratch *)ctx->filter_scratch;
    // the validity bitmaps of all nullable outputs in one block, cleared by one fill
    if (nullable_outs > 1) {
      out_bitmaps = device_buffer(ctx, nullable_outs * out_bb);
      QE_HIP(hipMemsetAsync(out_bitmaps->ptr, 0, nullable_outs * out_bb, ctx->stream));
    }
  }
  const bool scratch_cleared = pending != nullptr;  // (the first pass finds it zeroed)
  while (first || k < vm_list.size()) {
    Compiler comp(ctx, &in);
    FilterArgs args{};
    void *rowid_values = nullptr;
    if (first && need_rowid) {
      if (n > 2147483647LL) fail(QE_ERR_INVALID_ARGUMENT, "a selection over string columns takes batches of at most 2^31 - 1 rows");
      rowid = make_fixed_column(ctx, QE_TYPE_INT32, n, false, false);
      rowid_values = rowid.values->ptr;
    }
    k = lower_filter(comp, schema, pred, outs, types, vm_list, k, n, rowid_values, args, [&](int i, bool nullable, VmOut &o) {
      Column c = make_fixed_column(ctx, types[i], n, nullable, false,
                                   nullable && out_bitmaps && next_bitmap < nullable_outs ? sub_buffer(out_bitmaps, out_bitmaps_at + next_bitmap++ * out_bb, out_bb) : nullptr);
      o.values = c.values->ptr;
      o.validity = c.validity ? (uint32_t *)c.validity->ptr : nullptr;
      out.cols[i] = c;
    });
    if (!first && args.n_outs == 0) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression is too long for one kernel.");
    // (outputs that take more than one pass: every pass's result words are checked against the first's on the host)
    if (pending && k < vm_list.size()) pending.reset();
    args.scratch = scratch;
    if (!(first && scratch_cleared)) QE_HIP(launch_filter_reset(scratch, n, ctx->stream));
    const char *kname = "qe_filter_kernel";
    ctx->prof_begin(kname, n);
    QE_HIP(launch_filter(args, ctx->dev, ctx->stream, &kname));
    ctx->prof_end(kname);
    if (pending) {  // (one pass: see above)
      first = false;
      continue;
    }
    // every pass runs the ticket / look-back protocol anew: each one's result words are read back (a look-back that
    // timed out leaves rows unwritten)
    QE_HIP(hipMemcpyAsync(ctx->host_counters, scratch, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (ctx->host_counters[2] != 0) fail(QE_ERR_INTERNAL, "filter look-back timed out");
    if (first) m = n > 0 ? (int64_t)ctx->host_counters[1] : 0;
    else if (n > 0 && (int64_t)ctx->host_counters[1] != m) fail(QE_ERR_INTERNAL, "filter passes selected different numbers of rows");
    first = false;
  }
  if (pending) {
    out.rows = n;
    for (int i : vm_list) finish_fixed_column(out.cols[i], n);
    if (!utf8_cols.empty()) {
      std::vector<const Column *> srcs;
      for (size_t j = 0; j < utf8_cols.size(); ++j) {
        srcs.push_back(&in.cols[outs[utf8_cols[j]]->index]);
        pending->utf8_slot[utf8_cols[j]] = (int)j;
      }
      std::vector<Column> got = utf8_gather_columns(ctx, srcs, (const int32_t *)rowid.values->ptr, n, &scratch->total, pending.get(), 0, &arena);
      for (size_t j = 0; j < utf8_cols.size(); ++j) out.cols[utf8_cols[j]] = std::move(got[j]);
    }
    // the string columns' control blocks and, behind them, the filter's result words: one copy
    QE_HIP(hipMemcpyAsync(pending->host, (char *)arena.block->ptr + arena.control_at, (4 * utf8_cols.size() + 3) * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                          ctx->stream));
    QE_HIP(hipEventRecord(pending->ready, ctx->stream));
    out.pending = pending;
    return out;
  }
  out.rows = m;
  for (int i : vm_list) finish_fixed_column(out.cols[i], m);
  if (!utf8_cols.empty()) {
    std::vector<const Column *> srcs;
    for (int i : utf8_cols) srcs.push_back(&in.cols[outs[i]->index]);
    std::vector<Column> got = utf8_gather_columns(ctx, srcs, (const int32_t *)rowid.values->ptr, m);
    for (size_t j = 0; j < utf8_cols.size(); ++j) out.cols[utf8_cols[j]] = std::move(got[j]);
  }
  for (int i : utf8_lits) out.cols[i] = utf8_broadcast_column(ctx, outs[i]->sval, m);
  // (the row ids go back to the pool on return with the byte copies still reading them: the pool hands a block out
  // again only to work queued later on this stream; the host is free to prepare the next batch meanwhile)
  return out;
}

struct PipeIter : Iter {
  Context *ctx;
  Stream s;
  bool next(Batch &out) override {
    Batch in;
    if (!s.src->next(in)) return false;
    if (s.identity && !s.pred) {
      out = in;
      return true;
    }
    out = run_pipeline(ctx, in, stream_outputs(s, (int)in.cols.size()), s.pred);
    return true;
  }
  void reset() override { s.src->reset(); }
};

// ---------------------------------------------------------------- aggregate
struct AccDesc {
  VmAcc acc{};
  ExprPtr input;
};
struct AggSpec {
  std::vector<VmKey> keys;
  std::vector<ExprPtr> key_exprs;
  std::vector<AccDesc> accs;
  std::vector<FinOut> outs;  // pointers filled at finalize time
  // Key columns that are not unpacked from the 64-bit group key (a string key, or a tuple of keys identified by
  // its hash) but gathered from the source batches by a representative row of the group: output column (it holds
  // those row numbers until the gather), source column, type.
  struct KeyGather {
    int out, col, type;
  };
  std::vector<KeyGather> key_gathers;
  std::vector<int> agg_first_out;  // per aggregate: its (first) output column
};

static int add_acc(AggSpec &spec, uint8_t op, int in_type, bool track_nn, const ExprPtr &input) {
  // aggregates over the same input share one accumulator (AVG(x) next to SUM(x) / COUNT(x))
  // the row-number input differs between batches only in its base: one accumulator all the same
  std::string key = input ? (input->kind == EK_ROWID ? std::string("rowid") : expr_identity(input)) : std::string();
  for (size_t i = 0; i < spec.accs.size(); ++i) {
    AccDesc &d = spec.accs[i];
    if (d.acc.op == op && d.acc.in_type == (uint8_t)in_type && (d.input != nullptr) == (input != nullptr) &&
        (!input || (d.input->kind == EK_ROWID ? std::string("rowid") : expr_identity(d.input)) == key)) {
      if (track_nn) d.acc.track_nn = 1;
      return (int)i;
    }
  }
  if ((int)spec.accs.size() >= VM_MAX_ACCS)
    fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Too many aggregate accumulators for one kernel.");
  AccDesc d;
  d.acc.op = op;
  d.acc.in_type = (uint8_t)in_type;
  d.acc.track_nn = track_nn ? 1 : 0;
  d.acc.has_input = input ? 1 : 0;
  d.input = input;
  spec.accs.push_back(d);
  return (int)spec.accs.size() - 1;
}

static FinOut fin(uint8_t kind, int type, int a, int b, uint8_t cnt_src, uint8_t null_mode) {
  FinOut f{};
  f.kind = kind;
  f.type = (uint8_t)type;
  f.a = (int8_t)a;
  f.b = (int8_t)b;
  f.cnt_src = cnt_src;
  f.null_mode = null_mode;
  return f;
}

static ExprPtr make_to_f64(const ExprPtr &e) {
  auto n = std::make_shared<Expr>();
  n->kind = EK_BINARY;
  n->to_f64 = true;
  n->l = e;
  return n;
}

static bool numeric(int t) { return t == QE_TYPE_INT32 || t == QE_TYPE_INT64 || t == QE_TYPE_FLOAT64; }

// Group key tuples that do not fit the 64-bit packed key - a string next to other keys, several strings, more
// than 64 bits of integers, more fields than the packed key has - are identified by a 64-bit hash of the tuple.
constexpr int MAX_HASHED_KEYS = 12;
static bool key_tuple_hashed(const std::vector<int> &types) {
  int total_bits = 0, strings = 0;
  for (int t : types) {
    strings += t == QE_TYPE_UTF8;
    total_bits += t == QE_TYPE_INT32 ? 32 : t == QE_TYPE_INT64 ? 64 : t == QE_TYPE_BOOLEAN ? 1 : t == QE_TYPE_UTF8 ? 64 : 0;
  }
  return types.size() > 1 && (strings > 0 || total_bits > 64 || (int)types.size() > VM_MAX_KEYS);
}

// An order-independent Float64 sum of `x` at scale `S`: three integer sums of the limbs of x's fixed-point image and
// one of its infinity / NaN marks, in consecutive accumulators; the first one's index.
// `overflow_word`: device address of the word a kernel sets when a value does not fit scale S (0: S is known to fit).
static int add_fix_sum(AggSpec &spec, const ExprPtr &x, int S, bool track_nn, uint64_t overflow_word) {
  const int a = add_acc(spec, ACC_SUM_I64, QE_TYPE_INT64, track_nn, make_fix(FIX_LIMB0, x, S));
  const int b = add_acc(spec, ACC_SUM_I64, QE_TYPE_INT64, false, make_fix(FIX_LIMB1, x, S));
  const int c = add_acc(spec, ACC_SUM_I64, QE_TYPE_INT64, false, make_fix(FIX_LIMB2, x, S));
  const int d = add_acc(spec, ACC_SUM_I64, QE_TYPE_INT64, false, make_fix(FIX_SPECIAL, x, S, overflow_word));
  if (b != a + 1 || c != a + 2 || d != a + 3) fail(QE_ERR_INTERNAL, "the accumulators of an order-independent sum are not consecutive");
  return a;
}

// `fix_scales`: per aggregate the scale of its order-independent sum (QE_FLAG_DETERMINISTIC_SUM), or null.
static AggSpec build_agg_spec(uint32_t flags, const Schema &schema, const std::vector<ExprPtr> &groups,
                              const std::vector<AggExpr> &aggs, const std::vector<ExprPtr> &aggs_second, int mode,
                              int64_t row_base = 0, bool emit_row_number = false, const std::vector<int> *fix_scales = nullptr,
                              uint64_t fix_overflow_word = 0, bool explicit_counts = false) {
  // `explicit_counts`: whether a SUM (or an integer MIN / MAX) over a nullable input has seen a value at all is told by
  // a COUNT accumulator over the same input, not by the validity counters beside the accumulators - the many-groups
  // mode's log and partition passes carry accumulators only.
  AggSpec spec;
  int shift = 0;
  if ((int)groups.size() > MAX_HASHED_KEYS) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Too many group expressions.");
  // A hashed key tuple identifies groups as the reference identifies every group (physical_plan/mod.rs:317-358
  // sums per-column hashes); the key columns of the output are gathered from a representative row of each group,
  // so every key is a plain column here (AggState::update evaluates computed keys into columns beforehand).
  {
    std::vector<ExprPtr> norm;
    std::vector<int> key_types;
    for (auto &g : groups) {
      norm.push_back(expr_normalize(g, flags));
      key_types.push_back(expr_type(norm.back(), schema));
    }
    if (key_tuple_hashed(key_types)) {
      ExprPtr h;
      for (auto &e : norm) {
        const int t = expr_type(e, schema);
        if (!(t == QE_TYPE_INT32 || t == QE_TYPE_INT64 || t == QE_TYPE_BOOLEAN || t == QE_TYPE_UTF8))
          fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(t) + "` is not supported as a group key.");
        if (e->kind != EK_COLUMN)
          fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Group keys identified by a hash of the tuple must be plain columns.");
        h = make_key_hash(h, e);
      }
      if (flags & QE_FLAG_DEBUG_WEAK_KEY_HASH) h = make_binary(QE_OP_MOD, h, make_lit_i64(5));  // (testing: distinct keys collide)
      VmKey k{};
      k.type = (uint8_t)QE_TYPE_INT64;
      k.shift = 0;
      k.bits = 64;
      spec.keys.push_back(k);
      spec.key_exprs.push_back(h);
      const int a = add_acc(spec, ACC_MIN, QE_TYPE_INT64, false, make_rowid(row_base));
      for (auto &e : norm) {
        spec.key_gathers.push_back(AggSpec::KeyGather{(int)spec.outs.size(), e->index, expr_type(e, schema)});
        spec.outs.push_back(fin(FIN_ACC_ORD, QE_TYPE_INT64, a, a, CNT_ROWS, NULL_NEVER));
      }
      shift = 64;
    }
  }
  for (size_t g = 0; g < groups.size() && spec.key_gathers.empty(); ++g) {
    ExprPtr e = expr_normalize(groups[g], flags);
    int t = expr_type(e, schema);
    if (t == QE_TYPE_UTF8) {
      // A string key is aggregated by a 64-bit hash of its bytes (the reference identifies groups by
      // hash alone, physical_plan/mod.rs:317-358); the smallest row number of the group is kept as a
      // hidden accumulator and the output key column is gathered from those rows.
      if (groups.size() != 1 || e->kind != EK_COLUMN)
        fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "A Utf8 group key must be a plain column and the only group expression.");
      VmKey k{};
      k.type = (uint8_t)QE_TYPE_UTF8;
      k.shift = 0;
      k.bits = 64;
      spec.keys.push_back(k);
      spec.key_exprs.push_back((flags & QE_FLAG_DEBUG_WEAK_KEY_HASH) ? make_binary(QE_OP_MOD, make_hash_utf8(e), make_lit_i64(5)) : make_hash_utf8(e));
      int a = add_acc(spec, ACC_MIN, QE_TYPE_INT64, false, make_rowid(row_base));
      spec.key_gathers.push_back(AggSpec::KeyGather{(int)spec.outs.size(), e->index, QE_TYPE_UTF8});
      spec.outs.push_back(fin(FIN_ACC_ORD, QE_TYPE_INT64, a, a, CNT_ROWS, NULL_NEVER));
      shift = 64;
      continue;
    }
    int bits = t == QE_TYPE_INT32 ? 32 : t == QE_TYPE_INT64 ? 64 : t == QE_TYPE_BOOLEAN ? 1 : 0;
    // arrow2 cannot hash floats (the reference unwrap()s: physical_plan/mod.rs:317)
    if (bits == 0) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(t) + "` is not supported as a group key.");
    if (shift + bits > 64) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Group key wider than 64 bits is not supported.");
    VmKey k{};
    k.type = (uint8_t)t;
    k.shift = (uint8_t)shift;
    k.bits = (uint8_t)bits;
    spec.keys.push_back(k);
    spec.key_exprs.push_back(e);
    FinOut f = fin(FIN_KEY, t, (int)g, 0, CNT_ROWS, NULL_NEVER);
    f.shift = (uint8_t)shift;
    f.bits = (uint8_t)bits;
    spec.outs.push_back(f);
    shift += bits;
  }
  for (size_t i = 0; i < aggs.size(); ++i) {
    const AggExpr &ag = aggs[i];
    spec.agg_first_out.push_back((int)spec.outs.size());
    ExprPtr e = ag.input ? expr_normalize(ag.input, flags) : nullptr;
    int t = e ? expr_type(e, schema) : 0;
    bool nullable = e ? expr_nullable(e, schema) : false;
    if (e && e->kind == EK_LITERAL && !e->valid) fail(QE_ERR_SCALAR_TO_ARRAY, "Couldn't convert Scalar value `null` to array.");
    auto need_numeric = [&]() {
      if (!e || !numeric(t))
        fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + (e ? type_name(t) : "none") + "` is not supported.");
    };
    bool final_mode = mode == QE_AGG_MODE_FINAL;
    switch (ag.kind) {
      case QE_AGG_MAX:
      case QE_AGG_MIN: {
        need_numeric();
        bool is_min = ag.kind == QE_AGG_MIN;
        bool f64 = t == QE_TYPE_FLOAT64;
        if (explicit_counts && nullable && !f64) {
          const int a = add_acc(spec, is_min ? ACC_MIN : ACC_MAX, t, false, e), b = add_acc(spec, ACC_COUNT, t, false, e);
          spec.outs.push_back(fin(FIN_ACC_ORD, t, a, b, CNT_ACC, NULL_IF_COUNT_ZERO));
          break;
        }
        int a = add_acc(spec, is_min ? ACC_MIN : ACC_MAX, t, nullable && !f64, e);
        uint8_t nm = f64 ? (is_min ? NULL_IF_ACC_ALL_ONES : NULL_IF_ACC_ZERO) : (nullable ? NULL_IF_COUNT_ZERO : NULL_NEVER);
        spec.outs.push_back(fin(FIN_ACC_ORD, t, a, a, CNT_NN, nm));
        break;
      }
      case QE_AGG_SUM: {
        need_numeric();
        bool f64 = t == QE_TYPE_FLOAT64;
        if (f64 && fix_scales) {
          const int a = add_fix_sum(spec, e, (*fix_scales)[i], nullable, fix_overflow_word);
          FinOut f = fin(FIN_FIX_SUM, t, a, a, CNT_NN, nullable ? NULL_IF_COUNT_ZERO : NULL_NEVER);
          f.scale = (int16_t)(*fix_scales)[i];
          spec.outs.push_back(f);
          break;
        }
        if (explicit_counts && nullable) {
          const int a = add_acc(spec, f64 ? ACC_SUM_F64 : ACC_SUM_I64, t, false, e), b = add_acc(spec, ACC_COUNT, t, false, e);
          spec.outs.push_back(fin(f64 ? FIN_ACC_F64 : FIN_ACC_I64, t, a, b, CNT_ACC, NULL_IF_COUNT_ZERO));
          break;
        }
        int a = add_acc(spec, f64 ? ACC_SUM_F64 : ACC_SUM_I64, t, nullable, e);
        spec.outs.push_back(fin(f64 ? FIN_ACC_F64 : FIN_ACC_I64, t, a, a, CNT_NN, nullable ? NULL_IF_COUNT_ZERO : NULL_NEVER));
        break;
      }
      case QE_AGG_COUNT: {
        if (final_mode) {
          if (!e || t != QE_TYPE_INT64) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "COUNT partial column must be Int64.");
          int a = add_acc(spec, ACC_SUM_I64, t, false, e);
          spec.outs.push_back(fin(FIN_ACC_I64, QE_TYPE_INT64, a, a, CNT_ROWS, NULL_NEVER));
        } else if (!e || !nullable) {
          spec.outs.push_back(fin(FIN_COUNT, QE_TYPE_INT64, 0, 0, CNT_ROWS, NULL_NEVER));
        } else {
          int a = add_acc(spec, ACC_COUNT, t, false, e);
          spec.outs.push_back(fin(FIN_COUNT, QE_TYPE_INT64, a, a, CNT_ACC, NULL_NEVER));
        }
        break;
      }
      case QE_AGG_AVG: {
        if (final_mode) {
          ExprPtr cnt_e = aggs_second[i] ? expr_normalize(aggs_second[i], flags) : nullptr;
          if (!e || t != QE_TYPE_FLOAT64 || !cnt_e || expr_type(cnt_e, schema) != QE_TYPE_INT64)
            fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "AVG partial columns must be (Float64 sum, Int64 count).");
          int a = fix_scales ? add_fix_sum(spec, e, (*fix_scales)[i], false, fix_overflow_word) : add_acc(spec, ACC_SUM_F64, QE_TYPE_FLOAT64, false, e);
          int b = add_acc(spec, ACC_SUM_I64, QE_TYPE_INT64, false, cnt_e);
          FinOut f = fin(fix_scales ? FIN_FIX_AVG : FIN_AVG, QE_TYPE_FLOAT64, a, b, CNT_ACC, NULL_IF_COUNT_ZERO);
          if (fix_scales) f.scale = (int16_t)(*fix_scales)[i];
          spec.outs.push_back(f);
          break;
        }
        need_numeric();
        // (a Float64 input needs no cast: AVG(x) then shares SUM(x)'s accumulator)
        const ExprPtr as_f64 = t == QE_TYPE_FLOAT64 ? e : make_to_f64(e);
        int a = fix_scales ? add_fix_sum(spec, as_f64, (*fix_scales)[i], false, fix_overflow_word) : add_acc(spec, ACC_SUM_F64, QE_TYPE_FLOAT64, false, as_f64);
        int b = -1;
        uint8_t src = CNT_ROWS;
        if (nullable) {
          b = add_acc(spec, ACC_COUNT, t, false, e);
          src = CNT_ACC;
        }
        FinOut f;
        if (mode == QE_AGG_MODE_PARTIAL) {
          f = fin(fix_scales ? FIN_FIX_SUM : FIN_ACC_F64, QE_TYPE_FLOAT64, a, a, CNT_ROWS, NULL_NEVER);
        } else {
          f = fin(fix_scales ? FIN_FIX_AVG : FIN_AVG, QE_TYPE_FLOAT64, a, b < 0 ? 0 : b, src, NULL_IF_COUNT_ZERO);
        }
        if (fix_scales) f.scale = (int16_t)(*fix_scales)[i];
        spec.outs.push_back(f);
        if (mode == QE_AGG_MODE_PARTIAL) spec.outs.push_back(fin(FIN_COUNT, QE_TYPE_INT64, b < 0 ? 0 : b, b < 0 ? 0 : b, src, NULL_NEVER));
        break;
      }
      default:
        fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression `aggregate` is not supported.");
    }
  }
  if (emit_row_number) {
    // the smallest row number of the group, as a last output column: it names the group in every part of a split aggregation
    const int a = add_acc(spec, ACC_MIN, QE_TYPE_INT64, false, make_rowid(row_base));
    spec.outs.push_back(fin(FIN_ACC_ORD, QE_TYPE_INT64, a, a, CNT_ROWS, NULL_NEVER));
  }
  if (spec.outs.size() > sizeof(FinalizeArgs::outs) / sizeof(FinOut))
    fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Too many output columns for one aggregate.");
  return spec;
}

// The aggregate kernel's task list: predicate, key fields, group lookup, one fold per accumulator.
// `tracked` carries the validity tracking decided on the first batch of the scan.
static void lower_aggregate(Compiler &comp, const Schema &schema, const ExprPtr &pred, const AggSpec &bspec, const AggSpec &tracked,
                            AggArgs &args) {
  int nt = 0;
  if (pred) {
    if (expr_type(pred, schema) != QE_TYPE_BOOLEAN)
      fail(QE_ERR_NO_BOOLEAN_ARRAY_FOR_FILTER, "Expr doesn't evaluate to a boolean array, which is needed to filter.");
    args.tasks[nt++] = VmTask{comp.add(pred), ROLE_PRED, 0, 1, 0};
  }
  for (size_t i = 0; i < bspec.keys.size(); ++i) {
    args.keys[i] = bspec.keys[i];
    args.keys[i].pc = comp.add(bspec.key_exprs[i]);
    args.tasks[nt++] = VmTask{args.keys[i].pc, ROLE_KEY, (uint8_t)i, 1, 0};
  }
  args.tasks[nt++] = VmTask{0, ROLE_LOOKUP, 0, 0, 0};
  for (size_t i = 0; i < bspec.accs.size(); ++i) {
    args.accs[i] = bspec.accs[i].acc;
    args.accs[i].track_nn = tracked.accs[i].acc.track_nn;
    if (bspec.accs[i].input) args.accs[i].pc = comp.add(bspec.accs[i].input);
    args.tasks[nt++] = VmTask{args.accs[i].pc, ROLE_ACC, (uint8_t)i, (uint8_t)(bspec.accs[i].input ? 1 : 0), 0};
  }
  args.n_tasks = nt;
  args.prog = comp.prog;
  args.n_keys = (int)bspec.keys.size();
  args.n_accs = (int)bspec.accs.size();
}

// ------------------------------------------------------ high-cardinality mode
// With very many groups most rows miss the workgroups' LDS tables, and one scattered global atomic
// per row and accumulator is what the kernel then spends its time on.  When the caller's group-count
// hint says so, those rows are logged instead (AggSpill), partitioned by key hash and aggregated
// partition by partition in LDS: a few streaming passes over a fraction of the rows.
static int64_t spill_min_group_hint() {
  static int64_t v = [] {
    const char *e = getenv("QE_SPILL_MIN_GROUPS");
    return e && atoll(e) > 0 ? (int64_t)atoll(e) : (int64_t)1 << 16;
  }();
  return v;
}
constexpr int SPILL_MAX_INPUTS = 8;

struct SpillPlan {
  bool on = false;
  BufferPtr keys, vals, counts;
  int waves = 0;
  uint8_t acc_col[VM_MAX_ACCS];
  // the partition passes: hash bits that name a partition, workgroups, the cell tables [partition][workgroup]
  int bits = 0, blocks = 0;
  int sub_bits = 0;  // further hash bits by which the partition-wise aggregate folds a partition in several passes (PartArgs::sub_bits)
  BufferPtr cell_buf, cell1_buf, scan_buf;
  bool counted = false;  // the first pass filled the cell table and the log's length itself
  // The log can wait for the logs of the batches that follow and be partitioned and aggregated together with them
  // (SpillGroup): a first pass that counts its rows per partition itself, a log of pairs, the one-pass scatter.
  bool groupable = false, grouped = false;
  bool direct = false;  // the launch partitions its rows itself (AggSpill::direct_cap): `keys` are the stretches, `cell_buf` the rows in each
};

// The logs of consecutive batches of one aggregation, partitioned and aggregated together.  Every first pass ends by
// flushing its workgroups' tables, every partition-wise aggregate by flushing one table per partition - about a million
// groups each time with a million keys, whatever the number of rows in between - and the partition passes wait for the
// host to size their buffers.  Batch by batch that is paid per batch; here the first passes run back to back, each
// into its own log and its own block of columns of ONE cell table [partition][launch][workgroup], and a single scan,
// one wait, one scatter per log into one partitioned copy and one partition-wise aggregate follow when the group is
// full or the result is asked for.
struct SpillGroup {
  int bits = 0, sub_bits = 0, blocks = 0, slots = 0, n_accs = 0;
  BufferPtr cell_buf, scan_buf, ctrl;  // ctrl: [0] rows logged by the group's launches, [1] the partition ticket, [2] rows they selected, [3] rows partitioned directly
  struct Item {
    BufferPtr keys, vals, counts;
    AggSpill spill;
    int waves;
    bool direct;  // no log: `keys` holds the rows as the first pass partitioned them, `counts` the rows per (partition, workgroup) stretch
  };
  std::vector<Item> items;
  uint64_t log_rows = 0;  // capacity of the logs held, in rows
  VmAcc accs[VM_MAX_ACCS];
  uint8_t acc_col[VM_MAX_ACCS];
  AggTable table{};
};

// Partitions of the second phase: as few as keep a partition's distinct keys within 5/8 of its LDS table.
static int spill_partition_bits(int n_accs, int64_t group_hint) {
  const int64_t keys_per_partition = (int64_t)partitioned_lds_slots(n_accs) * 5 / 8;
  // (and no fewer than 256: a partition is one workgroup's work, and every compute unit should have some)
  int bits = 8;
  while (bits < 11 && ((int64_t)1 << bits) * keys_per_partition < group_hint) ++bits;
  static int forced_bits = [] {
    const char *e = getenv("QE_SPILL_BITS");
    return e ? atoi(e) : 0;
  }();
  if (forced_bits >= 1 && forced_bits <= 11 && forced_bits >= bits) bits = forced_bits;
  static int bits_cap = [] {
    const char *e = getenv("QE_SPILL_BITS_CAP");  // (experiments: fewer partitions than the hint asks for; what does not fit a partition's table goes to the global one row by row)
    return e ? atoi(e) : 0;
  }();
  if (bits_cap >= 8 && bits > bits_cap) bits = bits_cap;
  return bits;
}

static bool spill_enabled() {
  static int v = [] {
    const char *e = getenv("QE_AGGREGATE_SPILL");
    return e ? atoi(e) : 1;
  }();
  return v != 0;
}

// Decides on the mode for one launch and, when on, attaches the log buffers to `args`.
static SpillPlan plan_spill(Context *ctx, const Schema &schema, const AggSpec &bspec, AggArgs &args, int64_t group_hint, bool may_group = false,
                            bool want_direct = false) {
  SpillPlan sp;
  std::memset(args.spill_col, 0xff, sizeof(args.spill_col));
  std::memset(sp.acc_col, 0xff, sizeof(sp.acc_col));
  // When: the caller promises many groups; or, for batches large enough to pay for the second phase's fixed
  // costs, whenever the groups may outnumber a workgroup's LDS table - no hint at all, or one beyond the table.
  // (What does not fit the LDS table would otherwise go to the global table row by row, where a few thousand
  // keys mean heavily contended atomics: 64 M rows over 2 000 keys 44 ms, through the log 2.3 ms.)  A deferred
  // aggregate promises its caller not to wait for the device, which the second phase does.
  static const bool midrange = [] {
    const char *e = getenv("QE_SPILL_MIDRANGE");
    return e ? atoi(e) != 0 : true;
  }();
  const bool deferred = (ctx->flags & QE_FLAG_DEFERRED_GROUP_COUNT) != 0;
  const bool wanted = group_hint >= spill_min_group_hint() ||
                      (midrange && !deferred && args.n_rows >= (1 << 20) && (group_hint == 0 || group_hint >= 512));
  if (!spill_enabled() || !wanted || args.n_rows >= (1ll << 31) || args.n_accs == 0) return sp;
  // the log carries values only: an accumulator reads a non-nullable input, or one whose nulls can be marked
  std::vector<std::string> inputs;
  uint16_t null_cols = 0, null_f64 = 0;
  for (int i = 0; i < args.n_accs; ++i) {
    const AccDesc &d = bspec.accs[i];
    if (!d.input || args.accs[i].track_nn) return sp;
    // An input that can be null (a float64 MIN / MAX has no validity tracking of its own - a null is skipped like a
    // NaN - but its input can be null all the same; a COUNT counts the valid ones) is logged with a mark for nulls:
    // Float64 and Int32 inputs have bit patterns to spare for one, Int64 inputs have not.
    const bool marks_nulls = args.accs[i].op == ACC_COUNT || expr_nullable(d.input, schema);
    if (marks_nulls && args.accs[i].in_type != VT_F64 && args.accs[i].in_type != VT_I32) return sp;
    std::string key = expr_identity(d.input);
    size_t col = 0;
    while (col < inputs.size() && inputs[col] != key) ++col;
    if (col == inputs.size()) {
      if ((int)inputs.size() >= SPILL_MAX_INPUTS) return sp;
      inputs.push_back(key);
      args.spill_col[i] = (uint8_t)col;  // the first accumulator with this input logs it
    }
    sp.acc_col[i] = (uint8_t)col;
    if (marks_nulls) {
      null_cols |= (uint16_t)(1u << col);
      if (args.accs[i].in_type == VT_F64) null_f64 |= (uint16_t)(1u << col);
    }
  }
  // Rows will miss the workgroups' tables in numbers: when a kernel specialised for the shape exists, one wide workgroup
  // per compute unit with a table of four times the slots (fewer rows in the log, the same waves in flight)
  if (group_hint >= spill_min_group_hint()) args.tune |= 2u;
  if (aggregate_wide_available(args, ctx->dev)) {
    args.block_threads = AGG_WIDE_THREADS;
    args.lds_cap = aggregate_lds_slots(args, ctx->dev, 0, true);
    args.lds_limit = args.lds_cap - args.lds_cap / 4;
  }
  const int grid = aggregate_grid(args, ctx->dev);
  sp.waves = grid * (int)((args.block_threads ? args.block_threads : (uint32_t)VM_BLOCK_THREADS) / 64);
  const int64_t nsteps = (args.n_rows + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  const int64_t steps_per_wave = (nsteps + sp.waves - 1) / sp.waves;
  AggSpill &a = args.spill;
  // A region's stride is an odd number of 256-byte pieces: the waves fill their regions at about one pace, and with a
  // stride that is a multiple of a few KiB (whole steps of 256 rows: 512 KiB at 2^28 rows) they would all be writing
  // to the same few memory channels at any one time.
  static const int region_pad = [] {
    const char *e = getenv("QE_SPILL_REGION_PAD");
    return e ? atoi(e) : 0;
  }();
  // (a launch over several batches deals each batch's steps to the waves anew: a step more per batch and wave at most)
  a.rows_per_wave = (uint32_t)((steps_per_wave + (args.n_more > 0 ? args.n_more + 2 : 0)) * VM_WAVE_ROWS + region_pad);
  a.capacity = (uint64_t)sp.waves * a.rows_per_wave;
  a.n_inputs = (uint32_t)inputs.size();
  a.null_cols = null_cols;
  a.null_f64 = null_f64;
  static const bool pairs_on = [] {
    const char *e = getenv("QE_SPILL_PAIRS");  // 0: a key column and a value column whatever the number of value columns
    return e ? atoi(e) != 0 : true;
  }();
  a.pairs = pairs_on && a.n_inputs == 1 ? 1u : 0u;
  // the partition passes' geometry and tables, now: a first pass that can counts its logged rows per partition itself
  static int forced_blocks = [] {
    const char *e = getenv("QE_SPILL_BLOCKS");
    return e ? atoi(e) : 0;
  }();
  sp.bits = spill_partition_bits(args.n_accs, group_hint);
  // More groups than 512 partitions' tables hold: still 512 partitions - what the one-pass scatter, the first pass that
  // partitions directly and the grouped logs take - each folded in two or four passes over its rows by the next hash
  // bits.  (1024 partitions meant two scatter levels batch by batch: 2 M uniform keys, 500 M rows, 10.5 ms.)
  static const bool sub_passes = [] {
    const char *e = getenv("QE_SPILL_SUB_PASSES");
    return e ? atoi(e) != 0 : true;
  }();
  // (for logs of pairs: a log with several value columns takes the two scatter levels whatever the number of
  // partitions, and a second pass over it is all cost - two value columns, a million zipf keys: 10.2 -> 11.3 ms)
  if (sub_passes && a.pairs && sp.bits > 9) {
    sp.sub_bits = sp.bits - 9;
    sp.bits = 9;
  }
  sp.blocks = forced_blocks > 0 ? forced_blocks : ctx->dev.cus;
  if (want_direct && may_group && a.pairs && args.block_threads == (uint32_t)AGG_WIDE_THREADS && grid == sp.blocks && sp.blocks == ctx->dev.cus &&
      (args.tune & 2u) && sp.blocks * PART_MAX_DIRECT <= PART_MAX_DIRECT_UNITS && aggregate_direct_available(args, ctx->dev, sp.bits)) {
    // no log: one stretch of pairs per (partition, workgroup), the rows in each counted into a table of its own
    a.direct_cap = aggregate_direct_stretch_rows(args.n_rows, sp.bits, sp.blocks);
    const size_t stretches = spill_cells(sp.bits, sp.blocks);
    sp.keys = device_buffer(ctx, stretches * a.direct_cap * 16 + 256);
    sp.cell_buf = device_buffer(ctx, stretches * 4 + 64);
    a.keys = (unsigned long long *)sp.keys->ptr;
    a.vals = nullptr;
    a.capacity = (uint64_t)stretches * a.direct_cap;
    args.cells = (int32_t *)sp.cell_buf->ptr;
    args.cells_stride = (uint32_t)sp.blocks;
    args.cell_bits = sp.bits;
    sp.direct = sp.groupable = sp.on = true;
    return sp;
  }
  sp.keys = device_buffer(ctx, a.capacity * (a.pairs ? 16 : 8) + 64);
  sp.vals = device_buffer(ctx, a.pairs ? 64 : a.capacity * 8 * a.n_inputs + 64);
  sp.counts = device_buffer(ctx, (size_t)sp.waves * 4 + 64);
  QE_HIP(hipMemsetAsync(sp.counts->ptr, 0, (size_t)sp.waves * 4, ctx->stream));
  a.keys = (unsigned long long *)sp.keys->ptr;
  a.vals = (unsigned long long *)sp.vals->ptr;
  a.counts = (uint32_t *)sp.counts->ptr;
  static const int fuse_count = [] {
    const char *e = getenv("QE_SPILL_FUSED_COUNT");  // 0: off; 2 (timing experiments): the segment runs without the fused count
    return e ? atoi(e) : 1;
  }();
  const bool fused = fuse_count && args.block_threads == (uint32_t)AGG_WIDE_THREADS && grid == sp.blocks;
  static const bool group_on = [] {
    const char *e = getenv("QE_SPILL_GROUP");  // 0: every batch's log is partitioned and aggregated by itself
    return e ? atoi(e) != 0 : true;
  }();
  if (fused) a.seg_chunk = AGG_WIDE_THREADS / 64;  // one first-pass workgroup per column of the cell table: workgroup b's waves own segments [16 b, 16 b + 16)
  sp.groupable = may_group && group_on && fused && fuse_count != 2 && a.pairs && spill_onepass_available(sp.bits, a.n_inputs) &&
                 aggregate_counts_partitions(args, ctx->dev, sp.bits);
  if (!sp.groupable) {
    const size_t cells = spill_cells(sp.bits, sp.blocks), cells1 = spill_cells(spill_level1_bits(sp.bits), sp.blocks);
    sp.cell_buf = device_buffer(ctx, (cells + 1) * 4 + 64);
    sp.cell1_buf = device_buffer(ctx, (cells1 + 1) * 4 + 64);
    sp.scan_buf = device_buffer(ctx, scan_scratch_elems((int64_t)cells) * 4 + 64);
    QE_HIP(hipMemsetAsync(ctx->counters + 2, 0, 2 * sizeof(unsigned long long), ctx->stream));  // log length, partition ticket
    if (fused && fuse_count != 2) {
      args.cells = (int32_t *)sp.cell_buf->ptr;
      args.cells_total = ctx->counters + 2;
      args.cell_bits = sp.bits;
    }
  }
  sp.on = true;
  return sp;
}

// Second phase: partition the log by key hash (count, scan, one or two scatter passes), aggregate partition by partition.
static void aggregate_spilled(Context *ctx, const SpillPlan &sp, const AggArgs &args, int64_t group_hint) {
  (void)group_hint;
  const int bits = sp.bits, blocks = sp.blocks;
  const int bits1 = spill_level1_bits(bits);
  const size_t cells = spill_cells(bits, blocks), cells1 = spill_cells(bits1, blocks);
  const BufferPtr &scan_buf = sp.scan_buf;
  int32_t *cell = (int32_t *)sp.cell_buf->ptr, *cell1 = (int32_t *)sp.cell1_buf->ptr;
  unsigned long long *total = ctx->counters + 2;  // (zeroed before the first pass, with the partition ticket behind it)
  if (!sp.counted) {
    ctx->prof_begin("qe_spill_count_kernel");
    QE_HIP(launch_spill_count(args.spill, sp.waves, bits, blocks, cell, total, ctx->stream));
    ctx->prof_end();
  }
  const bool onepass = spill_onepass_available(bits, args.spill.n_inputs);
  if (bits1 < bits && !onepass) {
    QE_HIP(launch_spill_cells_level1(cell, cell1, bits, blocks, ctx->stream));
    QE_HIP(launch_exclusive_scan_i32(cell1, (int64_t)cells1, (int32_t *)scan_buf->ptr, ctx->stream));
  }
  QE_HIP(launch_exclusive_scan_i32(cell, (int64_t)cells, (int32_t *)scan_buf->ptr, ctx->stream));
  QE_HIP(hipMemcpyAsync(ctx->host_counters + 2, total, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync();
  const int64_t m = (int64_t)ctx->host_counters[2];
  static const bool verbose = getenv("QE_SPILL_VERBOSE") != nullptr;
  if (verbose) fprintf(stderr, "qe_gpu: high-cardinality mode: %lld of %lld rows logged, %d partitions\n", (long long)m, (long long)args.n_rows, 1 << bits);
  if (m == 0) return;
  const uint64_t stride = ((uint64_t)m + 63) & ~63ull;  // rows between the value columns of a partitioned copy
  const uint32_t n_in = args.spill.n_inputs;
  // One pass when the log has at most one value column: (key, value) pairs, written a whole 128-byte line at a time.
  // Otherwise whole-KiB bursts per column, which takes two passes for more than 64 partitions.
  BufferPtr out_keys, out_vals;
  if (onepass) {
    out_keys = device_buffer(ctx, stride * 16 + 256);
    ctx->prof_begin("qe_spill_scatter_onepass_kernel");
    QE_HIP(launch_spill_scatter_onepass(args.spill, sp.waves, bits, blocks, cell, blocks, out_keys->ptr, ctx->stream));
    ctx->prof_end();
  } else {
    out_keys = device_buffer(ctx, stride * 8 + 64);
    out_vals = device_buffer(ctx, stride * 8 * n_in + 64);
  }
  if (onepass) {
    // (done above)
  } else if (bits1 < bits) {
    BufferPtr mid_keys = device_buffer(ctx, stride * 8 + 64);
    BufferPtr mid_vals = device_buffer(ctx, stride * 8 * n_in + 64);
    ctx->prof_begin("qe_spill_scatter_kernel (level 1)");
    QE_HIP(launch_spill_scatter(args.spill, sp.waves, 1, bits1, bits1, blocks, nullptr, nullptr, 0, nullptr, m, cell1,
                                (unsigned long long *)mid_keys->ptr, (unsigned long long *)mid_vals->ptr, stride, ctx->stream));
    ctx->prof_end();
    ctx->prof_begin("qe_spill_scatter_kernel (level 2)");
    QE_HIP(launch_spill_scatter(args.spill, sp.waves, 2, bits, bits1, blocks, (const unsigned long long *)mid_keys->ptr,
                                (const unsigned long long *)mid_vals->ptr, stride, cell1, m, cell, (unsigned long long *)out_keys->ptr,
                                (unsigned long long *)out_vals->ptr, stride, ctx->stream));
    ctx->prof_end();
    // (the middle copy goes back to the pool here: later users are ordered behind these kernels on the stream)
  } else {
    ctx->prof_begin("qe_spill_scatter_kernel");
    QE_HIP(launch_spill_scatter(args.spill, sp.waves, 1, bits, bits, blocks, nullptr, nullptr, 0, nullptr, m, cell,
                                (unsigned long long *)out_keys->ptr, (unsigned long long *)out_vals->ptr, stride, ctx->stream));
    ctx->prof_end();
  }
  PartArgs pa{};
  pa.keys = (const unsigned long long *)out_keys->ptr;
  pa.vals = out_vals ? (const unsigned long long *)out_vals->ptr : nullptr;
  pa.interleaved = onepass ? 1 : 0;
  pa.capacity = stride;
  pa.offsets = cell;
  pa.n_parts = 1 << bits;
  pa.stride = blocks;
  pa.total = m;
  for (int i = 0; i < args.n_accs; ++i) {
    pa.accs[i] = args.accs[i];
    pa.acc_col[i] = sp.acc_col[i];
  }
  pa.n_accs = args.n_accs;
  pa.n_inputs = (int)n_in;
  pa.table = args.table;
  pa.lds_cap = partitioned_lds_slots(args.n_accs);
  pa.lds_limit = pa.lds_cap - pa.lds_cap / 4;
  pa.part_bits = bits;
  pa.sub_bits = sp.sub_bits;
  pa.null_cols = args.spill.null_cols;
  pa.null_f64 = args.spill.null_f64;
  pa.ticket = total + 1;
  ctx->prof_begin("qe_aggregate_partitioned_kernel");
  QE_HIP(launch_aggregate_partitioned(pa, ctx->dev, ctx->stream));
  ctx->prof_end();
  // the buffers go back to the pool on return: any later use is ordered behind these kernels on the stream
}

// Aggregation state fed one batch at a time: the device group table and what is needed to turn it
// into the output batch.  AggregateExec::execute's loop body (physical_plan/mod.rs:298-374) is
// `update`, its output assembly (:375-488) is `finish`.
// Thrown by AggState::update when a batch needs null tracking that the table, set up from earlier
// batches whose redundant validity buffers were ignored, does not have: the caller starts over with
// `strict` set, which takes every validity buffer at face value.
struct NeedsStrictNullability {};
// Thrown by AggState::finish when an order-independent sum ran at a scale chosen without having seen every value and
// a value turned out not to fit it: the caller starts over with `fix_exact` set, which finds every batch's scale by a
// pass of its own before the batch is folded in.
struct NeedsExactScales {};
// Rows of the first batch an order-independent sum looks at to choose its scale when it does not look at all of them.
constexpr int64_t FIX_PROBE_ROWS = (int64_t)1 << 22;
constexpr int FIX_PROBE_MARGIN = 8;  // bits of headroom above what those rows need

static Batch batch_prefix(const Batch &b, int64_t rows) {
  Batch p = b;
  p.rows = rows;
  for (auto &c : p.cols) {
    c.length = rows;
    if (c.null_count > 0) c.null_count = -1;
    if (type_width(c.type)) c.values_bytes = rows * type_width(c.type);
  }
  return p;
}

// Thrown by AggState::finish when a first pass that partitions its rows itself met a (partition, workgroup) stretch
// too small for its rows - keys far from evenly spread, which the rows it was chosen by did not show: the caller starts
// over with `direct_off` set, every batch through the table-and-log first pass.
struct NeedsLogPath {};
constexpr int64_t DIRECT_PROBE_ROWS = (int64_t)1 << 20;  // rows of the first large batch that choose between the two first passes

struct AggState {
  Context *ctx = nullptr;
  bool strict = false;
  std::vector<ExprPtr> groups;  // over the columns `outputs` describes (or the batch's own, when empty)
  std::vector<AggExpr> aggs;
  int mode = QE_AGG_MODE_SINGLE;
  int64_t cap_hint = 0;

  uint64_t cap = 0;
  bool have_table = false;
  AggSpec spec;
  AggTable table{};
  BufferPtr table_buf;
  std::vector<std::vector<Utf8KeySource>> key_sources;  // per gathered key column: its source in every batch
  // Groups identified by a hash of their key tuple: what it takes to compare every row's key with its group's
  // representative's once all batches are in (verify_hashed_keys) - per batch the columns as the kernel saw them
  // (temporaries of computed keys included), the predicate, the hash expression, the first row's number.
  struct VerifyItem {
    Batch in;
    ExprPtr pred, hash;
    int64_t row_base;
  };
  std::vector<VerifyItem> verify_items;
  int64_t row_base = 0;
  int deferred_finishes = 0;
  BufferPtr deferred_block;  // output block of the deferred mode, cleared by the table's init launch
  // More aggregates than one kernel has accumulators for: the list is split, every part aggregates the same rows
  // by the same keys plus a hidden MIN(row number), and the parts' outputs are put side by side by that number
  // (the reference keeps one accumulator object per group and aggregate, physical_plan/mod.rs:340-372: no limit).
  std::vector<std::unique_ptr<AggState>> parts;
  bool emit_row_number = false;  // this is such a part

  // Order-independent Float64 sums (QE_FLAG_DETERMINISTIC_SUM): the scale every such aggregate's limbs are at (INT_MIN:
  // no value seen yet), the rows folded in so far, and - for the aggregate that finds a batch's scales - no deferral.
  std::vector<int> fix_scale;
  int64_t fix_rows = 0;
  bool no_defer = false;
  bool fix_wanted() const { return (ctx->flags & QE_FLAG_DETERMINISTIC_SUM) != 0; }
  // How the scales are found.  Exact: a pass over every batch before it is folded in (one more read of the value
  // columns and a wait per batch).  Otherwise the scale comes from the first rows of the first batch that has any, with
  // some headroom, every later batch runs at it, and a value that does not fit sets `fix_overflow` on the device - read
  // with the group count at the end, upon which the aggregation starts over with exact scales.  Either way the scale
  // is a function of the data alone, so the sums' bits are too.  An aggregate that must not wait for the device
  // (deferred group count) cannot start over: exact scales.
  bool fix_exact = false;
  BufferPtr fix_overflow;
  bool fix_optimistic() const { return fix_wanted() && !fix_exact && !(ctx->flags & QE_FLAG_DEFERRED_GROUP_COUNT); }

  int acc_weight(const AggExpr &a) const {  // accumulators an aggregate may take
    const bool fix = fix_wanted();
    // (with very many groups a SUM, MIN or MAX over an input that turns out nullable brings a COUNT accumulator along:
    // build_agg_spec's explicit_counts)
    const int plain = effective_hint() >= spill_min_group_hint() && a.kind != QE_AGG_COUNT ? 2 : 1;
    return a.kind == QE_AGG_AVG ? (fix ? FIX_ACCS + 1 : 2) : (fix && a.kind == QE_AGG_SUM ? FIX_ACCS : plain);
  }
  static constexpr int PART_ACC_BUDGET = VM_MAX_ACCS - 2;                           // the row number, and one to spare

  // What the kernels' choices (LDS table size, high-cardinality mode) go by: the caller's group-count hint, or -
  // once a table has overflowed and the aggregation starts over with a larger one - what that taught.
  int64_t last_start_hint = 0, learned_hint = 0;
  int64_t effective_hint() const { return std::max(cap_hint, learned_hint); }

  // High-cardinality mode: the logs of the batches seen since the last partition passes (SpillGroup), and how many
  // batches the caller expects (sizes the group's cell table; 0 = not known).
  SpillGroup spill_group;
  int expected_batches = 0;
  // Which first pass the many-groups mode runs: 0 not chosen yet, 1 the one with a table and a log, 2 the one that
  // partitions its rows itself.  Chosen by the share of the first large batch's first rows that miss the table
  // (update); `direct_off`: never the second (an aggregation that started over after NeedsLogPath).
  int direct_mode = 0;
  bool direct_off = false;
  BufferPtr direct_overflow;  // set on the device when a stretch was too small; read with the group count at the end

  void spill_attach(SpillPlan &sp, AggArgs &args) {
    SpillGroup &g = spill_group;
    if (!g.items.empty() && (g.bits != sp.bits || g.sub_bits != sp.sub_bits || g.blocks != sp.blocks || g.items[0].spill.null_cols != args.spill.null_cols || g.n_accs != args.n_accs || (int)g.items.size() >= g.slots ||
                             g.log_rows + args.spill.capacity >= (1ull << 31)))
      spill_flush();
    if (g.items.empty()) {
      g.bits = sp.bits;
      g.sub_bits = sp.sub_bits;
      g.blocks = sp.blocks;
      g.n_accs = args.n_accs;
      g.slots = std::max(1, std::min(expected_batches > 0 ? expected_batches : 4, 8));
      const size_t cells = spill_cells(g.bits, g.blocks * g.slots);
      // the cell table (a slot no launch fills holds no rows), the group's control words and, for the first pass that
      // partitions directly, its overflow word: one block under one fill
      const size_t cell_bytes = ((cells + 1) * 4 + 255) & ~(size_t)255;
      g.cell_buf = device_buffer(ctx, cell_bytes + 512);
      g.scan_buf = device_buffer(ctx, scan_scratch_elems((int64_t)cells) * 4 + 64);
      g.ctrl = sub_buffer(g.cell_buf, cell_bytes, 256);
      QE_HIP(hipMemsetAsync(g.cell_buf->ptr, 0, cell_bytes + 512, ctx->stream));
      if (sp.direct && !direct_overflow) direct_overflow = sub_buffer(g.cell_buf, cell_bytes + 256, 256);
      g.log_rows = 0;
    }
    if (sp.direct) {
      if (!direct_overflow) {
        direct_overflow = device_buffer(ctx, 256);
        QE_HIP(hipMemsetAsync(direct_overflow->ptr, 0, 8, ctx->stream));
      }
      args.spill.counts = (uint32_t *)direct_overflow->ptr;
      args.cells_total = (unsigned long long *)g.ctrl->ptr + 3;
    } else {
      args.cells = (int32_t *)g.cell_buf->ptr + g.items.size() * (size_t)g.blocks;
      args.cells_stride = (uint32_t)(g.slots * g.blocks);
      args.cells_total = (unsigned long long *)g.ctrl->ptr;
      args.cell_bits = sp.bits;
    }
    sp.grouped = true;
  }

  void spill_push(const SpillPlan &sp, const AggArgs &args) {
    SpillGroup &g = spill_group;
    if (!sp.counted) fail(QE_ERR_INTERNAL, "a first pass that was to count its logged rows per partition did not");
    g.items.push_back(SpillGroup::Item{sp.keys, sp.vals, sp.direct ? sp.cell_buf : sp.counts, args.spill, sp.waves, sp.direct});
    g.log_rows += args.spill.capacity;
    for (int i = 0; i < args.n_accs; ++i) {
      g.accs[i] = args.accs[i];
      g.acc_col[i] = sp.acc_col[i];
    }
    g.table = args.table;
    // (the logs are sized for every row of their batch: no more of them at a time than a sixth of the card's memory holds)
    if ((int)g.items.size() >= g.slots || g.log_rows * 16 > ctx->dev.total_mem / 6) spill_flush();
  }

  // The partition passes over the group's batches: for the logs one scan of the cell table, one wait for the number of
  // rows and a scatter per log into one partitioned copy; then one partition-wise aggregate over that copy and the
  // stretches of the batches that were partitioned directly (which need no pass and no wait here).
  void spill_flush() {
    SpillGroup &g = spill_group;
    if (g.items.empty()) return;
    ctx->bind();
    ensure_counters(ctx);
    const int stride = g.slots * g.blocks;
    const size_t cells = spill_cells(g.bits, stride);
    int32_t *cell = (int32_t *)g.cell_buf->ptr;
    unsigned long long *total = (unsigned long long *)g.ctrl->ptr;
    int n_logs = 0, n_direct = 0;
    for (auto &it : g.items) (it.direct ? n_direct : n_logs)++;
    int64_t m = 0;
    BufferPtr out;
    static const bool verbose = getenv("QE_SPILL_VERBOSE") != nullptr;
    if (n_logs > 0) {
      QE_HIP(launch_exclusive_scan_i32(cell, (int64_t)cells, (int32_t *)g.scan_buf->ptr, ctx->stream));
      QE_HIP(hipMemcpyAsync(ctx->host_counters + 2, total, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
      ctx->sync();
      m = (int64_t)ctx->host_counters[2];
      if (m >= ((int64_t)1 << 31)) fail(QE_ERR_INTERNAL, "the logs of a group of batches hold more rows than the cell table can count");
    }
    if (verbose)
      fprintf(stderr, "qe_gpu: high-cardinality mode: %lld rows logged by %d launches, %d launches partitioned directly, %d partitions\n", (long long)m, n_logs,
              n_direct, 1 << g.bits);
    const uint64_t rows = ((uint64_t)m + 63) & ~63ull;
    if (m > 0) {
      out = device_buffer(ctx, rows * 16 + 256);
      for (size_t l = 0; l < g.items.size(); ++l) {
        if (g.items[l].direct) continue;
        ctx->prof_begin("qe_spill_scatter_onepass_kernel");
        QE_HIP(launch_spill_scatter_onepass(g.items[l].spill, g.items[l].waves, g.bits, g.blocks, cell + l * (size_t)g.blocks, stride, out->ptr, ctx->stream));
        ctx->prof_end();
      }
    }
    if (m > 0 || n_direct > 0) {
      PartArgs pa{};
      pa.keys = m > 0 ? (const unsigned long long *)out->ptr : nullptr;
      pa.vals = nullptr;
      pa.interleaved = 1;
      pa.capacity = rows;
      pa.offsets = cell;
      pa.n_parts = 1 << g.bits;
      pa.stride = stride;
      pa.total = m;
      for (int i = 0; i < g.n_accs; ++i) {
        pa.accs[i] = g.accs[i];
        pa.acc_col[i] = g.acc_col[i];
      }
      pa.n_accs = g.n_accs;
      pa.n_inputs = (int)g.items[0].spill.n_inputs;
      pa.table = g.table;
      pa.lds_cap = partitioned_lds_slots(g.n_accs);
      pa.lds_limit = pa.lds_cap - pa.lds_cap / 4;
      pa.part_bits = g.bits;
      pa.sub_bits = g.sub_bits;
      pa.null_cols = g.items[0].spill.null_cols;
      pa.null_f64 = g.items[0].spill.null_f64;
      pa.ticket = total + 1;
      pa.direct_blocks = g.blocks;
      for (auto &it : g.items)
        if (it.direct) {
          if (pa.n_direct >= PART_MAX_DIRECT) fail(QE_ERR_INTERNAL, "more directly partitioned batches in a group than the partition-wise aggregate takes");
          pa.direct[pa.n_direct++] = PartDirect{it.spill.keys, (const int32_t *)it.counts->ptr, it.spill.direct_cap, (uint32_t)g.blocks};
        }
      ctx->prof_begin("qe_aggregate_partitioned_kernel");
      QE_HIP(launch_aggregate_partitioned(pa, ctx->dev, ctx->stream));
      ctx->prof_end();
    }
    // (the buffers go back to the pool: any later use is ordered behind these kernels on the stream)
    g.items.clear();
    g.cell_buf.reset();
    g.scan_buf.reset();
    g.ctrl.reset();
    g.log_rows = 0;
  }

  // Chooses between the two first passes of the many-groups mode: the first rows of the batch go through the one with
  // a table and a log - results thrown away: no flush of the tables, a log that the batch's real launch overwrites -
  // and the share of the selected rows that missed the table decides.  `args` is the batch's launch as planned for that
  // first pass (log attached).
  void choose_first_pass(const AggArgs &args, const SpillPlan &sp) {
    direct_mode = 1;
    AggArgs probe = args;
    static const int64_t probe_rows = [] {
      const char *e = getenv("QE_AGGREGATE_DIRECT_PROBE_ROWS");
      return e && atoll(e) > 0 ? (int64_t)atoll(e) : DIRECT_PROBE_ROWS;
    }();
    probe.n_rows = std::min<int64_t>(args.n_rows, probe_rows);
    probe.n_more = 0;
    probe.tune |= 4u;  // the workgroups' tables are not flushed
    const size_t cells = spill_cells(sp.bits, sp.blocks);
    BufferPtr cell_buf = device_buffer(ctx, (cells + 1) * 4 + 64), ctrl = device_buffer(ctx, 256);
    QE_HIP(hipMemsetAsync(ctrl->ptr, 0, 8 * sizeof(unsigned long long), ctx->stream));
    probe.cells = (int32_t *)cell_buf->ptr;
    probe.cells_stride = (uint32_t)sp.blocks;
    probe.cells_total = (unsigned long long *)ctrl->ptr;
    probe.cell_bits = sp.bits;
    bool counted = false;
    const char *kname = "qe_aggregate_kernel";
    ctx->prof_begin(kname, probe.n_rows);
    QE_HIP(launch_aggregate(probe, ctx->dev, ctx->stream, &kname, &counted));
    ctx->prof_end(kname);
    QE_HIP(hipMemcpyAsync(ctx->host_counters + 24, ctrl->ptr, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (!counted) return;
    const double logged = (double)ctx->host_counters[24], selected = (double)ctx->host_counters[26];
    static const double threshold = [] {
      const char *e = getenv("QE_AGGREGATE_DIRECT_MISS_SHARE");
      return e && atof(e) > 0 ? atof(e) : 0.85;
    }();
    if (selected > 0 && logged >= threshold * selected) direct_mode = 2;
    static const bool verbose = getenv("QE_SPILL_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "qe_gpu: high-cardinality mode: %.0f of %.0f selected rows logged: %s first pass\n", logged, selected, direct_mode == 2 ? "partitioning" : "logging");
  }

  // Fresh table for about `hint` groups.
  void start(int64_t hint) {
    if (last_start_hint > 0 && hint > last_start_hint) learned_hint = std::max(learned_hint, hint);
    last_start_hint = hint;
    deferred_finishes = 0;
    deferred_block.reset();
    cap = 1024;
    while (cap < (uint64_t)hint * 2) cap <<= 1;
    if (cap > (1ull << 30)) fail(QE_ERR_AGG_CAPACITY, "group table would exceed 2^30 slots");
    parts.clear();
    int weight = 0;
    for (auto &a : aggs) weight += acc_weight(a);
    if (!emit_row_number && weight > PART_ACC_BUDGET) {
      for (size_t i = 0; i < aggs.size();) {
        auto p = std::make_unique<AggState>();
        p->ctx = ctx;
        p->strict = strict;
        p->fix_exact = fix_exact;
        p->groups = groups;
        p->mode = mode;
        p->cap_hint = effective_hint();
        p->expected_batches = expected_batches;
        p->direct_off = direct_off;
        p->emit_row_number = true;
        for (int w = 0; i < aggs.size() && (w == 0 || w + acc_weight(aggs[i]) <= PART_ACC_BUDGET); ++i) {
          w += acc_weight(aggs[i]);
          p->aggs.push_back(aggs[i]);
        }
        p->start(hint);
        parts.push_back(std::move(p));
      }
    }
    have_table = false;
    spill_group = SpillGroup();  // (logs of an aggregation that starts over are dropped with its table)
    direct_mode = 0;
    direct_overflow.reset();
    spec = AggSpec();
    table = AggTable{};
    table_buf.reset();
    key_sources.clear();
    verify_items.clear();
    row_base = 0;
    fix_scale.clear();
    fix_rows = 0;
    fix_overflow.reset();
  }

  // The scale each order-independent sum needs for this batch: the smallest multiple of 8 with |value| < 2^scale for
  // every finite input the predicate lets through (INT_MIN when there is none), found by a keyless MAX over the
  // inputs' exponent fields.
  std::vector<int> batch_scales(const Batch &in_raw, const std::vector<ExprPtr> &outputs, const ExprPtr &pred, const std::vector<char> &is_fix) {
    std::vector<int> need(aggs.size(), INT_MIN);
    if (in_raw.rows == 0) return need;
    AggState probe;
    probe.ctx = ctx;
    probe.strict = strict;
    probe.no_defer = true;
    probe.cap_hint = 1;
    std::vector<size_t> owner;
    for (size_t i = 0; i < aggs.size(); ++i)
      if (is_fix[i]) {
        AggExpr m;
        m.kind = QE_AGG_MAX;
        m.input = make_fix(FIX_EXPONENT, mode == QE_AGG_MODE_FINAL ? aggs[i].input : make_to_f64(aggs[i].input), 0);
        probe.aggs.push_back(m);
        owner.push_back(i);
      }
    probe.start(1);
    probe.update(in_raw, outputs, pred);
    Batch r;
    try {
      if (!probe.finish(r)) fail(QE_ERR_INTERNAL, "the scale pass of an order-independent sum overflowed its table");
    } catch (const Failure &f) {
      if (f.code == QE_ERR_EMPTY_HASHMAP_FOR_AGGREGATE) return need;  // the predicate let nothing through
      throw;
    }
    if (r.rows != 1 || r.cols.size() != owner.size()) fail(QE_ERR_INTERNAL, "the scale pass of an order-independent sum found more than one group");
    std::vector<int64_t> e(owner.size(), 0);
    std::vector<uint8_t> v(owner.size(), 1);
    for (size_t j = 0; j < owner.size(); ++j) {
      QE_HIP(hipMemcpyAsync(&e[j], r.cols[j].values_ptr(), 8, hipMemcpyDeviceToHost, ctx->stream));
      if (r.cols[j].validity) QE_HIP(hipMemcpyAsync(&v[j], r.cols[j].validity_ptr(), 1, hipMemcpyDeviceToHost, ctx->stream));
    }
    ctx->sync();
    for (size_t j = 0; j < owner.size(); ++j)
      if (v[j] & 1u) need[owner[j]] = (int)((std::max<int64_t>(e[j], 1) - 1022 + 7) & ~(int64_t)7);
    return need;
  }

  // Folds one batch in.  `outputs` (may be empty = the batch's columns) are the expressions the
  // group / aggregate expressions' column indices refer to; `pred` selects rows.
  // `more` (optional): the batches that follow, which a many-groups first pass takes in the same launch when it can
  // (its workgroups' tables are then filled and flushed once for all of them); otherwise they are folded in one by one.
  void update(const Batch &in_raw, const std::vector<ExprPtr> &outputs, const ExprPtr &pred, const std::vector<Batch> *more = nullptr) {
    if (!parts.empty()) {
      for (auto &p : parts) {
        p->update(in_raw, outputs, pred);
        if (more)
          for (auto &e : *more) p->update(e, outputs, pred);
      }
      return;
    }
    size_t more_taken = 0;  // batches of `more` that went into this batch's launch
    struct FoldRest {       // (whatever did not, on every way out that is not an exception)
      AggState *st;
      const std::vector<ExprPtr> &outputs;
      const ExprPtr &pred;
      const std::vector<Batch> *more;
      size_t &taken;
      bool armed = true;
      void run() {
        armed = false;
        if (more)
          for (size_t i = taken; i < more->size(); ++i) st->update((*more)[i], outputs, pred);
      }
    } rest{this, outputs, pred, more, more_taken};
    Batch in = strict ? in_raw : without_redundant_validity(in_raw);
    ctx->bind();
    ensure_counters(ctx);
    std::vector<ExprPtr> src_outs = outputs;
    if (src_outs.empty())
      for (size_t i = 0; i < in.cols.size(); ++i) src_outs.push_back(make_column((int)i));
    std::vector<ExprPtr> g2;
    std::vector<AggExpr> a2;
    std::vector<ExprPtr> a2_second;
    for (auto &g : groups) g2.push_back(expr_substitute(g, src_outs));
    for (auto &a : aggs) {
      AggExpr x;
      x.kind = a.kind;
      x.input = a.input ? expr_substitute(a.input, src_outs) : nullptr;
      a2.push_back(x);
      ExprPtr second;
      if (mode == QE_AGG_MODE_FINAL && a.kind == QE_AGG_AVG && a.input && a.input->kind == EK_COLUMN)
        second = expr_substitute(make_column(a.input->index + 1), src_outs);
      a2_second.push_back(second);
    }
    Schema schema = schema_of(in);
    ExprPtr pred_n = pred ? expr_normalize(pred, ctx->flags) : nullptr;
    // order-independent sums: this batch's scales first (a pass of its own), the table's limbs moved up where needed
    std::vector<char> is_fix(aggs.size(), 0);
    bool any_fix = false;
    if (fix_wanted())
      for (size_t i = 0; i < a2.size(); ++i)
        if (a2[i].input && (a2[i].kind == QE_AGG_AVG || (a2[i].kind == QE_AGG_SUM && expr_type(expr_normalize(a2[i].input, ctx->flags), schema) == QE_TYPE_FLOAT64))) {
          const int t = expr_type(expr_normalize(a2[i].input, ctx->flags), schema);
          if (!numeric(t)) continue;  // (build_agg_spec reports it)
          is_fix[i] = 1;
          any_fix = true;
        }
    std::vector<int> scales_now;
    if (any_fix) {
      fix_rows += in.rows;
      if (fix_rows > ((int64_t)1 << 31)) fail(QE_ERR_INVALID_ARGUMENT, "QE_FLAG_DETERMINISTIC_SUM: more than 2^31 rows in one aggregate");
      if (fix_scale.empty()) fix_scale.assign(aggs.size(), INT_MIN);
      std::vector<int> need(aggs.size(), INT_MIN);
      if (!fix_optimistic()) {
        need = batch_scales(in_raw, outputs, pred, is_fix);
      } else {
        if (!fix_overflow) {
          fix_overflow = device_buffer(ctx, 256);
          QE_HIP(hipMemsetAsync(fix_overflow->ptr, 0, 8, ctx->stream));
        }
        auto unknown = [&](const std::vector<int> &found) {
          for (size_t i = 0; i < aggs.size(); ++i)
            if (is_fix[i] && fix_scale[i] == INT_MIN && found[i] == INT_MIN) return true;
          return false;
        };
        if (unknown(need)) {  // (some sum has no scale yet)
          std::vector<int> found = in_raw.rows > FIX_PROBE_ROWS ? batch_scales(batch_prefix(in_raw, FIX_PROBE_ROWS), outputs, pred, is_fix)
                                                                : batch_scales(in_raw, outputs, pred, is_fix);
          if (in_raw.rows > FIX_PROBE_ROWS && unknown(found)) found = batch_scales(in_raw, outputs, pred, is_fix);  // the predicate let none of the first rows through
          for (size_t i = 0; i < aggs.size(); ++i)
            if (is_fix[i] && fix_scale[i] == INT_MIN && found[i] != INT_MIN) need[i] = found[i] + FIX_PROBE_MARGIN;
        }
      }
      std::vector<int> moved;
      for (size_t i = 0; i < aggs.size(); ++i) {
        if (!is_fix[i] || need[i] <= fix_scale[i]) continue;
        if (have_table && fix_scale[i] != INT_MIN) {
          const int acc = spec.outs[(size_t)spec.agg_first_out[i]].a;
          if (std::find(moved.begin(), moved.end(), acc) == moved.end()) {
            moved.push_back(acc);
            ctx->prof_begin("qe_fix_rescale_kernel");
            QE_HIP(launch_fix_rescale(table, acc, need[i] - fix_scale[i], ctx->stream));
            ctx->prof_end();
          }
        }
        fix_scale[i] = need[i];
      }
      for (size_t i = 0; i < aggs.size(); ++i) scales_now.push_back(fix_scale[i] == INT_MIN ? 0 : fix_scale[i]);
    }
    bool had_temporaries = false;
    {
      // predicate, keys and accumulator inputs share one program: whatever exceeds it is evaluated beforehand
      std::vector<ExprPtr *> all;
      if (pred_n) all.push_back(&pred_n);
      for (auto &g : g2) all.push_back(&g);
      for (size_t i = 0; i < a2.size(); ++i) {
        AggExpr &a = a2[i];
        if (a.input) {
          all.push_back(&a.input);
          if (a.kind == QE_AGG_AVG) all.push_back(&a.input);  // its sum and its count each evaluate the input
          if (is_fix[i])
            for (int k = 1; k < FIX_ACCS; ++k) all.push_back(&a.input);  // and so does every limb of an order-independent sum
        }
      }
      for (ExprPtr *p : all) {
        *p = expr_normalize(*p, ctx->flags);
        (void)expr_type(*p, schema);
      }
      ExprFitter fitter{ctx, in, schema};
      for (ExprPtr *p : all) *p = fitter.shrink(*p);
      std::vector<int> key_types;
      for (auto &g : g2) key_types.push_back(expr_type(g, fitter.schema));
      if (key_tuple_hashed(key_types))  // key columns are gathered from a representative row: computed keys become columns
        for (auto &g : g2)
          if (g->kind != EK_COLUMN) g = fitter.materialise(g);
      fitter.fit_together(all, 12 + 5 * (int)g2.size() + (int)a2.size(), 4);
      if (fitter.temporaries) schema = fitter.schema;
      had_temporaries = fitter.temporaries != 0;
    }
    static const bool spill_nullable = [] {
      const char *e = getenv("QE_SPILL_NULLABLE");  // 0: nullable accumulator inputs keep an aggregation out of the many-groups mode
      return e ? atoi(e) != 0 : true;
    }();
    AggSpec bspec = build_agg_spec(ctx->flags, schema, g2, a2, a2_second, mode, row_base, emit_row_number, any_fix ? &scales_now : nullptr,
                                   any_fix && fix_overflow ? (uint64_t)(uintptr_t)fix_overflow->ptr : 0,
                                   spill_nullable && spill_enabled() && effective_hint() >= spill_min_group_hint());
    key_sources.resize(bspec.key_gathers.size());
    for (size_t i = 0; i < bspec.key_gathers.size(); ++i) key_sources[i].push_back(Utf8KeySource{in.cols[bspec.key_gathers[i].col], row_base});
    if (!bspec.key_gathers.empty() && !(ctx->flags & QE_FLAG_HASH_ONLY_GROUP_IDENTITY))
      verify_items.push_back(VerifyItem{in, pred_n, bspec.key_exprs[0], row_base});
    row_base += in.rows;
    if (!have_table) {
      spec = bspec;
      int na = (int)spec.accs.size();
      table.cap = cap;
      table.stride = (cap + 1 + 63) & ~63ull;
      size_t words = table.stride * (2 + 2 * (size_t)na) + 8;
      table_buf = device_buffer(ctx, words * 8);
      unsigned long long *base = (unsigned long long *)table_buf->ptr;
      table.keys = base;
      table.rowcnt = base + table.stride;
      table.acc = base + 2 * table.stride;
      table.nn = table.acc + (size_t)na * table.stride;
      table.ctrl = table.nn + (size_t)na * table.stride;
      std::vector<VmAcc> va;
      for (auto &d : spec.accs) va.push_back(d.acc);
      void *zero = nullptr;
      size_t zero_bytes = 0;
      if (deferred_wanted()) {
        std::vector<size_t> at;
        zero_bytes = deferred_layout(at);
        deferred_block = device_buffer(ctx, zero_bytes + 256);
        zero = deferred_block->ptr;
      }
      ctx->prof_begin("qe_table_init_kernel");
      QE_HIP(launch_table_init(table, va.data(), na, ctx->stream, zero, zero_bytes));
      ctx->prof_end();
      have_table = true;
    } else {
      bool same = bspec.accs.size() == spec.accs.size() && bspec.keys.size() == spec.keys.size();
      for (size_t i = 0; same && i < spec.accs.size(); ++i)
        same = bspec.accs[i].acc.op == spec.accs[i].acc.op && bspec.accs[i].acc.in_type == spec.accs[i].acc.in_type;
      for (size_t i = 0; same && i < spec.keys.size(); ++i) same = bspec.keys[i].type == spec.keys[i].type;
      if (!same) {
        // nullability shapes the accumulator list (COUNT(x) of a column without nulls is the row count)
        if (!strict) throw NeedsStrictNullability{};
        fail(QE_ERR_INVALID_ARGUMENT, "batches of one scan must share a schema");
      }
      for (size_t i = 0; same && i < spec.outs.size() && i < bspec.outs.size(); ++i) spec.outs[i].scale = bspec.outs[i].scale;
      // a later batch may be nullable where the first was not: keep the stricter tracking
      for (size_t i = 0; i < spec.accs.size(); ++i)
        if (bspec.accs[i].acc.track_nn && !spec.accs[i].acc.track_nn) {
          if (!strict) throw NeedsStrictNullability{};
          fail(QE_ERR_INVALID_ARGUMENT, "batches of one scan must share nullability");
        }
    }
    if (in.rows == 0) {
      rest.run();
      return;
    }
    Compiler comp(ctx, &in);
    AggArgs args{};
    lower_aggregate(comp, schema, pred_n, bspec, spec, args);
    args.table = table;
    args.n_rows = in.rows;
    // The batches that follow, in this launch: for the first passes of the many-groups mode, when the program reads
    // nothing but its cached columns (they are all a further batch brings) and the batch is of the same shape -
    // lowered against it, the same kernel signature and literals.  Row numbers (hashed keys' representatives, the
    // parts of a split aggregation) count from each batch's own base: not for those.
    int64_t rows_in_launch = in.rows;
    static const bool merge_on = [] {
      const char *e = getenv("QE_SPILL_MERGE_BATCHES");  // 0: a launch per batch
      return e ? atoi(e) != 0 : true;
    }();
    // (Only there: with few groups a launch's fixed costs are small - one launch of the headline plan over four batches
    // measured 1 to 3 % faster than four, within the spread from run to run, and the batch loop cost the kernel ten
    // registers.)
    if (more && merge_on && effective_hint() >= spill_min_group_hint() && bspec.key_gathers.empty() && !emit_row_number && !any_fix && !had_temporaries &&
        comp.keepalive.empty() && comp.n_cols == (int)comp.prog.cache_n && in.rows >= (int64_t)ctx->dev.cus * VM_WAVE_ROWS * (AGG_WIDE_THREADS / 64)) {
      const std::vector<uint32_t> sig = signature(args);
      while (more_taken < more->size() && args.n_more < AGG_MORE_BATCHES) {
        const Batch &e_raw = (*more)[more_taken];
        if (e_raw.rows <= 0 || e_raw.cols.size() != in_raw.cols.size() || rows_in_launch + e_raw.rows > ((int64_t)1 << 30)) break;
        Batch e = strict ? e_raw : without_redundant_validity(e_raw);
        bool same = schema_of(e).size() == schema.size();
        for (size_t i = 0; same && i < schema.size(); ++i) same = e.cols[i].type == in.cols[i].type;
        if (!same) break;
        Compiler comp_e(ctx, &e);
        AggArgs args_e{};
        lower_aggregate(comp_e, schema, pred_n, bspec, spec, args_e);
        if (!comp_e.keepalive.empty() || comp_e.n_cols != (int)comp_e.prog.cache_n || comp_e.prog.cache_n != comp.prog.cache_n || signature(args_e) != sig ||
            std::memcmp(args_e.prog.lits, args.prog.lits, sizeof(args.prog.lits)) != 0)
          break;
        for (uint32_t i = 0; i < comp_e.prog.cache_n; ++i) args.more_cols[args.n_more][i] = args_e.prog.cols[args_e.prog.cache_col[i]];
        args.more_rows[args.n_more] = e.rows;
        ++args.n_more;
        rows_in_launch += e.rows;
        ++more_taken;
      }
    }
    args.lds_cap = aggregate_lds_slots(args, ctx->dev, effective_hint(), false);
    args.lds_limit = args.lds_cap - args.lds_cap / 4;
    args.n_rows = rows_in_launch;  // (what the log, the stretches and the launch are sized by)
    SpillPlan sp = plan_spill(ctx, schema, bspec, args, effective_hint(), true, direct_mode == 2);
    if (args.n_more > 0 && !(sp.on && sp.groupable)) {
      // not a first pass of the many-groups mode after all (or one whose log is partitioned batch by batch): the batches one by one
      more_taken -= (size_t)args.n_more;
      args.n_more = 0;
      rows_in_launch = in.rows;
      args.n_rows = in.rows;
      sp = plan_spill(ctx, schema, bspec, args, effective_hint(), true, direct_mode == 2);
    }
    if (sp.on && sp.groupable && !sp.direct && direct_mode == 0 && !direct_off && in.rows >= 8 * DIRECT_PROBE_ROWS && sp.blocks == ctx->dev.cus &&
        aggregate_direct_available(args, ctx->dev, sp.bits)) {
      choose_first_pass(args, sp);
      if (direct_mode == 2) {
        AggArgs again = args;
        again.spill = AggSpill{};
        again.block_threads = 0;
        again.tune = 0;
        again.lds_cap = aggregate_lds_slots(again, ctx->dev, effective_hint(), false);
        again.lds_limit = again.lds_cap - again.lds_cap / 4;
        SpillPlan direct = plan_spill(ctx, schema, bspec, again, effective_hint(), true, true);
        if (direct.direct) {  // (else: the log path as planned)
          args = again;
          sp = direct;
        }
      }
    }
    if (sp.on && sp.groupable) spill_attach(sp, args);
    else spill_flush();  // (a launch that goes another way: what is pending first, so that the table sees the batches in order)
    args.n_rows = in.rows;  // (the first batch's; the others' in more_rows)
    for (int i = 0; i < args.n_more; ++i) row_base += args.more_rows[i];
    const char *kname = "qe_aggregate_kernel";
    ctx->prof_begin(kname, rows_in_launch);
    if (sp.direct) {
      QE_HIP(launch_aggregate_direct(args, ctx->dev, ctx->stream, &kname));
      sp.counted = true;
    } else {
      QE_HIP(launch_aggregate(args, ctx->dev, ctx->stream, &kname, &sp.counted));
    }
    ctx->prof_end(kname);
    if (sp.on) {
      if (sp.grouped) spill_push(sp, args);
      else aggregate_spilled(ctx, sp, args, effective_hint());
    }
    if (!comp.keepalive.empty()) ctx->sync();
    rest.run();
  }

  // The output without a word from the device: one row per table slot, the rows past the group count
  // all zero (values and validity), every column a slice of one zeroed block.  The control words
  // travel to a pinned slot that the next synchronise examines.
  // Layout of the deferred output block: per output column the values slice, then (nullable outputs) the bitmap.
  size_t deferred_layout(std::vector<size_t> &at) const {
    const int64_t capacity = (int64_t)table.cap + 1;
    size_t total = 0;
    for (auto &f : spec.outs) {
      at.push_back(total);
      total += (f.type == QE_TYPE_BOOLEAN ? bitmap_bytes(capacity) : (((size_t)capacity * type_width(f.type) + 16 + 255) & ~(size_t)255));
      at.push_back(total);
      if (f.null_mode != NULL_NEVER) total += (bitmap_bytes(capacity) + 255) & ~(size_t)255;
    }
    return total;
  }
  bool deferred_wanted() const {
    return table.cap + 1 <= (1u << 16) && (ctx->flags & QE_FLAG_DEFERRED_GROUP_COUNT) && spec.key_gathers.empty() && !emit_row_number && !no_defer;
  }

  bool finish_deferred(Batch &result) {
    const int64_t capacity = (int64_t)table.cap + 1;
    std::vector<size_t> at;
    const size_t total = deferred_layout(at);
    BufferPtr block = deferred_block;  // cleared together with the table, if this is the first result taken
    deferred_block.reset();
    if (!block) {
      block = device_buffer(ctx, total + 256);
      QE_HIP(hipMemsetAsync(block->ptr, 0, total, ctx->stream));
    }
    FinalizeArgs fa{};
    fa.table = table;
    fa.n_outs = (int)spec.outs.size();
    result = Batch();
    result.ctx = ctx;
    result.rows = capacity;
    for (size_t i = 0; i < spec.outs.size(); ++i) {
      FinOut f = spec.outs[i];
      Column c;
      c.type = f.type;
      c.space = QE_MEM_DEVICE;
      c.length = capacity;
      c.values_bytes = f.type == QE_TYPE_BOOLEAN ? (capacity + 7) / 8 : capacity * type_width(f.type);
      c.values = sub_buffer(block, at[2 * i], (size_t)(at[2 * i + 1] - at[2 * i]));
      if (f.null_mode != NULL_NEVER) {
        c.validity = sub_buffer(block, at[2 * i + 1], bitmap_bytes(capacity));
        c.null_count = -1;
      }
      f.values = c.values->ptr;
      f.validity = c.validity ? (uint32_t *)c.validity->ptr : nullptr;
      fa.outs[i] = f;
      result.cols.push_back(c);
    }
    // the finalize kernel draws output positions from a counter of its own: the table's spare control word
    fa.out_count = table.ctrl + 2;
    fa.overflow_sink = ctx->deferred_sink();
    if (deferred_finishes++) QE_HIP(hipMemsetAsync(table.ctrl + 2, 0, sizeof(unsigned long long), ctx->stream));  // a running result came before
    ctx->prof_begin("qe_finalize_kernel");
    QE_HIP(launch_finalize(fa, ctx->stream));
    ctx->prof_end();
    return true;
  }

  // The output batch.  False when the table overflowed: the caller starts over with a larger one.
  // The parts' outputs side by side, in the first part's row order.
  bool finish_parts(Batch &result) {
    std::vector<Batch> rs(parts.size());
    for (size_t j = 0; j < parts.size(); ++j)
      if (!parts[j]->finish(rs[j])) return false;
    const int64_t n = rs[0].rows;
    auto row_numbers = [&](const Batch &b) {
      std::vector<int64_t> ids((size_t)n);
      QE_HIP(hipMemcpyAsync(ids.data(), b.cols.back().values_ptr(), (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
      ctx->sync();
      return ids;
    };
    result = Batch();
    result.ctx = ctx;
    result.rows = n;
    for (size_t c = 0; c + 1 < rs[0].cols.size(); ++c) result.cols.push_back(rs[0].cols[c]);
    std::unordered_map<int64_t, int64_t> at;
    const std::vector<int64_t> first = row_numbers(rs[0]);
    for (int64_t i = 0; i < n; ++i) at[first[(size_t)i]] = i;
    for (size_t j = 1; j < parts.size(); ++j) {
      if (rs[j].rows != n) fail(QE_ERR_INTERNAL, "the parts of a split aggregation found different groups");
      const std::vector<int64_t> ids = row_numbers(rs[j]);
      std::vector<int64_t> perm((size_t)n);
      for (int64_t i = 0; i < n; ++i) {
        auto it = at.find(ids[(size_t)i]);
        if (it == at.end()) fail(QE_ERR_INTERNAL, "the parts of a split aggregation found different groups");
        perm[(size_t)it->second] = i;
      }
      BufferPtr dev_perm = device_buffer(ctx, (size_t)n * 8 + 16);
      QE_HIP(hipMemcpyAsync(dev_perm->ptr, perm.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
      ctx->sync();  // `perm` (host) has been read
      for (size_t c = groups.size(); c + 1 < rs[j].cols.size(); ++c) {
        const Column &src = rs[j].cols[c];
        Column out = fixed_gather_rows(ctx, src.type, {Utf8KeySource{src, 0}}, (const int64_t *)dev_perm->ptr, n);
        if (src.validity) {
          Column bits;
          bits.type = QE_TYPE_BOOLEAN;
          bits.length = n;
          bits.values = src.validity;
          out.validity = fixed_gather_rows(ctx, QE_TYPE_BOOLEAN, {Utf8KeySource{bits, 0}}, (const int64_t *)dev_perm->ptr, n).values;
          out.null_count = -1;
        }
        result.cols.push_back(out);
      }
    }
    return true;
  }

  // Group identity by hash, checked: every selected row's key columns against those of the representative row of the
  // group its hash names.  Any difference means two distinct key tuples share one 64-bit hash and were merged - what the
  // reference does silently (physical_plan/mod.rs:310-358: groups are identified by the sum of column hashes alone).
  void verify_hashed_keys() {
    if (verify_items.empty() || spec.key_gathers.empty()) return;
    const int n_keys = (int)spec.key_gathers.size(), n_sources = (int)verify_items.size();
    if (n_keys > VERIFY_MAX_KEYS) fail(QE_ERR_INTERNAL, "too many key columns to verify");
    std::vector<KeyColumnSource> table((size_t)n_keys * n_sources);
    for (int k = 0; k < n_keys; ++k)
      for (int b = 0; b < n_sources; ++b) {
        const Utf8KeySource &src = key_sources[(size_t)k][(size_t)b];
        table[(size_t)k * n_sources + b] = KeyColumnSource{src.col.values_ptr(), src.col.offsets_ptr(), src.first_row};
      }
    BufferPtr dev_table = device_buffer(ctx, table.size() * sizeof(KeyColumnSource) + 16);
    QE_HIP(hipMemcpyAsync(dev_table->ptr, table.data(), table.size() * sizeof(KeyColumnSource), hipMemcpyHostToDevice, ctx->stream));
    ensure_counters(ctx);
    unsigned long long *mismatches = ctx->counters + 42;
    QE_HIP(hipMemsetAsync(mismatches, 0, sizeof(unsigned long long), ctx->stream));
    const uint32_t saved_flags = ctx->flags;
    ctx->flags &= ~(uint32_t)QE_FLAG_DEFERRED_ROW_COUNT;  // the launches below are sized by the row count
    try {
      for (int b = 0; b < n_sources; ++b) {
        const VerifyItem &it = verify_items[(size_t)b];
        if (it.in.rows == 0) continue;
        // the rows the aggregate kept, by number, with their tuple hashes (a row with a null key has a null hash)
        Batch t = run_pipeline_typed(ctx, it.in, {make_rowid(it.row_base), it.hash}, it.pred ? it.pred : make_lit_bool(true));
        VerifyArgs va{};
        va.table = table_args();
        va.rep_acc = spec.outs[(size_t)spec.key_gathers[0].out].a;
        va.row_ids = (const int64_t *)t.cols[0].values_ptr();
        va.hashes = (const unsigned long long *)t.cols[1].values_ptr();
        va.hash_validity = (const uint32_t *)t.cols[1].validity_ptr();
        va.m = t.rows;
        va.row_base = it.row_base;
        va.this_source = b;
        va.n_keys = n_keys;
        va.n_sources = n_sources;
        for (int k = 0; k < n_keys; ++k) va.key_type[k] = spec.key_gathers[(size_t)k].type;
        va.sources = (const KeyColumnSource *)dev_table->ptr;
        va.mismatches = mismatches;
        ctx->prof_begin("qe_verify_group_keys_kernel");
        QE_HIP(launch_verify_group_keys(va, ctx->stream));
        ctx->prof_end();
      }
    } catch (...) {
      ctx->flags = saved_flags;
      throw;
    }
    ctx->flags = saved_flags;
    QE_HIP(hipMemcpyAsync(ctx->host_counters + 42, mismatches, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();  // (`table` on the host has been read, the temporaries of the loop are done with)
    if (ctx->host_counters[42] != 0)
      fail(QE_ERR_INTERNAL, std::to_string(ctx->host_counters[42]) + " rows were grouped with a key that is not theirs: two distinct group keys share one 64-bit hash");
  }
  AggTable table_args() const { return table; }

  bool finish(Batch &result) {
    ctx->bind();
    if (!parts.empty()) return finish_parts(result);
    // empty input: `iter.next().ok_or(Error::EmptyHashmapForAggregate)` (physical_plan/mod.rs:377)
    if (!have_table) fail(QE_ERR_EMPTY_HASHMAP_FOR_AGGREGATE, "The hashmap for aggregation is empty.");
    spill_flush();  // the logged rows of the last batches
    // Small tables: the output columns are sized for every slot and the finalize pass is queued
    // before the group count is known, so that one trip to the host serves both.  Large tables
    // read the count first and allocate exactly.
    const bool eager = table.cap + 1 <= (1u << 16);
    if (deferred_wanted()) return finish_deferred(result);
    unsigned long long n_groups = 0, overflow = 0;
    // (the word an order-independent sum's kernels set when a value did not fit the scale travels with the control words)
    auto queue_scale_check = [&] {
      if (fix_overflow) QE_HIP(hipMemcpyAsync(ctx->host_counters + 20, fix_overflow->ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
      if (direct_overflow) QE_HIP(hipMemcpyAsync(ctx->host_counters + 21, direct_overflow->ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    };
    auto scale_check = [&] {
      if (direct_overflow && ctx->host_counters[21]) throw NeedsLogPath{};  // (rows are missing: nothing else about this result matters)
      if (fix_overflow && ctx->host_counters[20]) throw NeedsExactScales{};
    };
    if (!eager) {
      QE_HIP(hipMemcpyAsync(ctx->host_counters + 16, table.ctrl, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
      queue_scale_check();
      ctx->sync();
      scale_check();
      n_groups = ctx->host_counters[16];
      overflow = ctx->host_counters[17];
      if (overflow) return false;
      if (n_groups == 0) fail(QE_ERR_EMPTY_HASHMAP_FOR_AGGREGATE, "The hashmap for aggregation is empty.");
    }
    const int64_t capacity = eager ? (int64_t)table.cap + 1 : (int64_t)n_groups;
    // output assembly (physical_plan/mod.rs:375-488): key columns, then one column per aggregate
    FinalizeArgs fa{};
    fa.table = table;
    fa.n_outs = (int)spec.outs.size();
    result = Batch();
    result.ctx = ctx;
    // the nullable outputs' bitmaps and the counter the kernel draws output positions from: one block, one fill (a fill
    // per column is a small launch each, one after the other on the stream)
    const size_t out_bb = (bitmap_bytes(capacity) + 63) / 64 * 64;
    size_t nullable_outs = 0, next_bitmap = 0;
    for (auto &f : spec.outs) nullable_outs += f.null_mode != NULL_NEVER ? 1 : 0;
    BufferPtr zeroed = device_buffer(ctx, 64 + nullable_outs * out_bb);
    QE_HIP(hipMemsetAsync(zeroed->ptr, 0, 64 + nullable_outs * out_bb, ctx->stream));
    for (size_t i = 0; i < spec.outs.size(); ++i) {
      FinOut f = spec.outs[i];
      const bool nullable = f.null_mode != NULL_NEVER;
      Column c = make_fixed_column(ctx, f.type, capacity, nullable, false, nullable ? sub_buffer(zeroed, 64 + next_bitmap++ * out_bb, bitmap_bytes(capacity)) : nullptr);
      f.values = c.values->ptr;
      f.validity = c.validity ? (uint32_t *)c.validity->ptr : nullptr;
      fa.outs[i] = f;
      result.cols.push_back(c);
    }
    fa.out_count = (unsigned long long *)zeroed->ptr;
    ctx->prof_begin("qe_finalize_kernel");
    QE_HIP(launch_finalize(fa, ctx->stream));
    ctx->prof_end();
    if (eager) {
      QE_HIP(hipMemcpyAsync(ctx->host_counters + 16, table.ctrl, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
      queue_scale_check();
      ctx->sync();
      scale_check();
      n_groups = ctx->host_counters[16];
      overflow = ctx->host_counters[17];
      if (overflow) return false;
      if (n_groups == 0) fail(QE_ERR_EMPTY_HASHMAP_FOR_AGGREGATE, "The hashmap for aggregation is empty.");
    }
    result.rows = (int64_t)n_groups;
    for (auto &c : result.cols) finish_fixed_column(c, (int64_t)n_groups);
    verify_hashed_keys();
    for (size_t i = 0; i < spec.key_gathers.size(); ++i) {
      const AggSpec::KeyGather &kg = spec.key_gathers[i];
      const Column ids = result.cols[kg.out];  // representative row numbers; the buffer lives until the gather is done
      result.cols[kg.out] = kg.type == QE_TYPE_UTF8 ? utf8_gather_rows(ctx, key_sources[i], (const int64_t *)ids.values->ptr, (int64_t)n_groups)
                                                    : fixed_gather_rows(ctx, kg.type, key_sources[i], (const int64_t *)ids.values->ptr, (int64_t)n_groups);
    }
    if (!eager) ctx->sync();
    return true;
  }
};

struct AggIter : Iter {
  Context *ctx;
  Stream s;
  std::vector<ExprPtr> groups;  // over the child's output columns
  std::vector<AggExpr> aggs;
  int mode = QE_AGG_MODE_SINGLE;
  int64_t cap_hint = 0;
  bool done = false, yielded = false;
  Batch result;

  void compute() {
    AggState st;
    st.ctx = ctx;
    st.groups = groups;
    st.aggs = aggs;
    st.mode = mode;
    st.cap_hint = cap_hint;
    int64_t hint = cap_hint > 0 ? cap_hint : 32768;
    for (int attempt = 0; attempt < 7; ++attempt) {
      st.expected_batches = (int)std::min<int64_t>(s.src->batches_ahead(), 1 << 20);
      st.start(hint);
      s.src->reset();
      Batch in;
      try {
        // (with very many groups up to four batches go to the device at a time: one launch of the first pass can take them all)
        const bool ahead = st.effective_hint() >= spill_min_group_hint();
        while (s.src->next(in)) {
          std::vector<Batch> more;
          Batch e;
          while (ahead && in.rows > 0 && (int)more.size() < AGG_MORE_BATCHES && s.src->next(e)) more.push_back(e);
          st.update(in, stream_outputs(s, (int)in.cols.size()), s.pred, more.empty() ? nullptr : &more);
        }
      } catch (const NeedsStrictNullability &) {
        st.strict = true;  // a later batch has nulls where the earlier ones had none: once more, tracking them
        continue;
      }
      try {
        if (st.finish(result)) {
          done = true;
          return;
        }
      } catch (const NeedsExactScales &) {
        st.fix_exact = true;  // a value did not fit the scale the first rows suggested: once more, every batch's scale found first
        continue;
      } catch (const NeedsLogPath &) {
        st.direct_off = true;  // keys far from evenly spread: once more, every batch through the first pass with a table and a log
        continue;
      }
      // grow (a full table says little about how many groups there are: a big step, up to the largest table) and run the aggregation again
      if (st.cap >= (1ull << 30)) break;
      hint = std::min<int64_t>((int64_t)st.cap * 16, (int64_t)1 << 29);
    }
    fail(QE_ERR_AGG_CAPACITY, "group table still full after growing");
  }

  bool next(Batch &out) override {
    if (!done) compute();
    if (yielded) return false;
    yielded = true;
    out = result;
    return true;
  }
  void reset() override { yielded = false; }
};

// -------------------------------------------------------------------- plans
static Stream build_stream(std::unique_ptr<Plan> plan) {
  Context *ctx = plan->ctx;
  switch (plan->kind) {
    case PK_SCAN: {
      auto it = std::make_unique<ScanIter>();
      it->ctx = ctx;
      it->batches = std::move(plan->batches);
      it->parquet = plan->parquet;
      it->parquet_columns = plan->parquet_columns;
      if (plan->parquet) {
        const size_t have = plan->parquet->row_groups.size();
        it->first = std::min(have, (size_t)std::max(plan->parquet_first, 0));
        it->end = plan->parquet_end < 0 ? have : std::min(have, (size_t)plan->parquet_end);
        if (it->end < it->first) it->end = it->first;
      }
      Stream s;
      s.src = std::move(it);
      return s;
    }
    case PK_PROJECTION: {
      if (!plan->child) fail(QE_ERR_MISSING_INPUT_PHYSICAL_PLAN, "Physical plan `Projection` is missing its input.");
      Stream s = build_stream(std::move(plan->child));
      if (s.identity) {
        // column indices refer to the source directly; a wrong index surfaces at evaluation time
        s.outputs = plan->exprs;
      } else {
        std::vector<ExprPtr> o;
        for (auto &e : plan->exprs) o.push_back(expr_substitute(e, s.outputs));
        s.outputs = o;
      }
      s.identity = false;
      return s;
    }
    case PK_SELECTION: {
      if (!plan->child) fail(QE_ERR_MISSING_INPUT_PHYSICAL_PLAN, "Physical plan `Selection` is missing its input.");
      Stream s = build_stream(std::move(plan->child));
      ExprPtr p = s.identity ? plan->pred : expr_substitute(plan->pred, s.outputs);
      s.pred = s.pred ? make_binary(QE_OP_AND, s.pred, p) : p;
      return s;
    }
    case PK_AGGREGATE: {
      if (!plan->child) fail(QE_ERR_MISSING_INPUT_PHYSICAL_PLAN, "Physical plan `Aggregate` is missing its input.");
      auto it = std::make_unique<AggIter>();
      it->ctx = ctx;
      it->s = build_stream(std::move(plan->child));
      it->groups = plan->group_exprs;
      it->aggs = plan->aggs;
      it->mode = plan->mode;
      it->cap_hint = plan->cap_hint;
      Stream s;
      s.src = std::move(it);
      return s;
    }
  }
  fail(QE_ERR_PHYSICAL_PLAN_NOT_SUPPORTED, "Physical plan is not supported.");
}

static std::string plan_to_string(const Plan *p, int indent) {
  std::ostringstream o;
  o << std::string(indent * 2, ' ');
  switch (p->kind) {
    case PK_SCAN:
      if (p->parquet) {
        o << "Scan: " << p->parquet->path << "; projection=[";
        for (size_t i = 0; i < p->parquet_columns.size(); ++i) o << (i ? ", " : "") << p->parquet->columns[p->parquet_columns[i]].name;
        o << "], row_groups=" << p->parquet->row_groups.size();
        if (p->parquet_end >= 0) o << " [" << p->parquet_first << ", " << p->parquet_end << ")";
      } else {
        o << "Scan: memory, batches=" << p->batches.size();
      }
      break;
    case PK_PROJECTION: {
      o << "Projection: ";
      for (size_t i = 0; i < p->exprs.size(); ++i) o << (i ? ", " : "") << expr_to_string(p->exprs[i]);
      break;
    }
    case PK_SELECTION: o << "Selection: " << expr_to_string(p->pred); break;
    case PK_AGGREGATE: {
      o << "Aggregate: groupExpr=[";
      for (size_t i = 0; i < p->group_exprs.size(); ++i) o << (i ? ", " : "") << expr_to_string(p->group_exprs[i]);
      o << "], aggregateExpr=[";
      static const char *names[] = {"?", "max", "min", "sum", "count", "avg"};
      for (size_t i = 0; i < p->aggs.size(); ++i)
        o << (i ? ", " : "") << names[p->aggs[i].kind] << " " << (p->aggs[i].input ? expr_to_string(p->aggs[i].input) : "*");
      o << "]";
      break;
    }
  }
  o << "\n";
  if (p->child) o << plan_to_string(p->child.get(), indent + 1);
  return o.str();
}

// ------------------------------------------------------- lowering description
static void json_u32s(std::ostringstream &o, const char *key, const std::vector<uint32_t> &v) {
  o << "\"" << key << "\":[";
  for (size_t i = 0; i < v.size(); ++i) o << (i ? "," : "") << v[i];
  o << "]";
}

static std::string json_escape(const std::string &s) {
  std::string r;
  for (char ch : s) {
    if (ch == '"' || ch == '\\') r += '\\', r += ch;
    else if (ch == '\n') r += "\\n";
    else r += ch;
  }
  return r;
}

using LaunchVisitor = std::function<void(JitKind kind, const SpecFields &f)>;

// Lowers Scan(schema) -> [Selection pred] -> [Projection outs] -> [Aggregate groups, aggs] the way
// execution does and hands every kernel launch it would make to `visit`.  Touches no device.
static void for_each_lowered_launch(const Schema &schema, uint32_t flags, const ExprPtr &pred_raw, const std::vector<ExprPtr> &outs_raw,
                                    const std::vector<ExprPtr> &groups, const std::vector<AggExpr> &aggs, int mode,
                                    const LaunchVisitor &visit) {
  Batch in;
  in.rows = 1 << 20;
  for (size_t i = 0; i < schema.size(); ++i) {
    Column c;
    c.type = schema[i].type;
    c.space = QE_MEM_DEVICE;
    c.length = in.rows;
    // placeholder addresses: nothing here is ever dereferenced
    c.values = borrowed_buffer((const void *)(uintptr_t)(0x100000 * (3 * i + 1)), 0, QE_MEM_DEVICE);
    if (schema[i].nullable) c.validity = borrowed_buffer((const void *)(uintptr_t)(0x100000 * (3 * i + 2)), 0, QE_MEM_DEVICE);
    if (schema[i].type == QE_TYPE_UTF8) c.offsets = borrowed_buffer((const void *)(uintptr_t)(0x100000 * (3 * i + 3)), 0, QE_MEM_DEVICE);
    in.cols.push_back(c);
  }
  OutSink sink = [](int, bool nullable, VmOut &o) {
    o.values = (void *)(uintptr_t)0x10;
    o.validity = nullable ? (uint32_t *)(uintptr_t)0x10 : nullptr;
  };
  ExprPtr pred = pred_raw ? expr_normalize(pred_raw, flags) : nullptr;

  if (!groups.empty() || !aggs.empty()) {
    std::vector<ExprPtr> g2, second;
    std::vector<AggExpr> a2;
    for (auto &g : groups) g2.push_back(outs_raw.empty() ? g : expr_substitute(g, outs_raw));
    for (auto &a : aggs) {
      AggExpr x;
      x.kind = a.kind;
      x.input = a.input ? (outs_raw.empty() ? a.input : expr_substitute(a.input, outs_raw)) : nullptr;
      a2.push_back(x);
      ExprPtr sec;
      if (mode == QE_AGG_MODE_FINAL && a.kind == QE_AGG_AVG && a.input && a.input->kind == EK_COLUMN) {
        sec = make_column(a.input->index + 1);
        if (!outs_raw.empty()) sec = expr_substitute(sec, outs_raw);
      }
      second.push_back(sec);
    }
    // (order-independent sums: the launch has the same shape at every scale)
    const std::vector<int> scales(a2.size(), 0);
    AggSpec spec = build_agg_spec(flags, schema, g2, a2, second, mode, 0, false, (flags & QE_FLAG_DETERMINISTIC_SUM) ? &scales : nullptr);
    Compiler comp(nullptr, &in);
    AggArgs args{};
    lower_aggregate(comp, schema, pred, spec, spec, args);
    visit(JIT_AGGREGATE, spec_fields(args));
    return;
  }
  std::vector<ExprPtr> outs;
  for (auto &e : outs_raw) outs.push_back(expr_normalize(e, flags));
  std::vector<int> types, vm_list;
  for (auto &e : outs) types.push_back(expr_type(e, schema));
  for (size_t i = 0; i < outs.size(); ++i) {
    if (types[i] == QE_TYPE_UTF8) continue;             // gathered / broadcast by separate kernels
    if (!pred && outs[i]->kind == EK_COLUMN) continue;  // shared with the input, no kernel
    vm_list.push_back((int)i);
  }
  size_t k = 0;
  bool first = true;
  while ((pred && first) || k < vm_list.size()) {
    Compiler comp(nullptr, &in);
    if (pred) {
      FilterArgs args{};
      size_t k2 = lower_filter(comp, schema, pred, outs, types, vm_list, k, in.rows, nullptr, args, sink);
      if (!first && k2 == k) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression is too long for one kernel.");
      k = k2;
      visit(JIT_FILTER, spec_fields(args));
    } else {
      ProjectArgs args{};
      k = lower_project(comp, schema, outs, types, vm_list, k, in.rows, args, sink);
      visit(JIT_PROJECT, spec_fields(args));
    }
    first = false;
  }
}

static std::string describe_launch(JitKind kind, const SpecFields &f) {
  static const char *names[] = {"project", "filter", "aggregate"};
  std::ostringstream o;
  o << "{\"kind\":\"" << names[kind] << "\",";
  json_u32s(o, "sig", f.sig);
  o << ",";
  int n = program_words_used(*f.prog);
  json_u32s(o, "code", std::vector<uint32_t>(f.prog->insn, f.prog->insn + n));
  o << ",";
  std::vector<uint32_t> kinds, nullable;
  for (uint32_t i = 0; i < f.prog->cache_n; ++i) {
    kinds.push_back(f.prog->cache_kind[i]);
    nullable.push_back(f.prog->cols[f.prog->cache_col[i]].validity ? 1u : 0u);
  }
  json_u32s(o, "cache_kind", kinds);
  o << ",";
  json_u32s(o, "cache_nullable", nullable);
  o << ",\"specialisable\":" << (program_is_specialisable(*f.prog) ? 1 : 0);
  if (kind != JIT_AGGREGATE) {
    o << ",\"outs\":[";
    for (int i = 0; i < f.n_outs; ++i)
      o << (i ? "," : "") << "{\"pc\":" << f.outs[i].pc << ",\"type\":" << (int)f.outs[i].type << ",\"nullable\":" << (f.outs[i].validity ? 1 : 0) << "}";
    o << "]";
  }
  if (kind == JIT_FILTER)
    o << ",\"pred_pc\":" << f.pred_pc << ",\"pred_cache_mask\":" << f.pred_cache_mask << ",\"out_cache_mask\":" << f.out_cache_mask;
  if (kind == JIT_AGGREGATE) {
    o << ",\"tasks\":[";
    for (int i = 0; i < f.n_tasks; ++i)
      o << (i ? "," : "") << "{\"pc\":" << f.tasks[i].pc << ",\"role\":" << (int)f.tasks[i].role << ",\"idx\":" << (int)f.tasks[i].idx
        << ",\"eval\":" << (int)f.tasks[i].has_eval << "}";
    o << "],\"keys\":[";
    for (int i = 0; i < f.n_keys; ++i)
      o << (i ? "," : "") << "{\"type\":" << (int)f.keys[i].type << ",\"shift\":" << (int)f.keys[i].shift << ",\"bits\":" << (int)f.keys[i].bits << "}";
    o << "],\"accs\":[";
    for (int i = 0; i < f.n_accs; ++i)
      o << (i ? "," : "") << "{\"op\":" << (int)f.accs[i].op << ",\"type\":" << (int)f.accs[i].in_type << ",\"track\":" << (int)f.accs[i].track_nn
        << ",\"has_input\":" << (int)f.accs[i].has_input << "}";
    o << "]";
  }
  // the compile-time description a specialised kernel is built from; @STRUCT@ / @LABEL@ are filled in by the user
  o << ",\"spec_cpp\":\"" << json_escape(spec_struct_source("@STRUCT@", "@LABEL@", f)) << "\"}";
  return o.str();
}

}  // namespace qe

// ------------------------------------------------- scratch of the no-alloc entry point
// qe_selection_noalloc runs the selection kernels without taking any device memory: their scratch
// (ticket counter, result words, one look-back status word per tile) lives in this array instead of a
// buffer from hipMalloc.  Every GPU holds its own copy at its own address until the process ends; the
// array's name is only a handle for hipGetSymbolAddress, host code uses the pointer that call returns.
constexpr int64_t NOALLOC_MAX_ROWS = QE_NOALLOC_MAX_ROWS;
constexpr size_t NOALLOC_SCRATCH_WORDS = sizeof(qe::FilterScratch) / 8 + (size_t)(NOALLOC_MAX_ROWS / qe::FILTER_TILE_ROWS) + 8;
static __device__ __align__(256) unsigned long long qe_noalloc_scratch[NOALLOC_SCRATCH_WORDS];
constexpr int NOALLOC_MAX_DEVICES = 64;
static unsigned long long *g_noalloc_ptr[NOALLOC_MAX_DEVICES];  // per device: where its copy of the array lives
static std::mutex g_noalloc_setup_mutex;                         // contexts are created from any thread
// One per device: held by qe_selection_noalloc from the reset of the array to the arrival of its result, so that calls
// from several threads of one process (whatever their contexts and streams) take turns on the device's one array.
static std::mutex g_noalloc_call_mutex[NOALLOC_MAX_DEVICES];

// Once per device, with that device current (qe_context_create).
static void noalloc_scratch_setup(int device) {
  if (device < 0 || device >= NOALLOC_MAX_DEVICES) return;
  std::lock_guard<std::mutex> lock(g_noalloc_setup_mutex);
  if (g_noalloc_ptr[device]) return;
  void *ptr = nullptr;
  hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(qe_noalloc_scratch));
  if (e != hipSuccess || !ptr) qe::fail(QE_ERR_DEVICE, std::string("hipGetSymbolAddress(qe_noalloc_scratch): ") + hipGetErrorString(e));
  g_noalloc_ptr[device] = (unsigned long long *)ptr;
}
static unsigned long long *noalloc_scratch_of(int device) {
  if (device < 0 || device >= NOALLOC_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(g_noalloc_setup_mutex);
  return g_noalloc_ptr[device];
}

// ==================================================================== C ABI
using namespace qe;

struct qe_context { std::shared_ptr<Context> sp; Context &c; explicit qe_context(std::shared_ptr<Context> p) : sp(std::move(p)), c(*sp) {} };
struct qe_batch { Batch b; };
struct qe_expr { ExprPtr e; };
struct qe_agg_expr { AggExpr a; };
struct qe_plan { std::unique_ptr<Plan> p; };
struct qe_iter { std::unique_ptr<Iter> it; Context *ctx; };

static thread_local std::string g_last_error;

template <typename F>
static qe_status guarded(F &&f) {
  try {
    f();
    return QE_OK;
  } catch (const Failure &e) {
    g_last_error = e.msg;
    return e.code;
  } catch (const std::bad_alloc &) {
    g_last_error = "out of host memory";
    return QE_ERR_OUT_OF_MEMORY;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return QE_ERR_INTERNAL;
  } catch (...) {
    g_last_error = "unknown failure";
    return QE_ERR_INTERNAL;
  }
}

#define QE_REQUIRE(cond, what) \
  if (!(cond)) fail(QE_ERR_INVALID_ARGUMENT, what)

static qe_status write_string(const std::string &s, char *buf, int64_t buf_len, int64_t *needed) {
  if (needed) *needed = (int64_t)s.size();
  if (buf && buf_len > 0) {
    size_t nb = std::min<size_t>(s.size(), (size_t)buf_len - 1);
    std::memcpy(buf, s.data(), nb);
    buf[nb] = 0;
  }
  return QE_OK;
}

qe_status qe_guarded(const std::function<void()> &f) { return guarded(f); }
void qe_write_string(const std::string &s, char *buf, int64_t buf_len, int64_t *needed) { (void)write_string(s, buf, buf_len, needed); }
// The batch behind a handle, in its final shape: a Selection's output whose row count the host has not read yet
// (QE_FLAG_DEFERRED_ROW_COUNT) waits for it here, on its first use.
static qe::Batch &settled(const qe_batch *batch) {
  qe::Batch &b = const_cast<qe_batch *>(batch)->b;
  qe::batch_settle(b);
  return b;
}
const qe::Batch &qe_batch_ref(const qe_batch *batch) {
  if (!batch) fail(QE_ERR_INVALID_ARGUMENT, "batch is null");
  return settled(batch);
}
qe::Context *qe_context_ref(qe_context *ctx) { return &ctx->c; }
std::shared_ptr<qe::Context> qe_context_share(qe_context *ctx) { return ctx->sp; }
qe_batch *qe_batch_wrap(const qe::Batch &b) { return new qe_batch{b}; }
qe_plan *qe_plan_wrap(std::unique_ptr<qe::Plan> plan) { return new qe_plan{std::move(plan)}; }

extern "C" {

const char *qe_version(void) { return "qe_gpu 0.1 (gfx950)"; }
const char *qe_last_error_message(void) { return g_last_error.c_str(); }

const char *qe_status_name(qe_status s) {
  switch (s) {
    case QE_OK: return "Ok";
    case QE_END: return "End";
    case QE_ERR_EXCEEDING_BOUNDS: return "ExceedingBoundsError";
    case QE_ERR_NO_FIELD_IN_LOGICAL_PLAN: return "NoFieldInLogicalPlan";
    case QE_ERR_DIFFERENT_SIZES: return "DifferentSizes";
    case QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED: return "PhysicalExpressionNotSuported";
    case QE_ERR_PHYSICAL_PLAN_NOT_SUPPORTED: return "PhysicalPlanNotSuported";
    case QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED: return "PhysicalTypeNotSuported";
    case QE_ERR_PRIMITIVE_TYPE_NOT_SUPPORTED: return "PrimitiveTypeNotSuported";
    case QE_ERR_MISSING_CHILDREN: return "MissingChildren";
    case QE_ERR_MISSING_INPUT_PHYSICAL_PLAN: return "MissingInputPhysicalPlan";
    case QE_ERR_EMPTY_HASHMAP_FOR_AGGREGATE: return "EmptyHashmapForAggregate";
    case QE_ERR_DOWNCAST: return "DowncastError";
    case QE_ERR_SCALAR_TO_ARRAY: return "ScalarToArrayError";
    case QE_ERR_NO_BOOLEAN_ARRAY_FOR_FILTER: return "NoBooleanArrayForFilter";
    case QE_ERR_IO: return "IoError";
    case QE_ERR_ARROW: return "ArrowError";
    case QE_ERR_INVALID_ARGUMENT: return "InvalidArgument";
    case QE_ERR_DEVICE: return "DeviceError";
    case QE_ERR_OUT_OF_MEMORY: return "OutOfMemory";
    case QE_ERR_AGG_CAPACITY: return "AggregateCapacity";
    case QE_ERR_INTERNAL: return "Internal";
    default: return "Unknown";
  }
}

int32_t qe_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

qe_status qe_context_create(int32_t device, void *stream, qe_context **out) {
  return guarded([&] {
    QE_REQUIRE(out, "out is null");
    *out = nullptr;
    int n = qe_device_count();
    if (device < 0 || device >= n)
      fail(QE_ERR_DEVICE, "no usable GPU: device " + std::to_string(device) + " of " + std::to_string(n) +
                              " (this backend has no CPU fallback)");
    QE_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    QE_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      fail(QE_ERR_DEVICE, std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName);
    auto ctx = std::make_unique<qe_context>(std::make_shared<Context>());
    ctx->c.device = device;
    ctx->c.dev.cus = prop.multiProcessorCount;
    ctx->c.dev.max_lds = (int)prop.sharedMemPerBlock;
    ctx->c.dev.total_mem = (uint64_t)prop.totalGlobalMem;
    ctx->c.dev.device = device;
    ctx->c.dev.jit = jit_default_mode();
    if (stream) {
      ctx->c.stream = (hipStream_t)stream;
    } else {
      QE_HIP(hipStreamCreateWithFlags(&ctx->c.stream, hipStreamNonBlocking));
      ctx->c.own_stream = true;
    }
    noalloc_scratch_setup(device);
    *out = ctx.release();
  });
}

qe_status qe_context_set_flags(qe_context *ctx, uint32_t flags) {
  return guarded([&] {
    QE_REQUIRE(ctx, "ctx is null");
    ctx->c.flags = flags;
    ctx->c.dev.specialised = !(flags & QE_FLAG_NO_SPECIALIZED_KERNELS);
    ctx->c.dev.jit = (flags & QE_FLAG_NO_SPECIALIZED_KERNELS) ? (int)JIT_OFF : (flags & QE_FLAG_JIT_SYNC) ? (int)JIT_SYNC : jit_default_mode();
  });
}

qe_status qe_context_synchronize(qe_context *ctx) {
  return guarded([&] {
    QE_REQUIRE(ctx, "ctx is null");
    ctx->c.bind();
    ctx->c.sync();
    ctx->c.check_deferred();
  });
}

void qe_context_free(qe_context *ctx) { delete ctx; }

// ------------------------------------------------------------------ batches
static size_t values_size(const qe_column &c) {
  switch (c.type) {
    case QE_TYPE_BOOLEAN: return (size_t)((c.length + 7) / 8);
    case QE_TYPE_UTF8: return (size_t)c.values_bytes;
    default: return (size_t)c.length * type_width(c.type);
  }
}

qe_status qe_batch_import(qe_context *ctx, const qe_column *columns, int32_t n_columns, qe_batch **out) {
  return guarded([&] {
    QE_REQUIRE(ctx && out && (columns || n_columns == 0) && n_columns >= 0, "bad arguments");
    *out = nullptr;
    auto b = std::make_unique<qe_batch>();
    b->b.ctx = &ctx->c;
    int64_t rows = n_columns ? columns[0].length : 0;
    for (int i = 0; i < n_columns; ++i) {
      const qe_column &c = columns[i];
      if (c.type < QE_TYPE_INT32 || c.type > QE_TYPE_TIMESTAMP_NS)
        fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Physical type `" + std::to_string(c.type) + "` is not supported.");
      QE_REQUIRE(c.length >= 0, "negative length");
      if (c.length != rows)
        fail(QE_ERR_DIFFERENT_SIZES, "The expressions `column 0` and `column " + std::to_string(i) + "` have different sizes.");
      QE_REQUIRE(c.length == 0 || c.values || (c.type == QE_TYPE_UTF8 && c.values_bytes == 0), "values is null");
      QE_REQUIRE(c.type != QE_TYPE_UTF8 || c.offsets, "utf8 column needs offsets");
      QE_REQUIRE(c.memory_space == QE_MEM_HOST || c.memory_space == QE_MEM_DEVICE, "bad memory space");
      Column col;
      col.type = c.type;
      col.space = c.memory_space;
      col.length = c.length;
      col.null_count = c.validity ? c.null_count : 0;
      size_t vs = values_size(c);
      col.values_bytes = (int64_t)vs;
      col.values = borrowed_buffer(c.values, vs, c.memory_space);
      if (c.validity) col.validity = borrowed_buffer(c.validity, (size_t)((c.length + 7) / 8), c.memory_space);
      if (c.offsets) col.offsets = borrowed_buffer(c.offsets, (size_t)(c.length + 1) * 4, c.memory_space);
      if (c.memory_space == QE_MEM_DEVICE && type_width(c.type) && ((uintptr_t)c.values & 15u) && vs) {
        // kernels use 16-byte loads: move a misaligned buffer into an aligned one
        ctx->c.bind();
        BufferPtr aligned = device_buffer(&ctx->c, vs + 16);
        QE_HIP(hipMemcpyAsync(aligned->ptr, c.values, vs, hipMemcpyDeviceToDevice, ctx->c.stream));
        col.values = aligned;
      }
      b->b.cols.push_back(col);
    }
    b->b.rows = rows;
    *out = b.release();
  });
}

qe_status qe_batch_num_rows(const qe_batch *batch, int64_t *out) {
  return guarded([&] {
    QE_REQUIRE(batch && out, "bad arguments");
    *out = settled(batch).rows;
  });
}

qe_status qe_batch_num_columns(const qe_batch *batch, int32_t *out) {
  return guarded([&] {
    QE_REQUIRE(batch && out, "bad arguments");
    *out = (int32_t)batch->b.cols.size();
  });
}

static int64_t count_nulls(Context *ctx, const Column &c) {
  if (!c.validity) return 0;
  if (c.space == QE_MEM_HOST) {
    const uint8_t *bm = c.validity_ptr();
    int64_t z = 0;
    for (int64_t i = 0; i < c.length; ++i) z += !((bm[i >> 3] >> (i & 7)) & 1);
    return z;
  }
  ctx->bind();
  ensure_counters(ctx);
  QE_HIP(hipMemsetAsync(ctx->counters + 1, 0, sizeof(unsigned long long), ctx->stream));
  QE_HIP(launch_count_zero_bits((const uint32_t *)c.validity->ptr, c.length, ctx->counters + 1, ctx->stream));
  QE_HIP(hipMemcpyAsync(ctx->host_counters + 1, ctx->counters + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync();
  return (int64_t)ctx->host_counters[1];
}

qe_status qe_batch_column(const qe_batch *batch, int32_t index, qe_column *out) {
  return guarded([&] {
    QE_REQUIRE(batch && out, "bad arguments");
    if (index < 0 || index >= (int)batch->b.cols.size())
      fail(QE_ERR_EXCEEDING_BOUNDS, "The index `" + std::to_string(index) + "` is out of bounds.");
    Column &c = settled(batch).cols[index];
    if (c.null_count < 0) {
      // output bitmaps are produced by whole-word stores / atomics into padded buffers: count on demand
      // (a slice of one of this library's own blocks is padded like a block of its own; a borrowed bitmap may end mid-word)
      const Buffer *root = c.validity.get();
      while (root->parent) root = root->parent.get();
      bool word_ok = c.space == QE_MEM_HOST || root->owned;
      if (word_ok) c.null_count = count_nulls(batch->b.ctx, c);
    }
    out->type = c.type;
    out->memory_space = c.space;
    out->length = c.length;
    out->null_count = c.null_count;
    out->validity = c.validity_ptr();
    out->offsets = c.offsets_ptr();
    out->values = c.values_ptr();
    out->values_bytes = c.values_bytes;
  });
}

static Column copy_column(Context *ctx, const Column &c, int to_space) {
  Column o;
  o.type = c.type;
  o.space = to_space;
  o.length = c.length;
  o.null_count = c.null_count;
  o.values_bytes = c.values_bytes;
  hipMemcpyKind kind = (c.space == QE_MEM_HOST && to_space == QE_MEM_DEVICE)   ? hipMemcpyHostToDevice
                       : (c.space == QE_MEM_DEVICE && to_space == QE_MEM_HOST) ? hipMemcpyDeviceToHost
                       : (c.space == QE_MEM_DEVICE)                            ? hipMemcpyDeviceToDevice
                                                                               : hipMemcpyHostToHost;
  auto mk = [&](size_t bytes, size_t pad) -> BufferPtr {
    if (to_space == QE_MEM_DEVICE) {
      BufferPtr b = device_buffer(ctx, bytes + pad);
      QE_HIP(hipMemsetAsync(b->ptr, 0, bytes + pad, ctx->stream));
      return b;
    }
    return host_buffer(bytes + pad);
  };
  size_t vbytes = c.type == QE_TYPE_BOOLEAN ? (size_t)((c.length + 7) / 8) : (size_t)c.values_bytes;
  size_t vpad = c.type == QE_TYPE_BOOLEAN ? bitmap_bytes(c.length) - vbytes : 16;
  o.values = mk(vbytes, vpad);
  if (vbytes) QE_HIP(hipMemcpyAsync(o.values->ptr, c.values->ptr, vbytes, kind, ctx->stream));
  if (c.validity) {
    size_t bb = (size_t)((c.length + 7) / 8);
    o.validity = mk(bb, bitmap_bytes(c.length) - bb);
    if (bb) QE_HIP(hipMemcpyAsync(o.validity->ptr, c.validity->ptr, bb, kind, ctx->stream));
  }
  if (c.offsets) {
    size_t ob = (size_t)(c.length + 1) * 4;
    o.offsets = mk(ob, 16);
    QE_HIP(hipMemcpyAsync(o.offsets->ptr, c.offsets->ptr, ob, kind, ctx->stream));
  }
  return o;
}

static qe_status copy_batch(qe_context *ctx, const qe_batch *batch, qe_batch **out, int space) {
  return guarded([&] {
    QE_REQUIRE(ctx && batch && out, "bad arguments");
    *out = nullptr;
    ctx->c.bind();
    auto b = std::make_unique<qe_batch>();
    b->b.ctx = &ctx->c;
    b->b.rows = settled(batch).rows;
    for (auto &c : batch->b.cols) b->b.cols.push_back(copy_column(&ctx->c, c, space));
    ctx->c.sync();
    *out = b.release();
  });
}

qe_status qe_batch_to_device(qe_context *ctx, const qe_batch *batch, qe_batch **out) { return copy_batch(ctx, batch, out, QE_MEM_DEVICE); }
qe_status qe_batch_to_host(qe_context *ctx, const qe_batch *batch, qe_batch **out) { return copy_batch(ctx, batch, out, QE_MEM_HOST); }
void qe_batch_free(qe_batch *batch) { delete batch; }

// -------------------------------------------------------------- expressions
static qe_status new_expr(ExprPtr e, qe_expr **out) {
  return guarded([&] {
    QE_REQUIRE(out, "out is null");
    *out = new qe_expr{std::move(e)};
  });
}
qe_status qe_expr_column(int32_t index, qe_expr **out) { return new_expr(make_column(index), out); }
qe_status qe_expr_lit_i32(int32_t v, qe_expr **out) { return new_expr(make_lit_i32(v), out); }
qe_status qe_expr_lit_i64(int64_t v, qe_expr **out) { return new_expr(make_lit_i64(v), out); }
qe_status qe_expr_lit_f64(double v, qe_expr **out) { return new_expr(make_lit_f64(v), out); }
qe_status qe_expr_lit_bool(int32_t v, qe_expr **out) { return new_expr(make_lit_bool(v != 0), out); }
qe_status qe_expr_lit_utf8(const char *bytes, int64_t len, qe_expr **out) {
  return guarded([&] {
    QE_REQUIRE(out && (bytes || len == 0) && len >= 0, "bad arguments");
    *out = new qe_expr{make_lit_utf8(std::string(bytes ? bytes : "", (size_t)len))};
  });
}

qe_status qe_expr_binary(int32_t op, qe_expr *left, qe_expr *right, qe_expr **out) {
  return guarded([&] {
    QE_REQUIRE(left && right && out, "bad arguments");
    if (op < QE_OP_EQ || op > QE_OP_MOD) {
      delete left;
      delete right;
      fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression `op " + std::to_string(op) + "` is not supported.");
    }
    ExprPtr e = make_binary(op, left->e, right->e);
    delete left;
    delete right;
    *out = new qe_expr{e};
  });
}

qe_status qe_expr_clone(const qe_expr *expr, qe_expr **out) {
  return guarded([&] {
    QE_REQUIRE(expr && out, "bad arguments");
    *out = new qe_expr{expr->e};  // nodes are immutable: sharing is a deep copy in effect
  });
}

qe_status qe_expr_to_string(const qe_expr *expr, char *buf, int64_t buf_len, int64_t *needed) {
  return guarded([&] {
    QE_REQUIRE(expr, "expr is null");
    write_string(expr_to_string(expr->e), buf, buf_len, needed);
  });
}

qe_status qe_expr_evaluate(qe_context *ctx, const qe_expr *expr, const qe_batch *batch, qe_value *out) {
  return guarded([&] {
    QE_REQUIRE(ctx && expr && batch && out, "bad arguments");
    std::memset(out, 0, sizeof(*out));
    Schema schema = schema_of(settled(batch));
    ExprPtr e = expr_normalize(expr->e, ctx->c.flags);
    int t = expr_type(e, schema);
    out->type = t;
    if (e->kind == EK_LITERAL) {
      // literal expressions evaluate to ColumnarValue::Scalar (physical_expressions.rs:84-86 etc.)
      out->is_scalar = 1;
      out->scalar_valid = e->valid ? 1 : 0;
      out->scalar_i64 = e->ival;
      out->scalar_f64 = e->fval;
      if (t == QE_TYPE_UTF8) {
        char *s = (char *)std::malloc(e->sval.size() + 1);
        std::memcpy(s, e->sval.data(), e->sval.size());
        s[e->sval.size()] = 0;
        out->scalar_utf8 = s;
        out->scalar_utf8_len = (int64_t)e->sval.size();
      }
      return;
    }
    Batch r = run_pipeline(&ctx->c, batch->b, {e}, nullptr);
    out->array = new qe_batch{r};
  });
}

void qe_value_release(qe_value *value) {
  if (!value) return;
  if (value->scalar_utf8) std::free((void *)value->scalar_utf8);
  delete value->array;
  std::memset(value, 0, sizeof(*value));
}

void qe_expr_free(qe_expr *expr) { delete expr; }

qe_status qe_agg_expr_new(int32_t kind, qe_expr *input, qe_agg_expr **out) {
  return guarded([&] {
    QE_REQUIRE(out, "out is null");
    ExprPtr in = input ? input->e : nullptr;
    delete input;
    if (kind < QE_AGG_MAX || kind > QE_AGG_AVG)
      fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression `aggregate " + std::to_string(kind) + "` is not supported.");
    if (!in && kind != QE_AGG_COUNT) fail(QE_ERR_INVALID_ARGUMENT, "only COUNT may omit its input");
    *out = new qe_agg_expr{AggExpr{kind, in}};
  });
}
void qe_agg_expr_free(qe_agg_expr *agg) { delete agg; }

// -------------------------------------------------------------------- plans
qe_status qe_plan_scan_memory(qe_context *ctx, qe_batch *const *batches, int32_t n_batches, qe_plan **out) {
  return guarded([&] {
    QE_REQUIRE(ctx && out && (batches || n_batches == 0) && n_batches >= 0, "bad arguments");
    auto p = std::make_unique<Plan>();
    p->kind = PK_SCAN;
    p->ctx = &ctx->c;
    for (int i = 0; i < n_batches; ++i) {
      QE_REQUIRE(batches[i], "batch is null");
      (void)settled(batches[i]);
      for (auto &c : batches[i]->b.cols)
        if (c.space != QE_MEM_DEVICE) fail(QE_ERR_INVALID_ARGUMENT, "scan batches must be device resident (qe_batch_to_device)");
      p->batches.push_back(batches[i]->b);
    }
    *out = new qe_plan{std::move(p)};
  });
}

qe_status qe_plan_scan_parquet(qe_context *ctx, const char *path, const char *const *projection, int32_t n_projection, qe_plan **out) {
  return qe_plan_scan_parquet_row_groups(ctx, path, projection, n_projection, 0, -1, out);
}

qe_status qe_plan_scan_parquet_row_groups(qe_context *ctx, const char *path, const char *const *projection, int32_t n_projection,
                                          int32_t first_row_group, int32_t n_row_groups, qe_plan **out) {
  return guarded([&] {
    QE_REQUIRE(ctx && out && path && (projection || n_projection <= 0) && first_row_group >= 0, "bad arguments");
    auto p = std::make_unique<Plan>();
    p->parquet_first = first_row_group;
    p->parquet_end = n_row_groups < 0 ? -1 : first_row_group + n_row_groups;
    p->kind = PK_SCAN;
    p->ctx = &ctx->c;
    p->parquet = parquet_open(path);
    if (n_projection < 0) {
      for (size_t i = 0; i < p->parquet->columns.size(); ++i) p->parquet_columns.push_back((int)i);
    } else {
      std::vector<std::string> names;
      for (int i = 0; i < n_projection; ++i) {
        QE_REQUIRE(projection[i], "projection name is null");
        names.push_back(projection[i]);
      }
      p->parquet_columns = parquet_project(*p->parquet, names);
    }
    *out = new qe_plan{std::move(p)};
  });
}

qe_status qe_parquet_schema(const char *path, char *buf, int64_t buf_len, int64_t *needed) {
  return guarded([&] {
    QE_REQUIRE(path, "bad arguments");
    auto f = parquet_open(path);
    std::ostringstream o;
    o << "{\"fields\":[";
    for (size_t i = 0; i < f->columns.size(); ++i) {
      const ParquetColumn &c = f->columns[i];
      std::string name;
      for (char ch : c.name) {
        if (ch == '"' || ch == '\\') name += '\\';
        name += ch;
      }
      o << (i ? "," : "") << "{\"name\":\"" << name << "\",\"data_type\":\"" << (c.type ? type_name(c.type) : ("Unsupported(" + c.type_text + ")").c_str())
        << "\",\"type\":" << c.type << ",\"is_nullable\":" << (c.optional ? "true" : "false") << "}";
    }
    o << "],\"rows\":" << f->num_rows << ",\"row_groups\":" << f->row_groups.size() << "}";
    qe_write_string(o.str(), buf, buf_len, needed);
  });
}

qe_status qe_plan_projection(qe_plan *child, qe_expr *const *exprs, int32_t n_exprs, qe_plan **out) {
  return guarded([&] {
    QE_REQUIRE(out && (exprs || n_exprs == 0) && n_exprs >= 0, "bad arguments");
    auto p = std::make_unique<Plan>();
    p->kind = PK_PROJECTION;
    for (int i = 0; i < n_exprs; ++i) {
      QE_REQUIRE(exprs[i], "expr is null");
      p->exprs.push_back(exprs[i]->e);
      delete exprs[i];
    }
    if (!child) fail(QE_ERR_MISSING_INPUT_PHYSICAL_PLAN, "Physical plan `Projection` is missing its input.");
    p->ctx = child->p->ctx;
    p->child = std::move(child->p);
    delete child;
    *out = new qe_plan{std::move(p)};
  });
}

qe_status qe_plan_selection(qe_plan *child, qe_expr *predicate, qe_plan **out) {
  return guarded([&] {
    QE_REQUIRE(out && predicate, "bad arguments");
    auto p = std::make_unique<Plan>();
    p->kind = PK_SELECTION;
    p->pred = predicate->e;
    delete predicate;
    if (!child) fail(QE_ERR_MISSING_INPUT_PHYSICAL_PLAN, "Physical plan `Selection` is missing its input.");
    p->ctx = child->p->ctx;
    p->child = std::move(child->p);
    delete child;
    *out = new qe_plan{std::move(p)};
  });
}

qe_status qe_plan_aggregate(qe_plan *child, qe_expr *const *group_exprs, int32_t n_group_exprs,
                            qe_agg_expr *const *agg_exprs, int32_t n_agg_exprs, int32_t mode,
                            int64_t group_capacity_hint, qe_plan **out) {
  return guarded([&] {
    QE_REQUIRE(out && n_group_exprs >= 0 && n_agg_exprs >= 0, "bad arguments");
    QE_REQUIRE(mode >= QE_AGG_MODE_SINGLE && mode <= QE_AGG_MODE_FINAL, "bad aggregate mode");
    auto p = std::make_unique<Plan>();
    p->kind = PK_AGGREGATE;
    p->mode = mode;
    p->cap_hint = group_capacity_hint;
    for (int i = 0; i < n_group_exprs; ++i) {
      QE_REQUIRE(group_exprs[i], "group expr is null");
      p->group_exprs.push_back(group_exprs[i]->e);
      delete group_exprs[i];
    }
    for (int i = 0; i < n_agg_exprs; ++i) {
      QE_REQUIRE(agg_exprs[i], "aggregate expr is null");
      p->aggs.push_back(agg_exprs[i]->a);
      delete agg_exprs[i];
    }
    if (!child) fail(QE_ERR_MISSING_INPUT_PHYSICAL_PLAN, "Physical plan `Aggregate` is missing its input.");
    p->ctx = child->p->ctx;
    p->child = std::move(child->p);
    delete child;
    *out = new qe_plan{std::move(p)};
  });
}

// ------------------------------------------------------ streaming aggregation
struct qe_agg_state {
  AggState st;
  std::vector<Batch> history;  // what has been folded in, should the table have to grow
  int64_t hint = 0;
  bool started = false;
};

qe_status qe_agg_state_new(qe_context *ctx, qe_expr *const *group_exprs, int32_t n_group_exprs, qe_agg_expr *const *agg_exprs,
                           int32_t n_agg_exprs, int32_t mode, int64_t group_capacity_hint, qe_agg_state **out) {
  return guarded([&] {
    QE_REQUIRE(ctx && out && n_group_exprs >= 0 && n_agg_exprs >= 0 && (group_exprs || !n_group_exprs) && (agg_exprs || !n_agg_exprs), "bad arguments");
    QE_REQUIRE(mode >= QE_AGG_MODE_SINGLE && mode <= QE_AGG_MODE_FINAL, "bad aggregate mode");
    *out = nullptr;
    auto s = std::make_unique<qe_agg_state>();
    s->st.ctx = &ctx->c;
    s->st.mode = mode;
    s->st.cap_hint = group_capacity_hint;
    s->hint = group_capacity_hint > 0 ? group_capacity_hint : 32768;
    for (int i = 0; i < n_group_exprs; ++i) {
      QE_REQUIRE(group_exprs[i], "group expr is null");
      s->st.groups.push_back(group_exprs[i]->e);
      delete group_exprs[i];
    }
    for (int i = 0; i < n_agg_exprs; ++i) {
      QE_REQUIRE(agg_exprs[i], "aggregate expr is null");
      s->st.aggs.push_back(agg_exprs[i]->a);
      delete agg_exprs[i];
    }
    *out = s.release();
  });
}

qe_status qe_agg_update(qe_agg_state *state, const qe_batch *batch) {
  return guarded([&] {
    QE_REQUIRE(state && batch, "bad arguments");
    (void)settled(batch);
    for (auto &c : batch->b.cols)
      if (c.space != QE_MEM_DEVICE) fail(QE_ERR_INVALID_ARGUMENT, "batches must be device resident (qe_batch_to_device)");
    if (!state->started) {
      state->st.start(state->hint);
      state->started = true;
    }
    try {
      state->st.update(batch->b, {}, nullptr);
    } catch (const NeedsStrictNullability &) {
      state->st.strict = true;
      state->st.start(state->hint);
      for (auto &b : state->history) state->st.update(b, {}, nullptr);
      state->st.update(batch->b, {}, nullptr);
    }
    state->history.push_back(batch->b);
  });
}

qe_status qe_agg_finalize(qe_agg_state *state, qe_batch **out) {
  return guarded([&] {
    QE_REQUIRE(state && out, "bad arguments");
    *out = nullptr;
    if (!state->started) fail(QE_ERR_EMPTY_HASHMAP_FOR_AGGREGATE, "The hashmap for aggregation is empty.");
    Batch result;
    for (int attempt = 0; attempt < 7; ++attempt) {
      bool grow = true;
      try {
        if (state->st.finish(result)) {
          result.ctx = state->st.ctx;
          *out = new qe_batch{result};
          return;
        }
      } catch (const NeedsExactScales &) {
        state->st.fix_exact = true;  // an order-independent sum met a value beyond its scale: everything once more, scales found first
        grow = false;
      } catch (const NeedsLogPath &) {
        state->st.direct_off = true;  // a first pass that partitions directly met keys far from evenly spread: everything once more, through the log
        grow = false;
      }
      // the table overflowed: a larger one, and everything seen so far folded in again
      if (grow) {
        if (state->st.cap >= (1ull << 30)) break;
        state->hint = std::min<int64_t>((int64_t)state->st.cap * 16, (int64_t)1 << 29);
      }
      state->st.start(state->hint);
      for (auto &b : state->history) state->st.update(b, {}, nullptr);
    }
    fail(QE_ERR_AGG_CAPACITY, "group table still full after growing");
  });
}

void qe_agg_state_free(qe_agg_state *state) { delete state; }

qe_status qe_plan_to_string(const qe_plan *plan, char *buf, int64_t buf_len, int64_t *needed) {
  return guarded([&] {
    QE_REQUIRE(plan && plan->p, "plan is null");
    write_string(plan_to_string(plan->p.get(), 0), buf, buf_len, needed);
  });
}

qe_status qe_plan_num_children(const qe_plan *plan, int32_t *out) {
  return guarded([&] {
    QE_REQUIRE(plan && plan->p && out, "bad arguments");
    *out = plan->p->child ? 1 : 0;
  });
}

void qe_plan_free(qe_plan *plan) { delete plan; }

qe_status qe_plan_execute(qe_plan *plan, qe_iter **out) {
  return guarded([&] {
    QE_REQUIRE(plan && plan->p && out, "bad arguments");
    Context *ctx = plan->p->ctx;
    std::unique_ptr<Plan> p = std::move(plan->p);
    delete plan;
    auto it = std::make_unique<PipeIter>();
    it->ctx = ctx;
    it->s = build_stream(std::move(p));
    *out = new qe_iter{std::move(it), ctx};
  });
}

qe_status qe_iter_next(qe_iter *iter, qe_batch **out) {
  qe_status end = QE_OK;
  qe_status st = guarded([&] {
    QE_REQUIRE(iter && out, "bad arguments");
    *out = nullptr;
    Batch b;
    if (!iter->it->next(b)) {
      end = QE_END;
      return;
    }
    b.ctx = iter->ctx;
    *out = new qe_batch{b};
  });
  return st != QE_OK ? st : end;
}

void qe_iter_free(qe_iter *iter) { delete iter; }

// ------------------------------------------------------ no-alloc selection
static bool expr_needs_device_memory(const ExprPtr &e, const Batch &in) {
  // utf8 literals are copied to the device, utf8 / timestamp columns take gather passes with buffers of their own
  if (!e) return false;
  if (e->kind == EK_LITERAL) return e->type == QE_TYPE_UTF8;
  if (e->kind == EK_COLUMN)
    return e->index >= 0 && e->index < (int)in.cols.size() && (in.cols[e->index].type == QE_TYPE_UTF8 || in.cols[e->index].type == QE_TYPE_TIMESTAMP_NS);
  return expr_needs_device_memory(e->l, in) || expr_needs_device_memory(e->r, in);
}

qe_status qe_selection_noalloc(qe_context *ctx, const qe_batch *batch, const qe_expr *predicate, const qe_expr *const *exprs,
                               int32_t n_exprs, const qe_column *outs, int64_t *rows_out) {
  return guarded([&] {
    QE_REQUIRE(ctx && batch && predicate && rows_out && n_exprs >= 1 && exprs && outs, "bad arguments");
    *rows_out = 0;
    Context *c = &ctx->c;
    c->bind();
    const int64_t n = settled(batch).rows;
    if (n > NOALLOC_MAX_ROWS)
      fail(QE_ERR_INVALID_ARGUMENT, "qe_selection_noalloc: " + std::to_string(n) + " rows exceed the maximum of " +
                                        std::to_string(NOALLOC_MAX_ROWS) + " (QE_NOALLOC_MAX_ROWS) its fixed scratch array serves");
    if (n_exprs > VM_MAX_OUTS) fail(QE_ERR_INVALID_ARGUMENT, "qe_selection_noalloc: at most " + std::to_string(VM_MAX_OUTS) + " output expressions");
    unsigned long long *scratch_words = noalloc_scratch_of(c->device);
    if (!scratch_words) fail(QE_ERR_DEVICE, "qe_selection_noalloc: no scratch array on this device");
    for (auto &col : batch->b.cols)
      if (col.space != QE_MEM_DEVICE) fail(QE_ERR_INVALID_ARGUMENT, "batches must be device resident (qe_batch_to_device)");
    Batch in = without_redundant_validity(batch->b);
    Schema schema = schema_of(in);
    ExprPtr pred = expr_normalize(predicate->e, c->flags);
    if (expr_type(pred, schema) != QE_TYPE_BOOLEAN)
      fail(QE_ERR_NO_BOOLEAN_ARRAY_FOR_FILTER, "Expr doesn't evaluate to a boolean array, which is needed to filter.");
    if (pred->kind == EK_LITERAL && !pred->valid) fail(QE_ERR_SCALAR_TO_ARRAY, "Couldn't convert Scalar value `null` to array.");
    std::vector<ExprPtr> es;
    std::vector<int> types, vm_list;
    ExprFitter fitter{c, in, schema};
    if (expr_needs_device_memory(pred, in) || fitter.over(pred))
      fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "qe_selection_noalloc: the predicate needs device memory of its own (utf8, or too large for one kernel)");
    for (int i = 0; i < n_exprs; ++i) {
      QE_REQUIRE(exprs[i], "expr is null");
      ExprPtr e = expr_normalize(exprs[i]->e, c->flags);
      const int t = expr_type(e, schema);
      if (e->kind == EK_LITERAL && !e->valid) fail(QE_ERR_SCALAR_TO_ARRAY, "Couldn't convert Scalar value `null` to array.");
      if (expr_needs_device_memory(e, in) || t == QE_TYPE_UTF8 || fitter.over(e))
        fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "qe_selection_noalloc: output " + std::to_string(i) + " needs device memory of its own (utf8, timestamp, or too large for one kernel)");
      const qe_column &o = outs[i];
      if (o.type != t) fail(QE_ERR_INVALID_ARGUMENT, "qe_selection_noalloc: output " + std::to_string(i) + " has type " + type_name(o.type) + ", the expression gives " + type_name(t));
      if (o.memory_space != QE_MEM_DEVICE || !o.values) fail(QE_ERR_INVALID_ARGUMENT, "qe_selection_noalloc: output buffers must be device memory");
      if (o.length < n) fail(QE_ERR_INVALID_ARGUMENT, "qe_selection_noalloc: output " + std::to_string(i) + " holds " + std::to_string(o.length) + " rows, the batch has " + std::to_string(n));
      if (expr_nullable(e, schema) && !o.validity) fail(QE_ERR_INVALID_ARGUMENT, "qe_selection_noalloc: output " + std::to_string(i) + " can be null and needs a validity buffer");
      if (((uintptr_t)o.values & (t == QE_TYPE_INT32 || t == QE_TYPE_BOOLEAN ? 3u : 7u)) || ((uintptr_t)o.validity & 3u))
        fail(QE_ERR_INVALID_ARGUMENT, "qe_selection_noalloc: output buffers must be aligned to their element (bitmaps: 4 bytes)");
      es.push_back(e);
      types.push_back(t);
      vm_list.push_back(i);
    }
    if (n == 0) return;
    Compiler comp(c, &in);
    FilterArgs args{};
    const size_t placed = lower_filter(comp, schema, pred, es, types, vm_list, 0, n, nullptr, args, [&](int i, bool nullable, VmOut &o) {
      o.values = const_cast<void *>(outs[i].values);
      o.validity = nullable ? (uint32_t *)const_cast<void *>(outs[i].validity) : nullptr;
    });
    if (placed != vm_list.size() || !comp.keepalive.empty())
      fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "qe_selection_noalloc: the expressions do not fit one kernel launch");
    // the device's one array serves one call at a time: calls from other threads of this process wait here
    std::lock_guard<std::mutex> one_call_at_a_time(g_noalloc_call_mutex[c->device]);
    // bitmaps are written with atomic ORs into whole words: they start out zero
    const size_t words_bytes = (size_t)((n + 31) / 32) * 4;
    for (int i = 0; i < n_exprs; ++i) {
      if (types[i] == QE_TYPE_BOOLEAN) QE_HIP(hipMemsetAsync(const_cast<void *>(outs[i].values), 0, words_bytes, c->stream));
      if (args.outs[i].validity) QE_HIP(hipMemsetAsync(args.outs[i].validity, 0, words_bytes, c->stream));
    }
    FilterScratch *scratch = (FilterScratch *)scratch_words;
    args.scratch = scratch;
    QE_HIP(launch_filter_reset(scratch, n, c->stream));  // the array keeps the last call's tickets and status words
    DeviceInfo dev = c->dev;
    dev.jit = JIT_OFF;  // compiling a kernel loads a code object: device memory
    const char *kname = "qe_filter_kernel";
    c->prof_begin(kname, n);
    QE_HIP(launch_filter(args, dev, c->stream, &kname));
    c->prof_end(kname);
    unsigned long long result[3] = {0, 0, 0};
    QE_HIP(hipMemcpyAsync(result, scratch, sizeof(result), hipMemcpyDeviceToHost, c->stream));
    c->sync();
    if (result[2] != 0) fail(QE_ERR_INTERNAL, "filter look-back timed out");
    *rows_out = (int64_t)result[1];
  });
}

// ----------------------------------------------------------------- utilities
qe_status qe_generate_column(qe_context *ctx, int32_t kind, uint64_t seed, int64_t row0, int64_t n, double lo, double hi,
                             void *device_out) {
  return guarded([&] {
    QE_REQUIRE(ctx && (device_out || n == 0) && n >= 0 && kind >= 0 && kind <= 3, "bad arguments");
    ctx->c.bind();
    ctx->c.prof_begin("qe_generate_kernel");
    QE_HIP(launch_generate(kind, seed, row0, n, lo, hi, device_out, ctx->c.stream));
    ctx->c.prof_end();
  });
}

static void lowering_arguments(const int32_t *col_types, const uint8_t *col_nullable, int32_t n_cols, const qe_expr *const *exprs,
                               int32_t n_exprs, const qe_expr *const *group_exprs, int32_t n_group_exprs,
                               const qe_agg_expr *const *agg_exprs, int32_t n_agg_exprs, Schema &schema, std::vector<ExprPtr> &outs,
                               std::vector<ExprPtr> &groups, std::vector<AggExpr> &aggs) {
  QE_REQUIRE((col_types || n_cols == 0) && n_cols >= 0 && n_exprs >= 0 && n_group_exprs >= 0 && n_agg_exprs >= 0, "bad arguments");
  for (int i = 0; i < n_cols; ++i) schema.push_back(ColInfo{col_types[i], col_nullable ? col_nullable[i] != 0 : false});
  for (int i = 0; i < n_exprs; ++i) outs.push_back(exprs[i]->e);
  for (int i = 0; i < n_group_exprs; ++i) groups.push_back(group_exprs[i]->e);
  for (int i = 0; i < n_agg_exprs; ++i) aggs.push_back(agg_exprs[i]->a);
}

qe_status qe_lowering_describe(const int32_t *col_types, const uint8_t *col_nullable, int32_t n_cols, uint32_t flags,
                               const qe_expr *predicate, const qe_expr *const *exprs, int32_t n_exprs,
                               const qe_expr *const *group_exprs, int32_t n_group_exprs, const qe_agg_expr *const *agg_exprs,
                               int32_t n_agg_exprs, int32_t mode, char *buf, int64_t buf_len, int64_t *needed) {
  return guarded([&] {
    Schema schema;
    std::vector<ExprPtr> outs, groups;
    std::vector<AggExpr> aggs;
    lowering_arguments(col_types, col_nullable, n_cols, exprs, n_exprs, group_exprs, n_group_exprs, agg_exprs, n_agg_exprs, schema, outs,
                       groups, aggs);
    std::string json = "{\"launches\":[";
    bool first = true;
    for_each_lowered_launch(schema, flags, predicate ? predicate->e : nullptr, outs, groups, aggs, mode, [&](JitKind kind, const SpecFields &f) {
      json += (first ? "" : ",") + describe_launch(kind, f);
      first = false;
    });
    json += "]}";
    write_string(json, buf, buf_len, needed);
  });
}

qe_status qe_lowering_compile(const int32_t *col_types, const uint8_t *col_nullable, int32_t n_cols, uint32_t flags,
                              const qe_expr *predicate, const qe_expr *const *exprs, int32_t n_exprs,
                              const qe_expr *const *group_exprs, int32_t n_group_exprs, const qe_agg_expr *const *agg_exprs,
                              int32_t n_agg_exprs, int32_t mode, int32_t *n_kernels, int64_t *code_bytes) {
  return guarded([&] {
    QE_REQUIRE(n_kernels && code_bytes, "bad arguments");
    *n_kernels = 0;
    *code_bytes = 0;
    Schema schema;
    std::vector<ExprPtr> outs, groups;
    std::vector<AggExpr> aggs;
    lowering_arguments(col_types, col_nullable, n_cols, exprs, n_exprs, group_exprs, n_group_exprs, agg_exprs, n_agg_exprs, schema, outs,
                       groups, aggs);
    for_each_lowered_launch(schema, flags, predicate ? predicate->e : nullptr, outs, groups, aggs, mode, [&](JitKind kind, const SpecFields &f) {
      size_t n = jit_compile_only(kind, f);
      if (n) {
        *n_kernels += 1;
        *code_bytes += (int64_t)n;
      }
    });
  });
}

qe_status qe_context_profile(qe_context *ctx, int32_t enable) {
  return guarded([&] {
    QE_REQUIRE(ctx, "ctx is null");
    ctx->c.profiling = enable != 0;
    ctx->c.profile_min_rows = enable == 2 ? (int64_t)1 << 20 : 0;
  });
}

qe_status qe_context_profile_read(qe_context *ctx, char *buf, int64_t buf_len, int64_t *needed) {
  return guarded([&] {
    QE_REQUIRE(ctx, "ctx is null");
    Context &c = ctx->c;
    c.bind();
    c.sync();
    std::map<std::string, std::pair<int, double>> agg;
    for (auto &r : c.launches) {
      float ms = 0;
      QE_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
      auto &a = agg[r.name];
      a.first += 1;
      a.second += ms;
      (void)hipEventDestroy(r.start);
      (void)hipEventDestroy(r.stop);
    }
    c.launches.clear();
    std::ostringstream o;
    o << "{";
    bool first = true;
    for (auto &kv : agg) {
      o << (first ? "" : ", ") << "\"" << kv.first << "\": {\"launches\": " << kv.second.first << ", \"ms\": " << kv.second.second << "}";
      first = false;
    }
    o << "}";
    write_string(o.str(), buf, buf_len, needed);
  });
}

}  // extern "C"
