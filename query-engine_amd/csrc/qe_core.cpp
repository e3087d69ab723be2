// Context, device memory pool, containers, expression trees and the
// expression -> stack-machine compiler.
#include "qe_core.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace qe {

[[noreturn]] void fail(qe_status code, const std::string &msg) { throw Failure{code, msg}; }

void hip_check(hipError_t e, const char *what) {
  if (e != hipSuccess) fail(QE_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

// ------------------------------------------------------------------ context
static size_t bucket_of(size_t bytes) {
  // power-of-two buckets up to 1 MiB, then eighths of an octave (waste <= 12.5 %)
  if (bytes <= 256) return 256;
  size_t oct = 256;
  while (oct * 2 <= bytes) oct <<= 1;
  if (oct == bytes) return bytes;
  if (oct < ((size_t)1 << 20)) return oct * 2;
  size_t q = oct >> 3;
  return ((bytes + q - 1) / q) * q;
}

void Context::bind() const { QE_HIP(hipSetDevice(device)); }
void Context::sync() const { QE_HIP(hipStreamSynchronize(stream)); }

void *Context::alloc(size_t bytes, size_t *actual) {
  size_t b = bucket_of(bytes ? bytes : 1);
  *actual = b;
  auto it = pool.find(b);
  if (it != pool.end() && !it->second.empty()) {
    void *p = it->second.back();
    it->second.pop_back();
    return p;
  }
  bind();
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, b);
  if (e != hipSuccess) {
    // give cached blocks back and retry once
    for (auto &kv : pool)
      for (void *q : kv.second) (void)hipFree(q);
    pool.clear();
    pool_bytes = 0;
    e = hipMalloc(&p, b);
    if (e != hipSuccess) fail(QE_ERR_OUT_OF_MEMORY, "hipMalloc of " + std::to_string(b) + " bytes failed");
  }
  return p;
}

void Context::release(void *p, size_t actual) {
  if (!p) return;
  // Work queued on the stream may still read the block; blocks are only ever handed
  // out again to work queued later on the same stream, so reuse is ordered.
  pool[actual].push_back(p);
  pool_bytes += actual;
}

void Context::prof_begin(const char *name, int64_t rows) {
  prof_open = profiling && (profile_min_rows == 0 || rows >= profile_min_rows);  // small helpers (rows unknown = -1) are skipped when a floor is set
  if (!prof_open) return;
  LaunchRec r;
  r.name = name;
  QE_HIP(hipEventCreate(&r.start));
  QE_HIP(hipEventCreate(&r.stop));
  QE_HIP(hipEventRecord(r.start, stream));
  launches.push_back(r);
}
void Context::prof_end(const char *actual_name) {
  if (!prof_open || launches.empty()) return;
  prof_open = false;
  if (actual_name) launches.back().name = actual_name;
  QE_HIP(hipEventRecord(launches.back().stop, stream));
}

unsigned long long *Context::take_pending_words() {
  if (pending_free.empty()) {
    bind();
    constexpr int BLOCKS = 64;
    unsigned long long *slab = nullptr;
    QE_HIP(hipHostMalloc((void **)&slab, (size_t)BLOCKS * PENDING_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
    pending_slabs.push_back(slab);
    for (int i = 0; i < BLOCKS; ++i) pending_free.push_back(slab + (size_t)i * PENDING_WORDS);
  }
  unsigned long long *w = pending_free.back();
  pending_free.pop_back();
  return w;
}

hipEvent_t Context::take_pending_event() {
  if (!pending_events.empty()) {
    hipEvent_t e = pending_events.back();
    pending_events.pop_back();
    return e;
  }
  bind();
  hipEvent_t e = nullptr;
  QE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}

void PendingRows::wait() {
  if (waited) return;
  QE_HIP(hipEventSynchronize(ready));
  waited = true;
}

PendingRows::~PendingRows() {
  if (!ctx_ref) return;
  // (the copy into the words may still be queued: the block is handed out again only to a copy queued later on the
  // same stream, and nobody reads this one's any more)
  if (host) ctx_ref->pending_free.push_back(host);
  if (ready) ctx_ref->pending_events.push_back(ready);
}

void batch_settle(Batch &b) {
  if (!b.pending) return;
  std::shared_ptr<PendingRows> p = b.pending;
  p->wait();
  b.pending.reset();
  if (p->host[p->filter_at + 2] != 0) fail(QE_ERR_INTERNAL, "filter look-back timed out");
  const int64_t m = (int64_t)p->host[p->filter_at + 1];
  b.rows = m;
  for (size_t i = 0; i < b.cols.size(); ++i) {
    Column &c = b.cols[i];
    c.length = m;
    const int slot = i < p->utf8_slot.size() ? p->utf8_slot[i] : -1;
    if (c.type == QE_TYPE_UTF8 && slot >= 0) {
      const unsigned long long *w = p->host + 4 * slot;
      if (w[2] != 0) fail(QE_ERR_INTERNAL, "string gather look-back timed out");
      c.values_bytes = (int64_t)w[1];
    } else if (c.type == QE_TYPE_BOOLEAN) {
      c.values_bytes = (m + 7) / 8;
    } else if (type_width(c.type)) {
      c.values_bytes = m * type_width(c.type);
    }
  }
}

Context::~Context() {
  (void)hipSetDevice(device);
  (void)hipStreamSynchronize(stream);
  if (copy_stream) {
    (void)hipStreamSynchronize(copy_stream);
    (void)hipStreamDestroy(copy_stream);
  }
  for (hipEvent_t e : pending_events) (void)hipEventDestroy(e);
  for (unsigned long long *slab : pending_slabs) (void)hipHostFree(slab);
  for (auto &r : launches) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  for (auto &kv : pool)
    for (void *q : kv.second) (void)hipFree(q);
  for (Staging &s : staging) {
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.fence) (void)hipEventDestroy(s.fence);
    if (s.images) (void)hipEventDestroy(s.images);
    if (s.ptr) (void)hipHostFree(s.ptr);
  }
  if (filter_scratch) (void)hipFree(filter_scratch);
  if (counters) (void)hipFree(counters);
  if (host_counters) (void)hipHostFree(host_counters);
  if (own_stream && stream) (void)hipStreamDestroy(stream);
}

Buffer::~Buffer() {
  if (!owned || !ptr) return;
  if (space == QE_MEM_DEVICE) {
    if (ctx) ctx->release(ptr, actual);
  } else {
    std::free(ptr);
  }
}

BufferPtr device_buffer(Context *ctx, size_t bytes) {
  auto b = std::make_shared<Buffer>();
  b->ctx = ctx;
  b->ctx_ref = ctx->shared_from_this();
  b->ptr = ctx->alloc(bytes, &b->actual);
  b->bytes = bytes;
  b->space = QE_MEM_DEVICE;
  b->owned = true;
  return b;
}

BufferPtr host_buffer(size_t bytes) {
  auto b = std::make_shared<Buffer>();
  size_t padded = (bytes + 63) & ~(size_t)63;
  if (padded == 0) padded = 64;
  b->ptr = std::aligned_alloc(64, padded);
  if (!b->ptr) fail(QE_ERR_OUT_OF_MEMORY, "host allocation failed");
  std::memset(b->ptr, 0, padded);
  b->bytes = bytes;
  b->space = QE_MEM_HOST;
  b->owned = true;
  return b;
}

BufferPtr borrowed_buffer(const void *p, size_t bytes, int space) {
  auto b = std::make_shared<Buffer>();
  b->ptr = const_cast<void *>(p);
  b->bytes = bytes;
  b->space = space;
  b->owned = false;
  return b;
}

BufferPtr sub_buffer(const BufferPtr &parent, size_t offset, size_t bytes) {
  auto b = std::make_shared<Buffer>();
  b->ptr = (uint8_t *)parent->ptr + offset;
  b->bytes = bytes;
  b->space = parent->space;
  b->owned = false;
  b->parent = parent;
  return b;
}

unsigned long long *Context::deferred_sink() {
  bind();
  if (!counters) {
    QE_HIP(hipMalloc((void **)&counters, 64 * sizeof(unsigned long long)));
    QE_HIP(hipHostMalloc((void **)&host_counters, 128 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  if (!deferred_pending) {
    QE_HIP(hipMemsetAsync(counters + 40, 0, sizeof(unsigned long long), stream));
    deferred_pending = true;
  }
  return counters + 40;
}

void Context::check_deferred() {
  if (!deferred_pending) return;
  deferred_pending = false;
  QE_HIP(hipMemcpyAsync(host_counters + 40, counters + 40, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
  sync();
  if (host_counters[40])
    fail(QE_ERR_AGG_CAPACITY, "a deferred aggregate overflowed its group table (give a larger group_capacity_hint or clear QE_FLAG_DEFERRED_GROUP_COUNT), "
                               "or an exchange by key one of its peer blocks (give a larger max_rows_per_rank)");
}

size_t bitmap_bytes(int64_t n) { return (size_t)((n + 255) / 256) * 32 + 32; }

int type_width(int type) {
  switch (type) {
    case QE_TYPE_INT32: return 4;
    case QE_TYPE_INT64: return 8;
    case QE_TYPE_FLOAT64: return 8;
    case QE_TYPE_TIMESTAMP_NS: return 8;
    default: return 0;
  }
}

const char *type_name(int type) {
  switch (type) {
    case QE_TYPE_INT32: return "Int32";
    case QE_TYPE_INT64: return "Int64";
    case QE_TYPE_FLOAT64: return "Float64";
    case QE_TYPE_BOOLEAN: return "Boolean";
    case QE_TYPE_UTF8: return "Utf8";
    case QE_TYPE_TIMESTAMP_NS: return "Timestamp(Nanosecond, None)";
    default: return "Unknown";
  }
}

Schema schema_of(const Batch &b) {
  Schema s;
  for (const auto &c : b.cols) s.push_back({c.type, c.validity != nullptr});
  return s;
}

Batch without_redundant_validity(const Batch &b) {
  Batch out = b;
  for (auto &c : out.cols)
    if (c.known_all_valid && c.validity) {
      c.validity.reset();
      c.null_count = 0;
    }
  return out;
}

// ------------------------------------------------------------- expressions
ExprPtr make_column(int index) {
  auto e = std::make_shared<Expr>();
  e->kind = EK_COLUMN;
  e->index = index;
  return e;
}
static std::shared_ptr<Expr> lit(int type) {
  auto e = std::make_shared<Expr>();
  e->kind = EK_LITERAL;
  e->type = type;
  return e;
}
ExprPtr make_lit_i32(int32_t v) { auto e = lit(QE_TYPE_INT32); e->ival = v; return e; }
ExprPtr make_lit_i64(int64_t v) { auto e = lit(QE_TYPE_INT64); e->ival = v; return e; }
ExprPtr make_lit_f64(double v) { auto e = lit(QE_TYPE_FLOAT64); e->fval = v; return e; }
ExprPtr make_lit_bool(bool v) { auto e = lit(QE_TYPE_BOOLEAN); e->ival = v ? 1 : 0; return e; }
ExprPtr make_lit_utf8(const std::string &s) { auto e = lit(QE_TYPE_UTF8); e->sval = s; return e; }
static ExprPtr make_null(int type) { auto e = lit(type); e->valid = false; return e; }

ExprPtr make_binary(int op, ExprPtr l, ExprPtr r) {
  auto e = std::make_shared<Expr>();
  e->kind = EK_BINARY;
  e->op = op;
  e->l = std::move(l);
  e->r = std::move(r);
  return e;
}

ExprPtr make_hash_utf8(ExprPtr column) {
  auto n = std::make_shared<Expr>();
  n->kind = EK_HASH_UTF8;
  n->l = std::move(column);
  return n;
}

ExprPtr make_key_hash(ExprPtr before, ExprPtr column) {
  auto n = std::make_shared<Expr>();
  n->kind = EK_KEY_HASH;
  n->l = std::move(before);
  n->r = std::move(column);
  return n;
}

ExprPtr make_fix(int part, ExprPtr value, int scale, uint64_t overflow_word) {
  auto n = std::make_shared<Expr>();
  n->kind = EK_BINARY;
  n->fix = part;
  n->ival = scale;
  n->fix_overflow_word = overflow_word;
  n->l = std::move(value);
  return n;
}

ExprPtr make_rowid(int64_t base) {
  auto n = std::make_shared<Expr>();
  n->kind = EK_ROWID;
  n->ival = base;
  return n;
}

static const char *op_symbol(int op) {
  switch (op) {
    case QE_OP_EQ: return "==";
    case QE_OP_NEQ: return "!=";
    case QE_OP_GT: return ">";
    case QE_OP_GTEQ: return ">=";
    case QE_OP_LT: return "<";
    case QE_OP_LTEQ: return "<=";
    case QE_OP_AND: return "AND";
    case QE_OP_OR: return "OR";
    case QE_OP_ADD: return "+";
    case QE_OP_SUB: return "-";
    case QE_OP_MUL: return "*";
    case QE_OP_DIV: return "/";
    case QE_OP_MOD: return "%";
    default: return "?";
  }
}

// `impl Display`: "#<index>" for columns (physical_expressions.rs:64-68),
// "#<value>" for literals, "<left> <op> <right>" for binaries (:225-229,:338-342).
std::string expr_to_string(const ExprPtr &e) {
  std::ostringstream o;
  switch (e->kind) {
    case EK_COLUMN: o << "#" << e->index; break;
    case EK_LITERAL:
      if (!e->valid) { o << "#null"; break; }
      switch (e->type) {
        case QE_TYPE_FLOAT64: { char b[64]; snprintf(b, sizeof b, "#%.17g", e->fval); o << b; break; }
        case QE_TYPE_BOOLEAN: o << (e->ival ? "#true" : "#false"); break;
        case QE_TYPE_UTF8: o << "#\"" << e->sval << "\""; break;
        default: o << "#" << e->ival; break;
      }
      break;
    case EK_BINARY:
      if (e->to_f64) { o << "float64(" << expr_to_string(e->l) << ")"; break; }
      if (e->fix) { o << "fix" << e->fix << "(" << expr_to_string(e->l) << ", " << e->ival << ")"; break; }
      o << expr_to_string(e->l) << " " << op_symbol(e->op) << " " << expr_to_string(e->r);
      break;
    case EK_HASH_UTF8: o << "hash(" << expr_to_string(e->l) << ")"; break;
    case EK_KEY_HASH: o << "keyhash(" << (e->l ? expr_to_string(e->l) + ", " : std::string()) << expr_to_string(e->r) << ")"; break;
    case EK_ROWID: o << "rowid(" << e->ival << ")"; break;
  }
  return o.str();
}

// expr_to_string follows the reference's Display, which prints a column and an integer literal alike ("#2")
// and no parentheses; this one is shared by two expressions exactly when they are the same tree.
std::string expr_identity(const ExprPtr &e) {
  if (!e) return "-";
  std::ostringstream o;
  switch (e->kind) {
    case EK_COLUMN: o << "c" << e->index; break;
    case EK_LITERAL:
      o << "l" << e->type << ":";
      if (!e->valid) {
        o << "null";
      } else if (e->type == QE_TYPE_FLOAT64) {
        uint64_t bits;
        std::memcpy(&bits, &e->fval, 8);
        o << std::hex << bits << std::dec;
      } else if (e->type == QE_TYPE_UTF8) {
        o << e->sval.size() << ":" << e->sval;
      } else {
        o << e->ival;
      }
      break;
    case EK_BINARY:
      if (e->to_f64) o << "f64(" << expr_identity(e->l) << ")";
      else if (e->fix) o << "fix" << e->fix << "(" << expr_identity(e->l) << "," << e->ival << ")";
      else o << "(" << expr_identity(e->l) << " " << e->op << " " << expr_identity(e->r) << ")";
      break;
    case EK_HASH_UTF8: o << "hash(" << expr_identity(e->l) << ")"; break;
    case EK_KEY_HASH: o << "keyhash(" << expr_identity(e->l) << "," << expr_identity(e->r) << ")"; break;
    case EK_ROWID: o << "rowid(" << e->ival << ")"; break;
  }
  return o.str();
}

bool expr_is_scalar(const ExprPtr &e) {
  switch (e->kind) {
    case EK_COLUMN: return false;
    case EK_LITERAL: return true;
    case EK_HASH_UTF8:
    case EK_KEY_HASH:
    case EK_ROWID: return false;
    default: return expr_is_scalar(e->l) && (e->to_f64 || e->fix || expr_is_scalar(e->r));
  }
}

static bool is_math(int op) { return op >= QE_OP_ADD && op <= QE_OP_MOD; }
static bool is_cmp(int op) { return op >= QE_OP_EQ && op <= QE_OP_LTEQ; }
static bool is_bool(int op) { return op == QE_OP_AND || op == QE_OP_OR; }

int expr_type(const ExprPtr &e, const Schema &schema) {
  switch (e->kind) {
    case EK_COLUMN:
      // out-of-range index / unsupported type: physical_expressions.rs:57-60
      if (e->index < 0 || e->index >= (int)schema.size()) fail(QE_ERR_PRIMITIVE_TYPE_NOT_SUPPORTED, "Primitive type `Int32` is not supported.");
      return schema[e->index].type;
    case EK_LITERAL:
      return e->type;
    case EK_HASH_UTF8:
      if (expr_type(e->l, schema) != QE_TYPE_UTF8) fail(QE_ERR_INTERNAL, "hash of a non-utf8 column");
      return QE_TYPE_INT64;
    case EK_ROWID:
      return QE_TYPE_INT64;
    case EK_KEY_HASH: {
      if (e->l) (void)expr_type(e->l, schema);
      const int t = expr_type(e->r, schema);
      if (e->r->kind != EK_COLUMN || !(t == QE_TYPE_INT32 || t == QE_TYPE_INT64 || t == QE_TYPE_BOOLEAN || t == QE_TYPE_UTF8))
        fail(QE_ERR_INTERNAL, "key hash of something that is not a key column");
      return QE_TYPE_INT64;
    }
    default: {
      int lt = expr_type(e->l, schema);
      if (e->to_f64) {
        if (lt != QE_TYPE_INT32 && lt != QE_TYPE_INT64 && lt != QE_TYPE_FLOAT64)
          fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(lt) + "` is not supported.");
        return QE_TYPE_FLOAT64;
      }
      if (e->fix) {
        if (lt != QE_TYPE_FLOAT64) fail(QE_ERR_INTERNAL, "fixed-point image of something that is not Float64");
        return QE_TYPE_INT64;
      }
      int rt = expr_type(e->r, schema);
      if (lt != rt)
        fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(lt) + "` is not supported (operands of `" +
                                                      op_symbol(e->op) + "` are " + type_name(lt) + " and " + type_name(rt) + ").");
      if (is_math(e->op)) {
        bool ok = lt == QE_TYPE_INT32 || lt == QE_TYPE_INT64 || (lt == QE_TYPE_FLOAT64 && e->op != QE_OP_MOD);
        if (!ok) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(lt) + "` is not supported.");
        return lt;
      }
      if (is_cmp(e->op)) {
        bool eqne = e->op == QE_OP_EQ || e->op == QE_OP_NEQ;
        bool ok = lt == QE_TYPE_INT32 || lt == QE_TYPE_INT64 || lt == QE_TYPE_FLOAT64 ||
                  (eqne && (lt == QE_TYPE_BOOLEAN || lt == QE_TYPE_UTF8));
        if (!ok) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(lt) + "` is not supported.");
        return QE_TYPE_BOOLEAN;
      }
      if (is_bool(e->op)) {
        if (lt != QE_TYPE_BOOLEAN) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(lt) + "` is not supported.");
        return QE_TYPE_BOOLEAN;
      }
      fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression `" + expr_to_string(e) + "` is not supported.");
    }
  }
}

bool expr_nullable(const ExprPtr &e, const Schema &schema) {
  switch (e->kind) {
    case EK_COLUMN: return e->index >= 0 && e->index < (int)schema.size() && schema[e->index].nullable;
    case EK_LITERAL: return !e->valid;
    case EK_HASH_UTF8: return expr_nullable(e->l, schema);
    case EK_KEY_HASH: return (e->l && expr_nullable(e->l, schema)) || expr_nullable(e->r, schema);
    case EK_ROWID: return false;
    default: {
      if (e->to_f64 || e->fix) return expr_nullable(e->l, schema);
      bool n = expr_nullable(e->l, schema) || expr_nullable(e->r, schema);
      if ((e->op == QE_OP_DIV || e->op == QE_OP_MOD) && expr_type(e->l, schema) != QE_TYPE_FLOAT64) n = true;
      return n;
    }
  }
}

ExprPtr expr_substitute(const ExprPtr &e, const std::vector<ExprPtr> &outputs) {
  switch (e->kind) {
    case EK_COLUMN:
      if (e->index < 0 || e->index >= (int)outputs.size()) fail(QE_ERR_PRIMITIVE_TYPE_NOT_SUPPORTED, "Primitive type `Int32` is not supported.");
      return outputs[e->index];
    case EK_LITERAL:
    case EK_ROWID:
      return e;
    default: {
      auto n = std::make_shared<Expr>(*e);
      if (e->l) n->l = expr_substitute(e->l, outputs);
      if (e->r) n->r = expr_substitute(e->r, outputs);
      return n;
    }
  }
}

// (Scalar, Scalar) arms.  Math: f64 x f64 or i32 x i32 (physical_expressions.rs:270-320),
// plus i64 x i64; anything else is PhysicalTypeNotSuported.  A null operand gives a null
// result (:289-292).  Integer ops wrap (Rust release arithmetic).
static ExprPtr fold_scalar(int op, const ExprPtr &a, const ExprPtr &b) {
  if (is_math(op)) {
    if (a->type != b->type || !(a->type == QE_TYPE_FLOAT64 || a->type == QE_TYPE_INT32 || a->type == QE_TYPE_INT64) ||
        (a->type == QE_TYPE_FLOAT64 && op == QE_OP_MOD))
      fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(a->type) + "` is not supported.");
    if (!a->valid || !b->valid) return make_null(a->type);
    if (a->type == QE_TYPE_FLOAT64) {
      double x = a->fval, y = b->fval, r = 0;
      switch (op) {
        case QE_OP_ADD: r = x + y; break;
        case QE_OP_SUB: r = x - y; break;
        case QE_OP_MUL: r = x * y; break;
        default: r = x / y; break;
      }
      return make_lit_f64(r);
    }
    bool w32 = a->type == QE_TYPE_INT32;
    uint64_t x = (uint64_t)a->ival, y = (uint64_t)b->ival, r = 0;
    switch (op) {
      case QE_OP_ADD: r = x + y; break;
      case QE_OP_SUB: r = x - y; break;
      case QE_OP_MUL: r = x * y; break;
      default: {
        int64_t sx = a->ival, sy = b->ival;
        int64_t mn = w32 ? (int64_t)INT32_MIN : INT64_MIN;
        if (sy == 0 || (sx == mn && sy == -1)) return make_null(a->type);  // the reference panics here
        r = (uint64_t)(op == QE_OP_DIV ? sx / sy : sx % sy);
        break;
      }
    }
    return w32 ? make_lit_i32((int32_t)(uint32_t)r) : make_lit_i64((int64_t)r);
  }
  if (is_cmp(op)) {
    bool eqne = op == QE_OP_EQ || op == QE_OP_NEQ;
    if (eqne) {
      // `dyn Scalar` PartialEq: equal iff same type, same validity and same value
      bool eq = a->type == b->type && a->valid == b->valid;
      if (eq && a->valid) {
        switch (a->type) {
          case QE_TYPE_FLOAT64: eq = a->fval == b->fval; break;
          case QE_TYPE_UTF8: eq = a->sval == b->sval; break;
          default: eq = a->ival == b->ival; break;
        }
      }
      return make_lit_bool(op == QE_OP_EQ ? eq : !eq);
    }
    if (a->type != b->type || !(a->type == QE_TYPE_FLOAT64 || a->type == QE_TYPE_INT32 || a->type == QE_TYPE_INT64))
      fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(a->type) + "` is not supported.");
    if (!a->valid || !b->valid) return make_null(QE_TYPE_BOOLEAN);
    bool r;
    if (a->type == QE_TYPE_FLOAT64) {
      double x = a->fval, y = b->fval;
      r = op == QE_OP_GT ? x > y : op == QE_OP_GTEQ ? x >= y : op == QE_OP_LT ? x < y : x <= y;
    } else {
      int64_t x = a->ival, y = b->ival;
      r = op == QE_OP_GT ? x > y : op == QE_OP_GTEQ ? x >= y : op == QE_OP_LT ? x < y : x <= y;
    }
    return make_lit_bool(r);
  }
  if (is_bool(op)) {
    if (a->type != QE_TYPE_BOOLEAN || b->type != QE_TYPE_BOOLEAN)
      fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(a->type) + "` is not supported.");
    if (!a->valid || !b->valid) return make_null(QE_TYPE_BOOLEAN);
    return make_lit_bool(op == QE_OP_AND ? (a->ival && b->ival) : (a->ival || b->ival));
  }
  fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression is not supported.");
}

ExprPtr expr_normalize(const ExprPtr &e, uint32_t flags) {
  if (e->kind != EK_BINARY) return e;
  ExprPtr l = expr_normalize(e->l, flags);
  if (e->to_f64) {
    if (l->kind == EK_LITERAL) {
      if (!l->valid) return make_null(QE_TYPE_FLOAT64);
      return make_lit_f64(l->type == QE_TYPE_FLOAT64 ? l->fval : (double)l->ival);
    }
    auto n = std::make_shared<Expr>(*e);
    n->l = l;
    return n;
  }
  if (e->fix) {  // (not folded over a literal: the kernels compute it)
    auto n = std::make_shared<Expr>(*e);
    n->l = l;
    return n;
  }
  ExprPtr r = expr_normalize(e->r, flags);
  if (l->kind == EK_LITERAL && r->kind == EK_LITERAL) return fold_scalar(e->op, l, r);
  // (Scalar, Array): the reference calls `array OP scalar` (physical_expressions.rs:264-268)
  if ((flags & QE_FLAG_REFERENCE_SCALAR_ARRAY_SWAP) && l->kind == EK_LITERAL && r->kind != EK_LITERAL &&
      (e->op == QE_OP_SUB || e->op == QE_OP_DIV || e->op == QE_OP_MOD))
    std::swap(l, r);
  auto n = std::make_shared<Expr>(*e);
  n->l = l;
  n->r = r;
  return n;
}

// ----------------------------------------------------------------- compiler
Compiler::Compiler(Context *c, const Batch *b) : ctx(c), batch(b), schema(schema_of(*b)) {}

void expr_columns(const ExprPtr &e, std::map<int, int> &seen) {
  if (!e) return;
  if (e->kind == EK_COLUMN) seen[e->index] = 1;
  if (e->kind == EK_HASH_UTF8 || e->kind == EK_KEY_HASH || e->kind == EK_BINARY) {
    expr_columns(e->l, seen);
    expr_columns(e->r, seen);
  }
}

ExprCost expr_cost(const ExprPtr &e, const Schema &schema) {
  ExprCost c;
  switch (e->kind) {
    case EK_COLUMN:
    case EK_HASH_UTF8: c.insns = 1, c.slots = 1; return c;
    case EK_LITERAL:
      if (e->valid) c.insns = 1, c.lits = 1, c.slots = 1;
      else c.insns = 3, c.lits = 1, c.slots = 2;  // a null is made by dividing zero by zero
      return c;
    case EK_ROWID:
      c.insns = e->ival ? 3 : 1, c.lits = e->ival ? 1 : 0, c.slots = e->ival ? 2 : 1;
      return c;
    case EK_KEY_HASH: {
      ExprCost before = e->l ? expr_cost(e->l, schema) : ExprCost{};
      ExprCost part;
      if (expr_type(e->r, schema) == QE_TYPE_UTF8) part.insns = 1, part.slots = 1;
      else part = expr_cost(e->r, schema);
      c.insns = before.insns + part.insns + (e->l ? 4 : 3);
      c.lits = before.lits + part.lits + 1;
      c.slots = std::max(e->l ? std::max(before.slots, 2) : 1, 1 + part.slots);
      return c;
    }
    default: break;
  }
  if (e->to_f64 || e->fix) {
    c = expr_cost(e->l, schema);
    c.insns += 1;
    if (e->fix >= FIX_LIMB0 && e->fix <= FIX_LIMB2) c.lits += 1;
    if (e->fix == FIX_SPECIAL) c.lits += 2;  // the scale and the overflow word's address
    return c;
  }
  if (expr_type(e->l, schema) == QE_TYPE_UTF8) {  // one comparison instruction; a literal operand is (address, length)
    c.insns = 1, c.lits = 3, c.slots = 1;
    return c;
  }
  const ExprCost a = expr_cost(e->l, schema), b = expr_cost(e->r, schema);
  c.insns = a.insns + b.insns + 2;  // the operation, and a swap when the right operand goes first
  c.lits = a.lits + b.lits;
  c.slots = a.slots == b.slots ? a.slots + 1 : std::max(a.slots, b.slots);
  return c;
}

bool Compiler::fits(const ExprPtr &e) const {
  const ExprCost c = expr_cost(e, schema);
  if (n_insn + c.insns + 1 > VM_MAX_INSN) return false;
  if (n_lits + c.lits > VM_MAX_LITS) return false;
  std::map<int, int> seen;
  expr_columns(e, seen);
  int extra = 0;
  for (auto &kv : seen)
    if (!colmap.count(kv.first)) ++extra;
  return n_cols + extra <= VM_MAX_COLS;
}

void Compiler::push_insn(uint32_t op, uint32_t depth, uint32_t arg) {
  if (n_insn >= VM_MAX_INSN) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression is too long for one kernel.");
  prog.insn[n_insn++] = vm_insn(op, depth, arg);
}

int Compiler::col_slot(int index) {
  auto it = colmap.find(index);
  if (it != colmap.end()) return it->second;
  if (n_cols >= VM_MAX_COLS) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Too many columns for one kernel.");
  const Column &c = batch->cols[index];
  if (c.space != QE_MEM_DEVICE) fail(QE_ERR_INVALID_ARGUMENT, "batch is not device resident (call qe_batch_to_device first)");
  VmCol &v = prog.cols[n_cols];
  v.values = c.values_ptr();
  v.validity = c.validity_ptr();
  v.offsets = c.offsets_ptr();
  v.reserved = 0;
  colmap[index] = n_cols;
  return n_cols++;
}

int Compiler::lit_slot(uint64_t bits) {
  for (int i = 0; i < n_lits; ++i)
    if (prog.lits[i] == bits) return i;
  if (n_lits >= VM_MAX_LITS) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Too many literals for one kernel.");
  prog.lits[n_lits] = bits;
  return n_lits++;
}

// Registers (stack slots) an expression needs when its deeper operand is evaluated first.
int Compiler::need(const ExprPtr &e) const { return expr_cost(e, schema).slots; }

static uint32_t math_op(int op, int type) {
  int base = type == QE_TYPE_INT32 ? OP_ADD_I32 : type == QE_TYPE_INT64 ? OP_ADD_I64 : OP_ADD_F64;
  return (uint32_t)(base + (op - QE_OP_ADD));
}
static uint32_t cmp_op(int op, int type) {
  int base = type == QE_TYPE_FLOAT64 ? OP_EQ_F64 : OP_EQ_I64;
  switch (op) {
    case QE_OP_EQ: return base + 0;
    case QE_OP_NEQ: return base + 1;
    case QE_OP_LT: return base + 2;
    case QE_OP_LTEQ: return base + 3;
    case QE_OP_GT: return base + 4;
    default: return base + 5;  // GTEQ
  }
}
static int mirror_cmp(int op) {
  switch (op) {
    case QE_OP_LT: return QE_OP_GT;
    case QE_OP_LTEQ: return QE_OP_GTEQ;
    case QE_OP_GT: return QE_OP_LT;
    case QE_OP_GTEQ: return QE_OP_LTEQ;
    default: return op;
  }
}

void Compiler::emit(const ExprPtr &e, int depth) {
  if (depth >= VM_MAX_DEPTH) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression `" + expr_to_string(e) + "` is nested too deeply.");
  switch (e->kind) {
    case EK_COLUMN: {
      int t = expr_type(e, schema);
      if (t == QE_TYPE_UTF8) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Physical type `Utf8` is not supported in a computed expression.");
      int slot = col_slot(e->index);
      uint32_t kind = t == QE_TYPE_INT32 ? CK_32 : t == QE_TYPE_BOOLEAN ? CK_BOOL : CK_64;
      auto it = cachemap.find(e->index);
      int cs = -1;
      if (it != cachemap.end()) {
        cs = it->second;
      } else if ((int)prog.cache_n < VM_NCACHE) {
        cs = (int)prog.cache_n++;
        prog.cache_col[cs] = (uint8_t)slot;
        prog.cache_kind[cs] = (uint8_t)kind;
        cachemap[e->index] = cs;
      }
      if (cs >= 0) {
        last_cache_mask |= 1u << cs;
        all_cache_mask |= 1u << cs;
        push_insn(kind == CK_32 ? OP_PUSH_CACHED32 : kind == CK_BOOL ? OP_PUSH_CACHEDBOOL : OP_PUSH_CACHED64, depth, (uint32_t)cs);
        return;
      }
      switch (t) {
        case QE_TYPE_INT32: push_insn(OP_PUSH_COL32, depth, slot); break;
        case QE_TYPE_INT64:
        case QE_TYPE_FLOAT64: push_insn(OP_PUSH_COL64, depth, slot); break;
        default: push_insn(OP_PUSH_COLBOOL, depth, slot); break;
      }
      return;
    }
    case EK_LITERAL: {
      if (e->type == QE_TYPE_UTF8) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Physical type `Utf8` is not supported in a computed expression.");
      if (!e->valid) {
        // A null scalar as an operand of arithmetic or of a comparison makes the whole result null (the reference's
        // scalar arms, physical_expressions.rs:194-203 and :258-268, hand the scalar's `None` to arrow2); the one way this machine makes a
        // null out of nothing is an integer division by zero, which leaves zero bits with the validity cleared.
        if (depth + 1 >= VM_MAX_DEPTH) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression is nested too deeply.");
        const uint32_t zero = (uint32_t)lit_slot(0);
        push_insn(OP_PUSH_LIT, depth, zero);
        push_insn(OP_PUSH_LIT, depth + 1, zero);
        push_insn(OP_DIV_I64, depth + 2, 0);
        return;
      }
      uint64_t bits;
      if (e->type == QE_TYPE_FLOAT64) std::memcpy(&bits, &e->fval, 8);
      else bits = (uint64_t)e->ival;
      push_insn(OP_PUSH_LIT, depth, lit_slot(bits));
      return;
    }
    case EK_HASH_UTF8: {
      if (e->l->kind != EK_COLUMN) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "A Utf8 group key must be a column.");
      push_insn(OP_HASH_UTF8, depth, (uint32_t)col_slot(e->l->index));
      return;
    }
    case EK_KEY_HASH: {
      // h = mix64(h_before * K + part), h_first = mix64(part + K); part = the column's value (a string's hash)
      const uint64_t K = 0x9E3779B97F4A7C15ull;
      if (depth + 1 >= VM_MAX_DEPTH) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression is nested too deeply.");
      if (e->l) {
        emit(e->l, depth);
        push_insn(OP_PUSH_LIT, depth + 1, lit_slot(K));
        push_insn(OP_MUL_I64, depth + 2, 0);
      } else {
        push_insn(OP_PUSH_LIT, depth, lit_slot(K));
      }
      if (expr_type(e->r, schema) == QE_TYPE_UTF8) push_insn(OP_HASH_UTF8, depth + 1, (uint32_t)col_slot(e->r->index));
      else emit(e->r, depth + 1);
      push_insn(OP_ADD_I64, depth + 2, 0);
      push_insn(OP_MIX64, depth + 1, 0);
      return;
    }
    case EK_ROWID: {
      push_insn(OP_PUSH_ROWID, depth, 0);
      if (e->ival != 0) {
        if (depth + 1 >= VM_MAX_DEPTH) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression is nested too deeply.");
        push_insn(OP_PUSH_LIT, depth + 1, lit_slot((uint64_t)e->ival));
        push_insn(OP_ADD_I64, depth + 2, 0);
      }
      return;
    }
    default: break;
  }
  if (e->to_f64) {
    emit(e->l, depth);
    if (expr_type(e->l, schema) != QE_TYPE_FLOAT64) push_insn(OP_I64_TO_F64, depth + 1, 0);
    return;
  }
  if (e->fix) {
    (void)expr_type(e, schema);
    emit(e->l, depth);
    switch (e->fix) {
      case FIX_EXPONENT: push_insn(OP_FIX_EXP, depth + 1, 0); break;
      case FIX_SPECIAL: {
        const uint32_t scale = (uint32_t)lit_slot((uint64_t)e->ival), word = (uint32_t)lit_slot(e->fix_overflow_word);
        push_insn(OP_FIX_SPECIAL, depth + 1, scale | (word << 8));
        break;
      }
      default: push_insn(OP_FIX_LIMB0 + (uint32_t)(e->fix - FIX_LIMB0), depth + 1, (uint32_t)lit_slot((uint64_t)e->ival)); break;
    }
    return;
  }
  int lt = expr_type(e->l, schema);
  (void)expr_type(e, schema);  // operand type checks
  if (lt == QE_TYPE_UTF8) {
    bool ne = e->op == QE_OP_NEQ;
    const ExprPtr &a = e->l, &b = e->r;
    if (a->kind == EK_COLUMN && b->kind == EK_COLUMN) {
      push_insn(ne ? OP_UTF8_NE_COL : OP_UTF8_EQ_COL, depth, (uint32_t)col_slot(a->index) | ((uint32_t)col_slot(b->index) << 8));
      return;
    }
    const ExprPtr &col = a->kind == EK_COLUMN ? a : b;
    const ExprPtr &litv = a->kind == EK_COLUMN ? b : a;
    if (col->kind != EK_COLUMN || litv->kind != EK_LITERAL)
      fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression `" + expr_to_string(e) + "` is not supported.");
    if (!litv->valid) fail(QE_ERR_SCALAR_TO_ARRAY, "Couldn't convert Scalar value `null` to array.");
    if (litv->sval.size() <= 16) {
      // a short literal travels in the program's literals (two words of bytes and the length): no device copy, and
      // the comparison is one the shape-specialised kernels implement
      if (n_lits + 3 > VM_MAX_LITS) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Too many literals for one kernel.");
      uint64_t w[2] = {0, 0};
      for (size_t i = 0; i < litv->sval.size(); ++i) w[i >> 3] |= (uint64_t)(unsigned char)litv->sval[i] << (8 * (i & 7));
      const int li = n_lits;
      prog.lits[n_lits++] = w[0];
      prog.lits[n_lits++] = w[1];
      prog.lits[n_lits++] = (uint64_t)litv->sval.size();
      push_insn(ne ? OP_UTF8_NE_LIT16 : OP_UTF8_EQ_LIT16, depth, (uint32_t)col_slot(col->index) | ((uint32_t)li << 8));
      return;
    }
    if (!ctx) fail(QE_ERR_INVALID_ARGUMENT, "a utf8 literal needs a device context");
    BufferPtr buf = device_buffer(ctx, litv->sval.size() + 1);
    QE_HIP(hipMemcpyAsync(buf->ptr, litv->sval.data(), litv->sval.size(), hipMemcpyHostToDevice, ctx->stream));
    keepalive.push_back(buf);
    if (n_lits + 2 > VM_MAX_LITS) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Too many literals for one kernel.");
    int li = n_lits;
    prog.lits[n_lits++] = (uint64_t)(uintptr_t)buf->ptr;
    prog.lits[n_lits++] = (uint64_t)litv->sval.size();
    push_insn(ne ? OP_UTF8_NE_LIT : OP_UTF8_EQ_LIT, depth, (uint32_t)col_slot(col->index) | ((uint32_t)li << 8));
    return;
  }
  int op = e->op;
  // AND / OR broadcast a scalar operand into an array first, which a null scalar cannot be (columnar_value.rs:33);
  // arithmetic and comparisons hand the scalar to arrow2's *_scalar kernels, where a null scalar gives an all-null result
  if (is_bool(op))
    for (const ExprPtr &x : {e->l, e->r})
      if (x->kind == EK_LITERAL && !x->valid) fail(QE_ERR_SCALAR_TO_ARRAY, "Couldn't convert Scalar value `null` to array.");
  int na = need(e->l), nb = need(e->r);
  bool swapped = nb > na;
  if (swapped) {
    emit(e->r, depth);
    emit(e->l, depth + 1);
  } else {
    emit(e->l, depth);
    emit(e->r, depth + 1);
  }
  uint32_t code;
  bool need_swap = false;
  if (is_math(op)) {
    code = math_op(op, lt);
    need_swap = swapped && (op == QE_OP_SUB || op == QE_OP_DIV || op == QE_OP_MOD);
  } else if (is_cmp(op)) {
    code = cmp_op(swapped ? mirror_cmp(op) : op, lt);
  } else {
    code = op == QE_OP_AND ? OP_AND : OP_OR;
  }
  if (need_swap) push_insn(OP_SWAP, depth + 2, 0);
  push_insn(code, depth + 2, 0);
}

uint32_t Compiler::add(const ExprPtr &e) {
  uint32_t start = (uint32_t)n_insn;
  last_cache_mask = 0;
  emit(e, 0);
  push_insn(OP_END, 0, 0);
  return start;
}

uint32_t Compiler::add_rowid() {
  uint32_t start = (uint32_t)n_insn;
  push_insn(OP_PUSH_ROWID, 0, 0);
  push_insn(OP_END, 0, 0);
  return start;
}

}  // namespace qe
