// The callers of the physical path: logical expressions, logical plans, the planner that lowers
// them to physical plans, and the DataFrame builder — the reference's logical_plan/mod.rs,
// logical_plan/logical_expression.rs, logical_plan/optimizer.rs, query_planner.rs, dataframe.rs and
// prelude.rs, over device-resident batches instead of a Parquet reader.
//
// Differences from the reference, all deliberate:
//   * Selection keeps its input's schema.  The reference derives a one-field schema from the
//     predicate (logical_plan/mod.rs:189-194), after which no column can be found by name.
//   * The planner lowers every logical expression: Gt/GtEq/Lt/LtEq/And/Or/Mod and
//     Sum/Count/Avg fail with PhysicalExpressionNotSuported in the reference (query_planner.rs:82,99).
//   * Count yields Int64 and Avg Float64 (the reference's to_field copies the input's type,
//     logical_expression.rs:366-373,409-416).
//   * Errors the reference unwrap()s into panics (unknown column while deriving a schema,
//     logical_plan/mod.rs:138-145) are returned as NoFieldInLogicalPlan.
#include <algorithm>
#include <cstring>
#include <set>
#include <sstream>

#include "qe_core.h"
#include "qe_plan.h"

namespace qe {

// ------------------------------------------------------------------ schema
struct Field {
  std::string name;
  int type;  // QE_TYPE_*
  bool nullable;
};
using Fields = std::vector<Field>;

// --------------------------------------------------------- logical expressions
enum LogicalKind { LK_COLUMN, LK_LIT_BOOL, LK_LIT_STRING, LK_LIT_INT, LK_LIT_INT64, LK_LIT_FLOAT, LK_BINARY, LK_AGGREGATE };

struct LogicalExpr;
using LogicalPtr = std::shared_ptr<const LogicalExpr>;
struct LogicalExpr {
  LogicalKind kind = LK_COLUMN;
  std::string name;  // column name / string literal
  int64_t ival = 0;
  double fval = 0;
  int op = 0;        // QE_OP_* or QE_AGG_*
  LogicalPtr l, r;
};

static const char *binary_name(int op) {
  static const char *n[] = {"?", "eq", "neq", "gt", "gteq", "lt", "lteq", "and", "or", "add", "sub", "mul", "div", "mod"};
  return n[op];
}
static const char *binary_symbol(int op) {
  static const char *n[] = {"?", "==", "!=", ">", ">=", "<", "<=", "&&", "||", "+", "-", "*", "/", "%"};
  return n[op];
}
static const char *aggregate_name(int kind) {
  static const char *n[] = {"?", "max", "min", "sum", "count", "avg"};
  return n[kind];
}

// Rust's `{}` of an f64: shortest representation that round-trips, "1" for 1.0.
static std::string float_to_string(double v) {
  char b[64];
  for (int prec = 1; prec <= 17; ++prec) {
    snprintf(b, sizeof b, "%.*g", prec, v);
    if (strtod(b, nullptr) == v) break;
  }
  return b;
}

// `impl Display` (logical_expression.rs:123-127,153-157,276-280,378-382)
static std::string logical_to_string(const LogicalPtr &e) {
  switch (e->kind) {
    case LK_COLUMN: return "#" + e->name;
    case LK_LIT_BOOL: return std::string("'") + (e->ival ? "true" : "false") + "'";
    case LK_LIT_STRING: return "'" + e->name + "'";
    case LK_LIT_INT:
    case LK_LIT_INT64: return "'" + std::to_string(e->ival) + "'";
    case LK_LIT_FLOAT: return "'" + float_to_string(e->fval) + "'";
    case LK_BINARY: return logical_to_string(e->l) + " " + binary_symbol(e->op) + " " + logical_to_string(e->r);
    default: return std::string(aggregate_name(e->op)) + " (" + logical_to_string(e->l) + ")";
  }
}

static int field_index(const Fields &schema, const std::string &name) {
  for (size_t i = 0; i < schema.size(); ++i)
    if (schema[i].name == name) return (int)i;
  return -1;
}

// `to_field` (logical_expression.rs:39-65 and the per-type impls)
static Field logical_to_field(const LogicalPtr &e, const Fields &input) {
  switch (e->kind) {
    case LK_COLUMN: {
      int i = field_index(input, e->name);
      if (i < 0) fail(QE_ERR_NO_FIELD_IN_LOGICAL_PLAN, "No field with the name `" + e->name + "` in logical plan.");
      return input[i];
    }
    case LK_LIT_BOOL: return Field{e->ival ? "true" : "false", QE_TYPE_BOOLEAN, false};
    case LK_LIT_STRING: return Field{e->name, QE_TYPE_UTF8, false};
    case LK_LIT_INT: return Field{std::to_string(e->ival), QE_TYPE_INT32, false};
    case LK_LIT_INT64: return Field{std::to_string(e->ival), QE_TYPE_INT64, false};
    case LK_LIT_FLOAT: return Field{float_to_string(e->fval), QE_TYPE_FLOAT64, false};
    case LK_BINARY:
      if (e->op >= QE_OP_ADD) return Field{binary_name(e->op), logical_to_field(e->l, input).type, false};
      (void)logical_to_field(e->l, input);
      (void)logical_to_field(e->r, input);
      return Field{binary_name(e->op), QE_TYPE_BOOLEAN, false};
    default: {
      int t = logical_to_field(e->l, input).type;
      if (e->op == QE_AGG_COUNT) t = QE_TYPE_INT64;
      if (e->op == QE_AGG_AVG) t = QE_TYPE_FLOAT64;
      return Field{aggregate_name(e->op), t, false};
    }
  }
}

static void extract_columns(const LogicalPtr &e, std::set<std::string> &out) {  // optimizer.rs:71-138
  if (e->kind == LK_COLUMN) out.insert(e->name);
  if (e->l) extract_columns(e->l, out);
  if (e->r) extract_columns(e->r, out);
}

// --------------------------------------------------------------- logical plans
enum LogicalPlanKind { LP_SCAN, LP_PROJECTION, LP_SELECTION, LP_AGGREGATE };

struct LogicalPlan;
using LogicalPlanPtr = std::shared_ptr<const LogicalPlan>;
struct LogicalPlan {
  LogicalPlanKind kind = LP_SCAN;
  Fields schema;
  LogicalPlanPtr child;
  // scan
  std::string path;
  Fields source_schema;
  std::vector<Batch> batches;
  std::shared_ptr<ParquetFile> parquet;  // a file instead of batches
  bool has_projection = false;
  std::vector<std::string> projection;
  // projection / selection / aggregate
  std::vector<LogicalPtr> exprs;       // projection exprs, or [predicate]
  std::vector<LogicalPtr> group_exprs, aggregate_exprs;
};

static LogicalPlanPtr make_scan(const std::string &path, const Fields &source_schema, const std::vector<Batch> &batches,
                                const std::shared_ptr<ParquetFile> &parquet, bool has_projection, const std::vector<std::string> &projection) {
  auto p = std::make_shared<LogicalPlan>();
  p->kind = LP_SCAN;
  p->path = path;
  p->source_schema = source_schema;
  p->batches = batches;
  p->parquet = parquet;
  p->has_projection = has_projection;
  p->projection = projection;
  // derive_schema (logical_plan/mod.rs:79-91): the source's fields the projection names, in source order
  for (auto &f : source_schema)
    if (!has_projection || std::find(projection.begin(), projection.end(), f.name) != projection.end()) p->schema.push_back(f);
  return p;
}

static LogicalPlanPtr make_projection(const LogicalPlanPtr &input, const std::vector<LogicalPtr> &exprs) {
  auto p = std::make_shared<LogicalPlan>();
  p->kind = LP_PROJECTION;
  p->child = input;
  p->exprs = exprs;
  for (auto &e : exprs) p->schema.push_back(logical_to_field(e, input->schema));
  return p;
}

static LogicalPlanPtr make_selection(const LogicalPlanPtr &input, const LogicalPtr &expr) {
  auto p = std::make_shared<LogicalPlan>();
  p->kind = LP_SELECTION;
  p->child = input;
  p->exprs = {expr};
  (void)logical_to_field(expr, input->schema);  // unknown columns surface here
  p->schema = input->schema;
  return p;
}

static LogicalPlanPtr make_aggregate(const LogicalPlanPtr &input, const std::vector<LogicalPtr> &groups, const std::vector<LogicalPtr> &aggs) {
  auto p = std::make_shared<LogicalPlan>();
  p->kind = LP_AGGREGATE;
  p->child = input;
  p->group_exprs = groups;
  p->aggregate_exprs = aggs;
  for (auto &e : groups) p->schema.push_back(logical_to_field(e, input->schema));
  for (auto &e : aggs) p->schema.push_back(logical_to_field(e, input->schema));
  return p;
}

// `impl Display` of each node and format_logical_plan (logical_plan/mod.rs:48-58,94-109,148-159,197-201,251-263)
static std::string plan_line(const LogicalPlan &p) {
  std::string s;
  switch (p.kind) {
    case LP_SCAN:
      s = "Scan: " + p.path + "; projection=";
      if (!p.has_projection) return s + "None";
      for (auto &n : p.projection) s += n + ", ";
      return s;
    case LP_PROJECTION:
      s = "Projection: ";
      for (auto &e : p.exprs) s += logical_to_string(e) + ", ";
      return s;
    case LP_SELECTION: return "Selection: " + logical_to_string(p.exprs[0]) + ", ";
    default:
      s = "Aggregate: ";
      for (auto &e : p.group_exprs) s += logical_to_string(e) + ", ";
      for (auto &e : p.aggregate_exprs) s += logical_to_string(e) + ", ";
      return s;
  }
}
static std::string format_logical_plan(const LogicalPlanPtr &p, int indent) {
  std::string s;
  for (int i = 0; i < indent; ++i) s += " \t";
  s += plan_line(*p) + " \n";
  if (p->child) s += format_logical_plan(p->child, indent + 1);
  return s;
}

// Projection push-down (optimizer.rs:15-58): the columns the operators above the scan need are collected on
// the way down; the scan then reads only those.  `cols` is what the operators above `p` need of p's output.
// Where the reference only ever adds names on the way down, here
//  * a Projection or Aggregate starts the set afresh (what lies above them names their outputs, not the scan's
//    columns), which prunes more;
//  * a plan whose root is a Selection or the Scan itself needs every column of that root - they are the query's
//    output (the reference would hand such a scan the predicate's columns only, or none at all).
static LogicalPlanPtr push_down(const LogicalPlanPtr &p, std::set<std::string> &cols, bool root) {
  switch (p->kind) {
    case LP_SCAN: {
      if (root) return p;
      if (p->has_projection) {
        // an earlier pass already narrowed the scan: nothing outside its projection exists any more
        std::set<std::string> both;
        for (auto &n : p->projection)
          if (cols.count(n)) both.insert(n);
        cols = both;
      }
      return make_scan(p->path, p->source_schema, p->batches, p->parquet, true, std::vector<std::string>(cols.begin(), cols.end()));
    }
    case LP_AGGREGATE: {
      cols.clear();
      for (auto &e : p->group_exprs) extract_columns(e, cols);
      for (auto &e : p->aggregate_exprs) extract_columns(e, cols);
      return make_aggregate(push_down(p->child, cols, false), p->group_exprs, p->aggregate_exprs);
    }
    case LP_PROJECTION: {
      cols.clear();
      for (auto &e : p->exprs) extract_columns(e, cols);
      return make_projection(push_down(p->child, cols, false), p->exprs);
    }
    default: {
      // a Selection passes its input's columns through: what is needed above it, and what its predicate reads
      if (root)
        for (auto &f : p->schema) cols.insert(f.name);
      extract_columns(p->exprs[0], cols);
      return make_selection(push_down(p->child, cols, false), p->exprs[0]);
    }
  }
}

// ------------------------------------------------------------------- planner
// to_physical_expression (query_planner.rs:15-84)
static ExprPtr to_physical_expression(const LogicalPtr &e, const Fields &input) {
  switch (e->kind) {
    case LK_COLUMN: {
      int i = field_index(input, e->name);
      if (i < 0) fail(QE_ERR_NO_FIELD_IN_LOGICAL_PLAN, "No field with the name `#" + e->name + "` in logical plan.");
      return make_column(i);
    }
    case LK_LIT_BOOL: return make_lit_bool(e->ival != 0);
    case LK_LIT_STRING: return make_lit_utf8(e->name);
    case LK_LIT_INT: return make_lit_i32((int32_t)e->ival);
    case LK_LIT_INT64: return make_lit_i64(e->ival);
    case LK_LIT_FLOAT: return make_lit_f64(e->fval);
    case LK_BINARY: return make_binary(e->op, to_physical_expression(e->l, input), to_physical_expression(e->r, input));
    default: fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression `" + logical_to_string(e) + "` is not supported.");
  }
}

// to_physical_aggregate_expression (query_planner.rs:86-101)
static AggExpr to_physical_aggregate_expression(const LogicalPtr &e, const Fields &input) {
  if (e->kind != LK_AGGREGATE) fail(QE_ERR_PHYSICAL_EXPRESSION_NOT_SUPPORTED, "Physical expression `" + logical_to_string(e) + "` is not supported.");
  return AggExpr{e->op, to_physical_expression(e->l, input)};
}

// to_physical_plan (query_planner.rs:104-166)
static std::unique_ptr<Plan> to_physical_plan(Context *ctx, const LogicalPlanPtr &p) {
  auto out = std::make_unique<Plan>();
  out->ctx = ctx;
  switch (p->kind) {
    case LP_SCAN: {
      out->kind = PK_SCAN;
      if (p->parquet) {
        // DataSource::scan(projection) (data_source.rs:48-71): the named columns, in file order
        out->parquet = p->parquet;
        if (p->has_projection) out->parquet_columns = parquet_project(*p->parquet, p->projection);
        else
          for (size_t i = 0; i < p->parquet->columns.size(); ++i) out->parquet_columns.push_back((int)i);
        return out;
      }
      for (auto &b : p->batches) {
        if (!p->has_projection) {
          out->batches.push_back(b);
          continue;
        }
        Batch nb;  // the projected columns, shared with the source batch
        nb.ctx = b.ctx;
        nb.rows = b.rows;
        for (size_t i = 0; i < p->source_schema.size(); ++i)
          if (std::find(p->projection.begin(), p->projection.end(), p->source_schema[i].name) != p->projection.end()) nb.cols.push_back(b.cols[i]);
        out->batches.push_back(nb);
      }
      return out;
    }
    case LP_PROJECTION:
      out->kind = PK_PROJECTION;
      for (auto &e : p->exprs) out->exprs.push_back(to_physical_expression(e, p->child->schema));
      break;
    case LP_SELECTION:
      out->kind = PK_SELECTION;
      out->pred = to_physical_expression(p->exprs[0], p->child->schema);
      break;
    default:
      out->kind = PK_AGGREGATE;
      for (auto &e : p->group_exprs) out->group_exprs.push_back(to_physical_expression(e, p->child->schema));
      for (auto &e : p->aggregate_exprs) out->aggs.push_back(to_physical_aggregate_expression(e, p->child->schema));
      break;
  }
  out->child = to_physical_plan(ctx, p->child);
  return out;
}

static std::string schema_json(const Fields &schema) {
  std::ostringstream o;
  o << "[";
  for (size_t i = 0; i < schema.size(); ++i) {
    std::string name;
    for (char ch : schema[i].name) {
      if (ch == '"' || ch == '\\') name += '\\';
      name += ch;
    }
    o << (i ? "," : "") << "{\"name\":\"" << name << "\",\"data_type\":\"" << type_name(schema[i].type) << "\",\"type\":" << schema[i].type
      << ",\"is_nullable\":" << (schema[i].nullable ? "true" : "false") << "}";
  }
  o << "]";
  return o.str();
}

}  // namespace qe

// ==================================================================== C ABI
using namespace qe;

struct qe_logical_expr { LogicalPtr e; };
struct qe_dataframe { Context *ctx; LogicalPlanPtr plan; };

extern "C" {

static qe_status new_logical(std::shared_ptr<LogicalExpr> e, qe_logical_expr **out) {
  return qe_guarded([&] {
    if (!out) fail(QE_ERR_INVALID_ARGUMENT, "out is null");
    *out = new qe_logical_expr{std::move(e)};
  });
}

qe_status qe_col(const char *name, qe_logical_expr **out) {
  auto e = std::make_shared<LogicalExpr>();
  e->kind = LK_COLUMN;
  e->name = name ? name : "";
  return new_logical(e, out);
}
qe_status qe_lit_string(const char *value, qe_logical_expr **out) {
  auto e = std::make_shared<LogicalExpr>();
  e->kind = LK_LIT_STRING;
  e->name = value ? value : "";
  return new_logical(e, out);
}
qe_status qe_lit_int(int32_t value, qe_logical_expr **out) {
  auto e = std::make_shared<LogicalExpr>();
  e->kind = LK_LIT_INT;
  e->ival = value;
  return new_logical(e, out);
}
qe_status qe_lit_int64(int64_t value, qe_logical_expr **out) {
  auto e = std::make_shared<LogicalExpr>();
  e->kind = LK_LIT_INT64;
  e->ival = value;
  return new_logical(e, out);
}
qe_status qe_lit_float(double value, qe_logical_expr **out) {
  auto e = std::make_shared<LogicalExpr>();
  e->kind = LK_LIT_FLOAT;
  e->fval = value;
  return new_logical(e, out);
}
qe_status qe_lit_bool(int32_t value, qe_logical_expr **out) {
  auto e = std::make_shared<LogicalExpr>();
  e->kind = LK_LIT_BOOL;
  e->ival = value != 0;
  return new_logical(e, out);
}

qe_status qe_logical_binary(int32_t op, qe_logical_expr *left, qe_logical_expr *right, qe_logical_expr **out) {
  return qe_guarded([&] {
    if (!left || !right || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    LogicalPtr l = left->e, r = right->e;
    delete left;
    delete right;
    if (op < QE_OP_EQ || op > QE_OP_MOD) fail(QE_ERR_INVALID_ARGUMENT, "unknown binary operator");
    auto e = std::make_shared<LogicalExpr>();
    e->kind = LK_BINARY;
    e->op = op;
    e->l = l;
    e->r = r;
    *out = new qe_logical_expr{e};
  });
}

qe_status qe_logical_aggregate(int32_t kind, qe_logical_expr *expr, qe_logical_expr **out) {
  return qe_guarded([&] {
    if (!expr || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    LogicalPtr l = expr->e;
    delete expr;
    if (kind < QE_AGG_MAX || kind > QE_AGG_AVG) fail(QE_ERR_INVALID_ARGUMENT, "unknown aggregate");
    auto e = std::make_shared<LogicalExpr>();
    e->kind = LK_AGGREGATE;
    e->op = kind;
    e->l = l;
    *out = new qe_logical_expr{e};
  });
}

qe_status qe_logical_expr_to_string(const qe_logical_expr *expr, char *buf, int64_t buf_len, int64_t *needed) {
  return qe_guarded([&] {
    if (!expr) fail(QE_ERR_INVALID_ARGUMENT, "expr is null");
    qe_write_string(logical_to_string(expr->e), buf, buf_len, needed);
  });
}

void qe_logical_expr_free(qe_logical_expr *expr) { delete expr; }

qe_status qe_dataframe_parquet(qe_context *ctx, const char *path, qe_dataframe **out) {
  return qe_guarded([&] {
    if (!ctx || !path || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    auto file = parquet_open(path);
    Fields schema;  // infer_schema (data_source.rs:45-47)
    for (auto &c : file->columns) schema.push_back(Field{c.name, c.type, c.optional});
    *out = new qe_dataframe{qe_context_ref(ctx), make_scan(path, schema, {}, file, false, {})};
  });
}

qe_status qe_dataframe_from_batches(qe_context *ctx, const char *label, const char *const *column_names, int32_t n_columns,
                                    qe_batch *const *batches, int32_t n_batches, qe_dataframe **out) {
  return qe_guarded([&] {
    if (!ctx || !out || n_columns < 0 || n_batches < 0 || (n_columns && !column_names) || (n_batches && !batches))
      fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    Fields schema;
    std::vector<Batch> bs;
    for (int b = 0; b < n_batches; ++b) {
      const Batch &bt = qe_batch_ref(batches[b]);
      if ((int)bt.cols.size() != n_columns) fail(QE_ERR_DIFFERENT_SIZES, "The batches and the column names have different sizes.");
      for (auto &c : bt.cols)
        if (c.space != QE_MEM_DEVICE) fail(QE_ERR_INVALID_ARGUMENT, "scan batches must be device resident (qe_batch_to_device)");
      bs.push_back(bt);
    }
    for (int i = 0; i < n_columns; ++i) {
      Field f{column_names[i] ? column_names[i] : "", 0, true};
      if (n_batches) f.type = bs[0].cols[i].type;
      for (auto &bt : bs)
        if (bt.cols[i].type != f.type) fail(QE_ERR_INVALID_ARGUMENT, "batches of one scan must share a schema");
      schema.push_back(f);  // Parquet columns are nullable in the reference's schema (tests/mod.rs:31)
    }
    *out = new qe_dataframe{qe_context_ref(ctx), make_scan(label ? label : "memory", schema, bs, nullptr, false, {})};
  });
}

// Each builder consumes `df` like the reference's `self` methods (dataframe.rs:44-78).
qe_status qe_dataframe_project(qe_dataframe *df, qe_logical_expr *const *exprs, int32_t n_exprs, qe_dataframe **out) {
  return qe_guarded([&] {
    if (!df || !out || n_exprs < 0 || (n_exprs && !exprs)) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    std::unique_ptr<qe_dataframe> own(df);
    std::vector<LogicalPtr> es;
    for (int i = 0; i < n_exprs; ++i) {
      es.push_back(exprs[i]->e);
      delete exprs[i];
    }
    *out = new qe_dataframe{own->ctx, make_projection(own->plan, es)};
  });
}

qe_status qe_dataframe_filter(qe_dataframe *df, qe_logical_expr *expr, qe_dataframe **out) {
  return qe_guarded([&] {
    if (!df || !out || !expr) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    std::unique_ptr<qe_dataframe> own(df);
    LogicalPtr e = expr->e;
    delete expr;
    *out = new qe_dataframe{own->ctx, make_selection(own->plan, e)};
  });
}

qe_status qe_dataframe_aggregate(qe_dataframe *df, qe_logical_expr *const *group_by, int32_t n_group_by,
                                 qe_logical_expr *const *aggregate_expr, int32_t n_aggregate_expr, qe_dataframe **out) {
  return qe_guarded([&] {
    if (!df || !out || n_group_by < 0 || n_aggregate_expr < 0) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    std::unique_ptr<qe_dataframe> own(df);
    std::vector<LogicalPtr> gs, as;
    for (int i = 0; i < n_group_by; ++i) {
      gs.push_back(group_by[i]->e);
      delete group_by[i];
    }
    for (int i = 0; i < n_aggregate_expr; ++i) {
      as.push_back(aggregate_expr[i]->e);
      delete aggregate_expr[i];
    }
    *out = new qe_dataframe{own->ctx, make_aggregate(own->plan, gs, as)};
  });
}

qe_status qe_dataframe_optimize(qe_dataframe *df, qe_dataframe **out) {
  return qe_guarded([&] {
    if (!df || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    std::unique_ptr<qe_dataframe> own(df);
    std::set<std::string> cols;
    *out = new qe_dataframe{own->ctx, push_down(own->plan, cols, true)};
  });
}

qe_status qe_dataframe_schema(const qe_dataframe *df, char *buf, int64_t buf_len, int64_t *needed) {
  return qe_guarded([&] {
    if (!df) fail(QE_ERR_INVALID_ARGUMENT, "df is null");
    qe_write_string(schema_json(df->plan->schema), buf, buf_len, needed);
  });
}

qe_status qe_dataframe_logical_plan(const qe_dataframe *df, char *buf, int64_t buf_len, int64_t *needed) {
  return qe_guarded([&] {
    if (!df) fail(QE_ERR_INVALID_ARGUMENT, "df is null");
    qe_write_string(format_logical_plan(df->plan, 0), buf, buf_len, needed);
  });
}

qe_status qe_dataframe_to_physical_plan(qe_dataframe *df, qe_plan **out) {
  return qe_guarded([&] {
    if (!df || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    std::unique_ptr<qe_dataframe> own(df);
    *out = qe_plan_wrap(to_physical_plan(own->ctx, own->plan));
  });
}

qe_status qe_dataframe_execute(qe_dataframe *df, qe_iter **out) {
  qe_plan *plan = nullptr;
  qe_status st = qe_dataframe_to_physical_plan(df, &plan);
  if (st != QE_OK) return st;
  return qe_plan_execute(plan, out);
}

void qe_dataframe_free(qe_dataframe *df) { delete df; }

}  // extern "C"
