// Physical plan tree as the C ABI builds it, and the few helpers the ABI's translation units share.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "qe_core.h"
#include "qe_parquet.h"

namespace qe {

enum PlanKind { PK_SCAN = 0, PK_PROJECTION = 1, PK_SELECTION = 2, PK_AGGREGATE = 3 };

// enum PhysicalPlan { Scan, Projection, Selection, Aggregate } (physical_plan/mod.rs:34-39)
struct Plan {
  PlanKind kind = PK_SCAN;
  Context *ctx = nullptr;
  std::unique_ptr<Plan> child;
  std::vector<Batch> batches;        // scan over memory
  std::shared_ptr<ParquetFile> parquet;  // scan over a file: one batch per row group ...
  std::vector<int> parquet_columns;      // ... of these columns
  int parquet_first = 0, parquet_end = -1;  // row groups [first, end); end < 0 = all the file has
  std::vector<ExprPtr> exprs;        // projection
  ExprPtr pred;                      // selection
  std::vector<ExprPtr> group_exprs;  // aggregate
  std::vector<AggExpr> aggs;
  int mode = QE_AGG_MODE_SINGLE;
  int64_t cap_hint = 0;
};

}  // namespace qe

// Runs `f`, turning qe::Failure and other exceptions into a status + last-error message.
qe_status qe_guarded(const std::function<void()> &f);
void qe_write_string(const std::string &s, char *buf, int64_t buf_len, int64_t *needed);
const qe::Batch &qe_batch_ref(const qe_batch *batch);
qe::Context *qe_context_ref(qe_context *ctx);
qe_plan *qe_plan_wrap(std::unique_ptr<qe::Plan> plan);
