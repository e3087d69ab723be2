/* The C ABI without any Python: plain C, hipMalloc'ed columns, the headline plan
 *   Scan -> Selection(c > 0.5) -> Projection(k, a + b*c) -> Aggregate(GROUP BY k; SUM, COUNT, MIN)
 * built call by call the way a host engine's binding would, and the result checked on the CPU.
 * Build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/abi_example.c \
 *            -Lquery-engine_amd -lqe_gpu -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,... -lm
 * Exit status 0 = results agree. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qe_gpu.h"

#define CHECK(call)                                                                              \
  do {                                                                                           \
    qe_status st_ = (call);                                                                      \
    if (st_ != QE_OK) {                                                                          \
      fprintf(stderr, "%s -> %s: %s\n", #call, qe_status_name(st_), qe_last_error_message());    \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)
#define HIP(call)                                                                   \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                  \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static qe_expr *col(int i) {
  qe_expr *e = NULL;
  qe_expr_column(i, &e);
  return e;
}
static qe_expr *bin(int op, qe_expr *l, qe_expr *r) {
  qe_expr *e = NULL;
  qe_expr_binary(op, l, r, &e); /* consumes l and r */
  return e;
}

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 3000001;
  if (qe_device_count() < 1) {
    fprintf(stderr, "no gfx950 device\n");
    return 2;
  }
  qe_context *ctx = NULL;
  CHECK(qe_context_create(0, NULL, &ctx));

  /* device-resident columns k:Int64 (skewed, 10 values), a, b in [-1, 1), c in [0, 1) */
  void *dev[4];
  for (int i = 0; i < 4; ++i) HIP(hipMalloc(&dev[i], (size_t)n * 8 + 64));
  CHECK(qe_generate_column(ctx, 3, 43, 0, n, 0, 0, dev[0]));
  CHECK(qe_generate_column(ctx, 0, 44, 0, n, -1.0, 1.0, dev[1]));
  CHECK(qe_generate_column(ctx, 0, 45, 0, n, -1.0, 1.0, dev[2]));
  CHECK(qe_generate_column(ctx, 0, 46, 0, n, 0.0, 1.0, dev[3]));
  CHECK(qe_context_synchronize(ctx));

  qe_column desc[4];
  memset(desc, 0, sizeof desc);
  for (int i = 0; i < 4; ++i) {
    desc[i].type = i == 0 ? QE_TYPE_INT64 : QE_TYPE_FLOAT64;
    desc[i].memory_space = QE_MEM_DEVICE;
    desc[i].length = n;
    desc[i].values = dev[i];
    desc[i].values_bytes = n * 8;
  }
  qe_batch *batch = NULL;
  CHECK(qe_batch_import(ctx, desc, 4, &batch)); /* borrows the buffers */

  qe_plan *scan = NULL, *sel = NULL, *proj = NULL, *agg = NULL;
  CHECK(qe_plan_scan_memory(ctx, &batch, 1, &scan));
  qe_expr *half = NULL;
  CHECK(qe_expr_lit_f64(0.5, &half));
  CHECK(qe_plan_selection(scan, bin(QE_OP_GT, col(3), half), &sel));
  qe_expr *outs[2] = {col(0), bin(QE_OP_ADD, col(1), bin(QE_OP_MUL, col(2), col(3)))};
  CHECK(qe_plan_projection(sel, outs, 2, &proj));
  qe_expr *group = col(0);
  qe_agg_expr *aggs[3];
  CHECK(qe_agg_expr_new(QE_AGG_SUM, col(1), &aggs[0]));
  CHECK(qe_agg_expr_new(QE_AGG_COUNT, NULL, &aggs[1]));
  CHECK(qe_agg_expr_new(QE_AGG_MIN, col(1), &aggs[2]));
  CHECK(qe_plan_aggregate(proj, &group, 1, aggs, 3, QE_AGG_MODE_SINGLE, 64, &agg));
  char text[1024];
  CHECK(qe_plan_to_string(agg, text, sizeof text, NULL));
  printf("%s", text);

  qe_iter *it = NULL;
  CHECK(qe_plan_execute(agg, &it)); /* consumes the plan */
  qe_batch *out = NULL, *host = NULL;
  CHECK(qe_iter_next(it, &out));
  CHECK(qe_batch_to_host(ctx, out, &host));
  int64_t groups = 0;
  CHECK(qe_batch_num_rows(host, &groups));
  qe_column ck, cs, cc, cm;
  CHECK(qe_batch_column(host, 0, &ck));
  CHECK(qe_batch_column(host, 1, &cs));
  CHECK(qe_batch_column(host, 2, &cc));
  CHECK(qe_batch_column(host, 3, &cm));
  qe_batch *none = NULL;
  if (qe_iter_next(it, &none) != QE_END) {
    fprintf(stderr, "an aggregate yields exactly one batch\n");
    return 1;
  }

  /* the same query on the CPU */
  int64_t *k = malloc((size_t)n * 8);
  double *a = malloc((size_t)n * 8), *b = malloc((size_t)n * 8), *c = malloc((size_t)n * 8);
  HIP(hipMemcpy(k, dev[0], (size_t)n * 8, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(a, dev[1], (size_t)n * 8, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(b, dev[2], (size_t)n * 8, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(c, dev[3], (size_t)n * 8, hipMemcpyDeviceToHost));
  double sum[16] = {0}, mag[16] = {0}, mn[16];
  int64_t cnt[16] = {0};
  for (int g = 0; g < 16; ++g) mn[g] = INFINITY;
  for (int64_t i = 0; i < n; ++i) {
    if (!(c[i] > 0.5)) continue;
    volatile double t = b[i] * c[i]; /* two roundings, like two operator passes */
    double v = a[i] + t;
    int g = (int)k[i];
    sum[g] += v;
    mag[g] += fabs(v);
    cnt[g] += 1;
    if (v < mn[g]) mn[g] = v;
  }
  int bad = 0, seen = 0;
  for (int64_t r = 0; r < groups; ++r) {
    int g = (int)((const int64_t *)ck.values)[r];
    double s = ((const double *)cs.values)[r], m = ((const double *)cm.values)[r];
    int64_t q = ((const int64_t *)cc.values)[r];
    int ok = g >= 0 && g < 16 && q == cnt[g] && m == mn[g] && fabs(s - sum[g]) <= 1e-12 * mag[g];
    printf("k=%d count=%lld sum=%.6f min=%.6f %s\n", g, (long long)q, s, m, ok ? "ok" : "MISMATCH");
    bad += !ok;
    seen += 1;
  }
  for (int g = 0; g < 16; ++g) seen -= cnt[g] > 0;
  if (seen != 0) {
    fprintf(stderr, "group sets differ\n");
    bad += 1;
  }

  qe_batch_free(host);
  qe_batch_free(out);
  qe_iter_free(it);
  qe_batch_free(batch);
  qe_context_free(ctx);
  for (int i = 0; i < 4; ++i) HIP(hipFree(dev[i]));
  free(k), free(a), free(b), free(c);
  printf(bad ? "FAILED\n" : "C ABI example ok: %lld rows, %lld groups\n", (long long)n, (long long)groups);
  return bad ? 1 : 0;
}
