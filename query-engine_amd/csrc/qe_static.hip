// Registry of the kernels built ahead of time for one exact launch shape each.  The shapes come
// from qe_static_specs.h, which tools/gen_static_specs.py writes by running the host lowering
// (qe_lowering_describe) on a list of plans: no shape is written by hand, so a Spec can never
// disagree with what the host would hand the interpreter for the same plan.
#include <cstring>
#include <string>

#include "qe_kernels.h"
#include "qe_signature.h"
#include "qe_static.h"
#include "qe_static_specs.h"

namespace qe {
namespace {

struct Entry {
  const char *name;
  const uint32_t *sig;
  int n_sig;
  void (*project)(const ProjectArgs &, int, hipStream_t);
  void (*filter)(const FilterArgs &, int, hipStream_t);
  void (*aggregate)(const AggArgs &, int, int, size_t, hipStream_t);
  void (*aggregate_spill)(const AggArgs &, int, int, size_t, hipStream_t);  // the high-cardinality mode's first pass
  const char *(*spill_name)();
  void (*aggregate_direct)(const AggArgs &, int, int, size_t, hipStream_t);  // ... the one that partitions its rows itself
  const char *(*direct_name)();
};

template <class Spec>
void run_project(const ProjectArgs &a, int grid, hipStream_t s) {
  hipLaunchKernelGGL(qe_project_static<Spec>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, s, a);
}
template <class Spec>
void run_filter(const FilterArgs &a, int cus, hipStream_t s) {
  // tiles are larger than the interpreter kernel's (whose count sized the status array: enough),
  // and one workgroup per compute unit keeps the look-back walks short
  using G = FilterGeometry<Spec>;
  FilterArgs b = a;
  b.n_tiles = (a.n_rows + G::TILE_ROWS - 1) / G::TILE_ROWS;
  int64_t grid = b.n_tiles < cus ? b.n_tiles : cus;
  hipLaunchKernelGGL(qe_filter_static<Spec>, dim3((unsigned)grid), dim3(G::WAVES * 64), 0, s, b);
}
template <class Spec>
void run_aggregate(const AggArgs &a, int grid, int block_threads, size_t lds, hipStream_t s) {
  if (lds > 64 * 1024) {
    static bool raised = false;  // per kernel: this function exists once per Spec
    if (!raised) {
      (void)hipFuncSetAttribute((const void *)qe_aggregate_static<Spec>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipGetLastError();  // a runtime that takes no such attribute simply allows what the device has
      raised = true;
    }
  }
  hipLaunchKernelGGL(qe_aggregate_static<Spec>, dim3(grid), dim3(block_threads), lds, s, a);
}

template <class Spec>
void run_aggregate_spill(const AggArgs &a, int grid, int block_threads, size_t lds, hipStream_t s) {
  if (lds > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute((const void *)qe_aggregate_spill_static<Spec>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipGetLastError();
      raised = true;
    }
  }
  hipLaunchKernelGGL(qe_aggregate_spill_static<Spec>, dim3(grid), dim3(block_threads), lds, s, a);
}
template <class Spec>
const char *spill_kernel_name() {
  // "qe_aggregate_static<shape>" -> "qe_aggregate_spill_static<shape>"
  static const std::string name = std::string("qe_aggregate_spill") + (Spec::name + std::strlen("qe_aggregate"));
  return name.c_str();
}

template <class Spec>
void run_aggregate_direct(const AggArgs &a, int grid, int block_threads, size_t lds, hipStream_t s) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute((const void *)qe_aggregate_direct_static<Spec>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    raised = true;
  }
  hipLaunchKernelGGL(qe_aggregate_direct_static<Spec>, dim3(grid), dim3(block_threads), lds, s, a);
}
template <class Spec>
const char *direct_kernel_name() {
  static const std::string name = std::string("qe_aggregate_direct") + (Spec::name + std::strlen("qe_aggregate"));
  return name.c_str();
}

#define QE_ENTRY_PROJECT(S) Entry{S::name, S::sig, S::n_sig, run_project<S>, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
#define QE_ENTRY_FILTER(S) Entry{S::name, S::sig, S::n_sig, nullptr, run_filter<S>, nullptr, nullptr, nullptr, nullptr, nullptr},
#define QE_ENTRY_AGGREGATE(S) \
  Entry{S::name, S::sig, S::n_sig, nullptr, nullptr, run_aggregate<S>, run_aggregate_spill<S>, spill_kernel_name<S>, run_aggregate_direct<S>, direct_kernel_name<S>},

const Entry g_entries[] = {
    QE_STATIC_PROJECT_SPECS(QE_ENTRY_PROJECT) QE_STATIC_FILTER_SPECS(QE_ENTRY_FILTER)
        QE_STATIC_AGGREGATE_SPECS(QE_ENTRY_AGGREGATE) Entry{nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};

const Entry *find(const std::vector<uint32_t> &sig) {
  for (const Entry *e = g_entries; e->name; ++e)
    if ((size_t)e->n_sig == sig.size() && std::memcmp(e->sig, sig.data(), sig.size() * sizeof(uint32_t)) == 0) return e;
  return nullptr;
}

}  // namespace

bool static_launch_project(const ProjectArgs &args, int grid, hipStream_t stream, hipError_t *err, const char **kernel_name) {
  const Entry *e = find(signature(args));
  if (!e || !e->project) return false;
  e->project(args, grid, stream);
  *err = hipGetLastError();
  if (kernel_name) *kernel_name = e->name;
  return true;
}

bool static_launch_filter(const FilterArgs &args, int cus, hipStream_t stream, hipError_t *err, const char **kernel_name) {
  const Entry *e = find(signature(args));
  if (!e || !e->filter) return false;
  e->filter(args, cus, stream);
  *err = hipGetLastError();
  if (kernel_name) *kernel_name = e->name;
  return true;
}

bool static_has_aggregate(const AggArgs &args) {
  const Entry *e = find(signature(args));
  return e && e->aggregate;
}

bool static_launch_aggregate(const AggArgs &args, int grid, int block_threads, size_t lds_bytes, hipStream_t stream, hipError_t *err,
                             const char **kernel_name, int first_pass) {
  const Entry *e = find(signature(args));
  if (!e || !e->aggregate) return false;
  if (first_pass == 2) e->aggregate_direct(args, grid, block_threads, lds_bytes, stream);
  else if (first_pass == 1) e->aggregate_spill(args, grid, block_threads, lds_bytes, stream);
  else e->aggregate(args, grid, block_threads, lds_bytes, stream);
  *err = hipGetLastError();
  if (kernel_name) *kernel_name = first_pass == 2 ? e->direct_name() : first_pass == 1 ? e->spill_name() : e->name;
  return true;
}

int static_kernel_count() { return (int)(sizeof(g_entries) / sizeof(g_entries[0])) - 1; }
const char *static_kernel_name(int i) { return i >= 0 && i < static_kernel_count() ? g_entries[i].name : nullptr; }

}  // namespace qe
