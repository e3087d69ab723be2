// Host-callable launchers of the gfx950 kernels (defined in qe_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qe_vm.h"

namespace qe {

struct DeviceInfo {
  int cus;          // compute units
  int max_lds;      // bytes of LDS one workgroup may use
  bool specialised = true;  // use a kernel built for the launch's exact shape when there is one
  int device = 0;
  int jit = 0;               // JitMode: compile such a kernel at run time when none was built ahead
  int64_t jit_min_rows = 1 << 22;  // ... for batches of at least this many rows (background mode)
  uint64_t total_mem = 0;          // bytes of device memory
};

// Unfiltered projection: every expression of `args` evaluated for every row.
// `kernel_name` (optional) receives the name of the kernel variant that was launched.
hipError_t launch_project(const ProjectArgs &args, const DeviceInfo &dev, hipStream_t stream, const char **kernel_name = nullptr);

// Stable predicate -> compaction (+ projection).  `args.scratch` must hold
// filter_scratch_bytes(n_rows) bytes, zeroed by launch_filter_reset before every launch.
hipError_t launch_filter_reset(FilterScratch *scratch, int64_t n_rows, hipStream_t stream);
hipError_t launch_filter(const FilterArgs &args, const DeviceInfo &dev, hipStream_t stream, const char **kernel_name = nullptr);
size_t filter_scratch_bytes(int64_t n_rows);
int64_t filter_num_tiles(int64_t n_rows);

// Hash aggregate update over one batch.
// `counted` (may be null): whether the launch filled args.cells / args.cells_total (AggArgs::cells).
hipError_t launch_aggregate(const AggArgs &args, const DeviceInfo &dev, hipStream_t stream, const char **kernel_name = nullptr, bool *counted = nullptr);
// Slots of the workgroup's LDS table (`wide`: for the 1024-thread launch geometry, see AggArgs::block_threads).
uint32_t aggregate_lds_slots(const AggArgs &args, const DeviceInfo &dev, int64_t group_hint, bool wide);
// Whether a kernel that can be launched wide exists for this shape right now.
bool aggregate_wide_available(const AggArgs &args, const DeviceInfo &dev);
// The many-groups first pass that partitions its rows itself (AggSpill::direct_cap; qe_aggregate_direct_static):
// whether a kernel for this shape can be launched right now, the launch (workgroups = dev.cus, one column of the
// cell table each), and the rows a (partition, workgroup) stretch must hold for a batch of `n_rows`.
bool aggregate_direct_available(const AggArgs &args, const DeviceInfo &dev, int bits);
hipError_t launch_aggregate_direct(const AggArgs &args, const DeviceInfo &dev, hipStream_t stream, const char **kernel_name = nullptr);
uint32_t aggregate_direct_stretch_rows(int64_t n_rows, int bits, int blocks);
// Whether launch_aggregate, given a cell table of 2^bits partitions (AggArgs::cells), will run the many-groups first
// pass that counts its logged rows per partition itself (the launch then reports `counted`).
bool aggregate_counts_partitions(const AggArgs &args, const DeviceInfo &dev, int bits);

// High-cardinality mode, second phase (see AggSpill): partition the spill log by key hash
// (count, scan by the caller, move), then aggregate partition by partition.
size_t spill_cells(int bits, int blocks);
hipError_t launch_spill_count(const AggSpill &sp, int n_segments, int bits, int blocks, int32_t *cells, unsigned long long *total,
                              hipStream_t stream);
int spill_level1_bits(int bits);  // bits of the first scatter pass (== bits: there is only one)
hipError_t launch_spill_cells_level1(const int32_t *cells, int32_t *cells1, int bits, int blocks, hipStream_t stream);
hipError_t launch_spill_scatter(const AggSpill &sp, int n_segments, int level, int bits, int bits1, int blocks, const unsigned long long *src_keys,
                                const unsigned long long *src_vals, uint64_t src_stride, const int32_t *src_cells, int64_t total,
                                const int32_t *dst_cells, unsigned long long *dst_keys, unsigned long long *dst_vals, uint64_t dst_stride,
                                hipStream_t stream);
uint32_t partitioned_lds_slots(int n_accs);
hipError_t launch_aggregate_partitioned(const PartArgs &args, const DeviceInfo &dev, hipStream_t stream);
// The scatter in one pass: (key, value) pairs of 16 bytes, written a whole 128-byte line at a time.  For logs with at most
// one value column and at most 512 partitions; `dst_rows` holds 16 bytes per logged row.
bool spill_onepass_available(int bits, uint32_t n_inputs);
hipError_t launch_spill_scatter_onepass(const AggSpill &sp, int n_segments, int bits, int blocks, const int32_t *dst_cells, int cells_stride, void *dst_rows,
                                        hipStream_t stream);
// Workgroups launch_aggregate will use for these arguments (sizes the per-wave spill regions).
int aggregate_grid(const AggArgs &args, const DeviceInfo &dev);

// Fill a fresh global group table with the empty key / accumulator identities.
// `zero` (optional, 8-byte aligned, a multiple of 8 bytes): a buffer cleared by the same launch.
hipError_t launch_table_init(const AggTable &table, const VmAcc *accs, int n_accs, hipStream_t stream, void *zero = nullptr, size_t zero_bytes = 0);
// Compact occupied slots into output columns.
hipError_t launch_finalize(const FinalizeArgs &args, hipStream_t stream);
// Moves the limbs of the order-independent sum in accumulators acc .. acc+2 to a scale `delta` bits up.
hipError_t launch_fix_rescale(const AggTable &table, int acc, int delta, hipStream_t stream);

// utf8 helpers: gather rows `idx[0..m)` of a utf8 column.
hipError_t launch_exclusive_scan_i32(int32_t *data, int64_t n, int32_t *block_sums, hipStream_t stream);
size_t scan_scratch_elems(int64_t n);
// Rows idx[0..m) of up to UTF8_GATHER_MAX_COLS string columns, every column in the same launch (a launch per column
// leaves most of the chip idle while its first tiles start and its last ones end; the columns' tiles fill those gaps
// for each other).  Per column: control = 32 zeroed bytes whose second word receives the byte total and whose third is
// non-zero if the look-back gave up (the first column's first word is the launch's ticket counter); status =
// utf8_gather_offsets_status_bytes(m) zeroed bytes.
constexpr int UTF8_GATHER_MAX_COLS = 16;
struct Utf8GatherCol {
  const int32_t *src_offsets;
  const uint8_t *src_validity;  // or nullptr
  const uint8_t *src_bytes;
  int64_t src_total;            // bytes of the source column as its buffer bounds them
  int32_t *out_offsets;         // [m + 1], 16-byte aligned
  uint32_t *out_validity;       // or nullptr
  uint8_t *out_bytes;           // set before launch_utf8_gather_bytes
  void *control;
  void *status;
};
struct Utf8GatherCols {
  Utf8GatherCol col[UTF8_GATHER_MAX_COLS];
  int n;
};
size_t utf8_gather_offsets_status_bytes(int64_t m);
// out_offsets[0..m] and the validity bits of every column.
// `m_dev` (may be null): where the device holds the number of rows; `m` is then only a bound on it (grids and buffers are sized by it).
hipError_t launch_utf8_gather_offsets(const Utf8GatherCols &cols, const int32_t *idx, int64_t m, const unsigned long long *m_dev, int cus, hipStream_t stream);
// The bytes, once out_offsets are there and out_bytes has been sized from the totals.
hipError_t launch_utf8_gather_bytes(const Utf8GatherCols &cols, const int32_t *idx, int64_t m, const unsigned long long *m_dev, hipStream_t stream);
// Offsets and bytes in one launch, for a caller that has set out_bytes before it knows the byte totals (buffers sized
// by the sources): the tile that places its rows copies their bytes.  Same scratch as launch_utf8_gather_offsets.
bool utf8_gather_fused_available();
hipError_t launch_utf8_gather_fused(const Utf8GatherCols &cols, const int32_t *idx, int64_t m, const unsigned long long *m_dev, int cus, hipStream_t stream);
// Gather of strings by global row number over several source columns (group-by output keys).
// Pass out_lengths to get the lengths (first pass), dst_offsets + dst_bytes to copy the bytes.
struct Utf8Source {
  const int32_t *offsets;
  const uint8_t *bytes;
  int64_t first_row;  // global row number of the source's first row; sources are in ascending order
};
hipError_t launch_utf8_gather_rows(const Utf8Source *sources, int n_sources, const int64_t *ids, int64_t m, int32_t *out_lengths,
                                   const int32_t *dst_offsets, uint8_t *dst_bytes, hipStream_t stream);
// The same for fixed-width columns: width 8 or 4 bytes, 0 = bit-packed booleans (`out` zeroed by the caller).
struct FixedSource {
  const void *values;
  int64_t first_row;
};
hipError_t launch_gather_rows_fixed(const FixedSource *sources, int n_sources, const int64_t *ids, int64_t m, int width, void *out,
                                    hipStream_t stream);
// Key equality for groups that are identified by a 64-bit hash of their key tuple (string keys, wide tuples): every
// selected row's key columns against those of its group's representative row (the row whose number the group's hidden
// MIN(row number) accumulator holds).  The reference identifies groups by hash alone (physical_plan/mod.rs:310-358); two
// distinct keys with one hash would merge silently.  One launch per scanned batch, after every batch has been folded in.
constexpr int VERIFY_MAX_KEYS = 12;
struct KeyColumnSource {  // one key column of one scanned batch
  const void *values;     // fixed width: values (booleans bit-packed); utf8: bytes
  const int32_t *offsets; // utf8 only
  int64_t first_row;      // global row number of the batch's first row; batches in ascending order
};
struct VerifyArgs {
  AggTable table;
  int rep_acc;                       // the accumulator holding the order-preserving image of MIN(row number)
  const int64_t *row_ids;            // the batch's selected rows (global row numbers) and their tuple hashes
  const unsigned long long *hashes;
  const uint32_t *hash_validity;     // null: every hash is valid (a row with a null key has none and belongs to no group)
  int64_t m;
  int64_t row_base;                  // global row number of this batch's first row
  int this_source;                   // index of this batch among the sources
  int n_keys, n_sources;
  int key_type[VERIFY_MAX_KEYS];     // VmType (= QE_TYPE_*)
  const KeyColumnSource *sources;    // [n_keys][n_sources]
  unsigned long long *mismatches;    // += rows whose key differs from their group's representative's
};
hipError_t launch_verify_group_keys(const VerifyArgs &args, hipStream_t stream);
// Broadcast one utf8 literal to `n` rows (scalar_to_array for Utf8, columnar_value.rs:40-50).
hipError_t launch_utf8_broadcast(const uint8_t *lit, int32_t len, int64_t n, int32_t *out_offsets, uint8_t *out_bytes,
                                 hipStream_t stream);

// Number of zero bits among the first n bits of a bitmap (null count).
hipError_t launch_count_zero_bits(const uint32_t *bitmap, int64_t n, unsigned long long *out, hipStream_t stream);

// Kernels built for one exact launch shape (qe_static.hip).  Each returns false when no such
// kernel exists for `args`; otherwise it launches, sets *err and *kernel_name, and returns true.
bool static_launch_project(const ProjectArgs &args, int grid, hipStream_t stream, hipError_t *err, const char **kernel_name);
bool static_launch_filter(const FilterArgs &args, int cus, hipStream_t stream, hipError_t *err, const char **kernel_name);
bool static_launch_aggregate(const AggArgs &args, int grid, int block_threads, size_t lds_bytes, hipStream_t stream, hipError_t *err,
                             const char **kernel_name, int first_pass = 0);  // first_pass: 0 the aggregate kernel, 1 / 2 the many-groups first pass that logs / partitions
bool static_has_aggregate(const AggArgs &args);
int static_kernel_count();
const char *static_kernel_name(int i);

// Synthetic columns (see qe_generate_column in qe_gpu.h).
hipError_t launch_generate(int kind, uint64_t seed, int64_t row0, int64_t n, double lo, double hi, void *out,
                           hipStream_t stream);

}  // namespace qe
