// gfx950 kernels of the Parquet scan leaf (see qe_parquet_kernels.h for the host/device split).
// All of them are one-thread-per-row streaming kernels: a row finds its page and its run by binary
// search over small host-built tables (L1/L2 resident), reads its value from the chunk image and
// writes the Arrow buffer.  Value sections in a Parquet file have no alignment at all: multi-byte
// reads are misaligned global loads.
#include "qe_parquet_kernels.h"

namespace qe {
namespace {

// (The hardware takes misaligned global loads; given a copy from an address of unknown alignment the
// compiler emits one dword / dwordx2 load, not byte loads.)
__device__ __forceinline__ uint32_t load_u32(const uint8_t *image, uint32_t off) {
  uint32_t v;
  __builtin_memcpy(&v, image + off, 4);
  return v;
}
__device__ __forceinline__ uint64_t load_u64(const uint8_t *image, uint32_t off) {
  uint64_t v;
  __builtin_memcpy(&v, image + off, 8);
  return v;
}

// Last element of a sorted table whose key is <= x (the table's first key is <= x by construction).
template <typename T, typename KeyFn>
__device__ __forceinline__ int upper_slot(const T *table, int begin, int end, int x, KeyFn key) {
  int lo = begin, hi = end - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (key(table[mid]) <= x) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// Value `v` of the hybrid stream described by runs[begin, end).
__device__ __forceinline__ uint32_t run_value(const uint8_t *image, const PqRun *runs, int begin, int end, int v, int bit_width) {
  const PqRun r = runs[upper_slot(runs, begin, end, v, [](const PqRun &q) { return q.start; })];
  if ((uint32_t)(v - r.start) >= (uint32_t)r.count) return 0xffffffffu;  // beyond what the stream holds
  if (!r.packed) return r.data;
  const uint64_t bit = (uint64_t)(uint32_t)(v - r.start) * (uint32_t)bit_width;
  const uint64_t w = load_u64(image, r.data + (uint32_t)(bit >> 3));
  return (uint32_t)(w >> (bit & 7)) & (bit_width >= 32 ? 0xffffffffu : ((1u << bit_width) - 1u));
}

// Number of valid rows before `row` (row may equal the row count).
__device__ __forceinline__ int rank_of(const PqChunk &c, int row) {
  if (!c.validity) return row;
  const int w = row >> 6, b = row & 63;
  int r = c.word_rank[w];
  if (b) r += __popcll(c.validity[w] & ((1ull << b) - 1ull));
  return r;
}
__device__ __forceinline__ bool valid_at(const PqChunk &c, int row) { return !c.validity || ((c.validity[row >> 6] >> (row & 63)) & 1ull); }

__device__ __forceinline__ void flag(const PqChunk &c) { atomicOr(c.error, 1ull); }

__global__ __launch_bounds__(256) void qe_parquet_validity_kernel(const uint8_t *image, const PqRun *runs, int n_runs, int32_t n_rows,
                                                                  uint64_t *validity, int32_t *word_count) {
  const int64_t padded = ((int64_t)n_rows + 63) & ~63ll;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < padded; row += (int64_t)gridDim.x * blockDim.x) {
    uint32_t level = 0;
    if (row < n_rows) level = run_value(image, runs, 0, n_runs, (int)row, 1);
    const uint64_t word = __ballot(level != 0);
    if ((threadIdx.x & 63) == 0) {
      validity[row >> 6] = word;
      word_count[row >> 6] = __popcll(word);
    }
  }
}

__global__ __launch_bounds__(256) void qe_parquet_fixed_kernel(PqChunk c, void *out) {
  const int64_t padded = ((int64_t)c.n_rows + 63) & ~63ll;
  const int width = (c.physical == PQ_INT32 || c.physical == PQ_FLOAT) ? 4 : c.physical == PQ_INT96 ? 12 : 8;
  for (int64_t r64 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r64 < padded; r64 += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)r64;
    const bool in = r64 < c.n_rows;
    uint64_t v = 0;
    // The page of the wave's first row is nearly always the page of all its rows: looked up once, with scalar
    // loads; a lane whose row lies in another page searches for itself.
    const int row_first = __builtin_amdgcn_readfirstlane(row < c.n_rows ? row : c.n_rows - 1);
    PqPage p = c.pages[upper_slot(c.pages, 0, c.n_pages, row_first, [](const PqPage &q) { return q.first_row; })];
    if (in && valid_at(c, row)) {
      if (row < p.first_row || row >= p.first_row + p.num_rows)
        p = c.pages[upper_slot(c.pages, 0, c.n_pages, row, [](const PqPage &q) { return q.first_row; })];
      const int vi = rank_of(c, row) - rank_of(c, p.first_row);
      if (p.encoding == PQ_ENC_RLE_BOOL) {
        v = run_value(c.image, c.runs, p.run_begin, p.run_end, vi, 1);
        if (v > 1) flag(c);
      } else if (c.physical == PQ_BOOLEAN) {
        if ((uint32_t)(vi >> 3) < p.values_len) v = (c.image[p.values_off + (uint32_t)(vi >> 3)] >> (vi & 7)) & 1u;
        else flag(c);
      } else {
        uint32_t src = 0;
        bool ok;
        if (p.encoding == PQ_ENC_DICT) {
          const uint32_t idx = run_value(c.image, c.runs, p.run_begin, p.run_end, vi, p.bit_width);
          ok = idx < (uint32_t)c.dict_count;
          src = c.dict_off + idx * (uint32_t)width;
        } else {
          ok = ((uint64_t)vi + 1) * (uint32_t)width <= p.values_len;
          src = p.values_off + (uint32_t)vi * (uint32_t)width;
        }
        if (!ok) {
          flag(c);
        } else if (c.physical == PQ_INT32) {
          v = load_u32(c.image, src);
        } else if (c.physical == PQ_FLOAT) {  // widened: the engine's floating type is float64
          v = (uint64_t)__double_as_longlong((double)__uint_as_float(load_u32(c.image, src)));
        } else if (c.physical == PQ_INT96) {
          // nanoseconds of the day, then the Julian day number: nanoseconds since 1970-01-01
          const int64_t nanos = (int64_t)load_u64(c.image, src);
          const int64_t day = (int64_t)load_u32(c.image, src + 8);
          v = (uint64_t)((day - 2440588ll) * 86400000000000ll + nanos);
        } else {
          v = load_u64(c.image, src);
        }
      }
    }
    if (c.physical == PQ_BOOLEAN) {
      const uint64_t word = __ballot(v != 0);
      if ((threadIdx.x & 63) == 0) ((uint64_t *)out)[r64 >> 6] = word;
    } else if (in) {
      if (c.physical == PQ_INT32) ((uint32_t *)out)[row] = (uint32_t)v;
      else ((uint64_t *)out)[row] = v;
    }
  }
}

__global__ void qe_parquet_walk_kernel(PqChunk c, const PqWalk *walks, int n_walks, uint32_t *value_off, int32_t *value_len,
                                       uint32_t *dict_entry_off, int32_t *dict_entry_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_walks) return;
  const PqWalk w = walks[i];
  int count = w.count;
  uint32_t *offs = dict_entry_off;
  int32_t *lens = dict_entry_len;
  if (w.first_row >= 0) {
    const int r0 = rank_of(c, w.first_row);
    count = rank_of(c, w.end_row) - r0;
    offs = value_off + r0;
    lens = value_len + r0;
  }
  uint32_t p = w.off;
  const uint32_t end = w.off + w.len;
  int j = 0;
  for (; j < count; ++j) {
    if (end - p < 4) break;
    const uint32_t len = load_u32(c.image, p);
    if (len > end - p - 4) break;
    offs[j] = p + 4;
    lens[j] = (int32_t)len;
    p += 4 + len;
  }
  if (j < count) flag(c);
  for (; j < count; ++j) {
    offs[j] = 0;
    lens[j] = 0;
  }
}

__global__ __launch_bounds__(256) void qe_parquet_utf8_rows_kernel(PqChunk c, const uint32_t *value_off, const int32_t *value_len,
                                                                   const uint32_t *dict_entry_off, const int32_t *dict_entry_len,
                                                                   uint32_t *row_src, int32_t *row_len) {
  unsigned long long total = 0;
  for (int64_t r64 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r64 < c.n_rows; r64 += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)r64;
    uint32_t src = 0;
    int32_t len = 0;
    if (valid_at(c, row)) {
      const PqPage p = c.pages[upper_slot(c.pages, 0, c.n_pages, row, [](const PqPage &q) { return q.first_row; })];
      const int rank = rank_of(c, row);
      if (p.encoding == PQ_ENC_DICT) {
        const uint32_t idx = run_value(c.image, c.runs, p.run_begin, p.run_end, rank - rank_of(c, p.first_row), p.bit_width);
        if (idx < (uint32_t)c.dict_count) {
          src = dict_entry_off[idx];
          len = dict_entry_len[idx];
        } else {
          flag(c);
        }
      } else {
        src = value_off[rank];
        len = value_len[rank];
      }
    }
    row_src[row] = src;
    row_len[row] = len;
    total += (unsigned long long)(uint32_t)len;
  }
  // 64-bit sum of the lengths next to the error flag: the offsets are int32, the host refuses a
  // column whose bytes do not fit them before anything is copied
  // (one atomic per workgroup: every wave drawing on the one counter would queue up behind it)
  __shared__ unsigned long long s_total[4];
  for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off);
  if ((threadIdx.x & 63) == 0) s_total[threadIdx.x >> 6] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_total[0] + s_total[1] + s_total[2] + s_total[3];
    if (t) atomicAdd(c.error + 1, t);
  }
}

// Eight lanes per row: short strings dominate, and eight byte lanes keep a wave's stores within
// a few cache lines.
__global__ __launch_bounds__(256) void qe_parquet_utf8_copy_kernel(const uint8_t *image, const uint32_t *row_src, const int32_t *offsets,
                                                                   int32_t n_rows, uint8_t *out) {
  const int sub = threadIdx.x & 7;
  for (int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 3; row < n_rows; row += ((int64_t)gridDim.x * blockDim.x) >> 3) {
    const int32_t d0 = offsets[row], len = offsets[row + 1] - d0;
    const uint32_t s0 = row_src[row];
    for (int32_t b = sub; b < len; b += 8) out[d0 + b] = image[s0 + b];
  }
}

unsigned grid_rows(int64_t rows, int per_block, int max_blocks) {
  int64_t g = (rows + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (unsigned)g;
}

}  // namespace

hipError_t launch_pq_validity(const uint8_t *image, const PqRun *level_runs, int n_runs, int32_t n_rows, uint64_t *validity,
                              int32_t *word_count, hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_parquet_validity_kernel, dim3(grid_rows(n_rows, 256, 8192)), dim3(256), 0, stream, image, level_runs, n_runs, n_rows,
                     validity, word_count);
  return hipGetLastError();
}

hipError_t launch_pq_fixed(const PqChunk &chunk, void *out, hipStream_t stream) {
  if (chunk.n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_parquet_fixed_kernel, dim3(grid_rows(chunk.n_rows, 256, 8192)), dim3(256), 0, stream, chunk, out);
  return hipGetLastError();
}

hipError_t launch_pq_walk(const PqChunk &chunk, const PqWalk *walks, int n_walks, uint32_t *value_off, int32_t *value_len,
                          uint32_t *dict_entry_off, int32_t *dict_entry_len, hipStream_t stream) {
  if (n_walks <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_parquet_walk_kernel, dim3((n_walks + 63) / 64), dim3(64), 0, stream, chunk, walks, n_walks, value_off, value_len,
                     dict_entry_off, dict_entry_len);
  return hipGetLastError();
}

hipError_t launch_pq_utf8_rows(const PqChunk &chunk, const uint32_t *value_off, const int32_t *value_len, const uint32_t *dict_entry_off,
                               const int32_t *dict_entry_len, uint32_t *row_src, int32_t *row_len, hipStream_t stream) {
  if (chunk.n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_parquet_utf8_rows_kernel, dim3(grid_rows(chunk.n_rows, 256, 2048)), dim3(256), 0, stream, chunk, value_off, value_len,
                     dict_entry_off, dict_entry_len, row_src, row_len);
  return hipGetLastError();
}

hipError_t launch_pq_utf8_copy(const uint8_t *image, const uint32_t *row_src, const int32_t *offsets, int32_t n_rows, uint8_t *out,
                               hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_parquet_utf8_copy_kernel, dim3(grid_rows((int64_t)n_rows * 8, 256, 16384)), dim3(256), 0, stream, image, row_src, offsets,
                     n_rows, out);
  return hipGetLastError();
}

}  // namespace qe
