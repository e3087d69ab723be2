// Device side of the Parquet scan leaf: the descriptors the host builds while it walks a column
// chunk's page headers, and the launchers of the kernels that turn the chunk's bytes (resident in
// HBM as one "image") into Arrow columns.
//
// Division of work: everything that is sequential and tiny stays on the host (thrift page headers,
// the run headers of RLE / bit-packed hybrid streams: one header per up to 504 values); everything
// that touches every value runs on the device, one thread per row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qe {

enum PqPhysical { PQ_BOOLEAN = 0, PQ_INT32 = 1, PQ_INT64 = 2, PQ_INT96 = 3, PQ_FLOAT = 4, PQ_DOUBLE = 5, PQ_BYTE_ARRAY = 6 };
enum PqPageEncoding { PQ_ENC_PLAIN = 0, PQ_ENC_DICT = 1, PQ_ENC_RLE_BOOL = 2 };

// One run of an RLE / bit-packed hybrid stream.
struct PqRun {
  int32_t start;    // index of the run's first value: row in the row group (levels), value in the page (indices)
  int32_t count;
  uint32_t data;    // repeated run: the value; bit-packed run: image offset of the packed bytes
  uint32_t packed;  // 1 = bit-packed (LSB first)
};

// One data page.
struct PqPage {
  int32_t first_row;   // row in the row group of the page's first value
  int32_t num_rows;    // values in the page, nulls included
  uint32_t values_off; // image offset of the value section (after the levels)
  uint32_t values_len;
  int32_t encoding;    // PqPageEncoding
  int32_t bit_width;   // dictionary indices
  int32_t run_begin, run_end;  // the page's slice of the value runs (dictionary indices / RLE booleans)
};

struct PqChunk {
  const uint8_t *image;       // the column chunk's (decompressed) bytes; padded by 16 readable bytes
  uint32_t image_len;
  const PqPage *pages;
  int32_t n_pages;
  const PqRun *runs;          // value runs of all pages
  const uint64_t *validity;   // 64 rows per word, LSB first; null = no nulls
  const int32_t *word_rank;   // number of valid rows before each validity word
  int32_t n_rows;
  int32_t physical;           // PqPhysical
  uint32_t dict_off;          // image offset of the dictionary page's values (fixed width types)
  int32_t dict_count;
  unsigned long long *error;  // [0] set non-zero when the bytes contradict the descriptors; [1] += string bytes of the column
};

// Definition levels (bit width 1) -> validity words + the number of valid rows per word.
hipError_t launch_pq_validity(const uint8_t *image, const PqRun *level_runs, int n_runs, int32_t n_rows, uint64_t *validity,
                              int32_t *word_count, hipStream_t stream);
// BOOLEAN / INT32 / INT64 / DOUBLE / INT96 values, plain or dictionary encoded -> `out`
// (bit-packed for BOOLEAN, int64 nanoseconds since the epoch for INT96).  Null rows get 0.
hipError_t launch_pq_fixed(const PqChunk &chunk, void *out, hipStream_t stream);
// Length-prefixed byte arrays: one thread per segment walks the prefixes and records where each
// value starts and how long it is.  A segment is the value section of a plain-encoded data page
// (values numbered by their rank among the row group's valid rows) or the dictionary page.
struct PqWalk {
  uint32_t off, len;   // image range to walk
  int32_t first_row;   // data page: rows [first_row, end_row); dictionary: -1
  int32_t end_row;
  int32_t count;       // dictionary: number of entries
};
hipError_t launch_pq_walk(const PqChunk &chunk, const PqWalk *walks, int n_walks, uint32_t *value_off, int32_t *value_len,
                          uint32_t *dict_entry_off, int32_t *dict_entry_len, hipStream_t stream);
// Per row: where its string lies in the image and how long it is (0 for nulls).
hipError_t launch_pq_utf8_rows(const PqChunk &chunk, const uint32_t *value_off, const int32_t *value_len, const uint32_t *dict_entry_off,
                               const int32_t *dict_entry_len, uint32_t *row_src, int32_t *row_len, hipStream_t stream);
// Copies every row's bytes to its place in the Arrow values buffer.
hipError_t launch_pq_utf8_copy(const uint8_t *image, const uint32_t *row_src, const int32_t *offsets, int32_t n_rows, uint8_t *out,
                               hipStream_t stream);

}  // namespace qe
