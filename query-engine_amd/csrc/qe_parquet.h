// Parquet scan leaf: the reference's ParquetDataSource (data_source.rs:24-72) with the decode on
// the GPU.  The host reads the footer and the page headers; column chunks are shipped to HBM as
// they are in the file (after block decompression, where the file uses it) and decoded there.
#pragma once
#include <deque>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "qe_core.h"

namespace qe {

struct ParquetColumn {
  std::string name;
  int physical = 0;   // parquet physical type
  int type = 0;       // QE_TYPE_*, 0 = a type the engine cannot scan
  bool optional = false;
  std::string type_text;  // what the footer says, for error messages
};

struct ParquetChunkMeta {
  int64_t start = 0;  // file offset of the chunk's first page
  int64_t size = 0;   // bytes of all its pages, headers included
  int64_t num_values = 0;
  int codec = 0;
};

struct ParquetRowGroup {
  int64_t num_rows = 0;
  std::vector<ParquetChunkMeta> chunks;  // one per column, in schema order
};

struct ParquetFile {
  std::string path;
  int fd = -1;
  int64_t file_size = 0;
  int64_t num_rows = 0;
  std::vector<ParquetColumn> columns;
  std::vector<ParquetRowGroup> row_groups;
  ~ParquetFile();
};

// ParquetDataSource::new (data_source.rs:30-41): IoError when the file cannot be opened,
// ArrowError when it is not a Parquet file this reader understands.
std::shared_ptr<ParquetFile> parquet_open(const std::string &path);

// Indices of the columns a projection selects: every column whose name the projection holds, in
// file order (data_source.rs:49-62).
std::vector<int> parquet_project(const ParquetFile &file, const std::vector<std::string> &projection);

// Column chunks on their way from the file into pinned staging buffers, read ahead of the chunk that is being decoded
// (by reader threads of their own): while the device copies and decodes one chunk - and while the host waits for a row
// group's last one - the next chunks, the next row group's included, are already being read.  Owned by the scan.
struct ParquetReadAhead {
  struct Read {
    int row_group = -1, column = -1;
    Context::Staging *stage = nullptr;
    std::future<void> done;
  };
  std::deque<Read> queue;  // in scan order
};

// One row group, the given columns, decoded into a device-resident batch (the reference's
// FileReader yields one chunk per row group).  `ahead` (may be null) carries reads over to the scan's
// next call; `last_row_group` bounds what is read ahead (exclusive).
Batch parquet_read_row_group(Context *ctx, ParquetFile &file, int row_group, const std::vector<int> &columns, ParquetReadAhead *ahead = nullptr,
                             int last_row_group = -1);

}  // namespace qe
