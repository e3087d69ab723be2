// Parquet scan leaf: the reference's ParquetDataSource (data_source.rs:24-72) with the decode on
// the GPU.  The host reads the footer and the page headers; column chunks are shipped to HBM as
// they are in the file (after block decompression, where the file uses it) and decoded there.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
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

// Column chunks on their way from the file to the device, ahead of the chunk that is being decoded: every chunk is cut
// into slices, the slices of all chunks in flight form one queue that the process's reader threads drain in file order,
// and a reader that has read its slice into the chunk's pinned staging buffer queues that slice's copy to the device
// itself, on the context's copy stream.  The scan's own thread only waits for a chunk's last copy to be queued, walks
// the page headers and launches the decode behind an event: while it does that, while the device decodes, and while
// the host waits for a row group's last kernel, the next chunks - the next row group's included - keep flowing.
// Owned by the scan.
struct ParquetReadAhead {
  // What the reader jobs of one chunk and the scan share.
  struct Progress {
    std::mutex m;
    std::condition_variable cv;
    size_t n_slices = 0;
    size_t n_landed = 0;      // reader jobs through: their slices read, the copies they were to queue queued
    std::vector<uint16_t> group_landed;  // slices read, per group of slices that travel in one copy
    std::exception_ptr err;   // the first failure
    double first_start = 0.0, last_end = 0.0, reading = 0.0, queueing = 0.0, longest_queueing = 0.0;  // (trace only)
    std::atomic<bool> all_landed{false};
    void wait_all() {
      std::unique_lock<std::mutex> g(m);
      cv.wait(g, [this] { return n_landed == n_slices; });
    }
    // The scan's own wait.  Not asleep for the first while: with every CPU of the process busy reading, a thread woken
    // by a condition variable was seen to wait three milliseconds for a CPU, once per scan - as long as 150 MB take.
    void wait_all_awake() {
      if (n_slices == 0) return;
      const auto t0 = std::chrono::steady_clock::now();
      for (int spins = 0; !all_landed.load(std::memory_order_acquire); ++spins) {
        __builtin_ia32_pause();
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return wait_all();
      }
    }
  };
  struct Read {
    int row_group = -1, column = -1;
    Context *ctx = nullptr;
    Context::Staging *stage = nullptr;
    size_t bytes = 0;
    BufferPtr dev;  // where the chunk's image goes, with room for its tables
    std::shared_ptr<Progress> progress;
    Read() = default;
    Read(Read &&) = default;
    Read &operator=(Read &&o) {
      if (this != &o) {
        settle();
        row_group = o.row_group, column = o.column, ctx = o.ctx, stage = o.stage, bytes = o.bytes;
        dev = std::move(o.dev);
        progress = std::move(o.progress);
      }
      return *this;
    }
    // The reader jobs read the file, write the staging buffer and queue copies into the device block: nobody lets go of
    // any of them before the jobs are through and (the block goes back to a pool) the copies have been made.
    void settle();
    ~Read() { settle(); }
  };
  std::deque<Read> queue;  // in scan order
};

// One row group, the given columns, decoded into a device-resident batch (the reference's
// FileReader yields one chunk per row group).  `ahead` (may be null) carries reads over to the scan's
// next call; `last_row_group` bounds what is read ahead (exclusive).
Batch parquet_read_row_group(Context *ctx, ParquetFile &file, int row_group, const std::vector<int> &columns, ParquetReadAhead *ahead = nullptr,
                             int last_row_group = -1);

}  // namespace qe
