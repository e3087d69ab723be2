// Host side of the Parquet scan leaf: footer and page-header parsing (thrift compact protocol),
// block decompression, run tables for the RLE / bit-packed hybrid streams, and the orchestration of
// the decode kernels.  Follows the Apache Parquet format for flat schemas; what the reference does
// with the file is in data_source.rs:24-72 (open + read_metadata, infer_schema, projected scan)
// through arrow2's reader.
#include "qe_parquet.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>

#include "qe_parquet_kernels.h"

namespace qe {
namespace {

[[noreturn]] void bad_file(const std::string &what) { fail(QE_ERR_ARROW, "Arrow error: parquet: " + what); }

// ------------------------------------------------------------ thrift compact protocol
struct Thrift {
  const uint8_t *p, *end;
  uint8_t byte() {
    if (p >= end) bad_file("truncated metadata");
    return *p++;
  }
  uint64_t varint() {
    uint64_t r = 0;
    for (int s = 0; s < 64; s += 7) {
      const uint8_t c = byte();
      r |= (uint64_t)(c & 0x7f) << s;
      if (!(c & 0x80)) return r;
    }
    bad_file("overlong varint");
  }
  int64_t zigzag() {
    const uint64_t v = varint();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  std::string binary() {
    const uint64_t n = varint();
    if (n > (uint64_t)(end - p)) bad_file("truncated metadata");
    std::string s((const char *)p, (size_t)n);
    p += n;
    return s;
  }
  // Next field of the struct being read: false at its end.  `last` carries the previous field id.
  bool field(int &last, int &type) {
    const uint8_t h = byte();
    if (h == 0) return false;
    type = h & 15;
    const int delta = h >> 4;
    last = delta ? last + delta : (int)zigzag();
    return true;
  }
  void list(int &n, int &type) {
    const uint8_t h = byte();
    n = h >> 4;
    type = h & 15;
    if (n == 15) n = (int)varint();
  }
  void skip(int type, int depth = 0) {
    if (depth > 32) bad_file("metadata nested too deeply");
    switch (type) {
      case 1: case 2: return;
      case 3: byte(); return;
      case 4: case 5: case 6: varint(); return;
      case 7:
        if (end - p < 8) bad_file("truncated metadata");
        p += 8;
        return;
      case 8: binary(); return;
      case 9: case 10: {
        int n, t;
        list(n, t);
        for (int i = 0; i < n; ++i) {
          if (t == 1 || t == 2) byte();
          else skip(t, depth + 1);
        }
        return;
      }
      case 11: {
        const uint64_t n = varint();
        if (!n) return;
        const uint8_t kv = byte();
        for (uint64_t i = 0; i < n; ++i) {
          skip(kv >> 4, depth + 1);
          skip(kv & 15, depth + 1);
        }
        return;
      }
      case 12: {
        int last = 0, t;
        while (field(last, t)) skip(t, depth + 1);
        return;
      }
      default: bad_file("unknown thrift type");
    }
  }
};

enum { CODEC_NONE = 0, CODEC_SNAPPY = 1, CODEC_GZIP = 2, CODEC_ZSTD = 6, CODEC_LZ4_RAW = 7 };
enum { ENC_PLAIN = 0, ENC_PLAIN_DICTIONARY = 2, ENC_RLE = 3, ENC_BIT_PACKED = 4, ENC_RLE_DICTIONARY = 8 };
enum { PAGE_DATA = 0, PAGE_INDEX = 1, PAGE_DICTIONARY = 2, PAGE_DATA_V2 = 3 };

const char *physical_name(int t) {
  static const char *n[] = {"BOOLEAN", "INT32", "INT64", "INT96", "FLOAT", "DOUBLE", "BYTE_ARRAY", "FIXED_LEN_BYTE_ARRAY"};
  return t >= 0 && t < 8 ? n[t] : "?";
}

struct SchemaElement {
  int type = -1, repetition = 0, num_children = 0, converted = -1, logical = 0, time_unit = 0;
  std::string name;
};

SchemaElement read_schema_element(Thrift &t) {
  SchemaElement e;
  int last = 0, ty;
  while (t.field(last, ty)) {
    switch (last) {
      case 1: e.type = (int)t.zigzag(); break;
      case 3: e.repetition = (int)t.zigzag(); break;
      case 4: e.name = t.binary(); break;
      case 5: e.num_children = (int)t.zigzag(); break;
      case 6: e.converted = (int)t.zigzag(); break;
      case 10: {  // LogicalType union: the id of the one field that is set
        int l2 = 0, t2;
        while (t.field(l2, t2)) {
          e.logical = l2;
          if (l2 == 8 && t2 == 12) {  // TimestampType { isAdjustedToUTC, unit }
            int l3 = 0, t3;
            while (t.field(l3, t3)) {
              if (l3 == 2 && t3 == 12) {
                int l4 = 0, t4;
                while (t.field(l4, t4)) {
                  e.time_unit = l4;  // 1 millis, 2 micros, 3 nanos
                  t.skip(t4);
                }
              } else {
                t.skip(t3);
              }
            }
          } else {
            t.skip(t2);
          }
        }
        break;
      }
      default: t.skip(ty);
    }
  }
  return e;
}

ParquetChunkMeta read_column_chunk(Thrift &t, int *physical) {
  ParquetChunkMeta m;
  int64_t data_off = -1, dict_off = -1;
  int last = 0, ty;
  while (t.field(last, ty)) {
    if (last == 3 && ty == 12) {  // ColumnMetaData
      int l2 = 0, t2;
      while (t.field(l2, t2)) {
        switch (l2) {
          case 1: *physical = (int)t.zigzag(); break;
          case 4: m.codec = (int)t.zigzag(); break;
          case 5: m.num_values = t.zigzag(); break;
          case 7: m.size = t.zigzag(); break;
          case 9: data_off = t.zigzag(); break;
          case 11: dict_off = t.zigzag(); break;
          default: t.skip(t2);
        }
      }
    } else {
      t.skip(ty);
    }
  }
  if (data_off < 0) bad_file("column chunk without a data page offset");
  m.start = dict_off > 0 && dict_off < data_off ? dict_off : data_off;
  return m;
}

struct PageHeader {
  int type = -1, uncompressed = 0, compressed = 0;
  int num_values = 0, encoding = 0, def_encoding = ENC_RLE;
  int def_len = 0, rep_len = 0;
  bool v2_compressed = true;
};

PageHeader read_page_header(Thrift &t) {
  PageHeader h;
  int last = 0, ty;
  while (t.field(last, ty)) {
    if (last == 1) h.type = (int)t.zigzag();
    else if (last == 2) h.uncompressed = (int)t.zigzag();
    else if (last == 3) h.compressed = (int)t.zigzag();
    else if ((last == 5 || last == 7 || last == 8) && ty == 12) {
      const int which = last;
      int l2 = 0, t2;
      while (t.field(l2, t2)) {
        if (l2 == 1 && t2 != 12) h.num_values = (int)t.zigzag();
        else if (which == 5 && l2 == 2) h.encoding = (int)t.zigzag();
        else if (which == 5 && l2 == 3) h.def_encoding = (int)t.zigzag();
        else if (which == 7 && l2 == 2) h.encoding = (int)t.zigzag();
        else if (which == 8 && l2 == 4) h.encoding = (int)t.zigzag();
        else if (which == 8 && l2 == 5) h.def_len = (int)t.zigzag();
        else if (which == 8 && l2 == 6) h.rep_len = (int)t.zigzag();
        else if (which == 8 && l2 == 7) h.v2_compressed = t2 == 1;
        else t.skip(t2);
      }
    } else {
      t.skip(ty);
    }
  }
  if (h.type < 0 || h.compressed < 0 || h.uncompressed < 0 || h.num_values < 0 || h.def_len < 0 || h.rep_len < 0) bad_file("corrupt page header");
  return h;
}

// ------------------------------------------------------------------- block codecs
void snappy_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
  Thrift t{src, src + n};
  if (t.varint() != dst_len) bad_file("snappy: length mismatch");
  const uint8_t *p = t.p, *end = src + n;
  size_t o = 0;
  while (p < end) {
    const uint8_t tag = *p++;
    size_t len, off;
    switch (tag & 3) {
      case 0: {
        len = tag >> 2;
        if (len >= 60) {
          const size_t nb = len - 59;
          if ((size_t)(end - p) < nb) bad_file("snappy: truncated");
          len = 0;
          for (size_t i = 0; i < nb; ++i) len |= (size_t)p[i] << (8 * i);
          p += nb;
        }
        ++len;
        if ((size_t)(end - p) < len || dst_len - o < len) bad_file("snappy: corrupt literal");
        memcpy(dst + o, p, len);
        p += len;
        o += len;
        continue;
      }
      case 1:
        if (end - p < 1) bad_file("snappy: truncated");
        len = 4 + ((tag >> 2) & 7);
        off = ((size_t)(tag >> 5) << 8) | p[0];
        p += 1;
        break;
      case 2:
        if (end - p < 2) bad_file("snappy: truncated");
        len = 1 + (tag >> 2);
        off = p[0] | ((size_t)p[1] << 8);
        p += 2;
        break;
      default:
        if (end - p < 4) bad_file("snappy: truncated");
        len = 1 + (tag >> 2);
        off = p[0] | ((size_t)p[1] << 8) | ((size_t)p[2] << 16) | ((size_t)p[3] << 24);
        p += 4;
        break;
    }
    if (off == 0 || off > o || dst_len - o < len) bad_file("snappy: corrupt copy");
    if (off >= len) {
      memcpy(dst + o, dst + o - off, len);
    } else {
      for (size_t i = 0; i < len; ++i) dst[o + i] = dst[o - off + i];  // overlapping: byte by byte
    }
    o += len;
  }
  if (o != dst_len) bad_file("snappy: length mismatch");
}

void gzip_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
  z_stream z{};
  if (inflateInit2(&z, 15 + 32) != Z_OK) bad_file("gzip: cannot start");
  z.next_in = const_cast<Bytef *>(src);
  z.avail_in = (uInt)n;
  z.next_out = dst;
  z.avail_out = (uInt)dst_len;
  const int r = inflate(&z, Z_FINISH);
  const size_t got = z.total_out;
  inflateEnd(&z);
  if (r != Z_STREAM_END || got != dst_len) bad_file("gzip: corrupt page");
}

// zstd and lz4 are present on the system as run-time libraries only: bound by name when a file needs them.
void *library_symbol(const char *lib, const char *name) {
  void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
  return h ? dlsym(h, name) : nullptr;
}

void zstd_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
  using Fn = size_t (*)(void *, size_t, const void *, size_t);
  static Fn fn = (Fn)library_symbol("libzstd.so.1", "ZSTD_decompress");
  if (!fn) fail(QE_ERR_ARROW, "Arrow error: parquet: ZSTD pages need libzstd.so.1, which is not installed");
  if (fn(dst, dst_len, src, n) != dst_len) bad_file("zstd: corrupt page");
}

void lz4_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
  using Fn = int (*)(const char *, char *, int, int);
  static Fn fn = (Fn)library_symbol("liblz4.so.1", "LZ4_decompress_safe");
  if (!fn) fail(QE_ERR_ARROW, "Arrow error: parquet: LZ4_RAW pages need liblz4.so.1, which is not installed");
  if (fn((const char *)src, (char *)dst, (int)n, (int)dst_len) != (int)dst_len) bad_file("lz4: corrupt page");
}

void decompress(int codec, const uint8_t *src, size_t n, uint8_t *dst, size_t dst_len) {
  switch (codec) {
    case CODEC_NONE:
      if (n != dst_len) bad_file("page sizes disagree");
      memcpy(dst, src, n);
      return;
    case CODEC_SNAPPY: return snappy_decompress(src, n, dst, dst_len);
    case CODEC_GZIP: return gzip_decompress(src, n, dst, dst_len);
    case CODEC_ZSTD: return zstd_decompress(src, n, dst, dst_len);
    case CODEC_LZ4_RAW: return lz4_decompress(src, n, dst, dst_len);
    default: fail(QE_ERR_ARROW, "Arrow error: parquet: compression codec " + std::to_string(codec) + " is not supported");
  }
}

// Walks the run headers of a hybrid stream image[off, off+len) and appends one PqRun per run until
// `limit` values are covered or the bytes end.  `first` is the index given to the first value.
int walk_runs(const uint8_t *image, uint32_t off, uint32_t len, int bit_width, int first, int limit, std::vector<PqRun> &out) {
  Thrift t{image + off, image + off + len};
  int n = 0;
  const int value_bytes = (bit_width + 7) / 8;
  while (n < limit && t.p < t.end) {
    const uint64_t h = t.varint();
    if (h & 1) {
      const uint64_t groups = h >> 1;
      const uint64_t bytes = groups * (uint64_t)bit_width;
      uint64_t count = groups * 8;
      if (groups == 0) continue;
      // the last group of a page may be cut short when it only holds padding
      const uint64_t have = (uint64_t)(t.end - t.p);
      if (bytes > have) {
        if (bit_width == 0) bad_file("corrupt bit-packed run");
        count = have * 8 / (uint64_t)bit_width;
        if (count == 0) break;
      }
      if (count > (uint64_t)(limit - n)) count = (uint64_t)(limit - n);
      out.push_back(PqRun{first + n, (int32_t)count, (uint32_t)(t.p - image), 1u});
      t.p += std::min(bytes, have);
      n += (int)count;
    } else {
      uint64_t count = h >> 1;
      if ((uint64_t)(t.end - t.p) < (uint64_t)value_bytes) bad_file("truncated RLE run");
      uint32_t v = 0;
      for (int i = 0; i < value_bytes; ++i) v |= (uint32_t)t.p[i] << (8 * i);
      t.p += value_bytes;
      if (count == 0) continue;
      if (count > (uint64_t)(limit - n)) count = (uint64_t)(limit - n);
      out.push_back(PqRun{first + n, (int32_t)count, v, 0u});
      n += (int)count;
    }
  }
  return n;
}

size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// Host threads for the two bulk jobs of a scan: copying the file's bytes out of the page cache and
// decompressing pages.  Default: the CPUs this process may run on (its affinity mask, which is what a
// container's share shows up as), at most 16; QE_IO_THREADS overrides.
size_t io_threads() {
  static const size_t n = [] {
    const char *e = getenv("QE_IO_THREADS");
    long v = e ? atol(e) : 0;
    if (v <= 0) {
      cpu_set_t set;
      CPU_ZERO(&set);
      long allowed = sched_getaffinity(0, sizeof set, &set) == 0 ? (long)CPU_COUNT(&set) : (long)std::thread::hardware_concurrency();
      // (one CPU is left to the thread that runs the scan when the process has sixteen or fewer)
      v = std::min<long>(16, std::max<long>(1, allowed <= 16 ? allowed - 1 : allowed));
    }
    return (size_t)std::min<long>(v, 64);
  }();
  return n;
}

// The host threads of the bulk jobs, started once and kept: column chunks are read in slices by up to sixteen of them
// (jobs are taken in the order they were handed in, so the slices of the chunks read ahead land in file order), and
// page decompression runs on them.  Never torn down (the threads wait on a condition variable until the process ends);
// its jobs never wait for other jobs of the pool, so callers on any thread may share it.
class IoPool {
 public:
  static IoPool &get() {
    static IoPool *pool = new IoPool();
    return *pool;
  }
  void submit(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> g(m_);
      jobs_.push_back(std::move(job));
    }
    cv_.notify_one();
  }

 private:
  IoPool() {
    const size_t n = io_threads();
    for (size_t i = 0; i < n; ++i)
      std::thread([this] {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [this] { return !jobs_.empty(); });
            job = std::move(jobs_.front());
            jobs_.pop_front();
          }
          job();
        }
      }).detach();
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> jobs_;
};

template <typename F>
void parallel_for(size_t n, F fn) {
  const size_t t = std::min(n, io_threads());
  if (t <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  struct Shared {
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex m;
    std::condition_variable done;
    size_t helpers_left = 0;
  };
  auto sh = std::make_shared<Shared>();
  auto work = [sh, n, &fn] {
    try {
      for (;;) {
        const size_t i = sh->next.fetch_add(1);
        if (i >= n) break;
        fn(i);
      }
    } catch (...) {
      std::lock_guard<std::mutex> g(sh->m);
      if (!sh->err) sh->err = std::current_exception();
      sh->next.store(n);
    }
  };
  sh->helpers_left = t - 1;
  for (size_t k = 0; k + 1 < t; ++k)
    IoPool::get().submit([sh, work] {
      work();  // (`fn` lives until the caller has seen every helper leave: it waits below)
      std::lock_guard<std::mutex> g(sh->m);
      if (--sh->helpers_left == 0) sh->done.notify_all();
    });
  work();
  {
    std::unique_lock<std::mutex> g(sh->m);
    sh->done.wait(g, [&] { return sh->helpers_left == 0; });
  }
  if (sh->err) std::rethrow_exception(sh->err);
}

// The staging buffer to fill next, at least `bytes` large; waits until the copy that last read it is done.
Context::Staging &take_staging(Context *ctx, size_t bytes) {
  Context::Staging &s = ctx->staging[ctx->next_staging];
  ctx->next_staging = (ctx->next_staging + 1) % Context::STAGING_BUFFERS;
  ctx->bind();
  if (s.pending) {
    QE_HIP(hipEventSynchronize(s.copied));
    s.pending = false;
  }
  if (bytes > s.cap) {
    if (s.ptr) QE_HIP(hipHostFree(s.ptr));
    s.ptr = nullptr;
    s.cap = 0;
    const size_t cap = std::max(bytes + bytes / 4, (size_t)1 << 20);
    QE_HIP(hipHostMalloc(&s.ptr, cap, hipHostMallocDefault));
    s.cap = cap;
  }
  if (!s.copied) QE_HIP(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
  return s;
}

void read_exact(const ParquetFile &f, int64_t off, uint8_t *dst, size_t n) {
  size_t done = 0;
  while (done < n) {
    const ssize_t r = pread(f.fd, dst + done, n - done, off + (int64_t)done);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) fail(QE_ERR_IO, "IO error: parquet: cannot read " + f.path + (r < 0 ? std::string(": ") + strerror(errno) : ": unexpected end of file"));
    done += (size_t)r;
  }
}

void read_parallel(const ParquetFile &f, int64_t off, uint8_t *dst, size_t n) {
  const size_t slice = (size_t)4 << 20;
  parallel_for((n + slice - 1) / slice, [&](size_t i) { read_exact(f, off + (int64_t)(i * slice), dst + i * slice, std::min(slice, n - i * slice)); });
}

struct PageRec {
  PageHeader h;
  size_t body;  // offset of the page body in the raw chunk
};

// QE_PARQUET_TRACE=1: where the host's time goes, per column chunk, on stderr (waiting for a staging buffer, reading the
// file, walking the pages, queueing the copy and the decode kernels).
static bool parquet_trace() {
  static const bool on = getenv("QE_PARQUET_TRACE") != nullptr;
  return on;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Pinned bytes a chunk that is stored uncompressed needs: the chunk as it lies in the file, then room for the descriptor tables.
static size_t plain_chunk_staging_bytes(const ParquetChunkMeta &cm) {
  const size_t raw_len = (size_t)cm.size;
  return align16(raw_len + 16) + ((size_t)1 << 16) + raw_len / 8;
}

Column decode_chunk(Context *ctx, ParquetFile &f, const ParquetColumn &col, const ParquetChunkMeta &cm, int64_t n_rows, ParquetReadAhead::Read *ahead = nullptr) {
  const double t_start = parquet_trace() ? now_ms() : 0.0;
  double t_staged = 0.0, t_read = 0.0, t_walked = 0.0;
  if (!col.type)
    fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Physical type `" + col.type_text + "` is not supported (parquet column `" + col.name + "`).");
  if (cm.start < 4 || cm.size < 0 || cm.start + cm.size > f.file_size) bad_file("column chunk outside the file");
  if (n_rows >= (int64_t)1 << 31) bad_file("row groups of 2^31 rows or more are not supported");

  // ---- raw bytes and page headers
  // Sizes of the descriptor tables are only known after the pages have been walked; the worst
  // case (a run header every other byte) cannot exceed a small multiple of the chunk itself, and
  // the common case is tiny, so the staging buffer gets a fixed allowance and grows if it must.
  const size_t raw_len = (size_t)cm.size;
  const size_t table_room = ((size_t)1 << 16) + raw_len / 8;
  uint8_t *raw;
  Context::Staging *stage = nullptr;
  (void)table_room;
  BufferPtr dev;          // the chunk's image on the device, then its tables
  size_t dev_copied = 0;  // bytes of the staging buffer that are on their way there already
  if (ahead && ahead->stage) {
    // Read ahead of time into a staging buffer of its own, and copied to the device slice by slice by its readers:
    // wait until the last of those copies has been queued (a reader's failure is thrown here, once its fellow readers
    // are through), and put this stream's work on the chunk behind them.
    stage = ahead->stage;
    raw = (uint8_t *)stage->ptr;
    if (parquet_trace()) t_staged = now_ms();
    ParquetReadAhead::Progress &pr = *ahead->progress;
    pr.wait_all_awake();
    const double t_woken = parquet_trace() ? now_ms() : 0.0;
    if (pr.err) std::rethrow_exception(pr.err);
    ctx->bind();
    QE_HIP(hipStreamWaitEvent(ctx->stream, stage->images, 0));
    dev = ahead->dev;
    dev_copied = raw_len;
    if (parquet_trace())
      fprintf(stderr,
              "qe_gpu parquet: column %s: its %zu slices were read and queued in %.3f ms (%.1f GB/s; reading %.3f ms a slice, queueing %.3f, at most %.3f), the last %.3f ms %s its "
              "turn came; woken %.3f ms after it, stream wait queued in %.3f ms\n",
              col.name.c_str(), pr.n_slices, pr.last_end - pr.first_start, (double)raw_len / 1e6 / std::max(pr.last_end - pr.first_start, 1e-6),
              pr.reading / (double)std::max<size_t>(pr.n_slices, 1), pr.queueing / (double)std::max<size_t>(pr.n_slices, 1), pr.longest_queueing, std::abs(pr.last_end - t_staged),
              pr.last_end < t_staged ? "before" : "after", t_woken - std::max(pr.last_end, t_staged), now_ms() - t_woken);
    ahead->progress.reset();  // (from here on the device block is this stream's to hand back)
  } else {
    if (cm.codec == CODEC_NONE) {
      // the image is the chunk as it lies in the file: read it straight into pinned memory
      stage = &take_staging(ctx, plain_chunk_staging_bytes(cm));
      raw = (uint8_t *)stage->ptr;
    } else {
      if (ctx->host_scratch.size() < raw_len + 16) ctx->host_scratch.resize(raw_len + raw_len / 4 + 16);
      raw = ctx->host_scratch.data();
    }
    if (parquet_trace()) t_staged = now_ms();
    read_parallel(f, cm.start, raw, raw_len);
  }
  if (parquet_trace()) t_read = now_ms();
  std::vector<PageRec> recs;
  size_t image_need = 0;
  {
    size_t pos = 0;
    int64_t seen = 0;
    while (pos < raw_len && seen < cm.num_values) {
      Thrift t{raw + pos, raw + raw_len};
      PageRec r;
      r.h = read_page_header(t);
      r.body = (size_t)(t.p - raw);
      if ((size_t)r.h.compressed > raw_len - r.body) bad_file("page runs past its column chunk");
      pos = r.body + (size_t)r.h.compressed;
      if (r.h.type == PAGE_DATA || r.h.type == PAGE_DATA_V2) seen += r.h.num_values;
      if (r.h.type == PAGE_DATA || r.h.type == PAGE_DATA_V2 || r.h.type == PAGE_DICTIONARY) {
        recs.push_back(r);
        image_need += align16((size_t)r.h.uncompressed + 16);
      }
    }
    if (seen != n_rows) bad_file("column `" + col.name + "`: pages hold " + std::to_string(seen) + " values, the row group " + std::to_string(n_rows) + " rows");
  }

  // ---- image: page bodies, decompressed where needed (pages are independent: one task each)
  uint8_t *image;
  size_t image_len;
  std::vector<size_t> body_at(recs.size());  // image offset of each page's (uncompressed) body
  if (cm.codec == CODEC_NONE) {
    image = raw;
    image_len = raw_len;
    for (size_t i = 0; i < recs.size(); ++i) {
      if (recs[i].h.compressed != recs[i].h.uncompressed) bad_file("page sizes disagree");
      body_at[i] = recs[i].body;
    }
  } else {
    stage = &take_staging(ctx, image_need + ((size_t)1 << 16) + image_need / 8);
    image = (uint8_t *)stage->ptr;
    size_t at = 0;
    for (size_t i = 0; i < recs.size(); ++i) {
      body_at[i] = at;
      at += align16((size_t)recs[i].h.uncompressed + 16);
    }
    image_len = at;
    parallel_for(recs.size(), [&](size_t i) {
      const PageHeader &h = recs[i].h;
      const uint8_t *src = raw + recs[i].body;
      uint8_t *dst = image + body_at[i];
      if (h.type == PAGE_DATA_V2) {
        // levels are stored uncompressed in front of the (optionally compressed) values
        const size_t lv = (size_t)h.def_len + (size_t)h.rep_len;
        if (lv > (size_t)h.compressed || lv > (size_t)h.uncompressed) bad_file("corrupt v2 page");
        memcpy(dst, src, lv);
        decompress(h.v2_compressed ? cm.codec : CODEC_NONE, src + lv, (size_t)h.compressed - lv, dst + lv, (size_t)h.uncompressed - lv);
      } else {
        decompress(cm.codec, src, (size_t)h.compressed, dst, (size_t)h.uncompressed);
      }
      memset(dst + h.uncompressed, 0, align16((size_t)h.uncompressed + 16) - (size_t)h.uncompressed);
    });
  }
  if (image_len >= ((size_t)1 << 32) - 64) bad_file("column chunks of 4 GiB or more are not supported");

  // ---- descriptors
  std::vector<PqPage> pages;
  std::vector<PqRun> level_runs, runs;
  std::vector<PqWalk> walks;
  uint32_t dict_off = 0;
  int dict_count = 0;
  bool have_dict = false;
  int32_t row = 0;
  for (size_t i = 0; i < recs.size(); ++i) {
    const PageHeader &h = recs[i].h;
    const uint32_t body = (uint32_t)body_at[i], body_len = (uint32_t)h.uncompressed;
    if (h.type == PAGE_DICTIONARY) {
      if (h.encoding != ENC_PLAIN && h.encoding != ENC_PLAIN_DICTIONARY) bad_file("dictionary page encoding " + std::to_string(h.encoding) + " is not supported");
      if (have_dict) bad_file("more than one dictionary page in a column chunk");
      have_dict = true;
      dict_off = body;
      dict_count = h.num_values;
      if (col.physical == PQ_BYTE_ARRAY) {
        if ((uint64_t)dict_count * 4 > body_len) bad_file("corrupt dictionary page");  // every entry has a 4-byte length
        walks.push_back(PqWalk{body, body_len, -1, -1, dict_count});
      }
      else {
        const uint64_t w = (col.physical == PQ_INT32 || col.physical == PQ_FLOAT) ? 4 : col.physical == PQ_INT96 ? 12 : 8;
        if (col.physical == PQ_BOOLEAN || (uint64_t)dict_count * w > body_len) bad_file("corrupt dictionary page");
      }
      continue;
    }
    uint32_t p = body, end = body + body_len;
    if (h.type == PAGE_DATA) {
      if (col.optional) {
        if (h.def_encoding != ENC_RLE) bad_file("definition levels must be RLE encoded");
        if (body_len < 4) bad_file("truncated data page");
        uint32_t ln;
        memcpy(&ln, image + p, 4);
        if (ln > body_len - 4) bad_file("definition levels run past their page");
        if (walk_runs(image, p + 4, ln, 1, row, h.num_values, level_runs) != h.num_values) bad_file("definition levels end early");
        p += 4 + ln;
      }
    } else {
      if (h.rep_len) bad_file("repetition levels are not supported (nested data)");
      if ((uint32_t)h.def_len > body_len) bad_file("corrupt v2 page");
      if (col.optional) {
        if (walk_runs(image, p, (uint32_t)h.def_len, 1, row, h.num_values, level_runs) != h.num_values) bad_file("definition levels end early");
      }
      p += (uint32_t)h.def_len;
    }
    PqPage pg{};
    pg.first_row = row;
    pg.num_rows = h.num_values;
    pg.run_begin = pg.run_end = (int32_t)runs.size();
    if (h.encoding == ENC_PLAIN) {
      pg.encoding = PQ_ENC_PLAIN;
      pg.values_off = p;
      pg.values_len = end - p;
      if (col.physical == PQ_BYTE_ARRAY) walks.push_back(PqWalk{p, end - p, row, row + h.num_values, 0});
    } else if (h.encoding == ENC_PLAIN_DICTIONARY || h.encoding == ENC_RLE_DICTIONARY) {
      if (!have_dict) bad_file("dictionary encoded page without a dictionary");
      pg.encoding = PQ_ENC_DICT;
      if (end - p < 1) {
        if (h.num_values) {
          // a page of nulls only may carry no index bytes at all
          pg.bit_width = 0;
          runs.push_back(PqRun{0, h.num_values, 0u, 0u});
        }
      } else {
        pg.bit_width = image[p];
        if (pg.bit_width > 32) bad_file("corrupt dictionary index width");
        walk_runs(image, p + 1, end - p - 1, pg.bit_width, 0, h.num_values, runs);
      }
      pg.run_end = (int32_t)runs.size();
      if (pg.run_end == pg.run_begin) runs.push_back(PqRun{0, std::max(h.num_values, 1), 0u, 0u}), pg.run_end++;
    } else if (h.encoding == ENC_RLE && col.physical == PQ_BOOLEAN) {
      pg.encoding = PQ_ENC_RLE_BOOL;
      if (end - p < 4) bad_file("truncated boolean page");
      uint32_t ln;
      memcpy(&ln, image + p, 4);
      if (ln > end - p - 4) bad_file("boolean runs past their page");
      walk_runs(image, p + 4, ln, 1, 0, h.num_values, runs);
      pg.run_end = (int32_t)runs.size();
      if (pg.run_end == pg.run_begin) runs.push_back(PqRun{0, std::max(h.num_values, 1), 0u, 0u}), pg.run_end++;
    } else {
      fail(QE_ERR_ARROW, "Arrow error: parquet: encoding " + std::to_string(h.encoding) + " of column `" + col.name + "` is not supported");
    }
    if (h.num_values) pages.push_back(pg);
    row += h.num_values;
  }

  // ---- one host -> device copy: image, then the tables
  const size_t n_words = ((size_t)n_rows + 63) / 64;
  size_t at = align16(image_len + 16);
  const size_t pages_at = at;
  at = align16(at + pages.size() * sizeof(PqPage));
  const size_t runs_at = at;
  at = align16(at + runs.size() * sizeof(PqRun));
  const size_t level_runs_at = at;
  at = align16(at + level_runs.size() * sizeof(PqRun));
  const size_t walks_at = at;
  at = align16(at + walks.size() * sizeof(PqWalk));
  const size_t total = at + 16;
  if (total > stage->cap) {
    // tables did not fit behind the image: this staging buffer grows (nobody else holds it: its last copy was waited
    // for when it was handed out; copies of this chunk's own slices are waited for here)
    void *bigger = nullptr;
    const size_t cap = total + total / 4;
    ctx->bind();
    if (dev_copied) {
      ctx->sync();
      dev_copied = 0;
      dev.reset();
    }
    QE_HIP(hipHostMalloc(&bigger, cap, hipHostMallocDefault));
    memcpy(bigger, image, image_len);
    QE_HIP(hipHostFree(stage->ptr));
    stage->ptr = bigger;
    stage->cap = cap;
    image = (uint8_t *)bigger;
  }
  uint8_t *st = (uint8_t *)stage->ptr;
  memset(st + image_len, 0, pages_at - image_len);
  if (!pages.empty()) memcpy(st + pages_at, pages.data(), pages.size() * sizeof(PqPage));
  if (!runs.empty()) memcpy(st + runs_at, runs.data(), runs.size() * sizeof(PqRun));
  if (!level_runs.empty()) memcpy(st + level_runs_at, level_runs.data(), level_runs.size() * sizeof(PqRun));
  if (!walks.empty()) memcpy(st + walks_at, walks.data(), walks.size() * sizeof(PqWalk));

  if (parquet_trace()) t_walked = now_ms();
  ctx->bind();
  // the image (what of it is not on its way yet) and the tables behind it, in one copy
  if (!dev || total > dev->bytes) {
    dev = device_buffer(ctx, total);
    dev_copied = 0;
  }
  QE_HIP(hipMemcpyAsync((uint8_t *)dev->ptr + dev_copied, st + dev_copied, total - dev_copied, hipMemcpyHostToDevice, ctx->stream));
  QE_HIP(hipEventRecord(stage->copied, ctx->stream));
  stage->pending = true;
  const uint8_t *d = (const uint8_t *)dev->ptr;

  unsigned long long *err = ctx->counters + 32;  // cleared and checked once per row group

  Column out;
  out.type = col.type;
  out.space = QE_MEM_DEVICE;
  out.length = n_rows;
  out.null_count = 0;

  PqChunk ck{};
  ck.image = d;
  ck.image_len = (uint32_t)image_len;
  ck.pages = (const PqPage *)(d + pages_at);
  ck.n_pages = (int32_t)pages.size();
  ck.runs = (const PqRun *)(d + runs_at);
  ck.n_rows = (int32_t)n_rows;
  ck.physical = col.physical;
  ck.dict_off = dict_off;
  ck.dict_count = dict_count;
  ck.error = err;

  BufferPtr word_rank;
  if (col.optional) {
    // validity: the Arrow bitmap itself (64 rows per word), and the rank table the value kernels need
    out.validity = device_buffer(ctx, std::max(bitmap_bytes(n_rows), n_words * 8 + 32));
    QE_HIP(hipMemsetAsync(out.validity->ptr, 0, out.validity->bytes, ctx->stream));
    // An optional column without a single null is the common case (writers mark columns optional by
    // default): its levels are nothing but repeated runs of 1, which the host has just walked.  The
    // bitmap is then a fill, and the value kernels address values by row, as for a required column.
    bool all_valid = true;
    for (const PqRun &r : level_runs) all_valid = all_valid && !r.packed && r.data == 1u;
    if (all_valid) {
      out.null_count = 0;
      out.known_all_valid = true;
      if (n_rows / 8) QE_HIP(hipMemsetAsync(out.validity->ptr, 0xff, (size_t)(n_rows / 8), ctx->stream));
      if (n_rows % 8) QE_HIP(hipMemsetAsync((uint8_t *)out.validity->ptr + n_rows / 8, (1 << (n_rows % 8)) - 1, 1, ctx->stream));
    } else {
      word_rank = device_buffer(ctx, (n_words + 1) * 4 + 64);
      QE_HIP(hipMemsetAsync(word_rank->ptr, 0, (n_words + 1) * 4, ctx->stream));
      out.null_count = -1;
      ctx->prof_begin("qe_parquet_validity_kernel");
      QE_HIP(launch_pq_validity(d, (const PqRun *)(d + level_runs_at), (int)level_runs.size(), (int32_t)n_rows, (uint64_t *)out.validity->ptr,
                                (int32_t *)word_rank->ptr, ctx->stream));
      ctx->prof_end();
      BufferPtr tmp = device_buffer(ctx, scan_scratch_elems((int64_t)n_words + 1) * 4 + 64);
      QE_HIP(launch_exclusive_scan_i32((int32_t *)word_rank->ptr, (int64_t)n_words + 1, (int32_t *)tmp->ptr, ctx->stream));
      ck.validity = (const uint64_t *)out.validity->ptr;
      ck.word_rank = (const int32_t *)word_rank->ptr;
    }
  }

  if (col.physical != PQ_BYTE_ARRAY) {
    const bool boolean = col.type == QE_TYPE_BOOLEAN;
    const size_t vb = boolean ? std::max(bitmap_bytes(n_rows), n_words * 8 + 32) : (size_t)n_rows * (size_t)type_width(col.type) + 16;
    out.values = device_buffer(ctx, vb);
    out.values_bytes = boolean ? (n_rows + 7) / 8 : n_rows * type_width(col.type);
    if (boolean) QE_HIP(hipMemsetAsync(out.values->ptr, 0, vb, ctx->stream));
    ctx->prof_begin("qe_parquet_fixed_kernel");
    QE_HIP(launch_pq_fixed(ck, out.values->ptr, ctx->stream));
    ctx->prof_end();
  } else {
    // strings: prefix walk (one thread per page), per-row source and length, scan, copy
    BufferPtr value_off = device_buffer(ctx, (size_t)n_rows * 4 + 16), value_len = device_buffer(ctx, (size_t)n_rows * 4 + 16);
    BufferPtr dict_entry_off = device_buffer(ctx, (size_t)dict_count * 4 + 16), dict_entry_len = device_buffer(ctx, (size_t)dict_count * 4 + 16);
    BufferPtr row_src = device_buffer(ctx, (size_t)n_rows * 4 + 16);
    out.offsets = device_buffer(ctx, ((size_t)n_rows + 1) * 4 + 16);
    int32_t *offs = (int32_t *)out.offsets->ptr;
    QE_HIP(hipMemsetAsync(offs + n_rows, 0, 4, ctx->stream));
    ctx->prof_begin("qe_parquet_walk_kernel");
    QE_HIP(launch_pq_walk(ck, (const PqWalk *)(d + walks_at), (int)walks.size(), (uint32_t *)value_off->ptr, (int32_t *)value_len->ptr,
                          (uint32_t *)dict_entry_off->ptr, (int32_t *)dict_entry_len->ptr, ctx->stream));
    ctx->prof_end();
    ctx->prof_begin("qe_parquet_utf8_rows_kernel");
    QE_HIP(launch_pq_utf8_rows(ck, (const uint32_t *)value_off->ptr, (const int32_t *)value_len->ptr, (const uint32_t *)dict_entry_off->ptr,
                               (const int32_t *)dict_entry_len->ptr, (uint32_t *)row_src->ptr, offs, ctx->stream));
    ctx->prof_end();
    BufferPtr tmp = device_buffer(ctx, scan_scratch_elems(n_rows + 1) * 4 + 64);
    QE_HIP(launch_exclusive_scan_i32(offs, n_rows + 1, (int32_t *)tmp->ptr, ctx->stream));
    QE_HIP(hipMemcpyAsync(ctx->host_counters + 8, offs + n_rows, 4, hipMemcpyDeviceToHost, ctx->stream));
    QE_HIP(hipMemcpyAsync(ctx->host_counters + 10, err, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    QE_HIP(hipMemsetAsync(err + 1, 0, sizeof(unsigned long long), ctx->stream));
    ctx->sync();
    if (ctx->host_counters[10]) bad_file("column `" + col.name + "`: page contents contradict their headers");
    if (ctx->host_counters[11] > 2147483647ull)
      fail(QE_ERR_ARROW, "Arrow error: parquet: column `" + col.name + "` exceeds int32 offsets in one row group");
    const int32_t bytes = *(int32_t *)(ctx->host_counters + 8);
    out.values_bytes = bytes;
    out.values = device_buffer(ctx, (size_t)bytes + 16);
    ctx->prof_begin("qe_parquet_utf8_copy_kernel");
    QE_HIP(launch_pq_utf8_copy(d, (const uint32_t *)row_src->ptr, offs, (int32_t)n_rows, (uint8_t *)out.values->ptr, ctx->stream));
    ctx->prof_end();
  }
  // Temporaries go back to the pool here while the stream may still use them: the pool only hands
  // blocks to work queued later on the same stream.
  if (parquet_trace()) {
    const double t_end = now_ms();
    fprintf(stderr, "qe_gpu parquet: column %s, %.1f MB: staging buffer %.3f ms, read %.3f ms (%.1f GB/s), pages %.3f ms, queueing %.3f ms\n", col.name.c_str(),
            (double)raw_len / 1e6, t_staged - t_start, t_read - t_staged, (double)raw_len / 1e6 / std::max(t_read - t_staged, 1e-6), t_walked - t_read, t_end - t_walked);
  }
  return out;
}

}  // namespace

void ParquetReadAhead::Read::settle() {
  if (!progress) return;
  progress->wait_all();
  progress.reset();
  // never decoded (the scan went another way, or ended early): the copies queued into the block are waited for
  if (ctx && ctx->copy_stream) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->copy_stream);
  }
}

ParquetFile::~ParquetFile() {
  if (fd >= 0) close(fd);
}

std::shared_ptr<ParquetFile> parquet_open(const std::string &path) {
  auto f = std::make_shared<ParquetFile>();
  f->path = path;
  f->fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (f->fd < 0) fail(QE_ERR_IO, "IO error: " + path + ": " + strerror(errno));
  struct stat st;
  if (fstat(f->fd, &st) != 0) fail(QE_ERR_IO, "IO error: " + path + ": " + strerror(errno));
  f->file_size = st.st_size;
  if (f->file_size < 12) bad_file(path + " is too short to be a parquet file");
  uint8_t tail[8];
  read_exact(*f, f->file_size - 8, tail, 8);
  uint8_t head[4];
  read_exact(*f, 0, head, 4);
  if (memcmp(tail + 4, "PAR1", 4) != 0 || memcmp(head, "PAR1", 4) != 0) bad_file(path + " does not carry the PAR1 magic (encrypted footers are not supported)");
  uint32_t meta_len;
  memcpy(&meta_len, tail, 4);
  if ((int64_t)meta_len > f->file_size - 12) bad_file("footer length exceeds the file");
  std::vector<uint8_t> meta(meta_len);
  read_exact(*f, f->file_size - 8 - (int64_t)meta_len, meta.data(), meta_len);

  Thrift t{meta.data(), meta.data() + meta.size()};
  std::vector<SchemaElement> schema;
  int last = 0, ty;
  bool have_rows = false;
  std::vector<std::vector<std::pair<ParquetChunkMeta, int>>> groups;
  std::vector<int64_t> group_rows;
  while (t.field(last, ty)) {
    if (last == 2 && ty == 9) {
      int n, et;
      t.list(n, et);
      for (int i = 0; i < n; ++i) schema.push_back(read_schema_element(t));
    } else if (last == 3) {
      f->num_rows = t.zigzag();
      have_rows = true;
    } else if (last == 4 && ty == 9) {
      int n, et;
      t.list(n, et);
      for (int g = 0; g < n; ++g) {
        std::vector<std::pair<ParquetChunkMeta, int>> chunks;
        int64_t rows = -1;
        int l2 = 0, t2;
        while (t.field(l2, t2)) {
          if (l2 == 1 && t2 == 9) {
            int nc, ct;
            t.list(nc, ct);
            for (int c = 0; c < nc; ++c) {
              int physical = -1;
              ParquetChunkMeta m = read_column_chunk(t, &physical);
              chunks.emplace_back(m, physical);
            }
          } else if (l2 == 3) {
            rows = t.zigzag();
          } else {
            t.skip(t2);
          }
        }
        if (rows < 0) bad_file("row group without a row count");
        groups.push_back(chunks);
        group_rows.push_back(rows);
      }
    } else {
      t.skip(ty);
    }
  }
  if (schema.empty() || !have_rows) bad_file("footer without a schema");
  if (schema[0].num_children != (int)schema.size() - 1) bad_file("nested schemas are not supported");
  for (size_t i = 1; i < schema.size(); ++i) {
    const SchemaElement &e = schema[i];
    if (e.num_children > 0 || e.repetition == 2) bad_file("nested or repeated field `" + e.name + "` is not supported");
    ParquetColumn c;
    c.name = e.name;
    c.physical = e.type;
    c.optional = e.repetition == 1;
    c.type_text = physical_name(e.type);
    switch (e.type) {
      case PQ_BOOLEAN: c.type = QE_TYPE_BOOLEAN; break;
      case PQ_INT32: c.type = (e.converted < 0 || e.converted == 17) ? QE_TYPE_INT32 : 0; break;
      // plain 64-bit integers; millisecond / microsecond timestamps keep their raw count
      case PQ_INT64: c.type = (e.logical == 8 && e.time_unit == 3) ? QE_TYPE_TIMESTAMP_NS : QE_TYPE_INT64; break;
      case PQ_INT96: c.type = QE_TYPE_TIMESTAMP_NS; break;
      case PQ_DOUBLE: c.type = QE_TYPE_FLOAT64; break;
      case PQ_FLOAT: c.type = QE_TYPE_FLOAT64; break;  // widened on decode (arrow2 would say Float32, which the reference cannot evaluate)
      case PQ_BYTE_ARRAY: c.type = (e.converted == 0 || e.logical == 1) ? QE_TYPE_UTF8 : 0; break;
      default: c.type = 0;
    }
    if (!c.type && e.converted >= 0) c.type_text += " (converted type " + std::to_string(e.converted) + ")";
    f->columns.push_back(c);
  }
  for (size_t g = 0; g < groups.size(); ++g) {
    if (groups[g].size() != f->columns.size()) bad_file("row group with a different number of columns than the schema");
    ParquetRowGroup rg;
    rg.num_rows = group_rows[g];
    for (size_t c = 0; c < groups[g].size(); ++c) {
      if (groups[g][c].second != f->columns[c].physical) bad_file("column chunk type differs from the schema");
      rg.chunks.push_back(groups[g][c].first);
    }
    f->row_groups.push_back(rg);
  }
  int64_t group_total = 0;
  for (auto &rg : f->row_groups) group_total += rg.num_rows;
  if (group_total != f->num_rows)
    bad_file("the row groups hold " + std::to_string(group_total) + " rows, the footer says " + std::to_string(f->num_rows));
  return f;
}

std::vector<int> parquet_project(const ParquetFile &file, const std::vector<std::string> &projection) {
  std::vector<int> out;
  for (size_t i = 0; i < file.columns.size(); ++i)
    if (std::find(projection.begin(), projection.end(), file.columns[i].name) != projection.end()) out.push_back((int)i);
  return out;
}

// Chunks in flight ahead of the one being decoded, the current one included: as many as hold READ_AHEAD_BYTES between them
// (the readers must not run dry while the scan waits for the device at the end of a row group), at most
// READ_AHEAD_DEPTH.  Each holds a staging buffer of the ring; the ring has three more, so that the buffer a new read
// takes is one whose copy to the device was queued at least three chunks ago and has long finished - the host never
// waits for it.
constexpr int READ_AHEAD_DEPTH = Context::STAGING_BUFFERS - 3;
constexpr size_t READ_AHEAD_BYTES = (size_t)192 << 20;
// (slices of 1 MiB and less at the start of a scan, to get the first bytes to the device sooner, cost more in copies
// than they gained: 9.2-9.9 ms against 8.85 for the 392 MB file of the benchmark)
static size_t read_ahead_slice() {
  static const size_t v = [] {
    const char *e = getenv("QE_PARQUET_SLICE_KB");
    const long kb = e ? atol(e) : 4096;
    return (size_t)std::min<long>(std::max<long>(kb, 64), 65536) << 10;
  }();
  return v;
}

static size_t read_ahead_copy_slices() {
  static const size_t v = [] {
    const char *e = getenv("QE_PARQUET_COPY_SLICES");
    return (size_t)std::min<long>(std::max<long>(e ? atol(e) : 4, 1), 64);
  }();
  return v;
}

// Starts reading chunk (row_group, column) into the next staging buffer, a job of the pool per slice - for chunks that
// are stored uncompressed (the file's bytes are the image the device decodes); others are read when their turn comes.
static void read_ahead_issue(Context *ctx, ParquetFile &file, ParquetReadAhead &ahead, int row_group, int column, bool scans_last = false) {
  ParquetReadAhead::Read r;
  r.row_group = row_group;
  r.column = column;
  const ParquetChunkMeta &cm = file.row_groups[(size_t)row_group].chunks[(size_t)column];
  const bool plain = cm.codec == CODEC_NONE && file.columns[(size_t)column].type != 0 && cm.start >= 4 && cm.size >= 0 && cm.start + cm.size <= file.file_size;
  if (plain) {
    r.ctx = ctx;
    r.stage = &take_staging(ctx, plain_chunk_staging_bytes(cm));
    r.bytes = (size_t)cm.size;
    r.dev = device_buffer(ctx, plain_chunk_staging_bytes(cm));
    Context::Staging &st = *r.stage;
    if (!ctx->copy_stream) QE_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!st.fence) QE_HIP(hipEventCreateWithFlags(&st.fence, hipEventDisableTiming));
    if (!st.images) QE_HIP(hipEventCreateWithFlags(&st.images, hipEventDisableTiming));
    // the device block may be one that kernels queued on the context's stream still read (blocks go back to the pool
    // while the stream is busy with them): the copies into it wait for what that stream holds now
    QE_HIP(hipEventRecord(st.fence, ctx->stream));
    QE_HIP(hipStreamWaitEvent(ctx->copy_stream, st.fence, 0));
    r.progress = std::make_shared<ParquetReadAhead::Progress>();
    const size_t slice = read_ahead_slice(), len = r.bytes;
    const size_t n_slices = (len + slice - 1) / slice;
    r.progress->n_slices = n_slices;
    uint8_t *dst = (uint8_t *)st.ptr, *dev = (uint8_t *)r.dev->ptr;
    const ParquetFile *f = &file;  // (outlives the jobs: a Read is not let go of before they are through)
    const int64_t start = cm.start;
    const int device = ctx->device;
    hipStream_t copy_stream = ctx->copy_stream;
    hipEvent_t images = st.images;
    const bool trace = parquet_trace();
    if (n_slices == 0) QE_HIP(hipEventRecord(images, copy_stream));
    // slices are what a reader reads at a time, groups of them what travels in one copy (a copy has a fixed cost on
    // the device's side: 4 MiB copies reached 44 GB/s where 8 MiB ones reached 47-49)
    // (the scan's last chunk: a copy per slice - nothing else is left to keep the link busy while a group fills)
    const size_t per_group = scans_last ? 1 : read_ahead_copy_slices();
    r.progress->group_landed.assign((n_slices + per_group - 1) / per_group, 0);
    for (size_t i = 0; i < n_slices; ++i)
      IoPool::get().submit([pr = r.progress, f, start, dst, dev, len, slice, i, trace, device, copy_stream, images, per_group, n_slices] {
        const double t0 = trace ? now_ms() : 0.0;
        double t1 = 0.0;
        std::exception_ptr err;
        try {
          const size_t at = i * slice, n = std::min(slice, len - at);
          read_exact(*f, start + (int64_t)at, dst + at, n);
          if (trace) t1 = now_ms();
          const size_t g = i / per_group, in_group = std::min(per_group, n_slices - g * per_group);
          bool last_of_group;
          {
            std::lock_guard<std::mutex> lk(pr->m);
            last_of_group = ++pr->group_landed[g] == in_group;
          }
          if (last_of_group) {
            const size_t from = g * per_group * slice, to = std::min(len, from + per_group * slice);
            QE_HIP(hipSetDevice(device));
            QE_HIP(hipMemcpyAsync(dev + from, dst + from, to - from, hipMemcpyHostToDevice, copy_stream));
          }
        } catch (...) {
          err = std::current_exception();
        }
        std::lock_guard<std::mutex> g(pr->m);
        if (err && !pr->err) pr->err = err;
        if (trace) {
          if (pr->n_landed == 0 || t0 < pr->first_start) pr->first_start = t0;
          const double t2 = now_ms();
          pr->last_end = std::max(pr->last_end, t2);
          if (t1 > 0.0) pr->reading += t1 - t0, pr->queueing += t2 - t1, pr->longest_queueing = std::max(pr->longest_queueing, t2 - t1);
        }
        if (++pr->n_landed == pr->n_slices) {
          // every slice's copy has been queued: the event behind them all
          try {
            QE_HIP(hipEventRecord(images, copy_stream));
          } catch (...) {
            if (!pr->err) pr->err = std::current_exception();
          }
          pr->all_landed.store(true, std::memory_order_release);
          pr->cv.notify_all();
        }
      });
  }
  ahead.queue.push_back(std::move(r));
}

Batch parquet_read_row_group(Context *ctx, ParquetFile &file, int row_group, const std::vector<int> &columns, ParquetReadAhead *ahead, int last_row_group) {
  if (row_group < 0 || row_group >= (int)file.row_groups.size()) fail(QE_ERR_EXCEEDING_BOUNDS, "row group index out of range");
  const ParquetRowGroup &rg = file.row_groups[row_group];
  Batch b;
  b.ctx = ctx;
  b.rows = rg.num_rows;
  ctx->bind();
  if (!ctx->counters) {
    QE_HIP(hipMalloc((void **)&ctx->counters, 64 * sizeof(unsigned long long)));
    QE_HIP(hipHostMalloc((void **)&ctx->host_counters, 128 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  unsigned long long *err = ctx->counters + 32;
  QE_HIP(hipMemsetAsync(err, 0, 2 * sizeof(unsigned long long), ctx->stream));
  for (int c : columns)
    if (c < 0 || c >= (int)file.columns.size()) fail(QE_ERR_EXCEEDING_BOUNDS, "column index out of range");
  if (last_row_group < 0 || last_row_group > (int)file.row_groups.size()) last_row_group = (int)file.row_groups.size();
  // the scan order from a chunk on: the rest of this row group, then the next row groups' columns
  auto next_of = [&](int &g, size_t &ci) {
    if (++ci >= columns.size()) {
      ci = 0;
      ++g;
    }
    return g < last_row_group && !columns.empty();
  };
  for (size_t ci = 0; ci < columns.size(); ++ci) {
    const int c = columns[ci];
    ParquetReadAhead::Read mine;
    if (ahead) {
      // reads left over from a scan that went another way (a restart) are waited for and dropped
      while (!ahead->queue.empty() && (ahead->queue.front().row_group != row_group || ahead->queue.front().column != c)) ahead->queue.pop_front();
      if (ahead->queue.empty()) read_ahead_issue(ctx, file, *ahead, row_group, c);
      // top up: the chunks after the last one queued
      int g = ahead->queue.back().row_group;
      size_t at = 0;
      while (at < columns.size() && columns[at] != ahead->queue.back().column) ++at;
      size_t in_flight = 0;
      for (const ParquetReadAhead::Read &r : ahead->queue) in_flight += r.bytes;
      while ((int)ahead->queue.size() < READ_AHEAD_DEPTH && (ahead->queue.size() < 2 || in_flight < READ_AHEAD_BYTES) && at < columns.size() && next_of(g, at)) {
        read_ahead_issue(ctx, file, *ahead, g, columns[at], g == last_row_group - 1 && at + 1 == columns.size());
        in_flight += ahead->queue.back().bytes;
      }
      mine = std::move(ahead->queue.front());
      ahead->queue.pop_front();
    }
    b.cols.push_back(decode_chunk(ctx, file, file.columns[c], rg.chunks[c], rg.num_rows, mine.stage ? &mine : nullptr));
  }
  QE_HIP(hipMemcpyAsync(ctx->host_counters + 9, err, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  const double t_sync = parquet_trace() ? now_ms() : 0.0;
  ctx->sync();
  if (parquet_trace()) fprintf(stderr, "qe_gpu parquet: row group %d: waited %.3f ms for the device\n", row_group, now_ms() - t_sync);
  if (ctx->host_counters[9]) bad_file("row group " + std::to_string(row_group) + ": page contents contradict their headers");
  return b;
}

}  // namespace qe
