// The one exchange step of a multi-GPU aggregate, behind the C ABI: every rank aggregates its row range in
// QE_AGG_MODE_PARTIAL, the partial tables travel over RCCL (xGMI inside a node), a QE_AGG_MODE_FINAL aggregate merges
// what arrives.  The reference is a single-threaded pull iterator (physical_plan/mod.rs:58) and has no counterpart;
// the partial / final split follows its own batch-by-batch merge (physical_plan/mod.rs:298-374).
//
//   qe_exchange_all_gather  few groups: every rank receives every rank's table (one ncclAllGather per column, grouped
//                           into one launch), padded to a common size so that no sizes are negotiated and nothing is
//                           read back to the host;
//   qe_exchange_by_key      many groups: every partial row goes to the rank that owns its key (hash modulo ranks) - a
//                           device-side owner hash + scatter into fixed-size per-peer blocks, then grouped
//                           ncclSend / ncclRecv pairs, W - 1 direct transfers per rank at the same time (xGMI is point
//                           to point: an all-to-all uses every link, a ring would be bound by one).
// Rows that were not filled are all zero in both; a COUNT(*) column tells real rows apart (count > 0).
//
// RCCL is bound at run time (dlopen): a process that also runs PyTorch already has one loaded, and two copies of a
// collective library in one process do not share their bootstrap state.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <mutex>

#include "qe_core.h"
#include "qe_device.h"

namespace qe {
namespace {

// The part of RCCL's interface used here (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclSend / ncclRecv, groups).
struct NcclUniqueId {
  char internal[128];
};
static_assert(sizeof(NcclUniqueId) == QE_COMM_ID_BYTES, "unique id size");
typedef void *NcclComm;
constexpr int NCCL_UINT8 = 1;  // ncclUint8

struct Rccl {
  void *handle = nullptr;
  int (*GetVersion)(int *) = nullptr;
  int (*GetUniqueId)(NcclUniqueId *) = nullptr;
  int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  std::string error;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already in the process first (PyTorch's), then the system's
    const char *loaded[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : loaded)
      if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
      if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) {
      r.error = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "librccl.so");
      return;
    }
    auto sym = [&](const char *name) {
      void *p = dlsym(r.handle, name);
      if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + name;
      return p;
    };
    r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  });
  if (!r.error.empty()) fail(QE_ERR_DEVICE, r.error);
  return r;
}

void nccl_check(int rc, const char *what) {
  if (rc != 0) fail(QE_ERR_DEVICE, std::string(what) + ": " + rccl().GetErrorString(rc));
}
#define QE_NCCL(expr) nccl_check((expr), #expr)

// ---------------------------------------------------------------- kernels
constexpr int MAX_WORLD = 64;
constexpr int MAX_XCOLS = 24;

struct OwnerArgs {
  const void *key_values[4];
  int key_width[4];  // 4 or 8 bytes
  int n_keys;
  int world;
  int64_t rows;
  int64_t cap;        // rows per peer block
  const void *src[MAX_XCOLS];
  void *dst[MAX_XCOLS];  // [world][cap] per column
  int width[MAX_XCOLS];
  const uint8_t *src_valid[MAX_XCOLS];  // validity bitmap of a column that can be null, else nullptr
  uint32_t *dst_valid[MAX_XCOLS];       // [world][cap] bits, zeroed
  int n_cols;
  unsigned int *counters;          // [world], zeroed: rows placed per peer
  unsigned long long *overflow;    // incremented when a peer's block is full
};

__device__ __forceinline__ uint32_t owner_of(const OwnerArgs &A, int64_t r) {
  uint64_t h = 0;
  for (int k = 0; k < A.n_keys; ++k) {
    const uint64_t v = A.key_width[k] == 8 ? ((const uint64_t *)A.key_values[k])[r] : (uint64_t)(int64_t)((const int32_t *)A.key_values[k])[r];
    h = mix64(h * 0x9E3779B97F4A7C15ull + v);
  }
  return (uint32_t)((h >> 17) % (uint64_t)A.world);
}

// Every row to the block of the rank that owns its key.  Positions inside a block come from one atomic per wave and
// owner (ballot over the lanes with that owner), not one per row: a million rows on eight counters would queue up.
__global__ __launch_bounds__(256) void qe_exchange_scatter_kernel(const OwnerArgs A) {
  const uint32_t lane = threadIdx.x & 63u;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r0 = (int64_t)blockIdx.x * blockDim.x; r0 < A.rows; r0 += stride) {  // whole waves stay in the loop together
    const int64_t r = r0 + threadIdx.x;
    const bool live = r < A.rows;
    const uint32_t owner = live ? owner_of(A, r) : 0xffffffffu;
    int64_t pos = -1;
    for (int w = 0; w < A.world; ++w) {
      const uint64_t mine = __ballot(owner == (uint32_t)w);
      if (mine == 0) continue;
      unsigned int base = 0;
      if (lane == (uint32_t)__builtin_ctzll(mine)) base = atomicAdd(&A.counters[w], (unsigned int)__popcll(mine));
      base = (unsigned int)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(mine));
      if (owner == (uint32_t)w) pos = (int64_t)base + mbcnt64(mine);
    }
    if (!live) continue;
    if (pos >= A.cap) {
      atomicAdd(A.overflow, 1ull);
      continue;
    }
    const int64_t at = (int64_t)owner * A.cap + pos;
    for (int c = 0; c < A.n_cols; ++c) {
      if (A.width[c] == 8) ((uint64_t *)A.dst[c])[at] = ((const uint64_t *)A.src[c])[r];
      else ((uint32_t *)A.dst[c])[at] = ((const uint32_t *)A.src[c])[r];
      if (A.src_valid[c] && ((A.src_valid[c][r >> 3] >> (r & 7)) & 1u)) atomicOr(A.dst_valid[c] + (at >> 5), 1u << (at & 31));
    }
  }
}

}  // namespace

struct Comm {
  Context *ctx = nullptr;
  std::shared_ptr<Context> ctx_ref;
  NcclComm comm = nullptr;
  int rank = 0, world = 1;
  // No peers behind it: every exchange is laid out as rank `rank` of `world` would lay it out (owner hash over `world`
  // owners, this rank's block at rank * padded_rows), the transfers are device copies on the context's stream.
  bool loopback = false;
  BufferPtr counters;  // [MAX_WORLD] unsigned ints for the key-partitioned scatter
};

static Column exchange_column(Context *ctx, int type, int64_t rows, bool with_validity) {
  Column c;
  c.type = type;
  c.space = QE_MEM_DEVICE;
  c.length = rows;
  c.values_bytes = rows * type_width(type);
  c.values = device_buffer(ctx, (size_t)c.values_bytes + 16);
  if (with_validity) {
    c.validity = device_buffer(ctx, bitmap_bytes(rows));
    c.null_count = -1;
  }
  return c;
}

static void require_partial_table(const Batch &b) {
  if (b.cols.empty()) fail(QE_ERR_INVALID_ARGUMENT, "exchange: the partial table has no columns");
  if ((int)b.cols.size() > MAX_XCOLS) fail(QE_ERR_INVALID_ARGUMENT, "exchange: too many columns");
  for (auto &c : b.cols) {
    if (c.space != QE_MEM_DEVICE) fail(QE_ERR_INVALID_ARGUMENT, "exchange: partial tables must be device resident");
    if (!(c.type == QE_TYPE_INT32 || c.type == QE_TYPE_INT64 || c.type == QE_TYPE_FLOAT64))
      fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, std::string("Physical type `") + type_name(c.type) + "` is not supported in an exchanged partial table.");
  }
}

}  // namespace qe

using namespace qe;

struct qe_comm {
  Comm c;
};

// helpers defined in qe_exec.cpp
qe_status qe_guarded(const std::function<void()> &f);
const qe::Batch &qe_batch_ref(const qe_batch *batch);
qe::Context *qe_context_ref(qe_context *ctx);
qe_batch *qe_batch_wrap(const qe::Batch &b);
std::shared_ptr<qe::Context> qe_context_share(qe_context *ctx);

extern "C" {

qe_status qe_comm_unique_id(uint8_t *id) {
  return qe_guarded([&] {
    if (!id) fail(QE_ERR_INVALID_ARGUMENT, "id is null");
    NcclUniqueId u;
    QE_NCCL(rccl().GetUniqueId(&u));
    std::memcpy(id, u.internal, QE_COMM_ID_BYTES);
  });
}

qe_status qe_comm_init(qe_context *ctx, int32_t rank, int32_t world, const uint8_t *id, qe_comm **out) {
  return qe_guarded([&] {
    if (!ctx || !id || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) fail(QE_ERR_INVALID_ARGUMENT, "rank / world out of range");
    Context *c = qe_context_ref(ctx);
    c->bind();
    auto h = std::make_unique<qe_comm>();
    h->c.ctx = c;
    h->c.ctx_ref = qe_context_share(ctx);
    h->c.rank = rank;
    h->c.world = world;
    NcclUniqueId u;
    std::memcpy(u.internal, id, QE_COMM_ID_BYTES);
    Rccl &R = rccl();  // (a library that cannot be loaded fails here, before this rank enters the collective below)
    h->c.counters = device_buffer(c, MAX_WORLD * sizeof(unsigned int));  // and so does an allocation that cannot be had
    QE_NCCL(R.CommInitRank(&h->c.comm, world, u, rank));
    *out = h.release();
  });
}

qe_status qe_comm_init_loopback(qe_context *ctx, int32_t rank, int32_t world, qe_comm **out) {
  return qe_guarded([&] {
    if (!ctx || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) fail(QE_ERR_INVALID_ARGUMENT, "rank / world out of range");
    Context *c = qe_context_ref(ctx);
    c->bind();
    auto h = std::make_unique<qe_comm>();
    h->c.ctx = c;
    h->c.ctx_ref = qe_context_share(ctx);
    h->c.rank = rank;
    h->c.world = world;
    h->c.loopback = true;
    h->c.counters = device_buffer(c, MAX_WORLD * sizeof(unsigned int));
    *out = h.release();
  });
}

qe_status qe_comm_rccl_version(int32_t *version) {
  return qe_guarded([&] {
    if (!version) fail(QE_ERR_INVALID_ARGUMENT, "version is null");
    *version = 0;
    int v = 0;
    QE_NCCL(rccl().GetVersion(&v));
    *version = v;
  });
}

qe_status qe_comm_info(const qe_comm *comm, int32_t *rank, int32_t *world, int32_t *rccl_version) {
  return qe_guarded([&] {
    if (!comm) fail(QE_ERR_INVALID_ARGUMENT, "comm is null");
    if (rank) *rank = comm->c.rank;
    if (world) *world = comm->c.world;
    if (rccl_version) {
      int v = 0;
      if (!comm->c.loopback) QE_NCCL(rccl().GetVersion(&v));  // (a loopback communicator never binds RCCL: 0)
      *rccl_version = v;
    }
  });
}

void qe_comm_free(qe_comm *comm) {
  if (!comm) return;
  if (comm->c.comm) {
    (void)hipSetDevice(comm->c.ctx->device);
    (void)hipStreamSynchronize(comm->c.ctx->stream);
    (void)rccl().CommDestroy(comm->c.comm);
  }
  delete comm;
}

qe_status qe_exchange_all_gather(qe_comm *comm, const qe_batch *partial, int64_t padded_rows, qe_batch **out) {
  return qe_guarded([&] {
    if (!comm || !partial || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    Comm &C = comm->c;
    Context *ctx = C.ctx;
    ctx->bind();
    const Batch &in = qe_batch_ref(partial);
    require_partial_table(in);
    if (padded_rows < in.rows) fail(QE_ERR_INVALID_ARGUMENT, "exchange: the table has " + std::to_string(in.rows) + " rows, more than padded_rows = " + std::to_string(padded_rows));
    bool any_validity = false;
    for (auto &c : in.cols) any_validity = any_validity || c.validity;
    if (any_validity && (padded_rows % 32) != 0) fail(QE_ERR_INVALID_ARGUMENT, "exchange: padded_rows must be a multiple of 32 when a column can be null");
    Batch result;
    result.ctx = ctx;
    result.rows = padded_rows * C.world;
    // Every rank's block lands at rank * padded_rows of the output column.  A table that already has the padded size
    // (the deferred mode's one-row-per-slot block) is sent from where it lies; a shorter one is first put into its place
    // in the output, zero beyond its rows, and the collective runs in place.  Without peers (loopback) the table is
    // always put into its place and the other ranks' blocks are what empty tables would have left there: zeros.
    const bool as_it_is = in.rows == padded_rows && padded_rows > 0 && !C.loopback;
    for (auto &c : in.cols) {
      const size_t w = (size_t)type_width(c.type);
      Column o = exchange_column(ctx, c.type, result.rows, c.validity != nullptr);
      if (C.loopback) {
        if (result.rows) QE_HIP(hipMemsetAsync(o.values->ptr, 0, (size_t)result.rows * w, ctx->stream));
        if (o.validity && result.rows) QE_HIP(hipMemsetAsync(o.validity->ptr, 0, (size_t)(result.rows / 8), ctx->stream));
      }
      if (!as_it_is) {
        char *mine = (char *)o.values->ptr + (size_t)C.rank * padded_rows * w;
        if (in.rows) QE_HIP(hipMemcpyAsync(mine, c.values_ptr(), (size_t)in.rows * w, hipMemcpyDeviceToDevice, ctx->stream));
        if (padded_rows > in.rows) QE_HIP(hipMemsetAsync(mine + (size_t)in.rows * w, 0, (size_t)(padded_rows - in.rows) * w, ctx->stream));
        if (c.validity) {
          char *vb = (char *)o.validity->ptr + (size_t)C.rank * (padded_rows / 8);
          QE_HIP(hipMemsetAsync(vb, 0, (size_t)(padded_rows / 8), ctx->stream));
          if (in.rows) QE_HIP(hipMemcpyAsync(vb, c.validity_ptr(), (size_t)((in.rows + 7) / 8), hipMemcpyDeviceToDevice, ctx->stream));
        }
      }
      result.cols.push_back(o);
    }
    if (C.loopback) {
      *out = qe_batch_wrap(result);
      return;
    }
    Rccl &R = rccl();
    QE_NCCL(R.GroupStart());
    for (size_t ci = 0; ci < result.cols.size(); ++ci) {
      Column &o = result.cols[ci];
      const Column &c = in.cols[ci];
      const size_t block = (size_t)padded_rows * type_width(o.type);
      const void *send = as_it_is ? c.values_ptr() : (const void *)((char *)o.values->ptr + (size_t)C.rank * block);
      if (block) QE_NCCL(R.AllGather(send, o.values->ptr, block, NCCL_UINT8, C.comm, ctx->stream));
      if (o.validity && padded_rows) {
        const void *vsend = as_it_is ? (const void *)c.validity_ptr() : (const void *)((char *)o.validity->ptr + (size_t)C.rank * (padded_rows / 8));
        QE_NCCL(R.AllGather(vsend, o.validity->ptr, (size_t)(padded_rows / 8), NCCL_UINT8, C.comm, ctx->stream));
      }
    }
    QE_NCCL(R.GroupEnd());
    *out = qe_batch_wrap(result);
  });
}

qe_status qe_exchange_by_key(qe_comm *comm, const qe_batch *partial, int32_t n_key_columns, int64_t max_rows_per_rank, qe_batch **out) {
  return qe_guarded([&] {
    if (!comm || !partial || !out) fail(QE_ERR_INVALID_ARGUMENT, "bad arguments");
    *out = nullptr;
    Comm &C = comm->c;
    Context *ctx = C.ctx;
    ctx->bind();
    const Batch &in = qe_batch_ref(partial);
    require_partial_table(in);
    if (n_key_columns < 1 || n_key_columns > 4 || n_key_columns > (int)in.cols.size()) fail(QE_ERR_INVALID_ARGUMENT, "exchange by key: 1 to 4 key columns");
    if (max_rows_per_rank < in.rows) fail(QE_ERR_INVALID_ARGUMENT, "exchange by key: the table has more rows than max_rows_per_rank");
    for (int k = 0; k < n_key_columns; ++k)
      if (in.cols[k].type == QE_TYPE_FLOAT64) fail(QE_ERR_PHYSICAL_TYPE_NOT_SUPPORTED, "Physical type `Float64` is not supported as a group key.");
    // Rows per peer block, the same number on every rank: an even share of the largest table plus a quarter (a hash
    // spreads a million keys over eight owners to within a fraction of a per cent).  A block that overflows all the
    // same is reported by the next qe_context_synchronize: nothing here waits for the device.
    int64_t cap = max_rows_per_rank / C.world + max_rows_per_rank / (4 * C.world) + 1024;
    cap = (cap + 31) & ~(int64_t)31;
    Batch send, result;
    result.ctx = ctx;
    result.rows = cap * C.world;
    OwnerArgs a{};
    a.n_keys = n_key_columns;
    a.world = C.world;
    a.rows = in.rows;
    a.cap = cap;
    a.n_cols = (int)in.cols.size();
    for (int k = 0; k < n_key_columns; ++k) {
      a.key_values[k] = in.cols[k].values_ptr();
      a.key_width[k] = type_width(in.cols[k].type);
    }
    for (size_t i = 0; i < in.cols.size(); ++i) {
      const Column &c = in.cols[i];
      Column s = exchange_column(ctx, c.type, result.rows, c.validity != nullptr);
      QE_HIP(hipMemsetAsync(s.values->ptr, 0, (size_t)s.values_bytes, ctx->stream));  // unfilled rows travel as zeros
      if (s.validity) QE_HIP(hipMemsetAsync(s.validity->ptr, 0, (size_t)(result.rows / 8), ctx->stream));
      send.cols.push_back(s);
      result.cols.push_back(exchange_column(ctx, c.type, result.rows, c.validity != nullptr));
      a.src[i] = c.values_ptr();
      a.dst[i] = s.values->ptr;
      a.width[i] = type_width(c.type);
      a.src_valid[i] = c.validity_ptr();
      a.dst_valid[i] = s.validity ? (uint32_t *)s.validity->ptr : nullptr;
    }
    a.counters = (unsigned int *)C.counters->ptr;
    a.overflow = ctx->deferred_sink();
    QE_HIP(hipMemsetAsync(a.counters, 0, MAX_WORLD * sizeof(unsigned int), ctx->stream));
    if (in.rows > 0) {
      int grid = (int)std::min<int64_t>((in.rows + 255) / 256, (int64_t)ctx->dev.cus * 8);
      ctx->prof_begin("qe_exchange_scatter_kernel");
      hipLaunchKernelGGL(qe_exchange_scatter_kernel, dim3(grid), dim3(256), 0, ctx->stream, a);
      QE_HIP(hipGetLastError());
      ctx->prof_end();
    }
    if (C.loopback) {
      // no peers: block p of the output is what this rank would send to rank p (over a real communicator block p holds
      // what rank p sent here), moved by device copies where the grouped send / receive pairs would move it
      for (size_t i = 0; i < in.cols.size(); ++i) {
        const size_t bytes = (size_t)cap * type_width(in.cols[i].type) * (size_t)C.world;
        QE_HIP(hipMemcpyAsync(result.cols[i].values->ptr, send.cols[i].values->ptr, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        if (send.cols[i].validity)
          QE_HIP(hipMemcpyAsync(result.cols[i].validity->ptr, send.cols[i].validity->ptr, (size_t)(result.rows / 8), hipMemcpyDeviceToDevice, ctx->stream));
      }
      *out = qe_batch_wrap(result);
      return;
    }
    Rccl &R = rccl();
    QE_NCCL(R.GroupStart());
    for (size_t i = 0; i < in.cols.size(); ++i) {
      const size_t block = (size_t)cap * type_width(in.cols[i].type);
      for (int p = 0; p < C.world; ++p) {
        QE_NCCL(R.Send((const char *)send.cols[i].values->ptr + (size_t)p * block, block, NCCL_UINT8, p, C.comm, ctx->stream));
        QE_NCCL(R.Recv((char *)result.cols[i].values->ptr + (size_t)p * block, block, NCCL_UINT8, p, C.comm, ctx->stream));
        if (send.cols[i].validity) {  // cap is a multiple of 32: a peer's bits are whole bytes
          const size_t bits = (size_t)cap / 8;
          QE_NCCL(R.Send((const char *)send.cols[i].validity->ptr + (size_t)p * bits, bits, NCCL_UINT8, p, C.comm, ctx->stream));
          QE_NCCL(R.Recv((char *)result.cols[i].validity->ptr + (size_t)p * bits, bits, NCCL_UINT8, p, C.comm, ctx->stream));
        }
      }
    }
    QE_NCCL(R.GroupEnd());
    // the send blocks go back to the pool here; whoever takes them next is ordered behind the transfers on the stream
    *out = qe_batch_wrap(result);
  });
}

}  // extern "C"
