// Hand-written gfx950 (CDNA4, wave64) kernels of the physical-plan hot path:
//   * qe_project_kernel   — fused expression evaluation (ProjectionIterator::next,
//                           physical_plan/mod.rs:138-156 in the reference)
//   * qe_filter_kernel    — predicate -> stable compaction -> projection
//                           (SelectionIterator::next + filter_chunk, physical_plan/mod.rs:205-229)
//   * qe_aggregate_kernel — predicate/projection fused into a hash aggregate with a per-workgroup
//                           LDS table and a global open-addressing table
//                           (AggregateExec::execute, physical_plan/mod.rs:292-374)
//   * qe_finalize_kernel  — group table -> output batch (physical_plan/mod.rs:375-488)
// All of them are HBM-bandwidth bound byte/integer/f64 streaming work: wide coalesced loads,
// wave64 ballot/popcount ranking, LDS scans and LDS atomics; there is no dense contraction here.
//
// Code size is a first-order cost for the interpreter kernels: the instruction cache holds
// 64 KiB and a wave that jumps around a larger body stalls on instruction fetch.  Hence every
// kernel instantiates the interpreter ONCE, rare operations share one out-of-line-style copy
// that works on temporaries, and per-row work that is more than a few instructions long runs in
// a rolled loop that rotates the four row registers instead of being unrolled four times.
#include <cmath>
#include <cstdlib>

#include "qe_device.h"
#include "qe_jit.h"
#include "qe_kernels.h"

namespace qe {

// --------------------------------------------------------- projection kernel
template <bool FULL>
__global__ __launch_bounds__(VM_BLOCK_THREADS, FULL ? 2 : 3) void qe_project_kernel(const ProjectArgs A) {
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const int64_t n = A.n_rows;
  const int64_t nsteps = (n + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  const ProgRegs R = prog_regs(A.prog, lane);
  for (int64_t ws = (int64_t)blockIdx.x * VM_BLOCK_WAVES + wave; ws < nsteps;
       ws += (int64_t)gridDim.x * VM_BLOCK_WAVES) {
    RowCtx c;
    c.row0 = ws * VM_WAVE_ROWS;
    c.n = n;
    c.lane = lane;
    c.active = in_range_bits(c.row0, lane, n);
    ColCache K;
    cache_load(A.prog, (1u << A.prog.cache_n) - 1u, c, K);
#pragma unroll 1
    for (int e = 0; e < A.n_outs; ++e) {
      Slot S[VM_MAX_DEPTH];
      vm_eval<FULL>(A.prog, R, A.outs[e].pc, c, K, S);
      store_dense(A.outs[e], c, S[0]);
    }
  }
}

// ------------------------------------------------------------ filter kernel
// One ticket = one tile of 4096 rows = 4 steps of 1024.  Sub-steps 0..3 evaluate the predicate
// of each step and keep the 16 selection bits of each lane in a register; then the tile's count
// goes through the look-back; sub-steps 4..7 evaluate every output expression for the selected
// rows only and write it at its final position.  Rows keep their input order (the reference's
// filter_chunk is order preserving, tests/mod.rs:138-188); a row whose predicate is null is
// dropped.  Both phases run through one loop body so that the interpreter exists once.
template <bool FULL>
__global__ __launch_bounds__(VM_BLOCK_THREADS, FULL ? 2 : 3) void qe_filter_kernel(const FilterArgs A) {
  __shared__ unsigned long long s_tile;
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_cnt[FILTER_STEPS * VM_BLOCK_WAVES];
  __shared__ uint8_t s_flags[VM_BLOCK_WAVES][VM_WAVE_ROWS];  // a wave's validity / boolean bits on their way out
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const int64_t n = A.n_rows;
  unsigned long long *status = (unsigned long long *)(A.scratch + 1);
  const ProgRegs R = prog_regs(A.prog, lane);

  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&A.scratch->ticket, 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)s_tile;
    if (tile >= A.n_tiles) break;
    const int64_t tile_row0 = tile * FILTER_TILE_ROWS;

    uint32_t mask = 0;
    uint32_t pref[FILTER_STEPS] = {0, 0, 0, 0};
    unsigned long long base = 0;
#pragma unroll 1
    for (int u = 0; u < 2 * FILTER_STEPS; ++u) {
      const bool phase2 = u >= FILTER_STEPS;
      const int s = u & (FILTER_STEPS - 1);
      if (u == FILTER_STEPS) {
        // all four counts of every wave are in: prefix inside the tile, then across tiles
        __syncthreads();
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < FILTER_STEPS * VM_BLOCK_WAVES; ++i) {
          if ((uint32_t)(i & (VM_BLOCK_WAVES - 1)) == wave) pref[i / VM_BLOCK_WAVES] = run;
          run += s_cnt[i];
        }
        if (wave == 0) {
          unsigned long long excl = lookback_exclusive(status, tile, run, lane, &A.scratch->error);
          if (lane == 0) {
            s_base = excl;
            if (tile == A.n_tiles - 1) A.scratch->total = excl + run;
          }
        }
        __syncthreads();
        base = s_base;
      }
      RowCtx c;
      c.row0 = tile_row0 + s * VM_BLOCK_ROWS + wave * VM_WAVE_ROWS;
      c.n = n;
      c.lane = lane;
      c.active = phase2 ? ((mask >> (4 * s)) & 0xFu) : in_range_bits(c.row0, lane, n);
      const bool any = __ballot(c.active != 0) != 0;
      uint32_t sel = 0;
      if (any) {
        // phase 2 loads for the selected rows only: unselected lanes issue no load
        ColCache K;
        cache_load(A.prog, phase2 ? A.out_cache_mask : A.pred_cache_mask, c, K);
        uint32_t rank[VM_R];
        if (phase2) rank_rows(c.active, rank);
        const unsigned long long obase = base + (s == 0 ? pref[0] : s == 1 ? pref[1] : s == 2 ? pref[2] : pref[3]);
        const int ntask = phase2 ? A.n_outs : 1;
#pragma unroll 1
        for (int e = 0; e < ntask; ++e) {
          Slot S[VM_MAX_DEPTH];
          vm_eval<FULL>(A.prog, R, phase2 ? A.outs[e].pc : A.pred_pc, c, K, S);
          if (phase2)
            store_compact_ranked(A.outs[e], obase, c.active, rank, S[0], s_flags[wave], s_cnt[s * VM_BLOCK_WAVES + wave], lane);
          else
            sel = c.active & S[0].valid & bool_bits(S[0]);
        }
      }
      if (!phase2) {
        mask |= sel << (4 * s);
        uint32_t cnt = popc_sel(sel);
        if (lane == 0) s_cnt[s * VM_BLOCK_WAVES + wave] = cnt;
      }
    }
  }
}

// --------------------------------------------------------- aggregate kernel
// Each workgroup folds its rows into an LDS hash table (keys claimed by 64-bit LDS compare-swap,
// accumulators updated by LDS atomics) and flushes the table into the global open-addressing
// table once, at the end.  Keys that do not fit the LDS table go straight to the global table.
// A key that dominates a wave-step (low cardinality, skew) is reduced across the wave first, so
// that it costs one atomic per wave-step instead of one per row.
template <bool FULL>
__global__ __launch_bounds__(VM_BLOCK_THREADS, FULL ? 2 : 3) void qe_aggregate_kernel(const AggArgs A) {
  extern __shared__ unsigned long long lds_raw[];
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const int64_t n = A.n_rows;
  LdsTable L;
  agg_lds_setup(L, lds_raw, A.lds_cap, A.lds_limit, A.n_accs, agg_tracks_validity(A.accs, A.n_accs));
  agg_lds_clear(L, A.accs, A.n_accs);

  // wave-uniform descriptors, one per lane (see ProgRegs): task list, key fields, accumulators
  const ProgRegs R = prog_regs(A.prog, lane);
  const uint32_t tl = lane < VM_MAX_TASKS ? lane : 0;
  const uint32_t task_pc = A.tasks[tl].pc;
  const uint32_t task_w = (uint32_t)A.tasks[tl].role | ((uint32_t)A.tasks[tl].idx << 8) | ((uint32_t)A.tasks[tl].has_eval << 16);
  const uint32_t key_w = (uint32_t)A.keys[lane & (VM_MAX_KEYS - 1)].shift | ((uint32_t)A.keys[lane & (VM_MAX_KEYS - 1)].bits << 8);
  const uint32_t acc_w = (uint32_t)A.accs[lane & (VM_MAX_ACCS - 1)].op | ((uint32_t)A.accs[lane & (VM_MAX_ACCS - 1)].in_type << 8) |
                         ((uint32_t)A.accs[lane & (VM_MAX_ACCS - 1)].track_nn << 16) | ((uint32_t)A.spill_col[lane & (VM_MAX_ACCS - 1)] << 24);
  SpillState sp = spill_begin(A.spill, wave);

  const int64_t nsteps = (n + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS;
  GiveUpWatch steps_done;
  steps_done.on = A.spill.keys == nullptr;
  for (int64_t ws = (int64_t)blockIdx.x * VM_BLOCK_WAVES + wave; ws < nsteps;
       ws += (int64_t)gridDim.x * VM_BLOCK_WAVES) {
    if (table_given_up(A.table, steps_done)) break;
    RowCtx c;
    c.row0 = ws * VM_WAVE_ROWS;
    c.n = n;
    c.lane = lane;
    c.active = in_range_bits(c.row0, lane, n);
    ColCache K;
    cache_load(A.prog, (1u << A.prog.cache_n) - 1u, c, K);

    AggRows G;
#pragma unroll
    for (int j = 0; j < VM_R; ++j) G.key[j] = 0, G.slot[j] = SLOT_NONE;
    G.rem = 0;
    G.lead_mb = 0;
    G.lead_slot = SLOT_NONE;

#pragma unroll 1
    for (int t = 0; t < A.n_tasks; ++t) {
      const uint32_t tw = (uint32_t)__builtin_amdgcn_readlane((int)task_w, t);
      const uint32_t role = tw & 0xffu;
      const uint32_t idx = (tw >> 8) & 0xffu;
      Slot S[VM_MAX_DEPTH];
      S[0].valid = 0xFu;
#pragma unroll
      for (int j = 0; j < VM_R; ++j) S[0].v[j] = 1;
      if ((tw >> 16) & 1u) vm_eval<FULL>(A.prog, R, (uint32_t)__builtin_amdgcn_readlane((int)task_pc, t), c, K, S);

      if (role == ROLE_PRED) {
        c.active &= S[0].valid & bool_bits(S[0]);
        if (__ballot(c.active != 0) == 0) break;  // nothing of this wave-step survives the filter
      } else if (role == ROLE_KEY) {
        c.active &= S[0].valid;  // a null key drops the row (physical_plan/mod.rs:367)
        const uint32_t kw = (uint32_t)__builtin_amdgcn_readlane((int)key_w, (int)idx);
        const uint32_t kbits = (kw >> 8) & 0xffu;
        const uint64_t fmask = kbits >= 64 ? ~0ull : ((1ull << kbits) - 1ull);
        const uint32_t sh = kw & 0xffu;
#pragma unroll
        for (int j = 0; j < VM_R; ++j) G.key[j] |= (S[0].v[j] & fmask) << sh;
      } else if (role == ROLE_LOOKUP) {
        agg_lookup(L, A.table, c.active, lane, G, sp);
      } else {  // ROLE_ACC
        const uint32_t aw = (uint32_t)__builtin_amdgcn_readlane((int)acc_w, (int)idx);
        if (G.spill_mb != 0 && (aw >> 24) != 0xffu) spill_values(sp, aw >> 24, S[0], G);
        agg_fold(L, A.table, idx, aw & 0xffu, (aw >> 8) & 0xffu, ((aw >> 16) & 0xffu) != 0, S[0], G, lane);
      }
    }
  }
  spill_end(A.spill, sp, wave, lane);
  agg_flush(L, A.table, A.accs, A.n_accs);
}

// ------------------------------------------- high-cardinality second phase
// The spill log is a set of per-wave segments (segment w: rows [w * rows_per_wave, + counts[w])).  It is partitioned
// by the top bits of a hash of the key so that one partition's distinct keys fit an LDS table:
//   count    one pass over the logged keys: rows per (partition, workgroup) cell, for all `bits` at once;
//   scan     exclusive scan of the cells, partition-major: the scan IS the layout of the partitioned rows;
//   scatter  the rows move to their partition - in ONE pass when there are at most 64 partitions, else in TWO
//            (first by the top bits1 bits, then every first-level cell by the remaining bits).  HBM serves
//            scattered accesses at the rate at which it opens rows, a few G per second whatever their size, so a
//            partitioning pass is only as fast as its bursts are long: a workgroup collects the rows of a partition
//            in an LDS stage and writes them out KiB-wise, which 144 KiB of LDS afford for a few dozen partitions
//            at a time, not for five hundred (one pass over 512 partitions: 64-byte bursts, 0.8 TB/s);
//   aggregate one partition at a time into a fresh LDS table, flushed into the global table.
// The count pass serves both scatter levels: workgroup b of the second level takes, for every first-level partition
// q, exactly the rows workgroup b wrote there in the first level (cell (q, b)), so its rows per final partition are
// the counts it made itself in the count pass.
constexpr int PART_THREADS = 1024;
constexpr int PART_MAX_BITS = 11;
constexpr int SCATTER_MAX_STAGES = 512;
constexpr int SCATTER_ROWS = 4;       // rows per thread and round
constexpr int SCATTER_MAX_INPUTS = 8;  // = SPILL_MAX_INPUTS of the host

__global__ __launch_bounds__(PART_THREADS) void qe_spill_count_kernel(AggSpill sp, int n_segments, int bits, int32_t *cells, unsigned long long *total) {
  __shared__ uint32_t s_cell[1 << PART_MAX_BITS];
  const int P = 1 << bits;
  for (int p = threadIdx.x; p < P; p += blockDim.x) s_cell[p] = 0u;
  __syncthreads();
  unsigned long long mine = 0;
  const int seg0 = sp.seg_chunk ? (int)(blockIdx.x * sp.seg_chunk) : (int)blockIdx.x;
  const int seg_end = sp.seg_chunk ? min(n_segments, seg0 + (int)sp.seg_chunk) : n_segments;
  const int seg_step = sp.seg_chunk ? 1 : (int)gridDim.x;
  for (int seg = seg0; seg < seg_end; seg += seg_step) {
    const uint32_t cnt = sp.counts[seg];
    const uint64_t base = (uint64_t)seg * sp.rows_per_wave;
    for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) atomicAdd(&s_cell[partition_of(sp.keys[(base + i) << sp.pairs], bits)], 1u);
    if (threadIdx.x == 0) mine += cnt;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += blockDim.x) cells[p * gridDim.x + blockIdx.x] = (int32_t)s_cell[p];
  if (threadIdx.x == 0 && mine) atomicAdd(total, mine);
}

// First-level cells from the full-depth ones: cells1[q][b] = sum over the `fan` partitions below q of cells[.][b].
__global__ void qe_spill_cells_level1_kernel(const int32_t *cells, int32_t *cells1, int parts1, int fan, int blocks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= parts1 * blocks) return;
  const int q = i / blocks, b = i - q * blocks;
  int32_t sum = 0;
  for (int f = 0; f < fan; ++f) sum += cells[(q * fan + f) * blocks + b];
  cells1[i] = sum;
}

struct ScatterArgs {
  // level 1 source: the log (sp, n_segments).  Level 2 source: first-level cell (q, this workgroup) of src_keys / src_vals
  AggSpill sp;
  int n_segments;
  int level;                 // 1 or 2
  const unsigned long long *src_keys;
  const unsigned long long *src_vals;
  uint64_t src_stride;       // rows between value columns of the source
  const int32_t *src_cells;  // level 2: scanned first-level cells [q * blocks + b]
  int parts1;                // level 2: number of first-level partitions
  int64_t total;             // level 2: rows in all cells (end of the last one)
  int bits;                  // hash bits that name a row's partition at this level's depth
  int stages;                // partitions a workgroup has open at a time (level 1: all of them; level 2: the fan-out)
  int depth;                 // rows per stage
  const int32_t *dst_cells;  // scanned cells of this level's depth [p * blocks + b]: where this workgroup's rows of p go
  unsigned long long *dst_keys;
  unsigned long long *dst_vals;
  uint64_t dst_stride;
  uint32_t n_inputs;
};

// Dynamic LDS: keys[stages * depth], then vals[n_inputs][stages * depth].
__global__ __launch_bounds__(PART_THREADS) void qe_spill_scatter_kernel(const ScatterArgs A) {
  extern __shared__ unsigned long long lds_stage[];
  auto BAR = [&]() { lds_barrier(); };
  __shared__ uint32_t s_cur[SCATTER_MAX_STAGES];    // next output row of each open partition (this workgroup's cell)
  __shared__ uint32_t s_cnt[SCATTER_MAX_STAGES];    // rows offered to the stage since its last flush
  __shared__ uint32_t s_ready[SCATTER_MAX_STAGES];  // stages to write out in this round
  __shared__ uint32_t s_nready;
  const int PS = A.stages, S = A.depth;
  const uint32_t n_in = A.n_inputs;
  unsigned long long *st_keys = lds_stage;
  unsigned long long *st_vals = lds_stage + (size_t)PS * S;
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const uint32_t stage_mask = (uint32_t)PS - 1u;

  // Writes out every stage holding at least `threshold` rows, a wave to a stage: bursts of up to depth x 8 bytes per column.
  auto flush = [&](uint32_t threshold) {
    if (threadIdx.x < (uint32_t)PS && s_cnt[threadIdx.x] >= threshold && s_cnt[threadIdx.x] != 0) s_ready[atomicAdd(&s_nready, 1u)] = threadIdx.x;
    BAR();
    const uint32_t nready = s_nready;
    for (uint32_t q = wave; q < nready; q += n_waves) {
      const uint32_t p = s_ready[q];
      const uint32_t offered = s_cnt[p];
      const uint32_t n = offered < (uint32_t)S ? offered : (uint32_t)S;
      const uint32_t base = s_cur[p];
      for (uint32_t i = lane; i < n; i += 64) A.dst_keys[base + i] = st_keys[(size_t)p * S + i];
      for (uint32_t v = 0; v < n_in; ++v)
        for (uint32_t i = lane; i < n; i += 64) A.dst_vals[(uint64_t)v * A.dst_stride + base + i] = st_vals[((size_t)v * PS + p) * S + i];
      if (lane == 0) {  // the wave runs in lockstep: every lane has read s_cur / s_cnt above
        s_cur[p] = base + n;
        s_cnt[p] = 0;
      }
    }
    BAR();
    if (threadIdx.x == 0) s_nready = 0;
    BAR();
  };

  // The workgroup's work as a sequence of rounds: SCATTER_ROWS rows per thread out of one segment (level 1: a wave's
  // log; level 2: one first-level cell).  Everything here is uniform over the workgroup.
  struct Round {
    int seg;         // level 1: log segment; level 2: first-level partition
    uint64_t first;  // first row of the segment in the source
    uint32_t cnt;    // rows of the segment
    uint32_t i0;     // first row of the round within the segment
    bool valid;
  };
  const unsigned long long *src_keys = A.level == 1 ? A.sp.keys : A.src_keys;
  const unsigned long long *src_vals = A.level == 1 ? A.sp.vals : A.src_vals;
  const uint64_t src_stride = A.level == 1 ? A.sp.capacity : A.src_stride;
  const int64_t n_cells = (int64_t)A.parts1 * gridDim.x;
  // level 1: this workgroup's segments - every gridDim.x-th, or a run of seg_chunk (AggSpill::seg_chunk)
  const int seg_step = A.level == 1 ? (A.sp.seg_chunk ? 1 : (int)gridDim.x) : 1;
  const int seg_end = A.level == 1 ? (A.sp.seg_chunk ? (int)((blockIdx.x + 1) * A.sp.seg_chunk) : A.n_segments) : A.parts1;
  auto open_segment = [&](Round &r) {  // r.seg names a candidate: skips empty segments, invalid when none is left
    for (;;) {
      if (A.level == 1) {
        if (r.seg >= seg_end) break;
        const int seg = r.seg;
        if (seg >= A.n_segments) break;
        r.cnt = A.sp.counts[seg];
        r.first = (uint64_t)seg * A.sp.rows_per_wave;
      } else {
        if (r.seg >= A.parts1) break;
        const int64_t cell = (int64_t)r.seg * gridDim.x + blockIdx.x;
        const int64_t first = A.src_cells[cell];
        const int64_t end = cell + 1 < n_cells ? (int64_t)A.src_cells[cell + 1] : A.total;
        r.first = (uint64_t)first;
        r.cnt = (uint32_t)(end - first);
      }
      if (r.cnt != 0) {
        r.i0 = 0;
        r.valid = true;
        return;
      }
      r.seg += seg_step;
    }
    r.valid = false;
  };
  auto advance = [&](Round &r) {
    r.i0 += blockDim.x * (uint32_t)SCATTER_ROWS;
    if (r.i0 >= r.cnt) {
      r.seg += seg_step;
      open_segment(r);
    }
  };
  // keys and the first value column of a round's rows: loaded a round ahead, so that the trip to memory overlaps the
  // staging and the barriers of the round before
  auto fetch = [&](const Round &r, unsigned long long (&key)[SCATTER_ROWS], unsigned long long (&val0)[SCATTER_ROWS]) {
#pragma unroll
    for (int k = 0; k < SCATTER_ROWS; ++k) {
      const uint32_t i = r.i0 + (uint32_t)k * blockDim.x + threadIdx.x;
      key[k] = 0;
      val0[k] = 0;
      if (r.valid && i < r.cnt) {
        if (A.level == 1 && A.sp.pairs) {  // (key, value) pairs
          const KeyVal kv = load_pair_streaming((const KeyVal *)src_keys + (r.first + i));
          key[k] = kv.key;
          val0[k] = kv.val;
        } else {
          key[k] = __builtin_nontemporal_load(&src_keys[r.first + i]);
          if (n_in > 0) val0[k] = __builtin_nontemporal_load(&src_vals[r.first + i]);
        }
      }
    }
  };

  if (threadIdx.x == 0) s_nready = 0;
  if (threadIdx.x < (uint32_t)PS) {
    s_cur[threadIdx.x] = A.level == 1 ? (uint32_t)A.dst_cells[threadIdx.x * gridDim.x + blockIdx.x] : 0u;
    s_cnt[threadIdx.x] = 0;
  }
  BAR();
  Round cur;
  cur.seg = A.level == 1 ? (A.sp.seg_chunk ? (int)(blockIdx.x * A.sp.seg_chunk) : (int)blockIdx.x) : 0;
  cur.first = 0;
  cur.cnt = 0;
  cur.i0 = 0;
  cur.valid = false;
  open_segment(cur);
  unsigned long long key[SCATTER_ROWS], val0[SCATTER_ROWS], nkey[SCATTER_ROWS], nval0[SCATTER_ROWS];
  fetch(cur, key, val0);
  while (cur.valid) {  // uniform: barriers inside
    Round nxt = cur;
    advance(nxt);
    fetch(nxt, nkey, nval0);
    if (A.level == 2 && cur.i0 == 0) {
      // a new first-level cell: its rows go to the partitions below it
      if (threadIdx.x < (uint32_t)PS) {
        s_cur[threadIdx.x] = (uint32_t)A.dst_cells[((int64_t)cur.seg * PS + threadIdx.x) * gridDim.x + blockIdx.x];
        s_cnt[threadIdx.x] = 0;
      }
      BAR();
    }
    bool have[SCATTER_ROWS];
    uint32_t slot[SCATTER_ROWS], part[SCATTER_ROWS];
#pragma unroll
    for (int k = 0; k < SCATTER_ROWS; ++k) {
      have[k] = cur.i0 + (uint32_t)k * blockDim.x + threadIdx.x < cur.cnt;
      part[k] = partition_of(key[k], A.bits) & stage_mask;
      slot[k] = 0;
      if (have[k]) {
        slot[k] = atomicAdd(&s_cnt[part[k]], 1u);
        if (slot[k] < (uint32_t)S) {
          st_keys[(size_t)part[k] * S + slot[k]] = key[k];
          if (n_in > 0) st_vals[(size_t)part[k] * S + slot[k]] = val0[k];
        } else {
          slot[k] = 0x80000000u | atomicAdd(&s_cur[part[k]], 1u);  // stage full: straight to memory (rare)
          A.dst_keys[slot[k] & 0x7fffffffu] = key[k];
          if (n_in > 0) A.dst_vals[slot[k] & 0x7fffffffu] = val0[k];
        }
      }
    }
    for (uint32_t v = 1; v < n_in; ++v) {
      unsigned long long val[SCATTER_ROWS];
#pragma unroll
      for (int k = 0; k < SCATTER_ROWS; ++k) {
        const uint32_t i = cur.i0 + (uint32_t)k * blockDim.x + threadIdx.x;
        val[k] = have[k] ? src_vals[(uint64_t)v * src_stride + cur.first + i] : 0ull;
      }
#pragma unroll
      for (int k = 0; k < SCATTER_ROWS; ++k) {
        if (have[k]) {
          if (slot[k] & 0x80000000u) A.dst_vals[(uint64_t)v * A.dst_stride + (slot[k] & 0x7fffffffu)] = val[k];
          else st_vals[((size_t)v * PS + part[k]) * S + slot[k]] = val[k];
        }
      }
    }
    BAR();
    // half-full stages go out (a round brings a stage a fraction of its depth); everything at the end of a level-2
    // cell, whose successor's rows go to other partitions
    const bool cell_ends = A.level == 2 && (!nxt.valid || nxt.seg != cur.seg);
    flush(cell_ends ? 1u : (uint32_t)S / 2);
#pragma unroll
    for (int k = 0; k < SCATTER_ROWS; ++k) {
      key[k] = nkey[k];
      val0[k] = nval0[k];
    }
    cur = nxt;
  }
  flush(1);
}

// ---- the one-pass scatter: every logged row to its partition in a single pass over the log
// What a partitioning pass can afford is set by how its writes reach memory: a burst that covers whole 128-byte lines
// is taken at close to the streaming rate however short it is (128-byte bursts of 16-byte stores at pseudo-random
// places: 4.7 TB/s; tools/experiments/burst_probe.hip), one that starts or ends inside a line costs a read-modify-write
// of that line (64-byte bursts shifted by 8 bytes: 0.8 TB/s).  So the rows are written as (key, value) pairs of 16
// bytes, and a workgroup collects each partition's rows in a ring of 16 rows (two 128-byte lines) in LDS and writes a
// line when it is complete: 512 partitions x 256 bytes = 128 KiB of LDS, one pass where the staging of whole KiB
// bursts needed two.  A row's place in its partition is fixed when it arrives (a counter per partition), so a row that
// finds its ring full is written straight to its place instead (a partial line, rare: rows arrive at half the rings'
// capacity per round) and the line it belongs to is completed around it.  Only a cell's first and last line are partial.
constexpr int ONEPASS_RING = 16;      // rows a partition's ring holds (two lines of eight)
constexpr int ONEPASS_RING_STRIDE = 17;  // rows from one ring to the next
constexpr int ONEPASS_ROWS = 2;       // rows per thread and round
constexpr int ONEPASS_MAX_PARTS = 512;

struct OnePassArgs {
  AggSpill sp;               // the log: keys, one value column (or none), per-segment counts
  int n_segments;
  int bits;                  // partitions = 1 << bits
  const int32_t *dst_cells;  // scanned cells [p * cells_stride + b]: first row of this workgroup's rows of partition p
  KeyVal *dst;               // the partitioned rows
  uint32_t n_inputs;         // 0 or 1
  uint32_t cells_stride;     // columns in a row of the cell table (the logs of several first-pass launches partitioned together: a block of columns each)
};

__global__ __launch_bounds__(PART_THREADS) void qe_spill_scatter_onepass_kernel(const OnePassArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds_stage[];
  const int P = 1 << A.bits;
  // (a ring's stride is 17 rows: with 16 - 256 bytes - every partition's slot i would sit in the same LDS banks)
  KeyVal *ring = (KeyVal *)lds_stage;                                                        // [P][ONEPASS_RING_STRIDE]
  unsigned long long *s_state = (unsigned long long *)(ring + (size_t)P * ONEPASS_RING_STRIDE);  // [P] low word: rows arrived, counted from the line the cell starts in; high word: first row not yet written (a multiple of 8)
  unsigned long long *s_dst0 = s_state + P;                                                  // [P] row index in dst of the ring's row 0 (cell start rounded down to a line)
  uint32_t *s_valid = (uint32_t *)(s_dst0 + P);                                              // [P] bit i: ring slot i holds a row
  uint32_t *s_ready = s_valid + P;                                                           // [2 P] lines to write this round: partition | line number << 9
  uint32_t *s_nready = s_ready + 2 * P;                                                      // [2] their number, rounds taking turns
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  auto BAR = [&]() { lds_barrier(); };

  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const uint64_t base = (uint64_t)(uint32_t)A.dst_cells[(size_t)p * A.cells_stride + blockIdx.x];
    s_dst0[p] = base & ~7ull;
    s_state[p] = base & 7ull;
    s_valid[p] = 0;
  }
  if (threadIdx.x < 2) s_nready[threadIdx.x] = 0;
  BAR();

  // Writes the lines that are complete (every row of them has arrived); `all`: whatever has arrived, at the end.
  // Three barriers a round: after the rows' arrival, after the list of lines, after the writing.
  uint32_t turn = 0;
  auto flush = [&](bool all) {
    uint32_t *nready_at = s_nready + (turn & 1u);
    for (int t = threadIdx.x; t < 2 * P; t += blockDim.x) {  // (2 P is a multiple of 64: a wave's lanes stay together)
      const uint32_t p = (uint32_t)t >> 1, j = (uint32_t)t & 1u;
      const unsigned long long st = s_state[p];
      const uint32_t uw = (uint32_t)st, line0 = (uint32_t)(st >> 32) + 8u * j;  // rows [line0, line0 + 8)
      const bool ready = uw >= line0 + 8u || (all && uw > line0);
      const uint64_t m = __ballot(ready);
      if (m != 0) {  // (one atomic a wave)
        uint32_t at = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) at = atomicAdd(nready_at, (uint32_t)__popcll(m));
        at = (uint32_t)__builtin_amdgcn_readlane((int)at, __builtin_ctzll(m));
        if (ready) s_ready[at + mbcnt64(m)] = p | ((line0 >> 3) << 9);
      }
    }
    BAR();
    const uint32_t nready = *nready_at;
    if (threadIdx.x == 0) s_nready[(turn + 1u) & 1u] = 0;  // (last read before the barrier that ended the previous round)
    for (uint32_t q = wave * 8u; q < nready; q += n_waves * 8u) {
      const uint32_t e = q + (lane >> 3);
      if (e < nready) {
        const uint32_t entry = s_ready[e], p = entry & 511u, line = entry >> 9, r = lane & 7u;
        const uint32_t slot = (line & 1u) * 8u + r;
        if ((s_valid[p] >> slot) & 1u) A.dst[s_dst0[p] + line * 8u + r] = ring[(size_t)p * ONEPASS_RING_STRIDE + slot];
        // the written line leaves the window (a partition's two lines are listed in order, and both or the first)
        if (r == 0) {
          atomicAnd(&s_valid[p], ~(0xffu << (8u * (line & 1u))));
          atomicAdd(&s_state[p], 8ull << 32);
        }
      }
    }
    BAR();
    ++turn;
  };

  struct Round {
    int seg;
    uint64_t first;
    uint32_t cnt, i0;
    bool valid;
  };
  const int seg_step = A.sp.seg_chunk ? 1 : (int)gridDim.x;
  const int seg_end = A.sp.seg_chunk ? (int)((blockIdx.x + 1) * A.sp.seg_chunk) : A.n_segments;
  auto open_segment = [&](Round &r) {
    for (; r.seg < seg_end && r.seg < A.n_segments; r.seg += seg_step) {
      r.cnt = A.sp.counts[r.seg];
      r.first = (uint64_t)r.seg * A.sp.rows_per_wave;
      if (r.cnt != 0) {
        r.i0 = 0;
        r.valid = true;
        return;
      }
    }
    r.valid = false;
  };
  auto advance = [&](Round &r) {
    r.i0 += blockDim.x * (uint32_t)ONEPASS_ROWS;
    if (r.i0 >= r.cnt) {
      r.seg += seg_step;
      open_segment(r);
    }
  };
  auto fetch = [&](const Round &r, KeyVal (&row)[ONEPASS_ROWS]) {
#pragma unroll
    for (int k = 0; k < ONEPASS_ROWS; ++k) {
      const uint32_t i = r.i0 + (uint32_t)k * blockDim.x + threadIdx.x;
      row[k].key = 0;
      row[k].val = 0;
      if (r.valid && i < r.cnt) {
        if (A.sp.pairs) {
          row[k] = load_pair_streaming((const KeyVal *)A.sp.keys + (r.first + i));
        } else {
          row[k].key = __builtin_nontemporal_load(&A.sp.keys[r.first + i]);
          if (A.n_inputs > 0) row[k].val = __builtin_nontemporal_load(&A.sp.vals[r.first + i]);
        }
      }
    }
  };
  Round cur;
  cur.seg = A.sp.seg_chunk ? (int)(blockIdx.x * A.sp.seg_chunk) : (int)blockIdx.x;
  cur.first = 0;
  cur.cnt = 0;
  cur.i0 = 0;
  cur.valid = false;
  open_segment(cur);
  KeyVal row[ONEPASS_ROWS], nrow[ONEPASS_ROWS];
  fetch(cur, row);
  while (cur.valid) {  // uniform: barriers inside
    Round nxt = cur;
    advance(nxt);
    fetch(nxt, nrow);  // a round ahead: the trip to memory overlaps this round's staging and barriers
    // the rows take their places: one LDS operation numbers a row and tells it where the ring's window starts
    uint32_t part[ONEPASS_ROWS];
    unsigned long long st[ONEPASS_ROWS];
    bool have[ONEPASS_ROWS];
#pragma unroll
    for (int k = 0; k < ONEPASS_ROWS; ++k) {
      have[k] = cur.i0 + (uint32_t)k * blockDim.x + threadIdx.x < cur.cnt;
      part[k] = partition_of(row[k].key, A.bits);
      st[k] = 0;
      if (have[k]) st[k] = atomicAdd(&s_state[part[k]], 1ull);
    }
#pragma unroll
    for (int k = 0; k < ONEPASS_ROWS; ++k) {
      if (have[k]) {
        const uint32_t u = (uint32_t)st[k], ur = (uint32_t)(st[k] >> 32);
        if (u < ur + (uint32_t)ONEPASS_RING) {
          ring[(size_t)part[k] * ONEPASS_RING_STRIDE + (u & (ONEPASS_RING - 1))] = row[k];
          atomicOr(&s_valid[part[k]], 1u << (u & (ONEPASS_RING - 1)));
        } else {
          A.dst[s_dst0[part[k]] + u] = row[k];  // the ring is full: straight to its place
        }
      }
    }
    BAR();
    flush(false);
#pragma unroll
    for (int k = 0; k < ONEPASS_ROWS; ++k) row[k] = nrow[k];
    cur = nxt;
  }
  flush(true);  // (whatever lies beyond a ring's two lines was written straight to its place)
}

// One ticket = one partition: all its rows folded into a fresh LDS table, the table flushed into
// the global one.  A partition holds (groups / partitions) distinct keys, chosen by the host to fit.
// A partition's rows are its stretch of the partitioned copy (A.keys) and / or the stretches that first passes which
// partition directly wrote for it, one per source and first-pass workgroup (A.direct).  The waves take them as units
// from a counter: 256-row steps of the copy (pieces of a 64th of it when there are stretches too), whole stretches.
__global__ __launch_bounds__(PART_THREADS) void qe_aggregate_partitioned_kernel(const PartArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds_raw[];  // (the key slots are read in pairs)
  __shared__ unsigned long long s_ticket;
  __shared__ uint32_t s_unit;  // the current partition's next unit that no wave has taken yet
  __shared__ int32_t s_direct_count[PART_MAX_DIRECT_UNITS];  // rows in each of the current partition's stretches
  const uint32_t lane = lane_id();
  LdsTable L;
  agg_lds_setup(L, lds_raw, A.lds_cap, A.lds_limit, A.n_accs, false);
  L.home_mask = (A.lds_cap - 1) & ~1u;  // whole steps probe pairs of slots, the rest one slot at a time: one order for both
  SpillState none;
  none.keys = nullptr;
  none.vals = nullptr;
  none.capacity = 0;
  none.base = 0;
  none.cursor = 0;
  none.pairs = 0;
  const uint32_t direct_units = (uint32_t)A.n_direct * (uint32_t)A.direct_blocks;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) {
      s_ticket = atomicAdd(A.ticket, 1ull);
      s_unit = 0;
      // a global table that has been given up (the host starts over with a larger one): no further partition
      if (__hip_atomic_load(&A.table.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) s_ticket = (unsigned long long)A.n_parts << A.sub_bits;
    }
    __syncthreads();
    // (a partition's passes hold consecutive tickets: its rows are still in the caches when the next pass reads them)
    const int64_t p = (int64_t)(s_ticket >> A.sub_bits);
    const uint32_t sub = (uint32_t)s_ticket & ((1u << A.sub_bits) - 1u);
    if (p < A.n_parts)
      for (uint32_t t = threadIdx.x; t < direct_units; t += blockDim.x) {
        const PartDirect &d = A.direct[t / (uint32_t)A.direct_blocks];
        s_direct_count[t] = d.counts[(size_t)p * d.counts_stride + t % (uint32_t)A.direct_blocks];
      }
    agg_lds_clear(L, A.accs, A.n_accs);  // ends with a barrier
    if (p >= A.n_parts) break;
    GiveUpWatch watch;
    const int64_t start = A.keys ? A.offsets[p * A.stride] : 0;
    const int64_t end = A.keys ? (p + 1 < A.n_parts ? (int64_t)A.offsets[(p + 1) * A.stride] : A.total) : 0;
    // units of the copy: steps of 256 rows; with stretches to walk as well, 64 pieces (whole steps each)
    const int64_t piece = direct_units == 0 ? (int64_t)VM_WAVE_ROWS : ((((end - start) + 63) / 64 + VM_WAVE_ROWS - 1) / VM_WAVE_ROWS) * VM_WAVE_ROWS;
    const uint32_t copy_units = piece > 0 ? (uint32_t)((end - start + piece - 1) / piece) : 0u;
    const uint32_t n_units = copy_units + direct_units;
    // The keys and the first value column of a wave's next 256 rows are loaded while it works on the current ones:
    // the work is LDS round trips, one after the other, and a memory round trip on top of every step's would double it.
    // Units are taken from a counter, not dealt out in turn: the sixteen waves do not run at one pace (the oldest wave
    // of a SIMD is served first; dealt equal shares, wave 0 stood a fifth of a partition's time at the barrier).
    struct Cursor {
      const KeyVal *rows;  // the unit's pairs; nullptr: rows of the partitioned copy (A.keys / A.vals)
      int64_t pos, end;    // the next step's first row, the unit's end
    };
    auto next_unit = [&](Cursor &c) -> bool {  // (wave-uniform)
      for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&s_unit, 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= n_units) return false;
        if (t < copy_units) {
          c.rows = A.interleaved ? (const KeyVal *)A.keys : nullptr;
          c.pos = start + (int64_t)t * piece;
          c.end = c.pos + piece < end ? c.pos + piece : end;
        } else {
          const uint32_t du = t - copy_units, src = du / (uint32_t)A.direct_blocks, b = du % (uint32_t)A.direct_blocks;
          const PartDirect &d = A.direct[src];
          c.rows = (const KeyVal *)d.rows + ((uint64_t)p * (uint32_t)A.direct_blocks + b) * d.cap;
          c.pos = 0;
          c.end = s_direct_count[du];
        }
        if (c.pos < c.end) return true;
      }
    };
    unsigned long long nkey[VM_R], nval[VM_R];
    auto fetch = [&](const Cursor &c) {
#pragma unroll
      for (int j = 0; j < VM_R; ++j) {
        const int64_t r = c.pos + j * 64 + lane;
        nkey[j] = 0;
        nval[j] = 0;
        if (r < c.end) {
          if (c.rows) {
            const KeyVal kv = load_pair_streaming(c.rows + r);
            nkey[j] = kv.key;
            nval[j] = kv.val;
          } else {
            nkey[j] = __builtin_nontemporal_load(&A.keys[r]);
            if (A.n_inputs > 0) nval[j] = __builtin_nontemporal_load(&A.vals[r]);
          }
        }
      }
    };
    Cursor nxt;
    nxt.rows = nullptr;
    nxt.pos = nxt.end = 0;
    bool have = next_unit(nxt);
    if (have) fetch(nxt);
#ifdef QE_ABLATE  // (bit 4: a wave's clock around the parts of a step, printed by one wave at the end)
    unsigned long long t_wait = 0, t_look = 0, t_fold = 0, t_steps = 0, t_rounds = 0;
#endif
    while (have) {
      if (table_given_up(A.table, watch)) break;
#ifdef QE_ABLATE
      unsigned long long c0 = 0;
      if (A.ablate & 16) {
        c0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_wait += __builtin_readcyclecounter() - c0;
      }
#endif
      AggRows G;
      Slot in;
      in.valid = 0xFu;
      uint32_t active = 0;
      const int64_t r0 = nxt.pos;  // (rows of the copy: where the further value columns of this step are read)
#pragma unroll
      for (int j = 0; j < VM_R; ++j) {
        G.key[j] = nkey[j];
        in.v[j] = nval[j];
        G.slot[j] = SLOT_NONE;
        if (nxt.pos + j * 64 + lane < nxt.end) active |= 1u << j;
      }
      if (A.sub_bits) {  // (uniform) the rows of this pass only
#pragma unroll
        for (int j = 0; j < VM_R; ++j)
          if (((hash32(G.key[j]) >> (32 - A.part_bits - A.sub_bits)) & ((1u << A.sub_bits) - 1u)) != sub) active &= ~(1u << j);
      }
      nxt.pos += VM_WAVE_ROWS;
      if (nxt.pos >= nxt.end) have = next_unit(nxt);
      if (have) fetch(nxt);
      // The common step: every row finds (or claims) a slot in the LDS table - by construction a partition's keys fit
      // it.  No per-row bookkeeping of who goes where: four lookups, then one LDS operation per row and accumulator.
      // Anything else (a key the table has no room for) takes the general path below, which looks the rows up again
      // (a lookup changes nothing but the table's key slots).
      {
        int32_t slot[VM_R];
        uint32_t hit;
        if (A.ablate & 4) {
#pragma unroll
          for (int j = 0; j < VM_R; ++j) slot[j] = (int32_t)(hash32(G.key[j]) & (L.cap - 1));
          hit = active;
        } else {
          // a pair of slots a probe (one 16-byte LDS read): half the dependent round trips of the longest probe
          // sequence among the wave's 256 rows, which is what a step waits for
          uint32_t slot8[VM_R];
#ifdef QE_ABLATE
          if (A.ablate & 16) c0 = __builtin_readcyclecounter();
#endif
          hit = lds_pair_find_or_claim_rows(L, G.key, active, LDS_MAX_PROBE / 2, slot8);
#ifdef QE_ABLATE
          if (A.ablate & 16) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t_look += __builtin_readcyclecounter() - c0;
            ++t_steps;
          }
#endif
#pragma unroll
          for (int j = 0; j < VM_R; ++j) slot[j] = (int32_t)(slot8[j] >> 3);
        }
        if (__ballot(hit != active) == 0) {
          if (A.ablate & 2) continue;
#ifdef QE_ABLATE
          if (A.ablate & 16) c0 = __builtin_readcyclecounter();
#endif
#pragma unroll
          for (int j = 0; j < VM_R; ++j)
            if ((active >> j) & 1u) atomicAdd(&L.rowcnt[slot[j]], 1u);
#pragma unroll 1
          for (int v = 0; v < A.n_inputs; ++v) {
            if (v > 0) {  // (logs with several value columns are never pairs: rows of the copy)
#pragma unroll
              for (int j = 0; j < VM_R; ++j) in.v[j] = ((active >> j) & 1u) ? A.vals[(uint64_t)v * A.capacity + r0 + j * 64 + lane] : 0ull;
            }
            const uint32_t use = spill_valid_rows(A.null_cols, A.null_f64, (uint32_t)v, in.v, active);  // (a logged null is a row, not a value)
#pragma unroll 1
            for (int a = 0; a < A.n_accs; ++a) {
              if (A.acc_col[a] != v) continue;
              const uint32_t op = A.accs[a].op, in_type = A.accs[a].in_type;
              unsigned long long *acc = L.acc + (size_t)a * L.stride;
              if (op == ACC_MIN || op == ACC_MAX) {
                // (the atomic straight away: a read first, to skip atomics that cannot win, is a dependent LDS round trip per row)
#pragma unroll
                for (int j = 0; j < VM_R; ++j) {
                  const bool nan = in_type == VT_F64 && u2d(in.v[j]) != u2d(in.v[j]);  // NaN is ignored by min / max
                  if (!nan && ((use >> j) & 1u)) {
                    if (op == ACC_MIN) atomicMin(acc + slot[j], (unsigned long long)ord_image(in.v[j], in_type));
                    else atomicMax(acc + slot[j], (unsigned long long)ord_image(in.v[j], in_type));
                  }
                }
              } else {
#pragma unroll
                for (int j = 0; j < VM_R; ++j)
                  if ((use >> j) & 1u) acc_apply(acc + slot[j], op, op == ACC_COUNT ? 1ull : in.v[j]);
              }
            }
          }
#ifdef QE_ABLATE
          if (A.ablate & 16) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t_fold += __builtin_readcyclecounter() - c0;
          }
#endif
          continue;
        }
      }
      G.rem = 0;
      G.lead_mb = 0;
      G.lead_slot = SLOT_NONE;
      agg_lookup<false>(L, A.table, active, lane, G, none);
#pragma unroll 1
      for (int v = 0; v < A.n_inputs; ++v) {
        if (v > 0) {
#pragma unroll
          for (int j = 0; j < VM_R; ++j) in.v[j] = ((active >> j) & 1u) ? A.vals[(uint64_t)v * A.capacity + r0 + j * 64 + lane] : 0ull;
        }
        in.valid = spill_valid_rows(A.null_cols, A.null_f64, (uint32_t)v, in.v, 0xFu);
#pragma unroll 1
        for (int a = 0; a < A.n_accs; ++a)
          if (A.acc_col[a] == v) agg_fold(L, A.table, a, A.accs[a].op, A.accs[a].in_type, false, in, G, lane);
      }
    }
#ifdef QE_ABLATE
    if ((A.ablate & 16) && p == 7 && threadIdx.x == 128)
      printf("partition 7 wave 2: %llu steps, cycles a step: wait for rows %llu, lookup %llu, fold %llu\n", t_steps, t_wait / (t_steps + 1), t_look / (t_steps + 1),
             t_fold / (t_steps + 1));
    (void)t_rounds;
#endif
    if (!(A.ablate & 1)) agg_flush<true>(L, A.table, A.accs, A.n_accs);  // starts with a barrier
  }
}

// ------------------------------------------------------ table init/finalize
__global__ void qe_table_init_kernel(AggTable T, uint64_t packed_lo, uint64_t packed_hi, int n_accs, unsigned long long *zero, uint64_t zero_words) {
  // (an output block to clear rides along: one launch instead of two in front of a small aggregate)
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < zero_words; i += (uint64_t)gridDim.x * blockDim.x) zero[i] = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < T.stride;
       i += (uint64_t)gridDim.x * blockDim.x) {
    T.keys[i] = AGG_EMPTY_KEY;
    T.rowcnt[i] = 0;
    for (int a = 0; a < n_accs; ++a) {
      uint64_t packed = a < 8 ? packed_lo : packed_hi;
      T.acc[(uint64_t)a * T.stride + i] = acc_identity((uint32_t)(packed >> (8 * (a & 7))) & 0xffu);
      T.nn[(uint64_t)a * T.stride + i] = 0;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) T.ctrl[threadIdx.x] = 0;
}

// Order-independent sums at output time (FIN_FIX_SUM / FIN_FIX_AVG).
__device__ __forceinline__ double fin_fix_sum(const AggTable &T, const FinOut &f, uint64_t i) {
  const unsigned long long *acc = T.acc + (uint64_t)f.a * T.stride + i;
  return fix_combine(acc[0], acc[T.stride], acc[2 * T.stride], acc[3 * T.stride], (int)f.scale);
}

// The limbs of the sums in the table moved `delta` bits up: a later batch holds larger values than the scale so far allows.
__global__ void qe_fix_rescale_kernel(AggTable T, int acc, int delta) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < T.stride; i += (uint64_t)gridDim.x * blockDim.x) {
    unsigned long long *p = T.acc + (uint64_t)acc * T.stride + i;
    uint64_t l0 = p[0], l1 = p[T.stride], l2 = p[2 * T.stride];
    if ((l0 | l1 | l2) == 0) continue;
    fix_rescale(l0, l1, l2, delta);
    p[0] = l0;
    p[T.stride] = l1;
    p[2 * T.stride] = l2;
  }
}

hipError_t launch_fix_rescale(const AggTable &table, int acc, int delta, hipStream_t stream) {
  const uint64_t blocks = std::min<uint64_t>((table.stride + 255) / 256, 4096);
  qe_fix_rescale_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(table, acc, delta);
  return hipGetLastError();
}

// Validity bits of a wave's output rows.  The occupied lanes (`occ`) write consecutive rows from `first_pos` on, lane by
// lane: their bits are put together in the wave (`valid`: this lane's row is not null) and leave as at most three
// word-wide atomics, not one per row - a million groups with four nullable outputs are otherwise four million atomics.
// Called by the whole wave.
__device__ __forceinline__ void fin_store_validity(uint32_t *validity, uint64_t occ, bool valid, unsigned long long first_pos) {
  const uint64_t ok = __ballot(valid) & occ;
  if (ok == 0) return;
  const uint32_t n = (uint32_t)__popcll(occ), lane = lane_id();
  uint64_t packed;  // bit r: the r-th occupied lane's row is valid
  if (ok == occ) {
    packed = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  } else {
    packed = ((ok >> lane) & 1ull) ? 1ull << mbcnt64(occ) : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) packed |= __shfl_xor(packed, o);
  }
  const uint32_t sh = (uint32_t)(first_pos & 31u);
  if (lane < 3) {  // words 0..2 of the 96-bit value packed << sh
    uint32_t w;
    if (lane == 0) w = (uint32_t)(packed << sh);
    else if (lane == 1) w = (uint32_t)(sh ? packed >> (32 - sh) : packed >> 32);
    else w = sh ? (uint32_t)(packed >> (64 - sh)) : 0u;
    if (w != 0) atomicOr(validity + (first_pos >> 5) + lane, w);
  }
}

// N slots a lane (slot i[j], key k[j] with the reserved slot already told apart) to output rows pos[j], every output
// column.  Called by the whole wave; occ[j] = its lanes that hold a group in item j, whose rows are consecutive from
// first_pos[j] on.  Per output column all items' loads come first, then all stores.
template <int N>
__device__ __forceinline__ void fin_emit(const FinalizeArgs &A, const uint64_t (&i)[N], const unsigned long long (&k)[N], const bool (&occupied)[N],
                                         const uint64_t (&occ)[N], const unsigned long long (&pos)[N], const unsigned long long (&first_pos)[N]) {
  const AggTable &T = A.table;
  for (int o = 0; o < A.n_outs; ++o) {
    const FinOut &f = A.outs[o];
    uint64_t val[N];
    bool valid[N];
    // What the output reads of a group's row - a count, an accumulator - for all N items before anything is computed
    // or stored, with no branch between the loads (which arrays they come from is the same for every lane): the N
    // items' trips to memory overlap.  One item after the other, each behind the switch below, an output column was
    // N dependent trips, and a tile of a million-group table spent 80 microseconds on twenty of them.
    const bool need_cnt = f.null_mode == NULL_IF_COUNT_ZERO || f.kind == FIN_COUNT || f.kind == FIN_AVG || f.kind == FIN_FIX_AVG;
    const bool need_acc = f.null_mode == NULL_IF_ACC_ALL_ONES || f.null_mode == NULL_IF_ACC_ZERO || f.kind == FIN_ACC_I64 || f.kind == FIN_ACC_F64 ||
                          f.kind == FIN_ACC_ORD || f.kind == FIN_AVG;
    const unsigned long long *cnt_src = f.cnt_src == CNT_ROWS  ? T.rowcnt
                                        : f.cnt_src == CNT_NN ? T.nn + (uint64_t)f.b * T.stride
                                                              : T.acc + (uint64_t)f.b * T.stride;
    const unsigned long long *acc_src = T.acc + (uint64_t)f.a * T.stride;
    unsigned long long cnt[N], accv[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      cnt[j] = (need_cnt && occupied[j]) ? cnt_src[i[j]] : 1ull;
      accv[j] = (need_acc && occupied[j]) ? acc_src[i[j]] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      val[j] = 0;
      valid[j] = occupied[j];
      if (!occupied[j]) continue;
      const uint64_t at = i[j];
      if (f.null_mode == NULL_IF_COUNT_ZERO && cnt[j] == 0) valid[j] = false;
      if (f.null_mode == NULL_IF_ACC_ALL_ONES && accv[j] == ~0ull) valid[j] = false;
      if (f.null_mode == NULL_IF_ACC_ZERO && accv[j] == 0ull) valid[j] = false;
      switch (f.kind) {
        case FIN_KEY: {
          uint64_t fm = f.bits >= 64 ? ~0ull : ((1ull << f.bits) - 1ull);
          val[j] = (k[j] >> f.shift) & fm;
          if (f.type == VT_I32) val[j] = (uint64_t)(int64_t)(int32_t)(uint32_t)val[j];
          break;
        }
        case FIN_ACC_I64:
        case FIN_ACC_F64:
          val[j] = accv[j];
          break;
        case FIN_ACC_ORD:
          val[j] = valid[j] ? ord_inverse(accv[j], f.type) : 0;
          break;
        case FIN_COUNT:
          val[j] = cnt[j];
          break;
        case FIN_AVG:
          val[j] = valid[j] ? d2u(u2d(accv[j]) / (double)cnt[j]) : 0;
          break;
        case FIN_FIX_SUM:
          val[j] = d2u(fin_fix_sum(T, f, at));
          break;
        case FIN_FIX_AVG:
          val[j] = valid[j] ? d2u(fin_fix_sum(T, f, at) / (double)cnt[j]) : 0;
          break;
        default:
          break;
      }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (occupied[j]) {
        if (f.type == VT_I32)
          ((uint32_t *)f.values)[pos[j]] = (uint32_t)val[j];
        else if (f.type == VT_BOOL) {
          if (val[j] & 1u) atomicOr((uint32_t *)f.values + (pos[j] >> 5), 1u << (pos[j] & 31u));
        } else
          ((uint64_t *)f.values)[pos[j]] = val[j];
      }
      if (f.validity && occ[j] != 0) fin_store_validity(f.validity, occ[j], valid[j], first_pos[j]);
    }
  }
}

// Small tables: one atomic per wave is few enough, and no barriers.
__global__ __launch_bounds__(256) void qe_finalize_small_kernel(const FinalizeArgs A) {
  const AggTable &T = A.table;
  if (A.overflow_sink && blockIdx.x == 0 && threadIdx.x == 0 && T.ctrl[1]) atomicAdd(A.overflow_sink, 1ull);
  const uint64_t n_slots = T.cap + 1;
  // (whole waves go round: the output code talks across lanes)
  for (uint64_t i0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) & ~63ull; i0 < n_slots; i0 += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = i0 + lane_id();
    unsigned long long k = i < n_slots ? T.keys[i] : AGG_EMPTY_KEY;
    // output position: one atomic per wave for all its occupied slots, not one per slot
    const bool occupied = k != AGG_EMPTY_KEY;
    const uint64_t occ = __ballot(occupied);
    if (occ == 0) continue;
    unsigned long long wave_base = 0;
    if (lane_id() == (uint32_t)__builtin_ctzll(occ)) wave_base = atomicAdd(A.out_count, (unsigned long long)__popcll(occ));
    wave_base = readlane64(wave_base, __builtin_ctzll(occ));
    if (i == T.cap) k = AGG_EMPTY_KEY;  // reserved slot holds the marker-valued key
    fin_emit<1>(A, {i}, {k}, {occupied}, {occ}, {wave_base + mbcnt64(occ)}, {wave_base});
  }
}

// Output position of every occupied slot: one atomic per tile of 1024 slots (counts per wave and
// item meet in LDS), not one per slot or per wave - a million groups would otherwise queue tens of
// thousands of atomics on the one counter.  A lane's four slots are emitted together, an output column at a time with
// the four groups' loads in flight at once (fin_emit): item by item a tile was twenty dependent trips to memory, 70 to
// 140 microseconds of them (0.225 -> 0.142 ms at a million groups).  (Measured and not kept: one reservation per
// workgroup for a run of tiles, counted in a pass of its own - fewer, longer workgroups, 0.21 -> 0.34 ms.)
constexpr int FIN_ITEMS = 4;

__global__ __launch_bounds__(256) void qe_finalize_kernel(const FinalizeArgs A) {
  const AggTable &T = A.table;
  __shared__ uint32_t s_count[FIN_ITEMS][4];
  __shared__ unsigned long long s_base;
  const uint32_t wave = threadIdx.x >> 6;
  const uint64_t n_slots = T.cap + 1;
  const uint64_t tile_slots = 256ull * FIN_ITEMS;
  for (uint64_t tile = blockIdx.x * tile_slots; tile < n_slots; tile += (uint64_t)gridDim.x * tile_slots) {
    unsigned long long key[FIN_ITEMS];
    uint64_t occ[FIN_ITEMS];
#pragma unroll
    for (int j = 0; j < FIN_ITEMS; ++j) {
      const uint64_t i = tile + (uint64_t)j * 256 + threadIdx.x;
      key[j] = i < n_slots ? T.keys[i] : AGG_EMPTY_KEY;
      occ[j] = __ballot(key[j] != AGG_EMPTY_KEY);
      if ((threadIdx.x & 63) == 0) s_count[j][wave] = (uint32_t)__popcll(occ[j]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      for (int j = 0; j < FIN_ITEMS; ++j)
        for (int w = 0; w < 4; ++w) total += s_count[j][w];
      s_base = total ? atomicAdd(A.out_count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    uint64_t slot[FIN_ITEMS];
    unsigned long long kk[FIN_ITEMS], pos[FIN_ITEMS], first_pos[FIN_ITEMS];
    bool occupied[FIN_ITEMS];
#pragma unroll
    for (int j = 0; j < FIN_ITEMS; ++j) {
      slot[j] = tile + (uint64_t)j * 256 + threadIdx.x;
      kk[j] = key[j];
      occupied[j] = kk[j] != AGG_EMPTY_KEY;
      uint32_t before = 0;
      for (int jj = 0; jj < FIN_ITEMS; ++jj)
        for (int w = 0; w < 4; ++w)
          if (jj < j || (jj == j && (uint32_t)w < wave)) before += s_count[jj][w];
      if (slot[j] == T.cap) kk[j] = AGG_EMPTY_KEY;  // reserved slot holds the marker-valued key
      first_pos[j] = s_base + before;
      pos[j] = first_pos[j] + mbcnt64(occ[j]);
    }
    fin_emit<FIN_ITEMS>(A, slot, kk, occupied, occ, pos, first_pos);
    __syncthreads();  // s_count / s_base are reused by the next tile
  }
}

// Fixed-width values (4 or 8 bytes; width 0 = bit-packed booleans) of the rows `ids` (global row numbers over
// several source columns): the output key columns of an aggregate whose groups are identified by a hash.
__global__ void qe_gather_rows_fixed_kernel(const FixedSource *sources, int n_sources, const int64_t *ids, int64_t m, int width, void *out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = ids[i];
    int s = n_sources - 1;
    while (s > 0 && sources[s].first_row > id) --s;
    const int64_t r = id - sources[s].first_row;
    if (width == 8) ((uint64_t *)out)[i] = ((const uint64_t *)sources[s].values)[r];
    else if (width == 4) ((uint32_t *)out)[i] = ((const uint32_t *)sources[s].values)[r];
    else if ((((const uint8_t *)sources[s].values)[r >> 3] >> (r & 7)) & 1u) atomicOr((uint32_t *)out + (i >> 5), 1u << (i & 31));
  }
}

// A hashed group key's equality check: see VerifyArgs.  A thread to a selected row: the row's hash finds its group's slot
// (read-only probe), the slot's MIN(row number) names the representative row, and the two rows' key columns are
// compared value by value (strings: length, then bytes).
__global__ __launch_bounds__(256) void qe_verify_group_keys_kernel(const VerifyArgs A) {
  const AggTable &T = A.table;
  unsigned long long bad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < A.m; i += (int64_t)gridDim.x * blockDim.x) {
    if (A.hash_validity && !((A.hash_validity[i >> 5] >> (i & 31)) & 1u)) continue;
    const unsigned long long key = A.hashes[i];
    int64_t slot = -1;
    if (key == AGG_EMPTY_KEY) {
      slot = (int64_t)T.cap;
    } else {
      const uint64_t mask = T.cap - 1;
      uint64_t s = global_home(T, key, mix64(key));
      for (uint64_t p = 0; p <= mask; ++p) {
        const unsigned long long cur = T.keys[s];
        if (cur == key) {
          slot = (int64_t)s;
          break;
        }
        if (cur == AGG_EMPTY_KEY) break;
        s = (s + 1) & mask;
      }
    }
    if (slot < 0) {  // a selected row whose group is not in the table: the table and the batches disagree
      ++bad;
      continue;
    }
    const int64_t rep = (int64_t)ord_inverse(T.acc[(size_t)A.rep_acc * T.stride + (size_t)slot], VT_I64);
    const int64_t me = A.row_ids[i];
    if (rep == me) continue;
    int sb = A.n_sources - 1;
    while (sb > 0 && A.sources[sb].first_row > rep) --sb;
    const int64_t r1 = me - A.row_base, r2 = rep - A.sources[sb].first_row;
    bool same = true;
    for (int k = 0; k < A.n_keys && same; ++k) {
      const KeyColumnSource &mine = A.sources[(size_t)k * A.n_sources + A.this_source], &theirs = A.sources[(size_t)k * A.n_sources + sb];
      switch (A.key_type[k]) {
        case VT_I32: same = ((const int32_t *)mine.values)[r1] == ((const int32_t *)theirs.values)[r2]; break;
        case VT_I64: same = ((const int64_t *)mine.values)[r1] == ((const int64_t *)theirs.values)[r2]; break;
        case VT_BOOL:
          same = ((((const uint8_t *)mine.values)[r1 >> 3] >> (r1 & 7)) & 1u) == ((((const uint8_t *)theirs.values)[r2 >> 3] >> (r2 & 7)) & 1u);
          break;
        default: {  // utf8
          const int32_t a0 = mine.offsets[r1], a1 = mine.offsets[r1 + 1], b0 = theirs.offsets[r2], b1 = theirs.offsets[r2 + 1];
          same = a1 - a0 == b1 - b0;
          const uint8_t *x = (const uint8_t *)mine.values + a0, *y = (const uint8_t *)theirs.values + b0;
          for (int32_t b = 0; same && b < a1 - a0; ++b) same = x[b] == y[b];
          break;
        }
      }
    }
    if (!same) ++bad;
  }
  bad = wave_sum_u64(bad);
  if (lane_id() == 0 && bad) atomicAdd(A.mismatches, bad);
}

// ------------------------------------------------------------- utf8 helpers
// Output offsets of the selected rows of a string column in one pass: lengths by row id, scanned within a tile of 4096
// rows, the tile's first byte from a decoupled look-back over the tiles' byte counts (the filter's protocol: tiles
// handed out by a ticket counter, one status word a tile), validity bits gathered on the way.  scratch and status (a word per tile) start zeroed;
// scratch->total receives the byte count, out_offsets[m] too.
struct alignas(16) I32x4 { int32_t a, b, c, d; };
constexpr int OFFSETS_THREADS = 1024, OFFSETS_TILE_MIN = 2048;  // (the status array is sized for the smallest tile)
template <int R>  // rows a lane, consecutive: 4, 8 or 16 (measured: 4 and 8 alike, 16 much slower)
__global__ __launch_bounds__(OFFSETS_THREADS) void qe_utf8_gather_offsets_kernel(const Utf8GatherCols C, const int32_t *idx, int64_t m_bound,
                                                                                 const unsigned long long *m_dev) {
  constexpr int OFFSETS_TILE = OFFSETS_THREADS * R;
  // the number of rows: the host's, or - when the host has not waited for it - the word the filter kernel left it in
  const int64_t m = m_dev ? (int64_t)__hip_atomic_load(m_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : m_bound;
  const int64_t n_tiles = (m + OFFSETS_TILE - 1) / OFFSETS_TILE;
  if (m <= 0) {  // no tile ever writes offsets[m] = offsets[0] then
    if (blockIdx.x == 0 && (int)threadIdx.x < C.n) C.col[threadIdx.x].out_offsets[0] = 0;
    return;
  }
  constexpr uint32_t LANES_A_WORD = 32 / R;  // lanes whose rows share one 32-row validity word
  __shared__ int32_t s_wave[OFFSETS_THREADS / 64];
  __shared__ unsigned long long s_tile, s_base;
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  // One ticket = one tile of one column, the columns of a tile next to each other: a tile's predecessor in its column
  // holds an earlier ticket, as the look-back needs, and the tile's row ids are read by all columns at about the same time.
  FilterScratch *const tickets = (FilterScratch *)C.col[0].control;
  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&tickets->ticket, 1ull);
    __syncthreads();
    const int64_t ticket = (int64_t)s_tile;
    const int64_t tile = ticket / C.n;
    if (tile >= n_tiles) break;
    const Utf8GatherCol &col = C.col[ticket - tile * C.n];
    const int32_t *const src_offsets = col.src_offsets;
    const uint8_t *const src_validity = col.src_validity;
    int32_t *const out_offsets = col.out_offsets;
    uint32_t *const out_validity = src_validity ? col.out_validity : nullptr;
    FilterScratch *const scratch = (FilterScratch *)col.control;
    unsigned long long *const status = (unsigned long long *)col.status;
    const int64_t i0 = tile * OFFSETS_TILE + (int64_t)threadIdx.x * R;
    const bool whole = i0 + R <= m;
    int64_t r[R];
    if (whole) {
#pragma unroll
      for (int k = 0; k < R; k += 4) {
        const I32x4 four = *(const I32x4 *)__builtin_assume_aligned(idx + i0 + k, 16);
        r[k] = four.a, r[k + 1] = four.b, r[k + 2] = four.c, r[k + 3] = four.d;
      }
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) r[k] = i0 + k < m ? (int64_t)idx[i0 + k] : -1;
    }
    int32_t len[R];
    int32_t sum = 0;
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      len[k] = 0;
      if (r[k] >= 0) {
        int32_t two[2];  // offsets r and r + 1 in one (unaligned) 8-byte load
        __builtin_memcpy(two, src_offsets + r[k], 8);
        len[k] = two[1] - two[0];
        if (out_validity) bits |= ((src_validity[r[k] >> 3] >> (r[k] & 7)) & 1u) << k;
      }
      sum += len[k];
    }
    if (out_validity) {
      uint32_t w = bits << (R * (lane & (LANES_A_WORD - 1)));
#pragma unroll
      for (uint32_t o = 1; o < LANES_A_WORD; o <<= 1) w |= __shfl_xor(w, o);
      if ((lane & (LANES_A_WORD - 1)) == 0 && i0 < m) out_validity[i0 >> 5] = w;
    }
    int32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(incl, o);
      if ((int)lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < OFFSETS_THREADS / 64; ++w) {
      if ((uint32_t)w < wave) woff += s_wave[w];
      total += s_wave[w];
    }
    if (wave == 0) {
      const unsigned long long excl = lookback_exclusive(status, tile, (unsigned long long)total, lane, &scratch->error);
      if (lane == 0) {
        s_base = excl;
        if (tile == n_tiles - 1) {
          scratch->total = excl + (unsigned long long)total;
          out_offsets[m] = (int32_t)(excl + (unsigned long long)total);
        }
      }
    }
    __syncthreads();
    int32_t run = (int32_t)s_base + woff + incl - sum;
    if (whole) {
#pragma unroll
      for (int k = 0; k < R; k += 4) {
        I32x4 four;
        four.a = run, four.b = run + len[k], four.c = four.b + len[k + 1], four.d = four.c + len[k + 2];
        run = four.d + len[k + 3];
        *(I32x4 *)__builtin_assume_aligned(out_offsets + i0 + k, 16) = four;
      }
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (i0 + k < m) out_offsets[i0 + k] = run;
        run += len[k];
      }
    }
  }
}

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// In-place exclusive scan of one tile per block; tile sums go to block_sums.
__global__ __launch_bounds__(SCAN_BLOCK) void qe_scan_tiles_kernel(int32_t *data, int64_t n, int32_t *block_sums) {
  __shared__ int32_t s_wave[SCAN_BLOCK / 64];
  int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int32_t sum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? data[base + i] : 0;
    sum += v[i];
  }
  int32_t incl = sum;
  const uint32_t lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int32_t t = __shfl_up(incl, o);
    if ((int)lane >= o) incl += t;
  }
  uint32_t wave = threadIdx.x >> 6;
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int32_t woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
    if ((uint32_t)w < wave) woff += s_wave[w];
    total += s_wave[w];
  }
  int32_t run = woff + incl - sum;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) data[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 0 && block_sums) block_sums[blockIdx.x] = total;
}

__global__ void qe_scan_add_kernel(int32_t *data, int64_t n, const int32_t *block_offsets) {
  int64_t i = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x;
  int32_t off = block_offsets[blockIdx.x];
  for (int k = 0; k < SCAN_ITEMS; ++k, i += SCAN_BLOCK)
    if (i < n) data[i] += off;
}

// The bytes of the selected rows, a wave to every 64 output rows, a lane to a row.  The output of those rows is one
// contiguous run of bytes.  When none of the 64 rows is longer than 64 bytes - names, codes, dates, addresses: the usual
// string column - a lane fetches its row with 16-byte loads, drops the row's bytes at their place in a staging row of LDS,
// and the wave writes the run out of LDS as whole aligned dwords, 256 consecutive bytes per store instruction: about
// four vector instructions per output byte less than finding each output byte's row by a search over the offsets,
// which is what rows longer than that (and chunks that contain one) still get - there the lanes walk the output a
// dword each and find each byte's row by binary search in the wave's 65 output offsets (LDS).  (The search-only
// kernel spent 11.1 ms on the ten string columns of 100 M rows, bound by vector instruction issue, not by bytes:
// shortening the search's chain of LDS reads, or taking 256 rows and four dwords a lane at once, changed nothing.)
constexpr int GATHER_WAVES = 4;
constexpr int GATHER_SHORT = 64;  // the longest row a lane copies by itself, in fetches of 16 bytes

// The dword of output bytes p4 .. p4 + 3 (those of them inside [d0, d1)) of the rows whose output offsets are sd[0..rows]
// and source offsets ss[0..rows).
__device__ __forceinline__ uint32_t utf8_gather_dword(const int32_t *sd, const int32_t *ss, int rows, int32_t d0, int32_t d1, int32_t p4,
                                                      const uint8_t *__restrict__ src_bytes) {
  const int32_t first = p4 < d0 ? d0 : p4;
  int lo = 0, hi = rows;  // the last row q with sd[q] <= first (empty rows share their successor's offset)
#pragma unroll
  for (int step = 0; step < 7; ++step) {
    const int mid = (lo + hi) >> 1;
    if (hi - lo > 1) {
      if (sd[mid] <= first) lo = mid;
      else hi = mid;
    }
  }
  uint32_t word = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int32_t p = p4 + k;
    if (p >= d0 && p < d1) {
      while (sd[lo + 1] <= p) ++lo;  // on to the row that holds byte p (skips empty rows)
      word |= (uint32_t)src_bytes[ss[lo] + (p - sd[lo])] << (8 * k);
    }
  }
  return word;
}

// The bytes of one chunk - up to 64 consecutive output rows, a lane to a row - of one column.  sd[0..rows] are the rows'
// output offsets (sd[rows]: where the chunk ends), ss[0..rows) their source offsets, both in LDS and written before the
// call; `stage`: 64 * GATHER_SHORT + 16 bytes of LDS of the wave's own.  (src_total, the bytes of the source column: a
// 16-byte fetch must end inside them.)
__device__ __forceinline__ void utf8_copy_chunk(const uint8_t *__restrict__ src_bytes, int64_t src_total, uint8_t *__restrict__ dst_bytes, bool valid,
                                                int32_t my_src, int32_t my_len, int32_t my_dst, const int32_t *sd, const int32_t *ss, uint8_t *stage,
                                                int rows, uint32_t lane) {
  const int32_t d0 = sd[0], d1 = sd[rows];
  const int32_t a0 = d0 & ~3;  // the run's first dword
  const bool in_bounds = (int64_t)my_src + ((my_len + 15) & ~15) <= src_total;  // (a 16-byte fetch of the row's last piece ends inside the source)
  const bool is_short = my_len <= GATHER_SHORT && in_bounds;
  // rows a lane cannot copy by itself: longer than 64 bytes (or at the very end of the source)
  const uint64_t long_rows = __ballot(valid && !is_short);
  if (long_rows == 0 || d1 - a0 <= (int32_t)(64 * GATHER_SHORT)) {
    // ---- the run assembled in LDS: a lane fetches its own short row in pieces of 16 bytes; a row too long for that is
    // fetched by the whole wave, a lane to a piece, the rare one (a chunk with one such row used to send all its 64 rows
    // down the search path below: a column with 2 % long rows did so for 70 % of its chunks)
    const uint32_t short_len = (valid && is_short) ? (uint32_t)my_len : 0u;
    const int pieces = (__ballot(short_len > 0) != 0) + (__ballot(short_len > 16) != 0) + (__ballot(short_len > 32) != 0) + (__ballot(short_len > 48) != 0);
    auto stage_piece = [&](const uint8_t *from, uint8_t *at, int32_t here) {  // `here` (1..16) bytes of a 16-byte fetch
      uint32_t w[4];
      __builtin_memcpy(w, from, 16);  // (any alignment: one 16-byte load)
      // whole dwords as (unaligned) dword writes, the up to three bytes after them one by one: seven LDS
      // instructions a piece at most, not sixteen
      const int32_t full = here >> 2, tail = here & 3;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < full) __builtin_memcpy(at + 4 * q, &w[q], 4);
      if (tail) {
        const uint32_t tw = full == 0 ? w[0] : full == 1 ? w[1] : full == 2 ? w[2] : w[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (t < tail) at[4 * full + t] = (uint8_t)(tw >> (8 * t));
      }
    };
    for (int piece = 0; piece < pieces; ++piece) {
      const int32_t left = (int32_t)short_len - 16 * piece;
      if (left > 0) stage_piece(src_bytes + my_src + 16 * piece, stage + (my_dst - a0) + 16 * piece, left < 16 ? left : 16);
    }
    for (uint64_t rest = long_rows; rest != 0; rest &= rest - 1) {
      const int q = __builtin_ctzll(rest);
      const int32_t q_src = __builtin_amdgcn_readlane(my_src, q), q_dst = __builtin_amdgcn_readlane(my_dst, q), q_len = __builtin_amdgcn_readlane(my_len, q);
      for (int32_t off = 16 * (int32_t)lane; off < q_len; off += 16 * 64) {
        const int32_t left = q_len - off;
        if ((int64_t)q_src + off + 16 <= src_total) {
          stage_piece(src_bytes + q_src + off, stage + (q_dst - a0) + off, left < 16 ? left : 16);
        } else {  // the source's last bytes: one at a time
          for (int32_t b = 0; b < left && b < 16; ++b) stage[(q_dst - a0) + off + b] = src_bytes[q_src + off + b];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int32_t p4 = a0 + 4 * (int32_t)lane; p4 < d1; p4 += 256) {
      const uint32_t word = *(const uint32_t *)(stage + (p4 - a0));
      if (p4 >= d0 && p4 + 4 <= d1) {
        *(uint32_t *)(dst_bytes + p4) = word;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (p4 + k >= d0 && p4 + k < d1) dst_bytes[p4 + k] = (uint8_t)(word >> (8 * k));
      }
    }
  } else {
    // ---- a run too long for the staging row: a lane takes four consecutive output bytes (one aligned dword store; the
    // run's ragged ends go out byte by byte: the bytes next to them belong to other waves) and finds each byte's row
    for (int32_t p4 = a0 + 4 * (int32_t)lane; p4 < d1; p4 += 256) {
      const uint32_t word = utf8_gather_dword(sd, ss, rows, d0, d1, p4, src_bytes);
      if (p4 >= d0 && p4 + 4 <= d1) {
        *(uint32_t *)(dst_bytes + p4) = word;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (p4 + k >= d0 && p4 + k < d1) dst_bytes[p4 + k] = (uint8_t)(word >> (8 * k));
      }
    }
  }
}

// Every column of the launch, chunk by chunk.
__global__ __launch_bounds__(GATHER_WAVES * 64) void qe_utf8_gather_bytes_kernel(const Utf8GatherCols C, const int32_t *idx, int64_t m_bound,
                                                                                 const unsigned long long *m_dev) {
  const int64_t m = m_dev ? (int64_t)__hip_atomic_load(m_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : m_bound;
  __shared__ int32_t s_dst[GATHER_WAVES][65];
  __shared__ int32_t s_src[GATHER_WAVES][64];
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[GATHER_WAVES][64 * GATHER_SHORT + 16];
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  const int64_t n_chunks = (m + 63) / 64;
  for (int64_t chunk = (int64_t)blockIdx.x * GATHER_WAVES + wave; chunk < n_chunks; chunk += (int64_t)gridDim.x * GATHER_WAVES) {
    const int64_t i = chunk * 64 + lane;
    const int rows = (int)(m - chunk * 64 < 64 ? m - chunk * 64 : 64);
    // the chunk's row ids once for all columns; a column's offsets are fetched while the column before it is copied
    const int64_t r = i < m ? (int64_t)idx[i] : 0;
    int32_t n_two[2] = {0, 0}, n_dst = 0, n_end = 0;
    auto fetch_offsets = [&](int c) {
      if (c < C.n && i < m) {
        __builtin_memcpy(n_two, C.col[c].src_offsets + r, 8);
        n_dst = C.col[c].out_offsets[i];
      }
      if (c < C.n && lane == 0) n_end = C.col[c].out_offsets[chunk * 64 + rows];
    };
    fetch_offsets(0);
    for (int c = 0; c < C.n; ++c) {
      const Utf8GatherCol &col = C.col[c];
      int32_t my_src = 0, my_dst = 0, my_len = 0;
      if (i < m) {
        my_src = n_two[0];
        my_len = n_two[1] - n_two[0];
        my_dst = n_dst;
        s_src[wave][lane] = my_src;
        s_dst[wave][lane] = my_dst;
      }
      if (lane == 0) s_dst[wave][rows] = n_end;
      fetch_offsets(c + 1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave's own LDS writes, before its reads below
      __builtin_amdgcn_wave_barrier();
      utf8_copy_chunk(col.src_bytes, col.src_total, col.out_bytes, i < m, my_src, my_len, my_dst, s_dst[wave], s_src[wave], s_stage[wave], rows, lane);
      __builtin_amdgcn_wave_barrier();  // the next column (or chunk) overwrites the offsets and the staging row
    }
  }
}

// Offsets and bytes in one pass, for a caller that has the output bytes' buffers before it knows how many there are (a
// Selection that does not wait for its row count sizes them by the source): a tile's rows are scanned and placed as in
// the offsets kernel, and the workgroup then copies their bytes itself, a wave to 64 consecutive rows at a time - the
// source offsets, the output offsets and the row ids are in its registers, where the two-kernel form reads them from
// memory a second time (per selected row and column: 8 bytes of source offsets, scattered, 4 of output offsets, 4 of
// row id) and pays a launch's tail twice.
// (a tile of 2048 rows, three workgroups - 24 waves - to a compute unit, which is what their LDS allows: 6.3 ms for the ten
// string columns of 100 M rows, 7.1 for offsets and bytes in a launch each; tiles of 4096 rows with two workgroups to a
// unit 7.3, tiles of 1024 with six 7.0: fewer waves to hide the fetches behind, or more look-backs)
constexpr int FUSED_THREADS = 512, FUSED_R = 4;
constexpr size_t FUSED_WAVE_LDS = 64 * GATHER_SHORT + 16 + (64 * FUSED_R + 4) * 4 + 64 * FUSED_R * 4;
__global__ __launch_bounds__(FUSED_THREADS) void qe_utf8_gather_fused_kernel(const Utf8GatherCols C, const int32_t *idx, int64_t m_bound,
                                                                             const unsigned long long *m_dev) {
  constexpr int R = FUSED_R, TILE = FUSED_THREADS * R, WAVES = FUSED_THREADS / 64;
  const int64_t m = m_dev ? (int64_t)__hip_atomic_load(m_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : m_bound;
  const int64_t n_tiles = (m + TILE - 1) / TILE;
  if (m <= 0) {  // no tile ever writes offsets[m] = offsets[0] then
    if (blockIdx.x == 0 && (int)threadIdx.x < C.n) C.col[threadIdx.x].out_offsets[0] = 0;
    return;
  }
  constexpr uint32_t LANES_A_WORD = 32 / R;  // lanes whose rows share one 32-row validity word
  __shared__ int32_t s_wave[WAVES];
  __shared__ unsigned long long s_tile, s_base;
  // per wave (dynamic LDS, FUSED_WAVE_LDS bytes each): the staging row, the output offsets of the wave's rows (and where
  // they end), their source offsets
  extern __shared__ __attribute__((aligned(16))) uint8_t fused_lds[];
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  uint8_t *const my_stage = fused_lds + (size_t)wave * FUSED_WAVE_LDS;
  int32_t *const my_dst_row = (int32_t *)(my_stage + 64 * GATHER_SHORT + 16);
  int32_t *const my_src_row = my_dst_row + 64 * R + 4;
  FilterScratch *const tickets = (FilterScratch *)C.col[0].control;
  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&tickets->ticket, 1ull);
    __syncthreads();
    const int64_t ticket = (int64_t)s_tile;
    const int64_t tile = ticket / C.n;
    if (tile >= n_tiles) break;
    const Utf8GatherCol &col = C.col[ticket - tile * C.n];
    const int32_t *const src_offsets = col.src_offsets;
    const uint8_t *const src_validity = col.src_validity;
    int32_t *const out_offsets = col.out_offsets;
    uint32_t *const out_validity = src_validity ? col.out_validity : nullptr;
    FilterScratch *const scratch = (FilterScratch *)col.control;
    unsigned long long *const status = (unsigned long long *)col.status;
    const int64_t i0 = tile * TILE + (int64_t)threadIdx.x * R;
    const bool whole = i0 + R <= m;
    int64_t r[R];
    if (whole) {
#pragma unroll
      for (int k = 0; k < R; k += 4) {
        const I32x4 four = *(const I32x4 *)__builtin_assume_aligned(idx + i0 + k, 16);
        r[k] = four.a, r[k + 1] = four.b, r[k + 2] = four.c, r[k + 3] = four.d;
      }
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) r[k] = i0 + k < m ? (int64_t)idx[i0 + k] : -1;
    }
    int32_t len[R];
    int32_t sum = 0;
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      len[k] = 0;
      int32_t two[2] = {0, 0};  // offsets r and r + 1 in one (unaligned) 8-byte load
      if (r[k] >= 0) {
        __builtin_memcpy(two, src_offsets + r[k], 8);
        len[k] = two[1] - two[0];
        if (out_validity) bits |= ((src_validity[r[k] >> 3] >> (r[k] & 7)) & 1u) << k;
      }
      my_src_row[R * lane + k] = two[0];
      sum += len[k];
    }
    if (out_validity) {
      uint32_t w = bits << (R * (lane & (LANES_A_WORD - 1)));
#pragma unroll
      for (uint32_t o = 1; o < LANES_A_WORD; o <<= 1) w |= __shfl_xor(w, o);
      if ((lane & (LANES_A_WORD - 1)) == 0 && i0 < m) out_validity[i0 >> 5] = w;
    }
    int32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(incl, o);
      if ((int)lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      if ((uint32_t)w < wave) woff += s_wave[w];
      total += s_wave[w];
    }
    if (wave == 0) {
      const unsigned long long excl = lookback_exclusive(status, tile, (unsigned long long)total, lane, &scratch->error);
      if (lane == 0) {
        s_base = excl;
        if (tile == n_tiles - 1) {
          scratch->total = excl + (unsigned long long)total;
          out_offsets[m] = (int32_t)(excl + (unsigned long long)total);
        }
      }
    }
    __syncthreads();
    int32_t run = (int32_t)s_base + woff + incl - sum;
    {
      int32_t at = run;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        my_dst_row[R * lane + k] = at;
        at += len[k];
      }
      if (lane == 63) my_dst_row[64 * R] = at;
    }
    if (whole) {
#pragma unroll
      for (int k = 0; k < R; k += 4) {
        I32x4 four;
        four.a = run, four.b = run + len[k], four.c = four.b + len[k + 1], four.d = four.c + len[k + 2];
        run = four.d + len[k + 3];
        *(I32x4 *)__builtin_assume_aligned(out_offsets + i0 + k, 16) = four;
      }
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (i0 + k < m) out_offsets[i0 + k] = run;
        run += len[k];
      }
    }
    // ---- the bytes of the wave's 64 R rows, 64 rows at a time
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave's own LDS writes, before its reads below
    __builtin_amdgcn_wave_barrier();
    const int64_t w0 = tile * TILE + (int64_t)wave * 64 * R;  // the wave's first row
#pragma unroll 1
    for (int c = 0; c < R; ++c) {
      const int64_t first = w0 + 64 * c;
      if (first >= m) break;
      const int rows = (int)(m - first < 64 ? m - first : 64);
      const int32_t *sd = my_dst_row + 64 * c, *ss = my_src_row + 64 * c;
      const bool valid = (int)lane < rows;
      // (rows beyond m hold a length of 0 and the offset the last real row ends at: sd[rows] is where the chunk ends)
      const int32_t my_src = ss[lane], my_dst = sd[lane], my_len = valid ? sd[lane + 1] - my_dst : 0;
      utf8_copy_chunk(col.src_bytes, col.src_total, col.out_bytes, valid, my_src, my_len, my_dst, sd, ss, my_stage, rows, lane);
      __builtin_amdgcn_wave_barrier();  // the next chunk overwrites the staging row
    }
  }
}

__global__ void qe_utf8_gather_rows_kernel(const Utf8Source *sources, int n_sources, const int64_t *ids, int64_t m, int32_t *out_lengths,
                                           const int32_t *dst_offsets, uint8_t *dst_bytes) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = ids[i];
    int s = n_sources - 1;
    while (s > 0 && sources[s].first_row > id) --s;
    const int64_t r = id - sources[s].first_row;
    const int32_t s0 = sources[s].offsets[r], len = sources[s].offsets[r + 1] - s0;
    if (out_lengths) {
      out_lengths[i] = len;
    } else {
      const int32_t d0 = dst_offsets[i];
      for (int32_t b = 0; b < len; ++b) dst_bytes[d0 + b] = sources[s].bytes[s0 + b];
    }
  }
}

__global__ void qe_utf8_broadcast_kernel(const uint8_t *lit, int32_t len, int64_t n, int32_t *out_offsets,
                                         uint8_t *out_bytes) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
    out_offsets[i] = (int32_t)(i * len);
    if (i < n)
      for (int32_t b = 0; b < len; ++b) out_bytes[i * len + b] = lit[b];
  }
}

__global__ void qe_count_zero_bits_kernel(const uint32_t *bitmap, int64_t n, unsigned long long *out) {
  int64_t nwords = (n + 31) / 32;
  unsigned long long local = 0;
  for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = bitmap[w];
    int64_t rows = n - w * 32;
    uint32_t live = rows >= 32 ? 0xffffffffu : ((1u << rows) - 1u);
    local += __popc(~x & live);
  }
  local = wave_sum_u64(local);
  if (lane_id() == 0 && local) atomicAdd(out, local);
}

// ---------------------------------------------------------- synthetic data
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

__global__ void qe_generate_kernel(int kind, uint64_t seed, int64_t row0, int64_t n, double lo, double hi, void *out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)(row0 + i)));
    double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
    if (kind == 0) {
      ((double *)out)[i] = lo + u * (hi - lo);
    } else if (kind == 1) {
      int64_t a = (int64_t)lo, b = (int64_t)hi;
      uint64_t range = (uint64_t)(b - a);
      ((int64_t *)out)[i] = a + (int64_t)(range ? r % range : 0);
    } else if (kind == 2) {
      // power-law rank over {0..N-1} with exponent s = hi (> 1): inverse CDF of the continuous law
      double N = lo, s = hi;
      double e = 1.0 - s;
      double x = pow(1.0 + u * (pow(N + 1.0, e) - 1.0), 1.0 / e);
      int64_t rank = (int64_t)x - 1;
      if (rank < 0) rank = 0;
      if (rank >= (int64_t)N) rank = (int64_t)N - 1;
      // spread the ranks over the int64 domain with a bijection so hot keys are not neighbours
      ((int64_t *)out)[i] = (int64_t)splitmix64((uint64_t)rank ^ 0x5851f42d4c957f2dull);
    } else {
      // passenger_count style key over {0..9}: mode 1 (~70 %), then 2, 5, 3, 6, 4, 0, 7, 8, 9
      int64_t k;
      if (u < 0.70) k = 1;
      else if (u < 0.84) k = 2;
      else if (u < 0.88) k = 5;
      else if (u < 0.92) k = 3;
      else if (u < 0.95) k = 6;
      else if (u < 0.97) k = 4;
      else if (u < 0.99) k = 0;
      else if (u < 0.994) k = 7;
      else if (u < 0.997) k = 8;
      else k = 9;
      ((int64_t *)out)[i] = k;
    }
  }
}

// ---------------------------------------------------------------- launchers
// Does the program use an instruction only the FULL interpreter variant carries?
static bool program_needs_full(const VmProgram &P) {
  for (int i = 0; i < VM_MAX_INSN; ++i) {
    uint32_t op = P.insn[i] & 0xffu;
    bool fast = op == OP_END || op == OP_PUSH_LIT || vm_op_is_cached_push(op) || (op >= OP_ADD_I32 && op <= OP_MUL_I32) ||
                (op >= OP_ADD_I64 && op <= OP_MUL_I64) || (op >= OP_ADD_F64 && op <= OP_MUL_F64) ||
                (op >= OP_EQ_I64 && op <= OP_OR);
    if (!fast) return true;
  }
  return false;
}

// Workgroups per compute unit of each grid-stride kernel.  Defaults from measurement on MI355X; the
// environment variables exist for tuning runs (read once per process).
static int tuned(const char *env, int dflt) {
  const char *v = getenv(env);
  int x = v ? atoi(v) : 0;
  return x > 0 ? x : dflt;
}
static int project_blocks_per_cu() { static int v = tuned("QE_PROJECT_BLOCKS_PER_CU", 2); return v; }
static int filter_blocks_per_cu() { static int v = tuned("QE_FILTER_BLOCKS_PER_CU", 8); return v; }

static bool jit_wanted(const DeviceInfo &dev, int64_t n_rows) {
  return dev.specialised && dev.jit != JIT_OFF && (dev.jit == JIT_SYNC || n_rows >= dev.jit_min_rows);
}

static int grid_for(int64_t work_items, int per_block, int max_blocks) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

hipError_t launch_project(const ProjectArgs &args, const DeviceInfo &dev, hipStream_t stream, const char **kernel_name) {
  if (args.n_rows <= 0 || args.n_outs <= 0) return hipSuccess;
  int grid = grid_for(args.n_rows, VM_BLOCK_ROWS, dev.cus * project_blocks_per_cu());
  hipError_t err;
  if (dev.specialised && static_launch_project(args, grid, stream, &err, kernel_name)) return err;
  if (jit_wanted(dev, args.n_rows) &&
      jit_launch(JIT_PROJECT, spec_fields(args), &args, dev.jit, dev.device, grid, VM_BLOCK_THREADS, 0, stream, &err, kernel_name))
    return err;
  if (program_needs_full(args.prog))
    hipLaunchKernelGGL(qe_project_kernel<true>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, stream, args);
  else
    hipLaunchKernelGGL(qe_project_kernel<false>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, stream, args);
  return hipGetLastError();
}

int64_t filter_num_tiles(int64_t n_rows) { return (n_rows + FILTER_TILE_ROWS - 1) / FILTER_TILE_ROWS; }
size_t filter_scratch_bytes(int64_t n_rows) {
  return sizeof(FilterScratch) + (size_t)filter_num_tiles(n_rows) * sizeof(unsigned long long);
}

hipError_t launch_filter_reset(FilterScratch *scratch, int64_t n_rows, hipStream_t stream) {
  return hipMemsetAsync(scratch, 0, filter_scratch_bytes(n_rows), stream);
}

hipError_t launch_filter(const FilterArgs &args, const DeviceInfo &dev, hipStream_t stream, const char **kernel_name) {
  if (args.n_rows <= 0) return hipSuccess;
  int grid = grid_for(args.n_tiles, 1, dev.cus * filter_blocks_per_cu());
  hipError_t err;
  if (dev.specialised && static_launch_filter(args, dev.cus, stream, &err, kernel_name)) return err;
  if (jit_wanted(dev, args.n_rows)) {
    // geometry of the specialised filter kernels: bigger tiles, one workgroup per compute unit
    FilterArgs b = args;
    const int tile = filter_static_tile_rows((int)args.prog.cache_n);
    b.n_tiles = (args.n_rows + tile - 1) / tile;
    const unsigned g = (unsigned)(b.n_tiles < dev.cus ? b.n_tiles : dev.cus);
    if (jit_launch(JIT_FILTER, spec_fields(b), &b, dev.jit, dev.device, g, filter_static_waves((int)args.prog.cache_n) * 64, 0, stream, &err,
                   kernel_name))
      return err;
  }
  if (program_needs_full(args.prog))
    hipLaunchKernelGGL(qe_filter_kernel<true>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, stream, args);
  else
    hipLaunchKernelGGL(qe_filter_kernel<false>, dim3(grid), dim3(VM_BLOCK_THREADS), 0, stream, args);
  return hipGetLastError();
}

static size_t agg_lds_bytes(int n_accs, uint32_t slots, bool with_nn = true) {
  size_t stride = (size_t)slots + 1;
  return stride * 8 + stride * 8 * n_accs + stride * 4 + (with_nn ? stride * 4 * n_accs : 0) + 16;
}

// LDS table slots.  A table much larger than the number of groups only costs occupancy (with a
// 64 KiB table two workgroups fit a compute unit; a low-cardinality aggregate with a 1024-slot table
// runs four and measures 20-45 % faster), so the caller's group-count hint sizes it: 4 x hint slots,
// at least 256.  Without a hint the cardinality is unknown and the table takes the full budget.
// `wide`: the launch is one workgroup of AGG_WIDE_THREADS threads per compute unit, whose table may
// take nearly all of the unit's LDS.
uint32_t aggregate_lds_slots(const AggArgs &args, const DeviceInfo &dev, int64_t group_hint, bool wide) {
  const bool with_nn = agg_tracks_validity(args.accs, args.n_accs);
  size_t budget = wide ? (size_t)150 * 1024 : (size_t)64 * 1024;  // (wide: the admission tags come on top)
  if (!wide && (size_t)dev.max_lds < budget) budget = dev.max_lds;
  uint32_t slots = wide ? 8192 : 4096;
  static int cap = tuned("QE_AGGREGATE_LDS_SLOTS", 8192);
  while (slots > 16 && (agg_lds_bytes(args.n_accs, slots, with_nn) > budget || slots > (uint32_t)cap)) slots >>= 1;
  if (group_hint > 0) {
    uint64_t want = 256;
    while (want < (uint64_t)group_hint * 4 && want < slots) want <<= 1;
    if (want < slots) slots = (uint32_t)want;
  }
  return slots;
}

static bool aggregate_is_wide(const AggArgs &args) { return args.block_threads == (uint32_t)AGG_WIDE_THREADS; }

int aggregate_grid(const AggArgs &args, const DeviceInfo &dev) {
  if (aggregate_is_wide(args)) {
    static int wide_per_cu = tuned("QE_AGGREGATE_WIDE_BLOCKS_PER_CU", 1);
    return grid_for(args.n_rows, VM_WAVE_ROWS * (AGG_WIDE_THREADS / 64), dev.cus * wide_per_cu);
  }
  size_t lds = agg_lds_bytes(args.n_accs, args.lds_cap, agg_tracks_validity(args.accs, args.n_accs));
  // Resident workgroups per compute unit: enough waves to keep about 96 KiB of column loads in
  // flight per compute unit (a wave has 2 KiB per cached column outstanding), within what the LDS
  // tables allow.  Measured optima on MI355X: 3 workgroups for four 8-byte columns, 6 for two;
  // both more and fewer are slower by 5-15 %.
  int cols = args.prog.cache_n > 0 ? (int)args.prog.cache_n : 1;
  int per_cu = 12 / cols;
  per_cu = per_cu < 2 ? 2 : per_cu > 8 ? 8 : per_cu;
  int lds_fit = (int)((size_t)144 * 1024 / (lds + 512));
  if (lds_fit < 1) lds_fit = 1;
  if (per_cu > lds_fit) per_cu = lds_fit;
  static int forced = tuned("QE_AGGREGATE_BLOCKS_PER_CU", 0);
  if (forced > 0) per_cu = forced;
  int grid = grid_for(args.n_rows, VM_BLOCK_ROWS, dev.cus * per_cu);
  static int forced_grid = tuned("QE_AGGREGATE_GRID", 0);
  if (forced_grid > 0) grid = grid_for(args.n_rows, VM_BLOCK_ROWS, forced_grid);
  return grid;
}

// Can this launch use the wide geometry?  Only a kernel specialised for its shape can (the general kernels are built
// for 256-thread workgroups): one built ahead of time, or one the run-time compiler has ready - or makes ready now,
// when the context compiles on first use.
bool aggregate_wide_available(const AggArgs &args, const DeviceInfo &dev) {
  static int enabled = tuned("QE_AGGREGATE_WIDE", 1);
  if (enabled == 2 || !dev.specialised) return false;
  if (static_has_aggregate(args)) return true;
  if (!jit_wanted(dev, args.n_rows)) return false;
  // (the first-pass kernel of the many-groups mode is asked for by nobody else: its background compilation starts here)
  return (args.tune & 2u) ? jit_ready(JIT_AGGREGATE_SPILL, spec_fields_spill(args), dev.jit, true) : jit_ready(JIT_AGGREGATE, spec_fields(args), dev.jit);
}

// Dense-key bins (see qe_static.h) for aggregates with few groups whose inputs cannot be null, when
// the bins fit next to the LDS table without costing a resident workgroup.
static bool dense_bins_wanted(const AggArgs &args, size_t lds, int grid, const DeviceInfo &dev) {
  static int enabled = tuned("QE_AGGREGATE_DENSE_BINS", 1);
  if (enabled == 2 || args.n_accs == 0 || args.lds_cap > 1024 || args.spill.keys != nullptr) return false;
  for (int a = 0; a < args.n_accs; ++a)
    if (args.accs[a].track_nn) return false;
  const size_t total = ((lds + 15) & ~(size_t)15) + dense_bins_bytes(args.n_accs);
  const int per_cu = (grid + dev.cus - 1) / (dev.cus > 0 ? dev.cus : 1);
  return total <= 64 * 1024 && (size_t)per_cu * (total + 512) <= (size_t)144 * 1024;
}

// LDS of a wide launch before the partition histogram: the table, then the admission tags of the many-groups mode.
static size_t wide_lds_before_histogram(const AggArgs &args, uint32_t *door_off) {
  size_t lds = agg_lds_bytes(args.n_accs, args.lds_cap, agg_tracks_validity(args.accs, args.n_accs));
  static int door = tuned("QE_AGGREGATE_DOOR", 1);
  if (door_off) *door_off = 0;
  if (door != 2 && args.spill.keys != nullptr && lds + 16 + DOOR_ENTRIES * 2 <= (size_t)159 * 1024) {
    const uint32_t at = (uint32_t)((lds + 15) & ~(size_t)15);
    if (door_off) *door_off = at;
    lds = at + DOOR_ENTRIES * 2;
  }
  return lds;
}

static bool spill_kernel_wanted(const AggArgs &args) {
  // (QE_AGGREGATE_SPILL_KERNEL=2: the general specialised kernel instead of the first pass's own, for comparison)
  static int lean = tuned("QE_AGGREGATE_SPILL_KERNEL", 1);
  return lean != 2 && args.spill.keys != nullptr && (args.tune & 2u) != 0;
}

bool aggregate_counts_partitions(const AggArgs &args, const DeviceInfo &dev, int bits) {
  (void)dev;
  if (!aggregate_is_wide(args) || !spill_kernel_wanted(args) || args.spill.seg_chunk != (uint32_t)(AGG_WIDE_THREADS / 64)) return false;
  const size_t hist_at = (wide_lds_before_histogram(args, nullptr) + 15) & ~(size_t)15;
  return hist_at + ((size_t)4 << bits) <= (size_t)160 * 1024;
}

// Rows a (partition, workgroup) stretch holds: the workgroup's expected share of the batch's rows in that partition
// if every row is selected, eight standard deviations and a few lines on top, whole lines.  Keys spread evenly never
// need more; keys that do not are what the logging first pass is for.
uint32_t aggregate_direct_stretch_rows(int64_t n_rows, int bits, int blocks) {
  const double mean = (double)n_rows / ((double)((int64_t)1 << bits) * (double)blocks);
  const double rows = mean + 8.0 * std::sqrt(mean) + 64.0;
  return (uint32_t)(((int64_t)rows + 7) & ~(int64_t)7);
}

bool aggregate_direct_available(const AggArgs &args, const DeviceInfo &dev, int bits) {
  static int on = tuned("QE_AGGREGATE_DIRECT", 1);  // 2: off
  if (on == 2 || !dev.specialised || bits > 9 || (1 << bits) > DIRECT_MAX_PARTS || direct_lds_bytes(bits) > (size_t)160 * 1024) return false;
  if (static_has_aggregate(args)) return true;
  if (!jit_wanted(dev, args.n_rows)) return false;
  return jit_ready(JIT_AGGREGATE_DIRECT, spec_fields_direct(args), dev.jit, true);
}

hipError_t launch_aggregate_direct(const AggArgs &args_in, const DeviceInfo &dev, hipStream_t stream, const char **kernel_name) {
  if (args_in.n_rows <= 0) return hipSuccess;
  if (args_in.spill.direct_cap == 0 || (args_in.spill.direct_cap & 7u) || args_in.cells == nullptr || args_in.cells_total == nullptr ||
      args_in.spill.counts == nullptr || args_in.cell_bits > 9)
    return hipErrorInvalidValue;
  const size_t lds = direct_lds_bytes(args_in.cell_bits);
  const int grid = dev.cus;
  hipError_t err;
  AggArgs args = args_in;
  static int streaming = tuned("QE_DIRECT_STREAMING_STORES", 1);  // (2.12 -> 2.06 ms per 256 M rows of a million uniform keys)
  if (streaming == 1) args.tune |= 0x10u;  // (bit 4: complete lines leave as streaming stores)
#ifdef QE_ABLATE  // timing experiments only (results are wrong): compiled in by a build with -DQE_ABLATE, never in the shipped library
  static int ablate = tuned("QE_DIRECT_ABLATE", 0);  // 0x200: rows go nowhere, 0x400: lines are not written, 0x800: no rounds either
  args.tune |= (uint32_t)ablate & 0xe00u;
#endif
  if (static_launch_aggregate(args, grid, AGG_WIDE_THREADS, lds, stream, &err, kernel_name, 2)) return err;
  if (jit_launch(JIT_AGGREGATE_DIRECT, spec_fields_direct(args), &args, dev.jit, dev.device, grid, AGG_WIDE_THREADS, lds, stream, &err, kernel_name)) return err;
  return hipErrorInvalidConfiguration;
}

hipError_t launch_aggregate(const AggArgs &args_in, const DeviceInfo &dev, hipStream_t stream, const char **kernel_name, bool *counted) {
  if (counted) *counted = false;
  if (args_in.n_rows <= 0) return hipSuccess;
  AggArgs args = args_in;
  size_t lds = agg_lds_bytes(args.n_accs, args.lds_cap, agg_tracks_validity(args.accs, args.n_accs));
  int grid = aggregate_grid(args, dev);
  args.dense_off = 0;
  if (aggregate_is_wide(args)) {
    // the caller checked aggregate_wide_available: one of the two specialised routes takes the launch
    hipError_t werr;
    lds = wide_lds_before_histogram(args, &args.door_off);
    // the wide launch is the high-cardinality mode's first pass: a kernel of its own
    // Mid-range cardinalities (a few thousand groups, possibly skewed) keep the general kernel, whose hot keys in
    // registers take a dominant key off the LDS atomics.
    const bool many_groups = args.spill.keys != nullptr && (args.tune & 2u) != 0;
#ifdef QE_ABLATE  // timing experiments only (results are wrong): compiled in by a build with -DQE_ABLATE, never in the shipped library
    static int no_flush = tuned("QE_SPILL_ABLATE_FLUSH", 0);
    if (no_flush) args.tune |= 4u;
    static int first = tuned("QE_SPILL_ABLATE_FIRST", 0);  // 16: no log stores, 32: no look at the table, 64: no partition histogram, 128: no value stores
    args.tune |= (uint32_t)first & 0x1f0u;
#endif
    static int prio = tuned("QE_SPILL_PRIO", 1);
    if (prio) args.tune |= 8u;  // (bit 3: the waves of a SIMD take the high priority in turn)
    const bool spill_kernel = spill_kernel_wanted(args);
    if (many_groups && !spill_kernel) args.tune |= 1u;  // (with many groups the hot keys cost more than they save: 2.67 -> 2.42 ms per 256 M rows)
    // the first pass counts its logged rows per partition when the caller asked for it (args.cells), the launch has one
    // workgroup per column of the cell table, and the histogram fits next to the table
    const size_t hist_at = (lds + 15) & ~(size_t)15;
    const bool counts = spill_kernel && args.cells != nullptr && args.spill.seg_chunk == (uint32_t)(AGG_WIDE_THREADS / 64) &&
                        hist_at + ((size_t)4 << args.cell_bits) <= (size_t)160 * 1024;
    if (counts) {
      args.hist_off = (uint32_t)hist_at;
      lds = hist_at + ((size_t)4 << args.cell_bits);
    } else {
      args.cells = nullptr;
    }
    if (counted) *counted = counts;
    if (static_launch_aggregate(args, grid, AGG_WIDE_THREADS, lds, stream, &werr, kernel_name, spill_kernel ? 1 : 0)) return werr;
    if (spill_kernel) {
      if (jit_launch(JIT_AGGREGATE_SPILL, spec_fields_spill(args), &args, dev.jit, dev.device, grid, AGG_WIDE_THREADS, lds, stream, &werr, kernel_name)) return werr;
    } else if (jit_launch(JIT_AGGREGATE, spec_fields(args), &args, dev.jit, dev.device, grid, AGG_WIDE_THREADS, lds, stream, &werr, kernel_name)) {
      return werr;
    }
    return hipErrorInvalidConfiguration;
  }
  args.cells = nullptr;
  if (dev.specialised && dense_bins_wanted(args, lds, grid, dev)) {
    args.dense_off = (int32_t)((lds + 15) & ~(size_t)15);
    lds = (size_t)args.dense_off + dense_bins_bytes(args.n_accs);
    // with the bins a two-column group-by is at its best with five workgroups per compute unit
    // (measured: 4: +3 %, 6: +9 %, 8: +23 % time)
    static int forced = tuned("QE_AGGREGATE_BLOCKS_PER_CU", 0) + tuned("QE_AGGREGATE_GRID", 0);
    if (forced == 0 && grid > dev.cus * 5) grid = dev.cus * 5;
  }
  hipError_t err;
  if (dev.specialised && static_launch_aggregate(args, grid, VM_BLOCK_THREADS, lds, stream, &err, kernel_name)) return err;
  if (jit_wanted(dev, args.n_rows) &&
      jit_launch(JIT_AGGREGATE, spec_fields(args), &args, dev.jit, dev.device, grid, VM_BLOCK_THREADS, lds, stream, &err, kernel_name))
    return err;
  if (args.n_more > 0) return hipErrorInvalidConfiguration;  // (the interpreter kernels take one batch: aggregate_takes_more_batches said otherwise?)
  if (program_needs_full(args.prog))
    hipLaunchKernelGGL(qe_aggregate_kernel<true>, dim3(grid), dim3(VM_BLOCK_THREADS), lds, stream, args);
  else
    hipLaunchKernelGGL(qe_aggregate_kernel<false>, dim3(grid), dim3(VM_BLOCK_THREADS), lds, stream, args);
  return hipGetLastError();
}

size_t spill_cells(int bits, int blocks) { return ((size_t)1 << bits) * blocks; }

hipError_t launch_spill_count(const AggSpill &sp, int n_segments, int bits, int blocks, int32_t *cells, unsigned long long *total,
                              hipStream_t stream) {
  if (bits > PART_MAX_BITS) return hipErrorInvalidValue;
  hipLaunchKernelGGL(qe_spill_count_kernel, dim3(blocks), dim3(PART_THREADS), 0, stream, sp, n_segments, bits, cells, total);
  return hipGetLastError();
}

// How the partitioning is split into passes: bits of the first level (the second takes the rest; 0 = one pass).
int spill_level1_bits(int bits) {
  static int levels = tuned("QE_SPILL_LEVELS", 2);  // 1: one scatter pass whatever the number of partitions (an experiment)
  if (levels == 1) return bits;
  return bits <= 6 ? bits : bits - bits / 2;
}

hipError_t launch_spill_cells_level1(const int32_t *cells, int32_t *cells1, int bits, int blocks, hipStream_t stream) {
  const int bits1 = spill_level1_bits(bits);
  const int parts1 = 1 << bits1, fan = 1 << (bits - bits1);
  const int n = parts1 * blocks;
  hipLaunchKernelGGL(qe_spill_cells_level1_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, cells, cells1, parts1, fan, blocks);
  return hipGetLastError();
}

// One scatter pass.  level 1: the log -> dst by the top `bits` bits (dst_cells = scanned cells of that depth).
// level 2: first-level cells of src (src_cells scanned, parts1 of them per workgroup) -> dst by `bits` bits, of which
// the low bits - bits1 select among the partitions below one first-level partition.
hipError_t launch_spill_scatter(const AggSpill &sp, int n_segments, int level, int bits, int bits1, int blocks, const unsigned long long *src_keys,
                                const unsigned long long *src_vals, uint64_t src_stride, const int32_t *src_cells, int64_t total,
                                const int32_t *dst_cells, unsigned long long *dst_keys, unsigned long long *dst_vals, uint64_t dst_stride,
                                hipStream_t stream) {
  if (sp.n_inputs > (uint32_t)SCATTER_MAX_INPUTS) return hipErrorInvalidValue;
  ScatterArgs a{};
  a.sp = sp;
  a.n_segments = n_segments;
  a.level = level;
  a.src_keys = src_keys;
  a.src_vals = src_vals;
  a.src_stride = src_stride;
  a.src_cells = src_cells;
  a.parts1 = 1 << bits1;
  a.total = total;
  a.bits = bits;
  a.stages = level == 1 ? 1 << bits : 1 << (bits - bits1);
  if (a.stages > SCATTER_MAX_STAGES) return hipErrorInvalidValue;
  const size_t row_bytes = 8 * (1 + (size_t)sp.n_inputs);
  static int stage_kb = tuned("QE_SCATTER_LDS_KB", 144);
  size_t depth = (size_t)stage_kb * 1024 / ((size_t)a.stages * row_bytes);
  if (depth > 1024) depth = 1024;
  if (depth < 8) return hipErrorInvalidValue;
  a.depth = (int)depth;
  a.dst_cells = dst_cells;
  a.dst_keys = dst_keys;
  a.dst_vals = dst_vals;
  a.dst_stride = dst_stride;
  a.n_inputs = sp.n_inputs;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)qe_spill_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL(qe_spill_scatter_kernel, dim3(blocks), dim3(PART_THREADS), (size_t)a.stages * a.depth * row_bytes, stream, a);
  return hipGetLastError();
}

bool spill_onepass_available(int bits, uint32_t n_inputs) {
  static int on = tuned("QE_SPILL_ONEPASS", 1);  // 2: off (the two-level scatter of whole-KiB bursts instead)
  return on != 2 && bits <= 9 && (1 << bits) <= ONEPASS_MAX_PARTS && n_inputs <= 1;
}

hipError_t launch_spill_scatter_onepass(const AggSpill &sp, int n_segments, int bits, int blocks, const int32_t *dst_cells, int cells_stride, void *dst_rows,
                                        hipStream_t stream) {
  if (!spill_onepass_available(bits, sp.n_inputs)) return hipErrorInvalidValue;
  OnePassArgs a{};
  a.sp = sp;
  a.n_segments = n_segments;
  a.bits = bits;
  a.dst_cells = dst_cells;
  a.dst = (KeyVal *)dst_rows;
  a.n_inputs = sp.n_inputs;
  a.cells_stride = (uint32_t)(cells_stride > 0 ? cells_stride : blocks);
  const size_t P = (size_t)1 << bits;
  const size_t lds = P * ONEPASS_RING_STRIDE * sizeof(KeyVal) + P * (2 * sizeof(unsigned long long) + sizeof(uint32_t)) + 2 * P * sizeof(uint32_t) + 64;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)qe_spill_scatter_onepass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(qe_spill_scatter_onepass_kernel, dim3(blocks), dim3(PART_THREADS), lds, stream, a);
  return hipGetLastError();
}

// LDS slots of the partition-wise aggregate: as many as 144 KiB hold (one workgroup per compute unit).
uint32_t partitioned_lds_slots(int n_accs) {
  uint32_t slots = 8192;
  while (slots > 64 && agg_lds_bytes(n_accs, slots, false) > (size_t)148 * 1024) slots >>= 1;
  return slots;
}

hipError_t launch_aggregate_partitioned(const PartArgs &args, const DeviceInfo &dev, hipStream_t stream) {
  size_t lds = agg_lds_bytes(args.n_accs, args.lds_cap, false);
  static bool attr_set = false;
  if (!attr_set) {
    // (the table takes at most 148 KiB - partitioned_lds_slots - and the kernel's static arrays 8 KiB of the unit's 160)
    hipError_t e = hipFuncSetAttribute((const void *)qe_aggregate_partitioned_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  int grid = args.n_parts < dev.cus ? args.n_parts : dev.cus;
#ifdef QE_ABLATE  // timing experiments only (results are wrong): compiled in by a build with -DQE_ABLATE, never in the shipped library
  static int ablate = tuned("QE_PART_ABLATE", 0);
#else
  const int ablate = 0;
#endif
  PartArgs a2 = args;
  a2.ablate = ablate;
  hipLaunchKernelGGL(qe_aggregate_partitioned_kernel, dim3(grid), dim3(PART_THREADS), lds, stream, a2);
  return hipGetLastError();
}

hipError_t launch_table_init(const AggTable &table, const VmAcc *accs, int n_accs, hipStream_t stream, void *zero, size_t zero_bytes) {
  static_assert(VM_MAX_ACCS <= 16, "two packed words hold 16 accumulator ops");
  uint64_t packed[2] = {0, 0};
  for (int i = 0; i < n_accs && i < VM_MAX_ACCS; ++i) packed[i >> 3] |= (uint64_t)accs[i].op << (8 * (i & 7));
  int grid = grid_for((int64_t)table.stride, 256, 4096);
  hipLaunchKernelGGL(qe_table_init_kernel, dim3(grid), dim3(256), 0, stream, table, packed[0], packed[1], n_accs, (unsigned long long *)zero,
                     (uint64_t)(zero_bytes / 8));
  return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeArgs &args, hipStream_t stream) {
  if (args.table.cap + 1 <= (1u << 16)) {
    hipLaunchKernelGGL(qe_finalize_small_kernel, dim3(grid_for((int64_t)args.table.cap + 1, 256, 4096)), dim3(256), 0, stream, args);
    return hipGetLastError();
  }
  int grid = grid_for((int64_t)args.table.cap + 1, 256 * FIN_ITEMS, 4096);
  hipLaunchKernelGGL(qe_finalize_kernel, dim3(grid), dim3(256), 0, stream, args);
  return hipGetLastError();
}

size_t utf8_gather_offsets_status_bytes(int64_t m) { return (size_t)((m + OFFSETS_TILE_MIN - 1) / OFFSETS_TILE_MIN) * sizeof(unsigned long long); }

hipError_t launch_utf8_gather_offsets(const Utf8GatherCols &cols, const int32_t *idx, int64_t m, const unsigned long long *m_dev, int cus, hipStream_t stream) {
  if (m <= 0 || cols.n <= 0) return hipSuccess;
  if (cols.n > UTF8_GATHER_MAX_COLS || ((uintptr_t)idx & 15)) return hipErrorInvalidValue;
  for (int i = 0; i < cols.n; ++i)
    if ((uintptr_t)cols.col[i].out_offsets & 15) return hipErrorInvalidValue;
  constexpr int R = 4;
  const int64_t tile = (int64_t)OFFSETS_THREADS * R, n_tiles = (m + tile - 1) / tile;
  const int64_t tickets = n_tiles * cols.n, resident = (int64_t)cus * 2;
  const int grid = (int)(tickets < resident ? tickets : resident);
  hipLaunchKernelGGL(qe_utf8_gather_offsets_kernel<R>, dim3(grid), dim3(OFFSETS_THREADS), 0, stream, cols, idx, m, m_dev);
  return hipGetLastError();
}

size_t scan_scratch_elems(int64_t n) {
  // block sums of every level, laid end to end
  size_t total = 0;
  int64_t m = n;
  while (m > SCAN_TILE) {
    m = (m + SCAN_TILE - 1) / SCAN_TILE;
    total += (size_t)m;
  }
  return total + 1;
}

hipError_t launch_exclusive_scan_i32(int32_t *data, int64_t n, int32_t *block_sums, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (blocks == 1) {
    hipLaunchKernelGGL(qe_scan_tiles_kernel, dim3(1), dim3(SCAN_BLOCK), 0, stream, data, n, (int32_t *)nullptr);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(qe_scan_tiles_kernel, dim3((unsigned)blocks), dim3(SCAN_BLOCK), 0, stream, data, n, block_sums);
  hipError_t e = launch_exclusive_scan_i32(block_sums, blocks, block_sums + blocks, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(qe_scan_add_kernel, dim3((unsigned)blocks), dim3(SCAN_BLOCK), 0, stream, data, n, block_sums);
  return hipGetLastError();
}

bool utf8_gather_fused_available() {
  static int on = tuned("QE_UTF8_GATHER_FUSED", 1);  // 2: off (offsets and bytes in a launch each)
  return on != 2;
}

hipError_t launch_utf8_gather_fused(const Utf8GatherCols &cols, const int32_t *idx, int64_t m, const unsigned long long *m_dev, int cus, hipStream_t stream) {
  if (cols.n <= 0 || cols.n > UTF8_GATHER_MAX_COLS || ((uintptr_t)idx & 15)) return hipErrorInvalidValue;
  if (m <= 0) return hipSuccess;
  for (int i = 0; i < cols.n; ++i)
    if (((uintptr_t)cols.col[i].out_offsets & 15) || cols.col[i].out_bytes == nullptr) return hipErrorInvalidValue;
  static_assert(FUSED_THREADS * FUSED_R >= OFFSETS_TILE_MIN, "the status array is sized for tiles of OFFSETS_TILE_MIN rows");
  const int64_t tiles = (m + FUSED_THREADS * FUSED_R - 1) / (FUSED_THREADS * FUSED_R) * cols.n;
  static int per_cu = tuned("QE_UTF8_GATHER_FUSED_BLOCKS_PER_CU", 3);
  const int64_t most = (int64_t)cus * per_cu;
  const size_t lds = FUSED_WAVE_LDS * (FUSED_THREADS / 64);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)qe_utf8_gather_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(qe_utf8_gather_fused_kernel, dim3((unsigned)(tiles < most ? tiles : most)), dim3(FUSED_THREADS), lds, stream, cols, idx, m, m_dev);
  return hipGetLastError();
}

hipError_t launch_utf8_gather_bytes(const Utf8GatherCols &cols, const int32_t *idx, int64_t m, const unsigned long long *m_dev, hipStream_t stream) {
  if (m <= 0 || cols.n <= 0) return hipSuccess;
  if (cols.n > UTF8_GATHER_MAX_COLS) return hipErrorInvalidValue;
  hipLaunchKernelGGL(qe_utf8_gather_bytes_kernel, dim3(grid_for(m, 256, 8192)), dim3(GATHER_WAVES * 64), 0, stream, cols, idx, m, m_dev);
  return hipGetLastError();
}

hipError_t launch_utf8_gather_rows(const Utf8Source *sources, int n_sources, const int64_t *ids, int64_t m, int32_t *out_lengths,
                                   const int32_t *dst_offsets, uint8_t *dst_bytes, hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_utf8_gather_rows_kernel, dim3(grid_for(m, 256, 4096)), dim3(256), 0, stream, sources, n_sources, ids, m, out_lengths,
                     dst_offsets, dst_bytes);
  return hipGetLastError();
}

hipError_t launch_gather_rows_fixed(const FixedSource *sources, int n_sources, const int64_t *ids, int64_t m, int width, void *out,
                                    hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_gather_rows_fixed_kernel, dim3(grid_for(m, 256, 4096)), dim3(256), 0, stream, sources, n_sources, ids, m, width, out);
  return hipGetLastError();
}

hipError_t launch_verify_group_keys(const VerifyArgs &args, hipStream_t stream) {
  if (args.m <= 0) return hipSuccess;
  if (args.n_keys < 1 || args.n_keys > VERIFY_MAX_KEYS || args.n_sources < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(qe_verify_group_keys_kernel, dim3(grid_for(args.m, 256, 4096)), dim3(256), 0, stream, args);
  return hipGetLastError();
}

hipError_t launch_utf8_broadcast(const uint8_t *lit, int32_t len, int64_t n, int32_t *out_offsets, uint8_t *out_bytes,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(qe_utf8_broadcast_kernel, dim3(grid_for(n + 1, 256, 4096)), dim3(256), 0, stream, lit, len, n,
                     out_offsets, out_bytes);
  return hipGetLastError();
}

hipError_t launch_count_zero_bits(const uint32_t *bitmap, int64_t n, unsigned long long *out, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_count_zero_bits_kernel, dim3(grid_for((n + 31) / 32, 256, 1024)), dim3(256), 0, stream, bitmap,
                     n, out);
  return hipGetLastError();
}

hipError_t launch_generate(int kind, uint64_t seed, int64_t row0, int64_t n, double lo, double hi, void *out,
                           hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(qe_generate_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, stream, kind, seed, row0, n, lo,
                     hi, out);
  return hipGetLastError();
}

}  // namespace qe
