// Design-space probe for the fused filter+project kernel (c > 0.5 -> a + b*c, 28 B/row at 50 %):
// which part of a tile's life costs the time?  Variants switch off the look-back or the stores, change
// the tile size, the look-back width and the number of resident workgroups.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../query-engine_amd/csrc/qe_device.h"

using namespace qe;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Scratch { unsigned long long ticket, total, error, pad[5]; };

template <int LBW>
__device__ unsigned long long lookback_w(unsigned long long *status, int64_t tile, unsigned long long total, uint32_t lane) {
  if (tile == 0) {
    if (lane == 0) st_status(&status[0], ST_FLAG_PREFIX | total);
    return 0;
  }
  if (lane == 0) st_status(&status[tile], ST_FLAG_AGG | total);
  unsigned long long excl = 0;
  int64_t idx = tile - 1;
  for (int spins = 0; spins < (1 << 22);) {
    unsigned long long v[LBW];
#pragma unroll
    for (int k = 0; k < LBW; ++k) {
      int64_t t = idx - ((int64_t)lane * LBW + k);
      v[k] = (t >= 0) ? ld_status(&status[t]) : ST_FLAG_PREFIX;
    }
    unsigned long long sum_all = 0, sum_to_prefix = 0;
    bool has_prefix = false, inv_before = false, any_inv = false;
#pragma unroll
    for (int k = 0; k < LBW; ++k) {
      uint32_t flag = (uint32_t)(v[k] >> 62);
      unsigned long long val = v[k] & ST_VALUE_MASK;
      if (flag == 0u) { any_inv = true; if (!has_prefix) inv_before = true; }
      sum_all += val;
      if (!has_prefix) sum_to_prefix += val;
      if (flag == 2u) has_prefix = true;
    }
    uint64_t pre = __ballot(has_prefix);
    if (pre != 0) {
      int p = __builtin_ctzll(pre);
      bool blocked = ((int)lane < p && any_inv) || ((int)lane == p && inv_before);
      if (__ballot(blocked) == 0) {
        excl += wave_sum_u64((int)lane < p ? sum_all : ((int)lane == p ? sum_to_prefix : 0ull));
        break;
      }
    } else if (__ballot(any_inv) == 0) {
      excl += wave_sum_u64(sum_all);
      idx -= 64 * LBW;
      continue;
    }
    ++spins;
    __builtin_amdgcn_s_sleep(1);
  }
  if (lane == 0) st_status(&status[tile], ST_FLAG_PREFIX | ((excl + total) & ST_VALUE_MASK));
  return excl;
}

// MODE bit0: look-back on; bit1: stores on; bit2: late loads of a, b only for selected rows (else with c)
template <int STEPS, int LBW, int MODE, int MINB, int WAVES>
__global__ __launch_bounds__(WAVES * 64, MINB) void filt(const double *a, const double *b, const double *c, double *out, int64_t n,
                                                   int64_t n_tiles, Scratch *sc) {
  __shared__ unsigned long long s_tile, s_base;
  __shared__ uint32_t s_cnt[STEPS * WAVES];
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  unsigned long long *status = (unsigned long long *)(sc + 1);
  constexpr int TILE = STEPS * WAVES * 256;
  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(&sc->ticket, 1ull);
    __syncthreads();
    const int64_t tile = (int64_t)s_tile;
    if (tile >= n_tiles) break;
    U64x2 vc[STEPS][2], va[STEPS][2], vb[STEPS][2];
    uint32_t sel[STEPS], act[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      int64_t row0 = tile * TILE + s * (WAVES * 256) + wave * 256;
      act[s] = in_range_bits(row0, lane, n);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t r = row0 + h * 128 + 2 * lane;
        vc[s][h] = U64x2{0, 0};
        va[s][h] = U64x2{0, 0};
        vb[s][h] = U64x2{0, 0};
        if ((act[s] >> (2 * h)) & 3u) {
          vc[s][h] = stream_load_pair64((const uint64_t *)c + r);
          if (!(MODE & 4)) {
            va[s][h] = stream_load_pair64((const uint64_t *)a + r);
            vb[s][h] = stream_load_pair64((const uint64_t *)b + r);
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      int64_t row0 = tile * TILE + s * (WAVES * 256) + wave * 256;
      uint32_t m = 0;
      m |= (u2d(vc[s][0].x) > 0.5) ? 1u : 0u;
      m |= (u2d(vc[s][0].y) > 0.5) ? 2u : 0u;
      m |= (u2d(vc[s][1].x) > 0.5) ? 4u : 0u;
      m |= (u2d(vc[s][1].y) > 0.5) ? 8u : 0u;
      sel[s] = m & act[s];
      uint32_t cnt = popc_sel(sel[s]);
      if (lane == 0) s_cnt[s * WAVES + wave] = cnt;
      if (MODE & 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int64_t r = row0 + h * 128 + 2 * lane;
          if ((sel[s] >> (2 * h)) & 3u) {
            va[s][h] = stream_load_pair64((const uint64_t *)a + r);
            vb[s][h] = stream_load_pair64((const uint64_t *)b + r);
          }
        }
      }
    }
    __syncthreads();
    uint32_t pref[STEPS];
    uint32_t run = 0;
#pragma unroll
    for (int i = 0; i < STEPS * WAVES; ++i) {
      if ((uint32_t)(i % WAVES) == wave) pref[i / WAVES] = run;
      run += s_cnt[i];
    }
    unsigned long long base;
    if (MODE & 1) {
      if (wave == 0) {
        unsigned long long excl = lookback_w<LBW>(status, tile, run, lane);
        if (lane == 0) {
          s_base = excl;
          if (tile == n_tiles - 1) sc->total = excl + run;
        }
      }
      __syncthreads();
      base = s_base;
    } else {
      base = (unsigned long long)tile * (TILE / 2);
      if (run > TILE / 2 + 512) base = 0;  // keep `run` live
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      uint32_t rank[4];
      rank_rows(sel[s], rank);
      double r[4];
      r[0] = u2d(va[s][0].x) + u2d(vb[s][0].x) * u2d(vc[s][0].x);
      r[1] = u2d(va[s][0].y) + u2d(vb[s][0].y) * u2d(vc[s][0].y);
      r[2] = u2d(va[s][1].x) + u2d(vb[s][1].x) * u2d(vc[s][1].x);
      r[3] = u2d(va[s][1].y) + u2d(vb[s][1].y) * u2d(vc[s][1].y);
      if (MODE & 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((sel[s] >> j) & 1u) out[base + pref[s] + rank[j]] = r[j];
      } else {
        if (r[0] + r[1] + r[2] + r[3] == 12345.678) out[0] = 1.0;
      }
    }
  }
}

int main(int argc, char **argv) {
  int64_t n = argc > 1 ? atol(argv[1]) : 256000000L;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  double *a, *b, *c, *out;
  CK(hipMalloc(&a, n * 8 + 64));
  CK(hipMalloc(&b, n * 8 + 64));
  CK(hipMalloc(&c, n * 8 + 64));
  CK(hipMalloc(&out, n * 8 + 64));
  {
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0;
    for (int64_t o = 0; o < n; o += (int64_t)h.size()) {
      size_t m = (size_t)std::min<int64_t>(h.size(), n - o);
      CK(hipMemcpy(c + o, h.data(), m * 8, hipMemcpyHostToDevice));
    }
    CK(hipMemset(a, 0, n * 8));
    CK(hipMemset(b, 0, n * 8));
  }
  Scratch *sc;
  size_t sc_bytes = sizeof(Scratch) + (n / 1024 + 2) * 8;
  CK(hipMalloc(&sc, sc_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
#define RUN(STEPS, LBW, MODE, MINB, BPC, WAVES)                                                                             \
  {                                                                                                                  \
    int64_t nt = (n + STEPS * WAVES * 256 - 1) / (STEPS * WAVES * 256);                                                            \
    float best = 1e9;                                                                                                \
    unsigned long long total = 0;                                                                                    \
    for (int rep = 0; rep < 4; ++rep) {                                                                              \
      CK(hipMemsetAsync(sc, 0, sc_bytes, 0));                                                                        \
      CK(hipEventRecord(e0));                                                                                        \
      hipLaunchKernelGGL((filt<STEPS, LBW, MODE, MINB, WAVES>), dim3(cus * BPC), dim3(WAVES * 64), 0, 0, a, b, c, out, n, nt, sc); \
      CK(hipEventRecord(e1));                                                                                        \
      CK(hipEventSynchronize(e1));                                                                                   \
      float ms;                                                                                                      \
      CK(hipEventElapsedTime(&ms, e0, e1));                                                                          \
      if (rep && ms < best) best = ms;                                                                               \
    }                                                                                                                \
    CK(hipMemcpy(&total, &sc->total, 8, hipMemcpyDeviceToHost));                                                     \
    printf("waves=%d steps=%d lbw=%d lookback=%d stores=%d late=%d minb=%d blocks/cu=%d : %.3f ms %.1f Grows/s (selected %llu)\n", WAVES, STEPS, LBW, \
           MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, MINB, BPC, best, n / best / 1e6, total);                    \
  }
  RUN(4, 1, 7, 2, 1, 4)
  RUN(4, 1, 7, 2, 2, 4)
  RUN(4, 1, 7, 2, 3, 4)
  RUN(6, 1, 7, 1, 1, 4)
  RUN(6, 1, 7, 1, 2, 4)
  RUN(8, 1, 7, 1, 1, 4)
  RUN(4, 1, 7, 1, 1, 8)
  RUN(6, 1, 7, 1, 1, 8)
  RUN(8, 1, 7, 1, 1, 8)
  RUN(2, 1, 7, 1, 1, 16)
  RUN(3, 1, 7, 1, 1, 16)
  RUN(4, 1, 7, 1, 1, 16)
  RUN(8, 1, 6, 1, 1, 8)
  RUN(4, 1, 6, 1, 1, 16)
  return 0;
}
