// Ceiling probe: how fast can this access pattern (16-byte loads, row pair per lane, grid-stride over
// 256-row wave-steps) stream N columns on MI355X?  Variants: blocks per CU, plain / nontemporal loads,
// one or two wave-steps in flight per wave.  Prints GB/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct alignas(16) U2 { unsigned long long x, y; };

template <bool NT>
__device__ __forceinline__ U2 ld(const U2 *p) {
  if constexpr (NT) {
    U2 r;
    r.x = __builtin_nontemporal_load(&p->x);
    r.y = __builtin_nontemporal_load(&p->y);
    return r;
  } else {
    return *p;
  }
}

// reads NC columns, folds them into one value per lane (kept live through a final atomic)
template <int NC, bool NT, int DEPTH>
__global__ __launch_bounds__(256) void read_cols(const unsigned long long *const *cols_in, long n, unsigned long long *out) {
  const unsigned long long *cols[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) cols[c] = cols_in[c];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nsteps = n / 256;
  unsigned long long acc = 0;
  for (long ws = ((long)blockIdx.x * 4 + wave) * DEPTH; ws < nsteps; ws += (long)gridDim.x * 4 * DEPTH) {
    U2 v[DEPTH][NC][2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      long r0 = (ws + d) * 256 + 2 * lane;
      if (ws + d < nsteps) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          v[d][c][0] = ld<NT>((const U2 *)(cols[c] + r0));
          v[d][c][1] = ld<NT>((const U2 *)(cols[c] + r0 + 128));
        }
      } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) v[d][c][0] = v[d][c][1] = U2{0, 0};
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc += v[d][c][0].x ^ v[d][c][0].y ^ v[d][c][1].x ^ v[d][c][1].y;
  }
  if (acc == 0x1234567ull) atomicAdd(out, acc);
}

// out = a + b*c (24 B read + 8 B written per row)
template <bool NT>
__global__ __launch_bounds__(256) void triad(const double *a, const double *b, const double *c, double *o, long n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nsteps = n / 256;
  for (long ws = (long)blockIdx.x * 4 + wave; ws < nsteps; ws += (long)gridDim.x * 4) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      long r = ws * 256 + h * 128 + 2 * lane;
      U2 x = ld<NT>((const U2 *)(a + r)), y = ld<NT>((const U2 *)(b + r)), z = ld<NT>((const U2 *)(c + r));
      double r0 = __longlong_as_double(x.x) + __longlong_as_double(y.x) * __longlong_as_double(z.x);
      double r1 = __longlong_as_double(x.y) + __longlong_as_double(y.y) * __longlong_as_double(z.y);
      if constexpr (NT) {
        __builtin_nontemporal_store(r0, o + r);
        __builtin_nontemporal_store(r1, o + r + 1);
      } else {
        U2 w;
        w.x = __double_as_longlong(r0);
        w.y = __double_as_longlong(r1);
        *(U2 *)(o + r) = w;
      }
    }
  }
}

template <class F>
float time_ms(F &&f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv) {
  long n = argc > 1 ? atol(argv[1]) : 256000000L;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("device %s cus %d rows %ld\n", p.gcnArchName, cus, n);
  unsigned long long *col[4], *out;
  for (int i = 0; i < 4; ++i) {
    CK(hipMalloc(&col[i], n * 8 + 64));
    CK(hipMemset(col[i], i + 1, n * 8));
  }
  CK(hipMalloc(&out, 64));
  const unsigned long long **dcols;
  CK(hipMalloc(&dcols, 4 * sizeof(void *)));
  CK(hipMemcpy(dcols, col, 4 * sizeof(void *), hipMemcpyHostToDevice));
  for (int bpc : {2, 4, 8, 16, 32}) {
    int grid = cus * bpc;
#define RUN(NC, NT, D)                                                                                      \
  {                                                                                                         \
    float ms = time_ms([&] { hipLaunchKernelGGL((read_cols<NC, NT, D>), dim3(grid), dim3(256), 0, 0, dcols, n, out); }, 5); \
    printf("read  cols=%d nt=%d depth=%d blocks/cu=%2d  %.3f ms  %.0f GB/s\n", NC, NT, D, bpc, ms, NC * 8.0 * n / ms / 1e6);  \
  }
    RUN(4, false, 1) RUN(4, true, 1) RUN(4, false, 2) RUN(2, false, 1) RUN(2, true, 1) RUN(2, false, 2) RUN(2, false, 4)
    {
      float ms = time_ms([&] { hipLaunchKernelGGL(triad<false>, dim3(grid), dim3(256), 0, 0, (double *)col[0], (double *)col[1], (double *)col[2], (double *)col[3], n); }, 5);
      printf("triad nt=0 blocks/cu=%2d  %.3f ms  %.0f GB/s\n", bpc, ms, 32.0 * n / ms / 1e6);
      ms = time_ms([&] { hipLaunchKernelGGL(triad<true>, dim3(grid), dim3(256), 0, 0, (double *)col[0], (double *)col[1], (double *)col[2], (double *)col[3], n); }, 5);
      printf("triad nt=1 blocks/cu=%2d  %.3f ms  %.0f GB/s\n", bpc, ms, 32.0 * n / ms / 1e6);
    }
  }
  return 0;
}
