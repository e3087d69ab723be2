// Write-burst probe: how fast does MI355X take writes that come as bursts of B contiguous bytes at pseudo-random places
// (what a partitioning pass produces: one burst per flushed stage)?  B = 64 .. 4096 bytes, 8-byte and 16-byte stores per
// lane, bursts aligned to B or shifted by 8 bytes.  Prints GB/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

struct alignas(16) U2 { unsigned long long x, y; };

// W = bytes per lane and store (8 or 16)
template <int W>
__global__ __launch_bounds__(256) void bursts(char *buf, unsigned long long slots, int B, int skew, long n_bursts) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long)gridDim.x * blockDim.x) >> 6;
  const int per_store = 64 * W;                       // bytes one store instruction of the wave covers
  const int groups = B >= per_store ? 1 : per_store / B;  // bursts per store instruction
  const int lanes = 64 / groups;                      // lanes per burst
  const int reps = B >= per_store ? B / per_store : 1;  // store instructions per burst
  for (long b0 = wave * groups; b0 < n_bursts; b0 += n_waves * groups) {
    const long b = b0 + lane / lanes;
    if (b >= n_bursts) continue;
    char *at = buf + (mix((unsigned long long)b) % slots) * (unsigned long long)B + skew + (lane % lanes) * W;
    for (int r = 0; r < reps; ++r) {
      if constexpr (W == 8) *(unsigned long long *)(at + (long)r * per_store) = (unsigned long long)b;
      else *(U2 *)(at + (long)r * per_store) = U2{(unsigned long long)b, (unsigned long long)lane};
    }
  }
}

int main(int argc, char **argv) {
  const unsigned long long buf_bytes = 8ull << 30, total = argc > 1 ? strtoull(argv[1], nullptr, 10) : (4ull << 30);
  char *buf;
  CK(hipMalloc(&buf, buf_bytes + 65536));
  CK(hipMemset(buf, 0, buf_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int skew = 0; skew <= 8; skew += 8)
    for (int W = 8; W <= 16; W += 8)
      for (int B = 64; B <= 4096; B *= 2) {
        if (W == 16 && skew) continue;  // (16-byte stores want 16-byte alignment)
        const long n_bursts = (long)(total / B);
        const unsigned long long slots = buf_bytes / B;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          if (W == 8) hipLaunchKernelGGL(bursts<8>, dim3(256 * 8), dim3(256), 0, 0, buf, slots, B, skew, n_bursts);
          else hipLaunchKernelGGL(bursts<16>, dim3(256 * 8), dim3(256), 0, 0, buf, slots, B, skew, n_bursts);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf("burst %5d B  %2d B/lane  skew %d: %8.3f ms  %7.1f GB/s\n", B, W, skew, best, (double)total / best / 1e6);
      }
  return 0;
}
