// L2-resident read bandwidth per CU for the access shapes of the GEMM loader.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void rd(const uint4* src, uint4* out, int iters, int rows, int row_stride16, int mode) {
  // mode 0: each wave reads 1 KiB contiguous per load; mode 1: 8 rows x 128 B (row stride given) per wave load
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int wg = blockIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int idx;
      if (mode == 0) idx = ((wg * 131 + it * 8 + u) % rows) * row_stride16 + (wid * 64 + lane) % row_stride16;
      else { const int r = ((wg * 64 + (it * 8 + u) * 8 + wid * 8) + (lane >> 3)) % rows; idx = r * row_stride16 + (lane & 7) + ((it & 3) * 8) % row_stride16; }
      const uint4 v = src[idx];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  if (acc.x == 0x12345678) out[tid] = acc;
}
int main() {
  const int rows = 16000, stride16 = 32;  // 512 B rows (K=256 fp16) -> 8 MB
  uint4 *src, *out;
  hipMalloc(&src, (size_t)rows * stride16 * 16); hipMalloc(&out, 4096);
  hipMemset(src, 1, (size_t)rows * stride16 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int wgs : {256, 1024, 2048}) {
      const int iters = 64;
      rd<<<wgs, 256>>>(src, out, iters, rows, stride16, mode);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) rd<<<wgs, 256>>>(src, out, iters, rows, stride16, mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double bytes = (double)wgs * 256 * iters * 8 * 16;
      printf("mode %d wgs %4d: %.1f us, %.2f TB/s, %.1f GB/s per CU (%.1f B/clk @2.4GHz)\n", mode, wgs, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
    }
  return 0;
}
