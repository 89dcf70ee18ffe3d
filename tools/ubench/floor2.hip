// Un-profiled per-kernel cost of a dependent chain on one stream: hipGraph replay vs direct launches, empty vs streaming kernels.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_empty(float* out) { if (out == nullptr) out[0] = 1.f; }
template <int N>
__global__ __launch_bounds__(256) void k_stream(const uint4* in, float* out) {
  const uint4* p = in + ((size_t)blockIdx.x * N * 256) + threadIdx.x;
  uint4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 t = __builtin_nontemporal_load((const u4*)(p + i * 256)); v[i] = make_uint4(t.x, t.y, t.z, t.w); }
  uint32_t a = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) a ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (a == 0x12345678u) out[threadIdx.x] = 1.f;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t big = 1ull << 30;
  char* buf; float* out;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&out, 1 << 22)); CK(hipMemset(buf, 1, big));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int N = 2000;
  for (int variant = 0; variant < 4; ++variant) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < N; ++r) {
      const size_t off = ((size_t)r * 20u << 20) % (big - (64u << 20));
      if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, out);
      if (variant == 1) hipLaunchKernelGGL(k_empty, dim3(56), dim3(256), 0, st, out);
      if (variant == 2) hipLaunchKernelGGL((k_stream<14>), dim3(304), dim3(256), 0, st, (const uint4*)(buf + off), out);
      if (variant == 3) hipLaunchKernelGGL((k_stream<7>), dim3(56), dim3(256), 0, st, (const uint4*)(buf + off), out);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    double best = 1e9;
    for (int it = 0; it < 5; ++it) {
      const double t0 = now();
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      best = std::min(best, now() - t0);
    }
    printf("graph  variant %d: %.3f us per kernel\n", variant, best / N * 1e6);
    best = 1e9;
    for (int it = 0; it < 5; ++it) {
      const double t0 = now();
      for (int r = 0; r < N; ++r) {
        const size_t off = ((size_t)r * 20u << 20) % (big - (64u << 20));
        if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, out);
        if (variant == 1) hipLaunchKernelGGL(k_empty, dim3(56), dim3(256), 0, st, out);
        if (variant == 2) hipLaunchKernelGGL((k_stream<14>), dim3(304), dim3(256), 0, st, (const uint4*)(buf + off), out);
        if (variant == 3) hipLaunchKernelGGL((k_stream<7>), dim3(56), dim3(256), 0, st, (const uint4*)(buf + off), out);
      }
      const double t1 = now();
      CK(hipStreamSynchronize(st));
      best = std::min(best, now() - t0);
      if (it == 4) printf("direct variant %d: host enqueue %.3f us per kernel\n", variant, (t1 - t0) / N * 1e6);
    }
    printf("direct variant %d: %.3f us per kernel\n", variant, best / N * 1e6);
  }
  return 0;
}
