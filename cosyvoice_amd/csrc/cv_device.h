// Device-side helpers shared by the gfx950 kernels (wave64, MFMA 16x16, 16-byte LDS chunks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/cosyvoice_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define CV_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return CV_ERR_LAUNCH;           \
  } while (0)

// Host side: a per-function attribute such as the > 64 KiB dynamic-LDS opt-in (hipFuncSetAttribute) applies to the CURRENT device.
// Remembered per (call site, device); whichever thread gets to a new device first sets it (setting it twice is harmless).
struct PerDeviceOnce {
  std::atomic<bool> done[64];
  PerDeviceOnce() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
  template <typename F> void run(F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { f(); return; }
    if (!done[dev].load(std::memory_order_acquire)) { f(); done[dev].store(true, std::memory_order_release); }
  }
};

template <typename To, typename From>
__device__ __forceinline__ To bitcast(const From& f) {
  static_assert(sizeof(To) == sizeof(From), "size");
  To t;
  __builtin_memcpy(&t, &f, sizeof(To));
  return t;
}

// ---- 16-bit conversions (RNE; hipcc lowers the casts to v_cvt_pk_bf16_f32 / v_cvt_f16_f32) ----
template <int DT> struct Elem16;
template <> struct Elem16<CV_BF16> {
  static __device__ __forceinline__ uint16_t from_f32(float f) { return bitcast<uint16_t>((__bf16)f); }
  static __device__ __forceinline__ float to_f32(uint16_t u) { return bitcast<float>((uint32_t)u << 16); }
};
template <> struct Elem16<CV_F16> {
  static __device__ __forceinline__ uint16_t from_f32(float f) { return bitcast<uint16_t>((_Float16)f); }
  static __device__ __forceinline__ float to_f32(uint16_t u) { return (float)bitcast<_Float16>(u); }
};

// two fp32 -> one packed 16-bit pair, RNE: ONE instruction on gfx950 (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32); the scalar
// casts compile to cvt + cvt_sdwa + or for fp16.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <int DT>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2_t v = f32x2_t{a, b};
  if constexpr (DT == CV_BF16) return bitcast<uint32_t>(__builtin_convertvector(v, bf16x2_t));
  else return bitcast<uint32_t>(__builtin_convertvector(v, f16x2_t));
}

// One "block" MFMA step over the K extent of four 16-byte chunks (32 k for 16-bit, 16 k for f32).
// a: fragment of the operand whose rows become D rows; b: fragment whose rows become D columns.
// Lane l holds row (l&15), chunk (l>>4) of each.  D: col = l&15, row = 4*(l>>4) + reg.
template <int DT>
__device__ __forceinline__ f32x4_t mfma_block(const uint4& a, const uint4& b, f32x4_t c) {
  if constexpr (DT == CV_BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(bitcast<bf16x8_t>(a), bitcast<bf16x8_t>(b), c, 0, 0, 0);
  } else if constexpr (DT == CV_F16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8_t>(a), bitcast<f16x8_t>(b), c, 0, 0, 0);
  } else {
    // exact-f32 MFMA (v_mfma_f32_16x16x4_f32); element j of every lane's float4 forms k-step j
    const float4 fa = bitcast<float4>(a), fb = bitcast<float4>(b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, c, 0, 0, 0);
    return c;
  }
}

// ---- activations (fp32) ----
__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float act_mish(float x) {
  // x * tanh(softplus(x)); softplus threshold 20 as torch
  float sp = x > 20.0f ? x : log1pf(__expf(x));
  return x * tanhf(sp);
}
__device__ __forceinline__ float act_elu(float x) { return x > 0.0f ? x : expm1f(x); }
// sin(y) to ~2e-7 absolute for |y| < 1e3 (Snake arguments are O(10)): Cody-Waite reduction by pi in two terms to [-pi/2, pi/2],
// odd Taylor polynomial to r^13 (error 6e-8 at the interval ends), sign from the parity of the multiple — 16 straight-line
// instructions against ocml sinf's ~60 with a slow-path branch.  1.4 G Snake evaluations per batch-8 HiFT pass sit in GEMM epilogues.
__device__ __forceinline__ float sin_cw(float y) {
  const float k = rintf(y * 0.318309886183790672f);
  float r = fmaf(-k, 3.140625f, y);            // pi = 3.140625 (exact in 9 bits: k * hi is exact) + 9.67653589793e-4
  r = fmaf(-k, 9.67653589793e-4f, r);
  const float r2 = r * r;
  float p = fmaf(r2, 1.6059043836821613e-10f, -2.5052108385441720e-8f);   // 1/13!, -1/11!
  p = fmaf(p, r2, 2.7557319223985893e-6f);     // 1/9!
  p = fmaf(p, r2, -1.9841269841269841e-4f);    // -1/7!
  p = fmaf(p, r2, 8.3333333333333332e-3f);     // 1/5!
  p = fmaf(p, r2, -1.6666666666666666e-1f);    // -1/3!
  const float s = fmaf(p * r2, r, r);
  return __int_as_float(__float_as_int(s) ^ ((int)k << 31));
}
__device__ __forceinline__ float act_snake(float x, float alpha) {
  const float s = sin_cw(x * alpha);
  const float d = alpha + 1e-9f;
  float r = __builtin_amdgcn_rcpf(d);          // v_rcp_f32 (1 ulp) + one Newton step: the IEEE division costs ~10 instructions per element
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  return x + r * s * s;
}
__device__ __forceinline__ float apply_act(int act, float v, float param, float slope) {
  switch (act) {
    case CV_ACT_GELU: return act_gelu(v);
    case CV_ACT_SILU: return act_silu(v);
    case CV_ACT_MISH: return act_mish(v);
    case CV_ACT_LEAKY: return v > 0.0f ? v : v * slope;
    case CV_ACT_ELU: return act_elu(v);
    case CV_ACT_SNAKE: return act_snake(v, param);
    case CV_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
