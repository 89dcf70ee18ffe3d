// Flow-matching helper kernels (elementwise / gather, HBM-bound, float4 / 8-byte accesses).
#include "cv_device.h"

namespace {

template <int DT>
__device__ __forceinline__ void st4(void* base, int64_t i, float a, float b, float c, float d) {
  if constexpr (DT == CV_F32) {
    *(float4*)((float*)base + i) = make_float4(a, b, c, d);
  } else {
    uint2 u;
    u.x = pack2<DT>(a, b);
    u.y = pack2<DT>(c, d);
    *(uint2*)((uint16_t*)base + i) = u;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void embedding_kernel(const float* table, const int32_t* idx, void* out, int rows, int dim, int ldo) {
  const int d4 = dim >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * d4; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / d4), c = (int)(i - (int64_t)r * d4) * 4;
    const int id = idx[r];
    if (id == -2) continue;  // -2: leave the row untouched (multi-table assembly); -1: zero row
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (id >= 0) v = *(const float4*)(table + (int64_t)id * dim + c);
    st4<DT>(out, (int64_t)r * ldo + c, v.x, v.y, v.z, v.w);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void est_pack_kernel(const float* x, const float* mu, const float* spks, const float* cond,
                                                       void* xin, int B, int T, int C) {
  // one thread = 4 channels of one (b, t, segment s in 0..3); writes both CFG rows
  const int c4 = C >> 2;
  const int64_t total = (int64_t)B * T * 4 * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    int64_t r = i / c4;
    const int seg = (int)(r & 3);
    r >>= 2;
    const int t = (int)(r % T), b = (int)(r / T);
    const int64_t src = ((int64_t)b * T + t) * C + c;
    float4 v;
    if (seg == 0) v = *(const float4*)(x + src);
    else if (seg == 1) v = *(const float4*)(mu + src);
    else if (seg == 2) v = *(const float4*)(spks + (int64_t)b * C + c);
    else v = *(const float4*)(cond + src);
    const int64_t o0 = (((int64_t)(2 * b) * T + t) * 4 + seg) * C + c;
    const int64_t o1 = (((int64_t)(2 * b + 1) * T + t) * 4 + seg) * C + c;
    st4<DT>(xin, o0, v.x, v.y, v.z, v.w);
    if (seg == 0) st4<DT>(xin, o1, v.x, v.y, v.z, v.w);
    else st4<DT>(xin, o1, 0.f, 0.f, 0.f, 0.f);
  }
}

__global__ __launch_bounds__(256) void cfm_update_kernel(float* x, const float* v, int B, int64_t n4_per_b, float dt, float w) {
  const int64_t total = (int64_t)B * n4_per_b;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / n4_per_b);
    const int64_t o = i - (int64_t)b * n4_per_b;
    const float4 vc = ((const float4*)v)[(int64_t)(2 * b) * n4_per_b + o];
    const float4 vu = ((const float4*)v)[(int64_t)(2 * b + 1) * n4_per_b + o];
    float4 xv = ((float4*)x)[i];
    xv.x += dt * ((1.f + w) * vc.x - w * vu.x);
    xv.y += dt * ((1.f + w) * vc.y - w * vu.y);
    xv.z += dt * ((1.f + w) * vc.z - w * vu.z);
    xv.w += dt * ((1.f + w) * vc.w - w * vu.w);
    ((float4*)x)[i] = xv;
  }
}

// F.interpolate(mode='linear', align_corners=False) along T of a channels-last fp32 tensor (InterpolateRegulator.inference,
// flow/length_regulator.py:49-70): src = (t + 0.5) * T_in / T_out - 0.5 clamped at 0, fp32 index arithmetic as ATen's.
template <int DT>
__global__ __launch_bounds__(256) void interp_linear_kernel(const float* x, int ldx, int T_in, void* y, int ldy, int T_out, int C) {
  const float scale = (float)T_in / (float)T_out;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)T_out * C; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i / C), c = (int)(i % C);
    const float src = fmaxf(scale * ((float)t + 0.5f) - 0.5f, 0.f);
    const int i0 = min((int)src, T_in - 1), i1 = min(i0 + 1, T_in - 1);
    const float l1 = src - (float)i0, l0 = 1.f - l1;
    const float v = l0 * x[(int64_t)i0 * ldx + c] + l1 * x[(int64_t)i1 * ldx + c];
    if constexpr (DT == CV_F32) ((float*)y)[(int64_t)t * ldy + c] = v;
    else ((uint16_t*)y)[(int64_t)t * ldy + c] = Elem16<DT>::from_f32(v);
  }
}

inline int nblocks(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

}  // namespace

#define DISPATCH_DT(dt, CALL)                                   \
  switch (dt) {                                                 \
    case CV_F32: { constexpr int DT = CV_F32; CALL; } break;    \
    case CV_BF16: { constexpr int DT = CV_BF16; CALL; } break;  \
    case CV_F16: { constexpr int DT = CV_F16; CALL; } break;    \
    default: return CV_ERR_ARG;                                 \
  }

extern "C" int cv_embedding(const float* table, const int32_t* idx, void* out, int32_t dtype, int32_t rows, int32_t dim, int32_t ldo, void* stream) {
  if (!table || !idx || !out || rows <= 0 || dim <= 0 || (dim & 3) || (ldo & 3) || ldo < dim) return CV_ERR_ARG;
  DISPATCH_DT(dtype, hipLaunchKernelGGL(embedding_kernel<DT>, dim3(nblocks((int64_t)rows * (dim >> 2))), dim3(256), 0,
                                        (hipStream_t)stream, table, idx, out, rows, dim, ldo));
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_est_pack(const float* x, const float* mu, const float* spks, const float* cond, void* xin, int32_t dtype,
                           int32_t B, int32_t T, int32_t C, void* stream) {
  if (!x || !mu || !spks || !cond || !xin || B <= 0 || T <= 0 || C <= 0 || (C & 3)) return CV_ERR_ARG;
  DISPATCH_DT(dtype, hipLaunchKernelGGL(est_pack_kernel<DT>, dim3(nblocks((int64_t)B * T * C)), dim3(256), 0, (hipStream_t)stream,
                                        x, mu, spks, cond, xin, B, T, C));
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_cfm_update(float* x, const float* v, int32_t B, int32_t T, int32_t C, float dt, float cfg_rate, void* stream) {
  if (!x || !v || B <= 0 || T <= 0 || C <= 0 || ((T * C) & 3)) return CV_ERR_ARG;
  const int64_t n4 = (int64_t)T * C / 4;
  hipLaunchKernelGGL(cfm_update_kernel, dim3(nblocks((int64_t)B * n4)), dim3(256), 0, (hipStream_t)stream, x, v, B, n4, dt, cfg_rate);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_interp_linear_cl(const float* x, int32_t ldx, int32_t T_in, void* y, int32_t ldy, int32_t dtype, int32_t T_out,
                                   int32_t C, void* stream) {
  if (!x || !y || T_in <= 0 || T_out <= 0 || C <= 0 || ldx < C || ldy < C) return CV_ERR_ARG;
  DISPATCH_DT(dtype, hipLaunchKernelGGL(interp_linear_kernel<DT>, dim3(nblocks((int64_t)T_out * C)), dim3(256), 0, (hipStream_t)stream,
                                        x, ldx, T_in, y, ldy, T_out, C));
  CV_CHECK_LAUNCH();
  return CV_OK;
}
