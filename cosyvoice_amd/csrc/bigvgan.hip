// BigVGAN fused anti-aliased SnakeBeta activation for gfx950 (HBM-bound: one read + one write of the tensor).
// Workgroup = 256 outputs of one (batch, channel) row: x tile (+6 halo) -> LDS, 2x-upsampled + activated signal
// (2*256 + 11 values) -> LDS, each thread then takes its 12-tap stride-2 low-pass from LDS.  T is the contiguous dim,
// so global accesses are fully coalesced; alpha/beta are wave-uniform (one channel per workgroup).
#include "cv_device.h"

namespace {

constexpr int TT = 256;          // outputs per workgroup
constexpr int XH = 7;            // x halo on each side
constexpr int NU = 2 * TT + 11;  // upsampled values needed: n' in [2 t0 - 5, 2 t0 + 2 TT + 5]

template <int DT>
__device__ __forceinline__ float ld(const void* p, int64_t i) {
  if constexpr (DT == CV_F32) return ((const float*)p)[i];
  else return Elem16<DT>::to_f32(((const uint16_t*)p)[i]);
}
template <int DT>
__device__ __forceinline__ void stv(void* p, int64_t i, float v) {
  if constexpr (DT == CV_F32) ((float*)p)[i] = v;
  else ((uint16_t*)p)[i] = Elem16<DT>::from_f32(v);
}

template <int DT>
__global__ __launch_bounds__(256) void anti_alias_kernel(const void* x, void* y, int C, int T, const float* upf, const float* dnf,
                                                         const float* alog, const float* blog) {
  __shared__ float sx[TT + 2 * XH];
  __shared__ float su[NU + 1];
  __shared__ float sf[24];
  const int tid = threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * TT;
  const int64_t row = ((int64_t)b * C + c) * T;
  if (tid < 12) sf[tid] = upf[tid];
  else if (tid < 24) sf[tid] = dnf[tid - 12];
  // x[t0 - XH .. t0 + TT + XH) with replicate clamping (= the replicate pad of UpSample1d, resample.py:30)
  for (int i = tid; i < TT + 2 * XH; i += 256) sx[i] = ld<DT>(x, row + min(max(t0 - XH + i, 0), T - 1));
  __syncthreads();
  const float alpha = __expf(alog[c]);
  const float inv_beta = 1.0f / (__expf(blog[c]) + 1e-9f);
  // up[n] = 2 * sum_i xp[i] f[n + 15 - 2 i], xp[i] = x[clamp(i - 5)]  (conv_transpose stride 2, slice [15:-15])
  for (int j = tid; j < NU; j += 256) {
    int n = 2 * t0 - 5 + j;
    n = min(max(n, 0), 2 * T - 1);  // replicate pad of the low-pass input (filter.py:127)
    const int ilo = (n + 5) >> 1;   // ceil((n + 4) / 2)
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int i = ilo + q;          // index into xp
      const int k = n + 15 - 2 * i;   // filter tap, 0..11 by construction
      const int xi = min(max(i - 5, 0), T - 1);
      acc += sx[xi - (t0 - XH)] * sf[k];
    }
    const float u = 2.0f * acc;
    const float sn = sinf(u * alpha);
    su[j] = u + inv_beta * sn * sn;
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t < T) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += su[2 * tid + k] * sf[12 + k];
    stv<DT>(y, row + t, acc);
  }
}

// Channels-last form for the AMP blocks ([B][T][C] between the channels-last conv GEMMs, no layout round trips):
// workgroup = 32 outputs x 64 channels, thread = one channel (lane) x every 4th time step, so every global and LDS access
// runs along the contiguous channel dimension.  IDT / ODT: input / output element types (the residual stream is fp32,
// the following conv takes 16-bit or fp32 operands).
constexpr int CT = 32, CXH = 7, CNU = 2 * CT + 11;

template <int IDT, int ODT>
__global__ __launch_bounds__(256) void anti_alias_cl_kernel(const void* x, int ldx, void* y, int ldy, int T, int C, const float* upf,
                                                            const float* dnf, const float* alog, const float* blog) {
  __shared__ float sx[CT + 2 * CXH][64];
  __shared__ float su[CNU][64];
  __shared__ float sf[24];
  const int tid = threadIdx.x, cl = tid & 63, tr = tid >> 6;
  const int c = blockIdx.y * 64 + cl, b = blockIdx.z;
  const int t0 = blockIdx.x * CT;
  const int cc = min(c, C - 1);
  if (tid < 12) sf[tid] = upf[tid];
  else if (tid < 24) sf[tid] = dnf[tid - 12];
  for (int i = tr; i < CT + 2 * CXH; i += 4) {
    const int t = min(max(t0 - CXH + i, 0), T - 1);  // replicate pad of UpSample1d (resample.py:30)
    sx[i][cl] = ld<IDT>(x, ((int64_t)b * T + t) * ldx + cc);
  }
  const float alpha = __expf(alog[cc]);
  const float inv_beta = 1.0f / (__expf(blog[cc]) + 1e-9f);
  __syncthreads();
  for (int j = tr; j < CNU; j += 4) {
    int n = 2 * t0 - 5 + j;
    n = min(max(n, 0), 2 * T - 1);  // replicate pad of the low-pass input (filter.py:127)
    const int ilo = (n + 5) >> 1;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int i = ilo + q;
      const int k = n + 15 - 2 * i;
      const int xi = min(max(i - 5, 0), T - 1);
      acc += sx[xi - (t0 - CXH)][cl] * sf[k];
    }
    const float u = 2.0f * acc;
    const float sn = sinf(u * alpha);
    su[j][cl] = u + inv_beta * sn * sn;
  }
  __syncthreads();
  if (c >= C) return;
  for (int tt = tr; tt < CT; tt += 4) {
    const int t = t0 + tt;
    if (t >= T) break;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += su[2 * tt + k][cl] * sf[12 + k];
    stv<ODT>(y, ((int64_t)b * T + t) * ldy + c, acc);
  }
}

}  // namespace

extern "C" int cv_anti_alias_act(const void* x, void* y, int32_t dtype, int32_t B, int32_t C, int32_t T, const float* up_filter,
                                 const float* down_filter, const float* alpha_log, const float* beta_log, void* stream) {
  if (!x || !y || !up_filter || !down_filter || !alpha_log || !beta_log || B <= 0 || C <= 0 || T <= 0) return CV_ERR_ARG;
  if (C > 65535 || B > 65535) return CV_ERR_ARG;
  dim3 grid((T + TT - 1) / TT, C, B);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CV_F32: hipLaunchKernelGGL(anti_alias_kernel<CV_F32>, grid, dim3(256), 0, st, x, y, C, T, up_filter, down_filter, alpha_log, beta_log); break;
    case CV_BF16: hipLaunchKernelGGL(anti_alias_kernel<CV_BF16>, grid, dim3(256), 0, st, x, y, C, T, up_filter, down_filter, alpha_log, beta_log); break;
    case CV_F16: hipLaunchKernelGGL(anti_alias_kernel<CV_F16>, grid, dim3(256), 0, st, x, y, C, T, up_filter, down_filter, alpha_log, beta_log); break;
    default: return CV_ERR_ARG;
  }
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_anti_alias_act_cl(const void* x, int32_t ldx, int32_t in_dtype, void* y, int32_t ldy, int32_t out_dtype, int32_t B,
                                    int32_t T, int32_t C, const float* up_filter, const float* down_filter, const float* alpha_log,
                                    const float* beta_log, void* stream) {
  if (!x || !y || !up_filter || !down_filter || !alpha_log || !beta_log || B <= 0 || C <= 0 || T <= 0 || ldx < C || ldy < C) return CV_ERR_ARG;
  if (B > 65535) return CV_ERR_ARG;
  dim3 grid((T + CT - 1) / CT, (C + 63) / 64, B);
  hipStream_t st = (hipStream_t)stream;
#define AA_OUT(IDT_)                                                                                                              \
  switch (out_dtype) {                                                                                                            \
    case CV_F32: hipLaunchKernelGGL((anti_alias_cl_kernel<IDT_, CV_F32>), grid, dim3(256), 0, st, x, ldx, y, ldy, T, C, up_filter, down_filter, alpha_log, beta_log); break;  \
    case CV_BF16: hipLaunchKernelGGL((anti_alias_cl_kernel<IDT_, CV_BF16>), grid, dim3(256), 0, st, x, ldx, y, ldy, T, C, up_filter, down_filter, alpha_log, beta_log); break; \
    case CV_F16: hipLaunchKernelGGL((anti_alias_cl_kernel<IDT_, CV_F16>), grid, dim3(256), 0, st, x, ldx, y, ldy, T, C, up_filter, down_filter, alpha_log, beta_log); break;  \
    default: return CV_ERR_ARG;                                                                                                   \
  }
  switch (in_dtype) {
    case CV_F32: AA_OUT(CV_F32); break;
    case CV_BF16: AA_OUT(CV_BF16); break;
    case CV_F16: AA_OUT(CV_F16); break;
    default: return CV_ERR_ARG;
  }
#undef AA_OUT
  CV_CHECK_LAUNCH();
  return CV_OK;
}
