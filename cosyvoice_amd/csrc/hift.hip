// HiFT vocoder helper kernels for gfx950: layout changes, multi-alpha Snake, 16-point STFT / iSTFT, harmonic source.
// All are HBM-bound elementwise / small-stencil kernels: coalesced along the contiguous dim, fp32 math.
#include "cv_device.h"

namespace {

template <int DT>
__device__ __forceinline__ void st(void* base, int64_t i, float v) {
  if constexpr (DT == CV_F32) ((float*)base)[i] = v;
  else ((uint16_t*)base)[i] = Elem16<DT>::from_f32(v);
}

// ---- [B][C][T] fp32 -> [B][T][ldo] DT through a 32x33 LDS tile (coalesced on both sides)
template <int DT>
__global__ __launch_bounds__(256) void to_cl_kernel(const float* x, void* out, int C, int T, int ldo) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    tile[r][tx] = (c < C && t < T) ? x[((int64_t)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    if (t < T && c < ldo) st<DT>(out, ((int64_t)b * T + t) * ldo + c, tile[tx][r]);
  }
}

__global__ __launch_bounds__(256) void to_cf_kernel(const float* x, float* out, int C, int T, int ldx) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    tile[r][tx] = (c < C && t < T) ? x[((int64_t)b * T + t) * ldx + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    if (c < C && t < T) out[((int64_t)b * C + c) * T + t] = tile[tx][r];
  }
}

struct SnakeArgs {
  const float* alpha[4];
  void* out[4];
};

template <int DT>
__global__ __launch_bounds__(256) void snake_multi_kernel(const float* x, int64_t n4, int C, int ldx, int n, int ldo, SnakeArgs a) {
  // one float4 (4 channels of one row) per thread, grid-stride
  const int c4 = C >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / c4;
    const int col = (int)(i - row * c4) * 4;
    const float4 v = *(const float4*)(x + row * ldx + col);
    for (int k = 0; k < n; ++k) {
      const float4 al = *(const float4*)(a.alpha[k] + col);
      const float o0 = act_snake(v.x, al.x), o1 = act_snake(v.y, al.y), o2 = act_snake(v.z, al.z), o3 = act_snake(v.w, al.w);
      if constexpr (DT == CV_F32X3) {   // pre-split storage of the bf16x3 convs (cv_gemm_params.x3_flags)
        const float f[4] = {o0, o1, o2, o3};
        uint16_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          h[j] = Elem16<CV_BF16>::from_f32(f[j]);
          l[j] = Elem16<CV_BF16>::from_f32(f[j] - Elem16<CV_BF16>::to_f32(h[j]));
        }
        char* g = (char*)a.out[k] + (row * ldo + (col & ~7)) * 4 + ((col >> 2) & 1) * 8;   // 8-value group [8 hi | 8 lo], this lane's half
        *(uint2*)g = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        *(uint2*)(g + 16) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
      } else if constexpr (DT == CV_F32) {
        *(float4*)((float*)a.out[k] + row * ldo + col) = make_float4(o0, o1, o2, o3);
      } else {
        uint2 u;
        u.x = pack2<DT>(o0, o1);
        u.y = pack2<DT>(o2, o3);
        *(uint2*)((uint16_t*)a.out[k] + row * ldo + col) = u;
      }
    }
  }
}

__constant__ float c_cos16[16] = {1.f, 0.92387953251f, 0.70710678118f, 0.38268343236f, 0.f, -0.38268343236f,
                                  -0.70710678118f, -0.92387953251f, -1.f, -0.92387953251f, -0.70710678118f,
                                  -0.38268343236f, 0.f, 0.38268343236f, 0.70710678118f, 0.92387953251f};
__device__ __forceinline__ float hann16(int n) { return 0.5f - 0.5f * c_cos16[n & 15]; }

// ---- STFT: one thread per (frame, bin k in 0..8) computes real+imag
template <int DT>
__global__ __launch_bounds__(256) void stft16_kernel(const float* s, void* out, int S, int F, int ldo) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int f = idx / 12, k = idx - f * 12;  // 12 lanes per frame: 9 bins + 3 lanes for zero padding columns
  if (f >= F) return;
  const float* x = s + (int64_t)b * S;
  const int64_t ob = ((int64_t)b * F + f) * ldo;
  if (k >= 9) {
    for (int c = 18 + (k - 9); c < ldo; c += 3) st<DT>(out, ob + c, 0.f);
    return;
  }
  float re = 0.f, im = 0.f;
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    int j = f * 4 + n - 8;
    if (j < 0) j = -j;
    if (j >= S) j = 2 * (S - 1) - j;
    const float v = hann16(n) * x[j];
    const int ph = (k * n) & 15;
    re += v * c_cos16[ph];
    im -= v * c_cos16[(ph + 12) & 15];  // sin(2 pi ph/16) = cos(2 pi (ph-4)/16)
  }
  st<DT>(out, ob + k, re);
  st<DT>(out, ob + 9 + k, im);
}

// ---- iSTFT: block = 64 frames -> 256 output samples; frames (+3 halo before) staged in LDS as real/imag
__global__ __launch_bounds__(256) void istft16_kernel(const float* y, float* wav, int F, int ldy, float limit) {
  __shared__ float sre[67][9], sim[67][9];
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * 64;           // outputs n' in [4 f0, 4 f0 + 256)
  const int L = (F - 1) * 4;
  // frames covering sample n = n' + 8: f in [ceil((n-15)/4), floor(n/4)] -> for the block: f0 - 1 .. f0 + 65
  for (int i = threadIdx.x; i < 67 * 9; i += 256) {
    const int fl = i / 9, k = i - fl * 9;
    const int f = f0 - 1 + fl;
    float re = 0.f, im = 0.f;
    if (f >= 0 && f < F) {
      const float* row = y + ((int64_t)b * F + f) * ldy;
      const float mag = fminf(__expf(row[k]), 100.0f);
      const float ph = sinf(row[9 + k]);
      float sn, cs;
      sincosf(ph, &sn, &cs);
      re = mag * cs;
      im = mag * sn;
    }
    sre[fl][k] = re;
    sim[fl][k] = im;
  }
  __syncthreads();
  const int np = f0 * 4 + threadIdx.x;
  if (np >= L) return;
  const int n = np + 8;
  float acc = 0.f, env = 0.f;
  const int fhi = n >> 2;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int f = fhi - d;
    const int j = n - 4 * f;  // 0..15
    if (f < 0 || f >= F || j > 15) continue;
    const int fl = f - (f0 - 1);
    float v = sre[fl][0] + ((j & 1) ? -sre[fl][8] : sre[fl][8]);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const int ph = (k * j) & 15;
      v += 2.0f * (sre[fl][k] * c_cos16[ph] - sim[fl][k] * c_cos16[(ph + 12) & 15]);
    }
    const float w = hann16(j);
    acc += w * v * (1.0f / 16.0f);
    env += w * w;
  }
  float o = acc / env;
  o = fminf(fmaxf(o, -limit), limit);
  wav[(int64_t)b * L + np] = o;
}

// ---- source: (1) frame-start phases per (b, harmonic) in fp64, (2) per-sample synthesis + tanh(linear)
__global__ void source_prefix_kernel(const float* f0, double* work, int T, int up, int nh, float sr) {
  const int b = blockIdx.x, h = threadIdx.x;
  if (h >= nh) return;
  double acc = 0.0;
  double* w = work + ((int64_t)b * nh + h) * T;
  for (int t = 0; t < T; ++t) {
    w[t] = acc;
    // the reference forms F = f0*(h+1)/sr in fp32 (generator.py:145) and sums `up` copies of it
    const float Fm = f0[(int64_t)b * T + t] * (float)(h + 1) / sr;
    acc += (double)Fm * up;
    acc -= floor(acc);
  }
}

__global__ __launch_bounds__(256) void source_synth_kernel(const float* f0, const float* phase_vec, const float* noise,
                                                           const float* lin_w, const float* lin_b, const double* work,
                                                           float* s, int T, int up, int nh, float sr, float sine_amp,
                                                           float noise_std, float vthr) {
  const int b = blockIdx.y;
  const int64_t S = (int64_t)T * up;
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= S) return;
  const int t = (int)(n / up), i = (int)(n - (int64_t)t * up);
  const float f = f0[(int64_t)b * T + t];
  const float uv = f > vthr ? 1.f : 0.f;
  const float namp = uv * noise_std + (1.f - uv) * sine_amp / 3.f;
  float accum = lin_b[0];
  for (int h = 0; h < nh; ++h) {
    const float Fm = f * (float)(h + 1) / sr;
    double ph = work[((int64_t)b * nh + h) * T + t] + (double)Fm * (i + 1);
    ph -= floor(ph);
    const float theta = 6.283185307179586f * (float)ph;
    const float pv = h == 0 ? 0.f : phase_vec[b * nh + h];
    const float sw = sine_amp * sinf(theta + pv) * uv + namp * noise[((int64_t)b * nh + h) * S + n];
    accum += lin_w[h] * sw;
  }
  s[(int64_t)b * S + n] = tanhf(accum);
}

}  // namespace

#define DISPATCH_DT(dt, CALL)                                   \
  switch (dt) {                                                 \
    case CV_F32: { constexpr int DT = CV_F32; CALL; } break;    \
    case CV_BF16: { constexpr int DT = CV_BF16; CALL; } break;  \
    case CV_F16: { constexpr int DT = CV_F16; CALL; } break;    \
    default: return CV_ERR_ARG;                                 \
  }

extern "C" int cv_to_channels_last(const float* x, void* out, int32_t dtype, int32_t B, int32_t C, int32_t T, int32_t ldo, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || T <= 0 || ldo < C) return CV_ERR_ARG;
  dim3 grid((T + 31) / 32, (ldo + 31) / 32, B);
  DISPATCH_DT(dtype, hipLaunchKernelGGL(to_cl_kernel<DT>, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, T, ldo));
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_to_channels_first(const float* x, float* out, int32_t B, int32_t C, int32_t T, int32_t ldx, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || T <= 0 || ldx < C) return CV_ERR_ARG;
  dim3 grid((T + 31) / 32, (C + 31) / 32, B);
  hipLaunchKernelGGL(to_cf_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, T, ldx);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_snake_multi(const float* x, int32_t rows, int32_t C, int32_t ldx, int32_t n, const float* const* alpha,
                              void* const* out, int32_t ldo, int32_t dtype, void* stream) {
  if (!x || rows <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldo & 3) || n < 1 || n > 4 || !alpha || !out) return CV_ERR_ARG;
  // pre-split output: whole groups of 8 values ([8 hi | 8 lo], 32 bytes) — a half group at a row's end would put its lo half past the row
  if (dtype == CV_F32X3 && ((C & 7) || (ldo & 7))) return CV_ERR_ARG;
  SnakeArgs a{};
  for (int i = 0; i < n; ++i) { a.alpha[i] = alpha[i]; a.out[i] = out[i]; if (!alpha[i] || !out[i]) return CV_ERR_ARG; }
  const int64_t n4 = (int64_t)rows * (C >> 2);
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  if (dtype == CV_F32X3) hipLaunchKernelGGL(snake_multi_kernel<CV_F32X3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n4, C, ldx, n, ldo, a);
  else DISPATCH_DT(dtype, hipLaunchKernelGGL(snake_multi_kernel<DT>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n4, C, ldx, n, ldo, a));
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_stft16(const float* s, void* out, int32_t dtype, int32_t B, int32_t S, int32_t ldo, void* stream) {
  if (!s || !out || B <= 0 || S < 16 || (S & 3) || ldo < 18) return CV_ERR_ARG;
  const int F = S / 4 + 1;
  dim3 grid((F * 12 + 255) / 256, B);
  DISPATCH_DT(dtype, hipLaunchKernelGGL(stft16_kernel<DT>, grid, dim3(256), 0, (hipStream_t)stream, s, out, S, F, ldo));
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_istft16(const float* y, float* wav, int32_t B, int32_t F, int32_t ldy, float audio_limit, void* stream) {
  if (!y || !wav || B <= 0 || F < 2 || ldy < 18) return CV_ERR_ARG;
  dim3 grid((F - 1 + 63) / 64, B);
  hipLaunchKernelGGL(istft16_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, wav, F, ldy, audio_limit);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_hift_source(const float* f0, const float* phase_vec, const float* noise, const float* lin_w, const float* lin_b,
                              double* work, float* s, int32_t B, int32_t T, int32_t up, int32_t nh, float sampling_rate,
                              float sine_amp, float noise_std, float voiced_threshold, void* stream) {
  if (!f0 || !phase_vec || !noise || !lin_w || !lin_b || !work || !s || B <= 0 || T <= 0 || up <= 0 || nh < 1 || nh > 16) return CV_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(source_prefix_kernel, dim3(B), dim3(64), 0, st, f0, work, T, up, nh, sampling_rate);
  const int64_t S = (int64_t)T * up;
  dim3 grid((unsigned)((S + 255) / 256), B);
  hipLaunchKernelGGL(source_synth_kernel, grid, dim3(256), 0, st, f0, phase_vec, noise, lin_w, lin_b, work, s, T, up, nh,
                     sampling_rate, sine_amp, noise_std, voiced_threshold);
  CV_CHECK_LAUNCH();
  return CV_OK;
}
