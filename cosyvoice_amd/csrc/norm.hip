// cv_layernorm: row LayerNorm / RMSNorm over the last dim with fused affine, optional per-group add (time embedding),
// optional Mish, fp32 and/or 16-bit outputs.  One wave per row (dim <= 8192), float4 loads, two-pass in registers.
#include "cv_device.h"

namespace {

// MAXV = float4 per lane: 1 / 2 / 4 / 8 for dim <= 256 / 512 / 1024 / 2048 (no masked-out load instructions)
template <int ODT, int MAXV>
__global__ __launch_bounds__(256) void norm_kernel(const cv_norm_params p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const float* x = p.x + (int64_t)row * p.ldx;
  float4 v[MAXV];
  const int nv = p.dim >> 2;
  float s = 0.f;
  // unconditional clamped loads (a load under a per-element branch is waited for individually), masked afterwards
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    const float4 t = *(const float4*)(x + min(c, nv - 1) * 4);
    const float mk = c < nv ? 1.f : 0.f;
    v[i] = make_float4(t.x * mk, t.y * mk, t.z * mk, t.w * mk);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  float mean = 0.f;
  if (!p.rms) mean = wave_sum(s) / (float)p.dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)p.dim + p.eps);
  const float* add = p.add ? p.add + (int64_t)(row / p.rows_per_group) * p.add_ld : nullptr;
  const bool has_g = p.gamma != nullptr, has_b = p.beta != nullptr, has_a = add != nullptr;  // wave-uniform
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = i * 64 + lane;
    const int cc = min(c, nv - 1) * 4;
    // parameter vectors loaded as float4, unconditionally within the uniform branches
    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f), a4 = b4;
    if (has_g) g4 = *(const float4*)(p.gamma + cc);
    if (has_b) b4 = *(const float4*)(p.beta + cc);
    if (has_a) a4 = *(const float4*)(add + cc);
    if (c >= nv) continue;
    const int col = c * 4;
    float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w}, aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = (o[r] - mean) * rstd * gg[r] + bb[r];
      if (p.act == CV_ACT_MISH) y = act_mish(y);
      else if (p.act == CV_ACT_LEAKY) y = fmaxf(y, 0.f);   // ReLU (the params carry no slope): LegacyLinearNoSubsampling
      y += aa[r];
      o[r] = y * p.out_scale;
    }
    if (p.out_f32) *(float4*)(p.out_f32 + (int64_t)row * p.ldo32 + col) = make_float4(o[0], o[1], o[2], o[3]);
    if (p.out_act) {
      if constexpr (ODT == CV_F32) {
        *(float4*)((float*)p.out_act + (int64_t)row * p.ldoa + col) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        uint2 u;
        u.x = pack2<ODT>(o[0], o[1]);
        u.y = pack2<ODT>(o[2], o[3]);
        *(uint2*)((uint16_t*)p.out_act + (int64_t)row * p.ldoa + col) = u;
      }
    }
  }
}

// ---- GroupNorm over channels-last [B][T][C] (CosyVoice-v1: Block1D of the non-causal estimator, InterpolateRegulator) ----
// The statistics span all T rows of a group's channels, so the reduction is split over 32-row chunks: pass 1 writes one
// (mean, M2) pair per (batch, group, chunk) — two-pass inside the chunk, the second read hits L2 — and pass 2 merges the pairs
// in fixed order (Chan's parallel-variance formula: deterministic, no atomics, no E[x^2]-mean^2 cancellation) before
// normalising its own chunk with the fused affine / Mish / time-embedding add.
constexpr int GN_ROWS = 32;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const cv_groupnorm_params p) {
  __shared__ float red[4];
  const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int cpg = p.C / p.groups;
  const int t0 = chunk * GN_ROWS, nr = min(GN_ROWS, p.T - t0);
  const int n = nr * cpg;
  const float* x = p.x + (int64_t)b * p.x_bs + (int64_t)t0 * p.ldx + g * cpg;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[(int64_t)(i / cpg) * p.ldx + (i % cpg)];
  const float mean = block_sum(s, red) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = x[(int64_t)(i / cpg) * p.ldx + (i % cpg)] - mean;
    q += d * d;
  }
  const float m2 = block_sum(q, red);
  if (threadIdx.x == 0) {
    float* o = p.partial + (((int64_t)b * p.groups + g) * gridDim.x + chunk) * 2;
    o[0] = mean;
    o[1] = m2;
  }
}

template <int ODT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const cv_groupnorm_params p) {
  const int chunk = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int cpg = p.C / p.groups;
  const int nch = gridDim.x;
  // merge the chunk statistics (every thread redundantly: <= T/32 broadcast loads)
  const float* part = p.partial + ((int64_t)b * p.groups + g) * nch * 2;
  float mean = 0.f, m2 = 0.f, cnt = 0.f;
  for (int c = 0; c < nch; ++c) {
    const float nb = (float)(min(GN_ROWS, p.T - c * GN_ROWS) * cpg);
    const float mb = part[2 * c], qb = part[2 * c + 1];
    const float tot = cnt + nb, d = mb - mean;
    mean += d * (nb / tot);
    m2 += qb + d * d * (cnt * nb / tot);
    cnt = tot;
  }
  const float rstd = rsqrtf(m2 / cnt + p.eps);
  const int t0 = chunk * GN_ROWS, nr = min(GN_ROWS, p.T - t0);
  const int n = nr * cpg;
  const float* add = p.add ? p.add + (int64_t)b * p.add_ld : nullptr;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int t = t0 + i / cpg, c = g * cpg + (i % cpg);
    float y = (p.x[(int64_t)b * p.x_bs + (int64_t)t * p.ldx + c] - mean) * rstd;
    if (p.gamma) y *= p.gamma[c];
    if (p.beta) y += p.beta[c];
    if (p.act == CV_ACT_MISH) y = act_mish(y);
    if (add) y += add[c];
    if (p.out_f32) p.out_f32[(int64_t)b * p.o32_bs + (int64_t)t * p.ldo32 + c] = y;
    if (p.out_act) {
      const int64_t o = (int64_t)b * p.oa_bs + (int64_t)t * p.ldoa + c;
      if constexpr (ODT == CV_F32) ((float*)p.out_act)[o] = y;
      else ((uint16_t*)p.out_act)[o] = Elem16<ODT>::from_f32(y);
    }
  }
}

}  // namespace

extern "C" int cv_sizeof_groupnorm_params(void) { return (int)sizeof(cv_groupnorm_params); }

extern "C" int64_t cv_groupnorm_workspace_floats(int32_t B, int32_t T, int32_t groups) {
  if (B <= 0 || T <= 0 || groups <= 0) return 0;
  return (int64_t)B * groups * ((T + GN_ROWS - 1) / GN_ROWS) * 2;
}

extern "C" int cv_groupnorm_cl(const cv_groupnorm_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  const cv_groupnorm_params p = *pp;
  if (p.B <= 0 || p.T <= 0 || p.C <= 0 || p.groups <= 0 || p.C % p.groups || !p.x || !p.partial) return CV_ERR_ARG;
  if (p.B > 65535 || p.groups > 65535 || p.ldx < p.C || (!p.out_f32 && !p.out_act)) return CV_ERR_ARG;
  if ((p.out_f32 && p.ldo32 < p.C) || (p.out_act && p.ldoa < p.C)) return CV_ERR_ARG;
  if (p.act != CV_ACT_NONE && p.act != CV_ACT_MISH) return CV_ERR_ARG;
  dim3 grid((p.T + GN_ROWS - 1) / GN_ROWS, p.groups, p.B);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, st, p);
  switch (p.out_act ? p.out_dtype : CV_F32) {
    case CV_F32: hipLaunchKernelGGL(gn_apply_kernel<CV_F32>, grid, dim3(256), 0, st, p); break;
    case CV_BF16: hipLaunchKernelGGL(gn_apply_kernel<CV_BF16>, grid, dim3(256), 0, st, p); break;
    case CV_F16: hipLaunchKernelGGL(gn_apply_kernel<CV_F16>, grid, dim3(256), 0, st, p); break;
    default: return CV_ERR_ARG;
  }
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_layernorm(const cv_norm_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  cv_norm_params p = *pp;
  if (p.rows <= 0 || p.dim <= 0 || (p.dim & 3) || p.dim > 2048 || !p.x) return CV_ERR_ARG;
  if ((p.ldx & 3) || (!p.out_f32 && !p.out_act)) return CV_ERR_ARG;
  if (p.out_f32 && (p.ldo32 & 3)) return CV_ERR_ARG;
  if (p.out_act && (p.ldoa & 3)) return CV_ERR_ARG;
  if (p.out_scale == 0.f) p.out_scale = 1.f;
  if (p.rows_per_group <= 0) p.rows_per_group = p.rows;
  dim3 grid((p.rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
#define NORM_LAUNCH(ODT_)                                                                                    \
  do {                                                                                                       \
    if (p.dim <= 256) hipLaunchKernelGGL((norm_kernel<ODT_, 1>), grid, dim3(256), 0, st, p);                  \
    else if (p.dim <= 512) hipLaunchKernelGGL((norm_kernel<ODT_, 2>), grid, dim3(256), 0, st, p);             \
    else if (p.dim <= 1024) hipLaunchKernelGGL((norm_kernel<ODT_, 4>), grid, dim3(256), 0, st, p);            \
    else hipLaunchKernelGGL((norm_kernel<ODT_, 8>), grid, dim3(256), 0, st, p);                               \
  } while (0)
  switch (p.out_dtype) {
    case CV_F32: NORM_LAUNCH(CV_F32); break;
    case CV_BF16: NORM_LAUNCH(CV_BF16); break;
    case CV_F16: NORM_LAUNCH(CV_F16); break;
    default: return CV_ERR_ARG;
  }
#undef NORM_LAUNCH
  CV_CHECK_LAUNCH();
  return CV_OK;
}
