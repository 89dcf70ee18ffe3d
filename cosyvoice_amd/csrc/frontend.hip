// Prompt-feature front half: the two element-wise steps around the STFT / mel GEMMs of
// mel_spectrogram (cosyvoice/dataset/processor_kaldidata.py:37-74).  The DFT (windowed cos/sin basis) and the mel
// projection run on cv_gemm; these kernels are the |.| between them and the log-compression + layout change after.
#include "cv_device.h"

namespace {

// spec [rows][ld_spec] = [re(0..nbins) | im(0..nbins) | pad]  ->  mag [rows][ld_mag] = sqrt(re^2 + im^2 + eps), pad = 0
__global__ __launch_bounds__(256) void stft_mag_kernel(const float* spec, int ld_spec, float* mag, int ld_mag, int rows, int nbins, float eps) {
  const int r = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ld_mag) return;
  float v = 0.f;
  if (k < nbins) {
    const float re = spec[(int64_t)r * ld_spec + k], im = spec[(int64_t)r * ld_spec + nbins + k];
    v = sqrtf(re * re + im * im + eps);
  }
  mag[(int64_t)r * ld_mag + k] = v;
}

// mel [B][T][ld] -> out [B][n_mels][T] = log(max(mel, clip))   (dynamic_range_compression_torch, :27-28)
__global__ __launch_bounds__(256) void log_clamp_cf_kernel(const float* mel, int ld, float* out, int T, int n_mels, float clip) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, m = m0 + tx;
    tile[i][tx] = (t < T && m < n_mels) ? mel[((int64_t)b * T + t) * ld + m] : 1.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int m = m0 + i, t = t0 + tx;
    if (m < n_mels && t < T) out[((int64_t)b * n_mels + m) * T + t] = logf(fmaxf(tile[tx][i], clip));
  }
}

}  // namespace

extern "C" int cv_stft_magnitude(const float* spec, int32_t ld_spec, float* mag, int32_t ld_mag, int32_t rows, int32_t nbins, float eps,
                                 void* stream) {
  if (!spec || !mag || rows <= 0 || nbins <= 0 || ld_spec < 2 * nbins || ld_mag < nbins || rows > 65535) return CV_ERR_ARG;
  hipLaunchKernelGGL(stft_mag_kernel, dim3((ld_mag + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream, spec, ld_spec, mag, ld_mag,
                     rows, nbins, eps);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_log_clamp_channels_first(const float* mel, int32_t ld, float* out, int32_t B, int32_t T, int32_t n_mels, float clip,
                                           void* stream) {
  if (!mel || !out || B <= 0 || T <= 0 || n_mels <= 0 || ld < n_mels || B > 65535) return CV_ERR_ARG;
  hipLaunchKernelGGL(log_clamp_cf_kernel, dim3((T + 31) / 32, (n_mels + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, mel, ld, out, T,
                     n_mels, clip);
  CV_CHECK_LAUNCH();
  return CV_OK;
}
