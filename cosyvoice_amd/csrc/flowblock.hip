// cv_tblock_head / cv_tblock_tail: the CFM estimator's BasicTransformerBlock (flow/components/transformer.py:243-316,
// FeedForward :83-134; diffusers Attention / GELU semantics restated in SURVEY.md §8c) as TWO row-block kernels around
// the flash-attention launch, instead of eight GEMM / LayerNorm launches:
//
//   head:  LN(norm1) -> [Q | K] row-major + V^T                      (was: layernorm, QK GEMM, V^T GEMM)
//   tail:  to_out + bias + residual -> LN(norm3) -> Linear(256->1024) + erf-GELU -> Linear(1024->256) + residual
//                                                                    (was: GEMM, layernorm, GEMM, GEMM)
//
// Why: at these shapes (K = 256 .. 1024, N = 256 .. 1536) a 64x64-tile GEMM workgroup lives ~2.5 us for 32 MFMAs per
// wave: the stage ran at 0.46 MFMA-busy with 58 % of wave cycles parked in s_waitcnt (profiles/r01_k_flow_pmc_kernels.csv),
// bound by operand delivery L2 -> CU, and every LN / GELU intermediate made a round trip through L2.  Here a workgroup owns
// 64 rows of one sequence for the whole chain:
//   * the activation tile (LN output / attention output / GELU output) lives in LDS as an MFMA operand image
//     (row-major, 16-byte chunk slot XOR-swizzled by row & 15: conflict-free ds_read_b128 fragments);
//   * weights are pre-packed in MFMA fragment order (cv_pack_skinny: 1 KiB contiguous per fragment) and stream
//     L2 -> VGPR directly, split over the 4 waves by OUTPUT COLUMN, so every weight byte enters the CU exactly once per
//     64 rows (64 flop per weight byte instead of 32 per operand byte) and never touches LDS;
//   * a two-slot register ring (8 fragments each) keeps two fragment groups in flight behind the MFMAs;
//   * the 1024-wide GELU intermediate never leaves the CU (128-column chunks through a double-buffered LDS tile),
//     the residual row stays in the accumulators from the out-projection to the final store, and LayerNorm(norm3)
//     is computed from those accumulators (cross-wave row statistics through 2 KiB of LDS).
#include "cv_device.h"

namespace {

constexpr int TB_C = 256, TB_INNER = 512, TB_FF = 1024;
constexpr int BM = 64;        // rows per workgroup (4 MFMA row tiles)
constexpr int HC = 128;       // hidden columns per FFN chunk

// 16-byte chunk slot of (row, chunk) in an operand image whose rows hold >= 16 chunks: XOR the low 4 chunk bits with row & 15.
// A ds_read_b128 fragment (lane l: row l & 15, chunk 4 ks + (l >> 4)) then hits 16 distinct slots per 16-lane group.
__device__ __forceinline__ int swz16(int row, int chunk) { return (chunk & ~15) | ((chunk ^ row) & 15); }

// exact-erf GELU (diffusers GELU(approximate="none")): erfc by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// 16-bit rounding of the result), negative side computed as 0.5 x erfc(|z|) so there is no 1 - erf cancellation.
// ~14 VALU instead of ocml erff's ~40 with branches: the FFN evaluates 1024 of these per row per block.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float erfc_abs = pl * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);   // erfc(|z|)
  const float cdf2 = x < 0.f ? erfc_abs : 2.0f - erfc_abs;                                  // 1 + erf(z)
  return 0.5f * x * cdf2;
}

template <int DT>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  return make_uint2(pack2<DT>(a, b), pack2<DT>(c, d));
}

// LayerNorm of this workgroup's 64 rows straight from global memory into the K = 256 operand image (512-byte rows):
// wave w normalises rows 16w .. 16w+15, one row per wave-instruction (lane = 4 consecutive columns), two-pass in registers.
// Rows beyond T re-read row T-1 (finite values; their results are never stored).
template <int DT>
__device__ __forceinline__ void ln_rows_to_lds(const float* xs, int ldx, int t0, int T, const float* gamma, const float* beta,
                                               float eps, char* img, int wid, int lane) {
  const float4 g4 = *(const float4*)(gamma + 4 * lane);
  const float4 b4 = *(const float4*)(beta + 4 * lane);
  float4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = min(t0 + wid * 16 + i, T - 1);
    v[i] = *(const float4*)(xs + (int64_t)t * ldx + 4 * lane);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float mean = wave_sum((v[i].x + v[i].y) + (v[i].z + v[i].w)) * (1.0f / TB_C);
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    const float rstd = rsqrtf(wave_sum((a * a + b * b) + (c * c + d * d)) * (1.0f / TB_C) + eps);
    const int row = wid * 16 + i;
    *(uint2*)(img + row * 512 + (swz16(row, lane >> 1) << 4) + ((lane & 1) << 3)) =
        pack4<DT>(a * rstd * g4.x + b4.x, b * rstd * g4.y + b4.y, c * rstd * g4.z + b4.z, d * rstd * g4.w + b4.w);
  }
}

// ============================================================================================== head: LN -> Q | K | V^T
// Output columns: 1536 = 96 MFMA tiles (0..63 = [Q | K] row-major, 64..95 = V, stored transposed).  A wave-step is 4
// consecutive tiles x all 8 k-steps x the 4 row tiles (128 MFMAs); wave w takes tile groups w, w + 4, ... (6 steps).
// V steps swap the MFMA operands (D = xn . Wv^T has the frame index in the registers), so a lane holds 4 consecutive
// frames of one (head, channel) row of V^T: 8-byte stores, no transposing pass.
template <int DT>
__global__ __launch_bounds__(256, 2) void tblock_head_kernel(const cv_tblock_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const int r = blockIdx.y, t0 = blockIdx.x * BM;
  const float* xs = p.x + (int64_t)r * p.T * p.ldx;
  constexpr int NKS = TB_C / 32;   // 8

  const uint4* Wl = (const uint4*)p.wqkv_p + lane;
  uint4 s0[8], s1[8];
  auto ld = [&](uint4 (&s)[8], int tile0, int kq) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) s[j * 2 + u] = Wl[((tile0 + j) * NKS + kq * 2 + u) * 64];
  };
  ld(s0, wid * 4, 0);   // the first two fragment groups fly during the LayerNorm
  ld(s1, wid * 4, 1);

  ln_rows_to_lds<DT>(xs, p.ldx, t0, p.T, p.g1, p.b1n, p.eps, smem, wid, lane);
  __syncthreads();

  int aoff[4];   // byte offset of this lane's fragment chunk for k-step 0 of row tile i (k-step ks: chunk 4 ks + lg)
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (16 * i + lq) * 512;

  f32x4_t acc[4][4];
  auto compute = [&](uint4 (&s)[8], int kq, bool vpart) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ks = kq * 2 + u;
      uint4 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(smem + aoff[i] + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][j] = vpart ? mfma_block<DT>(a[i], s[j * 2 + u], acc[i][j]) : mfma_block<DT>(s[j * 2 + u], a[i], acc[i][j]);
    }
  };

  constexpr int NSTEP = 3 * TB_INNER / 16 / 16;   // 96 tiles / (4 waves x 4 tiles) = 6
  uint16_t* qk = (uint16_t*)p.qk + (int64_t)r * p.T * p.ldqk;
  uint16_t* vt = (uint16_t*)p.vt + (int64_t)r * (TB_INNER / 64) * 64 * p.vt_ld;
  for (int st = 0; st < NSTEP; ++st) {
    const int tile0 = (st * 4 + wid) * 4;
    const int ntile0 = (min(st + 1, NSTEP - 1) * 4 + wid) * 4;
    const bool vpart = tile0 >= (2 * TB_INNER) / 16;   // wave-uniform
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (!vpart) {
      compute(s0, 0, false); ld(s0, tile0, 2);
      compute(s1, 1, false); ld(s1, tile0, 3);
      compute(s0, 2, false); ld(s0, ntile0, 0);
      compute(s1, 3, false); ld(s1, ntile0, 1);
      // D rows = output column: lane holds row m = 16 i + lq, columns n .. n + 3
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int t = t0 + 16 * i + lq;
        if (t >= p.T) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(uint2*)(qk + (int64_t)t * p.ldqk + (tile0 + j) * 16 + 4 * lg) = pack4<DT>(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    } else {
      compute(s0, 0, true); ld(s0, tile0, 2);
      compute(s1, 1, true); ld(s1, tile0, 3);
      compute(s0, 2, true); ld(s0, ntile0, 0);
      compute(s1, 3, true); ld(s1, ntile0, 1);
      // D rows = frame: lane holds V^T row (head * 64 + d) = column tile * 16 + lq, frames t .. t + 3
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vrow = (tile0 + j) * 16 + lq - 2 * TB_INNER;
        uint16_t* dst = vt + (int64_t)vrow * p.vt_ld;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int t = t0 + 16 * i + 4 * lg;
          if (t + 3 < p.T) {
            *(uint2*)(dst + t) = pack4<DT>(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (t + e < p.T) dst[t + e] = Elem16<DT>::from_f32(acc[i][j][e]);
          }
        }
      }
    }
  }
}

// ============================================================================================== tail: to_out + LN + FFN
// LDS: [0, 64 K) attention-output image (64 rows x 1024 B) during the out-projection, afterwards xn image (64 x 512 B) at 0
// and the two GELU chunk tiles (64 x 256 B each) at 32 K / 48 K; [64 K, 66 K) cross-wave row statistics; [66 K, 70 K) bf1.
// Wave w owns output columns [64 w, 64 w + 64) of the 256-wide residual row for the whole kernel (acc2: 4 x 4 tiles).
template <int DT, bool OUTPROJ>
__global__ __launch_bounds__(256, 2) void tblock_tail_kernel(const cv_tblock_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ximg = smem;
  char* himg = smem + 32768;
  float* red = (float*)(smem + 65536);   // [2][4 waves][64 rows]
  const float* b1s = (const float*)(smem + 65536 + 2048);   // hidden-layer bias, staged once (a global load inside the chunk loop
                                                            // is sunk by hipcc to its use and drains the weight prefetch: vmcnt(0))
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const int r = blockIdx.y, t0 = blockIdx.x * BM;
  float* xs = p.x + (int64_t)r * p.T * p.ldx;
  const int ncol0 = wid * 64;
  ((float4*)(smem + 65536 + 2048))[tid] = ((const float4*)p.bf1)[tid];   // 1024 floats; visible after the first barrier

  const uint4* W1l = (const uint4*)p.w1_p + lane;   // [64 tiles][8 ks]
  const uint4* W2l = (const uint4*)p.w2_p + lane;   // [16 tiles][32 ks]
  uint4 s0[8], s1[8];
  // FFN fragment groups.  GEMM1 (chunk c, half hf): hidden tiles 8 c + 2 w + j (j < 2), k-steps 4 hf + u (u < 4): s[j * 4 + u]
  auto ld_g1 = [&](uint4 (&s)[8], int c, int hf) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) s[j * 4 + u] = W1l[((8 * c + 2 * wid + j) * (TB_C / 32) + 4 * hf + u) * 64];
  };
  // GEMM2 (chunk c, half hf): output tiles 4 w + j (j < 4), k-steps 4 c + 2 hf + u (u < 2): s[j * 2 + u]
  auto ld_g2 = [&](uint4 (&s)[8], int c, int hf) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) s[j * 2 + u] = W2l[((4 * wid + j) * (TB_FF / 32) + 4 * c + 2 * hf + u) * 64];
  };

  // residual rows in the accumulator layout (row 16 i + lq, columns ncol0 + 16 j + 4 lg ..+3)
  f32x4_t acc2[4][4];
  auto load_residual = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = min(t0 + 16 * i + lq, p.T - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *(const float4*)(xs + (int64_t)t * p.ldx + ncol0 + 16 * j + 4 * lg);
        acc2[i][j] = f32x4_t{v.x, v.y, v.z, v.w};
      }
    }
  };

  if constexpr (OUTPROJ) {
    // ---- attention output tile -> LDS operand image (1024-byte rows), one row per wave-instruction
    const uint16_t* aos = (const uint16_t*)p.ao + (int64_t)r * p.T * p.ldao;
    const uint4* Wol = (const uint4*)p.wo_p + lane;   // [16 tiles][16 ks]
    auto ld_o = [&](uint4 (&s)[8], int g) {           // group g: tiles 4 w + j, k-steps 2 g + u
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u) s[j * 2 + u] = Wol[((4 * wid + j) * (TB_INNER / 32) + 2 * g + u) * 64];
    };
    {
      uint4 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t = min(t0 + wid * 16 + i, p.T - 1);
        v[i] = *(const uint4*)(aos + (int64_t)t * p.ldao + lane * 8);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = wid * 16 + i;
        *(uint4*)(smem + row * 1024 + (swz16(row, lane) << 4)) = v[i];
      }
    }
    // (the 16 staging registers are dead before the residual rows and the first weight groups are requested: 256-VGPR budget)
    load_residual();
    ld_o(s0, 0);
    ld_o(s1, 1);
    __syncthreads();
    auto compute_o = [&](uint4 (&s)[8], int g) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ks = 2 * g + u;
        uint4 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(smem + (16 * i + lq) * 1024 + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc2[i][j] = mfma_block<DT>(s[j * 2 + u], a[i], acc2[i][j]);
      }
    };
    compute_o(s0, 0); ld_o(s0, 2);
    compute_o(s1, 1); ld_o(s1, 3);
    compute_o(s0, 2); ld_o(s0, 4);
    compute_o(s1, 3); ld_o(s1, 5);
    compute_o(s0, 4); ld_o(s0, 6);
    compute_o(s1, 5); ld_o(s1, 7);
    compute_o(s0, 6); ld_g1(s0, 0, 0);
    compute_o(s1, 7); ld_g1(s1, 0, 1);

    // ---- + to_out bias: acc2 is now the block's first residual output x1; LayerNorm(norm3) from the accumulators
    float4 gam[4], bet[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = ncol0 + 16 * j + 4 * lg;
      const float4 bo = *(const float4*)(p.bo + n);
      gam[j] = *(const float4*)(p.g3 + n);
      bet[j] = *(const float4*)(p.b3n + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc2[i][j][0] += bo.x; acc2[i][j][1] += bo.y; acc2[i][j][2] += bo.z; acc2[i][j][3] += bo.w; }
    }
    float mean[4], rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += (acc2[i][j][0] + acc2[i][j][1]) + (acc2[i][j][2] + acc2[i][j][3]);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lg == 0) red[wid * 64 + 16 * i + lq] = s;
    }
    __syncthreads();   // also: every wave is done reading the attention-output image
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * i + lq;
      mean[i] = ((red[m] + red[64 + m]) + (red[128 + m] + red[192 + m])) * (1.0f / TB_C);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = acc2[i][j][e] - mean[i]; q += d * d; }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (lg == 0) red[256 + wid * 64 + m] = q;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * i + lq;
      rstd[i] = rsqrtf(((red[256 + m] + red[320 + m]) + (red[384 + m] + red[448 + m])) * (1.0f / TB_C) + p.eps);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = ncol0 + 16 * j + 4 * lg;   // 4 consecutive columns = half of 16-byte chunk n >> 3
        const float sc = rstd[i];
        *(uint2*)(ximg + m * 512 + (swz16(m, n >> 3) << 4) + ((lg & 1) << 3)) =
            pack4<DT>((acc2[i][j][0] - mean[i]) * sc * gam[j].x + bet[j].x, (acc2[i][j][1] - mean[i]) * sc * gam[j].y + bet[j].y,
                      (acc2[i][j][2] - mean[i]) * sc * gam[j].z + bet[j].z, (acc2[i][j][3] - mean[i]) * sc * gam[j].w + bet[j].w);
      }
    }
  } else {
    ld_g1(s0, 0, 0);
    ld_g1(s1, 0, 1);
    load_residual();
    ln_rows_to_lds<DT>(xs, p.ldx, t0, p.T, p.g3, p.b3n, p.eps, ximg, wid, lane);
  }
  // + FFN output bias: the accumulators then collect x1 + b2 + sum_c gelu(..) . W2_c^T
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 b2 = *(const float4*)(p.bf2 + ncol0 + 16 * j + 4 * lg);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc2[i][j][0] += b2.x; acc2[i][j][1] += b2.y; acc2[i][j][2] += b2.z; acc2[i][j][3] += b2.w; }
  }
  __syncthreads();   // xn image complete

  constexpr int NC = TB_FF / HC;   // 8 chunks
  for (int c = 0; c < NC; ++c) {
    const int cn = min(c + 1, NC - 1);
    char* hb = himg + (c & 1) * 16384;
    // the accumulators of the hidden layer start from its bias (this wave's two tiles of the chunk)
    f32x4_t acc1[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 b1v = *(const float4*)(b1s + c * HC + (2 * wid + j) * 16 + 4 * lg);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc1[i][j] = f32x4_t{b1v.x, b1v.y, b1v.z, b1v.w};
    }
    auto compute_g1 = [&](uint4 (&s)[8], int hf) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ks = 4 * hf + u;
        uint4 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(ximg + (16 * i + lq) * 512 + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc1[i][j] = mfma_block<DT>(s[j * 4 + u], a[i], acc1[i][j]);
      }
    };
    auto compute_g2 = [&](uint4 (&s)[8], int hf) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ks = 2 * hf + u;   // k-step inside the chunk (4 x 32 = 128 hidden columns)
        uint4 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(hb + (16 * i + lq) * 256 + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc2[i][j] = mfma_block<DT>(s[j * 2 + u], a[i], acc2[i][j]);
      }
    };
    compute_g1(s0, 0); ld_g2(s0, c, 0);
    compute_g1(s1, 1); ld_g2(s1, c, 1);
    // bias + GELU -> 16-bit chunk tile (row 16 i + lq, hidden columns (2 w + j) * 16 + 4 lg ..+3 of the chunk)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * i + lq;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ch = (2 * wid + j) * 2 + (lg >> 1);
        *(uint2*)(hb + m * 256 + (swz16(m, ch) << 4) + ((lg & 1) << 3)) =
            pack4<DT>(gelu_erf(acc1[i][j][0]), gelu_erf(acc1[i][j][1]), gelu_erf(acc1[i][j][2]), gelu_erf(acc1[i][j][3]));
      }
    }
    __syncthreads();   // one barrier per chunk: the next chunk's GELU tile goes to the other buffer
    compute_g2(s0, 0); ld_g1(s0, cn, 0);
    compute_g2(s1, 1); ld_g1(s1, cn, 1);
  }

  // ---- store the block output (fp32 residual stream, in place) + optional 16-bit copy (skip connection / next conv operand)
  uint16_t* oa = p.out_act ? (uint16_t*)p.out_act + (int64_t)r * p.T * p.ldoa : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + 16 * i + lq;
    if (t >= p.T) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = ncol0 + 16 * j + 4 * lg;
      *(float4*)(xs + (int64_t)t * p.ldx + n) = make_float4(acc2[i][j][0], acc2[i][j][1], acc2[i][j][2], acc2[i][j][3]);
      if (oa) *(uint2*)(oa + (int64_t)t * p.ldoa + n) = pack4<DT>(acc2[i][j][0], acc2[i][j][1], acc2[i][j][2], acc2[i][j][3]);
    }
  }
}

int check_common(const cv_tblock_params& p) {
  if (p.dtype != CV_BF16 && p.dtype != CV_F16) return CV_ERR_UNSUPPORTED;
  if (p.C != TB_C || p.inner != TB_INNER || p.ff != TB_FF) return CV_ERR_UNSUPPORTED;
  if (p.R <= 0 || p.T <= 0 || !p.x || (p.ldx & 3) || ((uintptr_t)p.x & 15)) return CV_ERR_ARG;
  return CV_OK;
}

}  // namespace

extern "C" int cv_sizeof_tblock_params(void) { return (int)sizeof(cv_tblock_params); }

extern "C" int cv_tblock_head(const cv_tblock_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  const cv_tblock_params p = *pp;
  if (int rc = check_common(p)) return rc;
  if (!p.g1 || !p.b1n || !p.wqkv_p || !p.qk || !p.vt) return CV_ERR_ARG;
  if ((p.ldqk & 3) || (p.vt_ld & 3) || p.vt_ld < p.T || ((uintptr_t)p.qk & 7) || ((uintptr_t)p.vt & 7) || ((uintptr_t)p.wqkv_p & 15))
    return CV_ERR_ARG;
  dim3 grid((p.T + BM - 1) / BM, p.R);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 32768;
  if (p.dtype == CV_BF16) hipLaunchKernelGGL(tblock_head_kernel<CV_BF16>, grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(tblock_head_kernel<CV_F16>, grid, dim3(256), lds, st, p);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_tblock_tail(const cv_tblock_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  const cv_tblock_params p = *pp;
  if (int rc = check_common(p)) return rc;
  if (!p.g3 || !p.b3n || !p.w1_p || !p.bf1 || !p.w2_p || !p.bf2) return CV_ERR_ARG;
  if (((uintptr_t)p.w1_p & 15) || ((uintptr_t)p.w2_p & 15)) return CV_ERR_ARG;
  if (p.out_act && ((p.ldoa & 3) || ((uintptr_t)p.out_act & 7))) return CV_ERR_ARG;
  const bool outproj = p.ao != nullptr;
  if (outproj && (!p.wo_p || !p.bo || (p.ldao & 7) || ((uintptr_t)p.ao & 15) || ((uintptr_t)p.wo_p & 15))) return CV_ERR_ARG;
  dim3 grid((p.T + BM - 1) / BM, p.R);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 65536 + 2048 + 4096;
  static bool attr_set = false;
  if (!attr_set) {   // > 64 KiB of dynamic LDS needs the opt-in
    hipFuncSetAttribute((const void*)tblock_tail_kernel<CV_BF16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)tblock_tail_kernel<CV_F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)tblock_tail_kernel<CV_BF16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)tblock_tail_kernel<CV_F16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  if (p.dtype == CV_BF16) {
    if (outproj) hipLaunchKernelGGL((tblock_tail_kernel<CV_BF16, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((tblock_tail_kernel<CV_BF16, false>), grid, dim3(256), lds, st, p);
  } else {
    if (outproj) hipLaunchKernelGGL((tblock_tail_kernel<CV_F16, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((tblock_tail_kernel<CV_F16, false>), grid, dim3(256), lds, st, p);
  }
  CV_CHECK_LAUNCH();
  return CV_OK;
}
