// cv_gemm: LDS-tiled MFMA GEMM / implicit-im2col Conv1d for gfx950 (wave64).
//
// C[m][n] = epilogue( sum_k A[row(m,k)][ci(k)] * W[n][k] ),  A channels-last activations, W = [N][K] (torch Linear /
// repacked Conv1d layout).  256 threads = 4 waves (2x2); each wave owns a (BM/2)x(BN/2) patch of 16x16 MFMA tiles.
// K is consumed in tiles of 128 bytes per row (64 x 16-bit or 32 x f32): global -> registers (16-byte chunks, 8 lanes
// per row = one full 128-byte line) -> LDS -> ds_read_b128 fragments -> MFMA.
// LDS image: each (16 rows x 4 chunks) block is stored [chunk][row][16 B], so lane l of a wave reads bytes
// [16 l, 16 l + 16) of the block: linear, conflict-free for ds_read_b128 (MI355X_MICROARCH.md §LDS).
// One LDS stage + register prefetch: the next tile's global loads fly during the MFMAs; 32 KiB of LDS per 128x128
// workgroup keeps 4 workgroups resident per CU.
#include "cv_device.h"
#include <cstdlib>

namespace {

template <int DT> struct ElemSize { static constexpr int value = (DT == CV_F32) ? 4 : 2; };

__device__ __forceinline__ int lds_chunk_off(int row, int kc) {
  return (((row >> 4) * 2 + (kc >> 2)) << 10) + ((kc & 3) << 8) + ((row & 15) << 4);
}

template <int DT>
__device__ __forceinline__ void store_act(void* base, int64_t idx, float v) {
  if constexpr (DT == CV_F32) {
    ((float*)base)[idx] = v;
  } else {
    ((uint16_t*)base)[idx] = Elem16<DT>::from_f32(v);
  }
}

template <int DT>
__device__ __forceinline__ void store_act4(void* base, int64_t idx, float a, float b, float c, float d) {
  if constexpr (DT == CV_F32) {
    *(float4*)((float*)base + idx) = make_float4(a, b, c, d);
  } else {
    uint2 u;
    u.x = pack2<DT>(a, b);
    u.y = pack2<DT>(c, d);
    *(uint2*)((uint16_t*)base + idx) = u;
  }
}

template <int DT, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const cv_gemm_params p) {
  constexpr int ES = ElemSize<DT>::value;
  constexpr int CH = 16 / ES;    // elements per 16-byte chunk
  constexpr int BK = 128 / ES;   // elements per K tile
  constexpr int MT = BM / 32, NT = BN / 32;
  constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;
  constexpr int STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wave_m = wid >> 1, wave_n = wid & 1;

  const int mtiles = (p.M + BM - 1) / BM;
  const int tile_m = blockIdx.x % mtiles, tile_n = blockIdx.x / mtiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const int z1 = z / p.batch_inner, z0 = z - z1 * p.batch_inner;

  const char* Ab = (const char*)p.A + (z0 * p.a_bs0 + z1 * p.a_bs1) * ES;
  const char* Wb = (const char*)p.W + (z0 * p.w_bs0 + z1 * p.w_bs1) * ES;

  const int nk = (p.K + BK - 1) / BK;
  const bool conv = p.cin != p.K;

  uint4 ra[A_CH], rb[B_CH];
  uint32_t amask = 0, bmask = 0;  // validity bits of the prefetched chunks; applied when the tile is written to LDS

  // Tile loads are UNCONDITIONAL (addresses clamped into the operand, out-of-range chunks masked to zero afterwards):
  // `if (ok) v = load` makes hipcc branch around every load and wait vmcnt(0) each time, which serialises the K loop
  // on memory latency (cdna_hip_programming.md §5 "Three .s-level traps" (c)).
  const int a_row_max = p.a_rows - 1, n_max = p.N - 1, k_max = p.K - CH;
  auto load_tile = [&](int kt) {
    const int kbase = kt * BK;
    amask = 0;
    bmask = 0;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int c = i * 256 + tid;
      const int row = c >> 3, kc = c & 7;
      const int m = m0 + row;
      const int k = kbase + kc * CH;
      const int kcl = min(k, k_max);
      int tap = 0, ci = kcl;
      if (conv) { tap = kcl / p.cin; ci = kcl - tap * p.cin; }
      const int arow = m * p.a_row_stride + p.tap_base + tap * p.tap_step;
      const bool ok = (m < p.M) && (k < p.K) && (arow >= 0) && (arow <= a_row_max);
      const int arc = min(max(arow, 0), a_row_max);
      ra[i] = *(const uint4*)(Ab + ((int64_t)arc * p.lda + ci) * ES);
      amask |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int c = i * 256 + tid;
      const int row = c >> 3, kc = c & 7;
      const int n = n0 + row;
      const int k = kbase + kc * CH;
      const bool ok = (n < p.N) && (k < p.K);
      rb[i] = *(const uint4*)(Wb + ((int64_t)min(n, n_max) * p.ldw + min(k, k_max)) * ES);
      bmask |= (ok ? 1u : 0u) << i;
    }
  };
  auto store_tile = [&](int s) {
    char* sa = smem + s * STAGE;
    char* sb = sa + BM * 128;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int c = i * 256 + tid;
      const uint32_t mk = ((amask >> i) & 1u) ? 0xFFFFFFFFu : 0u;
      *(uint4*)(sa + lds_chunk_off(c >> 3, c & 7)) = make_uint4(ra[i].x & mk, ra[i].y & mk, ra[i].z & mk, ra[i].w & mk);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int c = i * 256 + tid;
      const uint32_t mk = ((bmask >> i) & 1u) ? 0xFFFFFFFFu : 0u;
      *(uint4*)(sb + lds_chunk_off(c >> 3, c & 7)) = make_uint4(rb[i].x & mk, rb[i].y & mk, rb[i].z & mk, rb[i].w & mk);
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // single LDS stage (32 KiB at 128x128 -> 4 workgroups per CU) + register prefetch of the next K tile:
  // occupancy, not a second LDS buffer, hides the global-load latency of these short-K GEMMs
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* sa = smem;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *(const uint4*)(sa + (((wave_m * MT + i) * 2 + ks) << 10) + lane * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *(const uint4*)(sb + (((wave_n * NT + j) * 2 + ks) << 10) + lane * 16);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_block<DT>(fb[j], fa[i], acc[i][j]);
    }
    if (kt + 1 < nk) {
      __syncthreads();
      store_tile(0);
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ epilogue
  // acc[i][j][r]: m = m0 + (wave_m*MT + i)*16 + (lane&15);  n = n0 + (wave_n*NT + j)*16 + 4*(lane>>4) + r
  // All bias / residual / act-param loads are issued UNCONDITIONALLY (clamped addresses, results masked) and hoisted in
  // front of the arithmetic: a load inside a per-element branch makes hipcc wait vmcnt(0) per element
  // (cdna_hip_programming.md §5 "Three .s-level traps" (c)) — that serialisation dominated the first version.
  const int lm = lane & 15, lg = lane >> 4;
  const int64_t res_off = z0 * p.res_bs0 + z1 * p.res_bs1;
  float* o32 = p.out_f32 ? p.out_f32 + (z0 * p.o32_bs0 + z1 * p.o32_bs1) : nullptr;
  char* oact = p.out_act ? (char*)p.out_act + (z0 * p.oa_bs0 + z1 * p.oa_bs1) * ES : nullptr;
  const bool vec = ((p.N & 3) == 0) && (p.act != CV_ACT_SWIGLU) && (p.out_mode == CV_OUT_ROWMAJOR);

  if (vec) {
    const bool has_res = p.res != nullptr, has_res2 = p.res2 != nullptr;
    float4 b4[NT], ap4[NT];
    int nbj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nb = n0 + (wave_n * NT + j) * 16 + 4 * lg;
      nbj[j] = nb;
      const int nc = nb < p.N ? nb : 0;  // N % 4 == 0: a 4-group is entirely in or out
      b4[j] = p.bias ? *(const float4*)(p.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
      ap4[j] = p.act_param ? *(const float4*)(p.act_param + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int orow[MT];
    bool rok[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + (wave_m * MT + i) * 16 + lm;
      const int r_ = m * p.out_row_stride + p.out_row_off;
      rok[i] = (m < p.M) && (r_ >= 0) && (r_ < p.out_rows);
      orow[i] = rok[i] ? r_ : (p.out_row_off >= 0 && p.out_row_off < p.out_rows ? p.out_row_off : 0);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float4 r4[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nc = nbj[j] < p.N ? nbj[j] : 0;
        r4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_res) r4[j] = *(const float4*)(p.res + res_off + (int64_t)orow[i] * p.ldres + nc);
      }
      if (has_res2) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int nc = nbj[j] < p.N ? nbj[j] : 0;
          const float4 q = *(const float4*)(p.res2 + res_off + (int64_t)orow[i] * p.ldres2 + nc);
          r4[j].x += q.x; r4[j].y += q.y; r4[j].z += q.z; r4[j].w += q.w;
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (!rok[i] || nbj[j] >= p.N) continue;
        const float v0 = (acc[i][j][0] + b4[j].x + r4[j].x) * p.out_scale;
        const float v1 = (acc[i][j][1] + b4[j].y + r4[j].y) * p.out_scale;
        const float v2 = (acc[i][j][2] + b4[j].z + r4[j].z) * p.out_scale;
        const float v3 = (acc[i][j][3] + b4[j].w + r4[j].w) * p.out_scale;
        if (o32) *(float4*)(o32 + (int64_t)orow[i] * p.ldo32 + nbj[j]) = make_float4(v0, v1, v2, v3);
        if (oact)
          store_act4<DT>(oact, (int64_t)orow[i] * p.ldoa + nbj[j], apply_act(p.act, v0, ap4[j].x, p.act_slope),
                         apply_act(p.act, v1, ap4[j].y, p.act_slope), apply_act(p.act, v2, ap4[j].z, p.act_slope),
                         apply_act(p.act, v3, ap4[j].w, p.act_slope));
      }
    }
    return;
  }

  // ---- general path (N % 4 != 0, SwiGLU pairs, QKV split): correctness first, used by a handful of small launches
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + (wave_m * MT + i) * 16 + lm;
    if (m >= p.M) continue;
    const int orow = m * p.out_row_stride + p.out_row_off;
    if (orow < 0 || orow >= p.out_rows) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nb = n0 + (wave_n * NT + j) * 16 + 4 * lg;
      if (nb >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      float ap[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        if (n < p.N) {
          if (p.bias) v[r] += p.bias[n];
          if (p.res) v[r] += p.res[res_off + (int64_t)orow * p.ldres + n];
          if (p.res2) v[r] += p.res2[res_off + (int64_t)orow * p.ldres2 + n];
          v[r] *= p.out_scale;
          if (p.act_param) ap[r] = p.act_param[n];
        }
      }
      if (p.out_mode == CV_OUT_QKV) {
        // 16-bit outputs only; boundaries are multiples of 64 so a 4-group never straddles
        if (nb < p.q_cols) {
          store_act4<DT>(oact, (int64_t)orow * p.ldoa + nb, v[0] * p.q_scale, v[1] * p.q_scale, v[2] * p.q_scale, v[3] * p.q_scale);
        } else if (nb < p.q_cols + p.k_cols) {
          char* kb = (char*)p.k_out + (int64_t)z * p.k_bs * ES;
          store_act4<DT>(kb, (int64_t)orow * p.ldk + (nb - p.q_cols), v[0], v[1], v[2], v[3]);
        } else {
          const int c = nb - p.q_cols - p.k_cols;
          const int h = c >> 6, d = c & 63;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            store_act<DT>(p.vt_out, ((int64_t)(z * p.vt_heads + h) * 64 + d + r) * p.vt_ld + orow, v[r]);
        }
        continue;
      }
      if (p.act == CV_ACT_SWIGLU) {
        // weights interleaved in 16-row blocks [gate16 | up16]: even j = gate, odd j = up (same lanes, same regs)
        if ((j & 1) == 0) continue;
        const int ncol = ((nb >> 5) << 4) + (nb & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nb + r >= p.N) continue;
          const float g = (acc[i][j > 0 ? j - 1 : 0][r] + (p.bias ? p.bias[nb - 16 + r] : 0.f)) * p.out_scale;
          const float hval = act_silu(g) * v[r];
          if (o32) o32[(int64_t)orow * p.ldo32 + ncol + r] = hval;
          if (oact) store_act<DT>(oact, (int64_t)orow * p.ldoa + ncol + r, hval);
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (nb + r >= p.N) continue;
        if (o32) o32[(int64_t)orow * p.ldo32 + nb + r] = v[r];
        if (oact) store_act<DT>(oact, (int64_t)orow * p.ldoa + nb + r, apply_act(p.act, v[r], ap[r], p.act_slope));
      }
    }
  }
}

template <int DT, int BM, int BN>
int launch(const cv_gemm_params& p, hipStream_t st) {
  const int mt = (p.M + BM - 1) / BM, nt = (p.N + BN - 1) / BN;
  dim3 grid(mt * nt, 1, p.batch);
  const size_t lds = (BM + BN) * 128;
  hipLaunchKernelGGL((gemm_kernel<DT, BM, BN>), grid, dim3(256), lds, st, p);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

// CV_GEMM_TILE=0|1|2 (128x128 | 128x64 | 64x64) overrides the heuristic: tuning aid only
static int g_tile_override = -2;

template <int DT>
int dispatch(const cv_gemm_params& p, hipStream_t st) {
  if (g_tile_override == -2) {
    const char* e = getenv("CV_GEMM_TILE");
    g_tile_override = e ? atoi(e) : -1;
  }
  const bool swiglu = p.act == CV_ACT_SWIGLU;
  int tile = g_tile_override;
  if (tile < 0) {
    // These GEMMs have short K (256..1024): latency is hidden by resident workgroups, not by pipeline depth, so prefer
    // the largest tile that still puts >= ~4 workgroups on every CU.
    // measured on MI355X (tools/gemm_bench.py, profiles/r01_gemm_tile_sweep.txt): 64x64 wins for K <= 512 and whenever
    // 128x64 would leave < ~3 workgroups per CU; 128x64 wins for long K with many tiles; 128x128 never wins here.
    const long long t12864 = (long long)((p.M + 127) / 128) * ((p.N + 63) / 64) * p.batch;
    if (p.K > 512 && t12864 >= 768) tile = 1;
    else tile = 2;
  }
  if (swiglu && tile == 0) tile = 1;
  if (tile == 0) return launch<DT, 128, 128>(p, st);
  if (tile == 1) return launch<DT, 128, 64>(p, st);
  return launch<DT, 64, 64>(p, st);
}

}  // namespace

extern "C" int cv_gemm(const cv_gemm_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  cv_gemm_params p = *pp;
  const int es = p.dtype == CV_F32 ? 4 : 2;
  const int ch = 16 / es;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0) return CV_ERR_ARG;
  if (!p.A || !p.W || (!p.out_f32 && !p.out_act)) return CV_ERR_ARG;
  if (p.batch_inner <= 0) p.batch_inner = p.batch;
  if (p.cin <= 0) p.cin = p.K;
  if (p.a_row_stride == 0) p.a_row_stride = 1;
  if (p.out_row_stride == 0) p.out_row_stride = 1;
  if (p.out_rows <= 0) p.out_rows = p.M * p.out_row_stride + p.out_row_off;
  if (p.a_rows <= 0) p.a_rows = p.M;
  if (p.out_scale == 0.f) p.out_scale = 1.f;
  // 16-byte chunk alignment rules of the loader
  if ((p.K % ch) || (p.cin % ch) || (p.lda % ch) || (p.ldw % ch)) return CV_ERR_ARG;
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return CV_ERR_ARG;
  if ((p.a_bs0 % ch) || (p.a_bs1 % ch) || (p.w_bs0 % ch) || (p.w_bs1 % ch)) return CV_ERR_ARG;
  if (p.K % p.cin) return CV_ERR_ARG;
  if (p.out_mode == CV_OUT_QKV) {
    if (p.dtype == CV_F32 || !p.out_act || !p.k_out || !p.vt_out) return CV_ERR_ARG;
    if ((p.q_cols & 63) || (p.k_cols & 63) || ((p.N - p.q_cols - p.k_cols) & 63)) return CV_ERR_ARG;
    if ((p.ldoa & 3) || (p.ldk & 3)) return CV_ERR_ARG;
    if (p.q_scale == 0.f) p.q_scale = 1.f;
  } else if (p.act == CV_ACT_SWIGLU) {
    if (p.N & 31) return CV_ERR_ARG;
  } else if ((p.N & 3) == 0) {
    // vector epilogue needs 16-byte (fp32) / 8-byte (16-bit) aligned rows
    if (p.out_f32 && ((p.ldo32 & 3) || ((uintptr_t)p.out_f32 & 15) || (p.o32_bs0 & 3) || (p.o32_bs1 & 3))) return CV_ERR_ARG;
    if (p.out_act && ((p.ldoa & 3) || ((uintptr_t)p.out_act & 15) || (p.oa_bs0 & 3) || (p.oa_bs1 & 3))) return CV_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (p.dtype) {
    case CV_F32: return dispatch<CV_F32>(p, st);
    case CV_BF16: return dispatch<CV_BF16>(p, st);
    case CV_F16: return dispatch<CV_F16>(p, st);
    default: return CV_ERR_ARG;
  }
}
