// cv_gemm: LDS-tiled MFMA GEMM / implicit-im2col Conv1d for gfx950 (wave64).
//
// C[m][n] = epilogue( sum_k A[row(m,k)][ci(k)] * W[n][k] ),  A channels-last activations, W = [N][K] (torch Linear /
// repacked Conv1d layout).  256 threads = 4 waves (2x2); each wave owns a (BM/2)x(BN/2) patch of 16x16 MFMA tiles.
// K is consumed in tiles of 128 bytes per row (64 x 16-bit or 32 x f32): global -> registers (16-byte chunks, 8 lanes
// per row = one full 128-byte line) -> LDS -> ds_read_b128 fragments -> MFMA.  A wave loads 8 rows x 8 chunks with
// lanes 0..7 on 8 different rows of one chunk column: each 8-lane ds_write_b128 group then covers 8 distinct 16-byte
// bank slots (the row-major lane order made every LDS write 8-way conflicted and the kernel LDS-bound).
// LDS image: each (16 rows x 4 chunks) block is stored [chunk][row][16 B], so lane l of a wave reads bytes
// [16 l, 16 l + 16) of the block: linear, conflict-free for ds_read_b128 (MI355X_MICROARCH.md §LDS).
// One LDS stage + register prefetch: the next tile's global loads fly during the MFMAs; 32 KiB of LDS per 128x128
// workgroup keeps 4 workgroups resident per CU.
#include "cv_device.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int DT> struct ElemSize { static constexpr int value = (DT == CV_F32) ? 4 : 2; };

// LDS tile image: plain row-major, 128 bytes (8 x 16-byte chunks) per row, chunk position XOR-swizzled by (row>>1)&7.
//  * global loads stay row-contiguous (8 lanes = one full 128-byte line; the transposed lane order measured 1.6x slower),
//  * the 8-lane ds_write_b128 groups write 8 distinct 16-byte slots of one row: conflict-free,
//  * the MFMA fragment read (lane = row l&15, chunk ks*4 + (l>>4)) hits 16 distinct slots per ds_read_b128 lane group.
__device__ __forceinline__ int lds_chunk_off(int row, int kc) {
  return (row << 7) + ((kc ^ ((row >> 1) & 7)) << 4);
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

// fp32 values in the PRE-SPLIT storage of the bf16x3 path: every aligned group of 8 values (32 bytes) holds [8 x bf16 hi | 8 x bf16 lo]
// (hi = bf16(v), lo = bf16(v - hi): exactly what the consumer's LDS store would compute, computed once by the producer instead of
// once per tap and per tile column by every consumer).  A lane holds 4 consecutive values (idx % 4 == 0) = half a group.
__device__ __forceinline__ void split_pack4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  const float f[4] = {a, b, c, d};
  uint16_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = Elem16<CV_BF16>::from_f32(f[j]);
    l[j] = Elem16<CV_BF16>::from_f32(f[j] - Elem16<CV_BF16>::to_f32(h[j]));
  }
  hi = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
  lo = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}

template <int DT, int MT, int NT, bool VEC_ONLY = false>   // VEC_ONLY: the caller guarantees N % 4 == 0, row-major output, no SwiGLU (conv_win_kernel)
__device__ __forceinline__ void gemm_epilogue(const cv_gemm_params& p, f32x4_t (&acc)[MT][NT], int m0, int n0, int wave_m, int wave_n,
                                              int lane, int z, int z0, int z1) {
  constexpr int ES = ElemSize<DT>::value;
  // ------------------------------------------------------------------ epilogue
  // acc[i][j][r]: m = m0 + (wave_m*MT + i)*16 + (lane&15);  n = n0 + (wave_n*NT + j)*16 + 4*(lane>>4) + r
  // All bias / residual / act-param loads are issued UNCONDITIONALLY (clamped addresses, results masked) and hoisted in
  // front of the arithmetic: a load inside a per-element branch makes hipcc wait vmcnt(0) per element
  // (cdna_hip_programming.md §5 "Three .s-level traps" (c)) — that serialisation dominated the first version.
  const int lm = lane & 15, lg = lane >> 4;
  const int64_t res_off = z0 * p.res_bs0 + z1 * p.res_bs1;
  float* o32 = p.out_f32 ? p.out_f32 + (z0 * p.o32_bs0 + z1 * p.o32_bs1) : nullptr;
  char* oact = p.out_act ? (char*)p.out_act + (z0 * p.oa_bs0 + z1 * p.oa_bs1) * ES : nullptr;
  const bool vec = VEC_ONLY || (((p.N & 3) == 0) && (p.act != CV_ACT_SWIGLU) && (p.out_mode == CV_OUT_ROWMAJOR));

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
        const bool live = rok[i] && nbj[j] < p.N;   // uniform over the lane pairs (lg, lg ^ 1) of the pre-split exchange below: same row, same 8-group
        const float v0 = (acc[i][j][0] + b4[j].x + r4[j].x) * p.out_scale;
        const float v1 = (acc[i][j][1] + b4[j].y + r4[j].y) * p.out_scale;
        const float v2 = (acc[i][j][2] + b4[j].z + r4[j].z) * p.out_scale;
        const float v3 = (acc[i][j][3] + b4[j].w + r4[j].w) * p.out_scale;
        if (o32 && live) *(float4*)(o32 + (int64_t)orow[i] * p.ldo32 + nbj[j]) = make_float4(v0, v1, v2, v3);
        if (oact) {
          float a0, a1, a2, a3;
          if (p.act == CV_ACT_NONE) {                // the two common cases first: apply_act's 7-way switch per element measured 35 - 60 us per conv
            a0 = v0; a1 = v1; a2 = v2; a3 = v3;      // launch at 8 x 60 000 x 64 (tools/hift_conv_bench.py) in a 64-element unrolled epilogue
          } else if (p.act == CV_ACT_SNAKE) {
            a0 = act_snake(v0, ap4[j].x); a1 = act_snake(v1, ap4[j].y); a2 = act_snake(v2, ap4[j].z); a3 = act_snake(v3, ap4[j].w);
          } else {
            a0 = apply_act(p.act, v0, ap4[j].x, p.act_slope); a1 = apply_act(p.act, v1, ap4[j].y, p.act_slope);
            a2 = apply_act(p.act, v2, ap4[j].z, p.act_slope); a3 = apply_act(p.act, v3, ap4[j].w, p.act_slope);
          }
          if (DT == CV_F32 && (p.x3_flags & 4)) {
            // pre-split output: the lanes lg and lg ^ 1 of a row hold the two halves of one 8-value group.  One v_permlane16_swap per register
            // hands the even lane both hi halves and the odd lane both lo halves: each lane writes ONE 16-byte chunk ([8 x hi] or [8 x lo]) instead
            // of two 8-byte pieces of both (cdna guide T21: an epilogue of narrow stores is store-issue bound).  Executed by every lane.
            uint2 hi, lo;
            split_pack4(a0, a1, a2, a3, hi, lo);
            const auto s0 = __builtin_amdgcn_permlane16_swap(hi.x, lo.x, false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(hi.y, lo.y, false, false);
            if (live) {
              const int64_t idx = (int64_t)orow[i] * p.ldoa + nbj[j];
              *(uint4*)(oact + (idx & ~(int64_t)7) * 4 + ((idx >> 2) & 1) * 16) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
          } else if (live) {
            store_act4<DT>(oact, (int64_t)orow[i] * p.ldoa + nbj[j], a0, a1, a2, a3);
          }
        }
      }
    }
    return;
  }
  if constexpr (VEC_ONLY) return;

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

// WM x WN waves per workgroup (NT_ = 64*WM*WN threads); each wave owns (BM/WM) x (BN/WN) of the tile.
// X3 (fp32 tensors only): every product is computed as three bf16 MFMAs on hi/lo splits, a = a_hi + a_lo with
// a_hi = bf16(a), a_lo = bf16(a - a_hi):  a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (relative error ~2^-16, fp32 accumulate).
// The split happens when a tile is written to LDS: the 128-byte row that held 32 floats holds 32 hi (chunks 0-3) + 32 lo
// (chunks 4-7) bf16 values, i.e. exactly the two fragment reads of the 16-bit path, and 3 MFMA 16x16x32 replace the
// 8 exact-f32 16x16x4 per K tile (the f32 MFMA peak is 1/16 of bf16: the HiFT / BigVGAN convs were MFMA-bound on it).
// PS (with X3): both operands arrive PRE-SPLIT (the 8-value group format above: activations written so by the producing launch's
// epilogue, weights converted once on the host) — the LDS store is then one 16-byte copy per chunk, conflict-free like the 16-bit path
// (the split form's 8-byte hi / lo stores showed 33 % LDS bank-conflict cycles, profiles/r01_k).  Without it the split costs ~150 VALU
// instructions per thread per K tile against 24 MFMAs per wave (the HiFT convs re-split every activation once per tap and tile column).
template <int DT, int BM, int BN, int WM = 2, int WN = 2, bool X3 = false, bool PS = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const cv_gemm_params p) {
  static_assert(!X3 || DT == CV_F32, "the bf16x3 split is a mode of the fp32 path");
  static_assert(!PS || X3, "pre-split operands belong to the bf16x3 path");
  constexpr int NTHR = 64 * WM * WN;
  constexpr int ES = ElemSize<DT>::value;
  constexpr int CH = 16 / ES;    // elements per 16-byte chunk
  constexpr int BK = 128 / ES;   // elements per K tile
  constexpr int MT = BM / (16 * WM), NT = BN / (16 * WN);
  constexpr int A_CH = BM * 8 / NTHR, B_CH = BN * 8 / NTHR;
  constexpr int STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wave_m = wid / WN, wave_n = wid % WN;
  // byte offset of this lane's fragment chunk inside a 16-row (2 KiB) block, for k half 0 / 1
  const int frag_off[2] = {((lane & 15) << 7) + ((((lane >> 4)) ^ (((lane & 15) >> 1) & 7)) << 4),
                           ((lane & 15) << 7) + (((4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) << 4)};

  // XCD-aware tile order (speed only, never correctness): blocks are dealt round-robin over the 8 XCDs, each with a
  // private 4 MiB L2.  Remap (bijective for any grid) so that every XCD walks a CONTIGUOUS range of tile ids, and order
  // ids so that consecutive tiles share the larger operand's tile: the small operand stays L2-resident, the large one
  // is fetched from HBM / Infinity Cache once instead of once per tile column.
  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    if (p.N <= p.M) { tile_n = id % ntiles; tile_m = id / ntiles; }   // n fastest: the A tile is reused across W tiles
    else { tile_m = id % mtiles; tile_n = id / mtiles; }              // m fastest: the W tile is reused across A tiles
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const int z1 = z / p.batch_inner, z0 = z - z1 * p.batch_inner;

  const char* Ab = (const char*)p.A + (z0 * p.a_bs0 + z1 * p.a_bs1) * ES;
  const char* Wb = (const char*)p.W + (z0 * p.w_bs0 + z1 * p.w_bs1) * ES;

  const int nk = (p.K + BK - 1) / BK;
  const bool conv = p.cin != p.K;

  // ---- per-thread loader state, hoisted out of the K loop (the first version recomputed rows, clamps, 64-bit
  // addresses and LDS offsets per tile: ~200 VALU instructions per 8 MFMAs made the 64x64 tile issue-bound)
  uint4 ra[A_CH], rb[B_CH];
  uint32_t amask = 0, bmask = 0;        // validity bits of the prefetched chunks; applied when written to LDS
  const int k_last = p.K - CH;          // last valid chunk start
  int a_m_ok[A_CH], a_rowbase[A_CH], a_tap[A_CH], a_ci[A_CH], a_lds[A_CH], a_lds2[A_CH];
  const char* a_ptr[A_CH];              // non-conv: fixed row, advances by 128 B per tile
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = i * NTHR + tid;
    const int row = c >> 3, kc = c & 7;  // 8 consecutive lanes = one 128-byte line of a row
    const int m = m0 + row;
    a_lds[i] = PS ? lds_chunk_off(row, (kc >> 1) + 4 * (kc & 1))                              // PS: whole hi / lo chunk
               : (X3 ? lds_chunk_off(row, kc >> 1) + ((kc & 1) << 3) : lds_chunk_off(row, kc));   // X3: hi half-chunk
    a_lds2[i] = lds_chunk_off(row, 4 + (kc >> 1)) + ((kc & 1) << 3);                            //     lo half-chunk
    a_m_ok[i] = m < p.M;
    a_rowbase[i] = m * p.a_row_stride + p.tap_base;
    const int k = kc * CH;
    a_tap[i] = conv ? k / p.cin : 0;
    a_ci[i] = conv ? k - a_tap[i] * p.cin : k;
    const int arow = a_rowbase[i];
    const int arc = min(max(arow, 0), p.a_rows - 1);
    a_ptr[i] = Ab + ((int64_t)arc * p.lda + k) * ES;
    if (!conv) a_m_ok[i] = a_m_ok[i] && (arow >= 0) && (arow < p.a_rows);
  }
  int b_ok[B_CH], b_lds[B_CH], b_lds2[B_CH], b_k[B_CH];
  const char* b_ptr[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = i * NTHR + tid;
    const int row = c >> 3, kc = c & 7;
    const int n = n0 + row;
    b_lds[i] = BM * 128 + (PS ? lds_chunk_off(row, (kc >> 1) + 4 * (kc & 1))
                              : (X3 ? lds_chunk_off(row, kc >> 1) + ((kc & 1) << 3) : lds_chunk_off(row, kc)));
    b_lds2[i] = BM * 128 + lds_chunk_off(row, 4 + (kc >> 1)) + ((kc & 1) << 3);
    b_ok[i] = n < p.N;
    b_k[i] = kc * CH;
    b_ptr[i] = Wb + ((int64_t)min(n, p.N - 1) * p.ldw + kc * CH) * ES;
  }

  // Loads are UNCONDITIONAL (clamped addresses; validity applied as a mask when the tile is written to LDS): a load
  // under a per-element branch makes hipcc wait vmcnt(0) per element and serialises the K loop on memory latency.
  auto load_tile = [&](int kt) {
    const int kbase = kt * BK;
    amask = 0;
    bmask = 0;
    if (!conv) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int k = kbase + a_ci[i];
        const int back = max(k - k_last, 0);  // elements to step back so a (masked) tail chunk stays inside the row
        ra[i] = *(const uint4*)(a_ptr[i] + ((int64_t)kbase - back) * ES);
        amask |= ((a_m_ok[i] && k < p.K) ? 1u : 0u) << i;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int k = kbase + ((i * NTHR + tid) & 7) * CH;
        const int arow = a_rowbase[i] + a_tap[i] * p.tap_step;
        const bool ok = a_m_ok[i] && (k < p.K) && (arow >= 0) && (arow < p.a_rows);
        const int arc = min(max(arow, 0), p.a_rows - 1);
        const int tapc = k < p.K ? a_ci[i] : 0;
        ra[i] = *(const uint4*)(Ab + ((int64_t)arc * p.lda + tapc) * ES);
        amask |= (ok ? 1u : 0u) << i;
        // advance (tap, ci) by one K tile
        a_ci[i] += BK;
        while (a_ci[i] >= p.cin) { a_ci[i] -= p.cin; a_tap[i] += 1; }
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int k = kbase + b_k[i];
      const int back = max(k - k_last, 0);
      rb[i] = *(const uint4*)(b_ptr[i] + ((int64_t)kbase - back) * ES);
      bmask |= ((b_ok[i] && k < p.K) ? 1u : 0u) << i;
    }
  };
  // X3: 4 floats -> 4 bf16 hi (8 bytes) + 4 bf16 lo (8 bytes)
  auto split4 = [](const uint4& r, uint32_t mk, uint2& hi, uint2& lo) {
    const float f[4] = {bitcast<float>(r.x & mk), bitcast<float>(r.y & mk), bitcast<float>(r.z & mk), bitcast<float>(r.w & mk)};
    uint16_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = Elem16<CV_BF16>::from_f32(f[j]);
      l[j] = Elem16<CV_BF16>::from_f32(f[j] - Elem16<CV_BF16>::to_f32(h[j]));
    }
    hi = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    lo = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const uint32_t mk = ((amask >> i) & 1u) ? 0xFFFFFFFFu : 0u;
      if constexpr (PS) {   // the 16-byte chunk is 8 hi (even chunk) or 8 lo (odd chunk) values: one conflict-free 16-byte store
        *(uint4*)(smem + a_lds[i]) = make_uint4(ra[i].x & mk, ra[i].y & mk, ra[i].z & mk, ra[i].w & mk);
      } else if constexpr (X3) {
        uint2 hi, lo;
        split4(ra[i], mk, hi, lo);
        *(uint2*)(smem + a_lds[i]) = hi;
        *(uint2*)(smem + a_lds2[i]) = lo;
      } else {
        *(uint4*)(smem + a_lds[i]) = make_uint4(ra[i].x & mk, ra[i].y & mk, ra[i].z & mk, ra[i].w & mk);
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const uint32_t mk = ((bmask >> i) & 1u) ? 0xFFFFFFFFu : 0u;
      if constexpr (PS) {
        *(uint4*)(smem + b_lds[i]) = make_uint4(rb[i].x & mk, rb[i].y & mk, rb[i].z & mk, rb[i].w & mk);
      } else if constexpr (X3) {
        uint2 hi, lo;
        split4(rb[i], mk, hi, lo);
        *(uint2*)(smem + b_lds[i]) = hi;
        *(uint2*)(smem + b_lds2[i]) = lo;
      } else {
        *(uint4*)(smem + b_lds[i]) = make_uint4(rb[i].x & mk, rb[i].y & mk, rb[i].z & mk, rb[i].w & mk);
      }
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // single LDS stage (occupancy hides latency) + register prefetch of the next K tile
  load_tile(0);
  store_tile();
  __syncthreads();
  const char* sa = smem;
  const char* sb = smem + BM * 128;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
    if constexpr (X3) {
      uint4 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        ah[i] = *(const uint4*)(sa + ((wave_m * MT + i) << 11) + frag_off[0]);
        al[i] = *(const uint4*)(sa + ((wave_m * MT + i) << 11) + frag_off[1]);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bh[j] = *(const uint4*)(sb + ((wave_n * NT + j) << 11) + frag_off[0]);
        bl[j] = *(const uint4*)(sb + ((wave_n * NT + j) << 11) + frag_off[1]);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          // small cross terms first, the dominant hi*hi last
          acc[i][j] = mfma_block<CV_BF16>(bl[j], ah[i], acc[i][j]);
          acc[i][j] = mfma_block<CV_BF16>(bh[j], al[i], acc[i][j]);
          acc[i][j] = mfma_block<CV_BF16>(bh[j], ah[i], acc[i][j]);
        }
    } else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *(const uint4*)(sa + ((wave_m * MT + i) << 11) + frag_off[ks]);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *(const uint4*)(sb + ((wave_n * NT + j) << 11) + frag_off[ks]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_block<DT>(fb[j], fa[i], acc[i][j]);
    }
    if (kt + 1 < nk) {
      __syncthreads();
      store_tile();
      __syncthreads();
    }
  }

  gemm_epilogue<DT, MT, NT>(p, acc, m0, n0, wave_m, wave_n, lane, z, z0, z1);
}

template <int DT, int BM, int BN, int WM = 2, int WN = 2, bool X3 = false, bool PS = false>
int launch(const cv_gemm_params& p, hipStream_t st) {
  const int mt = (p.M + BM - 1) / BM, nt = (p.N + BN - 1) / BN;
  dim3 grid(mt * nt, 1, p.batch);
  const size_t lds = (BM + BN) * 128;
  hipLaunchKernelGGL((gemm_kernel<DT, BM, BN, WM, WN, X3, PS>), grid, dim3(64 * WM * WN), lds, st, p);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

// ================================================================================================================
// gemm_ring_kernel: 128x128 tile, LDS-DMA (global_load_lds, 16 B per lane) into a STAGES-deep ring of 32 KiB stages,
// counted vmcnt + raw s_barrier so STAGES-1 tiles stay in flight across the MFMA block.  Motivation (measured,
// DESIGN.md §6): a CU pulls only ~10 B/clk through global loads, so the 64x64 / register-staged kernel is bound by
// operand traffic (262 MB for a 16000x1024x256 GEMM); a 128x128 tile halves the bytes per flop and the ring keeps the
// loads in flight at one workgroup per CU.  Same LDS fragment image and epilogue as gemm_kernel.
// Out-of-range chunks (M/N/K tails, conv padding rows) are fetched from a 16-byte zero word instead of being masked.
__device__ uint4 g_zero16;

template <int DT, int STAGES>
__global__ __launch_bounds__(256) void gemm_ring_kernel(const cv_gemm_params p) {
  constexpr int BM = 128, BN = 128, MT = 4, NT = 4;
  constexpr int ES = ElemSize<DT>::value;
  constexpr int CH = 16 / ES;
  constexpr int BK = 128 / ES;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NB = 8;  // 1 KiB blocks (16 rows x 4 chunks) per wave per tile: waves 0,1 load A, waves 2,3 load W
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid >> 1, wave_n = wid & 1;
  // byte offset of this lane's fragment chunk inside a 16-row (2 KiB) block, for k half 0 / 1
  const int frag_off[2] = {((lane & 15) << 7) + ((((lane >> 4)) ^ (((lane & 15) >> 1) & 7)) << 4),
                           ((lane & 15) << 7) + (((4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) << 4)};
  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    if (p.N <= p.M) { tile_n = id % ntiles; tile_m = id / ntiles; }
    else { tile_m = id % mtiles; tile_n = id / mtiles; }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const int z1 = z / p.batch_inner, z0 = z - z1 * p.batch_inner;
  const char* Ab = (const char*)p.A + (z0 * p.a_bs0 + z1 * p.a_bs1) * ES;
  const char* Wb = (const char*)p.W + (z0 * p.w_bs0 + z1 * p.w_bs1) * ES;
  const int nk = (p.K + BK - 1) / BK;
  const bool conv = p.cin != p.K;
  const bool load_a = wid < 2;
  const char* zero = (const char*)&g_zero16;

  // LDS-DMA writes 1 KiB per wave instruction at base + lane*16 = 8 consecutive rows x 8 chunk slots of the row-major
  // image; the XOR swizzle therefore goes on the SOURCE: lane (row lr8 = lane>>3, slot = lane&7) fetches chunk
  // slot ^ ((row>>1)&7) of its row (cdna guide rule 21: linear dest + swizzled source + swizzled read).
  // block j (of 8) of this wave covers rows (wid&1)*64 + j*8 .. +7 of the A (waves 0,1) or W (waves 2,3) tile.
  const int lr8 = lane >> 3, slot = lane & 7;
  int rowbase[NB];
  bool rok[NB];
  const char* rptr[NB];  // non-conv A / W: pointer to (row, chunk) at k = 0
  int kch[NB];           // source chunk index (0..7) of this lane in block j
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int row = (wid & 1) * 64 + j * 8 + lr8;
    kch[j] = slot ^ ((row >> 1) & 7);
    if (load_a) {
      const int m = m0 + row;
      rowbase[j] = m * p.a_row_stride + p.tap_base;
      rok[j] = m < p.M;
      const int arow = rowbase[j];
      if (!conv) rok[j] = rok[j] && arow >= 0 && arow < p.a_rows;
      rptr[j] = Ab + ((int64_t)min(max(arow, 0), p.a_rows - 1) * p.lda + kch[j] * CH) * ES;
    } else {
      const int n = n0 + row;
      rowbase[j] = 0;
      rok[j] = n < p.N;
      rptr[j] = Wb + ((int64_t)min(n, p.N - 1) * p.ldw + kch[j] * CH) * ES;
    }
  }

  auto issue = [&](int kt, int stage) {
    char* sbase = smem + stage * STAGE + (load_a ? 0 : BM * 128) + (wid & 1) * (NB * 1024);
    const int kbase = kt * BK;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int k = kbase + kch[j] * CH;
      const char* src;
      if (load_a && conv) {
        const int tap = k / p.cin, ci = k - tap * p.cin;
        const int arow = rowbase[j] + tap * p.tap_step;
        const bool ok = rok[j] && k < p.K && arow >= 0 && arow < p.a_rows;
        src = ok ? Ab + ((int64_t)arow * p.lda + ci) * ES : zero;
      } else {
        const bool ok = rok[j] && k < p.K;
        src = ok ? rptr[j] + (int64_t)kbase * ES : zero;
      }
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sbase + j * 1024), 16, 0, 0);
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // prologue: STAGES-1 tiles in flight
#pragma unroll
  for (int s2 = 0; s2 < STAGES - 1; ++s2)
    if (s2 < nk) issue(s2, s2);

  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt % STAGES;
    if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    // tiles still allowed in flight after this wait: min(STAGES-1, nk-1-kt), 8 loads each
    const int ahead = min(STAGES - 1, nk - 1 - kt);
    if (ahead >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* sa = smem + stage * STAGE;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 fa[MT], fb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = *(const uint4*)(sa + ((wave_m * MT + i) << 11) + frag_off[ks]);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[j] = *(const uint4*)(sb + ((wave_n * NT + j) << 11) + frag_off[ks]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_block<DT>(fb[j], fa[i], acc[i][j]);
    }
    // WAR: this stage is refilled by the issue() of the next iteration
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  gemm_epilogue<DT, MT, NT>(p, acc, m0, n0, wave_m, wave_n, lane, z, z0, z1);
}

template <int DT>
int launch_ring(const cv_gemm_params& p, hipStream_t st) {
  const int mt = (p.M + 127) / 128, nt = (p.N + 127) / 128;
  dim3 grid(mt * nt, 1, p.batch);
  constexpr int STAGES = 3;
  const size_t lds = (size_t)STAGES * 256 * 128;
  static PerDeviceOnce once;
  once.run([&] { hipFuncSetAttribute((const void*)gemm_ring_kernel<DT, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  hipLaunchKernelGGL((gemm_ring_kernel<DT, STAGES>), grid, dim3(256), lds, st, p);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

// ================================================================================================================
// conv_win_kernel: stride-1 Conv1d on the bf16x3 path with both operands pre-split (the HiFT / BigVGAN ResBlock convs: 91 % of the
// vocoder's time on gemm_kernel<64,64,X3,PS>).  The implicit-im2col GEMM above re-fetches every activation row once per tap and
// once per tile column through global loads -> registers -> LDS and synchronises twice per 32-deep K tile (12 MFMAs per wave): it
// is bound by operand staging (~6-15 B/clk/CU), not MFMA (0.10 of the bf16 peak issued, profiles/r03_hift_*).  Here a workgroup
// keeps the activation WINDOW of its rows — BM + (taps-1)*dilation rows x 64 channels, pre-split: 256 B per row — in LDS once per
// 64-channel chunk (LDS-DMA, XOR-swizzled on the source address) and every tap reads its A fragments from that image at a shifted
// row; the weights never touch LDS: each wave streams the fragments of its own 64 output columns L2 -> VGPR through a buffer
// descriptor (the (tap, chunk, k-step) offset is an SGPR), double-buffered one 32-deep k-step ahead.  Per k-step a wave issues
// 8 ds_read_b128 + 8 buffer loads for 48 MFMAs (wave tile 64 x 64) and a workgroup synchronises twice per CHUNK, not per K tile.
// Summation order per output: chunk-major, then tap, then k-step, cross terms before hi*hi — the same for every tile shape, so a
// conv's result does not depend on the batch size or on the tile the dispatcher picks.
template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_win_kernel(const cv_gemm_params p, const int win_rows) {
  constexpr int NW = WM * WN, MT = 4, NT = 4, BM = 64 * WM, BN = 64 * WN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wid / WN, wave_n = wid % WN;
  const int lq = lane & 15, lg = lane >> 4;

  const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.N + BN - 1) / BN;
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    tile_n = id % ntiles; tile_m = id / ntiles;   // column tiles of one window back to back on one XCD
    (void)mtiles;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.z;
  const int z1 = z / p.batch_inner, z0 = z - z1 * p.batch_inner;
  const char* Ab = (const char*)p.A + (z0 * p.a_bs0 + z1 * p.a_bs1) * 4;
  const char* Wb = (const char*)p.W + (z0 * p.w_bs0 + z1 * p.w_bs1) * 4;
  const int ktaps = p.K / p.cin, nchunks = p.cin >> 6;
  const char* zero = (const char*)&g_zero16;

  // ---- window loader: LDS-DMA, one wave instruction = 1 KiB = 4 window rows x 16 chunk slots; slot s of row r holds global chunk s ^ (r & 15)
  const int wr_in = lane >> 4, wslot = lane & 15;
  const int ngroups = (win_rows + 3) >> 2;
  auto load_window = [&](int c) {
    for (int g = wid; g < ngroups; g += NW) {
      const int w = g * 4 + wr_in;
      const int arow = m0 + p.tap_base + w;
      const bool ok = (w < win_rows) && (arow >= 0) && (arow < p.a_rows);
      const char* src = ok ? Ab + ((int64_t)arow * p.lda + c * 64) * 4 + ((wslot ^ (w & 15)) << 4) : zero;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(smem + g * 1024), 16, 0, 0);
    }
  };

  // ---- weight stream: lane = output column (lq) x 8-value group (lg) of the 32-deep k-step; [8 hi | 8 lo] = 32 contiguous bytes
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, (int)min((int64_t)p.N * p.ldw * 4, (int64_t)0x7FFFFFFF), 0x00020000);
  int woff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = min(n0 + (wave_n * NT + j) * 16 + lq, p.N - 1);
    woff[j] = (n * p.ldw + lg * 8) * 4;
  }
  u32x4_t wh[2][NT], wl[2][NT];
  auto load_w = [&](int set, int c, int t, int ks) {
    const int so = (t * p.cin + c * 64 + ks * 32) * 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      wh[set][j] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[j], so, 0);
      wl[set][j] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, woff[j] + 16, so, 0);
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int rbase = wave_m * (MT * 16) + lq;   // window row of this lane's A fragment row at tap 0, row tile 0
  auto step = [&](int set, int t, int ks) {
    uint4 ah[MT], al[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = rbase + i * 16 + t * p.tap_step;
      const int off = (row << 8) + ((((lg << 1) ^ (row & 15)) << 4) ^ (ks << 7));   // slot = chunk ^ (row & 15), chunk = ks*8 + lg*2 + {hi 0, lo 1}
      ah[i] = *(const uint4*)(smem + off);
      al[i] = *(const uint4*)(smem + (off ^ 16));
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint4 bh = bitcast<uint4>(wh[set][j]), bl = bitcast<uint4>(wl[set][j]);
        acc[i][j] = mfma_block<CV_BF16>(bl, ah[i], acc[i][j]);
        acc[i][j] = mfma_block<CV_BF16>(bh, al[i], acc[i][j]);
        acc[i][j] = mfma_block<CV_BF16>(bh, ah[i], acc[i][j]);
      }
  };

  load_w(0, 0, 0, 0);
  for (int c = 0; c < nchunks; ++c) {
    load_window(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ktaps; ++t) {
      // sched_barrier(0): left alone, hipcc sinks each refill to just before its first use and the prefetch distance collapses to zero
      load_w(1, c, t, 1);
      __builtin_amdgcn_sched_barrier(0);
      step(0, t, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < ktaps) load_w(0, c, t + 1, 0);
      else if (c + 1 < nchunks) load_w(0, c + 1, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      step(1, t, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < nchunks) __syncthreads();   // every wave is done with this chunk's image before the next one lands
  }
  gemm_epilogue<CV_F32, MT, NT, true>(p, acc, m0, n0, wave_m, wave_n, lane, z, z0, z1);
}

// CV_CONV_WIN=1 opts the qualifying convs in (default: gemm_kernel); CV_CONV_WIN_SHAPE=WMxWN forces one workgroup shape.  Both are read per
// call.  Measured (tools/hift_conv_bench.py, profiles/r03_hift_conv_bench.log, DESIGN.md status item 4): alone, on the c1-type launch (snake +
// pre-split output), the window kernel is 5 - 20 % faster on every batch-8 shape (1 282 vs 1 501 us over the nine shapes) and within +-10 % on
// single-utterance grids; inside a decode, where every second conv also reads the residual and writes the fp32 stream, the advantage shrinks to 3 %
// at batch 8 (12.3 vs 12.65 ms of conv time) and turns into a 4 % loss at batch 1 (3.67 vs 3.51 ms per decode) — its window load, taps and
// epilogue are serial inside a workgroup at 2 - 3 workgroups per CU.  Not the default.
constexpr int CONV_WIN_LDS_MAX = 80 * 1024;   // two workgroups per CU

template <int WM, int WN>
int launch_conv_win(const cv_gemm_params& p, hipStream_t st, int halo) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  const int win_rows = BM + halo;
  const size_t lds = (size_t)((win_rows + 3) & ~3) * 256;
  static PerDeviceOnce once;
  once.run([&] { hipFuncSetAttribute((const void*)conv_win_kernel<WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, CONV_WIN_LDS_MAX); });
  dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), 1, p.batch);
  hipLaunchKernelGGL((conv_win_kernel<WM, WN>), grid, dim3(64 * WM * WN), lds, st, p, win_rows);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

// returns CONV_WIN_NO when the launch does not qualify (the caller falls through to gemm_kernel)
constexpr int CONV_WIN_NO = 1;
int dispatch_conv_win(const cv_gemm_params& p, hipStream_t st) {
  int force_wm = 0, force_wn = 0;
  if (const char* e = getenv("CV_CONV_WIN"); !e || atoi(e) != 1) return CONV_WIN_NO;
  if (const char* s = getenv("CV_CONV_WIN_SHAPE"); s && s[0] >= '1' && s[0] <= '4' && s[1] == 'x' && s[2] >= '1' && s[2] <= '2') { force_wm = s[0] - '0'; force_wn = s[2] - '0'; }
  if (p.cin == p.K || (p.x3_flags & 3) != 3 || p.a_row_stride != 1 || p.tap_step <= 0) return CONV_WIN_NO;
  if ((p.cin & 63) || (p.N & 63) || p.act == CV_ACT_SWIGLU || p.out_mode != CV_OUT_ROWMAJOR) return CONV_WIN_NO;
  if ((int64_t)p.N * p.ldw * 4 > 0x7FFFFFFF) return CONV_WIN_NO;
  const int halo = (p.K / p.cin - 1) * p.tap_step;
  int wm, wn = (p.N >= 128) ? 2 : 1;
  if (force_wm) { wm = force_wm == 3 ? 2 : force_wm; wn = min(force_wn, wn); }
  else {
    // waves per workgroup by the number of 64 x 64 wave tiles in the launch (tools/hift_conv_bench.py: one-wave workgroups win on single-utterance
    // grids of 250 - 940 tiles, two waves at 2 000, four from 5 000); four waves = 128 x 128 where N allows, else 256 x 64
    const long long tiles = (long long)((p.M + 63) / 64) * ((p.N + 63) / 64) * p.batch;
    const int nw = tiles >= 4096 ? 4 : (tiles >= 1536 ? 2 : 1);
    if (nw < 2) wn = 1;
    wm = nw / wn;
  }
  while (wm > 1 && (size_t)((64 * wm + halo + 3) & ~3) * 256 > (size_t)CONV_WIN_LDS_MAX) wm >>= 1;
  if ((size_t)((64 * wm + halo + 3) & ~3) * 256 > (size_t)CONV_WIN_LDS_MAX) return CONV_WIN_NO;
  if (wm == 4 && wn == 1) return launch_conv_win<4, 1>(p, st, halo);
  if (wm == 2 && wn == 2) return launch_conv_win<2, 2>(p, st, halo);
  if (wm == 2 && wn == 1) return launch_conv_win<2, 1>(p, st, halo);
  if (wm == 1 && wn == 2) return launch_conv_win<1, 2>(p, st, halo);
  if (wm == 1 && wn == 1) return launch_conv_win<1, 1>(p, st, halo);
  return CONV_WIN_NO;
}

// CV_GEMM_TILE=0|1|2|3 (128x128 | 128x64 | 64x64 | 128x128 LDS-DMA ring) overrides the heuristic: tuning aid only
static int g_tile_override = -2;

// fp32 tensors, bf16x3 products: the two register-staged tiles only
int dispatch_x3(const cv_gemm_params& p, hipStream_t st) {
  if (const int rc = dispatch_conv_win(p, st); rc != CONV_WIN_NO) return rc;
  const long long t12864 = (long long)((p.M + 127) / 128) * ((p.N + 63) / 64) * p.batch;
  const bool big = p.act == CV_ACT_SWIGLU || (p.K > 512 && t12864 >= 768);
  if ((p.x3_flags & 3) == 3) {   // both operands pre-split
    if (big) return launch<CV_F32, 128, 64, 2, 2, true, true>(p, st);
    return launch<CV_F32, 64, 64, 2, 2, true, true>(p, st);
  }
  if (p.x3_flags & 3) return CV_ERR_UNSUPPORTED;   // one operand pre-split, the other not
  if (big) return launch<CV_F32, 128, 64, 2, 2, true>(p, st);
  return launch<CV_F32, 64, 64, 2, 2, true>(p, st);
}

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
    // very tall GEMMs with wide N and K (the LLM prefill of a 4-batch decode job: 9 024 rows): the 8-wave 128 x 128 tile halves the
    // operand bytes per flop (tools/prefill_probe.py: 18.8 -> 16.2 ms at 32 sequences; slower at 8 sequences = 2 256 rows)
    if (p.batch == 1 && p.M >= 8192 && p.N >= 512 && p.K >= 512) tile = 4;
  }
  if (swiglu && (tile == 0 || tile == 3)) tile = 1;
  if (tile == 3) return launch_ring<DT>(p, st);
  if (tile == 4) return launch<DT, 128, 128, 2, 4>(p, st);  // 8 waves, wave tile 64x32
  if (tile == 0) return launch<DT, 128, 128>(p, st);
  if (tile == 1) return launch<DT, 128, 64>(p, st);
  return launch<DT, 64, 64>(p, st);
}

}  // namespace

extern "C" int cv_gemm(const cv_gemm_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  cv_gemm_params p = *pp;
  const bool x3 = p.dtype == CV_F32X3;   // fp32 tensors, bf16x3 products: everything below sees an fp32 GEMM
  if (x3) p.dtype = CV_F32;
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
  if (p.x3_flags && (!x3 || (p.x3_flags & ~7))) return CV_ERR_ARG;          // pre-split storage belongs to CV_F32X3 launches
  if ((p.x3_flags & 4) && (!p.out_act || (p.N & 7) || (p.ldoa & 7) || p.act == CV_ACT_SWIGLU || p.out_mode != CV_OUT_ROWMAJOR)) return CV_ERR_ARG;
  if ((p.x3_flags & 3) && ((p.K & 7) || (p.cin & 7) || (p.lda & 7) || (p.ldw & 7))) return CV_ERR_ARG;   // 8-value groups
  hipStream_t st = (hipStream_t)stream;
  if (x3) return dispatch_x3(p, st);
  switch (p.dtype) {
    case CV_F32: return dispatch<CV_F32>(p, st);
    case CV_BF16: return dispatch<CV_BF16>(p, st);
    case CV_F16: return dispatch<CV_F16>(p, st);
    default: return CV_ERR_ARG;
  }
}
