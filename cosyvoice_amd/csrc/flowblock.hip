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
//   * a register ring of NS slots (8 fragments = 32 MFMA-steps of work each) keeps NS - 1 fragment groups in flight behind
//     the MFMAs.  First version: 2 slots at two waves per SIMD -> one group (512 cycles) of cover against a loaded L2
//     latency of 1-2 us.  Deeper rings at ONE wave per SIMD (6 slots in the 256 architectural VGPRs, accumulators in AGPRs;
//     hipcc does not place load destinations in AGPRs, an 8-slot ring was spilled inside the loop) measured no faster: with a
//     single wave per SIMD every latency of the chain is exposed (SQ counters: 25 % MFMA-busy, 35 % issue stalls, 40 % issuing;
//     ablations: output stores 10 of the head's 28 us, GELU 7 of the tail's 40 us).  So a workgroup is now EIGHT waves on the
//     same 64 rows (two per SIMD, the output columns split eight ways, 4 slots = 24 KiB in flight per wave): one wave's MFMAs
//     run under the other's GELU / stores / waits, at the same L2 -> CU traffic;
//   * the 1024-wide GELU intermediate never leaves the CU (128-column chunks through a double-buffered LDS tile),
//     the residual row stays in the accumulators from the out-projection to the final store, and LayerNorm(norm3)
//     is computed from those accumulators (cross-wave row statistics through 2 KiB of LDS).
// hipcc notes (ROCm 7.2): every ring refill is pinned with sched_barrier(0) — left alone, the scheduler sinks a refill to
// just before its use (the slot's registers are dead in between) and the prefetch distance collapses to zero; a batch of
// row loads is pinned with an empty asm that redefines all 16 registers — otherwise each load is sunk to its row's arithmetic
// (16 dependent L2 round trips); small global loads inside the chunk loop (bias) are likewise sunk behind the weight
// prefetch and drain it with vmcnt(0), so the hidden-layer bias is staged in LDS; the head's stores are raw buffer stores
// (rows beyond T dropped by the bounds check) because stores under an exec-masked branch make the in-order vmcnt
// bookkeeping of the prefetched loads conservative.
#include "cv_device.h"
#include <climits>
#include <cstdlib>

namespace {

constexpr int TB_C = 256, TB_INNER = 512, TB_FF = 1024;
constexpr int BM = 64;        // rows of the LDS operand images (LayerNorm / staging always cover 64 rows from the tile's first row)
constexpr int NW = 8;         // waves per workgroup (two per SIMD)
constexpr int NTHR = 64 * NW;
constexpr int NS = 4;         // ring slots per wave, 8 fragments (8 KiB per wave) each
constexpr int HC = 32 * NW;   // hidden columns per FFN chunk: two 16-column tiles per wave
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// 16-byte chunk slot of (row, chunk) in an operand image whose rows hold >= 16 chunks: XOR the low 4 chunk bits with row & 15.
// A ds_read_b128 fragment (lane l: row l & 15, chunk 4 ks + (l >> 4)) then hits 16 distinct slots per 16-lane group.
__device__ __forceinline__ int swz16(int row, int chunk) { return (chunk & ~15) | ((chunk ^ row) & 15); }

// exact-erf GELU (diffusers GELU(approximate="none")) = relu(x) - 0.5 |x| erfc(|x| / sqrt 2), erfc by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the 16-bit rounding of the result; no 1 - erf cancellation on the negative side).
// 13 VALU + 2 transcendental instead of ocml erff's ~40 with branches: the FFN evaluates 1024 of these per row per block.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));   // exp(-x^2 / 2)
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float erfc_abs = pl * t * e;
  return fmaf(-0.5f * ax, erfc_abs, fmaxf(x, 0.f));
}

// One weight fragment (1 KiB per wave: lane l reads bytes [16 l, 16 l + 16)) through a buffer descriptor: the fragment's byte
// offset is wave-uniform and travels in an SGPR (soffset), the only address VGPR is 16 * lane.  With plain pointers hipcc
// materialises a 64-bit VGPR address pair per 4 KiB of fragment offsets and hoists all of them to the kernel's top (the fully
// unrolled head kernel carried ~100 address registers and spilled its prefetch ring to scratch).
__device__ __forceinline__ uint4 frag_load(const __amdgpu_buffer_rsrc_t rs, int lane16, int frag) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, frag * 1024, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frag_rsrc(const void* base, int nfrag) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nfrag * 1024, 0x00020000);
}

// "redefine" 8 registers at this point: every load that fills them has to be issued before it (hipcc otherwise sinks each
// load of a batch to its own consumer and pays one dependent L2 round trip per row)
#define PIN8(v) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
#define PIN6(v) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]))
#define PIN4(v) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]))
#define PIN2(v) asm volatile("" : "+v"(v[0]), "+v"(v[1]))
template <int N, typename V>
__device__ __forceinline__ void pin_regs(V (&v)[N]) {
  static_assert(N == 2 || N == 4 || N == 6 || N == 8, "pin_regs");
  if constexpr (N == 8) PIN8(v);
  else if constexpr (N == 6) PIN6(v);
  else if constexpr (N == 4) PIN4(v);
  else PIN2(v);
}

template <int DT>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  return make_uint2(pack2<DT>(a, b), pack2<DT>(c, d));
}

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), every lane gets the total: 4 v_add_f32_dpp, no LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
  v += bitcast<float>(__builtin_amdgcn_update_dpp(0, bitcast<int>(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += bitcast<float>(__builtin_amdgcn_update_dpp(0, bitcast<int>(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += bitcast<float>(__builtin_amdgcn_update_dpp(0, bitcast<int>(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += bitcast<float>(__builtin_amdgcn_update_dpp(0, bitcast<int>(v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

// LayerNorm of this workgroup's rows straight from global memory into the K = 256 operand image (512-byte rows):
// wave w normalises rows RPW w .. RPW w + RPW - 1 (RPW = 8 for 48 / 64-row tiles, 4 for 16 / 32-row tiles: the image always
// holds a multiple of 32 rows), four rows per pass, 16 lanes per row (lane sub = l & 15 holds columns 64 k + 4 sub ..+3, k < 4:
// 256-byte coalesced segments), two-pass in registers, row sums by DPP.
// Rows beyond T re-read row T-1 (finite values; their results are never stored).
template <int DT, int MT>
__device__ __forceinline__ void ln_rows_to_lds(const float* xs, int ldx, int t0, int T, const float* gamma, const float* beta,
                                               float eps, char* img, int wid, int lane) {
  constexpr int NP = (16 * MT + 4 * NW - 1) / (4 * NW), RPW = 4 * NP;   // passes per wave (1 or 2), rows per wave
  const int sub = lane & 15, rr = lane >> 4;
  f32x4_t v[NP * 4];   // [pass][k]
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int t = min(t0 + wid * RPW + ps * 4 + rr, T - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[ps * 4 + k] = *(const f32x4_t*)(xs + (int64_t)t * ldx + 64 * k + 4 * sub);
  }
  float4 g4[4], b4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    g4[k] = *(const float4*)(gamma + 64 * k + 4 * sub);
    b4[k] = *(const float4*)(beta + 64 * k + 4 * sub);
  }
  pin_regs(v);
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (v[ps * 4 + k][0] + v[ps * 4 + k][1]) + (v[ps * 4 + k][2] + v[ps * 4 + k][3]);
    const float mean = row16_sum(s) * (1.0f / TB_C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[ps * 4 + k][e] - mean; v[ps * 4 + k][e] = d; q += d * d; }
    const float rstd = rsqrtf(row16_sum(q) * (1.0f / TB_C) + eps);
    const int row = wid * RPW + ps * 4 + rr;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *(uint2*)(img + row * 512 + (swz16(row, 8 * k + (sub >> 1)) << 4) + ((sub & 1) << 3)) =
          pack4<DT>(v[ps * 4 + k][0] * rstd * g4[k].x + b4[k].x, v[ps * 4 + k][1] * rstd * g4[k].y + b4[k].y,
                    v[ps * 4 + k][2] * rstd * g4[k].z + b4[k].z, v[ps * 4 + k][3] * rstd * g4[k].w + b4[k].w);
  }
}

// ============================================================================================== head: LN -> Q | K | V^T
// Output columns: 1536 = 96 MFMA tiles (0..63 = [Q | K] row-major, 64..95 = V, stored transposed).  A wave-step is 4
// consecutive tiles x all 8 k-steps x the 4 row tiles (128 MFMAs = 4 fragment groups of 2 k-steps); wave w takes tile
// groups w, w + 8, w + 16 (3 steps).  V steps swap the MFMA operands (D = xn . Wv^T has the frame index in the registers),
// so a lane holds 4 consecutive frames of one (head, channel) row of V^T: 8-byte stores, no transposing pass.
// Fragment group G = 4 step + k-quarter lives in ring slot G % 4 = k-quarter and is refilled with group G + 4 once computed.
// MT = MFMA row tiles per workgroup (rows per tile 16 MT = 64 or 48): 48-row tiles exist for grids that would otherwise run as
// two rounds with the second nearly empty (256 tiles of 64 rows on the 192 CUs the pipeline leaves the flow stage).
// fragment group G of the head's weight stream (clamped to the last step: harmless re-reads at the end): tiles (step * NW + w) * 4 + j,
// k-steps 2 kq + u, into one ring slot
__device__ __forceinline__ void head_ld(uint4 (&sl)[8], const __amdgpu_buffer_rsrc_t w_rs, int lane16, int wid, int G) {
  constexpr int NKS = TB_C / 32, NSTEP = 3 * TB_INNER / 16 / (4 * NW);
  const int st = min(G >> 2, NSTEP - 1), kq = G & 3;
  const int tile0 = (st * NW + wid) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int u = 0; u < 2; ++u) sl[j * 2 + u] = frag_load(w_rs, lane16, (tile0 + j) * NKS + kq * 2 + u);
  __builtin_amdgcn_sched_barrier(0);   // ring refills stay where they are written: NS - 1 fragment groups ahead of their use
}

// The head's GEMM + stores over the 16 MT-row operand image at `img` (LayerNorm output, 512-byte rows).  On entry ring slots 0..3 hold
// fragment groups 0..3 (head_ld) and the image is complete (barrier passed).
template <int DT, int MT, int ABL>
__device__ __forceinline__ void head_body(const cv_tblock_params& p, const char* img, uint4 (&s)[NS][8], const __amdgpu_buffer_rsrc_t w_rs,
                                          int lane16, int wid, int lq, int lg, int r, int t0) {
  constexpr int NSTEP = 3 * TB_INNER / 16 / (4 * NW);  // 96 tiles / (8 waves x 4 tiles) = 3
  auto ld = [&](uint4 (&sl)[8], int G) {
    if ((ABL & 2) && G >= NS) return;
    head_ld(sl, w_rs, lane16, wid, G);
  };
  f32x4_t acc[MT][4];
  auto compute = [&](uint4 (&sl)[8], int kq, bool vpart) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ks = kq * 2 + u;
      uint4 a[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *(const uint4*)(img + (16 * i + lq) * 512 + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc[i][j] = vpart ? mfma_block<DT>(a[i], sl[j * 2 + u], acc[i][j]) : mfma_block<DT>(sl[j * 2 + u], a[i], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // stores go through buffer descriptors of this sequence's [Q | K] rows / V^T block: rows beyond T fall outside the
  // descriptor and are dropped by the hardware bounds check (no exec-masked branches around the stores)
  const __amdgpu_buffer_rsrc_t qk_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((uint16_t*)p.qk + (int64_t)r * p.T * p.ldqk), 0, p.T * p.ldqk * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t vt_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((uint16_t*)p.vt + (int64_t)r * TB_INNER * p.vt_ld), 0, TB_INNER * p.vt_ld * 2, 0x00020000);
  const bool t_mod4 = (p.T & 3) != 0;

  // fully unrolled (3 steps): across a loop back-edge hipcc merges the pending-load state of the entry edge with the steady
  // state and falls back to conservative vmcnt values that drain most of the ring once per iteration
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) {
    const int tile0 = (st * NW + wid) * 4;
    constexpr int QK_STEPS = (2 * TB_INNER / 16) / (4 * NW);   // steps 0, 1: [Q | K] tiles for every wave; step 2: V tiles
    const bool vpart = st >= QK_STEPS;                         // compile-time after unrolling
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (!vpart) {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) { compute(s[kq], kq, false); ld(s[kq], st * 4 + kq + NS); }
      // D rows = output column: lane (lq, lg) holds row m = 16 i + lq, columns 4 lg ..+3 of each of the 4 column tiles j.  The four lanes
      // lg = 0..3 of a row exchange their pieces (4 x 4 transpose over (lane lg, tile j): permlane32_swap on bit 1, permlane16_swap on
      // bit 0) so that lane lg owns the 16 columns of tile lg = 32 contiguous bytes: two 16-byte stores instead of four 8-byte ones, 64
      // contiguous bytes per row and instruction instead of 32 (the stores were 10 of the head's 27 us, r02 ablation; cdna guide T21).
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int t = t0 + 16 * i + lq;
        uint32_t w[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint2 u = pack4<DT>(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          w[j][0] = u.x; w[j][1] = u.y;
        }
        if (ABL & 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(w[j][0]), "v"(w[j][1]));
          continue;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const auto a = __builtin_amdgcn_permlane32_swap(w[0][k], w[2][k], false, false);
          const auto b = __builtin_amdgcn_permlane32_swap(w[1][k], w[3][k], false, false);
          const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
          const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
          w[0][k] = c[0]; w[1][k] = c[1]; w[2][k] = d[0]; w[3][k] = d[1];
        }
        const int off = (t * p.ldqk + (tile0 + lg) * 16) * 2;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w[0][0], w[0][1], w[1][0], w[1][1]}, qk_rs, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w[2][0], w[2][1], w[3][0], w[3][1]}, qk_rs, off + 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) { compute(s[kq], kq, true); ld(s[kq], st * 4 + kq + NS); }
      // D rows = frame: lane holds V^T row (head * 64 + d) = column tile * 16 + lq, frames t .. t + 3 (the 4 x 4 exchange of the [Q | K]
      // stores was tried here too, over (lane lg, row tile i): 25.9 vs 25.1 us, no gain — the V^T rows are already 32-byte runs per lane group)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int vrow = (tile0 + j) * 16 + lq - 2 * TB_INNER;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int t = t0 + 16 * i + 4 * lg;
          const uint2 u = pack4<DT>(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          const int off = (vrow * p.vt_ld + t) * 2;
          if (ABL & 1) { asm volatile("" :: "v"(u.x), "v"(u.y)); continue; }
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{u.x, u.y}, vt_rs, (t + 3 < p.T) ? off : 0x7FFFFFF0, 0, 0);
        }
      }
      if (t_mod4) {   // T not a multiple of 4 (the v1 flow's odd lengths): the group that straddles T, element by element
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int vrow = (tile0 + j) * 16 + lq - 2 * TB_INNER;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int t = t0 + 16 * i + 4 * lg;
            const bool straddle = (t < p.T) && (t + 3 >= p.T);
#pragma unroll
            for (int e = 0; e < 3; ++e)
              __builtin_amdgcn_raw_buffer_store_b16((short)Elem16<DT>::from_f32(acc[i][j][e]), vt_rs,
                                                    (straddle && t + e < p.T) ? (vrow * p.vt_ld + t + e) * 2 : 0x7FFFFFF0, 0, 0);
          }
        }
      }
    }
  }
}

template <int DT, int MT, int ABL = 0>   // ABL: timing-only ablations (1: no output stores, 2: no ring refills)
__global__ __launch_bounds__(NTHR, 2) void tblock_head_kernel(const cv_tblock_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const int r = blockIdx.y, t0 = blockIdx.x * (16 * MT);
  const float* xs = p.x + (int64_t)r * p.T * p.ldx;
  const __amdgpu_buffer_rsrc_t w_rs = frag_rsrc(p.wqkv_p, 96 * (TB_C / 32));
  const int lane16 = lane * 16;
  uint4 s[NS][8];
  head_ld(s[0], w_rs, lane16, wid, 0);
  head_ld(s[1], w_rs, lane16, wid, 1);
  ln_rows_to_lds<DT, MT>(xs, p.ldx, t0, p.T, p.g1, p.b1n, p.eps, smem, wid, lane);
  head_ld(s[2], w_rs, lane16, wid, 2);
  head_ld(s[3], w_rs, lane16, wid, 3);
  __syncthreads();
  head_body<DT, MT, ABL>(p, smem, s, w_rs, lane16, wid, lq, lg, r, t0);
}

template <int MT>
__device__ __forceinline__ void ln_acc(f32x4_t (&acc)[MT][2], float* red, const float* gam, const float* bet, float eps, int wid, int lq, int lg);

// ============================================================================================== tail: to_out + LN + FFN
// LDS: [0, 64 K) attention-output image (64 rows x 1024 B) during the out-projection, afterwards the xn image (64 x 512 B) at 0
// and the first GELU chunk tile (64 x 512 B) at 32 K; [64 K, 96 K) second GELU chunk tile; [96 K, 100 K) cross-wave row
// statistics; [100 K, 104 K) bf1; [104 K, 108 K) bo, g3, b3n, bf2.
// Wave w owns output columns [32 w, 32 w + 32) of the 256-wide residual row for the whole kernel (acc2: 4 x 2 tiles) and hidden
// columns [32 w, 32 w + 32) of every 256-column FFN chunk (acc1: 4 x 2 tiles).  Every fragment group is 2 tiles x 4 k-steps
// (32 MFMAs): 4 out-projection groups (k-steps 4 g ..+3 of 16), then per chunk c the groups q = 0, 1 (hidden layer, k-steps
// 4 q ..+3 of 8) and q = 2, 3 (output layer, k-steps 8 c + 4 (q - 2) ..+3 of 32).  Group G lives in ring slot G % 4 and its slot
// is refilled with group G + 4 once computed.
// NEXT = true: the kernel continues with the HEAD of the following transformer block on the same rows (cv_tblock_tail_head): the block
// output stays in the accumulators, LayerNorm(norm1 of block i + 1) is taken from them (as norm3 above), the operand image replaces the xn
// image, and the [Q | K | V^T] GEMM of head_body runs on it — the last FFN refills of the register ring already fetch the head's first
// weight groups.  One launch and one 1 KB-per-row read of x less per block; the head fields of the parameter block (g1, b1n, wqkv_p, qk,
// vt) then describe block i + 1.  LDS: two more staged vectors (g1 | b1n) behind the four of the tail.
template <int DT, bool OUTPROJ, int MT, int ABL = 0, bool NEXT = false>   // ABL: timing-only ablations (1: GELU -> identity, 2: no ring refills)
__global__ __launch_bounds__(NTHR, 2) void tblock_tail_kernel(const cv_tblock_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ximg = smem;
  char* himg0 = smem + 32768;
  char* himg1 = smem + 65536;
  float* red = (float*)(smem + 98304);                // [2][8 waves][64 rows]
  const float* b1s = (const float*)(smem + 102400);   // hidden-layer bias, staged once (a global load inside the chunk loop
                                                      // is sunk by hipcc to its use and drains the weight prefetch: vmcnt(0))
  float* vecs = (float*)(smem + 106496);              // [4][256]: bo | g3 | b3n | bf2, likewise
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const int r = blockIdx.y, t0 = blockIdx.x * (16 * MT);
  float* xs = p.x + (int64_t)r * p.T * p.ldx;
  const int ncol0 = wid * 32;
  constexpr int NC = TB_FF / HC;   // 4 chunks
  if (tid < 256) ((float4*)(smem + 102400))[tid] = ((const float4*)p.bf1)[tid];   // 1024 floats; visible after the first barrier
  vecs[tid] = tid < 256 ? (OUTPROJ ? p.bo[tid] : 0.f) : p.g3[tid - 256];
  vecs[512 + tid] = tid < 256 ? p.b3n[tid] : p.bf2[tid - 256];
  if constexpr (NEXT) vecs[1024 + tid] = tid < 256 ? p.g1[tid] : p.b1n[tid - 256];

  const __amdgpu_buffer_rsrc_t w1_rs = frag_rsrc(p.w1_p, 64 * 8);    // [64 tiles][8 ks]
  const __amdgpu_buffer_rsrc_t w2_rs = frag_rsrc(p.w2_p, 16 * 32);   // [16 tiles][32 ks]
  const int lane16 = lane * 16;
  uint4 s[NS][8];
  const __amdgpu_buffer_rsrc_t wh_rs = frag_rsrc(NEXT ? p.wqkv_p : p.w1_p, NEXT ? 96 * (TB_C / 32) : 64 * 8);   // NEXT: the following head's weights
  auto ld_ffn = [&](uint4 (&sl)[8], int f) {   // FFN group f = 4 c + q (clamped to the last chunk: harmless re-reads at the end)
    if ((ABL & 2) && f >= NS) return;
    if constexpr (NEXT) {
      if (f >= 4 * NC) { head_ld(sl, wh_rs, lane16, wid, f - 4 * NC); return; }   // ring slot f % 4 = head group f - 16 (its k-quarter)
    }
    const int c = min(f >> 2, NC - 1), q = f & 3;
    if (q < 2) {   // hidden tiles 16 c + 2 w + j, k-steps 4 q + u
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) sl[j * 4 + u] = frag_load(w1_rs, lane16, (16 * c + 2 * wid + j) * (TB_C / 32) + 4 * q + u);
    } else {       // output tiles 2 w + j, k-steps 8 c + 4 (q - 2) + u
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) sl[j * 4 + u] = frag_load(w2_rs, lane16, (2 * wid + j) * (TB_FF / 32) + 8 * c + 4 * (q - 2) + u);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // residual rows in the accumulator layout (row 16 i + lq, columns ncol0 + 16 j + 4 lg ..+3)
  f32x4_t acc2[MT][2];
  auto load_residual = [&]() {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int t = min(t0 + 16 * i + lq, p.T - 1);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc2[i][j] = *(const f32x4_t*)(xs + (int64_t)t * p.ldx + ncol0 + 16 * j + 4 * lg);
    }
  };

  if constexpr (OUTPROJ) {
    const uint16_t* aos = (const uint16_t*)p.ao + (int64_t)r * p.T * p.ldao;
    const __amdgpu_buffer_rsrc_t wo_rs = frag_rsrc(p.wo_p, 16 * 16);   // [16 tiles][16 ks]
    auto ld_o = [&](uint4 (&sl)[8], int g) {          // group g: tiles 2 w + j, k-steps 4 g + u
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) sl[j * 4 + u] = frag_load(wo_rs, lane16, (2 * wid + j) * (TB_INNER / 32) + 4 * g + u);
      __builtin_amdgcn_sched_barrier(0);
    };
    {
      // ---- attention output tile -> LDS operand image (1024-byte rows), one row per wave-instruction, 2 MT rows per wave
      constexpr int AR = 2 * MT;
      u32x4_t v[AR];
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int t = min(t0 + wid * AR + i, p.T - 1);
        v[i] = *(const u32x4_t*)(aos + (int64_t)t * p.ldao + lane * 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      ld_o(s[0], 0);
      ld_o(s[1], 1);
      pin_regs(v);
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int row = wid * AR + i;
        *(u32x4_t*)(smem + row * 1024 + (swz16(row, lane) << 4)) = v[i];
      }
    }
    ld_o(s[2], 2);
    ld_o(s[3], 3);
    load_residual();
    __syncthreads();
    auto compute_o = [&](uint4 (&sl)[8], int g) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ks = 4 * g + u;
        uint4 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *(const uint4*)(smem + (16 * i + lq) * 1024 + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) acc2[i][j] = mfma_block<DT>(sl[j * 4 + u], a[i], acc2[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) { compute_o(s[g], g); ld_ffn(s[g], g); }   // the first FFN groups fly during the LayerNorm

    // ---- + to_out bias: acc2 is now the block's first residual output x1; LayerNorm(norm3) from the accumulators
    float4 gam[2], bet[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = ncol0 + 16 * j + 4 * lg;
      const float4 bo = *(const float4*)(vecs + n);
      gam[j] = *(const float4*)(vecs + 256 + n);
      bet[j] = *(const float4*)(vecs + 512 + n);
#pragma unroll
      for (int i = 0; i < MT; ++i) { acc2[i][j][0] += bo.x; acc2[i][j][1] += bo.y; acc2[i][j][2] += bo.z; acc2[i][j][3] += bo.w; }
    }
    float mean[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) sm += (acc2[i][j][0] + acc2[i][j][1]) + (acc2[i][j][2] + acc2[i][j][3]);
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      if (lg == 0) red[wid * 64 + 16 * i + lq] = sm;
    }
    __syncthreads();   // also: every wave is done reading the attention-output image
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = 16 * i + lq;
      float sm = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sm += red[w * 64 + m];
      mean[i] = sm * (1.0f / TB_C);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = acc2[i][j][e] - mean[i]; q += d * d; }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      if (lg == 0) red[NW * 64 + wid * 64 + m] = q;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = 16 * i + lq;
      float qs = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) qs += red[NW * 64 + w * 64 + m];
      const float sc = rsqrtf(qs * (1.0f / TB_C) + p.eps);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = ncol0 + 16 * j + 4 * lg;   // 4 consecutive columns = half of 16-byte chunk n >> 3
        *(uint2*)(ximg + m * 512 + (swz16(m, n >> 3) << 4) + ((lg & 1) << 3)) =
            pack4<DT>((acc2[i][j][0] - mean[i]) * sc * gam[j].x + bet[j].x, (acc2[i][j][1] - mean[i]) * sc * gam[j].y + bet[j].y,
                      (acc2[i][j][2] - mean[i]) * sc * gam[j].z + bet[j].z, (acc2[i][j][3] - mean[i]) * sc * gam[j].w + bet[j].w);
      }
    }
  } else {
    ld_ffn(s[0], 0);
    ld_ffn(s[1], 1);
    ln_rows_to_lds<DT, MT>(xs, p.ldx, t0, p.T, p.g3, p.b3n, p.eps, ximg, wid, lane);
    ld_ffn(s[2], 2);
    ld_ffn(s[3], 3);
    load_residual();
    __syncthreads();   // vecs staged
  }
  // + FFN output bias: the accumulators then collect x1 + b2 + sum_c gelu(..) . W2_c^T
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float4 b2 = *(const float4*)(vecs + 768 + ncol0 + 16 * j + 4 * lg);
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc2[i][j][0] += b2.x; acc2[i][j][1] += b2.y; acc2[i][j][2] += b2.z; acc2[i][j][3] += b2.w; }
  }
  __syncthreads();   // xn image complete

#pragma unroll   // 4 chunks, no loop back-edge: exact vmcnt bookkeeping for the ring (see the head kernel)
  for (int c = 0; c < NC; ++c) {
    char* hb = (c & 1) ? himg1 : himg0;
    // the accumulators of the hidden layer start from its bias (this wave's two tiles of the chunk)
    f32x4_t acc1[MT][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 b1v = *(const float4*)(b1s + c * HC + (2 * wid + j) * 16 + 4 * lg);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc1[i][j] = f32x4_t{b1v.x, b1v.y, b1v.z, b1v.w};
    }
    auto compute_g1 = [&](uint4 (&sl)[8], int hf) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ks = 4 * hf + u;
        uint4 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *(const uint4*)(ximg + (16 * i + lq) * 512 + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) acc1[i][j] = mfma_block<DT>(sl[j * 4 + u], a[i], acc1[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto compute_g2 = [&](uint4 (&sl)[8], int hf) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ks = 4 * hf + u;   // k-step inside the chunk (8 x 32 = 256 hidden columns)
        uint4 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *(const uint4*)(hb + (16 * i + lq) * 512 + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) acc2[i][j] = mfma_block<DT>(sl[j * 4 + u], a[i], acc2[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    compute_g1(s[0], 0); ld_ffn(s[0], 4 * c + 0 + NS);
    compute_g1(s[1], 1); ld_ffn(s[1], 4 * c + 1 + NS);
    // GELU -> 16-bit chunk tile (row 16 i + lq, hidden columns (2 w + j) * 16 + 4 lg ..+3 of the chunk)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = 16 * i + lq;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ch = (2 * wid + j) * 2 + (lg >> 1);
        *(uint2*)(hb + m * 512 + (swz16(m, ch) << 4) + ((lg & 1) << 3)) =
            (ABL & 1) ? pack4<DT>(acc1[i][j][0], acc1[i][j][1], acc1[i][j][2], acc1[i][j][3]) :
            pack4<DT>(gelu_erf(acc1[i][j][0]), gelu_erf(acc1[i][j][1]), gelu_erf(acc1[i][j][2]), gelu_erf(acc1[i][j][3]));
      }
    }
    __syncthreads();   // one barrier per chunk: the next chunk's GELU tile goes to the other buffer
    compute_g2(s[2], 0); ld_ffn(s[2], 4 * c + 2 + NS);
    compute_g2(s[3], 1); ld_ffn(s[3], 4 * c + 3 + NS);
  }

  // ---- store the block output (fp32 residual stream, in place) + optional 16-bit copy (skip connection / next conv operand)
  if constexpr (!NEXT) {
    uint16_t* oa = p.out_act ? (uint16_t*)p.out_act + (int64_t)r * p.T * p.ldoa : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int t = t0 + 16 * i + lq;
      if (t >= p.T) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = ncol0 + 16 * j + 4 * lg;
        *(float4*)(xs + (int64_t)t * p.ldx + n) = make_float4(acc2[i][j][0], acc2[i][j][1], acc2[i][j][2], acc2[i][j][3]);
        if (oa) *(uint2*)(oa + (int64_t)t * p.ldoa + n) = pack4<DT>(acc2[i][j][0], acc2[i][j][1], acc2[i][j][2], acc2[i][j][3]);
      }
    }
  } else {
    // raw buffer stores (rows beyond T fall outside the descriptor and are dropped): no exec-masked branch between the head's weight
    // prefetch, already in flight in the ring, and its first use
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, p.T * p.ldx * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int t = t0 + 16 * i + lq;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{bitcast<uint32_t>(acc2[i][j][0]), bitcast<uint32_t>(acc2[i][j][1]),
                                                       bitcast<uint32_t>(acc2[i][j][2]), bitcast<uint32_t>(acc2[i][j][3])},
                                               x_rs, (t * p.ldx + ncol0 + 16 * j + 4 * lg) * 4, 0, 0);
    }
    // LayerNorm(norm1 of the next block) from the accumulators -> operand image at ximg (free since the last chunk's barrier: the
    // hidden layer was its last reader; the output layer of that chunk reads himg1)
    ln_acc<MT>(acc2, red, vecs + 1024, vecs + 1280, p.eps, wid, lq, lg);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = 16 * i + lq;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = ncol0 + 16 * j + 4 * lg;
        *(uint2*)(ximg + m * 512 + (swz16(m, n >> 3) << 4) + ((lg & 1) << 3)) = pack4<DT>(acc2[i][j][0], acc2[i][j][1], acc2[i][j][2], acc2[i][j][3]);
      }
    }
    __syncthreads();
    head_body<DT, MT, 0>(p, ximg, s, wh_rs, lane16, wid, lq, lg, r, t0);
  }
}

// ============================================================================================== causal resnet block
// CausalResnetBlock1D of the estimator (flow/decoder.py:36-56, flow/components/decoder.py:54-59): two launches instead of five.
//   conv1:  h1  = Mish(LayerNorm_C(causal_conv_k3(a) + b1)) + time_term                 (was: conv GEMM, layernorm)
//   conv2:  out = Mish(LayerNorm_C(causal_conv_k3(h1) + b2)) + conv_1x1(a) + br         (was: conv GEMM, layernorm, 1x1 GEMM)
// Same row-block structure as the transformer-block kernels: 8 waves on 16 MT rows, output columns split 8 ways (2 tiles per
// wave), weights streamed once per workgroup through the 4-slot register ring, the LayerNorm over the 256 channels computed
// from the accumulators.  The conv operand image holds rows t0 - 2 .. t0 + 16 MT + 1 of the input (channels-last), so the three
// taps are the same image read at row offsets 0 / 1 / 2 (k = tap * cin + ci, the packed weight's K order); rows before the
// sequence start are zeros (the causal left padding, flow/decoder.py:59-85).
__device__ __forceinline__ float mish_fast(float x) {
  // x tanh(softplus(x)) with tanh(log(1 + e^x)) = ((1 + e^x)^2 - 1) / ((1 + e^x)^2 + 1) = w / (w + 2), w = e^x (e^x + 2)
  const float n = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.4426950408889634f);
  const float w = n * (n + 2.f);
  return x * w * __builtin_amdgcn_rcpf(w + 2.f);
}

// rows [t_first, t_first + nrows) of a channels-last 16-bit tensor -> LDS operand image (pitch bytes per row, 16-byte chunk slots
// swizzled by row & 15); rows outside [0, T) of the sequence: zeros before the start, the last row repeated beyond the end
__device__ __forceinline__ void stage_rows(const uint16_t* src, int ld, int cin, int t_first, int nrows, int T, char* img, int pitch, int tid) {
  const int cpr = cin >> 3;   // 16-byte chunks per row
  const int total = nrows * cpr;
  for (int id0 = 0; id0 < total; id0 += NTHR * 4) {
    u32x4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int id = min(id0 + u * NTHR + tid, total - 1);
      const int row = id / cpr, ch = id - row * cpr;
      const int t = t_first + row;
      v[u] = *(const u32x4_t*)(src + (int64_t)min(max(t, 0), T - 1) * ld + ch * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int id = id0 + u * NTHR + tid;
      if (id >= total) continue;
      const int row = id / cpr, ch = id - row * cpr;
      const bool zero = t_first + row < 0;
      *(u32x4_t*)(img + row * pitch + (swz16(row, ch) << 4)) = zero ? u32x4_t{0u, 0u, 0u, 0u} : v[u];
    }
  }
}

// LayerNorm over the 256 channels of rows held in the accumulator layout (wave w: columns 32 w + 16 j + 4 lg ..+3): statistics
// through `red` ([2][NW][64] floats).  On return acc holds (x - mean) * rstd * gamma + beta.  Two workgroup barriers.
template <int MT>
__device__ __forceinline__ void ln_acc(f32x4_t (&acc)[MT][2], float* red, const float* gam, const float* bet, float eps, int wid, int lq, int lg) {
  float mean[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) sm += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    if (lg == 0) red[wid * 64 + 16 * i + lq] = sm;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = 16 * i + lq;
    float sm = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sm += red[w * 64 + m];
    mean[i] = sm * (1.0f / TB_C);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = acc[i][j][e] - mean[i]; acc[i][j][e] = d; q += d * d; }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (lg == 0) red[NW * 64 + wid * 64 + m] = q;
  }
  __syncthreads();
  float4 g4[2], b4[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    g4[j] = *(const float4*)(gam + wid * 32 + 16 * j + 4 * lg);
    b4[j] = *(const float4*)(bet + wid * 32 + 16 * j + 4 * lg);
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = 16 * i + lq;
    float qs = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) qs += red[NW * 64 + w * 64 + m];
    const float sc = rsqrtf(qs * (1.0f / TB_C) + eps);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[i][j][0] = acc[i][j][0] * sc * g4[j].x + b4[j].x; acc[i][j][1] = acc[i][j][1] * sc * g4[j].y + b4[j].y;
      acc[i][j][2] = acc[i][j][2] * sc * g4[j].z + b4[j].z; acc[i][j][3] = acc[i][j][3] * sc * g4[j].w + b4[j].w;
    }
  }
}

// STAGE 1: a (cin channels) -> h1;  STAGE 2: h1 (256 channels) + a -> out.  cin in {256, 320, 512}.
// LDS: conv operand image at 0 (rows t0 - 2 ..., pitch round_up(K row bytes, 256)), stage 2's 1x1 operand image behind it,
// then [2][NW][64] statistics and the staged vectors (bias | gamma | beta | time term or 1x1 bias).
template <int DT, int STAGE, int MT>
__global__ __launch_bounds__(NTHR, 2) void resblock_kernel(const cv_resblock_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const int r = blockIdx.y, t0 = blockIdx.x * (16 * MT);
  const int cinc = STAGE == 1 ? p.cin : TB_C;                  // channels of the k3 conv's input
  const int pitch = ((cinc * 2 + 255) >> 8) << 8;              // bytes per image row (>= 16 chunks, multiple of 16 chunks)
  const int nrows = 16 * MT + 4;                               // t0 - 2 .. t0 + 16 MT + 1 (the last row only meets zero-padded k-steps)
  char* cimg = smem;
  const int pitch_a = ((p.cin * 2 + 255) >> 8) << 8;
  char* aimg = smem + nrows * pitch;                           // STAGE 2: rows t0 .. of `a` for the 1x1 conv
  float* red = (float*)(smem + nrows * pitch + (STAGE == 2 ? 16 * MT * pitch_a : 0));
  float* vecs = red + 2 * NW * 64;                             // [4][256]
  {
    const float* v0 = STAGE == 1 ? p.b1 : p.b2;
    const float* v1 = STAGE == 1 ? p.g1 : p.g2;
    const float* v2 = STAGE == 1 ? p.be1 : p.be2;
    const float* v3 = STAGE == 1 ? p.tadd : p.br;
    vecs[tid] = tid < 256 ? v0[tid] : v1[tid - 256];
    vecs[512 + tid] = tid < 256 ? v2[tid] : (v3 ? v3[tid - 256] : 0.f);
  }
  const int nks = (3 * cinc + 31) >> 5;          // k-steps of the k3 conv
  const int nks_pad = (nks + 3) & ~3;            // the packed weights are zero-padded to whole groups of 4 k-steps
  const int kpt = cinc >> 5;                     // k-steps per tap
  const __amdgpu_buffer_rsrc_t wc_rs = frag_rsrc(STAGE == 1 ? p.w1_p : p.w2_p, 16 * nks_pad);
  const int lane16 = lane * 16;
  uint4 s[NS][8];
  const int ngc = nks_pad >> 2;                  // groups of the k3 conv (2 tiles x 4 k-steps each)
  const int nkr = p.cin >> 5, ngr = STAGE == 2 ? (nkr + 3) >> 2 : 0;   // 1x1 conv: k-steps, groups (weights zero-padded likewise)
  const __amdgpu_buffer_rsrc_t wr_rs = frag_rsrc(STAGE == 2 ? p.wr_p : p.w1_p, 16 * (ngr > 0 ? ngr * 4 : 1));
  // group G: conv groups 0 .. ngc - 1, then the 1x1 groups; beyond the end: harmless re-reads of the last group
  auto ld = [&](uint4 (&sl)[8], int G) {
    const int Gc = min(G, ngc + ngr - 1);
    if (Gc < ngc) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) sl[j * 4 + u] = frag_load(wc_rs, lane16, (2 * wid + j) * nks_pad + 4 * Gc + u);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) sl[j * 4 + u] = frag_load(wr_rs, lane16, (2 * wid + j) * (ngr * 4) + 4 * (Gc - ngc) + u);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  ld(s[0], 0);
  ld(s[1], 1);
  const uint16_t* src = (const uint16_t*)(STAGE == 1 ? p.a : p.h1) + (int64_t)r * p.T * (STAGE == 1 ? p.lda : p.ldh1);
  stage_rows(src, STAGE == 1 ? p.lda : p.ldh1, cinc, t0 - 2, nrows, p.T, cimg, pitch, tid);
  if constexpr (STAGE == 2)
    stage_rows((const uint16_t*)p.a + (int64_t)r * p.T * p.lda, p.lda, p.cin, t0, 16 * MT, p.T, aimg, pitch_a, tid);
  ld(s[2], 2);
  ld(s[3], 3);
  __syncthreads();

  f32x4_t acc[MT][2], accr[MT][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float4 b = *(const float4*)(vecs + wid * 32 + 16 * j + 4 * lg);
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][j] = f32x4_t{b.x, b.y, b.z, b.w}; accr[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  }
  // k3 conv: k-step ks = tap * kpt + kc reads image row (16 i + lq + tap), chunk 4 kc + lg
  for (int g0 = 0; g0 < ngc; g0 += NS) {
#pragma unroll
    for (int gg = 0; gg < NS; ++gg) {
      const int g = g0 + gg;
      if (g < ngc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ks = min(4 * g + u, nks - 1 + 0 * nks);   // padded k-steps (zero weights) re-read the last real one
          const int tap = ks / kpt, kc = ks - tap * kpt;
          uint4 a[MT];
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int row = 16 * i + lq + tap;
            a[i] = *(const uint4*)(cimg + row * pitch + (swz16(row, kc * 4 + lg) << 4));
          }
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = mfma_block<DT>(s[gg][j * 4 + u], a[i], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        ld(s[gg], g + NS);
      }
    }
  }
  if constexpr (STAGE == 2) {
    // 1x1 conv of the block input into its own accumulators; its groups continue the ring numbering at ngc
    for (int g0 = 0; g0 < ngr; g0 += NS) {
#pragma unroll
      for (int gg = 0; gg < NS; ++gg) {
        const int g = g0 + gg;             // 1x1 group index; ring slot (ngc + g) % NS
        if (g < ngr) {
          const int slot = (ngc + g) % NS;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ks = min(4 * g + u, nkr - 1);
            uint4 a[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *(const uint4*)(aimg + (16 * i + lq) * pitch_a + (swz16(lq, ks * 4 + lg) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int i = 0; i < MT; ++i) {
                // slot index must be a compile-time constant for the ring to stay in registers: select over the 4 slots
                const uint4 wv = slot == 0 ? s[0][j * 4 + u] : (slot == 1 ? s[1][j * 4 + u] : (slot == 2 ? s[2][j * 4 + u] : s[3][j * 4 + u]));
                accr[i][j] = mfma_block<DT>(wv, a[i], accr[i][j]);
              }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  ln_acc<MT>(acc, red, vecs + 256, vecs + 512, p.eps, wid, lq, lg);
  // Mish, then the per-channel time term (stage 1) or the 1x1 conv + its bias (stage 2); store
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = wid * 32 + 16 * j + 4 * lg;
    const float4 ad = *(const float4*)(vecs + 768 + n);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int t = t0 + 16 * i + lq;
      float o0 = mish_fast(acc[i][j][0]) + ad.x, o1 = mish_fast(acc[i][j][1]) + ad.y;
      float o2 = mish_fast(acc[i][j][2]) + ad.z, o3 = mish_fast(acc[i][j][3]) + ad.w;
      if constexpr (STAGE == 2) { o0 += accr[i][j][0]; o1 += accr[i][j][1]; o2 += accr[i][j][2]; o3 += accr[i][j][3]; }
      if (t >= p.T) continue;
      if constexpr (STAGE == 1) *(uint2*)((uint16_t*)p.h1 + ((int64_t)r * p.T + t) * p.ldh1 + n) = pack4<DT>(o0, o1, o2, o3);
      else *(float4*)(p.out + ((int64_t)r * p.T + t) * p.ldo + n) = make_float4(o0, o1, o2, o3);
    }
  }
}

int check_common(const cv_tblock_params& p) {
  if (p.dtype != CV_BF16 && p.dtype != CV_F16) return CV_ERR_UNSUPPORTED;
  if (p.C != TB_C || p.inner != TB_INNER || p.ff != TB_FF) return CV_ERR_UNSUPPORTED;
  if (p.R <= 0 || p.T <= 0 || !p.x || (p.ldx & 3) || ((uintptr_t)p.x & 15)) return CV_ERR_ARG;
  return CV_OK;
}

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

template <typename K>
void set_lds(K kern, size_t lds) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }

constexpr size_t TAIL_LDS = 98304 + 4096 + 4096 + 4096 + 2048;
template <int DT, bool OP, int MT, int ABL, bool NEXT = false>
void launch_tail(const cv_tblock_params& p, hipStream_t st) {
  static PerDeviceOnce once;   // > 64 KiB of dynamic LDS needs the opt-in
  once.run([] { set_lds(tblock_tail_kernel<DT, OP, MT, ABL, NEXT>, TAIL_LDS); });
  dim3 grid((p.T + 16 * MT - 1) / (16 * MT), p.R);
  hipLaunchKernelGGL((tblock_tail_kernel<DT, OP, MT, ABL, NEXT>), grid, dim3(NTHR), TAIL_LDS, st, p);
}
template <int DT, int MT, int ABL>
void launch_head(const cv_tblock_params& p, hipStream_t st) {
  dim3 grid((p.T + 16 * MT - 1) / (16 * MT), p.R);
  hipLaunchKernelGGL((tblock_head_kernel<DT, MT, ABL>), grid, dim3(NTHR), 32768, st, p);
}
// rows per tile (16 MT, MT = 1 .. 4): the one that minimises (rounds of workgroups over the CUs the launch may use) x (time of one
// workgroup); one workgroup per CU is resident.  A workgroup's time is NOT proportional to its rows: every workgroup streams the block's
// whole weight set through its CU (0.75 - 1.25 MB at ~150 GB/s per CU) and pays the dependent chain of phases once, so the model is
// fixed + rows (fixed = MT_FIXED row tiles' worth; tools/tblock_bench.py at R = 2, T = 500: head 12.2 / 14.2 / 16.8 / 20.6 us and tail
// 15.6 / 19.0 / 23.1 / 27.3 us for MT = 1 / 2 / 3 / 4, i.e. about 9.4 + 2.8 MT and 11.7 + 3.9 MT).  At the
// batch-8 shapes (R = 16, T = 1000: 256 or 336 workgroups) it reproduces the r02 choice (64 rows on the whole chip, 48 on 192 CUs);
// at batch 1 (R = 2) it picks 16-row tiles (126 workgroups instead of 32 on 256 CUs).  p.cus = 0 means the whole chip.
// CV_TBLOCK_MT=1..4 overrides (tuning aid, read per call: tests switch it).
constexpr int MT_FIXED = 3;
int pick_mt(const cv_tblock_params& p) {
  const int forced = env_int("CV_TBLOCK_MT", 0);
  if (forced >= 1 && forced <= 4) return forced;
  const int cus = p.cus > 0 ? p.cus : 256;
  int best = 4;
  int64_t best_cost = INT64_MAX;
  for (int mt = 4; mt >= 1; --mt) {   // ties go to the larger tile (fewer weight re-reads)
    const int wgs = p.R * ((p.T + 16 * mt - 1) / (16 * mt));
    const int64_t cost = (int64_t)((wgs + cus - 1) / cus) * (mt + MT_FIXED);
    if (cost < best_cost) { best_cost = cost; best = mt; }
  }
  return best;
}
template <int DT>
void dispatch_head(const cv_tblock_params& p, hipStream_t st) {
  if constexpr (DT == CV_F16) {   // timing-only ablation builds (wrong results): CV_TBLOCK_ABL=1|2|3
    static const int abl = env_int("CV_TBLOCK_ABL", 0);
    if (abl == 1) return launch_head<DT, 4, 1>(p, st);
    if (abl == 2) return launch_head<DT, 4, 2>(p, st);
    if (abl == 3) return launch_head<DT, 4, 3>(p, st);
  }
  switch (pick_mt(p)) {
    case 1: return launch_head<DT, 1, 0>(p, st);
    case 2: return launch_head<DT, 2, 0>(p, st);
    case 3: return launch_head<DT, 3, 0>(p, st);
    default: return launch_head<DT, 4, 0>(p, st);
  }
}
template <int DT>
void dispatch_tail_head(const cv_tblock_params& p, hipStream_t st) {
  switch (pick_mt(p)) {
    case 1: return launch_tail<DT, true, 1, 0, true>(p, st);
    case 2: return launch_tail<DT, true, 2, 0, true>(p, st);
    case 3: return launch_tail<DT, true, 3, 0, true>(p, st);
    default: return launch_tail<DT, true, 4, 0, true>(p, st);
  }
}
template <int DT, bool OP>
void dispatch_tail(const cv_tblock_params& p, hipStream_t st) {
  if constexpr (DT == CV_F16 && OP) {
    static const int abl = env_int("CV_TBLOCK_ABL", 0);
    if (abl == 1) return launch_tail<DT, OP, 4, 1>(p, st);
    if (abl == 2) return launch_tail<DT, OP, 4, 2>(p, st);
    if (abl == 3) return launch_tail<DT, OP, 4, 3>(p, st);
  }
  switch (pick_mt(p)) {
    case 1: return launch_tail<DT, OP, 1, 0>(p, st);
    case 2: return launch_tail<DT, OP, 2, 0>(p, st);
    case 3: return launch_tail<DT, OP, 3, 0>(p, st);
    default: return launch_tail<DT, OP, 4, 0>(p, st);
  }
}

}  // namespace

extern "C" int cv_sizeof_tblock_params(void) { return (int)sizeof(cv_tblock_params); }

extern "C" int cv_tblock_head(const cv_tblock_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  const cv_tblock_params p = *pp;
  if (int rc = check_common(p)) return rc;
  if (!p.g1 || !p.b1n || !p.wqkv_p || !p.qk || !p.vt) return CV_ERR_ARG;
  if ((p.ldqk & 7) || (p.vt_ld & 3) || p.vt_ld < p.T || ((uintptr_t)p.qk & 15) || ((uintptr_t)p.vt & 7) || ((uintptr_t)p.wqkv_p & 15))
    return CV_ERR_ARG;   // [Q | K] rows are written with 16-byte stores
  if ((int64_t)p.T * p.ldqk * 2 >= (1ll << 31) || (int64_t)TB_INNER * p.vt_ld * 2 >= 0x7FFFFFF0ll) return CV_ERR_ARG;   // 32-bit buffer offsets
  hipStream_t st = (hipStream_t)stream;
  if (p.dtype == CV_BF16) dispatch_head<CV_BF16>(p, st);
  else dispatch_head<CV_F16>(p, st);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_tblock_tail(const cv_tblock_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  const cv_tblock_params p = *pp;
  if (int rc = check_common(p)) return rc;
  if (!p.g3 || !p.b3n || !p.w1_p || !p.bf1 || !p.w2_p || !p.bf2) return CV_ERR_ARG;
  if (((uintptr_t)p.w1_p & 15) || ((uintptr_t)p.w2_p & 15) || ((uintptr_t)p.bf1 & 15)) return CV_ERR_ARG;
  if (p.out_act && ((p.ldoa & 3) || ((uintptr_t)p.out_act & 7))) return CV_ERR_ARG;
  const bool outproj = p.ao != nullptr;
  if (outproj && (!p.wo_p || !p.bo || (p.ldao & 7) || ((uintptr_t)p.ao & 15) || ((uintptr_t)p.wo_p & 15))) return CV_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (p.dtype == CV_BF16) { if (outproj) dispatch_tail<CV_BF16, true>(p, st); else dispatch_tail<CV_BF16, false>(p, st); }
  else { if (outproj) dispatch_tail<CV_F16, true>(p, st); else dispatch_tail<CV_F16, false>(p, st); }
  CV_CHECK_LAUNCH();
  return CV_OK;
}

// tail of block i + head of block i + 1 in one launch: tail fields = block i (to_out path required), head fields = block i + 1
extern "C" int cv_tblock_tail_head(const cv_tblock_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  const cv_tblock_params p = *pp;
  if (int rc = check_common(p)) return rc;
  if (!p.g3 || !p.b3n || !p.w1_p || !p.bf1 || !p.w2_p || !p.bf2 || !p.ao || !p.wo_p || !p.bo) return CV_ERR_ARG;
  if (((uintptr_t)p.w1_p & 15) || ((uintptr_t)p.w2_p & 15) || ((uintptr_t)p.bf1 & 15) || (p.ldao & 7) || ((uintptr_t)p.ao & 15) || ((uintptr_t)p.wo_p & 15))
    return CV_ERR_ARG;
  if (p.out_act) return CV_ERR_ARG;   // the 16-bit copy belongs to the LAST block of a group, which has no following head
  if (!p.g1 || !p.b1n || !p.wqkv_p || !p.qk || !p.vt) return CV_ERR_ARG;
  if ((p.ldqk & 7) || (p.vt_ld & 3) || p.vt_ld < p.T || ((uintptr_t)p.qk & 15) || ((uintptr_t)p.vt & 7) || ((uintptr_t)p.wqkv_p & 15))
    return CV_ERR_ARG;
  if ((int64_t)p.T * p.ldqk * 2 >= (1ll << 31) || (int64_t)TB_INNER * p.vt_ld * 2 >= 0x7FFFFFF0ll || (int64_t)p.T * p.ldx * 4 >= (1ll << 31))
    return CV_ERR_ARG;   // 32-bit buffer offsets
  hipStream_t st = (hipStream_t)stream;
  if (p.dtype == CV_BF16) dispatch_tail_head<CV_BF16>(p, st);
  else dispatch_tail_head<CV_F16>(p, st);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

namespace {
template <int DT, int STAGE, int MT>
int launch_resblock(const cv_resblock_params& p, hipStream_t st) {
  const int cinc = STAGE == 1 ? p.cin : TB_C;
  const int pitch = ((cinc * 2 + 255) >> 8) << 8, pitch_a = ((p.cin * 2 + 255) >> 8) << 8;
  const size_t lds = (size_t)(16 * MT + 4) * pitch + (STAGE == 2 ? (size_t)16 * MT * pitch_a : 0) + 2 * NW * 64 * 4 + 4096;
  if (lds > 160 * 1024) return CV_ERR_UNSUPPORTED;
  static PerDeviceOnce once;
  once.run([] { set_lds(resblock_kernel<DT, STAGE, MT>, 160 * 1024); });
  dim3 grid((p.T + 16 * MT - 1) / (16 * MT), p.R);
  hipLaunchKernelGGL((resblock_kernel<DT, STAGE, MT>), grid, dim3(NTHR), lds, st, p);
  return hipGetLastError() == hipSuccess ? CV_OK : CV_ERR_LAUNCH;
}
template <int STAGE>
int dispatch_resblock(const cv_resblock_params& p, hipStream_t st) {
  cv_tblock_params q{};   // tile-size choice shared with the transformer-block kernels
  q.R = p.R; q.T = p.T; q.cus = p.cus;
  const int mt = pick_mt(q);
  if (p.dtype == CV_BF16) {
    switch (mt) {
      case 1: return launch_resblock<CV_BF16, STAGE, 1>(p, st);
      case 2: return launch_resblock<CV_BF16, STAGE, 2>(p, st);
      case 3: return launch_resblock<CV_BF16, STAGE, 3>(p, st);
      default: return launch_resblock<CV_BF16, STAGE, 4>(p, st);
    }
  }
  switch (mt) {
    case 1: return launch_resblock<CV_F16, STAGE, 1>(p, st);
    case 2: return launch_resblock<CV_F16, STAGE, 2>(p, st);
    case 3: return launch_resblock<CV_F16, STAGE, 3>(p, st);
    default: return launch_resblock<CV_F16, STAGE, 4>(p, st);
  }
}
int check_resblock(const cv_resblock_params& p, int stage) {
  if (p.dtype != CV_BF16 && p.dtype != CV_F16) return CV_ERR_UNSUPPORTED;
  if (p.C != TB_C || (p.cin != 256 && p.cin != 320 && p.cin != 512)) return CV_ERR_UNSUPPORTED;
  if (p.R <= 0 || p.T <= 0 || !p.a || (p.lda & 7) || p.lda < p.cin || ((uintptr_t)p.a & 15) || !p.h1 || (p.ldh1 & 7) || ((uintptr_t)p.h1 & 15)) return CV_ERR_ARG;
  if (stage == 1 && (!p.w1_p || !p.b1 || !p.g1 || !p.be1 || ((uintptr_t)p.w1_p & 15))) return CV_ERR_ARG;
  if (stage == 2 && (!p.w2_p || !p.b2 || !p.g2 || !p.be2 || !p.wr_p || !p.br || !p.out || (p.ldo & 3) || ((uintptr_t)p.out & 15) ||
                     ((uintptr_t)p.w2_p & 15) || ((uintptr_t)p.wr_p & 15))) return CV_ERR_ARG;
  return CV_OK;
}
}  // namespace

extern "C" int cv_sizeof_resblock_params(void) { return (int)sizeof(cv_resblock_params); }
extern "C" int cv_resblock_conv1(const cv_resblock_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  if (int rc = check_resblock(*pp, 1)) return rc;
  return dispatch_resblock<1>(*pp, (hipStream_t)stream);
}
extern "C" int cv_resblock_conv2(const cv_resblock_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  if (int rc = check_resblock(*pp, 2)) return rc;
  return dispatch_resblock<2>(*pp, (hipStream_t)stream);
}
