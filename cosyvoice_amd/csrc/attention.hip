// cv_attention: flash attention for gfx950, head_dim 64, 16-bit operands (bf16 / fp16), fp32 softmax + accumulation.
//
// Workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries (two 16-wide MFMA column tiles).
// Scores are computed TRANSPOSED, S^T = K . Q^T (MFMA rows = keys, cols = queries), so that
//   * a lane holds 16 keys of ONE query per 64-key tile: the softmax row reduction is in-lane + 2 cross-lane steps;
//   * the S^T accumulator registers are, after exp and 16-bit packing, directly the B operand of O^T += V^T . P^T
//     (k-slot (g,j<4) = key 16*kt0+4g+j, (g,j>=4) = key 16*kt1+4g+j-4): no LDS round trip for P;
//   * V arrives pre-transposed (V^T [d][key], written by the QKV producer), so its A fragment is ONE 16-byte
//     LDS read per lane from a [64 d][128 B] image with XOR-swizzled 16-byte chunk slots.
// K / V^T tiles (64 keys) are staged through LDS, double-buffered with register prefetch (one barrier per tile).
#include <cstdlib>
#include <type_traits>
#include "cv_device.h"

namespace {

constexpr int KT_BYTES = 8192;        // K tile: 64 keys x 128 B, row-major with XOR-swizzled 16-byte chunk slots
constexpr int VT_PITCH = 128;         // bytes per d-row of the V^T tile; 16-byte chunk slots XOR-swizzled by (row >> 1) & 7
constexpr int VT_BYTES = 64 * VT_PITCH;
constexpr int STAGE_BYTES = KT_BYTES + VT_BYTES;
constexpr float NEG_BIG = -1e30f;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// max over the four lanes that hold one query's scores (lane, lane^16, lane^32, lane^48): two v_permlane*_swap + v_max
// instead of two ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float xlane_max(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t w = __float_as_uint(m);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// launch bound 3 workgroups per CU (168 VGPRs, no spills): the kernel is stall-bound, not issue-bound (3 600 cycles per
// tile-wave against ~900 of issue), so a third resident wave per SIMD bought 97 -> 83 us; a fourth needs 128 VGPRs and spills.
// NWV = waves per workgroup: 4 (128 queries; the batch-8 shapes) or 2 (64 queries: twice the workgroups for grids that would
// leave most of the chip idle — one utterance, R = 2 CFG rows x 8 heads x T / 128 query blocks is 64 - 128 workgroups on 256 CUs).
template <int DT, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 3 : 2) void attn_kernel(const cv_attn_params p) {
  constexpr int NT = 64 * NWV, QW = 32 * NWV, NL = 512 / NT;   // threads, queries per workgroup, 16-byte chunks per thread per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (tile counts and buffer soffsets derive from it)
  const int lq = lane & 15, lg = lane >> 4;
  // Key order inside a 64-key tile: row lq of S^T sub-tile kt is key krow(kt) = (kt>>1)*32 + (lq>>2)*8 + (kt&1)*4 + (lq&3), so
  // that the scores a lane holds for sub-tiles 2s and 2s+1 are 8 CONSECUTIVE keys (s*32 + lg*8 ..+7): P^T is then already
  // in MFMA B-operand order and the V^T fragment is one 16-byte LDS read instead of two 8-byte ones.  The K image keeps
  // its 128-byte rows; the XOR swizzle is keyed on ((row>>1)&1) | ((row>>2)&6) so the 8 same-parity rows a 16-lane read
  // pass touches (0,2,8,10,16,18,24,26 + const) land in 8 different 16-byte slots.
  const int krow_lo = ((lq >> 2) << 3) + (lq & 3);
  auto kswz = [](int row) { return ((row >> 1) & 1) | ((row >> 2) & 6); };
  int kfrag0[4], kfrag1[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int row = ((kt >> 1) << 5) + ((kt & 1) << 2) + krow_lo;
    kfrag0[kt] = (row << 7) + ((lg ^ kswz(row)) << 4);
    kfrag1[kt] = (row << 7) + (((4 + lg) ^ kswz(row)) << 4);
  }
  // grid = (heads, query blocks, batch): consecutive workgroup ids go round-robin over the 8 XCDs, so with heads fastest
  // all query blocks of one (batch, head) run on ONE XCD and its K / V^T (256 KB at T = 1000) is fetched into one L2, not eight.
  const int b = blockIdx.z, h = blockIdx.x;
  const int hk = h / (p.H / p.Hkv);
  const int q_wg = blockIdx.y * QW;
  const int q0 = q_wg + wid * 32;

  const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + h * p.q_hs;
  const uint16_t* Kp = (const uint16_t*)p.k + (int64_t)b * p.k_bs + hk * p.k_hs;
  const uint16_t* Vt = (const uint16_t*)p.vt + (int64_t)(b * p.Hkv + hk) * 64 * p.vt_ld;
  const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;

  // keys this workgroup can ever need
  int limit = klen;
  const int q_max = min(p.Tq, q_wg + QW) - 1;
  if (p.causal) limit = min(limit, q_max + p.causal_off + 1);
  if (p.chunk > 0) limit = min(limit, ((q_max + p.q_off) / p.chunk + 1) * p.chunk);
  const int ntiles = (limit + 63) >> 6;

  // ---- Q fragments (B operand of S^T): lane = query (lq), d chunk = ks*4 + lg.  Loads are unconditional (clamped
  // rows, masked afterwards): a load under a per-element branch costs a vmcnt(0) wait each.
  uint4 qf[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int row = q0 + qt * 16 + lq;
    const uint32_t msk = row < p.Tq ? 0xFFFFFFFFu : 0u;
    const int rc = min(row, p.Tq - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 v = *(const uint4*)(Q + (int64_t)rc * p.ldq + (ks * 4 + lg) * 8);
      qf[qt][ks] = make_uint4(v.x & msk, v.y & msk, v.z & msk, v.w & msk);
    }
  }

  // K / V^T tile loads go through buffer descriptors: the per-thread offset is fixed, the tile offset is an SGPR (no
  // per-tile address VALU), and K rows >= klen fall outside num_records and read as zeros (no clamp, no mask).
  u32x4_t rk[NL], rv[NL];
  int tile_j0 = 0;  // key offset of the prefetched tile (the V^T tail mask is applied when it is written to LDS)
  const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Kp, 0, klen > 0 ? (int)min(((int64_t)(klen - 1) * p.ldk + 64) * 2, (int64_t)0x7FFFFFFF) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Vt, 0, (int)min((int64_t)64 * p.vt_ld * 2, (int64_t)0x7FFFFFFF), 0x00020000);
  int koff[NL], voff[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int c = i * NT + tid;
    const int r = c >> 3, dc = c & 7;  // K: 8 consecutive lanes = one 128-byte key row; V^T: r = d row, dc = key chunk
    koff[i] = (r * p.ldk + dc * 8) * 2;
    voff[i] = (r * p.vt_ld + dc * 8) * 2;
  }
  auto load_tile = [&](int t) {
    const int j0 = t << 6;
    tile_j0 = j0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      rk[i] = __builtin_amdgcn_raw_buffer_load_b128(k_rs, koff[i], j0 * p.ldk * 2, 0);
      rv[i] = __builtin_amdgcn_raw_buffer_load_b128(v_rs, voff[i], j0 * 2, 0);
    }
  };
  auto store_tile = [&](int s) {
    char* sk = smem + s * STAGE_BYTES;
    char* sv = sk + KT_BYTES;
    const bool tail = tile_j0 + 64 > klen;  // wave-uniform: only the last tile of a sequence needs the V^T key mask
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int c = i * NT + tid;
      const int r = c >> 3, dc = c & 7;
      *(u32x4_t*)(sk + (r << 7) + ((dc ^ kswz(r)) << 4)) = rk[i];  // row-major, XOR-swizzled chunk slot
      u32x4_t u = rv[i];
      if (tail) {  // zero every key >= klen of the V^T chunk (0 * garbage must not become NaN)
        const int nvalid = klen - (tile_j0 + dc * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t m_lo = (2 * e < nvalid) ? 0x0000FFFFu : 0u;
          const uint32_t m_hi = (2 * e + 1 < nvalid) ? 0xFFFF0000u : 0u;
          u[e] &= (m_lo | m_hi);
        }
      }
      // V^T image: 128-byte rows, chunk slot XOR (r >> 1) & 7 — 8 lanes fill one row (32 banks), the next 8 the other 32;
      // a fragment read (16 consecutive rows, one chunk) hits 8 distinct slots per bank half.  The padded 144-byte pitch it
      // replaces left 30 % of the kernel's LDS cycles as bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
      *(u32x4_t*)(sv + r * VT_PITCH + ((dc ^ ((r >> 1) & 7)) << 4)) = u;
    }
  };

  f32x4_t oacc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) oacc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float mrun[2] = {NEG_BIG, NEG_BIG};
  f32x2_t lrun[2] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}};
  const float sc = p.scale * 1.4426950408889634f;  // fold log2(e): softmax via exp2

  if (ntiles > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  // One 64-key tile.  MASKED = false is the interior version (every key of the tile visible to every query of the wave,
  // no bias): no key-index compares, no limit arithmetic, no branches between the MFMA blocks.  Both versions end in the
  // same single barrier, so waves of one workgroup may run different versions of the same tile.
  auto tile = [&](auto masked_tag, const int t) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int s = t & 1;
    if (t + 1 < ntiles) load_tile(t + 1);
    const char* sk = smem + s * STAGE_BYTES;
    const char* sv = sk + KT_BYTES;
    const int j0 = t << 6;

    // ---- S^T = K . Q^T
    f32x4_t sacc[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const uint4 k0 = *(const uint4*)(sk + kfrag0[kt]);
      const uint4 k1 = *(const uint4*)(sk + kfrag1[kt]);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
        a = mfma_block<DT>(k0, qf[qt][0], a);
        a = mfma_block<DT>(k1, qf[qt][1], a);
        sacc[kt][qt] = a;
      }
    }

    // ---- masks, online softmax (per query = per lane column).  r01 spent 730 VALU instructions per tile here, r02-a
    // ~450 (ACTIVE_INST_ANY, PMC): now the max is a v_max3 chain on raw scores (scale > 0 commutes with max) reduced over
    // lanes with two permlane swaps, scale and max-subtraction are packed fp32 FMAs feeding v_exp_f32, row sums and the
    // alpha rescale are packed too.
    uint4 pf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mnew;
      bool dead = false;  // query with no visible key so far
      if constexpr (MASKED) {
        const int i = q0 + qt * 16 + lq;
        int jlim = klen;
        if (p.causal) jlim = min(jlim, i + p.causal_off + 1);
        if (p.chunk > 0) jlim = min(jlim, ((i + p.q_off) / p.chunk + 1) * p.chunk);
        if (p.bias != nullptr) {
          const float* brow = p.bias + (int64_t)b * p.bias_bs + (int64_t)h * p.bias_hs + (int64_t)min(i, p.Tq - 1) * p.bias_ld;
          // the lane's 16 bias values (two runs of 8 consecutive keys; the rel-pos view has no 16-byte alignment) as ONE batch of loads:
          // written next to their use, hipcc paired every load with a vmcnt wait (16 dependent L2 round trips per q-tile)
          float bv[4][4];
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = j0 + ((kt >> 1) << 5) + (lg << 3) + ((kt & 1) << 2) + r;
              bv[kt][r] = brow[min(j, p.Tk - 1)];
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = j0 + ((kt >> 1) << 5) + (lg << 3) + ((kt & 1) << 2) + r;
              const float v = fmaf(sacc[kt][qt][r], sc, bv[kt][r] * 1.4426950408889634f);
              sacc[kt][qt][r] = (j < jlim) ? v : NEG_BIG;
            }
        } else {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = j0 + ((kt >> 1) << 5) + (lg << 3) + ((kt & 1) << 2) + r;
              sacc[kt][qt][r] = (j < jlim) ? sacc[kt][qt][r] * sc : NEG_BIG;
            }
        }
      }
      // 16 in-lane scores -> v_max3 chains (the file is built with -fno-honor-nans: no canonicalize per operand)
      float mx = fmaxf(fmaxf(sacc[0][qt][0], sacc[0][qt][1]), sacc[0][qt][2]);
      mx = fmaxf(fmaxf(mx, sacc[0][qt][3]), sacc[1][qt][0]);
      mx = fmaxf(fmaxf(mx, sacc[1][qt][1]), sacc[1][qt][2]);
      float my = fmaxf(fmaxf(sacc[1][qt][3], sacc[2][qt][0]), sacc[2][qt][1]);
      my = fmaxf(fmaxf(my, sacc[2][qt][2]), sacc[2][qt][3]);
      my = fmaxf(fmaxf(my, sacc[3][qt][0]), sacc[3][qt][1]);
      my = fmaxf(fmaxf(my, sacc[3][qt][2]), sacc[3][qt][3]);
      mx = xlane_max(fmaxf(mx, my));
      // interior tiles hold raw scores (scale folded into the FMA below), masked tiles hold scaled ones
      const float scl = MASKED ? 1.0f : sc;
      mnew = fmaxf(mrun[qt], mx * scl);
      if constexpr (MASKED) dead = mnew <= 0.5f * NEG_BIG;
      const f32x2_t sc2 = f32x2_t{scl, scl}, nm2 = f32x2_t{-mnew, -mnew};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const f32x2_t lo = f32x2_t{sacc[kt][qt][0], sacc[kt][qt][1]} * sc2 + nm2;
        const f32x2_t hi = f32x2_t{sacc[kt][qt][2], sacc[kt][qt][3]} * sc2 + nm2;
        // v_exp_f32 directly (exp2f's denormal handling costs four more VALU per element; a flushed 2^-127 is 0 here anyway)
        sacc[kt][qt][0] = __builtin_amdgcn_exp2f(lo[0]);
        sacc[kt][qt][1] = __builtin_amdgcn_exp2f(lo[1]);
        sacc[kt][qt][2] = __builtin_amdgcn_exp2f(hi[0]);
        sacc[kt][qt][3] = __builtin_amdgcn_exp2f(hi[1]);
      }
      if constexpr (MASKED) {
        // a query with no visible key so far (negative causal_off, klen 0): every masked score equals the running "max" and
        // exp2(0) = 1 would count the masked keys.  Zero the probabilities while the running max is still the mask value: the
        // row ends as zeros.  (An interior tile cannot hold such a row.)
        if (dead) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) sacc[kt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
      }
      const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
      const bool same_max = __all(mnew == mrun[qt]);  // running max unchanged for every query of the wave: alpha == 1
      mrun[qt] = mnew;
      f32x2_t ls = f32x2_t{sacc[0][qt][0], sacc[0][qt][1]} + f32x2_t{sacc[0][qt][2], sacc[0][qt][3]};
#pragma unroll
      for (int kt = 1; kt < 4; ++kt)
        ls += f32x2_t{sacc[kt][qt][0], sacc[kt][qt][1]} + f32x2_t{sacc[kt][qt][2], sacc[kt][qt][3]};
      const f32x2_t al2 = f32x2_t{alpha, alpha};
      lrun[qt] = lrun[qt] * al2 + ls;
      if (!same_max) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const f32x2_t lo = f32x2_t{oacc[dt][qt][0], oacc[dt][qt][1]} * al2;
          const f32x2_t hi = f32x2_t{oacc[dt][qt][2], oacc[dt][qt][3]} * al2;
          oacc[dt][qt] = f32x4_t{lo[0], lo[1], hi[0], hi[1]};
        }
      }
      // P^T fragments: k-step s2 = sub-tiles 2*s2 (elements 0..3) and 2*s2+1 (elements 4..7) = keys s2*32 + lg*8 ..+7
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint4 f;
        f.x = pack2<DT>(sacc[2 * s2][qt][0], sacc[2 * s2][qt][1]);
        f.y = pack2<DT>(sacc[2 * s2][qt][2], sacc[2 * s2][qt][3]);
        f.z = pack2<DT>(sacc[2 * s2 + 1][qt][0], sacc[2 * s2 + 1][qt][1]);
        f.w = pack2<DT>(sacc[2 * s2 + 1][qt][2], sacc[2 * s2 + 1][qt][3]);
        pf[qt][s2] = f;
      }
    }

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int vr = dt * 16 + lq;
      const char* vrow = sv + vr * VT_PITCH;
      const int vsw = (vr >> 1) & 7;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint4 vf = *(const uint4*)(vrow + (((lg + 4 * s2) ^ vsw) << 4));
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) oacc[dt][qt] = mfma_block<DT>(vf, pf[qt][s2], oacc[dt][qt]);
      }
    }

    if (t + 1 < ntiles) store_tile(s ^ 1);
    __syncthreads();
  };

  // interior tiles of this wave: all 64 keys below the limit of its first (= most restricted) query, and no bias
  int nfull = 0;
  if (p.bias == nullptr) {
    int jmin = klen;
    if (p.causal) jmin = min(jmin, q0 + p.causal_off + 1);
    if (p.chunk > 0) jmin = min(jmin, ((q0 + p.q_off) / p.chunk + 1) * p.chunk);
    nfull = min(max(jmin, 0) >> 6, ntiles);
  }
  int t = 0;
  for (; t < nfull; ++t) tile(std::false_type{}, t);
  for (; t < ntiles; ++t) tile(std::true_type{}, t);

  // ---- finalize: O[q][d] = O^T / l ; lane holds q = lq, d = dt*16 + 4*lg + r
  uint16_t* O = (uint16_t*)p.out + (int64_t)b * p.o_bs + h * 64;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float l = lrun[qt][0] + lrun[qt][1];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const int i = q0 + qt * 16 + lq;
    if (i >= p.Tq) continue;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 u;
      u.x = pack2<DT>(oacc[dt][qt][0] * inv, oacc[dt][qt][1] * inv);
      u.y = pack2<DT>(oacc[dt][qt][2] * inv, oacc[dt][qt][3] * inv);
      *(uint2*)(O + (int64_t)i * p.ldo + dt * 16 + 4 * lg) = u;
    }
  }
}

// ============================================================================================== 32x32x16 form
// The same algorithm on v_mfma_f32_32x32x16: a wave still owns 32 queries, but as ONE 32-wide column block — lanes l and l + 32 share
// query l & 31 and hold 16 + 16 of the 32 keys of a block.  Why: the kernel is ISSUE-bound (per 64-key tile a wave issues ~185 VALU +
// 32 MFMAs; a 16x16x32 MFMA blocks the SIMD's vector issue for 8 of its 16 cycles, a 32x32x16 for 8 of its 32 — MI355X_MICROARCH.md,
// cycle constants), so the same flops as 16 instead of 32 MFMAs free 128 issue cycles per tile-wave; the softmax bookkeeping (running
// max, alpha, rescale test) runs once per 32 queries instead of twice, and the row max needs one cross-lane step instead of two.
//   S^T block b (keys 32 b ..+31 of the tile) = K_b . Q^T: A = K rows, B = Q rows (k = d, 4 steps of 16);
//   D layout: column = query l & 31, register v <-> MFMA row (v & 3) + 8 (v >> 2) + 4 h (h = l >> 5).
// K rows are read PERMUTED (MFMA row m = key m with bits 2 and 3 swapped) so that registers 8 s .. 8 s + 7 of a lane are the 8
// CONSECUTIVE keys 16 s + 8 h ..+7: packed to 16 bits they are the B operand of O^T += V^T . P^T (k-step s of the block) and the V^T
// fragment is one 16-byte LDS read in natural key order ("An accumulator tile as the next MFMA's operand", cdna_hip_programming.md §3).
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int DT>
__device__ __forceinline__ f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
  if constexpr (DT == CV_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(bitcast<bf16x8_t>(a), bitcast<bf16x8_t>(b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8_t>(a), bitcast<f16x8_t>(b), c, 0, 0, 0);
}

template <int DT, int NWV, int ABL = 0>   // ABL: timing-only ablations (CV_ATTN_ABL; wrong results): 1 no K/V tile traffic after tile 0, 2 no softmax arithmetic
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 3 : 2) void attn32_kernel(const cv_attn_params p) {
  constexpr int NT = 64 * NWV, QW = 32 * NWV, NL = 512 / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // K image: 64 keys x 128 B, chunk slot XOR (row >> 1) & 7.  MFMA row lr of block b reads key row 32 b + kperm(lr).
  const int krow = (lr & 0x13) | ((lr & 4) << 1) | ((lr & 8) >> 1);   // bits 2 <-> 3
  const int b = blockIdx.z, h = blockIdx.x;
  const int hk = h / (p.H / p.Hkv);
  const int q_wg = blockIdx.y * QW;
  const int q0 = q_wg + wid * 32;

  const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + h * p.q_hs;
  const uint16_t* Kp = (const uint16_t*)p.k + (int64_t)b * p.k_bs + hk * p.k_hs;
  const uint16_t* Vt = (const uint16_t*)p.vt + (int64_t)(b * p.Hkv + hk) * 64 * p.vt_ld;
  const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;

  int limit = klen;
  const int q_max = min(p.Tq, q_wg + QW) - 1;
  if (p.causal) limit = min(limit, q_max + p.causal_off + 1);
  if (p.chunk > 0) limit = min(limit, ((q_max + p.q_off) / p.chunk + 1) * p.chunk);
  const int ntiles = (limit + 63) >> 6;

  // ---- Q fragments (B operand): lane = query lr, d chunk 2 ks + lh
  uint4 qf[4];
  {
    const int row = q0 + lr;
    const uint32_t msk = row < p.Tq ? 0xFFFFFFFFu : 0u;
    const int rc = min(row, p.Tq - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 v = *(const uint4*)(Q + (int64_t)rc * p.ldq + (2 * ks + lh) * 8);
      qf[ks] = make_uint4(v.x & msk, v.y & msk, v.z & msk, v.w & msk);
    }
  }

  u32x4_t rk[NL], rv[NL];
  int tile_j0 = 0;
  const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Kp, 0, klen > 0 ? (int)min(((int64_t)(klen - 1) * p.ldk + 64) * 2, (int64_t)0x7FFFFFFF) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Vt, 0, (int)min((int64_t)64 * p.vt_ld * 2, (int64_t)0x7FFFFFFF), 0x00020000);
  int koff[NL], voff[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int c = i * NT + tid;
    const int r = c >> 3, dc = c & 7;
    koff[i] = (r * p.ldk + dc * 8) * 2;
    voff[i] = (r * p.vt_ld + dc * 8) * 2;
  }
  auto load_tile = [&](int t) {
    const int j0 = t << 6;
    tile_j0 = j0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      rk[i] = __builtin_amdgcn_raw_buffer_load_b128(k_rs, koff[i], j0 * p.ldk * 2, 0);
      rv[i] = __builtin_amdgcn_raw_buffer_load_b128(v_rs, voff[i], j0 * 2, 0);
    }
  };
  auto store_tile = [&](int s) {
    char* sk = smem + s * STAGE_BYTES;
    char* sv = sk + KT_BYTES;
    const bool tail = tile_j0 + 64 > klen;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int c = i * NT + tid;
      const int r = c >> 3, dc = c & 7;
      *(u32x4_t*)(sk + (r << 7) + ((dc ^ ((r >> 1) & 7)) << 4)) = rk[i];
      u32x4_t u = rv[i];
      if (tail) {
        const int nvalid = klen - (tile_j0 + dc * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t m_lo = (2 * e < nvalid) ? 0x0000FFFFu : 0u;
          const uint32_t m_hi = (2 * e + 1 < nvalid) ? 0xFFFF0000u : 0u;
          u[e] &= (m_lo | m_hi);
        }
      }
      *(u32x4_t*)(sv + r * VT_PITCH + ((dc ^ ((r >> 1) & 7)) << 4)) = u;
    }
  };
  // fragment addresses: K block b, d step ks: row 32 b + krow, chunk 2 ks + lh; V^T block db, key chunk c: row 32 db + lr, chunk c
  int kaddr[2];
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) kaddr[bb] = ((32 * bb + krow) << 7);
  const int kswz = ((32 + krow) >> 1) & 7;   // (row >> 1) & 7 is the same for both blocks (32 >> 1 = 16 = 0 mod 8)
  int vaddr[2];
#pragma unroll
  for (int db = 0; db < 2; ++db) vaddr[db] = (32 * db + lr) * VT_PITCH;
  const int vswz = (lr >> 1) & 7;

  f32x16_t oacc[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int v = 0; v < 16; ++v) oacc[db][v] = 0.f;
  float mrun = NEG_BIG, lrun = 0.f;
  const float sc = p.scale * 1.4426950408889634f;

  if (ntiles > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();

  auto tile = [&](auto masked_tag, const int t) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int s = (ABL & 1) ? 0 : (t & 1);
    if (!(ABL & 1) && t + 1 < ntiles) load_tile(t + 1);
    const char* sk = smem + s * STAGE_BYTES;
    const char* sv = sk + KT_BYTES;
    const int j0 = t << 6;

    // ---- S^T = K . Q^T, two 32-key blocks
    f32x16_t sacc[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      f32x16_t a;
#pragma unroll
      for (int v = 0; v < 16; ++v) a[v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 kf = *(const uint4*)(sk + kaddr[bb] + (((2 * ks + lh) ^ kswz) << 4));
        a = mfma32<DT>(kf, qf[ks], a);
      }
      sacc[bb] = a;
    }

    // key index of register v in block bb: j0 + 32 bb + 16 (v >> 3) + 8 lh + (v & 7)
    float mnew;
    bool dead = false;
    if constexpr (MASKED) {
      const int i = q0 + lr;
      int jlim = klen;
      if (p.causal) jlim = min(jlim, i + p.causal_off + 1);
      if (p.chunk > 0) jlim = min(jlim, ((i + p.q_off) / p.chunk + 1) * p.chunk);
      if (p.bias != nullptr) {
        const float* brow = p.bias + (int64_t)b * p.bias_bs + (int64_t)h * p.bias_hs + (int64_t)min(i, p.Tq - 1) * p.bias_ld;
        // one batch of 16 loads per 32-key block (both blocks at once would hold 32 more registers: spills at 168)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          float bv[16];
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int j = j0 + 32 * bb + 16 * (v >> 3) + 8 * lh + (v & 7);
            bv[v] = brow[min(j, p.Tk - 1)];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int j = j0 + 32 * bb + 16 * (v >> 3) + 8 * lh + (v & 7);
            const float x = fmaf(sacc[bb][v], sc, bv[v] * 1.4426950408889634f);
            sacc[bb][v] = (j < jlim) ? x : NEG_BIG;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int j = j0 + 32 * bb + 16 * (v >> 3) + 8 * lh + (v & 7);
            sacc[bb][v] = (j < jlim) ? sacc[bb][v] * sc : NEG_BIG;
          }
      }
    }
    // 32 in-lane scores -> two v_max3 chains, then the other half of the query's keys from lane ^ 32
    float mx = fmaxf(fmaxf(sacc[0][0], sacc[0][1]), sacc[0][2]);
    float my = fmaxf(fmaxf(sacc[1][0], sacc[1][1]), sacc[1][2]);
#pragma unroll
    for (int v = 3; v + 1 < 16; v += 2) {
      mx = fmaxf(fmaxf(mx, sacc[0][v]), sacc[0][v + 1]);
      my = fmaxf(fmaxf(my, sacc[1][v]), sacc[1][v + 1]);
    }
    mx = fmaxf(fmaxf(mx, sacc[0][15]), fmaxf(my, sacc[1][15]));
    {
      const uint32_t u = __float_as_uint(mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float scl = MASKED ? 1.0f : sc;
    mnew = fmaxf(mrun, mx * scl);
    if constexpr (MASKED) dead = mnew <= 0.5f * NEG_BIG;
    const float nm = -mnew;
    float ls0 = 0.f, ls1 = 0.f;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int v = 0; v < 16; v += 2) {
        if (ABL & 2) continue;
        const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[bb][v], scl, nm));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[bb][v + 1], scl, nm));
        sacc[bb][v] = e0;
        sacc[bb][v + 1] = e1;
        ls0 += e0;
        ls1 += e1;
      }
    if constexpr (MASKED) {
      if (dead) {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int v = 0; v < 16; ++v) sacc[bb][v] = 0.f;
        ls0 = ls1 = 0.f;
      }
    }
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    const bool same_max = __all(mnew == mrun);
    mrun = mnew;
    lrun = fmaf(lrun, alpha, ls0 + ls1);
    if (!same_max) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int v = 0; v < 16; ++v) oacc[db][v] *= alpha;
    }
    // P^T fragments: block bb, k-step s2 = registers 8 s2 .. 8 s2 + 7 = keys 32 bb + 16 s2 + 8 lh ..+7
    uint4 pf[2][2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint4 f;
        f.x = pack2<DT>(sacc[bb][8 * s2 + 0], sacc[bb][8 * s2 + 1]);
        f.y = pack2<DT>(sacc[bb][8 * s2 + 2], sacc[bb][8 * s2 + 3]);
        f.z = pack2<DT>(sacc[bb][8 * s2 + 4], sacc[bb][8 * s2 + 5]);
        f.w = pack2<DT>(sacc[bb][8 * s2 + 6], sacc[bb][8 * s2 + 7]);
        pf[bb][s2] = f;
      }

    // ---- O^T += V^T . P^T: d block db (32 rows), key chunk c = 4 bb + 2 s2 + lh
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const uint4 vf = *(const uint4*)(sv + vaddr[db] + (((4 * bb + 2 * s2 + lh) ^ vswz) << 4));
          oacc[db] = mfma32<DT>(vf, pf[bb][s2], oacc[db]);
        }

    if (!(ABL & 1) && t + 1 < ntiles) store_tile(s ^ 1);
    __syncthreads();
  };

  int nfull = 0;
  if (p.bias == nullptr) {
    int jmin = klen;
    if (p.causal) jmin = min(jmin, q0 + p.causal_off + 1);
    if (p.chunk > 0) jmin = min(jmin, ((q0 + p.q_off) / p.chunk + 1) * p.chunk);
    nfull = min(max(jmin, 0) >> 6, ntiles);
  }
  int t = 0;
  for (; t < nfull; ++t) tile(std::false_type{}, t);
  for (; t < ntiles; ++t) tile(std::true_type{}, t);

  // ---- finalize: lane holds query lr, d = 32 db + 8 (v >> 2) + 4 lh + (v & 3)
  uint16_t* O = (uint16_t*)p.out + (int64_t)b * p.o_bs + h * 64;
  float l = lrun;
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const int i = q0 + lr;
  if (i < p.Tq) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 u;
        u.x = pack2<DT>(oacc[db][4 * g4 + 0] * inv, oacc[db][4 * g4 + 1] * inv);
        u.y = pack2<DT>(oacc[db][4 * g4 + 2] * inv, oacc[db][4 * g4 + 3] * inv);
        *(uint2*)(O + (int64_t)i * p.ldo + 32 * db + 8 * g4 + 4 * lh) = u;
      }
  }
}

// ============================================================================================== 64 queries per wave
// attn64_kernel: the 32x32x16 form with TWO 32-query blocks per wave (workgroup = 4 waves = 256 queries, 2 workgroups per CU at 256
// VGPRs) and the K / V^T tiles brought in by LDS-DMA (global_load_lds, 1 KiB per wave-instruction: no staging registers, none of the
// 13-cycle ds_write_b128 of the register-staged path).  Every K / V^T fragment read from LDS now feeds two MFMAs, and a tile is staged
// once per 256 queries instead of once per 128: the r03 ablations of the 128-query kernel put its tile traffic at 11 of 59 us and its
// MFMA + LDS-read skeleton at 38 us (3 x the MFMA floor).  No bias operand (the encoder's rel-pos attention keeps the 32-query forms).
// LDS-DMA writes lane l of a wave-instruction at base + 16 l, i.e. 8 rows x 8 chunk slots of the row-major image: the XOR swizzle
// goes on the SOURCE (lane (row, slot) fetches chunk slot ^ swz(row)).
template <int DT>
__global__ __launch_bounds__(256, 2) void attn64_kernel(const cv_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int krow = (lr & 0x13) | ((lr & 4) << 1) | ((lr & 8) >> 1);   // MFMA row -> key row: bits 2 <-> 3 (see attn32_kernel)
  const int b = blockIdx.z, h = blockIdx.x;
  const int hk = h / (p.H / p.Hkv);
  const int q_wg = blockIdx.y * 256;
  const int q0 = q_wg + wid * 64;

  const uint16_t* Q = (const uint16_t*)p.q + (int64_t)b * p.q_bs + h * p.q_hs;
  const uint16_t* Kp = (const uint16_t*)p.k + (int64_t)b * p.k_bs + hk * p.k_hs;
  const uint16_t* Vt = (const uint16_t*)p.vt + (int64_t)(b * p.Hkv + hk) * 64 * p.vt_ld;
  const int klen = p.klen ? min(p.klen[b], p.Tk) : p.Tk;

  int limit = klen;
  const int q_max = min(p.Tq, q_wg + 256) - 1;
  if (p.causal) limit = min(limit, q_max + p.causal_off + 1);
  if (p.chunk > 0) limit = min(limit, ((q_max + p.q_off) / p.chunk + 1) * p.chunk);
  const int ntiles = (limit + 63) >> 6;

  // ---- Q fragments (B operand): query block qb, d step ks: row q0 + 32 qb + lr, chunk 2 ks + lh
  uint4 qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = q0 + 32 * qb + lr;
    const uint32_t msk = row < p.Tq ? 0xFFFFFFFFu : 0u;
    const int rc = min(row, p.Tq - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 v = *(const uint4*)(Q + (int64_t)rc * p.ldq + (2 * ks + lh) * 8);
      qf[qb][ks] = make_uint4(v.x & msk, v.y & msk, v.z & msk, v.w & msk);
    }
  }

  // ---- LDS-DMA: wave w, instruction i (0, 1): the 1 KiB block 4 i + w of the K image and of the V^T image
  int drow[2], dchunk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    drow[i] = (4 * i + wid) * 8 + (lane >> 3);
    dchunk[i] = (lane & 7) ^ ((drow[i] >> 1) & 7);
  }
  auto dma_tile = [&](int t, int stage) {
    const int j0 = t << 6;
    char* sk = smem + stage * STAGE_BYTES;
    char* sv = sk + KT_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // rows / columns beyond the tensors are clamped to valid memory: their scores are masked (select, not multiply) and the
      // V^T columns beyond klen are zeroed in LDS before the tail tile is used
      const uint16_t* ks_ = Kp + (int64_t)min(j0 + drow[i], p.Tk - 1) * p.ldk + dchunk[i] * 8;
      const uint16_t* vs_ = Vt + (int64_t)drow[i] * p.vt_ld + min(j0 + dchunk[i] * 8, p.vt_ld - 8);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)ks_, (lds_ptr_t)(sk + (4 * i + wid) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)vs_, (lds_ptr_t)(sv + (4 * i + wid) * 1024), 16, 0, 0);
    }
  };
  // the tail tile (keys beyond klen inside it): zero those V^T columns (0 * garbage must not become NaN)
  auto fix_tail = [&](int t, int stage) {
    const int j0 = t << 6;
    char* sv = smem + stage * STAGE_BYTES + KT_BYTES;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = tid * 2 + e;
      const int d = c >> 3, slot = c & 7;
      const int kc = slot ^ ((d >> 1) & 7);
      const int nvalid = klen - (j0 + kc * 8);
      u32x4_t u = *(u32x4_t*)(sv + d * VT_PITCH + (slot << 4));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t m_lo = (2 * q < nvalid) ? 0x0000FFFFu : 0u;
        const uint32_t m_hi = (2 * q + 1 < nvalid) ? 0xFFFF0000u : 0u;
        u[q] &= (m_lo | m_hi);
      }
      *(u32x4_t*)(sv + d * VT_PITCH + (slot << 4)) = u;
    }
  };

  int kaddr[2], vaddr[2];
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
    kaddr[bb] = (32 * bb + krow) << 7;
    vaddr[bb] = (32 * bb + lr) * VT_PITCH;
  }
  const int kswz = (krow >> 1) & 7, vswz = (lr >> 1) & 7;

  f32x16_t oacc[2][2];   // [d block][query block]
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int v = 0; v < 16; ++v) oacc[db][qb][v] = 0.f;
  float mrun[2] = {NEG_BIG, NEG_BIG}, lrun[2] = {0.f, 0.f};
  const float sc = p.scale * 1.4426950408889634f;

  // 3-stage ring, two tiles in flight: the tile loop of the 2-stage form waited ~30 % of its wave cycles (SQ_WAIT_ANY) on the DMA it
  // had issued one tile earlier — a tile's arithmetic is shorter than a loaded L2 round trip
  if (ntiles > 0) dma_tile(0, 0);
  if (ntiles > 1) dma_tile(1, 1);
  if (ntiles > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile 0 landed (4 DMA instructions per wave and tile)
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ntiles == 1 && 64 > klen) { fix_tail(0, 0); __syncthreads(); }

  auto tile = [&](auto masked_tag, const int t) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int s = t % 3;
    if (t + 2 < ntiles) dma_tile(t + 2, (t + 2) % 3);   // that stage was last read during tile t - 1: every wave is past its barrier
    const char* sk = smem + s * STAGE_BYTES;
    const char* sv = sk + KT_BYTES;
    const int j0 = t << 6;

    // ---- software pipeline inside the tile (r03 counters: VALU-active 54 % and MFMA-busy 30 % of the kernel's time, adding up instead
    // of overlapping — both workgroups of a CU run the same phases in lockstep): the softmax of query block 0 is issued BETWEEN the
    // QK^T MFMAs of block 1, the softmax of block 1 between the PV MFMAs of block 0.  An MFMA occupies the vector issue port for 8 of
    // its 32 cycles and runs in the matrix pipe for the rest, so a wave's own VALU work fills the gaps.  sched_barrier(0) pins the
    // interleave the source spells out (hipcc otherwise clusters the MFMAs).
    f32x16_t sacc[2][2];   // [query block][key block]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int v = 0; v < 16; ++v) sacc[qb][bb][v] = 0.f;
    auto qk_step = [&](int qb, int i) {   // i = 4 bb + ks
      const int bb = i >> 2, ks = i & 3;
      const uint4 kf = *(const uint4*)(sk + kaddr[bb] + (((2 * ks + lh) ^ kswz) << 4));
      sacc[qb][bb] = mfma32<DT>(kf, qf[qb][ks], sacc[qb][bb]);
    };
    uint4 pf[2][2][2];   // [query block][key block][k-step]
    auto pv_step = [&](int qb, int i) {   // i = 4 db + 2 bb + s2
      const int db = i >> 2, bb = (i >> 1) & 1, s2 = i & 1;
      const uint4 vf = *(const uint4*)(sv + vaddr[db] + (((4 * bb + 2 * s2 + lh) ^ vswz) << 4));
      oacc[db][qb] = mfma32<DT>(vf, pf[qb][bb][s2], oacc[db][qb]);
    };
    // softmax of one query block in 8 slices
    float mx_[2], my_[2], mnew_[2], ls0_[2], ls1_[2], alpha_[2];
    bool same_[2], dead_[2];
    auto sm_step = [&](int qb, int i) {
      const float scl = MASKED ? 1.0f : sc;
      if (i == 0) {
        if constexpr (MASKED) {
          const int qi = q0 + 32 * qb + lr;
          int jlim = klen;
          if (p.causal) jlim = min(jlim, qi + p.causal_off + 1);
          if (p.chunk > 0) jlim = min(jlim, ((qi + p.q_off) / p.chunk + 1) * p.chunk);
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
              const int j = j0 + 32 * bb + 16 * (v >> 3) + 8 * lh + (v & 7);
              sacc[qb][bb][v] = (j < jlim) ? sacc[qb][bb][v] * sc : NEG_BIG;
            }
        }
        float mx = fmaxf(fmaxf(sacc[qb][0][0], sacc[qb][0][1]), sacc[qb][0][2]);
#pragma unroll
        for (int v = 3; v + 1 < 16; v += 2) mx = fmaxf(fmaxf(mx, sacc[qb][0][v]), sacc[qb][0][v + 1]);
        mx_[qb] = fmaxf(mx, sacc[qb][0][15]);
      } else if (i == 1) {
        float my = fmaxf(fmaxf(sacc[qb][1][0], sacc[qb][1][1]), sacc[qb][1][2]);
#pragma unroll
        for (int v = 3; v + 1 < 16; v += 2) my = fmaxf(fmaxf(my, sacc[qb][1][v]), sacc[qb][1][v + 1]);
        float mx = fmaxf(fmaxf(mx_[qb], my), sacc[qb][1][15]);
        const uint32_t u = __float_as_uint(mx);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        mnew_[qb] = fmaxf(mrun[qb], mx * scl);
        dead_[qb] = MASKED ? (mnew_[qb] <= 0.5f * NEG_BIG) : false;
        alpha_[qb] = __builtin_amdgcn_exp2f(mrun[qb] - mnew_[qb]);
        same_[qb] = __all(mnew_[qb] == mrun[qb]);
        mrun[qb] = mnew_[qb];
        ls0_[qb] = ls1_[qb] = 0.f;
      } else if (i < 6) {   // 8 scores per slice: registers 8 (i - 2) & 15 .. of key block (i - 2) >> 1
        const int bb = (i - 2) >> 1, v0 = ((i - 2) & 1) * 8;
        const float nm = -mnew_[qb];
#pragma unroll
        for (int v = v0; v < v0 + 8; v += 2) {
          const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[qb][bb][v], scl, nm));
          const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[qb][bb][v + 1], scl, nm));
          sacc[qb][bb][v] = e0;
          sacc[qb][bb][v + 1] = e1;
          ls0_[qb] += e0;
          ls1_[qb] += e1;
        }
        // the partial row sums are "redefined" here: MachineSink otherwise moves the whole slice (its results are only used after the
        // rescale branch of slice 7) below that branch, out of the MFMA gaps it was written into
        asm volatile("" : "+v"(ls0_[qb]), "+v"(ls1_[qb]));
      } else if (i == 6) {
        if constexpr (MASKED) {
          if (dead_[qb]) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
              for (int v = 0; v < 16; ++v) sacc[qb][bb][v] = 0.f;
            ls0_[qb] = ls1_[qb] = 0.f;
          }
        }
        lrun[qb] = fmaf(lrun[qb], alpha_[qb], ls0_[qb] + ls1_[qb]);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            uint4 f;
            f.x = pack2<DT>(sacc[qb][bb][8 * s2 + 0], sacc[qb][bb][8 * s2 + 1]);
            f.y = pack2<DT>(sacc[qb][bb][8 * s2 + 2], sacc[qb][bb][8 * s2 + 3]);
            f.z = pack2<DT>(sacc[qb][bb][8 * s2 + 4], sacc[qb][bb][8 * s2 + 5]);
            f.w = pack2<DT>(sacc[qb][bb][8 * s2 + 6], sacc[qb][bb][8 * s2 + 7]);
            pf[qb][bb][s2] = f;
          }
      } else {
        if (!same_[qb]) {
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int v = 0; v < 16; ++v) oacc[db][qb][v] *= alpha_[qb];
        }
      }
    };
    // phase A: S^T of query block 0
#pragma unroll
    for (int i = 0; i < 8; ++i) qk_step(0, i);
    __builtin_amdgcn_sched_barrier(0);
    // phase B: S^T of query block 1 under the softmax of block 0
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      qk_step(1, i);
      sm_step(0, i);
      __builtin_amdgcn_sched_barrier(0);
    }
    // phase C: O^T of query block 0 under the softmax of block 1
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      pv_step(0, i);
      sm_step(1, i);
      __builtin_amdgcn_sched_barrier(0);
    }
    // phase D: O^T of query block 1
#pragma unroll
    for (int i = 0; i < 8; ++i) pv_step(1, i);

    // this wave's share of tile t + 1 has landed (tile t + 2's four DMA instructions may still be in flight)
    if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 2 == ntiles && ((t + 1) << 6) + 64 > klen) {   // wave-uniform: the next tile is the tail tile
      fix_tail(t + 1, (t + 1) % 3);
      __syncthreads();
    }
  };

  int nfull = 0;
  {
    int jmin = klen;
    if (p.causal) jmin = min(jmin, q0 + p.causal_off + 1);
    if (p.chunk > 0) jmin = min(jmin, ((q0 + p.q_off) / p.chunk + 1) * p.chunk);
    nfull = min(max(jmin, 0) >> 6, ntiles);
  }
  int t = 0;
  for (; t < nfull; ++t) tile(std::false_type{}, t);
  for (; t < ntiles; ++t) tile(std::true_type{}, t);

  // ---- finalize: lane holds query 32 qb + lr, d = 32 db + 8 (v >> 2) + 4 lh + (v & 3)
  uint16_t* O = (uint16_t*)p.out + (int64_t)b * p.o_bs + h * 64;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l = lrun[qb];
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const int i = q0 + 32 * qb + lr;
    if (i >= p.Tq) continue;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        uint2 u;
        u.x = pack2<DT>(oacc[db][qb][4 * g4 + 0] * inv, oacc[db][qb][4 * g4 + 1] * inv);
        u.y = pack2<DT>(oacc[db][qb][4 * g4 + 2] * inv, oacc[db][qb][4 * g4 + 3] * inv);
        *(uint2*)(O + (int64_t)i * p.ldo + 32 * db + 8 * g4 + 4 * lh) = u;
      }
  }
}

}  // namespace

extern "C" int cv_attention(const cv_attn_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  cv_attn_params p = *pp;
  if (p.q_hs == 0) p.q_hs = 64;
  if (p.k_hs == 0) p.k_hs = 64;
  if ((p.q_hs & 7) || (p.k_hs & 7)) return CV_ERR_ARG;
  if (p.dtype != CV_BF16 && p.dtype != CV_F16) return CV_ERR_UNSUPPORTED;
  if (p.B <= 0 || p.H <= 0 || p.Hkv <= 0 || (p.H % p.Hkv) || p.Tq <= 0 || p.Tk <= 0 || p.q_off < 0) return CV_ERR_ARG;
  if (!p.q || !p.k || !p.vt || !p.out) return CV_ERR_ARG;
  if ((p.ldq & 7) || (p.ldk & 7) || (p.vt_ld & 7) || (p.ldo & 3) || (p.q_bs & 7) || (p.k_bs & 7) || (p.o_bs & 3)) return CV_ERR_ARG;
  if (p.vt_ld < p.Tk) return CV_ERR_ARG;
  if ((int64_t)p.Tk * p.ldk * 2 >= (int64_t)1 << 31 || (int64_t)64 * p.vt_ld * 2 >= (int64_t)1 << 31) return CV_ERR_UNSUPPORTED;   // 32-bit buffer offsets per (batch, head)
  if (((uintptr_t)p.q & 15) || ((uintptr_t)p.k & 15) || ((uintptr_t)p.vt & 15) || ((uintptr_t)p.out & 7)) return CV_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 2 * STAGE_BYTES;
  // 64-query workgroups when 128-query ones would not even fill the chip once (CV_ATTN_WAVES=2|4 overrides: tuning aid)
  const char* fe = getenv("CV_ATTN_WAVES");   // read per call: tests switch it
  const int forced = fe ? atoi(fe) : 0;
  const int64_t wgs4 = (int64_t)p.H * ((p.Tq + 127) / 128) * p.B;
  const bool two = forced == 2 || (forced != 4 && wgs4 < 256);
  // CV_ATTN_MFMA selects the alternative forms built and measured in round 3 (all tested against torch, tests/test_attention_gpu.py):
  //   16 (default) the 16x16x32 form above;
  //   32 the same tiling on 32x32x16 MFMAs (half the MFMA issue slots, one cross-lane step per row max): 58.2 vs 57.0 us;
  //   64 64 queries per wave, K / V^T by LDS-DMA into a 3-stage ring, the softmax of one query block issued between the MFMAs of the
  //      other (no bias operand): 57.0 - 58.2 us.
  // At the estimator's batch-8 shape (16 x 8 heads x T = 1000) all three take 57 - 58 us: the time is ~410 VALU + 32 MFMA issued per 64
  // queries x 64 keys (SQ_INSTS_VALU identical in every form) and ~30 % of wave cycles waiting, not the tile schedule.
  const char* me = getenv("CV_ATTN_MFMA");
  const int mf = me ? atoi(me) : 0;
  if (!p.bias && p.Tk >= 8 && p.vt_ld >= 8 && mf == 64) {
    dim3 grid(p.H, (p.Tq + 255) / 256, p.B);
    const size_t lds64 = 3 * STAGE_BYTES;
    if (p.dtype == CV_BF16) hipLaunchKernelGGL((attn64_kernel<CV_BF16>), grid, dim3(256), lds64, st, p);
    else hipLaunchKernelGGL((attn64_kernel<CV_F16>), grid, dim3(256), lds64, st, p);
    CV_CHECK_LAUNCH();
    return CV_OK;
  }
  const bool m32 = mf == 32;
  if (two) {
    dim3 grid(p.H, (p.Tq + 63) / 64, p.B);
    if (m32) {
      if (p.dtype == CV_BF16) hipLaunchKernelGGL((attn32_kernel<CV_BF16, 2>), grid, dim3(128), lds, st, p);
      else hipLaunchKernelGGL((attn32_kernel<CV_F16, 2>), grid, dim3(128), lds, st, p);
    } else {
      if (p.dtype == CV_BF16) hipLaunchKernelGGL((attn_kernel<CV_BF16, 2>), grid, dim3(128), lds, st, p);
      else hipLaunchKernelGGL((attn_kernel<CV_F16, 2>), grid, dim3(128), lds, st, p);
    }
  } else {
    dim3 grid(p.H, (p.Tq + 127) / 128, p.B);
    if (m32) {
      const char* ae = getenv("CV_ATTN_ABL");
      const int abl = ae ? atoi(ae) : 0;
      if (abl == 1 && p.dtype == CV_F16) hipLaunchKernelGGL((attn32_kernel<CV_F16, 4, 1>), grid, dim3(256), lds, st, p);
      else if (abl == 2 && p.dtype == CV_F16) hipLaunchKernelGGL((attn32_kernel<CV_F16, 4, 2>), grid, dim3(256), lds, st, p);
      else if (abl == 3 && p.dtype == CV_F16) hipLaunchKernelGGL((attn32_kernel<CV_F16, 4, 3>), grid, dim3(256), lds, st, p);
      else if (p.dtype == CV_BF16) hipLaunchKernelGGL((attn32_kernel<CV_BF16, 4>), grid, dim3(256), lds, st, p);
      else hipLaunchKernelGGL((attn32_kernel<CV_F16, 4>), grid, dim3(256), lds, st, p);
    } else {
      if (p.dtype == CV_BF16) hipLaunchKernelGGL((attn_kernel<CV_BF16, 4>), grid, dim3(256), lds, st, p);
      else hipLaunchKernelGGL((attn_kernel<CV_F16, 4>), grid, dim3(256), lds, st, p);
    }
  }
  CV_CHECK_LAUNCH();
  return CV_OK;
}
