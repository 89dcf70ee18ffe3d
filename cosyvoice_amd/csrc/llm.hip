// LLM decode-step kernels for gfx950: skinny MFMA GEMM over pre-packed weights (HBM-bound weight streaming),
// RMSNorm with split-K slab reduction, RoPE + KV append, single-query GQA attention, on-device RAS sampling.
#include "cv_device.h"
#include <algorithm>

namespace {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load16(const uint4* p) {
  const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// =========================================================================================== weight packing
// packed[(tile * nks + ks) * 64 + lane][j] = W[src_row(tile*16 + (lane&15))][ks*32 + 8*(lane>>4) + j]
__global__ __launch_bounds__(256) void pack_skinny_kernel(const uint16_t* W, uint16_t* Wp, int N, int K, int ntiles, int interleave) {
  const int nks = K >> 5;
  const int64_t total = (int64_t)ntiles * nks * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const int64_t t = i >> 6;
    const int ks = (int)(t % nks), tile = (int)(t / nks);
    int row = tile * 16 + (lane & 15);
    if (interleave) {
      // packed tile 2j = gate rows [16j,16j+16), tile 2j+1 = up rows N/2 + [16j,16j+16)
      const int j = tile >> 1;
      row = ((tile & 1) ? (N >> 1) : 0) + j * 16 + (lane & 15);
    }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N) v = *(const uint4*)(W + (int64_t)row * K + ks * 32 + 8 * (lane >> 4));
    *(uint4*)(Wp + i * 8) = v;
  }
}

// =========================================================================================== skinny GEMM
// One workgroup = TPW 16-column tiles x a K slice; the 4 waves split the slice, every wave streams its packed weight
// fragments (1 KiB per wave-load, non-temporal) straight into MFMA B operands and reduces through LDS at the end.
// NORM: the activation operand is produced in the prologue (residual + split-K slabs -> RMSNorm -> 16-bit, staged in
// LDS in fragment order); the weight loads of the first chunk are issued BEFORE the prologue so HBM latency hides under it.
// NORM: 0 = A from global memory, 1 + n = fused RMSNorm prologue that first adds n split-K slabs to the residual row.
// TPR = threads per activation row in the prologue (256/TPR rows): 64 for M <= 4, 32 for M <= 8, 16 otherwise.
// U = weight fragments a wave keeps in flight per tile (one batch of loads per U k-steps): 8 covers K <= 1024 in a
// single round trip, 16 does the same for the down projection's longer slices.
// Register diet for the gate/up form (TPW = 2): at 197 VGPRs two workgroups fit a CU, so on the ~100 CUs the pipeline
// leaves the decode loop its 304 workgroups ran as two rounds (11.2 us instead of 7.3).  Gamma is staged through LDS and
// read back just in time, the K = 896 slices use clamp-free immediate-offset loads (U = 7), and the kernel is bounded to
// 168 VGPRs = three workgroups per CU: one round on 104 CUs, 16 stragglers on 96.
// MR = 16-row groups of activation rows (M <= 16 * MR): every weight fragment a wave has loaded feeds MR MFMAs, so 32 rows (four
// 8-utterance batches in one token loop) cost ONE weight stream.  Plain-row forms only (NORM == 0: beyond 8 rows the input norm is
// its own launch); the split-norm partial sums are then pitched 16 * MR floats per part.
template <int DT, int TPW, int NORM, int TPR, int U, bool RS = false, int MR = 1>   // RS: consumer half of a split RMSNorm (rs_part)
__global__ __launch_bounds__(256, ((TPW == 2 && (TPR >= 32 || (MR == 2 && RS))) ? 3 : 2)) void skinny_kernel(const cv_skinny_params p) {
  static_assert(MR == 1 || NORM == 0, "row groups: plain activation rows only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float (*red)[TPW * MR][64][4] = (float (*)[TPW * MR][64][4])smem;   // [4][TPW * MR][64][4]
  char* aimg = smem + 4 * TPW * MR * 64 * 4 * sizeof(float);          // NORM: [nks][64 lanes][16 B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nks = p.K >> 5;
  const int per = (nks + p.ksplit - 1) / p.ksplit;
  const int kb = blockIdx.y * per, ke = min(nks, kb + per);
  const int cnt = max(ke - kb, 0), pw = (cnt + 3) >> 2;
  const int w0 = kb + wid * pw, w1 = min(ke, w0 + pw);
  const int tile0 = blockIdx.x * TPW;

  const uint4* Wt[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) Wt[t] = (const uint4*)p.Wp + (int64_t)(tile0 + t) * nks * 64 + lane;

  uint4 w[TPW][U];
  auto load_w = [&](int ks) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        // unconditional (clamped) load: a branch per element would serialise the stream with vmcnt(0) waits
        // U == 7: the host picked it because every wave's slice is exactly 7 k-steps (K = 896): no clamp, so the loads are
        // one base address + immediate offsets instead of 16 computed 64-bit addresses held in registers
        const int kk = (U == 7) ? ks + u : ((ks + u < w1) ? ks + u : max(w1 - 1, 0));
        w[t][u] = nt_load16(Wt[t] + (int64_t)kk * 64);
      }
  };

  // mode 1 (out += A W^T): the residual values this lane will update are fetched now, with the weights, instead of
  // after the reduction (one dependent HBM round trip less on the kernel's critical path)
  float resid[MR][TPW][4];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) resid[mr][t][r] = 0.f;
  if (p.mode == 1 && wid == 0) {
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      const int m = min((lane & 15) + 16 * mr, p.M - 1);
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          resid[mr][t][r] = p.out_f32[(int64_t)m * p.ldo + min((tile0 + t) * 16 + 4 * (lane >> 4) + r, p.N - 1)];
    }
  }
  // consumer half of the split RMSNorm: the producer's per-workgroup partial sums of squares of this lane's row, fetched with
  // the weights and summed in fixed order; 1/rms is applied in the epilogue
  // (lane (m, g) takes partial rows g, g + 4, ...: one batch of 16 independent loads issued ahead of the weight stream and only
  // consumed after the MFMAs; n_rs_part <= 64)
  float rs_p[MR][RS ? 16 : 1];
  if constexpr (RS) {
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int j = 0; j < 16; ++j) rs_p[mr][j] = 0.f;
    if (wid == 0) {
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) {
        const int m = min((lane & 15) + 16 * mr, p.M - 1), g4 = lane >> 4;
#pragma unroll
        for (int j = 0; j < 16; ++j) rs_p[mr][j] = p.rs_part[min(g4 + 4 * j, p.n_rs_part - 1) * (16 * MR) + m];
      }
    }
  }

  if constexpr (NORM != 0) {
    // TPR threads per row; thread handles float4 columns c = sub + TPR i.  Every load is unconditional (rows >= M and
    // columns beyond K are clamped and masked arithmetically) so the whole prologue is one batch of loads + one wait.
    constexpr int NV = 256 / TPR;  // float4 per thread for K <= 1024
    const int row = tid / TPR, sub = tid % TPR;
    const int nq = p.K >> 2;  // float4 per row
    const bool live = row < p.M;
    const int rowc = live ? row : 0;
    const float* xr = p.nx + (int64_t)rowc * p.ldnx;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *(const float4*)(xr + min(sub + TPR * i, nq - 1) * 4);
    if constexpr (NORM > 1) {
      const float* sl = p.nslabs + (int64_t)rowc * p.ld_nslab;
#pragma unroll
      for (int s2 = 0; s2 < NORM - 1; ++s2) {
        const float* sp = sl + (int64_t)s2 * p.nslab_stride;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const float4 q = *(const float4*)(sp + min(sub + TPR * i, nq - 1) * 4);
          v[i].x += q.x; v[i].y += q.y; v[i].z += q.z; v[i].w += q.w;
        }
      }
    }
    constexpr bool GAMMA_LDS = TPW == 2;  // the occupancy-bound form; the others keep gamma in registers (one barrier less)
    float4 gm[GAMMA_LDS ? 1 : NV];
    if constexpr (GAMMA_LDS) {
      float* sgam = (float*)smem;  // the reduction scratch is free until the MFMAs are done (K <= 1024 floats fit)
      const float4 g4 = *(const float4*)(p.ngamma + min(tid, nq - 1) * 4);
      if (tid < nq) *(float4*)(sgam + tid * 4) = g4;
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) gm[i] = *(const float4*)(p.ngamma + min(sub + TPR * i, nq - 1) * 4);
    }
    load_w(w0);  // weight stream issued behind the (L2-resident) activation loads; it lands during the norm arithmetic
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float msk = (sub + TPR * i < nq && live) ? 1.f : 0.f;
      ss += msk * (v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w);
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = rsqrtf(ss / (float)p.K + p.neps);
    const bool writer = p.nx_out && blockIdx.x == 0 && blockIdx.y == 0 && live;
    if constexpr (GAMMA_LDS) __syncthreads();  // gamma staged
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = sub + TPR * i;
      if (c >= nq) continue;
      float4 gmi;
      if constexpr (GAMMA_LDS) gmi = *(const float4*)((const float*)smem + c * 4);
      else gmi = gm[i];
      uint2 u = make_uint2(0, 0);
      if (live) {
        u.x = pack2<DT>(v[i].x * rstd * gmi.x, v[i].y * rstd * gmi.y);
        u.y = pack2<DT>(v[i].z * rstd * gmi.z, v[i].w * rstd * gmi.w);
      }
      if (writer) *(float4*)(p.nx_out + (int64_t)row * p.ldnx + c * 4) = v[i];
      // element k0 = 4c: chunk kc = c >> 1 (8 elements), half = c & 1; fragment slot (ks = kc >> 2, g = kc & 3, row)
      const int kc = c >> 1;
      *(uint2*)(aimg + ((((kc >> 2) * 64) + (kc & 3) * 16 + row) << 4) + (c & 1) * 8) = u;
    }
    if constexpr (TPR > 16) {
      // rows 256/TPR .. 15 of the fragment image are never written by the loop above: zero them (they feed MFMA lanes
      // whose results are discarded, but must be finite)
      constexpr int ROWS = 256 / TPR;
      const int nks_ = p.K >> 5;
      for (int idx = tid; idx < nks_ * 4 * (16 - ROWS); idx += 256) {
        const int r_ = ROWS + idx % (16 - ROWS), kg = idx / (16 - ROWS);  // kg = ks*4 + g
        *(uint4*)(aimg + (((kg >> 2) * 64 + (kg & 3) * 16 + r_) << 4)) = make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();
  } else {
    load_w(w0);
  }
  // the partial sums were issued ahead of the weights and return ahead of them (loads return in order): fold them now, under the
  // weight stream's latency, instead of carrying 16 registers per row group through the MFMA loop
  float rs_sum[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) rs_sum[mr] = 0.f;
  if constexpr (RS) {
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int j = 0; j < 16; ++j) rs_sum[mr] += ((lane >> 4) + 4 * j < p.n_rs_part) ? rs_p[mr][j] : 0.f;
  }

  // rows >= M are not fetched (their lanes re-read row M-1, the same cache lines, and are masked to zero): for M = 8 that
  // halves the activation traffic, which per workgroup is as large as the weight slice itself
  const uint16_t* Arow[MR];
  uint32_t rowmask[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    Arow[mr] = (NORM != 0) ? nullptr : (const uint16_t*)p.A + (int64_t)min((lane & 15) + 16 * mr, p.M - 1) * p.lda + 8 * (lane >> 4);
    rowmask[mr] = (lane & 15) + 16 * mr < p.M ? 0xFFFFFFFFu : 0u;
  }
  f32x4_t acc[TPW][MR];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[t][mr] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // SEQ_A (the gate/up form with two row groups): the second group's activation fragments are fetched only after the first group's
  // MFMAs, into the same registers — one more L2 round trip per workgroup, but 140 instead of 196 VGPRs: three workgroups per CU, and the
  // 304 workgroups of the launch take two rounds instead of three on the decode loops' 64 CUs.
  constexpr bool SEQ_A = MR == 2 && TPW == 2 && RS;
  for (int ks = w0; ks < w1; ks += U) {
    if (ks != w0) load_w(ks);
    auto load_a = [&](uint4 (&a)[U], int mr) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // unconditional clamped load, masked to zero beyond the slice (the clamped weight fragment then contributes 0)
        const bool ok = (U == 7) || (ks + u < w1);
        const int kk = ok ? ks + u : max(w1 - 1, 0);
        uint4 t;
        if constexpr (NORM != 0) t = *(const uint4*)(aimg + ((kk * 64 + lane) << 4));
        else t = *(const uint4*)(Arow[mr] + kk * 32);
        const uint32_t msk = ok ? ((NORM != 0) ? 0xFFFFFFFFu : rowmask[mr]) : 0u;
        a[u] = make_uint4(t.x & msk, t.y & msk, t.z & msk, t.w & msk);
      }
    };
    if constexpr (SEQ_A) {
      uint4 a[U];
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) {
        load_a(a, mr);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int t = 0; t < TPW; ++t) acc[t][mr] = mfma_block<DT>(w[t][u], a[u], acc[t][mr]);
        __builtin_amdgcn_sched_barrier(0);   // keep the second group's loads behind the first group's MFMAs (register reuse)
      }
    } else {
      uint4 a[MR][U];
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) load_a(a[mr], mr);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) acc[t][mr] = mfma_block<DT>(w[t][u], a[mr][u], acc[t][mr]);
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      red[wid][t * MR + mr][lane][0] = acc[t][mr][0]; red[wid][t * MR + mr][lane][1] = acc[t][mr][1];
      red[wid][t * MR + mr][lane][2] = acc[t][mr][2]; red[wid][t * MR + mr][lane][3] = acc[t][mr][3];
    }
  __syncthreads();
  if (wid != 0) return;
  const int g = lane >> 4;
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
  float v[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[t][r] = red[0][t * MR + mr][lane][r] + red[1][t * MR + mr][lane][r] + red[2][t * MR + mr][lane][r] + red[3][t * MR + mr][lane][r];

  const int m = (lane & 15) + 16 * mr;
  const bool live = m < p.M;   // (dead rows run the arithmetic on zeros and store nothing: the row sums below need every lane)
  if (p.mode == 2) {
    if constexpr (TPW >= 2) {   // TPW / 2 (gate, up) tile pairs
      float rstd = 1.f;
      if constexpr (RS) {
        float rs_t = rs_sum[mr];
        rs_t += __shfl_xor(rs_t, 16, 64);
        rs_t += __shfl_xor(rs_t, 32, 64);
        rstd = rsqrtf(rs_t / (float)p.K + p.rs_eps);
      }
#pragma unroll
      for (int pr = 0; pr < TPW / 2; ++pr) {
        const int nb = (tile0 + 2 * pr) * 16 + 4 * g;          // gate tile columns in packed order
        const int hcol = ((tile0 + 2 * pr) >> 1) * 16 + 4 * g;  // output column
        float h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float gt = v[2 * pr][r] * rstd, up = v[2 * pr + 1][r] * rstd;
          if (p.bias) { gt += p.bias[nb + r]; up += p.bias[nb + 16 + r]; }
          h[r] = act_silu(gt) * up;
        }
        uint2 u;
        u.x = pack2<DT>(h[0], h[1]);
        u.y = pack2<DT>(h[2], h[3]);
        if (live) *(uint2*)((uint16_t*)p.out_act + (int64_t)m * p.ldoa + hcol) = u;
      }
    }
    continue;
  }
  float ssq = 0.f;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int nb = (tile0 + t) * 16 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nb + r;
      if (n >= p.N || !live) continue;
      float o = v[t][r];
      if (p.bias && blockIdx.y == 0) o += p.bias[n];
      if (p.mode == 1) {
        const float nx = resid[mr][t][r] + o;
        p.out_f32[(int64_t)m * p.ldo + n] = nx;
        if (p.xb_out) {   // producer half of the split RMSNorm
          ((uint16_t*)p.xb_out)[(int64_t)m * p.ldxb + n] = Elem16<DT>::from_f32(nx);
          ssq += nx * nx;
        }
      }
      else if (p.mode == 3) ((uint16_t*)p.out_act)[(int64_t)m * p.ldoa + n] = Elem16<DT>::from_f32(fmaxf(o, 0.f));
      else p.out_f32[(int64_t)blockIdx.y * p.slab_stride + (int64_t)m * p.ldo + n] = o;
    }
  }
  if (p.mode == 1 && p.ss_part) {   // lanes m, m + 16, m + 32, m + 48 hold the four column groups of row m (all active together)
    ssq += __shfl_xor(ssq, 16, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    if (g == 0 && live) p.ss_part[blockIdx.x * (16 * MR) + m] = ssq;
  }
  }   // mr
}

// =========================================================================================== streaming skinny GEMM
// The decode step's form of the kernel above (every wave's K slice fits one batch of U fragment loads): a workgroup walks
// tile groups g = blockIdx.x, += gridDim.x with the NEXT group's weight fragments already in flight (two register sets,
// loads never conditional), so a CU keeps an HBM stream going instead of
// paying one cold round trip per workgroup — what matters when the decode loop owns 64-96 CUs (tts_batches' CU partition:
// 304 single-shot workgroups on 96 CUs ran as two full rounds, 11.2 us; see tools/llm_kernel_bench.py) and harmless with
// one group per workgroup.  The norm prologue and the activation fragments are produced once per workgroup.
// LOOP = false: one group per workgroup (grid = groups): no second register set, no dummy loads.
template <int DT, int TPW, int NORM, int TPR, int U, bool LOOP>
__global__ __launch_bounds__(256) void skinny_stream_kernel(const cv_skinny_params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 (*red)[4][TPW][64] = (float4 (*)[4][TPW][64])smem;            // [LOOP ? 2 : 1][4][TPW][64]
  char* aimg = smem + (LOOP ? 2 : 1) * 4 * TPW * 64 * sizeof(float4);   // NORM: [nks][64 lanes][16 B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nks = p.K >> 5;
  const int per = (nks + p.ksplit - 1) / p.ksplit;
  const int kb = blockIdx.y * per, ke = min(nks, kb + per);
  const int cnt = max(ke - kb, 0), pw = (cnt + 3) >> 2;
  const int w0 = kb + wid * pw, w1 = min(ke, w0 + pw);   // w1 - w0 <= U (host-checked)
  const int ngroups = ((p.N + 15) / 16) / TPW;
  const int stride = gridDim.x;
  const uint4* Wbase = (const uint4*)p.Wp + lane;

  auto load_group = [&](uint4 (&w)[TPW][U], float (&rs)[TPW][4], int g) {
    // never conditional, never a shared dummy line (2432 waves hitting one KiB serialise on its channel): a k-step past
    // the slice re-reads the wave's last fragment, a group past the end re-reads the group being computed (L2 hits)
    const int gs = g < ngroups ? g : g - stride;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int kk = (w0 + u < w1) ? w0 + u : max(w1 - 1, 0);
        w[t][u] = nt_load16(Wbase + ((int64_t)(gs * TPW + t) * nks + kk) * 64);
      }
    if (p.mode == 1) {  // kernel-uniform: the residual values of group g, fetched with its weights
      const int gc = min(g, ngroups - 1), m = min(lane & 15, p.M - 1);
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          rs[t][r] = p.out_f32[(int64_t)m * p.ldo + min((gc * TPW + t) * 16 + 4 * (lane >> 4) + r, p.N - 1)];
    }
  };

  uint4 wA[TPW][U], wB[TPW][U];
  float rA[TPW][4], rB[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) rA[t][r] = rB[t][r] = 0.f;

  if constexpr (NORM != 0) {
    constexpr int NV = 256 / TPR;  // float4 per thread for K <= 1024
    const int row = tid / TPR, sub = tid % TPR;
    const int nq = p.K >> 2;
    const bool live = row < p.M;
    const int rowc = live ? row : 0;
    const float* xr = p.nx + (int64_t)rowc * p.ldnx;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *(const float4*)(xr + min(sub + TPR * i, nq - 1) * 4);
    if constexpr (NORM > 1) {
      const float* sl = p.nslabs + (int64_t)rowc * p.ld_nslab;
#pragma unroll
      for (int s2 = 0; s2 < NORM - 1; ++s2) {
        const float* sp = sl + (int64_t)s2 * p.nslab_stride;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const float4 q = *(const float4*)(sp + min(sub + TPR * i, nq - 1) * 4);
          v[i].x += q.x; v[i].y += q.y; v[i].z += q.z; v[i].w += q.w;
        }
      }
    }
    float4 gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) gm[i] = *(const float4*)(p.ngamma + min(sub + TPR * i, nq - 1) * 4);
    load_group(wA, rA, blockIdx.x);  // behind the (L2-resident) activation loads; lands during the norm arithmetic
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float msk = (sub + TPR * i < nq && live) ? 1.f : 0.f;
      ss += msk * (v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w);
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = rsqrtf(ss / (float)p.K + p.neps);
    const bool writer = p.nx_out && blockIdx.x == 0 && blockIdx.y == 0 && live;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = sub + TPR * i;
      if (c >= nq) continue;
      uint2 u = make_uint2(0, 0);
      if (live) {
        u.x = pack2<DT>(v[i].x * rstd * gm[i].x, v[i].y * rstd * gm[i].y);
        u.y = pack2<DT>(v[i].z * rstd * gm[i].z, v[i].w * rstd * gm[i].w);
      }
      if (writer) *(float4*)(p.nx_out + (int64_t)row * p.ldnx + c * 4) = v[i];
      const int kc = c >> 1;
      *(uint2*)(aimg + ((((kc >> 2) * 64) + (kc & 3) * 16 + row) << 4) + (c & 1) * 8) = u;
    }
    if constexpr (TPR > 16) {
      constexpr int ROWS = 256 / TPR;
      for (int idx = tid; idx < nks * 4 * (16 - ROWS); idx += 256) {
        const int r_ = ROWS + idx % (16 - ROWS), kg = idx / (16 - ROWS);
        *(uint4*)(aimg + (((kg >> 2) * 64 + (kg & 3) * 16 + r_) << 4)) = make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();
  } else {
    load_group(wA, rA, blockIdx.x);
  }

  // activation fragments of this wave's K slice: the same for every group, read once
  uint4 a[U];
  {
    const uint16_t* Arow = (NORM != 0) ? nullptr : (const uint16_t*)p.A + (int64_t)min(lane & 15, p.M - 1) * p.lda + 8 * (lane >> 4);
    const uint32_t rowmask = ((NORM != 0) || (lane & 15) < p.M) ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = w0 + u < w1;
      const int kk = ok ? w0 + u : max(w1 - 1, 0);
      uint4 t;
      if constexpr (NORM != 0) t = *(const uint4*)(aimg + ((min(kk, nks - 1) * 64 + lane) << 4));
      else t = *(const uint4*)(Arow + min(kk, nks - 1) * 32);
      const uint32_t msk = ok ? rowmask : 0u;
      a[u] = make_uint4(t.x & msk, t.y & msk, t.z & msk, t.w & msk);
    }
  }

  int it = 0;
  auto compute = [&](uint4 (&w)[TPW][U], float (&rs)[TPW][4], int g) {
    f32x4_t acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = mfma_block<DT>(w[t][u], a[u], acc[t]);
    const int buf = it & 1;
#pragma unroll
    for (int t = 0; t < TPW; ++t) red[buf][wid][t][lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    if (wid == (it & 3)) {  // the epilogue rotates over the waves; the others run ahead into the next group
      float v[TPW][4];
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const float4 q0 = red[buf][0][t][lane], q1 = red[buf][1][t][lane], q2 = red[buf][2][t][lane], q3 = red[buf][3][t][lane];
        v[t][0] = q0.x + q1.x + q2.x + q3.x; v[t][1] = q0.y + q1.y + q2.y + q3.y;
        v[t][2] = q0.z + q1.z + q2.z + q3.z; v[t][3] = q0.w + q1.w + q2.w + q3.w;
      }
      const int m = lane & 15, gq = lane >> 4;
      const int tile0 = g * TPW;
      if (m < p.M) {
        if (p.mode == 2) {
          if constexpr (TPW == 2) {
            const int nb = tile0 * 16 + 4 * gq;
            const int hcol = (tile0 >> 1) * 16 + 4 * gq;
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float gt = v[0][r], up = v[1][r];
              if (p.bias) { gt += p.bias[nb + r]; up += p.bias[nb + 16 + r]; }
              h[r] = act_silu(gt) * up;
            }
            uint2 u;
            u.x = pack2<DT>(h[0], h[1]);
            u.y = pack2<DT>(h[2], h[3]);
            *(uint2*)((uint16_t*)p.out_act + (int64_t)m * p.ldoa + hcol) = u;
          }
        } else {
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const int nb = (tile0 + t) * 16 + 4 * gq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int n = nb + r;
              if (n >= p.N) continue;
              float o = v[t][r];
              if (p.bias && blockIdx.y == 0) o += p.bias[n];
              if (p.mode == 1) p.out_f32[(int64_t)m * p.ldo + n] = rs[t][r] + o;
              else p.out_f32[(int64_t)blockIdx.y * p.slab_stride + (int64_t)m * p.ldo + n] = o;
            }
          }
        }
      }
    }
    ++it;
  };

  if constexpr (!LOOP) {
    compute(wA, rA, blockIdx.x);
    return;
  }
  for (int g = blockIdx.x; g < ngroups;) {
    load_group(wB, rB, g + stride);
    compute(wA, rA, g);
    g += stride;
    if (g >= ngroups) break;
    load_group(wA, rA, g + stride);
    compute(wB, rB, g);
    g += stride;
  }
}

// =========================================================================================== rmsnorm + slab reduce
template <int DT>
__global__ __launch_bounds__(256) void rmsnorm_reduce_kernel(float* x, int ldx, const float* slabs, int nslab, int64_t slab_stride,
                                                             int ld_slab, const float* gamma, float eps, void* xn, int ldxn, int dim) {
  __shared__ float wsum[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  float* xr = x + (int64_t)row * ldx;
  const int nv = dim >> 2;
  float4 v[4];  // dim <= 4096
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    const int cc = min(c, nv - 1) * 4;  // unconditional clamped loads
    v[i] = *(const float4*)(xr + cc);
    for (int s = 0; s < nslab; ++s) {
      const float4 q = *(const float4*)(slabs + s * slab_stride + (int64_t)row * ld_slab + cc);
      v[i].x += q.x; v[i].y += q.y; v[i].z += q.z; v[i].w += q.w;
    }
    if (c < nv) {
      if (nslab > 0) *(float4*)(xr + c * 4) = v[i];
      ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  const float tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  const float rstd = rsqrtf(tot / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * 256 + tid;
    if (c >= nv) continue;
    const float4 gm = *(const float4*)(gamma + c * 4);
    const float o0 = v[i].x * rstd * gm.x, o1 = v[i].y * rstd * gm.y, o2 = v[i].z * rstd * gm.z, o3 = v[i].w * rstd * gm.w;
    if constexpr (DT == CV_F32) {
      *(float4*)((float*)xn + (int64_t)row * ldxn + c * 4) = make_float4(o0, o1, o2, o3);
    } else {
      uint2 u;
      u.x = pack2<DT>(o0, o1);
      u.y = pack2<DT>(o2, o3);
      *(uint2*)((uint16_t*)xn + (int64_t)row * ldxn + c * 4) = u;
    }
  }
}

// =========================================================================================== RoPE + KV append
template <int DT>
__global__ __launch_bounds__(256) void rope_append_kernel(const float* qkv, int ldqkv, const int32_t* pos_base, int rows_per_seq,
                                                          int Hq, int Hkv, const float* inv_freq, uint16_t* q_out, int ldq,
                                                          uint16_t* kcache, uint16_t* vtcache, int ctx_max) {
  const int r = blockIdx.x;
  const int b = r / rows_per_seq;
  const int pos = pos_base[b] + (r - b * rows_per_seq);
  const float* row = qkv + (int64_t)r * ldqkv;
  const int nq = Hq * 32, nk = Hkv * 32, nvv = Hkv * 64;
  for (int i = threadIdx.x; i < nq + nk + nvv; i += 256) {
    if (i < nq + nk) {
      const bool isq = i < nq;
      const int j = isq ? i : i - nq;
      const int h = j >> 5, d = j & 31;
      const float* src = row + (isq ? 0 : Hq * 64) + h * 64;
      const float x1 = src[d], x2 = src[d + 32];
      float sn, cs;
      sincosf((float)pos * inv_freq[d], &sn, &cs);
      const float o1 = x1 * cs - x2 * sn, o2 = x2 * cs + x1 * sn;
      if (isq) {
        q_out[(int64_t)r * ldq + h * 64 + d] = Elem16<DT>::from_f32(o1);
        q_out[(int64_t)r * ldq + h * 64 + d + 32] = Elem16<DT>::from_f32(o2);
      } else if (pos < ctx_max) {
        uint16_t* kd = kcache + (((int64_t)b * Hkv + h) * ctx_max + pos) * 64;
        kd[d] = Elem16<DT>::from_f32(o1);
        kd[d + 32] = Elem16<DT>::from_f32(o2);
      }
    } else if (pos < ctx_max) {
      const int j = i - nq - nk;
      const int h = j >> 6, d = j & 63;
      vtcache[(((int64_t)b * Hkv + h) * 64 + d) * ctx_max + pos] = Elem16<DT>::from_f32(row[(Hq + Hkv) * 64 + j]);
    }
  }
}

// =========================================================================================== fragment-tiled KV cache
// The fused decode attention keeps each (sequence, kv head) cache as ctx_max/64 tiles of 8 KB in MFMA FRAGMENT ORDER, so a
// wave's load of one fragment is 64 lanes x 16 contiguous bytes (8 full cache lines per instruction).  With the row-major
// caches a fragment load touched 16 lines (64 B of each for K, 32 B for V^T: 384 line accesses per 64-key tile instead of
// 128) and the one CU behind a (sequence, head) workgroup spent 5.4 of the kernel's 9 us getting 100 KB through its L1.
//   K tile:  element ((kt*2 + f)*64 + lq + 16*lg)*8 + e  =  key 16*kt + lq,             d = 32*f + 8*lg + e
//   V tile:  element ((dt*2 + s2)*64 + lq + 16*lg)*8 + e =  key 32*s2 + 16*(e>>2) + 4*lg + (e&3),   d = 16*dt + lq
constexpr int KV_TILE = 64 * 64;   // elements per tile
__device__ __forceinline__ int ktile_index(int key_rel, int d) {
  return ((((key_rel >> 4) * 2 + (d >> 5)) * 64 + (key_rel & 15) + 16 * ((d >> 3) & 3)) << 3) + (d & 7);
}
__device__ __forceinline__ int vtile_index(int key_rel, int d) {
  const int r = key_rel & 31;
  return ((((d >> 4) * 2 + (key_rel >> 5)) * 64 + (d & 15) + 16 * ((r & 15) >> 2)) << 3) + (r & 3) + 4 * (r >> 4);
}

// row-major caches ([B][Hkv][ctx_max][64] keys x d, [B][Hkv][64][ctx_max] d x keys: what cv_rope_append writes and
// cv_attention reads during prefill) -> fragment-tiled caches, tiles [0, ntiles)
__global__ __launch_bounds__(256) void kv_retile_kernel(const uint16_t* k_rm, const uint16_t* vt_rm, uint16_t* k_t, uint16_t* vt_t, int ctx_max) {
  const int t = blockIdx.x, bh = blockIdx.y;
  const uint16_t* ks = k_rm + (int64_t)bh * ctx_max * 64 + (int64_t)t * 64 * 64;
  const uint16_t* vs = vt_rm + (int64_t)bh * 64 * ctx_max + t * 64;
  uint16_t* kd = k_t + ((int64_t)bh * (ctx_max >> 6) + t) * KV_TILE;
  uint16_t* vd = vt_t + ((int64_t)bh * (ctx_max >> 6) + t) * KV_TILE;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + i * 256;          // 16-byte chunk of the tile: (fragment index, lane)
    const int fr = c >> 6, lq = c & 15, lg = (c >> 4) & 3;
    // K fragment fr = kt*2 + f: key 16*kt + lq, d = 32*f + 8*lg .. +7 (contiguous in the row-major row)
    *(uint4*)(kd + c * 8) = *(const uint4*)(ks + ((fr >> 1) * 16 + lq) * 64 + 32 * (fr & 1) + 8 * lg);
    // V fragment fr = dt*2 + s2: d = 16*dt + lq, keys 32*s2 + 4*lg .. +3 and + 16
    const uint16_t* vr = vs + (int64_t)((fr >> 1) * 16 + lq) * ctx_max + 32 * (fr & 1) + 4 * lg;
    const uint2 a = *(const uint2*)vr, b2 = *(const uint2*)(vr + 16);
    *(uint4*)(vd + c * 8) = make_uint4(a.x, a.y, b2.x, b2.y);
  }
}

// =========================================================================================== decode attention
constexpr float NEG_BIG = -1e30f;
constexpr int DA_WAVES = 8;

// One workgroup per (kv head, sequence); every wave owns key tiles wid, wid + 8 (64 keys each) and issues the loads of BOTH
// at kernel entry, together with the qkv row — nothing on the K/V stream waits for the new token: its roped K row and V
// column are exchanged through LDS and patched into the fragments of the wave whose tile holds position `pos`, while the
// global append happens on the side for later steps.  (The first version stored the new K/V, fenced and read them back,
// and loaded a wave's second tile only after finishing its first: two more dependent HBM round trips, 8.5 us per call.)
// FUSED is a template parameter, not `qkv != nullptr` at run time: a run-time branch around the qkv loads ends in PHIs
// with the zero defaults, and hipcc then waits for those loads before it issues the K/V tile loads.
template <int DT, bool FUSED, int ABL = 0>   // ABL: timing-only ablations (CV_DA_ABL; results are wrong): 1 no MFMA/softmax, 2 no K/V loads
__global__ __launch_bounds__(512) void decode_attn_kernel(const uint16_t* q, int ldq, uint16_t* kcache, uint16_t* vtcache,
                                                          const int32_t* ctx_len, int ctx_add, uint16_t* out, int ldo, int Hq, int Hkv,
                                                          int ctx_max, float scale, const float* qkv, int ldqkv, const float* inv_freq) {
  __shared__ float s_m[DA_WAVES][16], s_l[DA_WAVES][16];
  __shared__ float s_o[DA_WAVES][64][17];  // [wave][d][query col] (+1 pad)
  __shared__ __attribute__((aligned(16))) uint16_t s_newk[64], s_newv[64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int hk = blockIdx.x, b = blockIdx.y;
  const int G = Hq / Hkv;  // query heads per kv head (<= 16)
  // FUSED: fragment-tiled caches (above); otherwise row-major [ctx_max][64] / [64][ctx_max].  Same size per (b, hk) either way.
  uint16_t* Kb = kcache + ((int64_t)b * Hkv + hk) * ctx_max * 64;
  uint16_t* Vb = vtcache + ((int64_t)b * Hkv + hk) * 64 * ctx_max;
  // `pos` through the vector memory path (a uniform plain load becomes s_load + lgkmcnt(0) at the top of the kernel, ahead of
  // every other load): issued first, consumed after the qkv words and the first K/V tile are in flight (counted vmcnt).
  const int pos_v = __builtin_amdgcn_raw_buffer_load_b32(
      __builtin_amdgcn_make_buffer_rsrc((void*)ctx_len, 0, (int)gridDim.y * 4, 0x00020000), b * 4, 0, 0);

  // ---- loads of a tile pair: unconditional (rows beyond the cache end are clamped; invalid keys are masked later)
  uint4 kfA[4][2], kfB[4][2];
  uint2 vfA[4][2][2], vfB[4][2][2];
  auto load_tile = [&](uint4 (&kf)[4][2], uint2 (&vf)[4][2][2], int t) __attribute__((always_inline)) {
    if constexpr (FUSED) {
      const int tc = min(t, (ctx_max >> 6) - 1);
      const uint4* kp = (const uint4*)(Kb + (int64_t)tc * KV_TILE) + lane;
      const uint4* vp = (const uint4*)(Vb + (int64_t)tc * KV_TILE) + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          kf[i][f] = kp[(i * 2 + f) * 64];
          const uint4 v = vp[(i * 2 + f) * 64];
          vf[i][f][0] = make_uint2(v.x, v.y);
          vf[i][f][1] = make_uint2(v.z, v.w);
        }
      return;
    }
    const int j0 = min(t << 6, ctx_max - 64);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int key = j0 + kt * 16 + lq;
      kf[kt][0] = *(const uint4*)(Kb + (int64_t)key * 64 + lg * 8);
      kf[kt][1] = *(const uint4*)(Kb + (int64_t)key * 64 + 32 + lg * 8);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const uint16_t* vrow = Vb + (int64_t)(dt * 16 + lq) * ctx_max + j0 + 4 * lg;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        // the cache is zero-initialised and only ever holds finite values: stale keys beyond ctx meet P == 0
        vf[dt][s2][0] = *(const uint2*)(vrow + 32 * s2);
        vf[dt][s2][1] = *(const uint2*)(vrow + 32 * s2 + 16);
      }
    }
  };
  auto load_tiles = [&](int t0) __attribute__((always_inline)) {
    load_tile(kfA, vfA, t0);
    load_tile(kfB, vfB, t0 + DA_WAVES);
  };

  // Q fragments: column lq = query head hk*G + lq (zero beyond the group); d chunks (ks*4 + lg)*8
  uint4 qf[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  constexpr bool fused = FUSED;
  // fused RoPE (HF rotate_half: pairs (d, d+32)); a lane owns d in [8 lg, 8 lg + 8) and the partners + 32.  Every lane
  // loads (clamped head index) so the loads stay unconditional; lanes beyond the group zero their fragment afterwards.
  // Issue order = return order: the small qkv / rope-table words first, then the K/V tiles, so the rope arithmetic and the
  // LDS exchange run while the tiles are still in flight.
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, b0 = a0, b1 = a0, c0 = a0, c1 = a0, s0 = a0, s1 = a0;
  float kx1 = 0.f, kx2 = 0.f, vx = 0.f, kcs = 0.f, ksn = 0.f;
  const float* cs_row = nullptr;
  if constexpr (FUSED) {
    const float* row = qkv + (int64_t)b * ldqkv;
    const float* src = row + (hk * G + min(lq, G - 1)) * 64 + 8 * lg;
    a0 = *(const float4*)(src); a1 = *(const float4*)(src + 4);
    b0 = *(const float4*)(src + 32); b1 = *(const float4*)(src + 36);
    // new K (wave 0: d = lane & 31 pairs) and V (wave 1: d = lane); other waves load the same words and ignore them
    const float* ks_ = row + Hq * 64 + hk * 64;
    kx1 = ks_[lane & 31]; kx2 = ks_[(lane & 31) + 32];
    vx = row[(Hq + Hkv) * 64 + hk * 64 + lane];
  }
  // The first tile of every wave (keys < 512) is fetched unconditionally — its addresses do not depend on `pos`, so the loads
  // leave with the qkv words instead of one L2 round trip later (tiles beyond the context are in bounds of the cache and are
  // simply not used); the second tile (contexts beyond 512 keys) waits for `pos`.
  if constexpr (ABL == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kfA[i][0] = kfA[i][1] = kfB[i][0] = kfB[i][1] = make_uint4(0, 0, 0, 0);
      vfA[i][0][0] = vfA[i][0][1] = vfA[i][1][0] = vfA[i][1][1] = make_uint2(0, 0);
      vfB[i][0][0] = vfB[i][0][1] = vfB[i][1][0] = vfB[i][1][1] = make_uint2(0, 0);
    }
  } else {
    load_tile(kfA, vfA, min(wid, (ctx_max >> 6) - 1));
  }
  __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise hoists the wait for `pos` above half of these loads
  const int pos = __builtin_amdgcn_readfirstlane(pos_v);
  const int ctx = min(pos + ctx_add, ctx_max);
  const int ntiles = (ctx + 63) >> 6;
  if constexpr (FUSED) {   // the rope table row is the first load that needs `pos`
    cs_row = inv_freq + (int64_t)min(pos, ctx_max - 1) * 64;  // [pos][cos 32 | sin 32] table
    c0 = *(const float4*)(cs_row + 8 * lg); c1 = *(const float4*)(cs_row + 8 * lg + 4);
    s0 = *(const float4*)(cs_row + 32 + 8 * lg); s1 = *(const float4*)(cs_row + 32 + 8 * lg + 4);
    kcs = cs_row[lane & 31]; ksn = cs_row[32 + (lane & 31)];
  }
  if (ABL != 2 && wid + DA_WAVES < ntiles) load_tile(kfB, vfB, wid + DA_WAVES);
  if constexpr (FUSED) {
#define ROPE_LO(x1, x2, c, s_) ((x1) * (c) - (x2) * (s_))
#define ROPE_HI(x1, x2, c, s_) ((x2) * (c) + (x1) * (s_))
    const uint32_t qm = lq < G ? 0xFFFFFFFFu : 0u;
    qf[0] = make_uint4(pack2<DT>(ROPE_LO(a0.x, b0.x, c0.x, s0.x), ROPE_LO(a0.y, b0.y, c0.y, s0.y)) & qm,
                       pack2<DT>(ROPE_LO(a0.z, b0.z, c0.z, s0.z), ROPE_LO(a0.w, b0.w, c0.w, s0.w)) & qm,
                       pack2<DT>(ROPE_LO(a1.x, b1.x, c1.x, s1.x), ROPE_LO(a1.y, b1.y, c1.y, s1.y)) & qm,
                       pack2<DT>(ROPE_LO(a1.z, b1.z, c1.z, s1.z), ROPE_LO(a1.w, b1.w, c1.w, s1.w)) & qm);
    qf[1] = make_uint4(pack2<DT>(ROPE_HI(a0.x, b0.x, c0.x, s0.x), ROPE_HI(a0.y, b0.y, c0.y, s0.y)) & qm,
                       pack2<DT>(ROPE_HI(a0.z, b0.z, c0.z, s0.z), ROPE_HI(a0.w, b0.w, c0.w, s0.w)) & qm,
                       pack2<DT>(ROPE_HI(a1.x, b1.x, c1.x, s1.x), ROPE_HI(a1.y, b1.y, c1.y, s1.y)) & qm,
                       pack2<DT>(ROPE_HI(a1.z, b1.z, c1.z, s1.z), ROPE_HI(a1.w, b1.w, c1.w, s1.w)) & qm);
#undef ROPE_LO
#undef ROPE_HI
    if (wid == 0 && lane < 32) {
      const uint16_t k1 = Elem16<DT>::from_f32(kx1 * kcs - kx2 * ksn), k2 = Elem16<DT>::from_f32(kx2 * kcs + kx1 * ksn);
      s_newk[lane] = k1;
      s_newk[lane + 32] = k2;
      if (pos < ctx_max) {
        uint16_t* kt_ = Kb + (int64_t)(pos >> 6) * KV_TILE;
        kt_[ktile_index(pos & 63, lane)] = k1;
        kt_[ktile_index(pos & 63, lane + 32)] = k2;
      }
    } else if (wid == 1) {
      const uint16_t vv = Elem16<DT>::from_f32(vx);
      s_newv[lane] = vv;
      if (pos < ctx_max) Vb[(int64_t)(pos >> 6) * KV_TILE + vtile_index(pos & 63, lane)] = vv;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 t = *(const uint4*)(q + (int64_t)b * ldq + (hk * G + min(lq, G - 1)) * 64 + (ks * 4 + lg) * 8);
      const uint32_t qm = lq < G ? 0xFFFFFFFFu : 0u;
      qf[ks] = make_uint4(t.x & qm, t.y & qm, t.z & qm, t.w & qm);
    }
  }

  f32x4_t oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float mrun = NEG_BIG, lrun = 0.f;
  const float sc = scale * 1.4426950408889634f;

  // (always_inline: called four times; as a real call the fragment arrays would live in scratch)
  auto tile_step = [&](uint4 (&kf)[4][2], uint2 (&vf)[4][2][2], int t) __attribute__((always_inline)) {
    const int j0 = t << 6;
    if (fused && pos >= j0 && pos < j0 + 64 && pos < ctx_max) {
      // this tile holds the new token: its K row / V column come from LDS (the global copies were stale when loaded)
      const int rel = pos - j0;
      const uint4 nk0 = *(const uint4*)(s_newk + lg * 8), nk1 = *(const uint4*)(s_newk + 32 + lg * 8);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const bool hit = (kt * 16 + lq) == rel;
        // (component-wise: a ?: between two uint4 lvalues selects an address and drags the fragments into scratch)
        const uint4 o0 = kf[kt][0], o1 = kf[kt][1];
        kf[kt][0] = make_uint4(hit ? nk0.x : o0.x, hit ? nk0.y : o0.y, hit ? nk0.z : o0.z, hit ? nk0.w : o0.w);
        kf[kt][1] = make_uint4(hit ? nk1.x : o1.x, hit ? nk1.y : o1.y, hit ? nk1.z : o1.z, hit ? nk1.w : o1.w);
      }
      const int e = rel & 3;
      const uint32_t emask = (e & 1) ? 0xFFFF0000u : 0x0000FFFFu;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint32_t nv = (uint32_t)s_newv[dt * 16 + lq] << ((e & 1) * 16);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const bool hit = ((rel >> 2) == (8 * s2 + 4 * h + lg));
            const uint2 v = vf[dt][s2][h];
            const uint32_t x0 = (hit && e < 2) ? ((v.x & ~emask) | nv) : v.x;
            const uint32_t x1 = (hit && e >= 2) ? ((v.y & ~emask) | nv) : v.y;
            vf[dt][s2][h] = make_uint2(x0, x1);
          }
      }
    }
    if constexpr (ABL == 1) {   // keep the loads alive, skip the arithmetic
      uint32_t acc_ = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc_ ^= kf[i][0].x ^ kf[i][0].y ^ kf[i][0].z ^ kf[i][0].w ^ kf[i][1].x ^ kf[i][1].y ^ kf[i][1].z ^ kf[i][1].w;
        acc_ ^= vf[i][0][0].x ^ vf[i][0][0].y ^ vf[i][0][1].x ^ vf[i][0][1].y ^ vf[i][1][0].x ^ vf[i][1][0].y ^ vf[i][1][1].x ^ vf[i][1][1].y;
      }
      oacc[0][0] += __uint_as_float(acc_ & 0x3F800000u);
      return;
    }
    f32x4_t sacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
      a = mfma_block<DT>(kf[kt][0], qf[0], a);
      a = mfma_block<DT>(kf[kt][1], qf[1], a);
      sacc[kt] = a;
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + kt * 16 + 4 * lg + r;
        const float v = j < ctx ? sacc[kt][r] * sc : NEG_BIG;
        sacc[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mnew = fmaxf(mrun, mx);
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    mrun = mnew;
    float ls = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(sacc[kt][r] - mnew);
        sacc[kt][r] = e;
        ls += e;
      }
    lrun = lrun * alpha + ls;
    uint4 pf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      pf[s2].x = pack2<DT>(sacc[2 * s2][0], sacc[2 * s2][1]);
      pf[s2].y = pack2<DT>(sacc[2 * s2][2], sacc[2 * s2][3]);
      pf[s2].z = pack2<DT>(sacc[2 * s2 + 1][0], sacc[2 * s2 + 1][1]);
      pf[s2].w = pack2<DT>(sacc[2 * s2 + 1][2], sacc[2 * s2 + 1][3]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      oacc[dt][0] *= alpha; oacc[dt][1] *= alpha; oacc[dt][2] *= alpha; oacc[dt][3] *= alpha;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        oacc[dt] = mfma_block<DT>(make_uint4(vf[dt][s2][0].x, vf[dt][s2][0].y, vf[dt][s2][1].x, vf[dt][s2][1].y), pf[s2], oacc[dt]);
    }
  };

  if (wid < ntiles) tile_step(kfA, vfA, wid);  // wave-uniform conditions
  if (wid + DA_WAVES < ntiles) tile_step(kfB, vfB, wid + DA_WAVES);
  for (int t0 = wid + 2 * DA_WAVES; t0 < ntiles; t0 += 2 * DA_WAVES) {  // contexts beyond 1024 keys: one round trip per pair
    load_tiles(t0);
    tile_step(kfA, vfA, t0);
    if (t0 + DA_WAVES < ntiles) tile_step(kfB, vfB, t0 + DA_WAVES);
  }
  // ---- merge the waves (log-sum-exp)
  float l = lrun;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (lg == 0) { s_m[wid][lq] = mrun; s_l[wid][lq] = l; }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_o[wid][dt * 16 + 4 * lg + r][lq] = oacc[dt][r];
  __syncthreads();
  // thread -> (query col c = tid / 16, 4 d values); 16 cols x 16 d-quads = 256 threads
  if (tid < 256) {
    const int c = tid >> 4, d0 = (tid & 15) * 4;
    if (c < G) {
      float mm = NEG_BIG;
#pragma unroll
      for (int w_ = 0; w_ < DA_WAVES; ++w_) mm = fmaxf(mm, s_m[w_][c]);
      float lt = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w_ = 0; w_ < DA_WAVES; ++w_) {
        const float ww = __builtin_amdgcn_exp2f(s_m[w_][c] - mm);
        lt += s_l[w_][c] * ww;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += s_o[w_][d0 + r][c] * ww;
      }
      const float inv = lt > 0.f ? 1.f / lt : 0.f;
      uint2 u;
      u.x = pack2<DT>(o[0] * inv, o[1] * inv);
      u.y = pack2<DT>(o[2] * inv, o[3] * inv);
      *(uint2*)(out + (int64_t)b * ldo + (hk * G + c) * 64 + d0) = u;
    }
  }
}

// =========================================================================================== sampling
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

struct BestKV { float v; int i; };
__device__ __forceinline__ BestKV better(BestKV a, BestKV b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

constexpr int SV_PER = 32;  // values per thread -> V <= 8192

constexpr int SC_CAP = 256;  // candidate pool of the threshold pass
// lane l's value as a wave-uniform scalar (v_readlane_b32; l must be a constant after unrolling)
__device__ __forceinline__ float lane_bcast(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

__global__ __launch_bounds__(256) void sample_kernel(const cv_sample_params p) {
  __shared__ float s_red[4];
  __shared__ BestKV s_best[4];
  __shared__ float s_candp[64];
  __shared__ int s_candi[64];
  __shared__ __attribute__((aligned(16))) float s_scan[256];
  __shared__ float s_cv[SC_CAP];
  __shared__ int s_ci[SC_CAP];
  __shared__ int s_cnt;
  __shared__ float s_thr;
  __shared__ int s_flag[4];   // [0] need_fallback, [1] token, [2] done
  __shared__ float s_u2;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // every scalar of the bookkeeping is fetched up front, in one batch with the logits (each used to be its own
  // dependent round trip further down)
  const int fin0 = p.finished[b];
  const int step = p.step[b];
  const int min_len_b = p.min_len[b];
  const int n_em = p.n_emitted[b];
  const int max_len_b = p.max_len[b];
  const uint64_t key = p.seed ^ (p.nonce ? p.nonce[0] : 0ull);
  // the last win_size emitted ids, one per lane (repetition check, utils/common.py:111-113)
  const int widx = n_em - p.win_size + lane;
  const int recent = p.out_tokens[(int64_t)b * p.out_ld + min(max(widx, 0), p.out_ld - 1)];
  const bool recent_ok = lane < p.win_size && widx >= 0;
  int forced_tok = -1;
  if (p.forced) forced_tok = p.forced[(int64_t)b * p.forced_ld + min(n_em, p.forced_ld - 1)];
  const int last_tok = p.out_tokens[(int64_t)b * p.out_ld + min(max(n_em - 1, 0), p.out_ld - 1)];
  if (fin0 != 0) return;
  const float* lg = p.logits + (int64_t)b * p.ldl;
  const int V = p.V;

  float v[SV_PER];
  float mx = NEG_BIG;
#pragma unroll
  for (int j = 0; j < SV_PER; ++j) {
    const int i = tid + 256 * j;
    const float t = lg[min(i, V - 1)];  // unconditional clamped load
    v[j] = i < V ? t : NEG_BIG;
    mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx);
  if (lane == 0) s_red[wid] = mx;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < SV_PER; ++j) {
    v[j] = (tid + 256 * j) < V ? __expf(v[j] - mx) : 0.f;
    sum += v[j];
  }
  sum = wave_sum(sum);
  if (lane == 0) s_red[wid] = sum;
  __syncthreads();
  sum = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const float inv = 1.0f / sum;
  float lmax = -1.f;
#pragma unroll
  for (int j = 0; j < SV_PER; ++j) {
    v[j] *= inv;  // probabilities
    lmax = fmaxf(lmax, (tid + 256 * j) < V ? v[j] : -1.f);
  }

  // ---- nucleus candidates: descending prob, stable (lower index first); take while cum < top_p and n < top_k.
  // Threshold pass: the top_k-th largest of the 256 per-thread maxima bounds the top_k-th largest probability from below,
  // so every nucleus candidate is >= it; the few values above it are pooled in LDS and ordered by rank counting.  (The
  // first version extracted the maxima one by one: top_k x (32-value scan + 12 dependent cross-lane steps + 2 barriers),
  // 39 us per token.)  A pool overflow (near-constant logits) falls back to that loop.
  const int top_k = min(p.top_k, 64);
  // fallback_mode 1 (non_random_ras_sampling, utils/common.py:116-123): the repetition fallback is a second, wider nucleus
  // (top_p2, top_k2) over the same sorted candidates, so the candidate pass runs to the larger of the two k
  const int top_k2 = p.fallback_mode == 1 ? min(p.top_k2, 64) : 0;
  const int kth = max(top_k, top_k2);
  s_scan[tid] = lmax;
  __syncthreads();
  {
    int rank = 0;
#pragma unroll 8
    for (int i = 0; i < 256; i += 4) {
      const float4 o = *(const float4*)(s_scan + i);
      rank += (o.x > lmax || (o.x == lmax && i + 0 < tid)) ? 1 : 0;
      rank += (o.y > lmax || (o.y == lmax && i + 1 < tid)) ? 1 : 0;
      rank += (o.z > lmax || (o.z == lmax && i + 2 < tid)) ? 1 : 0;
      rank += (o.w > lmax || (o.w == lmax && i + 3 < tid)) ? 1 : 0;
    }
    if (rank == kth - 1) s_thr = lmax;
  }
  __syncthreads();
  const float thr = s_thr;
#pragma unroll
  for (int j = 0; j < SV_PER; ++j) {
    const int i = tid + 256 * j;
    if (i < V && v[j] >= thr) {
      const int slot = atomicAdd(&s_cnt, 1);
      if (slot < SC_CAP) { s_cv[slot] = v[j]; s_ci[slot] = i; }
    }
  }
  __syncthreads();
  const int pool = s_cnt;
  int nsorted;   // candidates in s_candp / s_candi, descending probability, lower index first on ties
  if (pool <= SC_CAP) {
    if (tid < pool) {
      const BestKV mine{s_cv[tid], s_ci[tid]};
      int rank = 0;
      for (int i = 0; i < pool; ++i) {
        const BestKV o{s_cv[i], s_ci[i]};
        rank += (o.v > mine.v || (o.v == mine.v && o.i < mine.i)) ? 1 : 0;
      }
      if (rank < 64) { s_candp[rank] = mine.v; s_candi[rank] = mine.i; }
    }
    nsorted = min(kth, pool);
    __syncthreads();
  } else {
    uint32_t taken = 0;
    int n = 0;
    const int want = min(kth, V);
    while (n < want) {
      BestKV best{-1.f, 0x7fffffff};
#pragma unroll
      for (int j = 0; j < SV_PER; ++j)
        if (!((taken >> j) & 1u) && (tid + 256 * j) < V && v[j] > best.v) best = BestKV{v[j], tid + 256 * j};
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        BestKV other{__shfl_xor(best.v, o, 64), __shfl_xor(best.i, o, 64)};
        best = better(best, other);
      }
      if (lane == 0) s_best[wid] = best;
      __syncthreads();
      best = better(better(s_best[0], s_best[1]), better(s_best[2], s_best[3]));
      __syncthreads();
      if ((best.i & 255) == tid) taken |= 1u << (best.i >> 8);
      if (tid == 0) { s_candp[n] = best.v; s_candi[n] = best.i; }
      ++n;
    }
    nsorted = want;
    __syncthreads();
  }
  // nucleus prefixes: sequential fp32 accumulation in candidate order (the reference's order), on wave-uniform values
  const float cp = s_candp[min(lane, max(nsorted - 1, 0))];
  int ncand = 0, ncand2 = 0;
  float cum = 0.f, cum2 = 0.f;
  {
    const int kmax = min(top_k, nsorted);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if (i >= kmax || cum >= p.top_p) break;
      cum += lane_bcast(cp, i);
      ++ncand;
    }
    const int kmax2 = min(top_k2, nsorted);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if (i >= kmax2 || cum2 >= p.top_p2) break;
      cum2 += lane_bcast(cp, i);
      ++ncand2;
    }
  }

  // ---- trials (llm.py:813-820): redraw while EOS is sampled before min_len
  const bool ignore_eos = step < min_len_b;
  int token = -1;
  int status = 0;
  for (int trial = 0;; ++trial) {
    if (wid == 0) {
      float u1, u2;
      if (p.uniforms) {
        u1 = p.uniforms[((int64_t)b * (p.max_trials + 1) + trial) * 2];
        u2 = p.uniforms[((int64_t)b * (p.max_trials + 1) + trial) * 2 + 1];
      } else {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)step, (uint32_t)b, (uint32_t)trial, 0u),
                                      make_uint2((uint32_t)key, (uint32_t)(key >> 32)));
        u1 = u01(r.x);
        u2 = u01(r.y);
      }
      // nucleus draw: inverse CDF over the candidate probabilities (renormalised); candidates live one per lane
      const float cpl = s_candp[min(lane, max(ncand - 1, 0))];
      const int cil = s_candi[min(lane, max(ncand - 1, 0))];
      const float cpl2 = s_candp[min(lane, max(ncand2 - 1, 0))];
      const int cil2 = s_candi[min(lane, max(ncand2 - 1, 0))];
      const float target = u1 * cum;
      float c = 0.f;
      int pick = ncand - 1;
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        if (i >= ncand) break;
        c += lane_bcast(cpl, i);
        if (c > target) { pick = i; break; }
      }
      const int tok = __shfl(cil, max(pick, 0), 64);
      // repetition check over the last win_size emitted tokens (utils/common.py:111-113)
      const int rep = __popcll(__ballot(recent_ok && recent == tok));
      const bool repeated = (float)rep >= (float)p.win_size * p.tau_r;
      int tok_out = tok;
      if (repeated && p.fallback_mode == 1) {   // second nucleus draw over the wider prefix (common.py:121-122)
        const float target2 = u2 * cum2;
        float c2 = 0.f;
        int pick2 = ncand2 - 1;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
          if (i >= ncand2) break;
          c2 += lane_bcast(cpl2, i);
          if (c2 > target2) { pick2 = i; break; }
        }
        tok_out = __shfl(cil2, max(pick2, 0), 64);
      }
      if (lane == 0) {
        s_flag[0] = (repeated && p.fallback_mode == 0) ? 1 : 0;
        s_flag[1] = tok_out;
        s_u2 = u2;
      }
    }
    __syncthreads();
    if (s_flag[0]) {
      // random_sampling over the full distribution: thread t owns the contiguous range [t*chunk, (t+1)*chunk)
      const int chunk = (V + 255) / 256;
      const int i0 = tid * chunk, i1 = min(V, i0 + chunk);
      float loc = 0.f;
      for (int i = i0; i < i1; ++i) loc += __expf(lg[i] - mx) * inv;
      s_scan[tid] = loc;
      __syncthreads();
      if (tid == 0) {
        float tot = 0.f;
        for (int i = 0; i < 256; ++i) tot += s_scan[i];
        const float target = s_u2 * tot;
        float c = 0.f;
        int owner = 255;
        for (int i = 0; i < 256; ++i) {
          if (c + s_scan[i] > target) { owner = i; break; }
          c += s_scan[i];
        }
        // walk the owner's range
        const int j0 = owner * chunk, j1 = min(V, j0 + chunk);
        int tok = max(j1 - 1, 0);
        for (int i = j0; i < j1; ++i) {
          c += __expf(lg[i] - mx) * inv;
          if (c > target) { tok = i; break; }
        }
        s_flag[1] = tok;
      }
      __syncthreads();
    }
    if (tid == 0) {
      int done = 1;
      if (ignore_eos && s_flag[1] == p.eos) {
        done = 0;
        if (trial + 1 > p.max_trials) done = 2;  // sampling stalled
      }
      s_flag[2] = done;
    }
    __syncthreads();
    const int done = s_flag[2];
    token = s_flag[1];
    __syncthreads();
    if (done) { status = done; break; }
  }

  // ---- bookkeeping (llm.py:866-874)
  if (p.forced) {
    const int f = (n_em < p.forced_ld) ? forced_tok : -2;
    if (f >= 0) { token = f; status = 1; }
    else if (f == -2) token = p.eos;  // forced list exhausted -> stop
  }
  bool emit = false;
  int fin = 0;
  if (status == 2) fin = 3;
  else if (token == p.eos) fin = 1;
  else if (token < p.eos) emit = true;
  if (emit && n_em >= p.out_ld) { emit = false; fin = 2; }
  // next-step input embedding.  Emitted id -> its embedding; skipped id (> EOS, llm.py:869-870 `continue`) -> the
  // previous input again = embedding of the last emitted id (the residual stream has overwritten x meanwhile).
  int next_in = emit ? token : ((!fin && n_em > 0) ? last_tok : -1);
  if (next_in >= 0) {
    const float* e = p.emb_table + (int64_t)next_in * p.emb_dim;
    for (int i = tid; i < p.emb_dim; i += 256) p.x[(int64_t)b * p.ldx + i] = e[i];
  }
  if (tid == 0) {
    if (emit) {
      p.out_tokens[(int64_t)b * p.out_ld + n_em] = token;
      p.n_emitted[b] = n_em + 1;
    }
    p.step[b] = step + 1;
    p.pos[b] += 1;
    if (!fin && step + 1 >= max_len_b) fin = 2;
    if (fin) p.finished[b] = fin;
  }
}

}  // namespace

#define DISPATCH_16(dt, CALL)                                   \
  switch (dt) {                                                 \
    case CV_BF16: { constexpr int DT = CV_BF16; CALL; } break;  \
    case CV_F16: { constexpr int DT = CV_F16; CALL; } break;    \
    default: return CV_ERR_UNSUPPORTED;                         \
  }

extern "C" int cv_sizeof_skinny_params(void) { return (int)sizeof(cv_skinny_params); }
extern "C" int cv_sizeof_sample_params(void) { return (int)sizeof(cv_sample_params); }

extern "C" int cv_pack_skinny(const void* W, void* Wp, int32_t N, int32_t K, int32_t interleave, void* stream) {
  if (!W || !Wp || N <= 0 || K <= 0 || (K & 31)) return CV_ERR_ARG;
  if (interleave && (N & 31)) return CV_ERR_ARG;
  const int ntiles = (N + 15) / 16;
  const int64_t total = (int64_t)ntiles * (K >> 5) * 64;
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_skinny_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)W, (uint16_t*)Wp, N, K,
                     ntiles, interleave);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_skinny_gemm(const cv_skinny_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  cv_skinny_params p = *pp;
  const bool norm = p.ngamma != nullptr;
  if (p.M <= 0 || p.M > 32 || p.N <= 0 || p.K <= 0 || (p.K & 31) || !p.Wp) return CV_ERR_ARG;
  if (p.M > 16 && (norm || p.mode == 3)) return CV_ERR_UNSUPPORTED;   // two row groups: plain-row forms of modes 0 / 1 / 2 only
  if (!norm && (!p.A || (p.lda & 7))) return CV_ERR_ARG;
  if (norm) {
    if (!p.nx || (p.K & 63) || p.K > 1024 || (p.ldnx & 3) || p.nx_out == p.nx) return CV_ERR_ARG;
    if (p.n_nslab > 0 && (!p.nslabs || (p.ld_nslab & 3) || (p.nslab_stride & 3))) return CV_ERR_ARG;
    if (p.n_nslab != 0 && p.n_nslab != 1 && p.n_nslab != 2 && p.n_nslab != 4) return CV_ERR_ARG;
  }
  if (p.ksplit <= 0) p.ksplit = 1;
  const int ntiles = (p.N + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  const size_t img = norm ? (size_t)(p.K >> 5) * 1024 : 0;
  const int nm = !norm ? 0 : 1 + p.n_nslab;
  const int tpr = p.M <= 4 ? 64 : (p.M <= 8 ? 32 : 16);
  const int tpw = p.mode == 2 ? 2 : 1;
  if (p.mode == 2) {
    if (p.ksplit != 1 || (ntiles & 1) || !p.out_act || (p.ldoa & 3)) return CV_ERR_ARG;
    if (p.rs_part) {
      if (norm || p.n_rs_part <= 0 || p.n_rs_part > 64) return CV_ERR_ARG;
      p.max_wgs = 0;
    }
  } else if (p.mode == 3) {
    if (p.ksplit != 1 || !p.out_act || norm) return CV_ERR_ARG;
    p.max_wgs = 0;   // single-shot kernel only
  } else if (p.mode == 1 && (p.xb_out || p.ss_part)) {
    if (!p.out_f32 || p.ksplit != 1 || !p.xb_out || !p.ss_part || p.ldxb < p.N) return CV_ERR_ARG;
    p.max_wgs = 0;
  } else {
    if (!p.out_f32) return CV_ERR_ARG;
    if (p.mode == 1 && p.ksplit != 1) return CV_ERR_ARG;
  }
  // k-steps one wave covers: the streaming kernel takes them as one batch of U fragment loads
  const int per_wave = ((((p.K >> 5) + p.ksplit - 1) / p.ksplit) + 3) >> 2;
  const int ngroups = ntiles / tpw;
  if (p.mode == 2 && p.rs_part && per_wave > 8) return CV_ERR_UNSUPPORTED;   // split-norm consumer: K <= 1024 only
  if (p.M > 16) {   // 17..32 rows: two 16-row groups share every weight fragment (MR = 2); single-shot workgroups
    dim3 grid(ngroups, p.ksplit);
    const size_t lds = 2 * 4 * tpw * 64 * 4 * sizeof(float);
    const bool exact7 = per_wave == 7 && ((p.K >> 5) == 28 * p.ksplit);
    if (tpw == 2) {
      if (!p.rs_part) return CV_ERR_UNSUPPORTED;
      // (two (gate, up) tile pairs per workgroup — skinny_kernel<DT, 4, 0, 16, 7, true, 2>, half the activation traffic — was measured at
      // 19.9 vs 20.7 us on 64 CUs and 10.8 vs 10.4 us on all: at 240 VGPRs its 152 workgroups still need two rounds; not dispatched)
      if (exact7) { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 2, 0, 16, 7, true, 2>), grid, dim3(256), lds, st, p)); }
      else { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 2, 0, 16, 8, true, 2>), grid, dim3(256), lds, st, p)); }
    } else if (per_wave > 8 && per_wave <= 10 && !(ntiles & 1)) {
      // long-K form (the down projection): with two row groups the activation slice a workgroup reads (32 rows x its K range) is twice
      // its weight tile, so two weight tiles share it (TPW = 2: half the workgroups, half the activation traffic)
      dim3 grid2(ngroups / 2, p.ksplit);
      const size_t lds2 = 2 * 4 * 2 * 64 * 4 * sizeof(float);
      DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 2, 0, 16, 10, false, 2>), grid2, dim3(256), lds2, st, p));
    } else if (exact7) { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 1, 0, 16, 7, false, 2>), grid, dim3(256), lds, st, p)); }
    else if (per_wave <= 8) { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 1, 0, 16, 8, false, 2>), grid, dim3(256), lds, st, p)); }
    else if (per_wave <= 12) { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 1, 0, 16, 12, false, 2>), grid, dim3(256), lds, st, p)); }
    else { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 1, 0, 16, 16, false, 2>), grid, dim3(256), lds, st, p)); }
    CV_CHECK_LAUNCH();
    return CV_OK;
  }
  if (per_wave > 16 || (tpw == 2 && per_wave > 8)) {
    if (norm) return CV_ERR_UNSUPPORTED;  // cannot happen: the prologue needs K <= 1024
    dim3 grid(ngroups, p.ksplit);
    const size_t lds = 4 * tpw * 64 * 4 * sizeof(float);
    if (tpw == 2) { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 2, 0, 16, 8>), grid, dim3(256), lds, st, p)); }
    else { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 1, 0, 16, 8>), grid, dim3(256), lds, st, p)); }
    CV_CHECK_LAUNCH();
    return CV_OK;
  }
  int gx = ngroups;
  if (p.max_wgs > 0) gx = std::min(ngroups, std::max(1, p.max_wgs / p.ksplit));
  dim3 grid(gx, p.ksplit);
  const bool loop = gx < ngroups;
  const size_t lds = (loop ? 2 : 1) * 4 * tpw * 64 * 4 * sizeof(float) + img;
  // every wave's K slice is exactly 7 k-steps (K = 896 * ksplit): single-shot launches use the clamp-free U = 7 form
  const bool exact7 = !loop && per_wave == 7 && ((p.K >> 5) == 28 * p.ksplit);
#define SK_LAUNCH7(TPW_, NM_, TPR_) \
  DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, TPW_, NM_, TPR_, 7>), grid, dim3(256), lds, st, p))
#define SK_LAUNCH(TPW_, NM_, TPR_, U_)                                                                                          \
  do {                                                                                                                          \
    if (loop) { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_stream_kernel<DT, TPW_, NM_, TPR_, U_, true>), grid, dim3(256), lds, st, p)); } \
    else { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, TPW_, NM_, TPR_, U_>), grid, dim3(256), lds, st, p)); }                   \
  } while (0)
#define SK_TPR(TPW_, NM_)                                                                                                  \
  do {                                                                                                                     \
    if (tpr == 64) { SK_LAUNCH(TPW_, NM_, 64, 8); }                                                                        \
    else if (tpr == 32) { if (exact7) { SK_LAUNCH7(TPW_, NM_, 32); } else { SK_LAUNCH(TPW_, NM_, 32, 8); } }               \
    else { if (exact7) { SK_LAUNCH7(TPW_, NM_, 16); } else { SK_LAUNCH(TPW_, NM_, 16, 8); } }                               \
  } while (0)
#define SK_NORM(TPW_)                          \
  do {                                         \
    if (nm == 1) { SK_TPR(TPW_, 1); }          \
    else if (nm == 2) { SK_TPR(TPW_, 2); }     \
    else if (nm == 3) { SK_TPR(TPW_, 3); }     \
    else { SK_TPR(TPW_, 5); }                  \
  } while (0)
  if (tpw == 2) {
    if (nm == 0 && p.rs_part) {
      if (exact7) { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 2, 0, 16, 7, true>), grid, dim3(256), lds, st, p)); }
      else { DISPATCH_16(p.dtype, hipLaunchKernelGGL((skinny_kernel<DT, 2, 0, 16, 8, true>), grid, dim3(256), lds, st, p)); }
    } else if (nm == 0) { SK_LAUNCH(2, 0, 16, 8); } else { SK_NORM(2); }
  } else if (nm == 0) {
    if (exact7) { SK_LAUNCH7(1, 0, 16); }
    else if (per_wave <= 8) { SK_LAUNCH(1, 0, 16, 8); } else if (per_wave <= 12) { SK_LAUNCH(1, 0, 16, 12); } else { SK_LAUNCH(1, 0, 16, 16); }
  } else {
    SK_NORM(1);
  }
#undef SK_NORM
#undef SK_TPR
#undef SK_LAUNCH7
#undef SK_LAUNCH
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_rmsnorm_reduce(float* x, int32_t ldx, const float* slabs, int32_t nslab, int64_t slab_stride, int32_t ld_slab,
                                 const float* gamma, float eps, void* xn, int32_t ldxn, int32_t dtype, int32_t rows, int32_t dim, void* stream) {
  if (!x || !gamma || !xn || rows <= 0 || dim <= 0 || (dim & 3) || dim > 4096 || (ldx & 3) || (ldxn & 3)) return CV_ERR_ARG;
  if (nslab > 0 && (!slabs || (ld_slab & 3) || (slab_stride & 3))) return CV_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case CV_F32: hipLaunchKernelGGL(rmsnorm_reduce_kernel<CV_F32>, dim3(rows), dim3(256), 0, st, x, ldx, slabs, nslab, slab_stride, ld_slab, gamma, eps, xn, ldxn, dim); break;
    case CV_BF16: hipLaunchKernelGGL(rmsnorm_reduce_kernel<CV_BF16>, dim3(rows), dim3(256), 0, st, x, ldx, slabs, nslab, slab_stride, ld_slab, gamma, eps, xn, ldxn, dim); break;
    case CV_F16: hipLaunchKernelGGL(rmsnorm_reduce_kernel<CV_F16>, dim3(rows), dim3(256), 0, st, x, ldx, slabs, nslab, slab_stride, ld_slab, gamma, eps, xn, ldxn, dim); break;
    default: return CV_ERR_ARG;
  }
  CV_CHECK_LAUNCH();
  return CV_OK;
}

namespace {
// CosyVoice-v1 TransformerLM cached decode step: the fp32 projection row [q + pos_bias_u | q + pos_bias_v | k | v] (4 D) of the
// new token t -> 16-bit query pair, K cache row t, V^T cache column t (TransformerEncoder.forward_chunk's att_cache append,
// transformer/encoder.py:185-274, transformer/attention.py:249-330).
template <int DT>
__global__ __launch_bounds__(256) void relpos_append_kernel(const float* qkv, void* qq, void* kc, void* vtc, int D, int t, int vt_ld) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 4 * D) return;
  const uint16_t v = Elem16<DT>::from_f32(qkv[i]);
  if (i < 2 * D) ((uint16_t*)qq)[i] = v;
  else if (i < 3 * D) ((uint16_t*)kc)[(int64_t)t * D + (i - 2 * D)] = v;
  else ((uint16_t*)vtc)[(int64_t)(i - 3 * D) * vt_ld + t] = v;
}
}  // namespace

extern "C" int cv_relpos_append(const float* qkv, void* qq, void* kcache, void* vtcache, int32_t dtype, int32_t D, int32_t t,
                                int32_t vt_ld, void* stream) {
  if (!qkv || !qq || !kcache || !vtcache || D <= 0 || t < 0 || t >= vt_ld) return CV_ERR_ARG;
  dim3 grid((4 * D + 255) / 256);
  switch (dtype) {
    case CV_BF16: hipLaunchKernelGGL(relpos_append_kernel<CV_BF16>, grid, dim3(256), 0, (hipStream_t)stream, qkv, qq, kcache, vtcache, D, t, vt_ld); break;
    case CV_F16: hipLaunchKernelGGL(relpos_append_kernel<CV_F16>, grid, dim3(256), 0, (hipStream_t)stream, qkv, qq, kcache, vtcache, D, t, vt_ld); break;
    default: return CV_ERR_ARG;
  }
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_rope_append(const float* qkv, int32_t ldqkv, const int32_t* pos_base, int32_t rows, int32_t rows_per_seq,
                              int32_t Hq, int32_t Hkv, const float* inv_freq, void* q_out, int32_t ldq, void* kcache, void* vtcache,
                              int32_t ctx_max, int32_t dtype, void* stream) {
  if (!qkv || !pos_base || !inv_freq || !q_out || !kcache || !vtcache || rows <= 0 || rows_per_seq <= 0 || Hq <= 0 || Hkv <= 0) return CV_ERR_ARG;
  DISPATCH_16(dtype, hipLaunchKernelGGL(rope_append_kernel<DT>, dim3(rows), dim3(256), 0, (hipStream_t)stream, qkv, ldqkv, pos_base,
                                        rows_per_seq, Hq, Hkv, inv_freq, (uint16_t*)q_out, ldq, (uint16_t*)kcache, (uint16_t*)vtcache, ctx_max));
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_kv_retile(const void* k_rm, const void* vt_rm, void* k_tiled, void* vt_tiled, int32_t B, int32_t Hkv,
                            int32_t ctx_max, int32_t n_keys, void* stream) {
  if (!k_rm || !vt_rm || !k_tiled || !vt_tiled || B <= 0 || Hkv <= 0 || ctx_max <= 0 || (ctx_max & 63) || n_keys < 0 || n_keys > ctx_max)
    return CV_ERR_ARG;
  if (((uintptr_t)k_rm | (uintptr_t)vt_rm | (uintptr_t)k_tiled | (uintptr_t)vt_tiled) & 15) return CV_ERR_ARG;
  const int ntiles = (n_keys + 63) >> 6;
  if (ntiles == 0) return CV_OK;
  hipLaunchKernelGGL(kv_retile_kernel, dim3(ntiles, B * Hkv), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)k_rm,
                     (const uint16_t*)vt_rm, (uint16_t*)k_tiled, (uint16_t*)vt_tiled, ctx_max);
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_decode_attention(const void* q, int32_t ldq, const void* kcache, const void* vtcache, const int32_t* ctx_len,
                                   int32_t ctx_add, void* out, int32_t ldo, int32_t B, int32_t Hq, int32_t Hkv, int32_t ctx_max,
                                   float scale, int32_t dtype, const float* qkv, int32_t ldqkv, const float* inv_freq, void* stream) {
  if (!kcache || !vtcache || !ctx_len || !out || B <= 0 || Hq <= 0 || Hkv <= 0 || (Hq % Hkv) || Hq / Hkv > 16) return CV_ERR_ARG;
  if (!qkv && (!q || (ldq & 7))) return CV_ERR_ARG;
  if (qkv && (!inv_freq || ldqkv < (Hq + 2 * Hkv) * 64)) return CV_ERR_ARG;
  if ((ldo & 3) || (ctx_max & 63)) return CV_ERR_ARG;
  dim3 grid(Hkv, B);
  static const int abl = getenv("CV_DA_ABL") ? atoi(getenv("CV_DA_ABL")) : 0;
  if (qkv && abl == 1) {
    DISPATCH_16(dtype, hipLaunchKernelGGL((decode_attn_kernel<DT, true, 1>), grid, dim3(512), 0, (hipStream_t)stream, (const uint16_t*)q, ldq,
                                          (uint16_t*)kcache, (uint16_t*)vtcache, ctx_len, ctx_add, (uint16_t*)out, ldo, Hq, Hkv,
                                          ctx_max, scale, qkv, ldqkv, inv_freq));
  } else if (qkv && abl == 2) {
    DISPATCH_16(dtype, hipLaunchKernelGGL((decode_attn_kernel<DT, true, 2>), grid, dim3(512), 0, (hipStream_t)stream, (const uint16_t*)q, ldq,
                                          (uint16_t*)kcache, (uint16_t*)vtcache, ctx_len, ctx_add, (uint16_t*)out, ldo, Hq, Hkv,
                                          ctx_max, scale, qkv, ldqkv, inv_freq));
  } else if (qkv) {
    DISPATCH_16(dtype, hipLaunchKernelGGL((decode_attn_kernel<DT, true>), grid, dim3(512), 0, (hipStream_t)stream, (const uint16_t*)q, ldq,
                                          (uint16_t*)kcache, (uint16_t*)vtcache, ctx_len, ctx_add, (uint16_t*)out, ldo, Hq, Hkv,
                                          ctx_max, scale, qkv, ldqkv, inv_freq));
  } else {
    DISPATCH_16(dtype, hipLaunchKernelGGL((decode_attn_kernel<DT, false>), grid, dim3(512), 0, (hipStream_t)stream, (const uint16_t*)q, ldq,
                                          (uint16_t*)kcache, (uint16_t*)vtcache, ctx_len, ctx_add, (uint16_t*)out, ldo, Hq, Hkv,
                                          ctx_max, scale, qkv, ldqkv, inv_freq));
  }
  CV_CHECK_LAUNCH();
  return CV_OK;
}

extern "C" int cv_sample_ras(const cv_sample_params* pp, void* stream) {
  if (!pp) return CV_ERR_ARG;
  const cv_sample_params& p = *pp;
  if (!p.logits || p.V <= 0 || p.V > 256 * SV_PER || p.B <= 0 || !p.min_len || !p.max_len || !p.step || !p.pos || !p.n_emitted ||
      !p.finished || !p.out_tokens || !p.emb_table || !p.x || p.top_k <= 0)
    return CV_ERR_ARG;
  hipLaunchKernelGGL(sample_kernel, dim3(p.B), dim3(256), 0, (hipStream_t)stream, p);
  CV_CHECK_LAUNCH();
  return CV_OK;
}
