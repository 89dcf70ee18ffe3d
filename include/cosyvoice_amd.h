/*
 * cosyvoice_amd — C ABI of the MI355X (gfx950) hot-path library  libcosyvoice_amd.so
 *
 * The reference (duj12/CosyVoice) has no C/FFI boundary: its plug-in seam is Python
 * duck-typing.  This header is the boundary the build adds underneath that
 * seam.  Conventions for every entry point:
 *   - raw device pointers (tensor.data_ptr()), caller owns every buffer and workspace;
 *   - explicit stream (hipStream_t passed as void*); nothing is launched on another stream;
 *   - returns 0 on success, a negative cv_status otherwise; no C++ exception crosses the ABI;
 *   - no hidden allocation or synchronisation inside launch functions (graph-capturable);
 *   - thread-compatible: distinct streams + workspaces may be driven concurrently.
 *
 * Each op names the reference code it replaces (file:line under /root/reference).
 */
#ifndef COSYVOICE_AMD_H
#define COSYVOICE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CV_OK = 0, CV_ERR_ARG = -1, CV_ERR_LAUNCH = -2, CV_ERR_UNSUPPORTED = -3 } cv_status;
/* CV_F32X3 (cv_gemm only): fp32 tensors whose products are computed as three bf16 MFMAs on hi/lo splits (relative error
 * ~2^-16, fp32 accumulate) instead of the exact-f32 MFMA, whose peak is 1/16 of the bf16 one. */
typedef enum { CV_F32 = 0, CV_BF16 = 1, CV_F16 = 2, CV_F32X3 = 3 } cv_dtype;
typedef enum {
  CV_ACT_NONE = 0, CV_ACT_GELU = 1, CV_ACT_SILU = 2, CV_ACT_MISH = 3, CV_ACT_LEAKY = 4,
  CV_ACT_ELU = 5, CV_ACT_SNAKE = 6, CV_ACT_TANH = 7, CV_ACT_SWIGLU = 8
} cv_act;
typedef enum { CV_OUT_ROWMAJOR = 0, CV_OUT_QKV = 1 } cv_out_mode;

int cv_version(void);
const char* cv_arch(void); /* "gfx950" */

/* ------------------------------------------------------------------------------------------
 * cv_gemm — C[z][m][n] = epilogue( sum_k A[z][row(m,k)][ci(k)] * W[n][k] ), MFMA, fp32 accumulate.
 * One kernel family for every Linear and Conv1d/ConvTranspose1d on the path (channels-last
 * activations, implicit im2col: k = tap*cin + ci, A row = m*a_row_stride + tap_base + tap*tap_step,
 * rows outside [0,a_rows) read as zero = the conv padding).
 * Replaces: torch.nn.Linear / Conv1d / ConvTranspose1d calls of
 *   cosyvoice/flow/decoder.py:36-85,222-334, flow/components/transformer.py:243-316,
 *   transformer/attention.py:69-76, transformer/positionwise_feed_forward.py:47-56,
 *   hifigan/generator.py:91-98,349-381, hifigan/f0_predictor.py:52-55, HF Qwen2 linears (llm/llm.py:754-766).
 * Epilogue: v = (acc + bias[n] + res[m][n] + res2[m][n]) * out_scale; out_f32 = v; out_act = T(act(v)).
 * ------------------------------------------------------------------------------------------ */
typedef struct cv_gemm_params {
  int32_t dtype;            /* cv_dtype of A, W and out_act */
  int32_t M, N, K;          /* per-batch output rows, output cols, reduction length (= taps*cin) */
  int32_t batch, batch_inner; /* grid.z = batch; z -> (z1 = z / batch_inner, z0 = z % batch_inner) */
  const void* A; int64_t a_bs0, a_bs1; int32_t lda; int32_t a_rows;
  int32_t cin, a_row_stride, tap_base, tap_step;
  const void* W; int64_t w_bs0, w_bs1; int32_t ldw;
  const float* bias;
  const float* res;  int64_t res_bs0, res_bs1;  int32_t ldres;
  const float* res2; int32_t ldres2;            /* same batch strides as res */
  float out_scale;
  int32_t act; const float* act_param; float act_slope;
  float* out_f32; int64_t o32_bs0, o32_bs1; int32_t ldo32;
  void* out_act;  int64_t oa_bs0, oa_bs1;   int32_t ldoa;
  int32_t out_row_stride, out_row_off, out_rows; /* output row = m*out_row_stride + out_row_off, stored if in [0,out_rows) */
  int32_t out_mode;         /* cv_out_mode */
  /* CV_OUT_QKV: columns [0,q_cols) -> out_act (ld ldoa) scaled by q_scale; [q_cols,q_cols+k_cols) -> k_out (ld ldk);
     the rest -> vt_out transposed: vt_out[((z*vt_heads + h)*64 + d) * vt_ld + m], head_dim 64 */
  int32_t q_cols, k_cols; float q_scale;
  void* k_out; int64_t k_bs; int32_t ldk;
  void* vt_out; int32_t vt_heads, vt_ld;
  /* CV_F32X3 only — PRE-SPLIT storage of fp32 tensors: every aligned group of 8 values (32 bytes) holds [8 x bf16 hi | 8 x bf16 lo]
   * with hi = bf16(v), lo = bf16(v - hi), i.e. what the bf16x3 kernel would otherwise compute each time it stages the values
   * (row lengths / K / cin / leading dimensions % 8 == 0).
   * bit 0: A is pre-split; bit 1: W is pre-split (both or neither); bit 2: out_act is WRITTEN pre-split (row-major, N % 8 == 0).
   * Results are bit-identical to the plain CV_F32X3 launch. */
  int32_t x3_flags; int32_t reserved_x3;
} cv_gemm_params;
int cv_gemm(const cv_gemm_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * cv_layernorm — y = act((x - mean) * rstd * gamma + beta + add[row_group]) over the last dim;
 * rms != 0 -> RMSNorm (no mean, no beta).  fp32 in, fp32 and/or 16-bit out.
 * Replaces nn.LayerNorm / Qwen2RMSNorm (+ Mish of CausalBlock1D, flow/decoder.py:36-49).
 * ------------------------------------------------------------------------------------------ */
typedef struct cv_norm_params {
  int32_t rows, dim; int32_t rms; float eps;
  const float* x; int32_t ldx;
  const float* gamma; const float* beta;
  const float* add; int32_t add_ld; int32_t rows_per_group; /* optional per-group vector added after affine (time-embedding) */
  int32_t act;            /* cv_act applied last (NONE or MISH) */
  float out_scale;
  int32_t out_dtype;      /* dtype of out_act */
  float* out_f32; int32_t ldo32;
  void* out_act; int32_t ldoa;
} cv_norm_params;
int cv_layernorm(const cv_norm_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * cv_groupnorm_cl — nn.GroupNorm(groups, C) on a channels-last fp32 tensor [B][T][C] (statistics over T x C/groups),
 * fused affine, optional Mish, optional per-batch-row vector added after the activation (time embedding).
 * Replaces Block1D's GroupNorm+Mish of the non-causal CFM estimator (matcha decoder Block1D via flow/decoder.py:129-130,
 * CosyVoice-v1) and InterpolateRegulator's GroupNorm (flow/length_regulator.py:36-38).  Two launches (chunk statistics,
 * merge + apply); `partial` is caller-owned scratch of cv_groupnorm_workspace_floats(B, T, groups) floats.
 * x may alias out_f32.
 * ------------------------------------------------------------------------------------------ */
typedef struct cv_groupnorm_params {
  int32_t B, T, C, groups; float eps;
  const float* x; int64_t x_bs; int32_t ldx;
  const float* gamma; const float* beta;
  const float* add; int32_t add_ld;
  int32_t act;            /* CV_ACT_NONE or CV_ACT_MISH */
  int32_t out_dtype;      /* dtype of out_act */
  float* out_f32; int64_t o32_bs; int32_t ldo32;
  void* out_act; int64_t oa_bs; int32_t ldoa;
  float* partial;
} cv_groupnorm_params;
int cv_sizeof_groupnorm_params(void);
int64_t cv_groupnorm_workspace_floats(int32_t B, int32_t T, int32_t groups);
int cv_groupnorm_cl(const cv_groupnorm_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * cv_attention — flash attention, head_dim 64, 16-bit operands, fp32 softmax/accumulate.
 *   S[i][j] = scale * q_i . k_j + bias[i][j];  masks: j < klen[b]; chunk: j < (i/chunk+1)*chunk; causal: j <= i + causal_off
 * Q [B][Tq][ldq] (head h at column h*64), K [B][Tk][ldk] (kv head at column hk*64),
 * Vt [B][Hkv][64][vt_ld] (keys contiguous), out [B][Tq][ldo] 16-bit.
 * Replaces diffusers Attention (flow/components/transformer.py:196-204), RelPositionMultiHeadedAttention
 * (transformer/attention.py:249-330; bias = the rel-shifted matrix_bd as a strided view) and HF Qwen2 SDPA prefill.
 * ------------------------------------------------------------------------------------------ */
typedef struct cv_attn_params {
  int32_t dtype; int32_t B, H, Hkv, Tq, Tk;
  const void* q; int64_t q_bs; int32_t ldq;
  const void* k; int64_t k_bs; int32_t ldk;
  const void* vt; int32_t vt_ld;
  void* out; int64_t o_bs; int32_t ldo;
  float scale;
  const int32_t* klen;      /* [B] or null (= Tk) */
  int32_t chunk;            /* 0 = off */
  int32_t causal; int32_t causal_off;
  const float* bias; int64_t bias_bs, bias_hs; int32_t bias_ld; /* fp32 additive bias view or null */
  int32_t q_off;            /* chunk mask: query row i sits at position q_off + i of the key sequence (incremental encoder: only the
                               rows behind the cached prefix are queries); 0 = queries and keys start together */
  int64_t q_hs, k_hs;       /* element stride between heads of Q / K (0 -> 64: heads packed inside a row) */
} cv_attn_params;
int cv_attention(const cv_attn_params* p, void* stream);


/* ------------------------------------------------------------------------------------------
 * cv_tblock_head / cv_tblock_tail — the CFM estimator's BasicTransformerBlock
 * (flow/components/transformer.py:243-316 forward, :159-236 ctor, FeedForward :83-134; diffusers 0.27.2 Attention / GELU
 * semantics as SURVEY.md §8c states them) as two row-block kernels around the cv_attention launch:
 *   head: xn = LayerNorm(norm1)(x);  qk[row] = [xn Wq^T | xn Wk^T] (row-major, ldqk);  vt = (xn Wv^T)^T per head
 *   tail: x1 = x + ao Wo^T + bo (skipped when ao == NULL: x1 = x);  xn = LayerNorm(norm3)(x1);
 *         x  = x1 + gelu_erf(xn W1^T + bf1) W2^T + bf2   (in place);  out_act = T(x) when given
 * x [R][T][ldx] fp32 residual stream (C = 256 channels), 64 rows of one sequence per workgroup; weights pre-packed
 * by cv_pack_skinny (MFMA fragment order) and streamed L2 -> VGPR once per 64 rows; the 1024-wide GELU intermediate
 * and the LayerNorm outputs never leave the CU.  Fixed widths C = 256, inner = 8 heads x 64 = 512, ff = 1024
 * (cosyvoice2 yaml :68-78): other widths return CV_ERR_UNSUPPORTED and the caller keeps the cv_gemm / cv_layernorm path.
 * ------------------------------------------------------------------------------------------ */
typedef struct cv_tblock_params {
  int32_t dtype;              /* CV_BF16 / CV_F16: element type of qk, vt, ao, out_act and the packed weights */
  int32_t R, T;               /* sequences, frames per sequence */
  int32_t C, inner, ff;       /* 256, 512, 1024 */
  float* x; int32_t ldx;      /* [R][T][ldx] fp32; read by head, updated in place by tail */
  float eps;                  /* LayerNorm eps (both norms) */
  /* head */
  const float* g1; const float* b1n;      /* norm1 weight / bias [C] */
  const void* wqkv_p;                     /* packed [Wq; Wk; Wv] (3 * inner rows, K = C), no bias */
  void* qk; int32_t ldqk;                 /* [R * T][ldqk]; 16-byte aligned, ldqk a multiple of 8 (rows are written with 16-byte stores) */
  void* vt; int32_t vt_ld;                /* [R][inner / 64][64][vt_ld] */
  /* tail */
  const void* ao; int32_t ldao;           /* attention output [R * T][ldao] or NULL */
  const void* wo_p; const float* bo;      /* packed to_out.0 (C rows, K = inner), bias [C] */
  const float* g3; const float* b3n;      /* norm3 weight / bias [C] */
  const void* w1_p; const float* bf1;     /* packed ff.net.0.proj (ff rows, K = C), bias [ff] */
  const void* w2_p; const float* bf2;     /* packed ff.net.2 (C rows, K = ff), bias [C] */
  void* out_act; int32_t ldoa;            /* optional 16-bit copy of the block output [R * T][ldoa] */
  int32_t cus;                            /* CUs the launch may use (a CU-masked stream's share); 0 = the whole chip.  Speed only:
                                             picks 48- or 64-row tiles so that the grid does not end in a nearly empty round */
} cv_tblock_params;
int cv_sizeof_tblock_params(void);
int cv_tblock_head(const cv_tblock_params* p, void* stream);
int cv_tblock_tail(const cv_tblock_params* p, void* stream);
/* tail of transformer block i followed, in the same launch and on the same rows, by the head of block i + 1: the tail fields describe
 * block i (ao / wo_p / bo required, out_act must be NULL), the head fields (g1, b1n, wqkv_p, qk, vt) block i + 1.  Same results as
 * cv_tblock_tail then cv_tblock_head up to the summation order of the fp32 LayerNorm statistics; x is written once and not re-read. */
int cv_tblock_tail_head(const cv_tblock_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * cv_resblock_conv1 / cv_resblock_conv2 — CausalResnetBlock1D of the CFM estimator (flow/decoder.py:36-56 CausalBlock1D /
 * CausalResnetBlock1D, flow/components/decoder.py:54-59 ResnetBlock1D.forward; CausalConv1d :59-85 = left padding k - 1) as
 * two row-block kernels (same structure as cv_tblock_*):
 *   conv1: h1  = Mish(LayerNorm_C(conv_k3(a) + b1)) + tadd            tadd [C] = mlp(mish(time embedding)) of this step / block
 *   conv2: out = Mish(LayerNorm_C(conv_k3(h1) + b2)) + conv_1x1(a) + br
 * a [R][T][lda] 16-bit channels-last (first cin channels), h1 [R][T][ldh1] 16-bit, out [R][T][ldo] fp32.  Weights packed by
 * cv_pack_skinny from [C][K] matrices with k = tap * cin + ci (K zero-padded to a multiple of 128).  C = 256, cin in
 * {256, 320, 512}; anything else returns CV_ERR_UNSUPPORTED (the caller keeps the cv_gemm / cv_layernorm launches).
 * ------------------------------------------------------------------------------------------ */
typedef struct cv_resblock_params {
  int32_t dtype, R, T, C, cin;
  const void* a; int32_t lda;
  const void* w1_p; const float* b1; const float* g1; const float* be1; const float* tadd;
  void* h1; int32_t ldh1;
  const void* w2_p; const float* b2; const float* g2; const float* be2;
  const void* wr_p; const float* br;
  float* out; int32_t ldo;
  float eps; int32_t cus;
} cv_resblock_params;
int cv_sizeof_resblock_params(void);
int cv_resblock_conv1(const cv_resblock_params* p, void* stream);
int cv_resblock_conv2(const cv_resblock_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout / elementwise helpers (HBM-bound, coalesced, fp32 math).
 * ------------------------------------------------------------------------------------------ */
/* x [B][C][T] fp32 (the reference's channel-first tensors) -> out [B][T][ldo] of `dtype` (channels-last; columns
 * [C,ldo) zero-filled).  Replaces the implicit layout of torch Conv1d inputs (generator.py:353, f0_predictor.py:53). */
int cv_to_channels_last(const float* x, void* out, int32_t dtype, int32_t B, int32_t C, int32_t T, int32_t ldo, void* stream);
/* x [B][T][ldx] fp32 -> out [B][C][T] fp32 */
int cv_to_channels_first(const float* x, float* out, int32_t B, int32_t C, int32_t T, int32_t ldx, void* stream);
/* Snake x + sin^2(a x)/(a+1e-9) (transformer/activation.py:73-84) of x [rows][C] fp32 with up to 4 alpha vectors,
 * one `dtype` output per alpha: the first activation of the parallel ResBlocks (generator.py:366-372).
 * dtype CV_F32X3 = fp32-sized outputs in the pre-split chunk format of cv_gemm_params.x3_flags. */
int cv_snake_multi(const float* x, int32_t rows, int32_t C, int32_t ldx, int32_t n, const float* const* alpha,
                   void* const* out, int32_t ldo, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * HiFT source / (i)STFT kernels, n_fft 16, hop 4, periodic hann, center=True (generator.py:333-347).
 * ------------------------------------------------------------------------------------------ */
/* s [B][S] fp32 -> out [B][S/4+1][ldo] `dtype`: columns 0..8 real, 9..17 imag, [18,ldo) zero (torch.stft, reflect pad) */
int cv_stft16(const float* s, void* out, int32_t dtype, int32_t B, int32_t S, int32_t ldo, void* stream);
/* y [B][F][ldy] fp32 (conv_post output: 9 log-magnitudes, 9 phase pre-activations) -> wav [B][(F-1)*4]:
 * mag = min(exp(y),100), phase = sin(y), iSTFT with window-envelope normalisation, clamp +-audio_limit (generator.py:376-381) */
int cv_istft16(const float* y, float* wav, int32_t B, int32_t F, int32_t ldy, float audio_limit, void* stream);
/* SourceModuleHnNSF (generator.py:137-220): f0 [B][T] (Hz, >=0) -> s [B][T*up]; harmonics nh (<=16);
 * phase_vec [B][nh] (row 0 ignored = 0), noise [B][nh][T*up] standard normal, lin_w [nh], lin_b [1];
 * work [B][nh][T] doubles (frame-start phases).  The phase scan is done per frame in fp64 (exact value of the
 * reference's order-dependent fp32 cumsum). */
int cv_hift_source(const float* f0, const float* phase_vec, const float* noise, const float* lin_w, const float* lin_b,
                   double* work, float* s, int32_t B, int32_t T, int32_t up, int32_t nh, float sampling_rate,
                   float sine_amp, float noise_std, float voiced_threshold, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flow-matching helpers (flow/flow.py:286-317, flow/flow_matching.py:72-124).
 * ------------------------------------------------------------------------------------------ */
/* out[r][0:dim] = table[idx[r]][0:dim]; idx == -1 -> zero row; idx == -2 -> row left untouched; out dtype `dtype`, ld ldo */
int cv_embedding(const float* table, const int32_t* idx, void* out, int32_t dtype, int32_t rows, int32_t dim, int32_t ldo, void* stream);
/* Estimator input for classifier-free guidance, channels-last: for utterance b, row pair (2b, 2b+1):
 *   xin[2b]   = [x | mu | spks (broadcast over T) | cond]   xin[2b+1] = [x | 0 | 0 | 0]      (flow_matching.py:95-108,
 *   flow/decoder.py:243-249).  x, mu, cond [B][T][C] fp32, spks [B][C] fp32 -> xin [2B][T][4C] `dtype`. */
int cv_est_pack(const float* x, const float* mu, const float* spks, const float* cond, void* xin, int32_t dtype,
                int32_t B, int32_t T, int32_t C, void* stream);
/* Euler step with CFG: x += dt * ((1+w) * v[2b] - w * v[2b+1]);  v [2B][T][C] fp32 (flow_matching.py:116-118) */
int cv_cfm_update(float* x, const float* v, int32_t B, int32_t T, int32_t C, float dt, float cfg_rate, void* stream);
/* F.interpolate(mode="linear", align_corners=False) along T, channels-last: x [T_in][ldx] fp32 -> y [T_out][ldy] `dtype`
 * (InterpolateRegulator.inference, flow/length_regulator.py:49-70). */
int cv_interp_linear_cl(const float* x, int32_t ldx, int32_t T_in, void* y, int32_t ldy, int32_t dtype, int32_t T_out,
                        int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Prompt-feature front half: mel_spectrogram (cosyvoice/dataset/processor_kaldidata.py:37-74; 24 kHz: n_fft = win 1920,
 * hop 480, 80 mels, center=False with (n_fft-hop)/2 reflect padding).  The windowed DFT and the mel projection are cv_gemm
 * calls (frames are a strided view of the padded signal: lda = hop); these are the two element-wise steps.
 * ------------------------------------------------------------------------------------------ */
/* spec [rows][ld_spec] = [re(nbins) | im(nbins) | pad] -> mag [rows][ld_mag] = sqrt(re^2 + im^2 + eps) (:66), columns >= nbins zeroed */
int cv_stft_magnitude(const float* spec, int32_t ld_spec, float* mag, int32_t ld_mag, int32_t rows, int32_t nbins, float eps, void* stream);
/* mel [B][T][ld] -> out [B][n_mels][T] = log(max(mel, clip)) (dynamic_range_compression_torch :27-28) */
int cv_log_clamp_channels_first(const float* mel, int32_t ld, float* out, int32_t B, int32_t T, int32_t n_mels, float clip, void* stream);

/* ------------------------------------------------------------------------------------------
 * hipGraph capture of a launch sequence issued through this ABI on `stream` (the reference's counterpart is
 * torch.cuda.CUDAGraph capture in llm/qwen2_5.py:97-124).  begin/end bracket the launches; launch replays them.
 * ------------------------------------------------------------------------------------------ */
int cv_graph_begin(void* stream);
int cv_graph_end(void* stream, void** graph_out);
int cv_graph_launch(void* graph, void* stream);
/* The same launch sequence issued one by one (hipLaunchKernel per captured node) instead of as a hipGraphExec replay:
 * hipGraph replays ignore the CU mask of the stream they run on, direct launches honour it.  Used to give the decode loop
 * (the reference's LLM thread, cli/model.py:119) and the flow/vocoder stages (the caller's thread) disjoint CU sets.
 * cv_graph_num_launches: length of that list, or CV_ERR_UNSUPPORTED when the capture holds a node kind it cannot replay. */
int cv_graph_launch_direct(void* graph, void* stream);
int cv_graph_num_launches(void* graph);
int cv_graph_destroy(void* graph);
/* Stream restricted to the CUs of `mask` (nwords x 32 bits; bit i = CU slot i / 8 of XCD i % 8 on an 8-XCD part). */
int cv_stream_create_cumask(const uint32_t* mask, int32_t nwords, void** stream_out);
int cv_stream_destroy(void* stream);

/* ------------------------------------------------------------------------------------------
 * LLM decode-step kernels (Qwen2 backbone; HF Qwen2ForCausalLM called from llm/llm.py:754-766; the reference's own
 * CUDA-graph decode path llm/qwen2_5.py:97-179,265-320 is the behavioural spec of the captured token loop).
 * ------------------------------------------------------------------------------------------ */
/* Skinny GEMM for M <= 16 rows (decode batch): out[m][n] = sum_k A[m][k] * W[n][k], weights pre-packed by
 * cv_pack_skinny into MFMA 16x16x32 B-fragments [N/16][K/32][64 lanes][8] so every wave-load is 1 KiB contiguous
 * (HBM-bound weight streaming).  A: [16][lda] 16-bit (rows >= M must be finite, e.g. zero).
 * mode 0: out_f32[ks][m][n] (+bias on slice 0), ksplit slabs of slab_stride floats;
 * mode 1: out_f32[m][n] += acc + bias (in-place residual, ksplit must be 1);
 * mode 2: SwiGLU: tiles alternate [gate16 | up16]; out_act[m][16*pair + i] = silu(g) * u (ksplit must be 1);
 * mode 3: out_act[m][n] = T(relu(acc + bias)) (ksplit must be 1; the v1 TransformerEncoderLayer's FFN, encoder_layer.py:24-115). */
typedef struct cv_skinny_params {
  int32_t dtype, M, N, K;
  const void* A; int32_t lda;
  const void* Wp;
  const float* bias;
  int32_t ksplit, mode;
  float* out_f32; int32_t ldo; int64_t slab_stride;
  void* out_act; int32_t ldoa;
  /* optional fused prologue (ngamma != NULL; A is then ignored): v[m][:] = nx[m][:] + sum_s nslabs[s][m][:];
     A[m][:] = T(rmsnorm(v[m]) * ngamma); workgroup (0,0) also stores v to nx_out (must not alias nx). K <= 1024. */
  const float* nx; int32_t ldnx;
  const float* nslabs; int32_t n_nslab; int64_t nslab_stride; int32_t ld_nslab;
  const float* ngamma; float neps;
  float* nx_out;
  /* > 0: cap on the workgroups launched (x ksplit slices); each then walks several tile groups with the next group's
     weights prefetched.  Set to about 2 x the CUs the calling stream owns; 0 = one workgroup per tile group. */
  int32_t max_wgs;
  /* RMSNorm split across two launches (the decode step's o_proj -> gate/up pair), so that the consumer needs no prologue:
     producer (mode 1): xb_out != NULL also stores the updated rows as `dtype` (un-normalised) and ss_part[blockIdx.x][16] =
     this workgroup's partial sum of squares of each row (fixed order: deterministic, no atomics);
     consumer (mode 2, A = those 16-bit rows, gamma folded into the packed weights): rs_part != NULL scales gate and up by
     rsqrt(sum_i rs_part[i][m] / K + rs_eps) in the epilogue. */
  void* xb_out; int32_t ldxb; float* ss_part;
  const float* rs_part; int32_t n_rs_part; float rs_eps;
} cv_skinny_params;
int cv_skinny_gemm(const cv_skinny_params* p, void* stream);
/* CosyVoice-v1 TransformerLM cached decode step (TransformerEncoder.forward_chunk's att_cache, transformer/encoder.py:185-274):
 * the new token's fp32 projection row qkv = [q + pos_bias_u | q + pos_bias_v | k | v] (4 D) -> qq [2 D] `dtype`, K cache row t
 * (kcache [ctx][D]) and V^T cache column t (vtcache [D][vt_ld], row = head * 64 + channel). */
int cv_relpos_append(const float* qkv, void* qq, void* kcache, void* vtcache, int32_t dtype, int32_t D, int32_t t,
                     int32_t vt_ld, void* stream);
/* W [N][K] row-major 16-bit (device) -> packed (device), N padded up to a multiple of 16 with zeros.
 * interleave != 0: rows are taken as [gate(N/2) ; up(N/2)] and emitted as alternating 16-row blocks. */
int cv_pack_skinny(const void* W, void* Wp, int32_t N, int32_t K, int32_t interleave, void* stream);

/* x[b][:] += sum_s slab[s][b][:] (nslab may be 0); xn[b][:] = T(rmsnorm(x[b]) * gamma).  One workgroup per row. */
int cv_rmsnorm_reduce(float* x, int32_t ldx, const float* slabs, int32_t nslab, int64_t slab_stride, int32_t ld_slab,
                      const float* gamma, float eps, void* xn, int32_t ldxn, int32_t dtype, int32_t rows, int32_t dim, void* stream);

/* RoPE (HF rotate_half convention) + KV-cache append.  qkv fp32 [rows][ldqkv] = [q (Hq*64) | k (Hkv*64) | v (Hkv*64)];
 * row r belongs to sequence b = r / rows_per_seq at position pos_base[b] + r % rows_per_seq.
 * q_out [rows][ldq] 16-bit; kcache [B][Hkv][ctx_max][64]; vtcache [B][Hkv][64][ctx_max] (16-bit). */
int cv_rope_append(const float* qkv, int32_t ldqkv, const int32_t* pos_base, int32_t rows, int32_t rows_per_seq,
                   int32_t Hq, int32_t Hkv, const float* inv_freq, void* q_out, int32_t ldq, void* kcache, void* vtcache,
                   int32_t ctx_max, int32_t dtype, void* stream);

/* Single-query GQA attention over the KV cache: for sequence b, kv head hk, the Hq/Hkv query heads of the group
 * form the (<=16) columns of one MFMA tile; 8 waves split the keys and merge by log-sum-exp.
 * q [B][ldq] 16-bit (row b), ctx_len[b] + ctx_add keys valid; out [B][ldo] 16-bit.
 * Fused form (qkv != NULL): q is ignored; the kernel first applies RoPE to row b of qkv fp32 [B][ldqkv]
 * (= [q | k | v], position ctx_len[b]; inv_freq is then the table rope[ctx_max][64] = [cos 32 | sin 32] per position),
 * appends K / V^T of its kv head to the caches, then attends (ctx_add = 1).  In the fused form BOTH caches are
 * FRAGMENT-TILED (cv_kv_retile below): per (sequence, kv head) ctx_max/64 tiles of 64 keys x 64 d in MFMA operand order, so every
 * fragment load is 64 lanes x 16 contiguous bytes; the un-fused form reads the row-major caches cv_rope_append writes. */
int cv_decode_attention(const void* q, int32_t ldq, const void* kcache, const void* vtcache, const int32_t* ctx_len,
                        int32_t ctx_add, void* out, int32_t ldo, int32_t B, int32_t Hq, int32_t Hkv, int32_t ctx_max,
                        float scale, int32_t dtype, const float* qkv, int32_t ldqkv, const float* inv_freq, void* stream);

/* Row-major KV caches (cv_rope_append's layout, also what cv_attention reads during prefill) -> the fragment-tiled caches
 * of the fused cv_decode_attention, keys [0, n_keys) rounded up to whole 64-key tiles.  Same buffer sizes
 * ([B][Hkv][ctx_max*64] 16-bit elements each); ctx_max % 64 == 0.  Tile layout (element index inside a 4096-element tile):
 *   K:   ((kt*2 + f)*64 + lq + 16*lg)*8 + e   = key 16*kt + lq,                         d = 32*f + 8*lg + e
 *   V^T: ((dt*2 + s2)*64 + lq + 16*lg)*8 + e  = key 32*s2 + 16*(e >> 2) + 4*lg + (e & 3), d = 16*dt + lq */
int cv_kv_retile(const void* k_rowmajor, const void* vt_rowmajor, void* k_tiled, void* vt_tiled, int32_t B, int32_t Hkv,
                 int32_t ctx_max, int32_t n_keys, void* stream);

/* Repetition-aware sampling on device (utils/common.py:109-146 ras_sampling/nucleus_sampling/random_sampling,
 * llm/llm.py:806-821 sampling_ids, :861-874 loop bookkeeping).  One workgroup per sequence. */
typedef struct cv_sample_params {
  const float* logits; int32_t ldl; int32_t V; int32_t B;
  int32_t eos; int32_t top_k; float top_p; int32_t win_size; float tau_r;
  uint64_t seed;
  /* repetition fallback: 0 = random_sampling over the full distribution (ras_sampling, utils/common.py:106-112);
     1 = a second nucleus draw with (top_p2, top_k2) (non_random_ras_sampling :116-123: top_p + 0.15, top_k * expand_scale) */
  int32_t fallback_mode; float top_p2; int32_t top_k2;
  const float* uniforms;      /* optional [B][max_trials+1][2] injected uniforms (tests); null -> Philox */
  int32_t max_trials;
  const int32_t* min_len; const int32_t* max_len;   /* [B] */
  const int32_t* forced; int32_t forced_ld;         /* [B][forced_ld] teacher-forced emitted ids, -1 = none; or null */
  int32_t* step;              /* [B] loop index i of llm.py:861 */
  int32_t* pos;               /* [B] KV length, +1 per step */
  int32_t* n_emitted;         /* [B] */
  int32_t* finished;          /* [B] 0 running, 1 eos, 2 max_len, 3 sampling stalled (max_trials) */
  int32_t* out_tokens; int32_t out_ld;              /* [B][out_ld] */
  const float* emb_table; int32_t emb_dim;          /* speech_embedding.weight fp32 */
  float* x; int32_t ldx;      /* next-step input embedding [B][ldx] */
  /* optional per-request nonce (device, one uint64): XORed into the Philox key, so a captured step graph (seed is baked into
     its kernel arguments) still draws a fresh stream for every request — the reference draws from torch's global RNG
     (utils/common.py:139).  The counter stays (step, row, trial). */
  const uint64_t* nonce;
} cv_sample_params;
int cv_sample_ras(const cv_sample_params* p, void* stream);
/* ------------------------------------------------------------------------------------------
 * Stage-level entry points: a whole stage step behind one call, so that the composition of a stage does not
 * live only in this package's Python classes.
 * cv_llm_step_*: one decode step of the Qwen2 backbone for B <= 16 sequences — 24 x [RMSNorm + QKV -> RoPE + KV append +
 * GQA attention -> o_proj (+ residual) -> RMSNorm -> gate/up + SwiGLU -> down (+ residual)] -> final norm -> llm_decoder ->
 * repetition-aware sampling, loop bookkeeping and the next input embedding; positions / EOS state live on the device.
 * Replaces the reference's graph decode path (llm/qwen2_5.py:97-179,265-320: four CUDA graphs per layer replayed from Python,
 * positions on the host; llm/qwen2_infer.py:50-105) and the loop body of llm/llm.py:861-874.
 * All pointers are caller-owned device buffers; `layers` is a host array read at enqueue / capture time only.
 * cv_llm_step_enqueue issues the launches on `stream`; cv_llm_step_graph_create captures them (on `capture_stream`, which must
 * not be the default stream) into a graph handle that cv_llm_step_graph_launch / cv_graph_launch_direct replay.
 * ------------------------------------------------------------------------------------------ */
typedef struct cv_llm_layer {
  const void* p_qkv; const float* bqkv;   /* cv_pack_skinny([Wq; Wk; Wv]) (K = hidden), bias [q_dim + 2 kv_dim] fp32 */
  const void* p_o;                        /* cv_pack_skinny(o_proj) (hidden rows, K = q_dim) */
  const void* p_gu;                       /* cv_pack_skinny([gate; up] * post_attention_layernorm.weight, interleave = 1) */
  const void* p_down;                     /* cv_pack_skinny(down_proj) (hidden rows, K = intermediate) */
  const float* g_in;                      /* input_layernorm.weight [hidden] */
  void* kcache; void* vtcache;            /* [B_max][Hkv][ctx_max*64] each, FRAGMENT-TILED (cv_kv_retile) */
} cv_llm_layer;
typedef struct cv_llm_step_desc {
  int32_t dtype, B, num_layers, hidden, num_heads, num_kv_heads, inter, ctx_max, down_ksplit; float rms_eps;
  int32_t split_qkv_norm;                 /* != 0: input RMSNorm (+ slab reduce) as its own cv_rmsnorm_reduce launch, residual updated in place in x
                                             (pays off beyond 8 rows); 0: fused into the QKV kernel's prologue, residual ping-pongs x / x2 */
  int32_t reserved;
  const cv_llm_layer* layers;             /* host array [num_layers] */
  /* state rows: 16 for B <= 16, 32 for B <= 32 (two MFMA row groups share every weight stream; needs split_qkv_norm);
     slabs / ssp are pitched by that row count */
  float* x; float* x2;                    /* [rows][hidden] fp32: input embedding of the step / residual ping-pong */
  void* xn; void* xb;                     /* [16][hidden] 16-bit scratch rows */
  float* ssp; int32_t n_ssp;              /* [hidden / 16][16] partial sums of squares */
  float* qkv;                             /* [16][q_dim + 2 kv_dim] fp32 */
  void* ao; void* h;                      /* [16][q_dim], [16][inter] 16-bit */
  float* slabs;                           /* [down_ksplit][16][hidden] fp32 */
  float* logits; int32_t vpad;            /* [16][vpad] fp32 */
  const float* rope_table;                /* [ctx_max][cos 32 | sin 32] */
  const float* g_final; const void* p_dec; const float* dec_b; int32_t out_vocab;   /* model.norm, llm_decoder (packed), bias */
  cv_sample_params sample;                /* sampler configuration + device state; logits / ldl / V / B / x / ldx / emb_dim are filled in */
} cv_llm_step_desc;
int cv_sizeof_llm_step_desc(void);
int cv_sizeof_llm_layer(void);
int cv_llm_step_enqueue(const cv_llm_step_desc* d, void* stream);
int cv_llm_step_graph_create(const cv_llm_step_desc* d, void* capture_stream, void** graph_out);
int cv_llm_step_graph_launch(void* graph, void* stream);
int cv_llm_step_graph_destroy(void* graph);

/* cv_flow_euler_*: the whole flow-matching solve behind one call — n_steps x [cv_est_pack -> CFM estimator ->
 * cv_cfm_update] for B utterances (2B CFG rows), T frames.  Replaces ConditionalCFM.solve_euler
 * (flow/flow_matching.py:72-124) and ConditionalDecoder.forward (flow/decoder.py:224-334: down block, 12 mid blocks, up block,
 * final block, final_proj), each block = causal resnet (cv_resblock_conv1/2) + n_tb transformer blocks
 * (cv_tblock_head -> cv_attention -> cv_tblock_tail); the down / up slots and the final block are causal k3 convs
 * (cv_gemm with taps) + LayerNorm + Mish (cv_layernorm) + final_proj (cv_gemm).
 * `blocks`, every `tb` array and `dts` are HOST arrays read at enqueue / capture time only; all other pointers are caller-owned
 * device buffers.  Weights: *_p = cv_pack_skinny fragment order (resnet conv K zero-padded to a multiple of 128 per tap group as
 * cv_resblock_params describes); down_w / up_w / fin_w [C][3*C] and proj_w [out_ch][C] row-major 16-bit (k = tap*C + ci).
 * tadd[step][block*C + c] = the block's time-MLP output for that step (fp32, depends only on t: computed once per schedule). */
typedef struct cv_flow_resnet {
  const void* w1_p; const float* b1; const float* g1; const float* be1;
  const void* w2_p; const float* b2; const float* g2; const float* be2;
  const void* wr_p; const float* br;
  int32_t cin; int32_t reserved;
} cv_flow_resnet;
typedef struct cv_flow_tblock {
  const float* g1; const float* b1n; const void* wqkv_p;
  const void* wo_p; const float* bo; const float* g3; const float* b3n;
  const void* w1_p; const float* bf1; const void* w2_p; const float* bf2;
} cv_flow_tblock;
typedef struct cv_flow_block { cv_flow_resnet res; const cv_flow_tblock* tb; int32_t n_tb;
  int32_t fuse_tail_head;   /* != 0: blocks j < n_tb - 1 run as cv_tblock_tail_head (tail of j + head of j + 1 in one launch) */
} cv_flow_block;
typedef struct cv_flow_solver_desc {
  int32_t dtype, B, T, Tp;                 /* Tp = V^T row pitch (T rounded up to 8) */
  int32_t C, inner, ff, heads, in_ch, out_ch;   /* 256, 512, 1024, 8, 320, 80 */
  int32_t n_blocks, n_steps, cus; float cfg_rate, eps;
  const cv_flow_block* blocks;             /* host [n_blocks]: down, mid..., up */
  const void* down_w; const float* down_b; const void* up_w; const float* up_b;
  const void* fin_w; const float* fin_b; const float* fin_g; const float* fin_be; const void* proj_w; const float* proj_b;
  const float* tadd;                       /* device [n_steps][n_blocks * C] */
  const float* dts;                        /* host [n_steps] */
  float* x; const float* mu; const float* spks; const float* cond;   /* [B][T][out_ch] state (in place), mu, cond; spks [B][out_ch] */
  const int32_t* klen;                     /* [2B] valid frames per CFG row, or NULL */
  /* workspace for R = 2B rows: xin [R][T][in_ch], h1 [R][T][C], qk [R][T][2*inner], vt [R][heads][64][Tp] (zero-initialised),
   * ao [R][T][inner], cat [R][T][2C], d [R][T][C] (all `dtype`); x32, c32a [R][T][C] fp32; v [R][T][out_ch] fp32 */
  void* xin; void* h1; float* x32; void* qk; void* vt; void* ao; void* cat; void* d; float* v; float* c32a;
} cv_flow_solver_desc;
int cv_sizeof_flow_solver_desc(void);
int cv_sizeof_flow_block(void);
int cv_sizeof_flow_tblock(void);
int cv_flow_euler_enqueue(const cv_flow_solver_desc* d, void* stream);
int cv_flow_euler_graph_create(const cv_flow_solver_desc* d, void* capture_stream, void** graph_out);
int cv_flow_euler_graph_launch(void* graph, void* stream);
int cv_flow_euler_graph_destroy(void* graph);

/* cv_hift_decode_*: HiFTGenerator.decode (hifigan/generator.py:349-381) behind one call — source STFT -> conv_pre ->
 * n_stages x [source_downs + source_resblocks, ups (ConvTranspose1d as u phase GEMMs, reflect pad on the last stage) + source
 * fusion, num_kernels parallel ResBlocks (Snake activations, mean) + leaky-relu] -> conv_post -> exp / sin -> iSTFT -> clamp.
 * Every conv is a cv_gemm launch over channels-last tensors with the activation / residual / mean fused into its epilogue.
 * `dtype` = element type of the activation tensors and weights (CV_F32 for the reference-exact vocoder); `gemm_dtype` = what
 * cv_gemm multiplies in (CV_F32 exact, CV_F32X3 = bf16 hi/lo split products, or the 16-bit dtype).
 * `stages`, every `units` / `rbs` / `phases` / `xa` array are HOST arrays read at enqueue / capture time only. */
typedef struct cv_hift_conv {        /* channels-last Conv1d: w [cout][k*cin] with k index = tap*cin + ci; cin = channel pitch of its input */
  const void* w; const float* b; int32_t k, cin, cout, dilation, pad_left, stride;
  int32_t x3_flags, reserved;        /* gemm_dtype CV_F32X3: cv_gemm_params.x3_flags of this conv (bit 2 is dropped when it has no 16/32-bit activation output) */
} cv_hift_conv;
typedef struct cv_hift_resunit { cv_hift_conv c1, c2; const float* a1; const float* a2; } cv_hift_resunit;   /* Snake alphas [C] */
typedef struct cv_hift_resblock { const cv_hift_resunit* units; int32_t n_units, reserved; } cv_hift_resblock;
typedef struct cv_hift_phase { const void* w; int32_t ntaps, tap_base; } cv_hift_phase;   /* w [c][ntaps*up_cin] */
typedef struct cv_hift_stage {
  const cv_hift_phase* phases; const float* up_b; int32_t u, up_cin, up_flags, reserved;   /* up_flags: x3_flags of the phase GEMMs */
  cv_hift_conv source_down; cv_hift_resblock source_rb;
  const cv_hift_resblock* rbs;       /* [n_kernels] */
  int32_t t_out, c;                  /* frames and channels after this stage */
  /* workspace, all [B][t_out][c]: x32, r0/r1, acc0/acc1, si0/si1 fp32; xa[n_kernels], ta, ra, out `dtype` */
  float* x32; void* const* xa; float* r0; float* r1; void* ta; void* ra; float* acc0; float* acc1; float* si0; float* si1; void* out;
} cv_hift_stage;
typedef struct cv_hift_decode_desc {
  int32_t dtype, gemm_dtype, B, T, S, n_stages, n_kernels, stft_ld, hop;
  int32_t presplit;                  /* != 0 (gemm_dtype CV_F32X3): activation tensors are kept in the pre-split chunk format (cv_snake_multi writes it too) */
  float lrelu_slope, audio_limit;
  cv_hift_conv conv_pre, conv_post;
  const cv_hift_stage* stages;
  const void* mel_cl;                /* [B][T][conv_pre.cin] `dtype` (cv_to_channels_last of the mel) */
  const float* s;                    /* [B][S] source signal */
  void* stft;                        /* [B][S/hop + 1][stft_ld] `dtype` */
  void* a_pre;                       /* [B][T][conv_pre.cout] `dtype` */
  float* post;                       /* [B][t_last][stft_ld] fp32 */
  float* wav;                        /* [B][(t_last - 1) * hop] fp32 output */
} cv_hift_decode_desc;
int cv_sizeof_hift_decode_desc(void);
int cv_sizeof_hift_stage(void);
int cv_sizeof_hift_resunit(void);
int cv_hift_decode_enqueue(const cv_hift_decode_desc* d, void* stream);
int cv_hift_decode(const cv_hift_decode_desc* d, void* stream);   /* = cv_hift_decode_enqueue (the name SURVEY.md §8b lists) */
int cv_hift_decode_graph_create(const cv_hift_decode_desc* d, void* capture_stream, void** graph_out);

int cv_sizeof_gemm_params(void);
int cv_sizeof_norm_params(void);
int cv_sizeof_attn_params(void);
int cv_sizeof_skinny_params(void);
int cv_sizeof_sample_params(void);

/* ------------------------------------------------------------------------------------------
 * BigVGAN fused anti-aliased activation — the reference's only native kernel
 * (cosyvoice/BigVGAN/alias_free_activation/cuda/anti_alias_activation_cuda.cu:44-181, binding
 * anti_alias_activation.cpp:19-23, wrapper cuda/activation1d.py:13-76; semantics = torch path
 * alias_free_activation/torch/{act,resample,filter}.py): per channel, replicate-pad(5) -> 2x upsample with the 12-tap
 * kaiser-sinc filter (x2 gain) -> SnakeBeta x + sin^2(x e^a)/(e^b + 1e-9) -> replicate-pad(5,6) -> 12-tap low-pass,
 * stride 2.  x, y [B][C][T] of `dtype` (f32 / bf16 / f16), filters and log-scale alpha/beta fp32, fp32 accumulation.
 * ------------------------------------------------------------------------------------------ */
int cv_anti_alias_act(const void* x, void* y, int32_t dtype, int32_t B, int32_t C, int32_t T, const float* up_filter,
                      const float* down_filter, const float* alpha_log, const float* beta_log, void* stream);
/* The same activation on channels-last tensors x [B][T][ldx] -> y [B][T][ldy] (first C channels), input / output element
 * types chosen independently (fp32 residual stream in, 16-bit or fp32 conv operand out): the form the AMP blocks of
 * BigVGAN.forward (BigVGAN/bigvgan.py:128-137,384-438) use between the channels-last conv GEMMs of this library. */
int cv_anti_alias_act_cl(const void* x, int32_t ldx, int32_t in_dtype, void* y, int32_t ldy, int32_t out_dtype, int32_t B,
                         int32_t T, int32_t C, const float* up_filter, const float* down_filter, const float* alpha_log,
                         const float* beta_log, void* stream);

#ifdef __cplusplus
}
#endif
#endif
