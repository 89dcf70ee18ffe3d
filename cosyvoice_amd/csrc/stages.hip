// Stage-level entry points: the composition of a whole stage step behind ONE C call, so that any host (not only this
// package's Python classes) can drive it through the ABI.  cv_llm_step_*: the Qwen2 decode step — per layer
//   skinny QKV (+ RMSNorm prologue, + split-K slab reduce of the previous layer's down projection) -> fused RoPE + KV append +
//   single-query GQA attention -> skinny o_proj (in-place residual; also emits the 16-bit rows + partial sums of squares of the
//   split post-attention RMSNorm) -> skinny gate/up + SwiGLU (1/rms in the epilogue) -> skinny down (split-K slabs),
// then final norm (+ slab reduce) -> speech-token head -> on-device sampling + bookkeeping + next-input embedding.
// Behavioural spec: the reference's graph decode path, /root/reference/cosyvoice/llm/qwen2_5.py:97-179,265-320 (4 CUDA graphs per
// layer replayed from Python with host-side positions) and llm/llm.py:861-874 (the loop body); here the step is one launch
// sequence with device-side position / EOS state, captured into one hipGraph by cv_llm_step_graph_create.
#include "cv_device.h"

namespace {
int enqueue_step(const cv_llm_step_desc& d, hipStream_t st) {
  const int H = d.hidden, I = d.inter, KS = d.down_ksplit;
  const int q_dim = d.num_heads * 64, kv_dim = d.num_kv_heads * 64, qkv_dim = q_dim + 2 * kv_dim;
  const float scale = 0.125f;   // 1 / sqrt(64)
  float* cur = d.x;
  float* nxt = d.x2;
  for (int li = 0; li < d.num_layers; ++li) {
    const cv_llm_layer& L = d.layers[li];
    {   // RMSNorm(input_layernorm) of (residual + previous down-projection slabs) -> QKV; workgroup (0,0) stores the summed residual
      cv_skinny_params p{};
      p.dtype = d.dtype; p.M = d.B; p.N = qkv_dim; p.K = H;
      p.A = d.xn; p.lda = H; p.Wp = L.p_qkv; p.bias = L.bqkv; p.ksplit = 1; p.mode = 0;
      p.out_f32 = d.qkv; p.ldo = qkv_dim;
      p.nx = cur; p.ldnx = H; p.ngamma = L.g_in; p.neps = d.rms_eps; p.nx_out = nxt;
      if (li > 0) { p.nslabs = d.slabs; p.n_nslab = KS; p.nslab_stride = 16 * (int64_t)H; p.ld_nslab = H; }
      if (int rc = cv_skinny_gemm(&p, st)) return rc;
    }
    if (int rc = cv_decode_attention(d.xn, H, L.kcache, L.vtcache, d.sample.pos, 1, d.ao, q_dim, d.B, d.num_heads, d.num_kv_heads,
                                     d.ctx_max, scale, d.dtype, d.qkv, qkv_dim, d.rope_table, st)) return rc;
    {   // o_proj, in-place residual; producer half of the split post-attention RMSNorm
      cv_skinny_params p{};
      p.dtype = d.dtype; p.M = d.B; p.N = H; p.K = q_dim;
      p.A = d.ao; p.lda = q_dim; p.Wp = L.p_o; p.ksplit = 1; p.mode = 1;
      p.out_f32 = nxt; p.ldo = H;
      p.xb_out = d.xb; p.ldxb = H; p.ss_part = d.ssp;
      if (int rc = cv_skinny_gemm(&p, st)) return rc;
    }
    {   // gate/up + SwiGLU, consumer half (gamma folded into the packed weights)
      cv_skinny_params p{};
      p.dtype = d.dtype; p.M = d.B; p.N = 2 * I; p.K = H;
      p.A = d.xb; p.lda = H; p.Wp = L.p_gu; p.ksplit = 1; p.mode = 2;
      p.out_act = d.h; p.ldoa = I;
      p.rs_part = d.ssp; p.n_rs_part = d.n_ssp; p.rs_eps = d.rms_eps;
      if (int rc = cv_skinny_gemm(&p, st)) return rc;
    }
    {   // down projection as split-K slabs (summed by the next layer's prologue / the final norm)
      cv_skinny_params p{};
      p.dtype = d.dtype; p.M = d.B; p.N = H; p.K = I;
      p.A = d.h; p.lda = I; p.Wp = L.p_down; p.ksplit = KS; p.mode = 0;
      p.out_f32 = d.slabs; p.ldo = H; p.slab_stride = 16 * (int64_t)H;
      if (int rc = cv_skinny_gemm(&p, st)) return rc;
    }
    float* t = cur; cur = nxt; nxt = t;
  }
  if (int rc = cv_rmsnorm_reduce(cur, H, d.slabs, KS, 16 * (int64_t)H, H, d.g_final, d.rms_eps, d.xn, H, d.dtype, d.B, H, st)) return rc;
  {   // speech-token head
    cv_skinny_params p{};
    p.dtype = d.dtype; p.M = d.B; p.N = d.out_vocab; p.K = H;
    p.A = d.xn; p.lda = H; p.Wp = d.p_dec; p.bias = d.dec_b; p.ksplit = 1; p.mode = 0;
    p.out_f32 = d.logits; p.ldo = d.vpad;
    if (int rc = cv_skinny_gemm(&p, st)) return rc;
  }
  cv_sample_params sp = d.sample;
  sp.logits = d.logits; sp.ldl = d.vpad; sp.V = d.out_vocab; sp.B = d.B;
  sp.x = d.x; sp.ldx = H; sp.emb_dim = H;
  return cv_sample_ras(&sp, st);
}

int check_desc(const cv_llm_step_desc* d) {
  if (!d || !d->layers || d->num_layers <= 0) return CV_ERR_ARG;
  if (d->dtype != CV_BF16 && d->dtype != CV_F16) return CV_ERR_UNSUPPORTED;
  if (d->B <= 0 || d->B > 16 || d->down_ksplit < 2 || (d->hidden & 63) || d->num_kv_heads <= 0 || d->num_heads % d->num_kv_heads) return CV_ERR_UNSUPPORTED;
  if (!d->x || !d->x2 || !d->xn || !d->xb || !d->ssp || !d->qkv || !d->ao || !d->h || !d->slabs || !d->logits || !d->rope_table ||
      !d->g_final || !d->p_dec || !d->sample.pos || !d->sample.step || !d->sample.n_emitted || !d->sample.finished ||
      !d->sample.out_tokens || !d->sample.emb_table || !d->sample.min_len || !d->sample.max_len) return CV_ERR_ARG;
  return CV_OK;
}
}  // namespace

extern "C" int cv_sizeof_llm_step_desc(void) { return (int)sizeof(cv_llm_step_desc); }
extern "C" int cv_sizeof_llm_layer(void) { return (int)sizeof(cv_llm_layer); }

extern "C" int cv_llm_step_enqueue(const cv_llm_step_desc* d, void* stream) {
  if (int rc = check_desc(d)) return rc;
  return enqueue_step(*d, (hipStream_t)stream);
}

extern "C" int cv_llm_step_graph_create(const cv_llm_step_desc* d, void* capture_stream, void** graph_out) {
  if (!graph_out || !capture_stream) return CV_ERR_ARG;   // capture needs a non-default stream
  if (int rc = check_desc(d)) return rc;
  if (int rc = cv_graph_begin(capture_stream)) return rc;
  const int rc_body = enqueue_step(*d, (hipStream_t)capture_stream);
  void* g = nullptr;
  const int rc_end = cv_graph_end(capture_stream, &g);
  if (rc_body != CV_OK) { if (g) cv_graph_destroy(g); return rc_body; }
  if (rc_end != CV_OK) return rc_end;
  *graph_out = g;
  return CV_OK;
}
extern "C" int cv_llm_step_graph_launch(void* graph, void* stream) { return cv_graph_launch(graph, stream); }
extern "C" int cv_llm_step_graph_destroy(void* graph) { return cv_graph_destroy(graph); }
