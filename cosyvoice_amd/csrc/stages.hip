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
  const int RP = d.B <= 16 ? 16 : 32;   // row pitch of the split-K slabs (one or two 16-row MFMA groups)
  float* cur = d.x;
  float* nxt = d.split_qkv_norm ? d.x : d.x2;   // split form: residual updated in place
  for (int li = 0; li < d.num_layers; ++li) {
    const cv_llm_layer& L = d.layers[li];
    if (d.split_qkv_norm) {   // one workgroup per row: residual += slabs, normalised 16-bit row; then QKV over plain rows
      if (int rc = cv_rmsnorm_reduce(cur, H, li > 0 ? d.slabs : nullptr, li > 0 ? KS : 0, RP * (int64_t)H, H, L.g_in, d.rms_eps, d.xn, H,
                                     d.dtype, d.B, H, st)) return rc;
      cv_skinny_params p{};
      p.dtype = d.dtype; p.M = d.B; p.N = qkv_dim; p.K = H;
      p.A = d.xn; p.lda = H; p.Wp = L.p_qkv; p.bias = L.bqkv; p.ksplit = 1; p.mode = 0;
      p.out_f32 = d.qkv; p.ldo = qkv_dim;
      if (int rc = cv_skinny_gemm(&p, st)) return rc;
    } else {   // RMSNorm(input_layernorm) of (residual + previous down-projection slabs) -> QKV; workgroup (0,0) stores the summed residual
      cv_skinny_params p{};
      p.dtype = d.dtype; p.M = d.B; p.N = qkv_dim; p.K = H;
      p.A = d.xn; p.lda = H; p.Wp = L.p_qkv; p.bias = L.bqkv; p.ksplit = 1; p.mode = 0;
      p.out_f32 = d.qkv; p.ldo = qkv_dim;
      p.nx = cur; p.ldnx = H; p.ngamma = L.g_in; p.neps = d.rms_eps; p.nx_out = nxt;
      if (li > 0) { p.nslabs = d.slabs; p.n_nslab = KS; p.nslab_stride = RP * (int64_t)H; p.ld_nslab = H; }
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
      p.out_f32 = d.slabs; p.ldo = H; p.slab_stride = RP * (int64_t)H;
      if (int rc = cv_skinny_gemm(&p, st)) return rc;
    }
    float* t = cur; cur = nxt; nxt = t;
  }
  if (int rc = cv_rmsnorm_reduce(cur, H, d.slabs, KS, RP * (int64_t)H, H, d.g_final, d.rms_eps, d.xn, H, d.dtype, d.B, H, st)) return rc;
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
  if (d->B <= 0 || d->B > 32 || (d->B > 16 && !d->split_qkv_norm) || d->down_ksplit < 2 || (d->hidden & 63) || d->num_kv_heads <= 0 || d->num_heads % d->num_kv_heads) return CV_ERR_UNSUPPORTED;
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


// ================================================================================================ flow-matching solver
// cv_flow_euler_*: n_steps x [pack CFG rows -> estimator -> Euler update].  Behavioural spec: flow/flow_matching.py:72-124
// (solve_euler, CFG batch of 2 per utterance) and flow/decoder.py:224-334 (ConditionalDecoder.forward); the launch sequence is the
// one cosyvoice_amd/flow.py composes (ConditionalDecoder.forward_cl + CausalConditionalCFM.solve) and is tested equal to it.
namespace {
int conv_k3(const cv_flow_solver_desc& d, const void* a, int lda, const void* w, const float* bias, void* out_act, float* out_f32,
            hipStream_t st) {
  const int R = 2 * d.B, T = d.T, C = d.C;
  cv_gemm_params p{};
  p.dtype = d.dtype; p.M = T; p.N = C; p.K = 3 * C; p.batch = R; p.batch_inner = 0;
  p.A = a; p.a_bs0 = (int64_t)T * lda; p.lda = lda; p.a_rows = T;
  p.cin = C; p.a_row_stride = 1; p.tap_base = -2; p.tap_step = 1;
  p.W = w; p.ldw = 3 * C; p.bias = bias; p.out_scale = 1.0f;
  p.out_row_stride = 1;
  if (out_act) { p.out_act = out_act; p.oa_bs0 = (int64_t)T * C; p.ldoa = C; }
  if (out_f32) { p.out_f32 = out_f32; p.o32_bs0 = (int64_t)T * C; p.ldo32 = C; }
  return cv_gemm(&p, st);
}

int enqueue_estimator(const cv_flow_solver_desc& d, const float* tadd_row, hipStream_t st) {
  const int R = 2 * d.B, T = d.T, C = d.C, inner = d.inner;
  const size_t esz = 2;
  const void* a_in = d.xin;
  int lda = d.in_ch, cin = d.in_ch;
  for (int bi = 0; bi < d.n_blocks; ++bi) {
    const cv_flow_block& blk = d.blocks[bi];
    {   // causal resnet block: conv + LN + Mish + time term | conv + LN + Mish + 1x1 conv of the input
      cv_resblock_params p{};
      p.dtype = d.dtype; p.R = R; p.T = T; p.C = C; p.cin = cin;
      p.a = a_in; p.lda = lda;
      p.w1_p = blk.res.w1_p; p.b1 = blk.res.b1; p.g1 = blk.res.g1; p.be1 = blk.res.be1; p.tadd = tadd_row + (size_t)bi * C;
      p.h1 = d.h1; p.ldh1 = C;
      p.w2_p = blk.res.w2_p; p.b2 = blk.res.b2; p.g2 = blk.res.g2; p.be2 = blk.res.be2;
      p.wr_p = blk.res.wr_p; p.br = blk.res.br;
      p.out = d.x32; p.ldo = C; p.eps = d.eps; p.cus = d.cus;
      if (int rc = cv_resblock_conv1(&p, st)) return rc;
      if (int rc = cv_resblock_conv2(&p, st)) return rc;
    }
    for (int j = 0; j < blk.n_tb; ++j) {
      const cv_flow_tblock& tb = blk.tb[j];
      const bool last = j == blk.n_tb - 1;
      const bool fuse = blk.fuse_tail_head != 0;
      cv_tblock_params p{};
      p.dtype = d.dtype; p.R = R; p.T = T; p.C = C; p.inner = inner; p.ff = d.ff;
      p.x = d.x32; p.ldx = C; p.eps = d.eps; p.cus = d.cus;
      p.g1 = tb.g1; p.b1n = tb.b1n; p.wqkv_p = tb.wqkv_p;
      p.qk = d.qk; p.ldqk = 2 * inner; p.vt = d.vt; p.vt_ld = d.Tp;
      if (j == 0 || !fuse) {   // with fuse_tail_head the head of block j > 0 ran inside the previous block's tail launch
        if (int rc = cv_tblock_head(&p, st)) return rc;
      }
      cv_attn_params a{};
      a.dtype = d.dtype; a.B = R; a.H = d.heads; a.Hkv = d.heads; a.Tq = T; a.Tk = T;
      a.q = d.qk; a.q_bs = (int64_t)T * 2 * inner; a.ldq = 2 * inner;
      a.k = (const char*)d.qk + (size_t)inner * esz; a.k_bs = (int64_t)T * 2 * inner; a.ldk = 2 * inner;
      a.vt = d.vt; a.vt_ld = d.Tp;
      a.out = d.ao; a.o_bs = (int64_t)T * inner; a.ldo = inner;
      a.scale = 0.125f; a.klen = d.klen;
      if (int rc = cv_attention(&a, st)) return rc;
      p.ao = d.ao; p.ldao = inner; p.wo_p = tb.wo_p; p.bo = tb.bo; p.g3 = tb.g3; p.b3n = tb.b3n;
      p.w1_p = tb.w1_p; p.bf1 = tb.bf1; p.w2_p = tb.w2_p; p.bf2 = tb.bf2;
      if (last) {   // 16-bit copy of the block output: skip tensor / next block's conv input (decoder.py:277,301)
        if (bi == 0) { p.out_act = (char*)d.cat + (size_t)C * esz; p.ldoa = 2 * C; }
        else if (bi == d.n_blocks - 2) { p.out_act = d.cat; p.ldoa = 2 * C; }
        else { p.out_act = d.d; p.ldoa = C; }
        if (int rc = cv_tblock_tail(&p, st)) return rc;
      } else if (fuse) {   // tail of block j + head of block j + 1 in one launch
        const cv_flow_tblock& nx = blk.tb[j + 1];
        p.g1 = nx.g1; p.b1n = nx.b1n; p.wqkv_p = nx.wqkv_p;
        if (int rc = cv_tblock_tail_head(&p, st)) return rc;
      } else {
        if (int rc = cv_tblock_tail(&p, st)) return rc;
      }
    }
    if (bi == 0) {   // downsample slot = CausalConv1d k3 on the skip tensor (decoder.py:278)
      if (int rc = conv_k3(d, (const char*)d.cat + (size_t)C * esz, 2 * C, d.down_w, d.down_b, d.d, nullptr, st)) return rc;
      a_in = d.d; lda = C; cin = C;
    } else if (bi == d.n_blocks - 2) {
      a_in = d.cat; lda = 2 * C; cin = 2 * C;
    } else {
      a_in = d.d; lda = C; cin = C;
    }
  }
  // upsample slot, final block (conv + LN + Mish), final_proj (decoder.py:331-334)
  if (int rc = conv_k3(d, d.d, C, d.up_w, d.up_b, d.h1, nullptr, st)) return rc;
  if (int rc = conv_k3(d, d.h1, C, d.fin_w, d.fin_b, nullptr, d.c32a, st)) return rc;
  const int rows = R * T;
  {
    cv_norm_params n{};
    n.rows = rows; n.dim = C; n.rms = 0; n.eps = d.eps; n.x = d.c32a; n.ldx = C; n.gamma = d.fin_g; n.beta = d.fin_be;
    n.act = CV_ACT_MISH; n.out_scale = 1.0f; n.out_dtype = d.dtype; n.out_act = d.h1; n.ldoa = C;
    if (int rc = cv_layernorm(&n, st)) return rc;
  }
  cv_gemm_params g{};
  g.dtype = d.dtype; g.M = rows; g.N = d.out_ch; g.K = C; g.batch = 1;
  g.A = d.h1; g.lda = C; g.a_row_stride = 1; g.W = d.proj_w; g.ldw = C; g.bias = d.proj_b; g.out_scale = 1.0f;
  g.out_f32 = d.v; g.ldo32 = d.out_ch; g.out_row_stride = 1;
  return cv_gemm(&g, st);
}

int enqueue_solver(const cv_flow_solver_desc& d, hipStream_t st) {
  for (int i = 0; i < d.n_steps; ++i) {
    if (int rc = cv_est_pack(d.x, d.mu, d.spks, d.cond, d.xin, d.dtype, d.B, d.T, d.out_ch, st)) return rc;
    if (int rc = enqueue_estimator(d, d.tadd + (size_t)i * d.n_blocks * d.C, st)) return rc;
    if (int rc = cv_cfm_update(d.x, d.v, d.B, d.T, d.out_ch, d.dts[i], d.cfg_rate, st)) return rc;
  }
  return CV_OK;
}

int check_flow_desc(const cv_flow_solver_desc* d) {
  if (!d || !d->blocks || !d->dts || d->n_blocks < 3 || d->n_steps <= 0) return CV_ERR_ARG;
  if (d->dtype != CV_BF16 && d->dtype != CV_F16) return CV_ERR_UNSUPPORTED;
  if (d->C != 256 || d->inner != 512 || d->ff != 1024 || d->heads != 8 || d->in_ch != 4 * d->out_ch) return CV_ERR_UNSUPPORTED;   // the fused kernels' shapes
  if (d->B <= 0 || d->T <= 0 || d->Tp < d->T || (d->Tp & 7)) return CV_ERR_ARG;
  if (!d->x || !d->mu || !d->spks || !d->cond || !d->tadd || !d->xin || !d->h1 || !d->x32 || !d->qk || !d->vt || !d->ao || !d->cat ||
      !d->d || !d->v || !d->c32a || !d->down_w || !d->up_w || !d->fin_w || !d->fin_g || !d->fin_be || !d->proj_w) return CV_ERR_ARG;
  for (int i = 0; i < d->n_blocks; ++i) {
    const cv_flow_block& b = d->blocks[i];
    if (b.n_tb <= 0 || !b.tb || !b.res.w1_p || !b.res.w2_p || !b.res.wr_p) return CV_ERR_ARG;
    const int want = i == 0 ? d->in_ch : (i == d->n_blocks - 1 ? 2 * d->C : d->C);
    if (b.res.cin != want) return CV_ERR_ARG;
  }
  return CV_OK;
}
}  // namespace

extern "C" int cv_sizeof_flow_solver_desc(void) { return (int)sizeof(cv_flow_solver_desc); }
extern "C" int cv_sizeof_flow_block(void) { return (int)sizeof(cv_flow_block); }
extern "C" int cv_sizeof_flow_tblock(void) { return (int)sizeof(cv_flow_tblock); }

extern "C" int cv_flow_euler_enqueue(const cv_flow_solver_desc* d, void* stream) {
  if (int rc = check_flow_desc(d)) return rc;
  return enqueue_solver(*d, (hipStream_t)stream);
}

extern "C" int cv_flow_euler_graph_create(const cv_flow_solver_desc* d, void* capture_stream, void** graph_out) {
  if (!graph_out || !capture_stream) return CV_ERR_ARG;
  if (int rc = check_flow_desc(d)) return rc;
  if (int rc = cv_graph_begin(capture_stream)) return rc;
  const int rc_body = enqueue_solver(*d, (hipStream_t)capture_stream);
  void* g = nullptr;
  const int rc_end = cv_graph_end(capture_stream, &g);
  if (rc_body != CV_OK) { if (g) cv_graph_destroy(g); return rc_body; }
  if (rc_end != CV_OK) return rc_end;
  *graph_out = g;
  return CV_OK;
}
extern "C" int cv_flow_euler_graph_launch(void* graph, void* stream) { return cv_graph_launch(graph, stream); }
extern "C" int cv_flow_euler_graph_destroy(void* graph) { return cv_graph_destroy(graph); }


// ================================================================================================ HiFT decode
// cv_hift_decode_*: HiFTGenerator.decode, hifigan/generator.py:349-381 (ResBlock :91-98, SourceModule STFT :333-347).  The launch
// sequence is the one cosyvoice_amd/hift.py composes (_decode_cl_impl / _resblock) and is tested equal to it.
namespace {
struct Epi {   // epilogue of one conv launch
  const float* res = nullptr; const float* res2 = nullptr; float out_scale = 1.0f;
  int act = CV_ACT_NONE; const float* act_param = nullptr; float act_slope = 0.0f;
  float* out_f32 = nullptr; int ldo32 = 0; void* out_act = nullptr;
};

int hconv(const cv_hift_decode_desc& d, const cv_hift_conv& c, const void* x, int T_in, int T_out, const Epi& e, hipStream_t st) {
  cv_gemm_params p{};
  p.dtype = d.gemm_dtype; p.M = T_out; p.N = c.cout; p.K = c.k * c.cin; p.batch = d.B;
  p.A = x; p.a_bs0 = (int64_t)T_in * c.cin; p.lda = c.cin; p.a_rows = T_in;
  p.cin = c.cin; p.a_row_stride = c.stride; p.tap_base = -c.pad_left; p.tap_step = c.dilation;
  p.W = c.w; p.ldw = c.k * c.cin; p.bias = c.b;
  if (e.res) { p.res = e.res; p.res_bs0 = (int64_t)T_out * c.cout; p.ldres = c.cout; }
  if (e.res2) { p.res2 = e.res2; p.ldres2 = c.cout; }
  p.out_scale = e.out_scale; p.act = e.act; p.act_param = e.act_param; p.act_slope = e.act_slope;
  if (e.out_f32) { const int ld = e.ldo32 ? e.ldo32 : c.cout; p.out_f32 = e.out_f32; p.o32_bs0 = (int64_t)T_out * ld; p.ldo32 = ld; }
  if (e.out_act) { p.out_act = e.out_act; p.oa_bs0 = (int64_t)T_out * c.cout; p.ldoa = c.cout; }
  p.out_row_stride = 1;
  p.x3_flags = e.out_act ? c.x3_flags : (c.x3_flags & ~4);
  return cv_gemm(&p, st);
}

// ResBlock (generator.py:91-98): x -> x + c2(snake(c1(snake(x)))) per unit; xa = snake_{a1[0]}(x32) already computed
int hresblock(const cv_hift_decode_desc& d, const cv_hift_resblock& rb, const cv_hift_stage& sg, const float* x32, const void* xa,
              const Epi& fin, hipStream_t st) {
  const float* cur32 = x32;
  const void* cur_a = xa;
  const int t = sg.t_out;
  for (int j = 0; j < rb.n_units; ++j) {
    const cv_hift_resunit& u = rb.units[j];
    Epi e1; e1.act = CV_ACT_SNAKE; e1.act_param = u.a2; e1.out_act = sg.ta;
    if (int rc = hconv(d, u.c1, cur_a, t, t, e1, st)) return rc;
    if (j < rb.n_units - 1) {
      float* nxt = (j & 1) ? sg.r1 : sg.r0;
      Epi e2; e2.res = cur32; e2.out_f32 = nxt; e2.act = CV_ACT_SNAKE; e2.act_param = rb.units[j + 1].a1; e2.out_act = sg.ra;
      if (int rc = hconv(d, u.c2, sg.ta, t, t, e2, st)) return rc;
      cur32 = nxt; cur_a = sg.ra;
    } else {
      Epi e2 = fin; e2.res = cur32;
      if (int rc = hconv(d, u.c2, sg.ta, t, t, e2, st)) return rc;
    }
  }
  return CV_OK;
}

int enqueue_hift_decode(const cv_hift_decode_desc& d, hipStream_t st) {
  const int B = d.B, F = d.S / d.hop + 1;
  if (int rc = cv_stft16(d.s, d.stft, d.dtype, B, d.S, d.stft_ld, st)) return rc;
  {
    Epi e; e.act = CV_ACT_LEAKY; e.act_slope = d.lrelu_slope; e.out_act = d.a_pre;
    if (int rc = hconv(d, d.conv_pre, d.mel_cl, d.T, d.T, e, st)) return rc;
  }
  const void* cur_a = d.a_pre;
  int t_in = d.T;
  for (int i = 0; i < d.n_stages; ++i) {
    const cv_hift_stage& sg = d.stages[i];
    const int t_out = sg.t_out, c = sg.c;
    const bool last = i == d.n_stages - 1;
    {   // source branch (generator.py:361-363): strided conv of the source STFT, then its ResBlock -> si1
      Epi e; e.out_f32 = sg.si0; e.act = CV_ACT_SNAKE; e.act_param = sg.source_rb.units[0].a1; e.out_act = sg.xa[0];
      if (int rc = hconv(d, sg.source_down, d.stft, F, t_out, e, st)) return rc;
      Epi fin; fin.out_f32 = sg.si1;
      if (int rc = hresblock(d, sg.source_rb, sg, sg.si0, sg.xa[0], fin, st)) return rc;
    }
    // ups[i] as u phase GEMMs (+ reflect pad on the last stage) + source fusion: x = ups(x) + si (generator.py:355-364)
    auto phase = [&](const cv_hift_phase& ph, int M, int row_off) {
      cv_gemm_params p{};
      p.dtype = d.gemm_dtype; p.M = M; p.N = c; p.K = ph.ntaps * sg.up_cin; p.batch = B;
      p.A = cur_a; p.a_bs0 = (int64_t)t_in * sg.up_cin; p.lda = sg.up_cin; p.a_rows = t_in;
      p.cin = sg.up_cin; p.a_row_stride = 1; p.tap_base = ph.tap_base; p.tap_step = -1;
      p.W = ph.w; p.ldw = ph.ntaps * sg.up_cin; p.bias = sg.up_b;
      p.res = sg.si1; p.res_bs0 = (int64_t)t_out * c; p.ldres = c; p.out_scale = 1.0f;
      p.out_f32 = sg.x32; p.o32_bs0 = (int64_t)t_out * c; p.ldo32 = c;
      p.out_row_stride = sg.u; p.out_row_off = row_off; p.out_rows = t_out;
      p.x3_flags = sg.up_flags;
      return cv_gemm(&p, st);
    };
    const int off = last ? 1 : 0;
    for (int r = 0; r < sg.u; ++r)
      if (int rc = phase(sg.phases[r], t_in, r + off)) return rc;
    if (last)   // ReflectionPad1d((1, 0)): padded[0] = ups_out[1] = phase r = 1, q = 0
      if (int rc = phase(sg.phases[1], 1, 0)) return rc;
    // parallel ResBlocks, mean over kernels (generator.py:366-372), then leaky-relu (:353 / :374)
    {
      const float* alphas[8]; void* outs[8];
      for (int j = 0; j < d.n_kernels; ++j) { alphas[j] = sg.rbs[j].units[0].a1; outs[j] = sg.xa[j]; }
      if (int rc = cv_snake_multi(sg.x32, B * t_out, c, c, d.n_kernels, alphas, outs, c, d.presplit ? (int)CV_F32X3 : d.dtype, st)) return rc;
    }
    const float slope = last ? 0.01f : d.lrelu_slope;   // F.leaky_relu default after the loop (generator.py:374)
    for (int j = 0; j < d.n_kernels; ++j) {
      Epi fin;
      if (j < d.n_kernels - 1) fin.out_f32 = (j & 1) ? sg.acc1 : sg.acc0;
      else { fin.out_scale = 1.0f / d.n_kernels; fin.act = CV_ACT_LEAKY; fin.act_slope = slope; fin.out_act = sg.out; }
      if (j > 0) fin.res2 = ((j - 1) & 1) ? sg.acc1 : sg.acc0;
      if (int rc = hresblock(d, sg.rbs[j], sg, sg.x32, sg.xa[j], fin, st)) return rc;
    }
    cur_a = sg.out;
    t_in = t_out;
  }
  {
    Epi e; e.out_f32 = d.post; e.ldo32 = d.stft_ld;
    if (int rc = hconv(d, d.conv_post, cur_a, t_in, t_in, e, st)) return rc;
  }
  return cv_istft16(d.post, d.wav, B, t_in, d.stft_ld, d.audio_limit, st);
}

int check_hift_desc(const cv_hift_decode_desc* d) {
  if (!d || !d->stages || d->n_stages <= 0 || d->n_kernels <= 0 || d->n_kernels > 4) return CV_ERR_ARG;
  if (d->B <= 0 || d->T <= 0 || d->S <= 0 || d->hop != 4 || d->stft_ld < 18) return CV_ERR_ARG;
  if (!d->mel_cl || !d->s || !d->stft || !d->a_pre || !d->post || !d->wav || !d->conv_pre.w || !d->conv_post.w) return CV_ERR_ARG;
  for (int i = 0; i < d->n_stages; ++i) {
    const cv_hift_stage& g = d->stages[i];
    if (!g.phases || g.u < 2 || !g.rbs || !g.xa || !g.x32 || !g.r0 || !g.r1 || !g.ta || !g.ra || !g.acc0 || !g.acc1 || !g.si0 || !g.si1 ||
        !g.out || !g.source_rb.units || g.source_rb.n_units <= 0 || g.t_out <= 0 || g.c <= 0) return CV_ERR_ARG;
    for (int j = 0; j < d->n_kernels; ++j)
      if (!g.rbs[j].units || g.rbs[j].n_units <= 0 || !g.xa[j]) return CV_ERR_ARG;
    // pre-split activation storage ([8 hi | 8 lo] groups of 8 values): every channel count it is applied to is a whole number of groups
    const bool any_x3 = d->presplit || g.up_flags || g.source_down.x3_flags;
    if (any_x3 && (g.c & 7)) return CV_ERR_ARG;
  }
  if ((d->presplit || d->conv_pre.x3_flags) && (d->conv_pre.cout & 7)) return CV_ERR_ARG;
  return CV_OK;
}
}  // namespace

extern "C" int cv_sizeof_hift_decode_desc(void) { return (int)sizeof(cv_hift_decode_desc); }
extern "C" int cv_sizeof_hift_stage(void) { return (int)sizeof(cv_hift_stage); }
extern "C" int cv_sizeof_hift_resunit(void) { return (int)sizeof(cv_hift_resunit); }

extern "C" int cv_hift_decode_enqueue(const cv_hift_decode_desc* d, void* stream) {
  if (int rc = check_hift_desc(d)) return rc;
  return enqueue_hift_decode(*d, (hipStream_t)stream);
}

// SURVEY.md §8b names the entry point cv_hift_decode: same call
extern "C" int cv_hift_decode(const cv_hift_decode_desc* d, void* stream) { return cv_hift_decode_enqueue(d, stream); }

extern "C" int cv_hift_decode_graph_create(const cv_hift_decode_desc* d, void* capture_stream, void** graph_out) {
  if (!graph_out || !capture_stream) return CV_ERR_ARG;
  if (int rc = check_hift_desc(d)) return rc;
  if (int rc = cv_graph_begin(capture_stream)) return rc;
  const int rc_body = enqueue_hift_decode(*d, (hipStream_t)capture_stream);
  void* g = nullptr;
  const int rc_end = cv_graph_end(capture_stream, &g);
  if (rc_body != CV_OK) { if (g) cv_graph_destroy(g); return rc_body; }
  if (rc_end != CV_OK) return rc_end;
  *graph_out = g;
  return CV_OK;
}
