"""Un-profiled cost of each decode-step kernel: a hipGraph holding only that kernel for all 24 layers (each layer's own
weights, so the stream stays HBM-cold as in the real step), replayed back to back; wall time / launches.
rocprofv3's per-kernel durations carry a ~3 us/kernel profiler floor on this stack (tools/ubench/floor*.hip) and cannot
rank kernels this short."""
import math, sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench as B
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig, LlmConfig
from cosyvoice_amd.llm import Qwen2LM
from cosyvoice_amd.weights import llm_state_dict

lc, fc = LlmConfig.full(), FlowConfig.full()
llm = Qwen2LM(lc, dtype=torch.bfloat16, max_batch=int(__import__("os").environ.get("LKB_B", "8")), ctx_max=704, max_out=258)
llm.load_state_dict(llm_state_dict(lc))
texts, forced, ptext, pspeech, pfeat, emb = B.make_inputs(lc, fc, 100, n_utts=int(__import__("os").environ.get("LKB_B", "8")))
dev = "cuda"; Bn = int(__import__("os").environ.get("LKB_B", "8"))
texts_d = [t.to(dev) for t in texts]; pt, ps = ptext.to(dev), pspeech.to(dev)
llm.generate_batch(texts_d, [pt]*Bn, [ps]*Bn, forced=forced, max_steps=int(__import__('os').environ.get('LKB_STEPS', '120')))   # caches filled to ctx ~400, state valid
torch.cuda.synchronize()
cfg, st = llm.cfg, llm.st
H, I = cfg.hidden_size, cfg.intermediate_size
qkv_dim = cfg.q_dim + 2 * cfg.kv_dim
KS = llm.DOWN_KSPLIT
scale = 1.0 / math.sqrt(cfg.head_dim)
x, x2 = st["x"], st["x2"]
CAP = 0   # max_wgs passed to the skinny kernels while a graph is captured

def k_qkv_plain(li, lay):   # the > 8-row form: input norm as its own launch (k_norm), plain rows into the QKV kernel
    ops.skinny_gemm(st["xn"], lay["p_qkv"], Bn, qkv_dim, H, bias=lay["bqkv"], out_f32=st["qkv"], ldo=qkv_dim)
def k_norm(li, lay): ops.rmsnorm_reduce(x, lay["g_in"], cfg.rms_eps, st["xn"], Bn, slabs=st["slabs"], nslab=KS, slab_stride=llm._rp(Bn) * H, ld_slab=H)
def k_qkv(li, lay):
    if Bn > 16: return k_qkv_plain(li, lay)
    ops.skinny_gemm(st["xn"], lay["p_qkv"], Bn, qkv_dim, H, bias=lay["bqkv"], out_f32=st["qkv"], ldo=qkv_dim,
                    norm=dict(x=x, gamma=lay["g_in"], eps=cfg.rms_eps, x_out=x2, slabs=st["slabs"], nslab=KS, slab_stride=llm._rp(Bn) * H, ld_slab=H), max_wgs=CAP)
def k_attn(li, lay):
    ops.decode_attention(st["q"], llm.kcache[li], llm.vtcache[li], st["pos"], 1, st["ao"], Bn, cfg.num_heads, cfg.num_kv_heads,
                         llm.ctx_max, scale, qkv=st["qkv"], inv_freq=llm.rope_table)
def k_o(li, lay): ops.skinny_gemm(st["ao"], lay["p_o"], Bn, H, cfg.q_dim, mode=1, out_f32=x2, ldo=H, max_wgs=CAP)
def k_gu(li, lay):
    if Bn > 16: return
    ops.skinny_gemm(st["xn"], lay["p_gu"], Bn, 2 * I, H, mode=2, out_act=st["h"], ldoa=I, norm=dict(x=x2, gamma=lay["g_post"], eps=cfg.rms_eps), max_wgs=CAP)
def k_gu_nonorm(li, lay):   # ablation: the same weight stream and SwiGLU epilogue without the RMSNorm prologue
    if Bn > 16:   # two row groups: only the split-norm consumer form exists (what the step uses)
        return ops.skinny_gemm(st["xb"], lay["p_gu_g"], Bn, 2 * I, H, mode=2, out_act=st["h"], ldoa=I,
                               split_in=dict(rs=st["ssp"], n=st["ssp"].shape[0], eps=cfg.rms_eps))
    ops.skinny_gemm(st["xn"], lay["p_gu"], Bn, 2 * I, H, mode=2, out_act=st["h"], ldoa=I, max_wgs=CAP)
def k_down(li, lay): ops.skinny_gemm(st["h"], lay["p_down"], Bn, H, I, ksplit=KS, out_f32=st["slabs"], ldo=H, slab_stride=llm._rp(Bn) * H, max_wgs=CAP)
def k_final(li, lay): ops.rmsnorm_reduce(x, llm.g_final, cfg.rms_eps, st["xn"], Bn, slabs=st["slabs"], nslab=KS, slab_stride=llm._rp(Bn) * H, ld_slab=H)
def k_head(li, lay): ops.skinny_gemm(st["xn"], llm.p_dec, Bn, cfg.out_vocab, H, bias=llm.dec_b, out_f32=st["logits"], ldo=llm.Vpad, max_wgs=CAP)
def k_sample(li, lay):
    st["finished"].zero_()
    llm._head_and_sample(Bn, True, False)

def bench(name, fn, per_rep, configs):
    global CAP
    out = []
    for s, cap in configs:
        CAP = cap
        def body():
            for li, lay in enumerate(llm.layers): fn(li, lay)
        g = ops.Graph().capture(body)
        with torch.cuda.stream(s):
            for _ in range(3): g.launch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): g.launch()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / (20 * per_rep) * 1e6)
    CAP = 0
    print(f"{name:10s} " + "  ".join(f"{t:6.2f} us" for t in out), flush=True)

full, part = torch.cuda.current_stream(), ops.masked_stream(lambda s_, x_: s_ < 8)
configs = [(full, 0), (part, 0)]
print("kernel      all CUs | 64 CUs")
only = __import__('os').environ.get('LKB_ONLY')
for name, fn, per in (("in_norm", k_norm, 24), ("qkv_plain", k_qkv_plain, 24), ("qkv", k_qkv, 24), ("attn", k_attn, 24), ("o_proj", k_o, 24), ("gate_up", k_gu, 24), ("gu_nonorm", k_gu_nonorm, 24), ("down", k_down, 24),
                      ("final_norm", k_final, 24), ("head", k_head, 24)):
    if only and name != only: continue
    bench(name, fn, per, configs if name not in ("attn", "final_norm") else configs[:2])
if not only: bench("head+samp", k_sample, 24, configs[:1])   # per (head + memset + sampler) triple

if __import__('os').environ.get('LKB_PAIR'):
    # the same layer's launch twice in a row: does the second one run from L2 (weights streamed with the nt policy)?
    def pair(fn):
        return lambda li, lay: (fn(li, lay), fn(li, lay))
    print("pairs (us per PAIR)")
    for name, fn in (("gate_up", k_gu_nonorm), ("down", k_down), ("o_proj", k_o)):
        bench(name + "x2", pair(fn), 24, configs)
