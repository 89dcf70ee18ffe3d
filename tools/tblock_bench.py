"""Per-kernel timing of the estimator's transformer block at the bench's batch-8 shape (R = 16 CFG rows x T = 1000 frames):
head (LN + QKV), flash attention, tail (to_out + LN + FFN), each as a hipGraph over the 56 blocks' own weights (so the weight
stream comes from L2 / Infinity Cache as in the real pass), replayed between events.  Prints us per call and TFLOP/s."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd import ops
from cosyvoice_amd.config import FlowConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.weights import flow_state_dict

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dt = torch.float16
fc = FlowConfig.full()
flow = CausalMaskedDiffWithXvec(fc, dtype=dt).load_state_dict(flow_state_dict(fc))
est = flow.decoder.estimator
ws = est._workspace(R, T)
ws["x32"].normal_()
tbs = [tb for blk in est.blocks for tb in blk["tb"]]
C, inner, H, Tp = 256, 512, 8, ws["Tp"]

def params(tb):
    p = ops.tblock_params(ws["x32"], R, T, 1e-5, dt)
    p.g1, p.b1n, p.wqkv_p = tb["g1"].data_ptr(), tb["b1"].data_ptr(), tb["wqkv_p"].data_ptr()
    p.qk, p.ldqk, p.vt, p.vt_ld = ws["qk"].data_ptr(), 2 * inner, ws["vt"].data_ptr(), Tp
    p.ao, p.ldao, p.wo_p, p.bo = ws["ao"].data_ptr(), inner, tb["wo_p"].data_ptr(), tb["bo"].data_ptr()
    p.g3, p.b3n = tb["g3"].data_ptr(), tb["b3"].data_ptr()
    p.w1_p, p.bf1, p.w2_p, p.bf2 = tb["wf1_p"].data_ptr(), tb["bf1"].data_ptr(), tb["wf2_p"].data_ptr(), tb["bf2"].data_ptr()
    return p

ps = [params(tb) for tb in tbs]
def head():
    for p in ps: ops.tblock_head(p)
def attn():
    for _ in ps:
        ops.attention(ws["qk"], ws["qk"][:, :, inner:], ws["vt"], ws["ao"], B=R, H=H, Hkv=H, Tq=T, Tk=T, scale=0.125,
                      q_bs=T * 2 * inner, ldq=2 * inner, k_bs=T * 2 * inner, ldk=2 * inner, vt_ld=Tp, o_bs=T * inner, ldo=inner)
def tail():
    for p in ps: ops.tblock_tail(p)
def tail_head():   # tail of block i + head of block i + 1 in one launch (cv_tblock_tail_head): what blocks 0..2 of every group of 4 run
    for i, p in enumerate(ps):
        q = ps[(i + 1) % len(ps)]
        p.g1, p.b1n, p.wqkv_p = q.g1, q.b1n, q.wqkv_p
        ops.tblock_tail_head(p)

def timed(fn, flop):
    fn(); torch.cuda.synchronize()
    g = ops.Graph().capture(fn)
    g.launch(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): g.launch()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (n * len(ps))
    return us, flop / us / 1e6

rows = R * T
for name, fn, flop in (("head", head, 2.0 * rows * 256 * 1536), ("attention", attn, 4.0 * R * H * T * T * 64),
                       ("tail", tail, 2.0 * rows * (512 * 256 + 2 * 256 * 1024)),
                       ("tail+head", tail_head, 2.0 * rows * (512 * 256 + 2 * 256 * 1024 + 256 * 1536))):
    us, tf = timed(fn, flop)
    ws["x32"].normal_()   # the tail updates x in place: keep it bounded
    print(f"{name:10s} R={R} T={T}: {us:7.1f} us per call  {tf:6.0f} TFLOP/s", flush=True)
