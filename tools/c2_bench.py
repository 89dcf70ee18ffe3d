"""BASELINE config C2 / batch-1 flow: ONE utterance through the flow stage (encoder + 10 CFG Euler steps as one hipGraph), at
T = 500 (250 tokens, no prompt: C2), T = 650 (C1: 75-token prompt + 250) and T = 1000 (C4's utterance: 250 + 250), all CUs.
CV_TBLOCK_MT / CV_ATTN_WAVES override the tile choices (tuning).  Prints ms and TFLOP/s on SURVEY.md §8d's flop counts."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd.config import FlowConfig
from cosyvoice_amd.flow import CausalMaskedDiffWithXvec
from cosyvoice_amd.weights import flow_state_dict

dt = torch.float16
fc = FlowConfig.full()
flow = CausalMaskedDiffWithXvec(fc, dtype=dt).load_state_dict(flow_state_dict(fc))
flow.decoder.use_graph = True
g = torch.Generator().manual_seed(3)
dev = "cuda"
for n_p, n_g, tf in ((0, 250, 1.895), (75, 250, 2.687), (250, 250, 4.937)):
    tok = torch.randint(0, fc.vocab_size, (1, n_g), generator=g, dtype=torch.int32).to(dev)
    ptok = torch.randint(0, fc.vocab_size, (1, n_p), generator=g, dtype=torch.int32).to(dev)
    pfeat = torch.clamp(torch.randn(1, 2 * n_p, 80, generator=g) * 2 - 6, -11.5, 2.0).to(dev)
    emb = torch.randn(1, fc.spk_embed_dim, generator=g).to(dev)
    for _ in range(3):
        flow.inference_batch(tok, ptok, pfeat, emb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        flow.inference_batch(tok, ptok, pfeat, emb)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"flow batch 1, T={2 * (n_p + n_g)} [MT={os.environ.get('CV_TBLOCK_MT', 'auto')}, attn waves={os.environ.get('CV_ATTN_WAVES', 'auto')}]: "
          f"{ms:.2f} ms  {tf / ms * 1e3:.0f} TFLOP/s ({tf / ms * 1e3 / 2500:.4f} of 2.5 PF)", flush=True)
