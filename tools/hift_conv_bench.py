"""Times the HiFT v2 ResBlock conv launches (stride-1 bf16x3 convs on pre-split operands) one shape at a time: gemm_kernel (implicit
im2col, the default) against conv_win_kernel (CV_CONV_WIN=1) in every workgroup shape, each as a 10-launch hipGraph.  BENCH_C2=1: the second conv of a ResBlock unit (fp32 residual in, fp32 stream out as well).   python tools/hift_conv_bench.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosyvoice_amd import _lib as L, ops
from cosyvoice_amd.hift import _presplit


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [1, 8]
    dev = "cuda"
    stages = [(256, 4000), (128, 20000), (64, 60000)]   # channels, frames per utterance (500 mel frames x 8 x 5 x 3)
    for B in Bs:
        for C, T in stages:
            for k, dil in ((3, 1), (7, 3), (11, 5)):
                torch.manual_seed(0)
                x = _presplit(torch.randn(B * T, C)).view(torch.float32).reshape(B, T, C).to(dev)
                W = _presplit(torch.randn(C, k * C) / (C * k) ** 0.5).view(torch.float32).to(dev)
                bias = torch.randn(C, device=dev)
                alpha = torch.rand(C, device=dev) + 0.5
                oa = torch.empty(B, T, C, device=dev)
                c2 = os.environ.get("BENCH_C2", "0") == "1"      # the ResBlock's second conv: + fp32 residual in, + fp32 stream out
                res = torch.randn(B, T, C, device=dev) if c2 else None
                o32 = torch.empty(B, T, C, device=dev) if c2 else None
                flops = 2.0 * B * T * C * C * k * 3
                row = []
                for tag, env in (("gemm", {"CV_CONV_WIN": "0"}), ("auto", {"CV_CONV_WIN": "1"}), ("4x1", {"CV_CONV_WIN": "1", "CV_CONV_WIN_SHAPE": "4x1"}), ("2x2", {"CV_CONV_WIN": "1", "CV_CONV_WIN_SHAPE": "2x2"}),
                                 ("2x1", {"CV_CONV_WIN": "1", "CV_CONV_WIN_SHAPE": "2x1"}), ("1x2", {"CV_CONV_WIN": "1", "CV_CONV_WIN_SHAPE": "1x2"}), ("1x1", {"CV_CONV_WIN": "1", "CV_CONV_WIN_SHAPE": "1x1"})):
                    for kk in ("CV_CONV_WIN", "CV_CONV_WIN_SHAPE"):
                        os.environ.pop(kk, None)
                    os.environ.update(env)
                    if tag in ("2x2", "1x2") and C < 128:
                        row.append("   -  ")
                        continue

                    def run():
                        ops.conv1d_cl(x, W, k, dilation=dil, pad_left=(k * dil - dil) // 2, bias=bias, act=ops.ACT_SNAKE, act_param=alpha, out_act=oa,
                                      res=res, out_f32=o32, dtype=L.CV_F32X3, x3_flags=7)
                    run()
                    torch.cuda.synchronize()
                    reps = 10

                    def many():
                        for _ in range(reps):
                            run()
                    g = ops.Graph().capture(many)     # replayed: the Python launch path (~20 us per call) would otherwise bound the small shapes
                    g.launch()
                    torch.cuda.synchronize()
                    n = 5
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        g.launch()
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / (n * reps)
                    row.append(f"{tag} {us:7.1f} us {flops / us * 1e-6:5.0f} TF")
                print(f"B={B} C={C:3d} T={T:5d} k={k:2d} d={dil}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
