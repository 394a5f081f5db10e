"""Micro-benchmark of tts_conv1d on the vocoder's residual-conv shapes (run on the MI355X box).

    python tools/microbench_conv.py [--batch 32] [--frames 640]

For every stage (C = 256/128/64/32 at 8/48/192/384 rows per frame), kernel size and input activation it times the
launch with HIP events and prints algorithmic TFLOP/s and the activation GB/s (read x + read res + write y, fp32)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import capi, engine, packing
from ims_toucan_prosody_variance_amd.ragged import Ragged


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=640)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = engine.Ops(dev)
    filt = torch.from_numpy(packing.kaiser_sinc_filter12()).to(dev)
    print(f"{'C':>4} {'k':>3} {'dil':>3} {'pre':>6} {'dtype':>5} {'us':>9} {'TFLOP/s':>8} {'GB/s':>7}")
    for C, mult in ((256, 8), (128, 48), (64, 192), (32, 384)):
        rows = args.frames * mult
        rag = Ragged([rows] * args.batch, dev)
        R = rag.total_rows
        x = torch.randn(R, C, device=dev)
        res = torch.randn(R, C, device=dev)
        y = torch.empty(R, C, device=dev)
        sn = (torch.zeros(C, device=dev), torch.zeros(C, device=dev), filt)
        for k, dil in ((3, 1), (7, 3), (11, 5)):
            w = (np.random.RandomState(0).randn(C, C, k) / np.sqrt(C * k)).astype(np.float32)
            cw = packing.pack_conv(w, np.zeros(C, np.float32), dev, dil=dil, bf16=True)
            for pre, pname in ((capi.PRE_NONE, "none"), (capi.PRE_LRELU, "lrelu"), (capi.PRE_SNAKE, "snake")):
                for comp, cname in ((capi.COMPUTE_F32, "f32"), (capi.COMPUTE_BF16, "bf16")):
                    tag = f"{C}-{k}-{pname}-{cname}"
                    if args.only and args.only not in tag:
                        continue
                    run = lambda: ops.conv(cw, x, y, rag, pre=pre, pre_slope=0.1, snake=sn if pre == capi.PRE_SNAKE else None, res=res,
                                           compute=comp)
                    run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.reps):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    us = 1e3 * e0.elapsed_time(e1) / args.reps
                    flops = 2.0 * R * C * C * k
                    gbytes = 3.0 * R * C * 4 / 1e9
                    print(f"{C:>4} {k:>3} {dil:>3} {pname:>6} {cname:>5} {us:9.1f} {flops / us / 1e6:8.1f} {gbytes / (us * 1e-6):7.0f}", flush=True)


if __name__ == "__main__":
    main()
