"""The vocoder's stage-1 convs (256 -> 256 channels, 163 840 rows at batch 32) and its up-samplers as they run in the bf16 / fp16
configuration (16-bit tensors in and out, residual add on the second conv of a step), back to back:

    python tools/microbench_conv256.py [--reps 10]

us per launch, algorithmic TFLOP/s and GB/s (x + res + y once)."""
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
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = engine.Ops(dev)
    print(f"{'shape':>24} {'k':>3} {'dil':>3} {'res':>4} {'us':>9} {'TFLOP/s':>8} {'GB/s':>7}")
    shapes = [(256, 256, 8, k, dil, res) for k, dil in ((3, 1), (7, 3), (11, 5)) for res in (False, True)]
    shapes += [(512, 2048, 1, 3, 1, False), (256, 768, 8, 3, 1, False), (128, 256, 48, 3, 1, False)]
    for cin, cout, mult, k, dil, with_res in shapes:
        rag = Ragged([args.frames * mult] * args.batch, dev)
        R = rag.total_rows
        rs = np.random.RandomState(0)
        cw = packing.pack_conv((rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32), np.zeros(cout, np.float32), dev, dil=dil, bf16=True)
        x = torch.randn(R, cin, device=dev).to(torch.bfloat16)
        y = torch.empty(R, cout, device=dev, dtype=torch.bfloat16)
        res = torch.randn(R, cout, device=dev).to(torch.bfloat16) if with_res else None
        run = lambda: ops.conv(cw, x, y, rag, res=res, compute=capi.COMPUTE_BF16)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / args.reps
        flops = 2.0 * R * cin * cout * k
        gb = (R * cin * 2 + R * cout * 2 * (2 if with_res else 1)) / 1e9
        print(f"{f'{cin}->{cout} r{R}':>24} {k:>3} {dil:>3} {str(with_res)[0]:>4} {us:9.1f} {flops / us / 1e6:8.1f} {gb / (us * 1e-6):7.0f}", flush=True)


if __name__ == "__main__":
    main()
