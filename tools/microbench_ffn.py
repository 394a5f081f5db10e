"""Micro-benchmark of tts_ffn_fused against the launches it replaces (run on the MI355X box)."""
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
    ap.add_argument("--rows", default="20480,4096,640")
    ap.add_argument("--hidden", type=int, default=1536)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = engine.Ops(dev)
    rs = np.random.RandomState(0)
    Cc, H = 192, args.hidden
    w1 = (rs.randn(H, Cc, 1) / np.sqrt(Cc)).astype(np.float32)
    w2 = (rs.randn(Cc, H, 1) / np.sqrt(H)).astype(np.float32)
    b1, b2 = np.zeros(H, np.float32), np.zeros(Cc, np.float32)
    c1 = packing.pack_conv(w1, b1, dev, bf16="bf16")
    c2 = packing.pack_conv(w2, b2, dev, bf16="bf16")
    pk = packing.pack_ffn(w1, b1, w2, dev, "bf16")
    g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    print(f"{'rows':>7} {'fused us':>9} {'TFLOP/s':>8} {'unfused us':>11}")
    for R in [int(r) for r in args.rows.split(",")]:
        x = torch.randn(R, Cc, device=dev)
        ln = torch.empty(R, Cc, device=dev)
        hid = torch.empty(R, H, device=dev, dtype=torch.bfloat16)
        rag = Ragged([R], dev)

        def fused():
            ops.ffn_fused(x, x, (g, b), pk, c2.bias, R, capi.COMPUTE_BF16, post=(g, b))

        def unfused():
            ops.layernorm(x, ln, g, b, R, Cc)
            ops.conv(c1, ln, hid, rag, act=capi.ACT_RELU, compute=capi.COMPUTE_BF16)
            ops.conv(c2, hid, x, rag, alpha=0.5, res=x, compute=capi.COMPUTE_BF16)
            ops.layernorm(x, x, g, b, R, Cc)

        out = []
        for fn in (fused, unfused):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.append(1e3 * e0.elapsed_time(e1) / args.reps)
        print(f"{R:>7} {out[0]:9.1f} {4.0 * R * Cc * H / out[0] / 1e6:8.1f} {out[1]:11.1f}", flush=True)
        diag = getattr(ops.lib, "tts_ffn_diag_clock", None)
        if diag is not None:  # (diagnostic library built with -DFF_DIAG_CLOCK: shader cycles per chunk, workgroup 0 / wavefront 0)
            import ctypes as C_
            buf = (C_.c_ulonglong * 8)()
            fused()
            torch.cuda.synchronize()
            diag(buf)
            n = max(1, buf[5])
            names = ("wait+barrier", "issue+reads", "first product", "relu/pack", "second product")
            print("        cycles per chunk: " + "  ".join(f"{nm} {buf[i] / n:.0f}" for i, nm in enumerate(names)) + f"  | total {sum(buf[:5]) / n:.0f}", flush=True)


if __name__ == "__main__":
    main()
