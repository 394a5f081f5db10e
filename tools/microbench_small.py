"""Batch-1 conv shapes of the acoustic model, launched back to back (warm instruction cache, warm L2) - compare with the
in-pipeline durations of the same launches in a kernel trace to separate cold-start cost from steady-state cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import capi, engine, packing
from ims_toucan_prosody_variance_amd.ragged import Ragged


def main():
    dev = torch.device("cuda:0")
    ops = engine.Ops(dev)
    if len(sys.argv) > 1:  # python tools/microbench_small.py M [M ...]: the 1-tap shapes at other row counts
        shapes = [(f"K{cin} N{cout}", int(m), cin, cout, 1, capi.MODE_LINEAR) for m in sys.argv[1:] for cin, cout in ((192, 192), (192, 576), (1536, 192))]
    else:
        shapes = [("WN gated k5", 320, 192, 384, 5, capi.MODE_GATED), ("FFN2 dec", 640, 1536, 192, 1, capi.MODE_LINEAR),
                  ("FFN1 dec", 640, 192, 1536, 1, capi.MODE_LINEAR), ("qkv dec", 640, 192, 576, 1, capi.MODE_LINEAR),
                  ("FFN2 enc", 128, 1536, 192, 1, capi.MODE_LINEAR), ("out enc", 128, 192, 192, 1, capi.MODE_LINEAR),
                  ("FFN1 enc", 128, 192, 1536, 1, capi.MODE_LINEAR), ("WN res/skip", 320, 192, 384, 1, capi.MODE_LINEAR),
                  ("pred k3", 128, 256, 256, 3, capi.MODE_LINEAR), ("postnet k5", 640, 512, 512, 5, capi.MODE_LINEAR),
                  ("dec conv k3", 640, 192, 192, 3, capi.MODE_LINEAR)]
    for name, M, cin, cout, k, mode in shapes:
        rs = np.random.RandomState(0)
        cw = packing.pack_conv((rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32), np.zeros(cout, np.float32), dev, mode=mode, bf16=True)
        cw3 = packing.pack_conv((rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32), np.zeros(cout, np.float32), dev, mode=mode, bf16="x3")
        rag = Ragged([M], dev)
        x = torch.randn(rag.total_rows, cin, device=dev)
        y = torch.empty(rag.total_rows, cout // 2 if mode != capi.MODE_LINEAR else cout, device=dev)
        for comp, cname in ((capi.COMPUTE_F32, "f32"), (capi.COMPUTE_BF16, "bf16"), (capi.COMPUTE_F32X3, "f32x3")):
            for split in ((False, True) if comp == capi.COMPUTE_F32 else (False,)):
                ops.split_k = int(split)  # fp32: also the split-K form the acoustic model opts into (TTS_IO_SPLIT_K)
                run = lambda: ops.conv(cw3 if comp == capi.COMPUTE_F32X3 else cw, x, y, rag, compute=comp)
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    run()
                e1.record()
                torch.cuda.synchronize()
                print(f"{name:12s} M={M:4d} {cin:4d}->{cout:4d} k={k} {cname + (' split-K' if split else ''):13s}: {1e3 * e0.elapsed_time(e1) / 50:7.1f} us / launch", flush=True)


if __name__ == "__main__":
    main()
