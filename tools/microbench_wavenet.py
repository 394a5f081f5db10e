"""Micro-benchmark of tts_wavenet_layer vs the two-launch form at the bench shape (32 utterances x 320 squeezed frames)."""
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
    B, T, H = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 320, 192
    rag = Ragged([T] * B, dev)
    R = rag.total_rows
    rs = np.random.RandomState(0)
    inl = packing.pack_conv((rs.randn(2 * H, H, 5) / np.sqrt(5 * H)).astype(np.float32), np.zeros(2 * H, np.float32), dev, mode=capi.MODE_GATED, bf16=True)
    res = packing.pack_conv((rs.randn(2 * H, H, 1) / np.sqrt(H)).astype(np.float32), np.zeros(2 * H, np.float32), dev, bf16=True)
    hs, out = torch.randn(R, 2 * H, device=dev), torch.empty(R, 2 * H, device=dev)
    cond = torch.randn(R, 8 * H, device=dev)[:, : 2 * H]
    acts = torch.empty(R, H, device=dev, dtype=torch.bfloat16)

    def t(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps

    fused = t(lambda: ops.wavenet_layer(inl, res, hs, out, cond, rag))

    def two():
        ops.conv(inl, hs[:, :H], acts, rag, preadd=cond, compute=capi.COMPUTE_BF16)
        ops.conv(res, acts, hs, rag, accumulate=True, compute=capi.COMPUTE_BF16)

    print(f"rows {R}: fused {fused:.1f} us   two launches {t(two):.1f} us   ({2 * R * H * 384 * 6 / fused / 1e6:.0f} TFLOP/s fused)")


if __name__ == "__main__":
    main()
