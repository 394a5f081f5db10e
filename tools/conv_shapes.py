"""Per-shape timing of every matrix-core launch of one bench step (stage API, tts_profile mode 2): where the acoustic model's
small-GEMM time goes.  Run on the GPU box: python tools/conv_shapes.py [--dtype bf16|fp16|fp32] [--batch 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import fixture_weights as fw, native, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    prec = {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}[args.dtype]
    pipe = native.NativePipeline(fw.acoustic_state_dict(), fw.bigvgan_state_dict(), "bigvgan", dev, precision=prec)
    B, L = args.batch, 128
    texts = [torch.from_numpy(syn.utterance_features(u, L, word_boundaries=False)).to(dev) for u in range(B)]
    embs = torch.from_numpy(np.stack([syn.utterance_embedding(u) for u in range(B)])).to(dev)
    durs = [torch.full((L,), 5, dtype=torch.int32, device=dev) for _ in range(B)]
    zs = [torch.from_numpy(syn.postflow_noise(u, 5 * L)).to(dev) for u in range(B)]
    run = lambda: pipe.forward(texts, embs, [syn.LANG_EN] * B, durations=durs, z_noise=zs)
    run()
    run()
    pipe.profile(2)
    run()
    torch.cuda.synchronize()
    s = pipe.profile_summary()
    tot = sum(v["total_ms"] for v in s.values())
    print(f"{tot:.2f} ms in {sum(v['launches'] for v in s.values())} matrix-core launches")
    for k, v in sorted(s.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"{v['total_ms']:8.3f} ms  n={v['launches']:3d}  avg {v['avg_us']:8.1f} us  {v['tflops']:7.1f} TF/s  {k}")


if __name__ == "__main__":
    main()
