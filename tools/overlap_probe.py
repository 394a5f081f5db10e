"""Does the acoustic model of batch k+1 overlap the vocoder of batch k when they run on two HIP streams?  (run on the GPU box)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import fixture_weights as fw, native, synthetic as syn


def main():
    dev = torch.device("cuda:0")
    pipe = native.NativePipeline(fw.acoustic_state_dict(), fw.bigvgan_state_dict(), "bigvgan", dev, precision="bf16")
    B, L, T = 32, 128, 640
    texts = [torch.from_numpy(syn.utterance_features(u, L, word_boundaries=False)).to(dev) for u in range(B)]
    embs = torch.from_numpy(np.stack([syn.utterance_embedding(u) for u in range(B)])).to(dev)
    durs = [torch.full((L,), 5, dtype=torch.int32, device=dev) for _ in range(B)]
    zs = [torch.from_numpy(syn.postflow_noise(u, T)).to(dev) for u in range(B)]
    packed = pipe.pack_inputs(texts, embs, [syn.LANG_EN] * B, durations=durs)
    z_sq = pipe.squeeze_noise(zs, [T] * B)
    K = 8

    def serial():
        for _ in range(K):
            out = pipe.forward(None, None, packed=packed, z_sq=z_sq, vocode=False)
            wav, _ = pipe.vocode_batch(out["rag_mel"])
        return wav

    sa, sv = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def pipelined():
        wav = None
        for _ in range(K):
            with torch.cuda.stream(sa):
                out = pipe.forward(None, None, packed=packed, z_sq=z_sq, vocode=False)
                mel = out["mel_packed"]
                ev = torch.cuda.Event()
                ev.record(sa)
            with torch.cuda.stream(sv):
                sv.wait_event(ev)
                mel.record_stream(sv)
                wav, _ = pipe.vocode(mel, out["rag_mel"])
        return wav

    for name, fn in (("serial", serial), ("pipelined", pipelined), ("serial", serial), ("pipelined", pipelined)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"{name}: {1e3 * dt:.2f} ms/step  {B * T / dt:.0f} frames/s  checksum {float(w[:1000].abs().sum()):.4f}")


if __name__ == "__main__":
    main()
