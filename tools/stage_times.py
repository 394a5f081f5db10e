"""Time of every stage entry of the stage API at the bench's configuration (run on the MI355X box): HIP events around each call."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import capi, fixture_weights as fw, native, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--phones", type=int, default=128)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    pipe = native.NativePipeline(fw.acoustic_state_dict(), fw.bigvgan_state_dict(), "bigvgan", dev, precision=args.precision)
    B, L = args.batch, args.phones
    texts = [torch.from_numpy(syn.utterance_features(i, L, word_boundaries=False)).to(dev) for i in range(B)]
    embs = torch.stack([torch.from_numpy(syn.utterance_embedding(i)) for i in range(B)]).to(dev)
    durs = [torch.full((L,), 5, dtype=torch.int32, device=dev) for _ in range(B)]
    zs = [torch.from_numpy(syn.postflow_noise(i, 5 * L)).to(dev) for i in range(B)]
    packed = pipe.pack_inputs(texts, embs, [syn.LANG_EN] * B, durations=durs)
    z_sq = pipe.squeeze_noise(zs, [5 * L] * B)
    lib, h = pipe.lib, pipe.h
    st = torch.cuda.current_stream(dev).cuda_stream
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    lens = (C.c_int32 * B)(*packed["Ls"])
    frames = (C.c_int32 * B)()
    pipe.forward(None, None, packed=packed, z_sq=z_sq)  # warm-up (workspace, tables)
    torch.cuda.synchronize()
    stages = [
        ("encoder", lambda: lib.tts_encoder(h, ptr(packed["text"]), ptr(packed["emb"]), ptr(packed["lang"]), lens, B, st)),
        ("variance_predictors", lambda: lib.tts_variance_predictors(h, ptr(packed["gp"]), ptr(packed["ge"]), ptr(packed["gd"]), st)),
        ("control_and_regulate", lambda: lib.tts_control_and_regulate(h, 1.0, 1.0, 1.0, 1.0, frames, st)),
        ("decoder", lambda: lib.tts_decoder(h, st)),
        ("postnet", lambda: lib.tts_postnet(h, st)),
        ("postflow", lambda: lib.tts_postflow(h, ptr(z_sq), st)),
        ("vocoder", lambda: pipe.vocode_batch(__import__("ims_toucan_prosody_variance_amd").ragged.Ragged([int(f) for f in frames], dev, align=2).halved().doubled())),
    ]
    tot = {n: 0.0 for n, _ in stages}
    for rep in range(args.reps + 1):
        for name, fn in stages:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn()
            if isinstance(rc, int):
                capi.check(rc, name)
            e1.record()
            e1.synchronize()
            if rep:
                tot[name] += e0.elapsed_time(e1)
    for name, _ in stages:
        print(f"{name:>22}: {tot[name] / args.reps:8.3f} ms")
    print(f"{'sum':>22}: {sum(tot.values()) / args.reps:8.3f} ms")


if __name__ == "__main__":
    main()
