"""Latency / throughput of the other BASELINE.json configurations on one MI355X (run on the GPU box).

  configs[1]: batch = 1, 128 phonemes, acoustic model fp32, mel only (no vocoder)
  configs[0]-like: one ~20-phoneme utterance end to end with the Avocodo vocoder (fp32)
Prints one JSON line per configuration (median of `--reps` after 3 warm-ups, host wall clock around a device sync)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import capi, engine, fixture_weights as fw, native, synthetic as syn


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--graphs", action="store_true", help="replay the pass as HIP graphs (fixed shapes; python sequencer)")
    ap.add_argument("--sequencer", default="native", choices=["native", "python"])
    ap.add_argument("--only-configs1", action="store_true", help="configs[1] alone (for a kernel trace of that pass)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    use_native = args.sequencer == "native" and not args.graphs
    if use_native:
        pipe = native.NativePipeline(fw.acoustic_state_dict(), fw.hifigan_state_dict(), "hifigan", dev)
    else:
        ac = engine.AcousticEngine(fw.acoustic_state_dict(), dev, use_graphs=args.graphs)
        voc = engine.VocoderEngine(fw.hifigan_state_dict(), "hifigan", dev, use_graphs=args.graphs)
    tag = "native stage API" if use_native else "python sequencer"

    L = 128
    text = [torch.from_numpy(syn.utterance_features(0, L, word_boundaries=False)).to(dev)]
    emb = torch.from_numpy(syn.utterance_embedding(0))[None].to(dev)
    dur = [torch.full((L,), 5, dtype=torch.int32, device=dev)]
    z = [torch.from_numpy(syn.postflow_noise(0, 5 * L)).to(dev)]
    if use_native:
        run1 = lambda: pipe.forward(text, emb, [syn.LANG_EN], durations=dur, z_noise=z, vocode=False)
    else:
        run1 = lambda: ac.forward(text, emb, [syn.LANG_EN], durations=dur, z_noise=z)
    c0 = capi.CALLS
    run1()
    calls = capi.CALLS - c0
    t = timeit(run1, args.reps)
    print(json.dumps({"config": "configs[1]: batch=1 x 128 phonemes, acoustic fp32, mel only", "graphs": args.graphs, "sequencer": tag,
                      "abi_calls": calls, "ms": 1e3 * t, "mel_frames_per_s": 5 * L / t}))
    if args.only_configs1:
        return

    L = 20
    text = [torch.from_numpy(syn.utterance_features(1, L)).to(dev)]
    emb = torch.from_numpy(syn.utterance_embedding(1))[None].to(dev)

    def e2e():
        if use_native:
            out = pipe.forward(text, emb, [syn.LANG_EN])
            b, n = out["wav_spans"][0]
            return out["wav"][b:b + n]
        out = ac.forward(text, emb, [syn.LANG_EN])
        w, rw = voc.forward(out["mel_packed"], out["rag_mel"])
        return w[: rw.lengths[0]]

    t = timeit(e2e, args.reps)
    wav = e2e()
    print(json.dumps({"config": "configs[0]-like: one 20-phoneme utterance, predicted durations, acoustic + Avocodo fp32, end to end",
                      "graphs": args.graphs, "sequencer": tag, "ms": 1e3 * t, "audio_s": wav.numel() / 24000.0, "rtf": t / (wav.numel() / 24000.0)}))

    if use_native:
        # one 128-phoneme sentence (10.2 s of audio) end to end with BigVGAN, the way read_to_file meets a single sentence: the exact
        # configuration (fp32 everywhere) and the mixed one (fp32 acoustic model = the same mel bit for bit, fp16 vocoder)
        L = 128
        text = [torch.from_numpy(syn.utterance_features(0, L, word_boundaries=False)).to(dev)]
        emb = torch.from_numpy(syn.utterance_embedding(0))[None].to(dev)
        dur = [torch.full((L,), 5, dtype=torch.int32, device=dev)]
        z = [torch.from_numpy(syn.postflow_noise(0, 5 * L)).to(dev)]
        for name, vp in (("fp32", None), ("mixed (fp32 acoustic + fp16 vocoder)", "f16")):
            p2 = native.NativePipeline(fw.acoustic_state_dict(), fw.bigvgan_state_dict(), "bigvgan", dev, precision="f32", vocoder_precision=vp)
            run = lambda: p2.forward(text, emb, [syn.LANG_EN], durations=dur, z_noise=z)
            t = timeit(run, args.reps)
            print(json.dumps({"config": f"one 128-phoneme sentence (640 frames, 10.24 s of audio), acoustic + BigVGAN, {name}, end to end",
                              "sequencer": tag, "ms": 1e3 * t, "rtf": t / 10.24}))


if __name__ == "__main__":
    main()
