"""One-off parity check far outside the benchmark shape: ONE utterance of 1024 phonemes (5120 frames, 1.97 M samples) through the
fp32 HIP path vs the CPU oracle (run on the GPU box: python tools/long_utterance_check.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw, synthetic as syn
from oracle import toucan_oracle as orc


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    dev = torch.device("cuda:0")
    feats = syn.utterance_features(5, L, word_boundaries=False)
    emb = syn.utterance_embedding(5)
    durs = np.full(L, 5, dtype=np.int64)
    z = syn.postflow_noise(5, 5 * L)
    ac_sd, voc_sd = fw.acoustic_state_dict(), fw.hifigan_state_dict()
    t0 = time.time()
    o = orc.AcousticOracle(ac_sd)(torch.from_numpy(feats), torch.from_numpy(emb), syn.LANG_EN, z_noise=torch.from_numpy(z),
                                  durations=torch.from_numpy(durs))
    wav_o = orc.VocoderOracle(voc_sd, "hifigan")(o["mel"].t().contiguous())
    print(f"oracle: {time.time() - t0:.1f} s, mel {tuple(o['mel'].shape)}, wav {wav_o.numel()}", flush=True)
    ac = engine.AcousticEngine(ac_sd, dev)
    voc = engine.VocoderEngine(voc_sd, "hifigan", dev)
    out = ac.forward([torch.from_numpy(feats).to(dev)], torch.from_numpy(emb)[None].to(dev), [syn.LANG_EN],
                     durations=[torch.from_numpy(durs).to(dev)], z_noise=[torch.from_numpy(z).to(dev)])
    wav, _ = voc.forward(out["mel_packed"], out["rag_mel"])
    torch.cuda.synchronize()
    mel = out["mel"][0].cpu().numpy()
    err = np.abs(mel - o["mel"].numpy())
    werr = np.abs(wav.cpu().numpy()[: wav_o.numel()] - wav_o.numpy())
    print(f"mel max err {err.max():.3e}  L1 {err.mean():.3e}   wav max err {werr.max():.3e}")
    assert err.mean() < 1e-4 and err.max() < 2e-3 and werr.max() < 2e-3
    print("long utterance ok")


if __name__ == "__main__":
    main()
