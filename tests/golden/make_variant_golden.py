"""Golden vectors for the checkpoint variants the reference interface falls back to (ToucanTTSInterface.py:55-63):
multi-speaker single-language (``lang_embs=None``) and single-speaker (``lang_embs=None, utt_embed_dim=None``).
Runs ONLY where /root/reference exists.  Same recipe as make_golden.py: the build's seeded fixture weights go through the
reference's strict ``load_state_dict``, the reference forward runs with injected Glow noise, the oracle must reproduce it, and
inputs + outputs are stored (data only) as ``tests/golden/V20_monolingual.npz`` / ``V20_single.npz``.

    python tests/golden/make_variant_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stand-in modules + sys.path)
import torch  # noqa: E402

from ims_toucan_prosody_variance_amd import fixture_weights as fw, synthetic as syn  # noqa: E402
from oracle import toucan_oracle as orc  # noqa: E402


def main():
    variants = [("V20_monolingual", dict(n_lang=None, multispeaker=True), dict(lang_embs=None)),
                ("V20_single", dict(n_lang=None, multispeaker=False), dict(lang_embs=None, utt_embed_dim=None))]
    for name, fw_kw, ref_kw in variants:
        sd = fw.acoustic_state_dict(**fw_kw)
        ref = mg.RefAcoustic(sd, **ref_kw)
        oracle = orc.AcousticOracle(sd)
        u, L = 30, 20
        feats = syn.utterance_features(u, L)
        emb = syn.utterance_embedding(u)
        text, e = torch.from_numpy(feats), torch.from_numpy(emb)
        probe = oracle(text, e, syn.LANG_EN, run_postflow=False)
        T = int(probe["durations"].sum())
        z = torch.from_numpy(syn.postflow_noise(u, T))
        with torch.inference_mode():
            mel_r, dur_r, pitch_r, energy_r = ref(text, e, syn.LANG_EN, z)
        o = oracle(text, e, syn.LANG_EN, z_noise=z)
        errs = dict(mel=mg.close("mel", o["mel"], mel_r, 2e-5), pitch=mg.close("pitch", o["pitch"], pitch_r, 1e-5),
                    energy=mg.close("energy", o["energy"], energy_r, 1e-5))
        assert torch.equal(o["durations"], dur_r.reshape(-1))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), text=feats, utt_emb=emb, lang_id=np.int64(syn.LANG_EN), z=z.numpy(),
                            mel=mel_r.numpy(), durations=dur_r.reshape(-1).numpy(), pitch=pitch_r.reshape(-1).numpy(),
                            energy=energy_r.reshape(-1).numpy(), decoded=ref.taps["decoded"].numpy(), enc_out=ref.taps["enc_out"].numpy(),
                            fixture=json.dumps(fw_kw))
        print(name, "T", T, "oracle vs reference", errs)


if __name__ == "__main__":
    main()
