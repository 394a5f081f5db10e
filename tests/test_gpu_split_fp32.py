"""The split fp32 product (TTS_COMPUTE_F32X3: every dense product of the frame stages / the vocoder as three fp16 MFMAs on split
operands, fp32 accumulation, fp32 tensors everywhere) gated on EVERY fp32 reference golden with the fp32 tolerances: durations bit
exact, mel max-abs 5e-4 / L1 < 1e-4 (the north-star bound), waveform max-abs 5e-4 - through both sequencers, alone and inside a
full-size batch of 32 (where every conv of the frame stages really takes the split form: on small grids the exact fp32 split-K
form is the fast one and stays)."""
import json
import os

import numpy as np
import pytest
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw, native, synthetic as syn
from ims_toucan_prosody_variance_amd.ragged import Ragged

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _gold(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _kw(g):
    kw = json.loads(str(g["ctrl"]))
    if "gold_durations" in g.files:
        kw["durations"] = [torch.from_numpy(g["gold_durations"])]
    if "gold_pitch" in g.files:
        kw["pitch"], kw["energy"] = [torch.from_numpy(g["gold_pitch"])], [torch.from_numpy(g["gold_energy"])]
    return kw


@pytest.fixture(scope="module")
def paths():
    ac, hf = fw.acoustic_state_dict(), fw.hifigan_state_dict()
    return dict(engine=engine.AcousticEngine(ac, DEV, precision="f32x3"), native=native.NativePipeline(ac, hf, "hifigan", DEV, precision="f32x3"))


@pytest.mark.parametrize("name", ["L7_pred", "L20_pred", "L20_ctrl", "L20_gold_odd", "L20_gold_prosody", "L128_gold5"])
def test_split_fp32_acoustic_matches_every_reference_golden(paths, name):
    g = _gold(name)
    args = ([torch.from_numpy(g["text"])], torch.from_numpy(g["utt_emb"])[None], [int(g["lang_id"])])
    for which, p in paths.items():
        out = p.forward(*args, z_noise=[torch.from_numpy(g["z"])], **_kw(g))
        assert np.array_equal(out["durations"][0].cpu().numpy(), g["durations"]), (which, "durations must be bit exact")
        np.testing.assert_allclose(out["pitch"][0].cpu().numpy(), g["pitch"], atol=5e-5)
        err = np.abs(out["mel"][0].cpu().numpy() - g["mel"])
        assert err.max() < 5e-4 and err.mean() < 1e-4, (which, name, float(err.max()), float(err.mean()))
        if which == "native" and "wav_hifigan" in g.files:
            b, n = out["wav_spans"][0]
            assert np.abs(out["wav"][b:b + n].cpu().numpy() - g["wav_hifigan"]).max() < 5e-4


@pytest.mark.parametrize("kind", ["hifigan", "bigvgan"])
def test_split_fp32_vocoders_match_the_reference_goldens(kind):
    sd = fw.hifigan_state_dict() if kind == "hifigan" else fw.bigvgan_state_dict()
    voc = engine.VocoderEngine(sd, kind, DEV, precision="f32x3")
    pipe = native.NativePipeline(fw.acoustic_state_dict(), sd, kind, DEV, precision="f32x3")
    for name in ("L7_pred", "L20_pred"):
        g = _gold(name)
        mel = torch.from_numpy(g["mel"]).to(DEV).contiguous()
        rag = Ragged([mel.shape[0]], DEV)
        for wav, r in (voc.forward(mel, rag), pipe.vocode(mel, rag)):
            assert np.abs(wav.cpu().numpy()[: r.lengths[0]] - g["wav_" + kind]).max() < 5e-4, (kind, name)
    g = _gold("L128_gold5")  # full length: head / tail / checksum
    mel = torch.from_numpy(g["mel"]).to(DEV).contiguous()
    wav, _ = pipe.vocode(mel, Ragged([mel.shape[0]], DEV))
    wav = wav.cpu().numpy()
    assert wav.shape[0] == int(g["wav_len"])
    assert np.abs(wav[:8192] - g[f"wav_{kind}_head"]).max() < 5e-4 and np.abs(wav[-8192:] - g[f"wav_{kind}_tail"]).max() < 5e-4
    assert abs(float(np.abs(wav.astype(np.float64)).sum()) - float(g[f"wav_{kind}_abs_sum"])) < 1e-4 * wav.shape[0]


def test_split_fp32_full_size_batch_keeps_the_fp32_tolerances_and_exact_durations():
    """Batch 32 x 128 phonemes (every frame-stage conv on the split product): utterance 0 = the reference golden L128_gold5 within
    the fp32 tolerances; predicted durations of a 32-batch bit-identical to the exact fp32 pipeline's (the phoneme stages are the
    same arithmetic in both); the two sequencers agree bit for bit."""
    g = _gold("L128_gold5")
    B, L = 32, 128
    texts = [torch.from_numpy(g["text"])] + [torch.from_numpy(syn.utterance_features(u, L, word_boundaries=False)) for u in range(1, B)]
    embs = torch.stack([torch.from_numpy(g["utt_emb"])] + [torch.from_numpy(syn.utterance_embedding(u)) for u in range(1, B)])
    durs = [torch.from_numpy(g["gold_durations"]).to(torch.int32)] + [torch.full((L,), 5, dtype=torch.int32) for _ in range(1, B)]
    zs = [torch.from_numpy(g["z"])] + [torch.from_numpy(syn.postflow_noise(u, 5 * L)) for u in range(1, B)]
    langs = [int(g["lang_id"])] * B
    ac_sd = fw.acoustic_state_dict()
    x3 = native.NativePipeline(ac_sd, None, None, DEV, precision="f32x3")
    out = x3.forward(texts, embs, langs, durations=durs, z_noise=zs)
    err = np.abs(out["mel"][0].cpu().numpy() - g["mel"])
    print("split fp32, batch 32: mel max-abs", float(err.max()), "L1", float(err.mean()))
    assert err.max() < 5e-4 and err.mean() < 1e-4, (float(err.max()), float(err.mean()))
    eng = engine.AcousticEngine(ac_sd, DEV, precision="f32x3")
    ref = eng.forward(texts, embs, langs, durations=durs, z_noise=zs)
    for u in (0, 1, 31):
        assert torch.equal(out["mel"][u], ref["mel"][u]), u
    exact = native.NativePipeline(ac_sd, None, None, DEV, precision="f32")
    gen = lambda: torch.Generator(device=DEV).manual_seed(7)
    a = x3.forward(texts, embs, langs, generator=gen(), run_postflow=False)
    b = exact.forward(texts, embs, langs, generator=gen(), run_postflow=False)
    assert torch.equal(a["durations_packed"], b["durations_packed"]) and torch.equal(a["pitch_packed"], b["pitch_packed"])
    e2 = float((a["mel_packed"] - b["mel_packed"]).abs().max())
    print("split fp32 vs exact fp32, batch 32, predicted durations: mel max-abs difference", e2)
    assert e2 < 5e-4
