"""GST style embedding + log-mel front end (set_utterance_embedding(path), ToucanTTSInterface.py:103-114).

* oracle (CPU restatement) vs the golden captured from the reference's own StyleEmbedding (tests/golden/make_style_golden.py);
* the host packing (Conv2d stack as banded 2-tap convs, GRU, token attention) through the numpy ABI emulator vs the same golden;
* the HIP path vs the golden and the log-mel kernels vs the float64 oracle (-m gpu).
The log-mel stage is PARITY UNPINNED (librosa is absent: its STFT / mel basis are restated from the documented algorithm)."""
import os

import numpy as np
import pytest
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import fixture_weights as fw, style
from oracle import toucan_oracle as orc

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "style.npz"))
N = len(GOLD["lengths"])


def test_oracle_matches_reference_golden():
    o = orc.StyleOracle(fw.style_state_dict())
    for u in range(N):
        emb, ref = o(torch.from_numpy(GOLD[f"spec{u}"]), return_ref=True)
        np.testing.assert_allclose(ref.numpy(), GOLD[f"ref{u}"], atol=2e-5)
        np.testing.assert_allclose(emb.numpy(), GOLD[f"emb{u}"], atol=2e-5)


def _check_engine(eng, atol_ref, atol_emb):
    specs = [torch.from_numpy(GOLD[f"spec{u}"]) for u in range(N)]
    emb, ref = eng.forward(specs, return_ref=True)  # one batch of four lengths (37 .. 1000 frames)
    for u in range(N):
        np.testing.assert_allclose(ref[u].cpu().numpy(), GOLD[f"ref{u}"], atol=atol_ref)
        np.testing.assert_allclose(emb[u].cpu().numpy(), GOLD[f"emb{u}"], atol=atol_emb)
    one = eng.forward(specs[1:2])  # batching never changes an utterance
    assert torch.equal(one[0], emb[1])


def test_host_packing_matches_reference_golden(monkeypatch):
    from tests import abi_emulator
    abi_emulator.install(monkeypatch)
    _check_engine(style.StyleEngine(fw.style_state_dict(), "cpu"), 3e-5, 3e-5)


def test_logmel_host_pieces():
    fb = style.mel_filterbank()
    assert fb.shape == (80, 513) and np.array_equal(fb, orc.mel_filterbank()) and (fb >= 0).all()
    assert fb[:, : int(40 / (16000 / 1024))].sum() == 0  # nothing below fmin
    assert np.allclose(fb.sum(1) * (16000 / 1024), 1.0, atol=0.2)  # area-normalised triangles (Slaney)
    x = np.sin(2 * np.pi * 440 * np.arange(16000) / 44100.0)
    y = style.resample_sinc(x, 44100, 16000)
    assert y.shape[0] == int(np.ceil(16000 * 16000 / 44100)) and np.abs(y).max() < 1.05
    t = np.arange(y.shape[0]) / 16000.0
    assert np.abs(y[200:-200] - np.sin(2 * np.pi * 440 * t)[200:-200]).max() < 2e-2
    a = style.normalize_reference_audio(np.stack([x, 0.5 * x], axis=1), 16000)
    assert a.ndim == 1 and abs(np.abs(a).max() - 1.0) < 1e-6


@pytest.mark.gpu
def test_hip_style_embedding_matches_reference_golden():
    _check_engine(style.StyleEngine(fw.style_state_dict(), "cuda:0"), 5e-5, 5e-5)


@pytest.mark.gpu
def test_hip_logmel_matches_oracle():
    rs = np.random.RandomState(3)
    t = np.arange(16000 * 2) / 16000.0
    audio = (0.4 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t) + 0.05 * rs.randn(t.size)).astype(np.float32)
    got = style.LogMel("cuda:0").forward(audio).cpu().numpy()
    want = orc.logmel(audio)
    assert got.shape == want.shape == (1 + audio.size // 256, 80)
    assert np.abs(got - want).max() < 2e-3  # log10 of fp32 sums of ~1000 terms vs float64


@pytest.mark.gpu
def test_set_utterance_embedding_from_a_wav_file(tmp_path, monkeypatch):
    """The drop-in's set_utterance_embedding(path): wav file -> normalisation -> log-mel -> GST, all stages on the GPU; equals the
    oracle chain on the same file."""
    import wave
    from ims_toucan_prosody_variance_amd import interface
    models = tmp_path / "Models"
    interface.write_fixture_checkpoints(str(models), n_lang=20)
    monkeypatch.setattr(interface, "MODELS_DIR", str(models))
    sr = 22050
    t = np.arange(int(1.3 * sr)) / sr
    sig = 0.3 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.1 * np.sin(2 * np.pi * 2500 * t)
    path = tmp_path / "ref.wav"
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((sig * 32767).astype("<i2").tobytes())
    tts = interface.ToucanTTSInterface(device="cuda", tts_model_path="Meta")
    before = tts.default_utterance_embedding.clone()
    tts.set_utterance_embedding(str(path))
    emb = tts.default_utterance_embedding
    assert emb.shape == (64,) and emb.is_cuda and not torch.equal(emb.cpu(), before.cpu())
    data, rate = style.read_audio(str(path))
    audio = style.normalize_reference_audio(data, rate)
    want = orc.StyleOracle(fw.style_state_dict())(torch.from_numpy(orc.logmel(audio)))
    np.testing.assert_allclose(emb.cpu().numpy(), want.numpy(), atol=2e-3)
    wav = tts("~həlˈoʊ~#", input_is_phones=True)  # and the new voice is what the next utterance is conditioned on
    assert torch.isfinite(wav).all()
