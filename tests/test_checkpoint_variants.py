"""The checkpoint variants the reference interface falls back to (ToucanTTSInterface.py:55-63): multi-speaker single-language
(``lang_embs=None``) and single-speaker (``lang_embs=None, utt_embed_dim=None``: LayerNorm predictors, no utterance-embedding
projection).  Goldens ``tests/golden/V20_*.npz`` were captured from the reference's own ``ToucanTTS`` with these constructor
arguments (tests/golden/make_variant_golden.py).  CPU: oracle and host pipeline (ABI emulator); GPU: the HIP path."""
import json
import os

import numpy as np
import pytest
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw
from oracle import toucan_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VARIANTS = ["V20_monolingual", "V20_single"]


def _gold(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return g, fw.acoustic_state_dict(**json.loads(str(g["fixture"])))


def _check(out, g, tol_mel, tol_small):
    assert np.array_equal(np.asarray(out["durations"]), g["durations"]), "durations must be bit exact"
    np.testing.assert_allclose(np.asarray(out["pitch"]), g["pitch"], atol=tol_small)
    np.testing.assert_allclose(np.asarray(out["energy"]), g["energy"], atol=tol_small)
    mel = np.asarray(out["mel"])
    assert mel.shape == g["mel"].shape
    assert np.abs(mel - g["mel"]).max() < tol_mel
    assert np.abs(mel - g["mel"]).mean() < 1e-4  # mel L1 bound of the north star


@pytest.mark.parametrize("name", VARIANTS)
def test_fixture_schema_of_the_variant(name):
    g, sd = _gold(name)
    assert ("encoder.language_embedding.weight" in sd) is False
    assert ("encoder.hs_emb_projection.weight" in sd) is (name == "V20_monolingual")
    assert ("pitch_predictor.norms.0.weight" in sd) is (name == "V20_single")


@pytest.mark.parametrize("name", VARIANTS)
def test_oracle_matches_reference_golden(name):
    g, sd = _gold(name)
    o = orc.AcousticOracle(sd)(torch.from_numpy(g["text"]), torch.from_numpy(g["utt_emb"]), int(g["lang_id"]), z_noise=torch.from_numpy(g["z"]))
    _check({k: o[k].numpy() for k in ("durations", "pitch", "energy", "mel")}, g, 2e-4, 1e-5)


def _engine_outputs(eng, g):
    out = eng.forward([torch.from_numpy(g["text"])], torch.from_numpy(g["utt_emb"])[None], [int(g["lang_id"])], z_noise=[torch.from_numpy(g["z"])])
    return {k: out[k][0].cpu().numpy() for k in ("durations", "pitch", "energy", "mel")}


@pytest.mark.parametrize("name", VARIANTS)
def test_host_pipeline_matches_reference_golden(name, monkeypatch):
    from tests import abi_emulator
    abi_emulator.install(monkeypatch)
    g, sd = _gold(name)
    eng = engine.AcousticEngine(sd, "cpu")
    assert eng.multilingual is False and eng.multispeaker is (name == "V20_monolingual")
    _check(_engine_outputs(eng, g), g, 3e-4, 3e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", VARIANTS)
def test_hip_path_matches_reference_golden(name):
    assert torch.cuda.is_available()
    g, sd = _gold(name)
    _check(_engine_outputs(engine.AcousticEngine(sd, "cuda:0"), g), g, 5e-4, 5e-5)
