"""Host logic (engine sequencing, weight packing, ragged layout) on CPU: the real engine drives the numpy
ABI emulator (tests/abi_emulator.py) and must reproduce the oracle / reference goldens.  No GPU."""
import json
import os

import numpy as np
import pytest
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw
from tests import abi_emulator

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def emu(monkeypatch):
    return abi_emulator.install(monkeypatch)


@pytest.fixture(scope="module")
def ac_sd():
    return fw.acoustic_state_dict()


def _gold(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _inputs(gs):
    texts = [torch.from_numpy(g["text"]) for g in gs]
    embs = torch.stack([torch.from_numpy(g["utt_emb"]) for g in gs])
    langs = [int(g["lang_id"]) for g in gs]
    zs = [torch.from_numpy(g["z"]) for g in gs]
    return texts, embs, langs, zs


def test_acoustic_engine_single_utterance_matches_reference_golden(emu, ac_sd):
    g = _gold("L7_pred")
    eng = engine.AcousticEngine(ac_sd, "cpu")
    texts, embs, langs, zs = _inputs([g])
    taps = {}
    out = eng.forward(texts, embs, langs, z_noise=zs, taps=taps)
    assert np.array_equal(out["durations"][0].numpy(), g["durations"])
    for b in range(6):
        np.testing.assert_allclose(taps[f"enc_block{b}"].numpy(), g[f"tap_enc_block{b}"], atol=3e-5, err_msg=f"enc{b}")
    np.testing.assert_allclose(taps["enc_out"].numpy(), g["enc_out"], atol=3e-5)
    np.testing.assert_allclose(out["pitch"][0].numpy(), g["pitch"], atol=3e-5)
    np.testing.assert_allclose(out["energy"][0].numpy(), g["energy"], atol=3e-5)
    np.testing.assert_allclose(taps["upsampled"].numpy()[: g["tap_upsampled"].shape[0]], g["tap_upsampled"], atol=3e-5)
    for b in range(6):
        np.testing.assert_allclose(taps[f"dec_block{b}"].numpy()[: g["tap_dec_block0"].shape[0]], g[f"tap_dec_block{b}"], atol=5e-5)
    np.testing.assert_allclose(out["decoded_packed"].numpy()[: g["decoded"].shape[0]], g["decoded"], atol=1e-4)
    np.testing.assert_allclose(taps["glow_g"].numpy()[: g["tap_glow_g"].shape[1]], g["tap_glow_g"].T, atol=1e-4)
    for b in (17, 8, 0):
        np.testing.assert_allclose(taps[f"glow_z_after_block{b}"].numpy(), g[f"tap_glow_z_after_block{b}"].T, atol=3e-4)
    mel = out["mel"][0].numpy()
    assert mel.shape == g["mel"].shape
    np.testing.assert_allclose(mel, g["mel"], atol=3e-4)
    assert np.abs(mel - g["mel"]).mean() < 1e-4


def test_ragged_batch_equals_one_by_one_and_goldens(emu, ac_sd):
    """Batch-1 semantics under batching: L20 cases incl. an odd frame count, mixed gold/predicted are
    not mixable in one call, so batch the two predicted-duration cases and compare each with its golden."""
    gs = [_gold("L7_pred"), _gold("L20_pred")]
    eng = engine.AcousticEngine(ac_sd, "cpu")
    texts, embs, langs, zs = _inputs(gs)
    out = eng.forward(texts, embs, langs, z_noise=zs)
    for u, g in enumerate(gs):
        assert np.array_equal(out["durations"][u].numpy(), g["durations"])
        assert out["mel"][u].shape == g["mel"].shape
        np.testing.assert_allclose(out["mel"][u].numpy(), g["mel"], atol=3e-4)


def test_gold_durations_controls_and_odd_length(emu, ac_sd):
    g = _gold("L20_gold_odd")
    eng = engine.AcousticEngine(ac_sd, "cpu")
    texts, embs, langs, zs = _inputs([g])
    kw = json.loads(str(g["ctrl"]))
    out = eng.forward(texts, embs, langs, z_noise=zs, durations=[torch.from_numpy(g["gold_durations"])], **kw)
    assert out["mel"][0].shape[0] == int(g["gold_durations"].sum()) - 1
    np.testing.assert_allclose(out["mel"][0].numpy(), g["mel"], atol=3e-4)
    np.testing.assert_allclose(out["pitch"][0].numpy(), g["pitch"], atol=3e-5)

    g = _gold("L20_ctrl")
    texts, embs, langs, zs = _inputs([g])
    out = eng.forward(texts, embs, langs, z_noise=zs, **json.loads(str(g["ctrl"])))
    assert np.array_equal(out["durations"][0].numpy(), g["durations"])
    np.testing.assert_allclose(out["energy"][0].numpy(), g["energy"], atol=3e-5)
    np.testing.assert_allclose(out["mel"][0].numpy(), g["mel"], atol=3e-4)


@pytest.mark.parametrize("kind", ["hifigan", "bigvgan"])
def test_vocoder_engine_matches_reference_golden(emu, kind):
    g = _gold("L7_pred")
    sd = fw.hifigan_state_dict() if kind == "hifigan" else fw.bigvgan_state_dict()
    voc = engine.VocoderEngine(sd, kind, "cpu")
    from ims_toucan_prosody_variance_amd.ragged import Ragged
    mel = torch.from_numpy(g["mel"]).contiguous()
    taps = {}
    wav, rag = voc.forward(mel, Ragged([mel.shape[0]], "cpu"), taps)
    assert wav.numel() == 384 * mel.shape[0]
    np.testing.assert_allclose(taps["voc_stage0"].numpy(), g[f"tap_{kind}_stage0"].T, atol=1e-4)
    np.testing.assert_allclose(wav.numpy(), g["wav_" + kind], atol=2e-4)


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("kind", ["hifigan", "bigvgan"])
def test_bf16_vocoder_engine_with_fused_residual_steps(emu, kind, precision):
    """16-bit configurations (fused tts_resblock_step for C <= 128) stay within the stated bf16 tolerance of the fp32 golden."""
    g = _gold("L7_pred")
    sd = fw.hifigan_state_dict() if kind == "hifigan" else fw.bigvgan_state_dict()
    voc = engine.VocoderEngine(sd, kind, "cpu", precision=precision)
    from ims_toucan_prosody_variance_amd.ragged import Ragged
    mel = torch.from_numpy(g["mel"]).contiguous()
    wav, rag = voc.forward(mel, Ragged([mel.shape[0]], "cpu"))
    assert emu.calls.get("resblock_step", 0) == 27  # 3 stages (C <= 128) x 3 blocks x 3 dilations
    err = np.abs(wav.numpy() - g["wav_" + kind])
    assert err.mean() < (2e-2 if precision == "bf16" else 4e-3), float(err.mean())


@pytest.mark.parametrize("precision,bound", [("bf16", 0.05), ("f16", 0.01)])
def test_16bit_acoustic_engine_within_stated_tolerance(emu, ac_sd, precision, bound):
    """bf16 / fp16 MFMA configurations of the acoustic model (fp32 statistics, flow state and predictors) against the fp32 golden;
    the golden's own durations are passed as gold durations so the frame count cannot move."""
    g = _gold("L7_pred")
    eng = engine.AcousticEngine(ac_sd, "cpu", precision=precision)
    texts, embs, langs, zs = _inputs([g])
    out = eng.forward(texts, embs, langs, z_noise=zs, durations=[torch.from_numpy(g["durations"])])
    err = np.abs(out["mel"][0].numpy() - g["mel"])
    assert err.mean() < bound, (precision, float(err.mean()))


def test_bigvgan_checkpoint_with_stored_antialias_filter(emu):
    """Real BigVGAN checkpoints carry the Activation1d filters as buffers (``...upsample.filter`` /
    ``...downsample.lowpass.filter``); when present they replace the restated Kaiser-sinc design, in the engine and in the oracle."""
    from oracle import toucan_oracle as orc
    from ims_toucan_prosody_variance_amd import packing
    from ims_toucan_prosody_variance_amd.ragged import Ragged
    sd = fw.bigvgan_state_dict()
    assert packing.stored_antialias_filter(sd) is None
    f = packing.kaiser_sinc_filter12().astype(np.float64)
    f = f * (1.0 + 0.05 * np.cos(np.arange(12)))  # a different (still low-pass, unit-sum) filter
    f = (f / f.sum()).astype(np.float32)
    sd2 = dict(sd)
    for b in range(12):
        for a in range(6):
            sd2[f"resblocks.{b}.activations.{a}.upsample.filter"] = f.reshape(1, 1, 12)
            sd2[f"resblocks.{b}.activations.{a}.downsample.lowpass.filter"] = f.reshape(1, 1, 12)
    sd2["activation_post.upsample.filter"] = f.reshape(1, 1, 12)
    sd2["activation_post.downsample.lowpass.filter"] = f.reshape(1, 1, 12)
    np.testing.assert_array_equal(packing.stored_antialias_filter(sd2), f)
    mel = torch.from_numpy(_gold("L7_pred")["mel"]).contiguous()
    voc = engine.VocoderEngine(sd2, "bigvgan", "cpu")
    wav, _ = voc.forward(mel, Ragged([mel.shape[0]], "cpu"))
    want = orc.VocoderOracle(sd2, "bigvgan")(mel.t().contiguous()).numpy()
    np.testing.assert_allclose(wav.numpy(), want, atol=2e-4)
    base, _ = engine.VocoderEngine(sd, "bigvgan", "cpu").forward(mel, Ragged([mel.shape[0]], "cpu"))
    assert np.abs(wav.numpy() - base.numpy()).max() > 1e-3  # the stored filter is really the one in use
    sd3 = dict(sd2)
    sd3["resblocks.3.activations.1.downsample.lowpass.filter"] = packing.kaiser_sinc_filter12().reshape(1, 1, 12)
    with pytest.raises(NotImplementedError):
        engine.VocoderEngine(sd3, "bigvgan", "cpu")
