"""Parity of the PRODUCT path on a real MI355X - the code a user of the drop-in and bench.py actually execute:
 (a1) ``ToucanTTSInterface(device="cuda")(phones, input_is_phones=True)`` against the CPU oracle (acoustic + vocoder) on the same
      fixture checkpoints (ToucanTTSInterface.py:132-169: phoneme tensor, default embedding, language id, mel transpose, even-frame
      truncation by the flow's squeeze), incl. the cloner-style gold-prosody call (UtteranceCloner.py:163);
 (c)  the exact path bench.py times - NativePipeline.pack_inputs / squeeze_noise / forward(packed=, z_sq=) on one HIP stream beside
      vocode() on another, batch 32 x 128 phonemes, bf16 - against the Python sequencer (engine.py), bit for bit;
 (d)  GPU twins of tests/test_edge_cases_cpu.py: the reference's corner cases through the HIP kernels end to end, against the oracle.
Tolerances (fp32 path, as in test_gpu_e2e.py): durations bit exact, mel max-abs 5e-4 / L1 1e-4, waveform max-abs 1e-3."""
import json
import os

import numpy as np
import pytest
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw, native, synthetic as syn
from ims_toucan_prosody_variance_amd.phonemes import IDX, phone_table, phones_to_features
from oracle import toucan_oracle as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"
N_LANG = 20


def _gold(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


# ---- (a1) the interface -----------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def models_dir(tmp_path_factory):
    from ims_toucan_prosody_variance_amd import interface
    d = tmp_path_factory.mktemp("Models")
    interface.write_fixture_checkpoints(str(d), n_lang=N_LANG)
    return str(d)


@pytest.fixture(scope="module")
def oracles():
    return dict(ac=orc.AcousticOracle(fw.acoustic_state_dict(n_lang=N_LANG)),
                hifigan=orc.VocoderOracle(fw.hifigan_state_dict(), "hifigan"),
                bigvgan=orc.VocoderOracle(fw.bigvgan_state_dict(), "bigvgan"))


PHONES = "~wˈʌns əpˈɑːn ɐ mˈɪdnaɪt dɹˈɪɹi, wˈaɪl aɪ pˈɑːndɚd~#"


@pytest.mark.parametrize("faster_vocoder", [True, False])
def test_interface_call_matches_the_oracle(models_dir, oracles, monkeypatch, faster_vocoder):
    """tts(phones, input_is_phones=True, <all four scales>) == VocoderOracle(AcousticOracle(...)): predicted durations bit exact, mel and
    waveform within the fp32 tolerances.  The PostFlow noise (torch.randn at Glow.py:363 in the reference) is the one injected input."""
    from ims_toucan_prosody_variance_amd import interface
    monkeypatch.setattr(interface, "MODELS_DIR", models_dir)
    from InferenceInterfaces.ToucanTTSInterface import ToucanTTSInterface
    tts = ToucanTTSInterface(device="cuda", tts_model_path="Meta", faster_vocoder=faster_vocoder)
    tts.set_language("en")
    assert tts.pipe is not None, "the stage API must be the path behind the interface on a GPU"
    kind = "hifigan" if faster_vocoder else "bigvgan"
    feats = phones_to_features(PHONES)  # (bit exact against the reference's string_to_tensor: tests/test_frontend_golden.py)
    emb = torch.from_numpy(fw.default_utterance_embedding())
    kw = dict(duration_scaling_factor=1.1, pitch_variance_scale=1.2, energy_variance_scale=0.9, pause_duration_scaling_factor=1.3)
    oa = oracles["ac"]
    probe = oa(torch.from_numpy(feats), emb, syn.LANG_EN, run_postflow=False, **kw)
    T = int(probe["durations"].sum())
    z = torch.from_numpy(syn.postflow_noise(555, T))
    wav = tts(PHONES, input_is_phones=True, z_noise=z, **kw)
    o = oa(torch.from_numpy(feats), emb, syn.LANG_EN, z_noise=z, **kw)
    w = oracles[kind](o["mel"].t().contiguous())  # ToucanTTSInterface.py:168: the mel goes to the vocoder as [80, T]
    assert np.array_equal(tts.last_durations[0].cpu().numpy(), o["durations"].numpy())
    np.testing.assert_allclose(tts.last_pitch[0].cpu().numpy(), o["pitch"].numpy(), atol=5e-5)
    np.testing.assert_allclose(tts.last_energy[0].cpu().numpy(), o["energy"].numpy(), atol=5e-5)
    mel = tts.last_mel[0].cpu().numpy()
    assert mel.shape == tuple(o["mel"].shape) == (T - T % 2, 80)
    err = np.abs(mel - o["mel"].numpy())
    assert err.max() < 5e-4 and err.mean() < 1e-4, (float(err.max()), float(err.mean()))
    assert wav.is_cuda and wav.dim() == 1 and wav.numel() == w.numel() == 384 * (T - T % 2)
    assert np.abs(wav.cpu().numpy() - w.numpy()).max() < 1e-3


def test_interface_gold_prosody_call_matches_the_oracle(models_dir, oracles, monkeypatch):
    """The cloner-style call (UtteranceCloner.py:163): gold durations, pitch and energy through ``forward`` / ``read_to_file``'s
    lists, with variance scaling on top - the fork's namesake feature - against the oracle; gold tensors are not written to."""
    from ims_toucan_prosody_variance_amd import interface
    monkeypatch.setattr(interface, "MODELS_DIR", models_dir)
    tts = interface.ToucanTTSInterface(device="cuda", tts_model_path="Meta", faster_vocoder=True)
    feats = phones_to_features(PHONES)
    L = feats.shape[0]
    emb = torch.from_numpy(fw.default_utterance_embedding())
    dur = torch.from_numpy(syn.ragged_durations(41, feats))
    gp = torch.from_numpy((0.2 + 1.5 * fw.uniform01("t.gp", L, 77)).astype(np.float32)).reshape(L, 1)  # [L, 1] like the reference's callers
    ge = torch.from_numpy((0.1 + 2.0 * fw.uniform01("t.ge", L, 78)).astype(np.float32)).reshape(L, 1)
    gp0, ge0, d0 = gp.clone(), ge.clone(), dur.clone()
    T = int(dur.sum())
    z = torch.from_numpy(syn.postflow_noise(556, T))
    kw = dict(pitch_variance_scale=1.4, energy_variance_scale=0.6)
    wav = tts(PHONES, input_is_phones=True, durations=dur, pitch=gp, energy=ge, z_noise=z, **kw)
    assert torch.equal(gp, gp0) and torch.equal(ge, ge0) and torch.equal(dur, d0)
    o = oracles["ac"](torch.from_numpy(feats), emb, syn.LANG_EN, z_noise=z, durations=dur, pitch=gp, energy=ge, **kw)
    assert np.array_equal(tts.last_durations[0].cpu().numpy(), o["durations"].numpy())
    np.testing.assert_allclose(tts.last_pitch[0].cpu().numpy(), o["pitch"].numpy(), atol=1e-5)
    np.testing.assert_allclose(tts.last_energy[0].cpu().numpy(), o["energy"].numpy(), atol=1e-5)
    unvoiced = feats[:, IDX["voiced"]] == 0
    assert unvoiced.any() and not np.allclose(o["pitch"].numpy()[~unvoiced], gp0.reshape(-1).numpy()[~unvoiced])
    err = np.abs(tts.last_mel[0].cpu().numpy() - o["mel"].numpy())
    assert err.max() < 5e-4 and err.mean() < 1e-4, (float(err.max()), float(err.mean()))
    w = oracles["hifigan"](o["mel"].t().contiguous())
    assert np.abs(wav.cpu().numpy() - w.numpy()).max() < 1e-3


# ---- (c) the path bench.py times ---------------------------------------------------------------------------------------------------
def test_bench_path_full_size_bf16_two_streams_equals_the_python_sequencer():
    """BASELINE.json configs[2] at full size the way bench.py runs it: inputs packed once (pack_inputs / squeeze_noise), three steps
    with the acoustic model of step k+1 on one HIP stream beside the vocoder of step k on another (bench.py's step_overlapped
    through the same helper), bf16.  Every step's mel and waveform equal the Python sequencer's (engine.py: the path the golden /
    oracle parity tests drive) bit for bit, and utterance 0 - the reference golden L128_gold5 - keeps the stated bf16 tolerance."""
    import bench
    g = _gold("L128_gold5")
    B, L = 32, 128
    texts = [torch.from_numpy(g["text"])] + [torch.from_numpy(syn.utterance_features(u, L, word_boundaries=False)) for u in range(1, B)]
    embs = torch.stack([torch.from_numpy(g["utt_emb"])] + [torch.from_numpy(syn.utterance_embedding(u)) for u in range(1, B)])
    durs = [torch.from_numpy(g["gold_durations"]).to(torch.int32)] + [torch.full((L,), 5, dtype=torch.int32) for _ in range(1, B)]
    zs = [torch.from_numpy(g["z"])] + [torch.from_numpy(syn.postflow_noise(u, 5 * L)) for u in range(1, B)]
    langs = [int(g["lang_id"])] * B
    ac_sd, voc_sd = fw.acoustic_state_dict(), fw.bigvgan_state_dict()
    ac = engine.AcousticEngine(ac_sd, DEV, precision="bf16")
    voc = engine.VocoderEngine(voc_sd, "bigvgan", DEV, precision="bf16")
    ref = ac.forward(texts, embs, langs, durations=durs, z_noise=zs)
    wref, rw = voc.forward(ref["mel_packed"], ref["rag_mel"])
    torch.cuda.synchronize()
    pipe = native.NativePipeline(ac_sd, voc_sd, "bigvgan", DEV, precision="bf16")
    dev = torch.device(DEV)
    packed = pipe.pack_inputs([t.to(dev) for t in texts], embs.to(dev), langs, durations=[d.to(dev) for d in durs])
    z_sq = pipe.squeeze_noise([z.to(dev) for z in zs], [int(d.sum()) for d in durs])
    runner = bench.TwoStreamRunner(pipe, packed, z_sq, {}, dev)
    results = [runner.step() for _ in range(3)]
    torch.cuda.synchronize()
    for k, (out, wav) in enumerate(results):
        for u in range(B):
            assert torch.equal(out["mel"][u], ref["mel"][u]), f"step {k} utterance {u}: mel differs from the Python sequencer"
            b0, n = rw.begins[u], rw.lengths[u]
            assert torch.equal(wav[b0:b0 + n], wref[b0:b0 + n]), f"step {k} utterance {u}: waveform differs"
    err = np.abs(results[-1][0]["mel"][0].cpu().numpy() - g["mel"])
    assert err.mean() < 0.05, float(err.mean())  # TOL_16BIT["bf16"] of test_gpu_e2e.py
    v = bench.verify_against_single_run(pipe, results[-1][0], results[-1][1], packed, z_sq, {}, B)
    assert v["mel_bit_identical"] and v["wav_bit_identical"], v


# ---- (d) GPU twins of the CPU edge cases ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sd():
    return fw.acoustic_state_dict(n_lang=N_LANG)


@pytest.fixture(scope="module")
def paths(sd):
    """Both sequencers of the HIP path + the oracle."""
    hf = fw.hifigan_state_dict()
    return dict(oracle=orc.AcousticOracle(sd), voc_oracle=orc.VocoderOracle(hf, "hifigan"),
                engine=engine.AcousticEngine(sd, DEV), native=native.NativePipeline(sd, hf, "hifigan", DEV))


def _run_all(paths, feats_list, embs, **kw):
    oracle = paths["oracle"]
    per = lambda i: {k: (v[i] if isinstance(v, list) else v) for k, v in kw.items()}
    probe = [oracle(torch.from_numpy(f), torch.from_numpy(e), syn.LANG_EN, run_postflow=False, **per(i)) for i, (f, e) in enumerate(zip(feats_list, embs))]
    zs = [torch.from_numpy(syn.postflow_noise(50 + i, max(2, int(p["upsampled"].shape[0])))) for i, p in enumerate(probe)]
    args = ([torch.from_numpy(f) for f in feats_list], torch.from_numpy(np.stack(embs)), [syn.LANG_EN] * len(feats_list))
    outs = [paths["engine"].forward(*args, z_noise=zs, **kw), paths["native"].forward(*args, z_noise=zs, **kw)]
    refs = [oracle(torch.from_numpy(f), torch.from_numpy(e), syn.LANG_EN, z_noise=z, **per(i)) for i, (f, e, z) in enumerate(zip(feats_list, embs, zs))]
    return outs, refs


def _check(outs, refs, paths=None, atol=5e-4):
    for out in outs:
        for u, r in enumerate(refs):
            assert np.array_equal(out["durations"][u].cpu().numpy(), r["durations"].numpy())
            assert tuple(out["mel"][u].shape) == tuple(r["mel"].shape)
            err = np.abs(out["mel"][u].cpu().numpy() - r["mel"].numpy())
            assert err.max() < atol and err.mean() < 1e-4, (u, float(err.max()), float(err.mean()))
    if paths is not None:  # the waveform of the stage API's vocoder on these tiny mels
        out = outs[1]
        for u, r in enumerate(refs):
            w = paths["voc_oracle"](r["mel"].t().contiguous()).numpy()
            b0, n = out["wav_spans"][u]
            assert n == w.shape[0]
            assert np.abs(out["wav"][b0:b0 + n].cpu().numpy() - w).max() < 1e-3


def test_gpu_single_phoneme_and_two_frame_utterances_in_one_batch(paths):
    table = phone_table()
    feats = [table["a"][None, :].copy(), syn.utterance_features(7, 9)]
    embs = [syn.utterance_embedding(1), syn.utterance_embedding(2)]
    durs = [torch.tensor([2]), torch.tensor([1, 0, 3, 2, 0, 1, 4, 1, 2])]
    outs, refs = _run_all(paths, feats, embs, durations=durs)
    assert refs[0]["mel"].shape[0] == 2
    _check(outs, refs, paths)


def test_gpu_all_zero_durations_fall_back_to_one_frame_per_phoneme(paths):
    """Layers/LengthRegulator.py:52-53 end to end through the HIP kernels."""
    outs, refs = _run_all(paths, [syn.utterance_features(3, 6)], [syn.utterance_embedding(3)], durations=[torch.zeros(6, dtype=torch.long)])
    assert refs[0]["mel"].shape[0] == 6
    _check(outs, refs, paths)


def test_gpu_word_boundaries_get_zero_frames_and_unvoiced_zero_pitch(paths):
    """InferenceToucanTTS.py:214-222 on a real phoneme string with predicted durations: ' ' -> 0 frames, unvoiced -> pitch 0 (then
    shifted by _scale_variance and clamped), '~' -> energy 0."""
    feats = phones_to_features("~ˈaɪ sˈi tˈu~#")
    outs, refs = _run_all(paths, [feats], [syn.utterance_embedding(4)], pitch_variance_scale=1.3, pause_duration_scaling_factor=0.7)
    wb = feats[:, IDX["word_boundary"]] == 1
    for out in outs:
        d = out["durations"][0].cpu().numpy()
        assert wb.any() and (d[wb] == 0).all()
        np.testing.assert_allclose(out["pitch"][0].cpu().numpy(), refs[0]["pitch"].numpy(), atol=5e-5)
        np.testing.assert_allclose(out["energy"][0].cpu().numpy(), refs[0]["energy"].numpy(), atol=5e-5)
    _check(outs, refs, paths)


def test_gpu_1024_phoneme_utterance_regrows_the_position_tables(paths):
    """1 024 phonemes (the handle's initial table size) x 2 frames = 2 048 frames: encoder at the table's limit, decoder beyond it -
    the tables regrow inside forward() - against the oracle; then a short utterance again on the regrown tables."""
    L = 1024
    feats = [syn.utterance_features(11, L, word_boundaries=False)]
    durs = [torch.full((L,), 2, dtype=torch.long)]
    outs, refs = _run_all(paths, feats, [syn.utterance_embedding(11)], durations=durs)
    assert refs[0]["mel"].shape[0] == 2048
    _check(outs, refs)
    outs, refs = _run_all(paths, [syn.utterance_features(3, 6)], [syn.utterance_embedding(3)], durations=[torch.tensor([1, 2, 0, 3, 1, 2])])
    _check(outs, refs, paths)


def test_gpu_fp32_durations_do_not_depend_on_the_batch(paths):
    """Predicted durations are a rounding of exp(log d): the duration predictor's arithmetic must not depend on the grid a launch
    happens to get, or an utterance's frame count could differ between B = 1, B = 32 and an N-rank shard.  32 utterances of
    different lengths: durations (and pitch / energy) of the batch == the same utterance alone, bit for bit, through both sequencers."""
    B = 32
    Ls = [16 + (7 * u) % 48 for u in range(B)]
    feats = [torch.from_numpy(syn.utterance_features(700 + u, L)) for u, L in enumerate(Ls)]
    embs = torch.from_numpy(np.stack([syn.utterance_embedding(700 + u) for u in range(B)]))
    for name in ("engine", "native"):
        p = paths[name]
        kw = dict(run_postflow=False)
        if name == "native":
            kw["vocode"] = False
        big = p.forward(feats, embs, [syn.LANG_EN] * B, **kw)
        d_big = [d.clone() for d in big["durations"]]
        p_big = [x.clone() for x in big["pitch"]]
        for u in range(0, B, 5):
            one = p.forward([feats[u]], embs[u:u + 1], [syn.LANG_EN], **kw)
            assert torch.equal(one["durations"][0], d_big[u]), (name, u)
            assert torch.equal(one["pitch"][0], p_big[u]), (name, u)
