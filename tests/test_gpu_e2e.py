"""End-to-end parity on a real MI355X: the HIP path (engine.py -> libtoucan_hip.so) against
 (1) the committed golden vectors captured from the reference's own modules (tests/golden/*.npz), and
 (2) the CPU oracle run live on the same seeded inputs.
Stated tolerances (fp32 path): mel max-abs 5e-4 on |mel| ~ 4 (L1 < 1e-4, the north-star bound);
waveform max-abs 5e-4 on |wav| <= 1.  bf16 vocoder: waveform mean-abs error < 2e-2."""
import json
import os

import numpy as np
import pytest
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import engine, fixture_weights as fw, synthetic as syn
from ims_toucan_prosody_variance_amd.ragged import Ragged

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _gold(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="module")
def acoustic():
    assert torch.cuda.is_available()
    return engine.AcousticEngine(fw.acoustic_state_dict(), DEV)


@pytest.fixture(scope="module")
def vocoders():
    return {"hifigan": engine.VocoderEngine(fw.hifigan_state_dict(), "hifigan", DEV),
            "bigvgan": engine.VocoderEngine(fw.bigvgan_state_dict(), "bigvgan", DEV)}


def _inputs(gs):
    texts = [torch.from_numpy(g["text"]) for g in gs]
    embs = torch.stack([torch.from_numpy(g["utt_emb"]) for g in gs])
    langs = [int(g["lang_id"]) for g in gs]
    zs = [torch.from_numpy(g["z"]) for g in gs]
    return texts, embs, langs, zs


def _check_mel(mel, g, name=""):
    mel = mel.cpu().numpy()
    assert mel.shape == g["mel"].shape, name
    err = np.abs(mel - g["mel"])
    assert err.max() < 5e-4, f"{name}: mel max abs err {err.max():.3e}"
    assert err.mean() < 1e-4, f"{name}: mel L1 {err.mean():.3e}"


@pytest.mark.parametrize("name", ["L7_pred", "L20_pred", "L20_ctrl", "L20_gold_odd", "L20_gold_prosody", "L128_gold5"])
def test_acoustic_matches_reference_golden(acoustic, name):
    g = _gold(name)
    texts, embs, langs, zs = _inputs([g])
    kw = json.loads(str(g["ctrl"]))
    if "gold_durations" in g.files:
        kw["durations"] = [torch.from_numpy(g["gold_durations"])]
    if "gold_pitch" in g.files:  # gold prosody overrides (InferenceToucanTTS.py:209-210; UtteranceCloner.py:163): zeroed and scaled like predictions
        kw["pitch"], kw["energy"] = [torch.from_numpy(g["gold_pitch"])], [torch.from_numpy(g["gold_energy"])]
    taps = {}
    out = acoustic.forward(texts, embs, langs, z_noise=zs, taps=taps, **kw)
    assert np.array_equal(out["durations"][0].cpu().numpy(), g["durations"]), "durations must be bit exact"
    np.testing.assert_allclose(out["pitch"][0].cpu().numpy(), g["pitch"], atol=5e-5)
    np.testing.assert_allclose(out["energy"][0].cpu().numpy(), g["energy"], atol=5e-5)
    np.testing.assert_allclose(taps["enc_out"].cpu().numpy(), g["enc_out"], atol=5e-5)
    n = g["decoded"].shape[0]
    np.testing.assert_allclose(out["decoded_packed"].cpu().numpy()[:n], g["decoded"], atol=2e-4)
    _check_mel(out["mel"][0], g, name)


def test_ragged_batch_matches_each_utterance_run_alone(acoustic):
    """batch of 4 ragged utterances == the reference run once per utterance (goldens R128/R97/R64/R20)."""
    names = ["R128", "R97", "R64", "R20"]
    gs = [_gold(n) for n in names]
    texts, embs, langs, zs = _inputs(gs)
    out = acoustic.forward(texts, embs, langs, z_noise=zs, durations=[torch.from_numpy(g["gold_durations"]) for g in gs])
    for u, (n, g) in enumerate(zip(names, gs)):
        _check_mel(out["mel"][u], g, n)


@pytest.mark.parametrize("kind", ["hifigan", "bigvgan"])
def test_vocoder_matches_reference_golden(vocoders, kind):
    for name in ("L7_pred", "L20_pred"):
        g = _gold(name)
        mel = torch.from_numpy(g["mel"]).to(DEV).contiguous()
        taps = {}
        wav, rag = vocoders[kind].forward(mel, Ragged([mel.shape[0]], DEV), taps)
        wav = wav.cpu().numpy()
        assert wav.shape[0] == 384 * g["mel"].shape[0]
        if f"tap_{kind}_stage0" in g.files:
            np.testing.assert_allclose(taps["voc_stage0"].cpu().numpy(), g[f"tap_{kind}_stage0"].T, atol=2e-4)
        assert np.abs(wav - g["wav_" + kind]).max() < 5e-4, (kind, name)


@pytest.mark.parametrize("kind", ["hifigan", "bigvgan"])
def test_vocoder_full_length_against_golden_checksums(vocoders, kind):
    g = _gold("L128_gold5")
    mel = torch.from_numpy(g["mel"]).to(DEV).contiguous()
    wav, _ = vocoders[kind].forward(mel, Ragged([mel.shape[0]], DEV))
    wav = wav.cpu().numpy()
    assert wav.shape[0] == int(g["wav_len"])
    assert np.abs(wav[:8192] - g[f"wav_{kind}_head"]).max() < 5e-4
    assert np.abs(wav[-8192:] - g[f"wav_{kind}_tail"]).max() < 5e-4
    assert abs(float(np.abs(wav.astype(np.float64)).sum()) - float(g[f"wav_{kind}_abs_sum"])) < 1e-4 * wav.shape[0]


@pytest.mark.parametrize("kind", ["hifigan", "bigvgan"])
def test_vocoder_ragged_batch_equals_one_by_one(vocoders, kind):
    gs = [_gold(n) for n in ("L20_pred", "L7_pred", "L20_gold_odd")]
    rag = Ragged([g["mel"].shape[0] for g in gs], DEV, align=2)
    mel = torch.zeros(rag.total_rows, 80)
    for g, b in zip(gs, rag.begins):
        mel[b:b + g["mel"].shape[0]] = torch.from_numpy(g["mel"])
    wav, rw = vocoders[kind].forward(mel.to(DEV), rag)
    wav = wav.cpu().numpy()
    for g, b, n in zip(gs, rw.begins, rw.lengths):
        if "wav_" + kind in g.files:
            assert np.abs(wav[b:b + n] - g["wav_" + kind]).max() < 5e-4


def test_against_live_oracle_with_device_side_control_path(acoustic, vocoders):
    """Seeded synthetic batch (SURVEY 8d parity shape), predicted durations + prosody scaling, vs the CPU oracle."""
    from oracle import toucan_oracle as orc
    oa = orc.AcousticOracle(fw.acoustic_state_dict())
    ov = orc.VocoderOracle(fw.hifigan_state_dict(), "hifigan")
    us, Ls = [300, 301, 302], [48, 21, 33]
    feats = [syn.utterance_features(u, L) for u, L in zip(us, Ls)]
    embs = np.stack([syn.utterance_embedding(u) for u in us])
    kw = dict(duration_scaling_factor=0.9, pitch_variance_scale=1.2, energy_variance_scale=0.8, pause_duration_scaling_factor=1.3)
    probe = [oa(torch.from_numpy(f), torch.from_numpy(e), syn.LANG_EN, run_postflow=False, **kw) for f, e in zip(feats, embs)]
    zs = [torch.from_numpy(syn.postflow_noise(u, int(p["durations"].sum()))) for u, p in zip(us, probe)]
    out = acoustic.forward([torch.from_numpy(f) for f in feats], torch.from_numpy(embs), [syn.LANG_EN] * 3, z_noise=zs, **kw)
    wav, rw = vocoders["hifigan"].forward(out["mel_packed"], out["rag_mel"])
    wav = wav.cpu().numpy()
    for u in range(3):
        o = oa(torch.from_numpy(feats[u]), torch.from_numpy(embs[u]), syn.LANG_EN, z_noise=zs[u], **kw)
        assert np.array_equal(out["durations"][u].cpu().numpy(), o["durations"].numpy())
        err = np.abs(out["mel"][u].cpu().numpy() - o["mel"].numpy())
        assert err.max() < 5e-4 and err.mean() < 1e-4
        w = ov(o["mel"].t().contiguous()).numpy()
        b, n = rw.begins[u], rw.lengths[u]
        assert np.abs(wav[b:b + n] - w).max() < 1e-3


def test_bf16_vocoder_within_stated_tolerance():
    g = _gold("L20_pred")
    mel = torch.from_numpy(g["mel"]).to(DEV).contiguous()
    for kind, sd in (("hifigan", fw.hifigan_state_dict()), ("bigvgan", fw.bigvgan_state_dict())):
        voc = engine.VocoderEngine(sd, kind, DEV, bf16=True)
        wav, _ = voc.forward(mel, Ragged([mel.shape[0]], DEV))
        err = np.abs(wav.cpu().numpy() - g["wav_" + kind])
        assert err.mean() < 2e-2, (kind, float(err.mean()), float(err.max()))


def test_fused_snake_path_agrees_with_unfused(vocoders):
    g = _gold("L20_pred")
    mel = torch.from_numpy(g["mel"]).to(DEV).contiguous()
    voc = engine.VocoderEngine(fw.bigvgan_state_dict(), "bigvgan", DEV, fuse_snake=True)
    wav, _ = voc.forward(mel, Ragged([mel.shape[0]], DEV))
    assert np.abs(wav.cpu().numpy() - g["wav_bigvgan"]).max() < 5e-4


def test_bf16_acoustic_within_stated_tolerance():
    """configs[2] precision: bf16 MFMA GEMMs, fp32 activations/statistics.  Stated tolerance vs the fp32 reference golden:
    mel mean-abs error < 0.05 (|mel| ~ 4, i.e. ~1 %); gold durations so the frame count is identical."""
    g = _gold("L128_gold5")
    ac = engine.AcousticEngine(fw.acoustic_state_dict(), DEV, bf16=True)
    texts, embs, langs, zs = _inputs([g])
    out = ac.forward(texts, embs, langs, z_noise=zs, durations=[torch.from_numpy(g["gold_durations"])])
    err = np.abs(out["mel"][0].cpu().numpy() - g["mel"])
    print("bf16 acoustic: mel mean abs err", float(err.mean()), "max", float(err.max()))
    assert err.mean() < 0.05, float(err.mean())


def _configs2_batch(B=32, L=128):
    """configs[2] / configs[4] shape: B utterances x 128 phonemes, gold durations.  Utterance 0 is the reference golden
    L128_gold5 (565 frames), the others are the seeded synthetic utterances of bench.py (5 frames per phoneme -> 640 frames)."""
    g = _gold("L128_gold5")
    texts = [torch.from_numpy(g["text"])] + [torch.from_numpy(syn.utterance_features(u, L, word_boundaries=False)) for u in range(1, B)]
    embs = torch.stack([torch.from_numpy(g["utt_emb"])] + [torch.from_numpy(syn.utterance_embedding(u)) for u in range(1, B)])
    durs = [torch.from_numpy(g["gold_durations"]).to(torch.int32)] + [torch.full((L,), 5, dtype=torch.int32) for _ in range(1, B)]
    zs = [torch.from_numpy(g["z"])] + [torch.from_numpy(syn.postflow_noise(u, 5 * L)) for u in range(1, B)]
    return g, texts, embs, durs, zs


# Stated tolerances of the 16-bit configurations against the fp32 reference golden L128_gold5 (|mel| mean 4.1, max 63; |wav| <= 1).
# mel: mean-abs error; wav: mean-abs error of the vocoder alone on the golden mel (head and tail 8192 samples).
# Measured on MI355X (this test prints them, DESIGN.md section 4 records them): see MEASURED_16BIT in DESIGN.md.
TOL_16BIT = {"bf16": dict(mel_mean=0.05, wav_mean=2e-2), "f16": dict(mel_mean=0.01, wav_mean=4e-3)}


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_full_size_batch32_16bit_bigvgan_equals_per_utterance_runs(precision):
    """BASELINE.json configs[2] (bf16) and the single-GPU shard of configs[4] (fp16, pitch 1.3 / energy 0.7) at FULL size:
    batch 32 x 128 phonemes, 16-bit acoustic MFMA + 16-bit BigVGAN with fused residual steps and a 16-bit residual stream.
    (1) every utterance of the batch is BIT-IDENTICAL (mel and waveform) to the same utterance run alone (B = 1): batching
        never changes an utterance's arithmetic (same kernels, same K order, per-utterance tiles);
    (2) utterance 0 stays within the stated tolerance of the fp32 reference golden (mel), and the vocoder alone - fed the golden
        mel as utterance 0 of a 32-batch - within the stated waveform tolerance.  The measured errors are printed and written
        to gpurun_out/parity_16bit_<precision>.json."""
    g, texts, embs, durs, zs = _configs2_batch()
    B = len(texts)
    langs = [int(g["lang_id"])] * B
    kw = dict(pitch_variance_scale=1.3, energy_variance_scale=0.7) if precision == "f16" else {}
    ac = engine.AcousticEngine(fw.acoustic_state_dict(), DEV, precision=precision)
    voc = engine.VocoderEngine(fw.bigvgan_state_dict(), "bigvgan", DEV, precision=precision)
    assert voc.fuse_step and voc.store_bf16
    out = ac.forward(texts, embs, langs, durations=durs, z_noise=zs, **kw)
    wav, rw = voc.forward(out["mel_packed"], out["rag_mel"])
    mels = [m.clone() for m in out["mel"]]
    wavs = [wav[b:b + n].clone() for b, n in zip(rw.begins, rw.lengths)]
    pitch_b = [p.clone() for p in out["pitch"]]
    for u in range(B):
        o1 = ac.forward([texts[u]], embs[u:u + 1], [langs[u]], durations=[durs[u]], z_noise=[zs[u]], **kw)
        w1, r1 = voc.forward(o1["mel_packed"], o1["rag_mel"])
        assert torch.equal(o1["pitch"][0], pitch_b[u]), f"utterance {u}: pitch differs between B=32 and B=1"
        assert torch.equal(o1["mel"][0], mels[u]), f"utterance {u}: mel differs between B=32 and B=1 (max {float((o1['mel'][0] - mels[u]).abs().max()):.3e})"
        assert torch.equal(w1[: r1.lengths[0]], wavs[u]), f"utterance {u}: waveform differs between B=32 and B=1"
    rec = {"precision": precision, "batch": B}
    if precision == "bf16":  # the golden was captured without prosody scaling
        err = np.abs(mels[0].cpu().numpy() - g["mel"])
        rec.update(mel_mean_abs=float(err.mean()), mel_max_abs=float(err.max()), mel_ref_mean_abs=float(np.abs(g["mel"]).mean()))
        assert err.mean() < TOL_16BIT[precision]["mel_mean"], rec
    else:
        o0 = ac.forward([texts[0]], embs[:1], [langs[0]], durations=[durs[0]], z_noise=[zs[0]])
        err = np.abs(o0["mel"][0].cpu().numpy() - g["mel"])
        rec.update(mel_mean_abs=float(err.mean()), mel_max_abs=float(err.max()), mel_ref_mean_abs=float(np.abs(g["mel"]).mean()))
        assert err.mean() < TOL_16BIT[precision]["mel_mean"], rec
    # vocoder alone on the golden mel, inside the full batch
    mp = out["mel_packed"].clone()
    rm = out["rag_mel"]
    mp[rm.begins[0]:rm.begins[0] + rm.lengths[0]] = torch.from_numpy(g["mel"]).to(DEV)
    wav2, rw2 = voc.forward(mp, rm)
    w0 = wav2[rw2.begins[0]:rw2.begins[0] + rw2.lengths[0]].cpu().numpy()
    assert w0.shape[0] == int(g["wav_len"])
    eh, et = np.abs(w0[:8192] - g["wav_bigvgan_head"]), np.abs(w0[-8192:] - g["wav_bigvgan_tail"])
    rec.update(wav_mean_abs=float((eh.mean() + et.mean()) / 2), wav_max_abs=float(max(eh.max(), et.max())),
               wav_ref_mean_abs=float(np.abs(g["wav_bigvgan_head"]).mean()))
    print("16-bit parity at full size:", json.dumps(rec))
    os.makedirs(os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", f"parity_16bit_{precision}.json"), "w") as f:
        json.dump(rec, f)
    assert rec["wav_mean_abs"] < TOL_16BIT[precision]["wav_mean"], rec


def test_fp16_configuration_within_stated_tolerance():
    """configs[4] precision (fp16 MFMA GEMMs, fp32 statistics / flow state) on the small goldens: acoustic mel and both vocoders."""
    g = _gold("L20_pred")
    ac = engine.AcousticEngine(fw.acoustic_state_dict(), DEV, precision="f16")
    texts, embs, langs, zs = _inputs([g])
    out = ac.forward(texts, embs, langs, z_noise=zs, durations=[torch.from_numpy(g["durations"])])
    err = np.abs(out["mel"][0].cpu().numpy() - g["mel"])
    print("fp16 acoustic: mel mean abs err", float(err.mean()), "max", float(err.max()))
    assert err.mean() < TOL_16BIT["f16"]["mel_mean"]
    mel = torch.from_numpy(g["mel"]).to(DEV).contiguous()
    for kind, sd in (("hifigan", fw.hifigan_state_dict()), ("bigvgan", fw.bigvgan_state_dict())):
        voc = engine.VocoderEngine(sd, kind, DEV, precision="f16")
        wav, _ = voc.forward(mel, Ragged([mel.shape[0]], DEV))
        e = np.abs(wav.cpu().numpy() - g["wav_" + kind])
        print("fp16", kind, "wav mean abs err", float(e.mean()), "max", float(e.max()))
        assert e.mean() < TOL_16BIT["f16"]["wav_mean"], (kind, float(e.mean()))


def test_hip_graph_replay_is_bit_identical_to_eager():
    """use_graphs: the two shape-static halves of the acoustic pass and the vocoder are captured and replayed."""
    g = _gold("L20_pred")
    texts, embs, langs, zs = _inputs([g])
    eager = engine.AcousticEngine(fw.acoustic_state_dict(), DEV)
    graphed = engine.AcousticEngine(fw.acoustic_state_dict(), DEV, use_graphs=True)
    ve = engine.VocoderEngine(fw.hifigan_state_dict(), "hifigan", DEV)
    vg = engine.VocoderEngine(fw.hifigan_state_dict(), "hifigan", DEV, use_graphs=True)
    ref = eager.forward(texts, embs, langs, z_noise=zs)
    wref, rw = ve.forward(ref["mel_packed"], ref["rag_mel"])
    n = rw.lengths[0]
    for _ in range(3):  # first call captures, later calls replay
        out = graphed.forward(texts, embs, langs, z_noise=zs)
        w, _ = vg.forward(out["mel_packed"], out["rag_mel"])
        assert torch.equal(out["mel"][0], ref["mel"][0])
        assert torch.equal(out["durations_packed"], ref["durations_packed"])
        assert torch.equal(w[:n], wref[:n])  # rows beyond the utterance are alignment padding (never written)
    _check_mel(out["mel"][0], g, "graph replay")


def test_hip_graphs_survive_table_regrowth_and_layout_cache_turnover():
    """Graph replay after the buffers a capture read by raw pointer were replaced: a short utterance, then one with more than 256
    frames (the relative-position tables regrow and are re-allocated), then the short one again - and the same after Ragged's
    layout cache was emptied.  Replays must equal the eager result bit for bit (stale tables / freed tile tables would not)."""
    from ims_toucan_prosody_variance_amd import ragged
    gs, gl = _gold("L20_pred"), _gold("L128_gold5")
    eager = engine.AcousticEngine(fw.acoustic_state_dict(), DEV)
    graphed = engine.AcousticEngine(fw.acoustic_state_dict(), DEV, use_graphs=True)

    def run(eng, g):
        texts, embs, langs, zs = _inputs([g])
        kw = {"durations": [torch.from_numpy(g["gold_durations"])]} if "gold_durations" in g.files else {}
        return eng.forward(texts, embs, langs, z_noise=zs, **kw)["mel"][0].clone()

    ref_s, ref_l = run(eager, gs), run(eager, gl)
    assert torch.equal(run(graphed, gs), ref_s)      # captures with pmax = 256
    assert torch.equal(run(graphed, gl), ref_l)      # 565 frames: tables regrow
    assert torch.equal(run(graphed, gs), ref_s)      # must not replay against the released tables
    assert torch.equal(run(graphed, gs), ref_s)      # (replay of the re-captured graph)
    ragged.Ragged._cache.clear()                     # the layouts the graphs captured leave the cache ...
    junk = [torch.empty(1 << 20, device=DEV) for _ in range(8)]  # ... and freed blocks would be handed out again
    for j in junk:
        j.fill_(1e30)
    assert torch.equal(run(graphed, gl), ref_l)
    assert torch.equal(run(graphed, gs), ref_s)


def test_drop_in_interface_on_the_gpu(tmp_path, monkeypatch):
    """The reference's entry point (import path, ctor keywords, read_to_file) end to end on cuda with reference-format checkpoints."""
    import wave
    from ims_toucan_prosody_variance_amd import interface
    models = tmp_path / "Models"
    interface.write_fixture_checkpoints(str(models), n_lang=20)
    monkeypatch.setattr(interface, "MODELS_DIR", str(models))
    from InferenceInterfaces.ToucanTTSInterface import ToucanTTSInterface
    for faster in (True, False):
        tts = ToucanTTSInterface(device="cuda", tts_model_path="Meta", faster_vocoder=faster)
        tts.set_language("en")
        wav = tts("~həlˈoʊ wˈɜːld~#", input_is_phones=True)
        assert wav.is_cuda and wav.dim() == 1 and torch.isfinite(wav).all() and float(wav.abs().max()) <= 1.0
        frames = int(tts.last_durations[0].sum())
        assert wav.numel() == 384 * (frames - frames % 2)
        out = tmp_path / f"x{int(faster)}.wav"
        tts.read_to_file(["~həlˈoʊ~#", "", "~wˈɜːld~#"], str(out), silent=True, input_is_phones=True)
        with wave.open(str(out)) as f:
            assert f.getframerate() == 24000 and f.getnframes() > 3 * 10600
        both = tts.synthesize_batch(["~həlˈoʊ~#", "~wˈɜːld tˈu~#"])
        assert len(both) == 2 and all(w.is_cuda for w in both)
    import matplotlib
    matplotlib.use("Agg")
    monkeypatch.chdir(tmp_path)
    wav2, png = tts("~həlˈoʊ wˈɜːld~#", input_is_phones=True, return_plot_as_filepath=True)  # ToucanTTSInterface.py:171-226
    assert png == "tmp.png" and (tmp_path / "tmp.png").stat().st_size > 10_000 and wav2.is_cuda


@pytest.mark.parametrize("kind", ["hifigan", "bigvgan"])
def test_chunked_vocoder_is_bit_identical_on_a_long_utterance(kind):
    """Overlap-save vocoding (streaming.py): 5 120 frames (1 024 phonemes x 5, tools/long_utterance_check.py's case) cut into
    512-frame chunks with the default 24-frame halo == the whole mel vocoded at once, bit for bit, in fp32 and in the bf16
    configuration with fused residual steps - through both sequencers.  A halo below the stack's reach must NOT be identical
    (the test would be vacuous otherwise)."""
    from ims_toucan_prosody_variance_amd import native, streaming
    T = 5120
    mel = (torch.from_numpy(syn.postflow_noise(77, T)).t().contiguous() * 2.0).to(DEV)  # any mel-shaped signal will do
    sd = fw.hifigan_state_dict() if kind == "hifigan" else fw.bigvgan_state_dict()
    for precision in ("f32", "bf16"):
        voc = engine.VocoderEngine(sd, kind, DEV, precision=precision)
        whole, rw = voc.forward(mel, Ragged([T], DEV))
        whole = whole[: rw.lengths[0]]
        pieces = list(streaming.stream_vocode(voc.forward, mel, chunk_frames=512, max_batch=4))
        assert len(pieces) == 10 and all(p.numel() == 512 * 384 for p in pieces)
        assert torch.equal(torch.cat(pieces), whole), (kind, precision)
        short = streaming.chunked_vocode(voc.forward, mel, chunk_frames=512, halo_frames=4)
        assert short.shape == whole.shape and not torch.equal(short, whole)
    pipe = native.NativePipeline(fw.acoustic_state_dict(), sd, kind, DEV)
    w2 = streaming.chunked_vocode(pipe.vocode, mel, chunk_frames=640, max_batch=3)  # ragged last chunk, another batch size
    ref, rr = engine.VocoderEngine(sd, kind, DEV).forward(mel, Ragged([T], DEV))
    assert torch.equal(w2, ref[: rr.lengths[0]])
    assert streaming.REACH_FRAMES <= streaming.DEFAULT_HALO


def test_interface_stream_equals_one_shot_call(tmp_path, monkeypatch):
    from ims_toucan_prosody_variance_amd import interface
    models = tmp_path / "Models"
    interface.write_fixture_checkpoints(str(models), n_lang=20)
    monkeypatch.setattr(interface, "MODELS_DIR", str(models))
    tts = interface.ToucanTTSInterface(device="cuda", tts_model_path="Meta", faster_vocoder=False)
    phones = "~" + "wˈʌns əpˈɑːn ɐ mˈɪdnaɪt dɹˈɪɹi " * 6 + "~#"
    L = int(tts.text2phone.string_to_tensor(phones, input_phonemes=True).shape[0])
    dur = torch.full((L,), 6, dtype=torch.long)
    z = torch.randn(80, 6 * L, generator=torch.Generator().manual_seed(3)) * 0.8
    pieces = list(tts.stream(phones, input_is_phones=True, chunk_frames=128, durations=dur, z_noise=z))
    assert len(pieces) > 3
    one = tts.synthesize_batch([phones], durations=[dur], z_noise=[z])[0]
    assert torch.equal(torch.cat(pieces), one)


def test_file_reader_harness_end_to_end(tmp_path, monkeypatch):
    """run_phoneme_file_reader.py (counterpart of the reference's run_text_to_file_reader.py:8-41): the fourteen lines of the poem through the drop-in
    interface on the GPU (stage API), one ragged batch, into a 24 kHz file with 10 600 samples of silence around every sentence."""
    import importlib.util
    import wave
    from ims_toucan_prosody_variance_amd import interface
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    models = tmp_path / "Models"
    interface.write_fixture_checkpoints(str(models), n_lang=20)
    monkeypatch.setattr(interface, "MODELS_DIR", str(models))
    monkeypatch.chdir(tmp_path)
    spec = importlib.util.spec_from_file_location("reader_harness", os.path.join(root, "run_phoneme_file_reader.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.the_raven(version="test", model_id="Meta", exec_device="cuda", speed_over_quality=True)
    with wave.open(str(tmp_path / "audios" / "the_raven_test.wav")) as f:
        assert f.getframerate() == 24000 and f.getnchannels() == 1
        n = f.getnframes()
    assert n > 15 * 10600 and (n - 15 * 10600) % 768 == 0  # 14 sentences of an even number of frames, 15 silences


def test_native_library_is_the_one_loaded():
    from ims_toucan_prosody_variance_amd import capi
    import ctypes
    assert isinstance(capi.lib(), ctypes.CDLL)
    with open("/proc/self/maps") as f:
        assert "libtoucan_hip.so" in f.read()
