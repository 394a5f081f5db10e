"""Drop-in boundary on CPU (numpy ABI emulator underneath): reference-format checkpoints, read_to_file behaviour,
the reference's import path, the batch API and the 2-rank gloo path."""
import os
import sys
import wave

import numpy as np
import pytest
import torch

import ims_toucan_prosody_variance_amd  # noqa: F401
from ims_toucan_prosody_variance_amd import distributed as dd, interface
from tests import abi_emulator

PHONES_A = "~həlˈoʊ wˈɜːld~#"
PHONES_B = "~tˈɛst~#"


@pytest.fixture(scope="module")
def models_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("Models")
    interface.write_fixture_checkpoints(str(d), n_lang=20)
    return str(d)


@pytest.fixture()
def tts(monkeypatch, models_dir):
    abi_emulator.install(monkeypatch)
    monkeypatch.setattr(interface, "MODELS_DIR", models_dir)
    return interface.ToucanTTSInterface(device="cpu", tts_model_path="Meta", faster_vocoder=True)


def test_reference_import_path_and_signature():
    from InferenceInterfaces.ToucanTTSInterface import ToucanTTSInterface
    assert ToucanTTSInterface is interface.ToucanTTSInterface
    import inspect
    params = list(inspect.signature(ToucanTTSInterface.__init__).parameters)
    assert params == ["self", "device", "tts_model_path", "embedding_model_path", "vocoder_model_path", "faster_vocoder", "language"]
    fwd = list(inspect.signature(ToucanTTSInterface.forward).parameters)
    assert fwd[:12] == ["self", "text", "view", "duration_scaling_factor", "pitch_variance_scale", "energy_variance_scale",
                        "pause_duration_scaling_factor", "durations", "pitch", "energy", "input_is_phones", "return_plot_as_filepath"]


def test_forward_returns_24khz_wave_with_384_samples_per_frame(tts):
    torch.manual_seed(0)
    wav = tts(PHONES_B, input_is_phones=True)
    assert wav.dim() == 1 and wav.dtype == torch.float32
    frames = int(sum(int(d.sum()) for d in tts.last_durations))
    assert wav.numel() == 384 * (frames - frames % 2)
    assert float(wav.abs().max()) <= 1.0


def test_forward_draws_the_reference_figure(monkeypatch, tmp_path, tts):
    """return_plot_as_filepath=True: (wave, "tmp.png") like ToucanTTSInterface.py:222-226, the file written to the working directory."""
    import matplotlib
    matplotlib.use("Agg")
    monkeypatch.chdir(tmp_path)
    plain = tts(PHONES_B, input_is_phones=True)
    wav, path = tts(PHONES_B, input_is_phones=True, return_plot_as_filepath=True)
    assert path == "tmp.png" and (tmp_path / "tmp.png").stat().st_size > 10_000
    assert wav.shape == plain.shape and torch.isfinite(wav).all()  # (every call draws its own PostFlow noise, like the reference)
    assert tts(PHONES_B, input_is_phones=True, view=True).shape == plain.shape  # (plt.show() is a no-op on the Agg backend)


def test_missing_checkpoint_and_unsupported_paths_fail_loudly(monkeypatch, models_dir, tts):
    monkeypatch.setattr(interface, "MODELS_DIR", models_dir)
    with pytest.raises(FileNotFoundError):
        interface.ToucanTTSInterface(device="cpu", tts_model_path="DoesNotExist")
    with pytest.raises(RuntimeError, match="espeak"):
        tts("plain text needs a phonemizer")
    tts.set_language("de")
    assert int(tts.lang_id[0]) == 1
    tts.set_utterance_embedding(embedding=torch.ones(1, 64))
    assert tts.default_utterance_embedding.shape == (64,)


def test_default_cpu_device_is_refused_by_the_real_library(monkeypatch, models_dir):
    """The reference's default device="cpu" cannot be served (there is no CPU path): a clear error, not a GPU memory fault."""
    from ims_toucan_prosody_variance_amd import capi, engine
    monkeypatch.setattr(capi, "_LIB", None)  # the real libtoucan_hip.so (loads without a GPU)
    monkeypatch.setattr(interface, "MODELS_DIR", models_dir)
    with pytest.raises(capi.ToucanHipError, match="no CPU path"):
        engine.Ops("cpu")
    with pytest.raises(capi.ToucanHipError, match="device='cuda'"):
        interface.ToucanTTSInterface(tts_model_path="Meta")


def test_read_to_file_silence_layout_and_compat_mode(tts, tmp_path):
    torch.manual_seed(0)
    out = tmp_path / "a.wav"
    tts.read_to_file([PHONES_B, "   ", PHONES_B], str(out), silent=True, input_is_phones=True)
    with wave.open(str(out)) as f:
        assert f.getframerate() == 24000 and f.getsampwidth() == 2
        n = f.getnframes()
        data = np.frombuffer(f.readframes(n), dtype="<i2")
    assert len(tts.last_durations) == 2  # both sentences went through the engines as one ragged batch
    frames = int(tts.last_durations[0].sum())
    assert frames == int(tts.last_durations[1].sum())
    per = 384 * (frames - frames % 2)
    assert n == 10600 + 2 * (per + 10600)  # silence, sentence, silence, sentence, silence; blank string skipped
    assert not data[:10600].any() and not data[-10600:].any()
    out2 = tmp_path / "b.wav"
    tts.read_to_file([PHONES_B], str(out2), silent=True, input_is_phones=True, increased_compatibility_mode=True)
    with wave.open(str(out2)) as f:
        assert f.getframerate() == 48000 and f.getnframes() == 2 * (10600 + per + 10600)


def test_float2pcm_matches_reference_formula():
    x = np.array([-1.0, -0.5, 0.0, 0.5, 0.99997, 1.0, -0.99999, 3.0e-5, -3.0e-5, 2.0, -2.0], dtype=np.float32)
    assert interface.float2pcm(x).tolist() == [-32768, -16384, 0, 16384, 32767, 32767, -32767, 0, 0, 32767, -32768]  # truncation, saturation
    assert interface.float2pcm(np.array([-1.0, 0.0, 1.0]), "uint8").tolist() == [0, 128, 255]
    with pytest.raises(TypeError):
        interface.float2pcm(np.array([1, 2]))


def test_read_to_file_with_per_sentence_gold_prosody_and_ensemble(tts, tmp_path):
    """dur_list / pitch_list / energy_list shorter than text_list (the reference's zip_longest semantics): sentences with and
    without gold values are grouped into separate batches; the cloner-style ensemble averages several voices in one batch."""
    torch.manual_seed(0)
    L = int(tts.text2phone.string_to_tensor(PHONES_B, input_phonemes=True).shape[0])
    gold = torch.full((L,), 3, dtype=torch.long)
    out = tmp_path / "g.wav"
    tts.read_to_file([PHONES_B, PHONES_B], str(out), silent=True, input_is_phones=True, dur_list=[gold])
    with wave.open(str(out)) as f:
        n = f.getnframes()
    feats = tts.text2phone.string_to_tensor(PHONES_B, input_phonemes=True)
    gold_frames = int(gold[feats[:, 21] != 1].sum())
    pred_frames = int(tts.last_durations[0].sum())  # the last batch was the sentence without gold durations
    assert n == 10600 + (384 * (gold_frames - gold_frames % 2) + 10600) + (384 * (pred_frames - pred_frames % 2) + 10600)
    embs = [torch.randn(64, generator=torch.Generator().manual_seed(s)) for s in (1, 2, 3)]
    z = [torch.randn(80, 200, generator=torch.Generator().manual_seed(9)) * 0.8] * 3
    mean = tts.synthesize_ensemble(PHONES_B, embs, durations=gold, z_noise=z)
    singles = tts.synthesize_batch([PHONES_B] * 3, utterance_embeddings=embs, durations=[gold] * 3, z_noise=z)
    assert torch.allclose(mean, torch.stack(singles).mean(0), atol=1e-6)


def test_batch_equals_one_by_one(tts):
    z = [torch.randn(80, 400, generator=torch.Generator().manual_seed(i)) * 0.8 for i in range(2)]
    both = tts.synthesize_batch([PHONES_A, PHONES_B], z_noise=z)
    one = [tts.synthesize_batch([p], z_noise=[zz])[0] for p, zz in zip((PHONES_A, PHONES_B), z)]
    for a, b in zip(both, one):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) < 1e-5


def test_deal_by_length_is_balanced_and_complete():
    lengths = [128, 97, 64, 20, 111, 5, 77, 128, 64]
    shards = dd.deal_by_length(lengths, 4)
    assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(lengths)


def _rank_main(rank, world, port, models_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tests import abi_emulator as emu
    emu.install()
    interface.MODELS_DIR = models_dir
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    tts = interface.ToucanTTSInterface(device="cpu", tts_model_path="Meta", faster_vocoder=True)
    texts = [PHONES_A, PHONES_B, "~ˈa~#"]
    z = [torch.randn(80, 400, generator=torch.Generator().manual_seed(i)) * 0.8 for i in range(3)]
    kw = {}
    if world == 4:  # ragged counts: 3 utterances on 4 ranks leave one shard EMPTY; gold durations on this leg (host-side cost key)
        L = [int(tts.text2phone.string_to_tensor(t, input_phonemes=True).shape[0]) for t in texts]
        kw["durations"] = [torch.full((n,), 2 + i, dtype=torch.long) for i, n in enumerate(L)]
    waves = tts.synthesize_batch(texts, z_noise=z, distributed=True, **kw)
    if rank == 0:
        single = tts.synthesize_batch(texts, z_noise=z, **kw)
        q.put([float((a - b).abs().max()) if a.shape == b.shape else 1e9 for a, b in zip(waves, single)])
    else:
        q.put([w.numel() for w in waves])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sharding_matches_single_process(models_dir, world):
    """world 2: predicted durations (stage A + frame-count all-gather + re-deal by frames); world 4: gold durations, ragged
    shard sizes with one empty shard.  Sharded == single process, bit for bit, on every rank's copy."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, models_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errs = [r for r in res if isinstance(r[0], float)][0]
    assert max(errs) == 0.0, errs  # same kernels, same per-utterance arithmetic -> identical
    sizes = [r for r in res if not isinstance(r[0], float)]
    assert all(s == sizes[0] for s in sizes) and len(sizes[0]) == 3  # every rank holds every waveform
