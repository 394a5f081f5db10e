"""Drop-in counterpart of InferenceInterfaces/ToucanTTSInterface.py (:21-309) on the HIP engines.

Same class name, constructor keywords, methods and behaviour as the reference's ``ToucanTTSInterface`` so that a
script written against the reference (run_text_to_file_reader.py:8-16) works unchanged when this repository's
``InferenceInterfaces`` package is the one on ``sys.path``:

* path shorthand ``"Meta"`` -> ``Models/ToucanTTS_Meta/best.pt``; vocoder default ``Models/{Avocodo|BigVGAN}/best.pt`` (:33-40)
* checkpoints are the reference's own formats: ``{"model": state_dict, "default_emb": tensor}`` and ``{"generator": state_dict}``
  (run_weight_averaging.py:108-116); weight norm folding / flow inverses happen in ``packing.py``
* ``forward`` / ``__call__`` (:132-229), ``read_to_file`` (:231-285: 10 600 samples of silence around sentences, blank
  strings skipped, 24 kHz output or sample-doubled 48 kHz PCM16), ``read_aloud`` (:287-309), the language / embedding setters
* additive API: ``synthesize_batch`` (ragged batches, optionally sharded over the ranks of torch.distributed)

What is NOT here (outside the hot path, unavailable offline): grapheme-to-phoneme conversion (espeak-ng),
the GST style-embedding network behind ``set_utterance_embedding(path)``, plotting.  These raise explicit errors.
"""
import itertools
import os
import wave as _wave

import numpy as np
import torch

from . import engine
from .phonemes import ArticulatoryCombinedTextFrontend, get_language_id
from .ragged import Ragged

MODELS_DIR = os.environ.get("TOUCAN_MODELS_DIR", "Models/")  # Utility/storage_config.py:1


def float2pcm(sig, dtype="int16"):
    """Utility/utils.py:20-33."""
    sig = np.asarray(sig)
    if sig.dtype.kind != "f":
        raise TypeError("'sig' must be a float array")
    dtype = np.dtype(dtype)
    if dtype.kind not in "iu":
        raise TypeError("'dtype' must be an integer type")
    i = np.iinfo(dtype)
    abs_max = 2 ** (i.bits - 1)
    offset = i.min + abs_max
    return (sig * abs_max + offset).clip(i.min, i.max).astype(dtype)


def write_wav(path, data, samplerate):
    """soundfile.write(file, data, samplerate) for a .wav target (PCM_16 is soundfile's WAV default); uses soundfile when present."""
    try:
        import soundfile
        soundfile.write(file=path, data=data, samplerate=samplerate, subtype="PCM_16")
        return
    except ImportError:
        pass
    pcm = data if np.asarray(data).dtype.kind in "iu" else float2pcm(np.asarray(data, dtype=np.float32))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with _wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(samplerate)
        f.writeframes(np.asarray(pcm, dtype="<i2").tobytes())


def _load_checkpoint(path):
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: checkpoint not found (the reference downloads it with run_model_downloader.py; "
                                f"offline, write fixture checkpoints with ims_toucan_prosody_variance_amd.interface.write_fixture_checkpoints)")
    return torch.load(path, map_location="cpu", weights_only=True)


def _to_numpy_sd(sd):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}


def write_fixture_checkpoints(models_dir=MODELS_DIR, n_lang=8000):
    """Write the seeded fixture weights in the reference's checkpoint layout (Models/ToucanTTS_Meta, Avocodo, BigVGAN)."""
    from . import fixture_weights as fw
    t = lambda sd: {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    for sub, obj in (("ToucanTTS_Meta", {"model": t(fw.acoustic_state_dict(n_lang=n_lang)),
                                         "default_emb": torch.from_numpy(fw.default_utterance_embedding())}),
                     ("Avocodo", {"generator": t(fw.hifigan_state_dict())}),
                     ("BigVGAN", {"generator": t(fw.bigvgan_state_dict())})):
        os.makedirs(os.path.join(models_dir, sub), exist_ok=True)
        torch.save(obj, os.path.join(models_dir, sub, "best.pt"))


class ToucanTTSInterface(torch.nn.Module):

    def __init__(self,
                 device="cpu",
                 tts_model_path=os.path.join(MODELS_DIR, "ToucanTTS_Meta", "best.pt"),
                 embedding_model_path=None,
                 vocoder_model_path=None,
                 faster_vocoder=True,
                 language="en"):
        super().__init__()
        self.device = device
        if not tts_model_path.endswith(".pt"):
            tts_model_path = os.path.join(MODELS_DIR, f"ToucanTTS_{tts_model_path}", "best.pt")
        if vocoder_model_path is None:
            vocoder_model_path = os.path.join(MODELS_DIR, "Avocodo" if faster_vocoder else "BigVGAN", "best.pt")

        self.text2phone = ArticulatoryCombinedTextFrontend(language=language, add_silence_to_end=True)

        checkpoint = _load_checkpoint(tts_model_path)
        sd = _to_numpy_sd(checkpoint["model"])
        # variant detection: the reference retries load_state_dict (:55-63); the schema tells us directly
        self.use_lang_id = "encoder.language_embedding.weight" in sd
        self.phone2mel = engine.AcousticEngine(sd, device)

        self.embedding_model_path = embedding_model_path  # GST network: not on the hot path (see module docstring)

        voc = _load_checkpoint(vocoder_model_path)
        self.mel2wav = engine.VocoderEngine(_to_numpy_sd(voc["generator"]), "hifigan" if faster_vocoder else "bigvgan", device)

        self.default_utterance_embedding = checkpoint["default_emb"].to(self.device)
        self.lang_id = get_language_id_tensor(language) if self.use_lang_id else None
        self.eval()

    # ---- setters (:103-130) -------------------------------------------------------------------------
    def set_utterance_embedding(self, path_to_reference_audio="", embedding=None):
        if embedding is not None:
            self.default_utterance_embedding = embedding.squeeze().to(self.device)
            return
        assert os.path.exists(path_to_reference_audio)
        raise NotImplementedError("computing a style embedding from audio needs the GST network and the librosa front-end "
                                  "(outside the MI355X hot path); pass embedding=<tensor[64]> instead")

    def set_language(self, lang_id):
        self.set_phonemizer_language(lang_id=lang_id)
        self.set_accent_language(lang_id=lang_id)

    def set_phonemizer_language(self, lang_id):
        self.text2phone = ArticulatoryCombinedTextFrontend(language=lang_id, add_silence_to_end=True)

    def set_accent_language(self, lang_id):
        self.lang_id = get_language_id_tensor(lang_id).to(self.device) if self.use_lang_id else None

    # ---- synthesis -----------------------------------------------------------------------------------
    def _lang(self):
        return None if self.lang_id is None else int(self.lang_id.reshape(-1)[0])

    def forward(self,
                text,
                view=False,
                duration_scaling_factor=1.0,
                pitch_variance_scale=1.0,
                energy_variance_scale=1.0,
                pause_duration_scaling_factor=1.0,
                durations=None,
                pitch=None,
                energy=None,
                input_is_phones=False,
                return_plot_as_filepath=False):
        if view or return_plot_as_filepath:
            raise NotImplementedError("plotting (matplotlib/librosa) is outside the hot path")
        with torch.inference_mode():
            phones = self.text2phone.string_to_tensor(text, input_phonemes=input_is_phones)
            wavs = self._synthesize([phones], [self.default_utterance_embedding], [self._lang()],
                                    durations=None if durations is None else [durations],
                                    pitch=None if pitch is None else [pitch],
                                    energy=None if energy is None else [energy],
                                    duration_scaling_factor=duration_scaling_factor, pitch_variance_scale=pitch_variance_scale,
                                    energy_variance_scale=energy_variance_scale,
                                    pause_duration_scaling_factor=pause_duration_scaling_factor)
        return wavs[0]

    def _synthesize(self, phones, embs, langs, z_noise=None, **kw):
        emb = torch.stack([e.reshape(-1).to(torch.float32).cpu() for e in embs])
        lang_ids = None if any(l is None for l in langs) else langs
        out = self.phone2mel.forward(phones, emb, lang_ids, z_noise=z_noise, **kw)
        wav, rag = self.mel2wav.forward(out["mel_packed"], out["rag_mel"])
        self.last_durations, self.last_pitch, self.last_energy = out["durations"], out["pitch"], out["energy"]
        return [wav[b:b + n] for b, n in zip(rag.begins, rag.lengths)]

    def synthesize_batch(self, texts, input_is_phones=True, utterance_embeddings=None, z_noise=None, durations=None, pitch=None,
                         energy=None, duration_scaling_factor=1.0, pitch_variance_scale=1.0, energy_variance_scale=1.0,
                         pause_duration_scaling_factor=1.0, distributed=False):
        """Additive API: a ragged batch in one pass; each utterance equals the reference run on it alone.
        texts: phoneme strings (or [L,62] feature tensors).  With ``distributed=True`` and an initialised process group the
        utterances are dealt over the ranks by length and every rank returns all waveforms (one all-gather)."""
        feats = [t if torch.is_tensor(t) else self.text2phone.string_to_tensor(t, input_phonemes=input_is_phones) for t in texts]
        embs = utterance_embeddings if utterance_embeddings is not None else [self.default_utterance_embedding] * len(feats)
        kw = dict(duration_scaling_factor=duration_scaling_factor, pitch_variance_scale=pitch_variance_scale,
                  energy_variance_scale=energy_variance_scale, pause_duration_scaling_factor=pause_duration_scaling_factor)
        if not distributed:
            with torch.inference_mode():
                return self._synthesize(feats, embs, [self._lang()] * len(feats), z_noise=z_noise, durations=durations, pitch=pitch,
                                        energy=energy, **kw)
        from . import distributed as dd
        return dd.synthesize_sharded(self, feats, embs, z_noise, durations, pitch, energy, kw)

    def read_to_file(self,
                     text_list,
                     file_location,
                     duration_scaling_factor=1.0,
                     pitch_variance_scale=1.0,
                     energy_variance_scale=1.0,
                     silent=False,
                     dur_list=None,
                     pitch_list=None,
                     energy_list=None,
                     increased_compatibility_mode=False,
                     input_is_phones=False):
        if not dur_list:
            dur_list = []
        if not pitch_list:
            pitch_list = []
        if not energy_list:
            energy_list = []
        silence = torch.zeros([10600])
        wav = silence.clone()
        for (text, durations, pitch, energy) in itertools.zip_longest(text_list, dur_list, pitch_list, energy_list):
            if text.strip() != "":
                if not silent:
                    print("Now synthesizing: {}".format(text))
                spoken_sentence = self(text,
                                       durations=durations.to(self.device) if durations is not None else None,
                                       pitch=pitch.to(self.device) if pitch is not None else None,
                                       energy=energy.to(self.device) if energy is not None else None,
                                       duration_scaling_factor=duration_scaling_factor,
                                       pitch_variance_scale=pitch_variance_scale,
                                       energy_variance_scale=energy_variance_scale,
                                       input_is_phones=input_is_phones).cpu()
                wav = torch.cat((wav, spoken_sentence, silence), 0)
        if increased_compatibility_mode:
            doubled = np.repeat(wav.numpy(), 2)  # 24 kHz -> 48 kHz by sample doubling (:282)
            write_wav(file_location, float2pcm(doubled), 48000)
        else:
            write_wav(file_location, wav.numpy(), 24000)

    def read_aloud(self, text, view=False, duration_scaling_factor=1.0, pitch_variance_scale=1.0, energy_variance_scale=1.0,
                   blocking=False, increased_compatibility_mode=False):
        if text.strip() == "":
            return
        try:
            import sounddevice
        except ImportError as e:
            raise RuntimeError("read_aloud needs the sounddevice package") from e
        wav = self(text, view, duration_scaling_factor=duration_scaling_factor, pitch_variance_scale=pitch_variance_scale,
                   energy_variance_scale=energy_variance_scale).cpu()
        wav = torch.cat((wav, torch.zeros([12000])), 0).numpy()
        if increased_compatibility_mode:
            sounddevice.play(float2pcm(np.repeat(wav, 2)), samplerate=48000)
        else:
            sounddevice.play(wav, samplerate=24000)
        if blocking:
            sounddevice.wait()


def get_language_id_tensor(language):
    """Preprocessing/TextFrontend.py:490-524 returns a LongTensor([id])."""
    i = get_language_id(language)
    if i is None:
        raise ValueError(f"language {language!r} has no id (TextFrontend.py:490-524)")
    return torch.LongTensor([i])
