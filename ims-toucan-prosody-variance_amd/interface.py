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

* ``set_utterance_embedding(path)`` (:103-114): reference audio -> log-mel -> GST style embedding on the GPU (style.py)

What is NOT here (unavailable offline): grapheme-to-phoneme conversion (espeak-ng / phonemizer: raw text
raises - pass phoneme strings with ``input_is_phones=True``) and the silero voice-activity trim of the reference audio.
"""
import os
import wave as _wave

import numpy as np
import torch

from . import engine
from .phonemes import ArticulatoryCombinedTextFrontend, get_language_id
from .ragged import Ragged

MODELS_DIR = os.environ.get("TOUCAN_MODELS_DIR", "Models/")  # Utility/storage_config.py:1


def float2pcm(sig, dtype="int16"):
    """Float waveform in [-1, 1) -> integer PCM the way the reference's writer does it (Utility/utils.py:20-33): scale by half the
    integer range around the type's mid point, saturate, and let the integer cast drop the fraction (no rounding step)."""
    sig, dt = np.asarray(sig), np.dtype(dtype)
    if sig.dtype.kind != "f":
        raise TypeError("'sig' must be a float array")
    if dt.kind not in "iu":
        raise TypeError("'dtype' must be an integer type")
    lo, hi = int(np.iinfo(dt).min), int(np.iinfo(dt).max)
    half_range = float(1 << (8 * dt.itemsize - 1))
    mid = lo + half_range  # 0 for signed types, the centre of the range for unsigned ones
    return np.clip(sig * half_range + mid, lo, hi).astype(dt)


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
                     ("BigVGAN", {"generator": t(fw.bigvgan_state_dict())}),
                     ("Embedding", {"style_emb_func": t(fw.style_state_dict())})):
        os.makedirs(os.path.join(models_dir, sub), exist_ok=True)
        torch.save(obj, os.path.join(models_dir, sub, "embedding_function.pt" if sub == "Embedding" else "best.pt"))


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
        # precision of the MFMA GEMMs: fp32 (exact-parity default), TOUCAN_PRECISION=bf16 / f16 (BASELINE.json configs[2] / [4]),
        # f32x3 (fp32 tensors, dense products as three fp16 MFMAs on split operands: keeps the fp32 tolerances), or mixed / mixed3:
        # acoustic model in fp32 / f32x3 (the mel keeps the fp32 parity), vocoder on the fp16 matrix cores
        precision = os.environ.get("TOUCAN_PRECISION", "f32")
        voc_precision = "f16" if precision in ("mixed", "mixed3") else precision
        precision = {"mixed": "f32", "mixed3": "f32x3"}.get(precision, precision)
        voc = _load_checkpoint(vocoder_model_path)
        voc_sd, kind = _to_numpy_sd(voc["generator"]), "hifigan" if faster_vocoder else "bigvgan"
        # On a GPU the whole pass runs through the stage API (native.NativePipeline -> csrc/pipeline.hip: the kernels are sequenced in
        # C++, a dozen C calls per batch).  TOUCAN_PY_SEQUENCER=1 - and the CPU test emulator - keep engine.py's Python sequencing.
        import ctypes
        from . import capi
        self.pipe = None
        if torch.device(device).type == "cuda" and isinstance(capi.lib(), ctypes.CDLL) and not os.environ.get("TOUCAN_PY_SEQUENCER"):
            from . import native
            self.pipe = native.NativePipeline(sd, voc_sd, kind, device, precision=precision, vocoder_precision=voc_precision)
            self.phone2mel = self.mel2wav = self.pipe  # (the reference's attribute names; both stages live in the one handle)
        else:
            self.phone2mel = engine.AcousticEngine(sd, device, precision=precision)
            self.mel2wav = engine.VocoderEngine(voc_sd, kind, device, precision=voc_precision)

        self.embedding_model_path = embedding_model_path  # GST network: loaded on the first set_utterance_embedding(path)
        self._style = None


        self.default_utterance_embedding = checkpoint["default_emb"].to(self.device)
        self.lang_id = get_language_id_tensor(language) if self.use_lang_id else None
        self.eval()

    # ---- setters (:103-130) -------------------------------------------------------------------------
    def set_utterance_embedding(self, path_to_reference_audio="", embedding=None):
        if embedding is not None:
            self.default_utterance_embedding = embedding.squeeze().to(self.device)
            return
        assert os.path.exists(path_to_reference_audio)
        # reference audio -> mono, peak-normalised, 16 kHz -> log-mel -> GST style embedding, the last two on the GPU (style.py).
        # Deviation (network-only dependency): the reference also trims leading / trailing silence with the silero VAD.
        from . import style
        if self._style is None:
            path = self.embedding_model_path or os.path.join(MODELS_DIR, "Embedding", "embedding_function.pt")  # :71-77
            self._style = (style.LogMel(self.device), style.StyleEngine(_to_numpy_sd(_load_checkpoint(path)["style_emb_func"]), self.device))
        data, sr = style.read_audio(path_to_reference_audio)
        logmel, gst = self._style
        spec = logmel.forward(style.normalize_reference_audio(data, sr))
        self.default_utterance_embedding = gst.forward([spec])[0].to(self.device)

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
                return_plot_as_filepath=False,
                z_noise=None):
        """The reference's signature (ToucanTTSInterface.py:132-143) plus one additive keyword: ``z_noise`` [80, T] - the PostFlow
        noise 0.8 N(0,1) the reference draws inside the model (Glow.py:363) - so that a call can be reproduced and checked
        against the oracle; None draws it on the device."""
        with torch.inference_mode():
            phones = self.text2phone.string_to_tensor(text, input_phonemes=input_is_phones)
            wavs = self._synthesize([phones], [self.default_utterance_embedding], [self._lang()],
                                    z_noise=None if z_noise is None else [z_noise],
                                    durations=None if durations is None else [durations],
                                    pitch=None if pitch is None else [pitch],
                                    energy=None if energy is None else [energy],
                                    duration_scaling_factor=duration_scaling_factor, pitch_variance_scale=pitch_variance_scale,
                                    energy_variance_scale=energy_variance_scale,
                                    pause_duration_scaling_factor=pause_duration_scaling_factor)
        if view or return_plot_as_filepath:  # ToucanTTSInterface.py:171-226 (matplotlib only: plotting.py)
            from . import plotting
            if input_is_phones:
                labels = text.replace(" ", "|")
            else:
                labels = self.text2phone.get_phone_string(text, for_plot_labels=True)
            fig = plotting.draw(wavs[0].cpu().numpy(), self.last_mel[0].cpu().numpy(), self.last_durations[0].cpu().numpy(),
                                self.last_pitch[0].cpu().numpy(), labels, text)
            if return_plot_as_filepath:
                return wavs[0], plotting.show_or_save(fig, "tmp.png")
            plotting.show_or_save(fig)
        return wavs[0]

    def _synthesize_packed(self, phones, embs, langs, z_noise=None, **kw):
        """One ragged batch through both engines.  Returns (packed waveform, [(first sample, sample count)] per utterance)."""
        emb = torch.stack([e.reshape(-1).to(torch.float32).cpu() for e in embs])
        lang_ids = None if any(l is None for l in langs) else langs
        if self.pipe is not None:
            out = self.pipe.forward(phones, emb, lang_ids, z_noise=z_noise, **kw)
            self.last_durations, self.last_pitch, self.last_energy = out["durations"], out["pitch"], out["energy"]
            self.last_mel = out["mel"]
            return out["wav"], out["wav_spans"]
        out = self.phone2mel.forward(phones, emb, lang_ids, z_noise=z_noise, **kw)
        wav, rag = self.mel2wav.forward(out["mel_packed"], out["rag_mel"])
        self.last_durations, self.last_pitch, self.last_energy = out["durations"], out["pitch"], out["energy"]
        self.last_mel = out["mel"]
        return wav, list(zip(rag.begins, rag.lengths))

    def _synthesize(self, phones, embs, langs, z_noise=None, **kw):
        wav, spans = self._synthesize_packed(phones, embs, langs, z_noise=z_noise, **kw)
        return [wav[b:b + n] for b, n in spans]

    def predict_frame_counts(self, feats, embs, pitch=None, energy=None, **kw):
        """Mel frames each utterance will get (stage A only) - the balancing key of the multi-GPU deal."""
        emb = torch.stack([e.reshape(-1).to(torch.float32).cpu() for e in embs])
        langs = None if self.lang_id is None else [self._lang()] * len(feats)
        return self.phone2mel.predict_frame_counts(feats, emb, langs, pitch=pitch, energy=energy, **kw)

    def synthesize_batch(self, texts, input_is_phones=True, utterance_embeddings=None, z_noise=None, durations=None, pitch=None,
                         energy=None, duration_scaling_factor=1.0, pitch_variance_scale=1.0, energy_variance_scale=1.0,
                         pause_duration_scaling_factor=1.0, distributed=False):
        """Additive API: a ragged batch in one pass; each utterance equals the reference run on it alone (16-bit configurations: bit
        for bit whatever the batch; fp32: durations / pitch / energy bit for bit, the mel to fp32 rounding order - DESIGN.md section 4).
        texts: phoneme strings (or [L,62] feature tensors).  durations / pitch / energy: optional per-utterance gold prosody (the
        cloner-style call, UtteranceCloner.py:163).  With ``distributed=True`` and an initialised process group the utterances are
        dealt over the ranks by frame count and every rank returns all waveforms (distributed.py)."""
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

    def synthesize_ensemble(self, text, utterance_embeddings, durations=None, pitch=None, energy=None, input_is_phones=True, z_noise=None):
        """Several voices speaking one text with the same prosody, averaged (UtteranceCloner.py:166-194's ensemble) - as ONE batch
        over the voices instead of a loop that swaps the default embedding.  Without gold durations the voices may predict
        different lengths; the mean is then taken over the common prefix."""
        n = len(utterance_embeddings)
        rep = lambda v: None if v is None else [v] * n
        waves = self.synthesize_batch([text] * n, input_is_phones=input_is_phones, utterance_embeddings=list(utterance_embeddings),
                                      durations=rep(durations), pitch=rep(pitch), energy=rep(energy), z_noise=z_noise)
        m = min(w.numel() for w in waves)
        return torch.stack([w[:m] for w in waves]).mean(dim=0)

    def stream(self, text, input_is_phones=False, chunk_frames=512, halo_frames=None, max_batch=4, z_noise=None, **prosody):
        """Generator of waveform pieces for ONE (long) text: the acoustic model runs once, the vocoder chunk-wise with overlap
        (streaming.py) - the concatenation equals ``self(text, ...)`` bit for bit, the first piece is ready after one chunk and the
        vocoder's workspace is bounded by ``max_batch`` chunks.  prosody: the keyword arguments of ``forward`` (gold durations /
        pitch / energy, scaling factors)."""
        from . import streaming
        halo = streaming.DEFAULT_HALO if halo_frames is None else halo_frames
        listed = {k: [v] for k, v in prosody.items() if k in ("durations", "pitch", "energy") and v is not None}
        scales = {k: v for k, v in prosody.items() if k.endswith("_factor") or k.endswith("_scale")}
        with torch.inference_mode():
            phones = self.text2phone.string_to_tensor(text, input_phonemes=input_is_phones)
            emb = self.default_utterance_embedding.reshape(1, -1).to(torch.float32).cpu()
            langs = None if self.lang_id is None else [self._lang()]
            if self.pipe is not None:
                out = self.pipe.forward([phones], emb, langs, z_noise=None if z_noise is None else [z_noise], vocode=False, **listed, **scales)
                vocode = self.pipe.vocode
            else:
                out = self.phone2mel.forward([phones], emb, langs, z_noise=None if z_noise is None else [z_noise], **listed, **scales)
                vocode = self.mel2wav.forward
            self.last_durations, self.last_pitch, self.last_energy = out["durations"], out["pitch"], out["energy"]
            yield from streaming.stream_vocode(vocode, out["mel"][0].contiguous(), chunk_frames, halo, max_batch)

    SILENCE_SAMPLES = 10600   # between sentences in read_to_file (ToucanTTSInterface.py:267)
    MAX_FILE_BATCH = 32       # sentences synthesised per ragged batch by read_to_file

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
        """Same result as the reference's sentence loop (:231-285: silence, sentence, silence, ... with blank strings skipped,
        24 kHz float file or sample-doubled 48 kHz PCM16), but the sentences go through the engines as ragged batches of up to
        MAX_FILE_BATCH utterances and the file is assembled once on the host."""
        n = len(text_list)
        column = lambda lst: list(lst) + [None] * (n - len(lst)) if lst else [None] * n
        durs, pits, enes = column(dur_list), column(pitch_list), column(energy_list)
        spoken = [i for i, t in enumerate(text_list) if t.strip() != ""]
        pieces = {}
        for lo in range(0, len(spoken), self.MAX_FILE_BATCH):
            chunk = spoken[lo:lo + self.MAX_FILE_BATCH]
            if not silent:
                for i in chunk:
                    print("Now synthesizing: {}".format(text_list[i]))
            # gold prosody is per sentence and optional per sentence: sentences that share the same set of gold quantities
            # form one batch (the engines take a quantity either for every utterance of a batch or for none)
            groups = {}
            for i in chunk:
                groups.setdefault((durs[i] is not None, pits[i] is not None, enes[i] is not None), []).append(i)
            for (has_d, has_p, has_e), idx in groups.items():
                waves = self.synthesize_batch([text_list[i] for i in idx], input_is_phones=input_is_phones,
                                              durations=[durs[i] for i in idx] if has_d else None,
                                              pitch=[pits[i] for i in idx] if has_p else None,
                                              energy=[enes[i] for i in idx] if has_e else None,
                                              duration_scaling_factor=duration_scaling_factor, pitch_variance_scale=pitch_variance_scale,
                                              energy_variance_scale=energy_variance_scale)
                for i, w in zip(idx, waves):
                    pieces[i] = w.cpu().numpy()
        gap = self.SILENCE_SAMPLES
        total = gap + sum(pieces[i].shape[0] + gap for i in spoken)
        audio = np.zeros(total, dtype=np.float32)
        at = gap
        for i in spoken:
            audio[at:at + pieces[i].shape[0]] = pieces[i]
            at += pieces[i].shape[0] + gap
        if increased_compatibility_mode:  # 24 kHz is less widely supported than 48 kHz: every sample twice, 16-bit integers
            write_wav(file_location, float2pcm(np.repeat(audio, 2)), 48000)
        else:
            write_wav(file_location, audio, 24000)

    def read_aloud(self, text, view=False, duration_scaling_factor=1.0, pitch_variance_scale=1.0, energy_variance_scale=1.0,
                   blocking=False, increased_compatibility_mode=False):
        """Synthesise and play one text through the sound card (:287-309); half a second of silence is appended."""
        if not text.strip():
            return
        try:
            import sounddevice
        except ImportError as e:
            raise RuntimeError("read_aloud needs the sounddevice package") from e
        speech = self(text, view, duration_scaling_factor=duration_scaling_factor, pitch_variance_scale=pitch_variance_scale,
                      energy_variance_scale=energy_variance_scale)
        audio = np.concatenate([speech.cpu().numpy(), np.zeros(12000, dtype=np.float32)])
        if increased_compatibility_mode:
            sounddevice.play(float2pcm(np.repeat(audio, 2)), samplerate=48000)
        else:
            sounddevice.play(audio, samplerate=24000)
        if blocking:
            sounddevice.wait()


def get_language_id_tensor(language):
    """Preprocessing/TextFrontend.py:490-524 returns a LongTensor([id])."""
    i = get_language_id(language)
    if i is None:
        raise ValueError(f"language {language!r} has no id (TextFrontend.py:490-524)")
    return torch.LongTensor([i])
