"""The per-speaker path behind ``set_utterance_embedding(path)`` (ToucanTTSInterface.py:103-114) on the HIP kernels:

* ``LogMel``: AudioPreprocessor.logmelfilterbank (Preprocessing/AudioPreprocessor.py:96-117) - the centred, Hann-windowed STFT
  (n_fft 1024, hop 256) as ONE 4-tap conv over the audio viewed as rows of 256 samples (the window is folded into a real DFT
  basis [4][256][513 re | 513 im]), magnitude, the 80-band mel projection, log10(max(1e-10, .)).
* ``StyleEngine``: StyleEmbedding.forward + the GST encoder (TrainingInterfaces/Spectrogram_to_Embedding/StyleEmbedding.py:21-57,
  GST.py:59-243) - the eight stride-2 Conv2d + BatchNorm(eval) + ReLU layers as banded 2-tap convs over row PAIRS of a
  [time, frequency x channels] tensor (stride 2 in time = a re-view, stride 2 in frequency = the band structure of the packed
  weight; BatchNorm folded), a two-layer GRU, and the 2000-token attention with one query per utterance.

Everything dense goes through tts_conv1d in fp32 (exact fp32 MFMA); tts_gru_layer, tts_style_tokens, tts_complex_magnitude and
tts_log10_floor are the rest (csrc/style.hip).  Host code here is weight layout and launch order only.

What is NOT reproduced (third-party / network): the silero voice-activity trim (``cut_silence=True`` needs
torch.hub) and - PARITY UNPINNED - librosa's mel basis and torchaudio's resampler, both restated from their documented algorithms.
"""
import ctypes as C
import math
import os
import wave as _wave

import numpy as np
import torch

from . import capi, engine, packing
from .capi import ACT_NONE, ACT_RELU, MODE_LINEAR
from .ragged import Ragged

N_FFT, HOP, N_MELS, SR = 1024, 256, 80, 16000


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=40.0, fmax=8000.0):
    """librosa.filters.mel defaults (Slaney scale, area-normalised triangles) - third party, PARITY UNPINNED, restated."""
    hz2mel = lambda f: np.where(np.asarray(f, np.float64) >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0),
                                np.asarray(f, np.float64) / (200.0 / 3))
    mel2hz = lambda m: np.where(np.asarray(m, np.float64) >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (np.asarray(m, np.float64) - 15.0)),
                                np.asarray(m, np.float64) * (200.0 / 3))
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.maximum(0.0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return (w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]).astype(np.float32)


class LogMel:
    def __init__(self, device, sr=SR):
        self.ops = engine.Ops(device)
        self.device = self.ops.device
        self.sr = sr
        n = np.arange(N_FFT, dtype=np.float64)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)  # periodic Hann (scipy get_window("hann", fftbins=True))
        k = np.arange(N_FFT // 2 + 1, dtype=np.float64)
        ang = 2.0 * np.pi * np.outer(n, k) / N_FFT
        basis = np.concatenate([win[:, None] * np.cos(ang), -win[:, None] * np.sin(ang)], axis=1)  # [1024, 1026]
        self.dft = packing.ConvWeights(basis.reshape(4, HOP, -1).astype(np.float32), None, MODE_LINEAR, 1, 0, self.device)
        self.mel = packing.pack_conv(mel_filterbank(sr), None, self.device)
        self.bins = N_FFT // 2 + 1

    @torch.inference_mode()
    def forward(self, audio):
        """audio: 1-D float array at self.sr -> log-mel [frames, 80] on the device (frames = 1 + len // 256)."""
        ops, dev = self.ops, self.device
        x = np.asarray(audio, dtype=np.float32).reshape(-1)
        assert x.size > N_FFT // 2, "reference audio is too short for the centred STFT"
        frames = 1 + x.size // HOP
        x = np.pad(x, N_FFT // 2, mode="reflect")  # librosa.stft(center=True, pad_mode="reflect")
        rows = -(-x.size // HOP)
        buf = np.zeros(rows * HOP, dtype=np.float32)
        buf[: x.size] = x
        xd = torch.from_numpy(buf).to(dev).view(rows, HOP)
        rag = Ragged([rows], dev)
        spec = ops.conv(self.dft, xd, ops.empty(rows, 2 * self.bins), rag, compute=capi.COMPUTE_F32)
        mag = ops.empty(rows, self.bins)
        capi.check(ops.lib.tts_complex_magnitude(spec.data_ptr(), 2 * self.bins, mag.data_ptr(), self.bins, rows, self.bins, ops.stream()),
                   "tts_complex_magnitude")
        melp = ops.conv(self.mel, mag, ops.empty(rows, N_MELS), rag, compute=capi.COMPUTE_F32)
        out = ops.empty(rows, N_MELS)
        capi.check(ops.lib.tts_log10_floor(melp.data_ptr(), N_MELS, out.data_ptr(), N_MELS, rows, N_MELS, 1e-10, ops.stream()), "tts_log10_floor")
        return out[:frames]


class StyleEngine:
    CHANS = (32, 32, 64, 64, 128, 128, 256, 256)
    WINDOW = 812  # StyleEmbedding.py:40

    def __init__(self, state_dict, device):
        self.ops = engine.Ops(device)
        self.device = dev = self.ops.device
        sd = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in state_dict.items()}
        self.layers = []
        f_in, c_in = N_MELS, 1
        for i, c_out in enumerate(self.CHANS):
            w = sd[f"gst.ref_enc.convs.{3 * i}.weight"].astype(np.float64)  # [c_out, c_in, kh, kw] (kh: time, kw: frequency)
            q = f"gst.ref_enc.convs.{3 * i + 1}."
            scale = sd[q + "weight"].astype(np.float64) / np.sqrt(sd[q + "running_var"].astype(np.float64) + 1e-5)  # BatchNorm2d eval
            shift = sd[q + "bias"].astype(np.float64) - sd[q + "running_mean"].astype(np.float64) * scale
            f_out = (f_in + 1) // 2
            k_in = f_in * c_in
            dense = np.zeros((2, 2 * k_in, f_out * c_out), dtype=np.float64)
            # output frame t reads input frames 2t-1, 2t, 2t+1 (kh = 0, 1, 2); with frames paired into rows, frame 2t-1 is the second
            # half of row t-1 (tap 0) and frames 2t, 2t+1 are row t (tap 1); column layout of every tensor: [frequency][channel]
            for kh, (tap, half) in enumerate(((0, 1), (1, 0), (1, 1))):
                for fo in range(f_out):
                    for kw in range(3):
                        fi = 2 * fo - 1 + kw
                        if 0 <= fi < f_in:
                            r0 = half * k_in + fi * c_in
                            dense[tap, r0:r0 + c_in, fo * c_out:(fo + 1) * c_out] = (w[:, :, kh, kw] * scale[:, None]).T
            self.layers.append((packing.ConvWeights(dense.astype(np.float32), np.tile(shift, f_out).astype(np.float32), MODE_LINEAR, 1, 1, dev),
                                k_in, f_out * c_out))
            f_in, c_in = f_out, c_out
        assert f_in == 1
        g = "gst.ref_enc.gst."
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        self.gru = [(t(sd[g + f"weight_ih_l{l}"].T), t(sd[g + f"weight_hh_l{l}"].T), t(sd[g + f"bias_ih_l{l}"]), t(sd[g + f"bias_hh_l{l}"]))
                    for l in range(2)]
        m = "gst.stl.mha."
        toks = np.tanh(sd["gst.stl.gst_embs"].astype(np.float64))  # GST.py:213: the tokens pass through tanh; constants of the model
        self.k = t(toks @ sd[m + "linear_k.weight"].astype(np.float64).T + sd[m + "linear_k.bias"])
        self.v = t(toks @ sd[m + "linear_v.weight"].astype(np.float64).T + sd[m + "linear_v.bias"])
        self.lin_q = packing.pack_conv(sd[m + "linear_q.weight"], sd[m + "linear_q.bias"], dev)
        self.lin_out = packing.pack_conv(sd[m + "linear_out.weight"], sd[m + "linear_out.bias"], dev)
        self.n_tokens = toks.shape[0]

    @torch.inference_mode()
    def forward(self, specs, return_ref=False):
        """specs: list of [T_u, 80] log-mel tensors -> style embeddings [B, 64] (and the reference embeddings [B, 256])."""
        ops, dev = self.ops, self.device
        B, T = len(specs), self.WINDOW
        x = torch.zeros(B, T, N_MELS, dtype=torch.float32, device=dev)
        for b, s in enumerate(specs):  # StyleEmbedding.py:40-52: repeat to at least 812 frames, keep the first 812
            s = torch.as_tensor(s, dtype=torch.float32).to(dev)
            reps = max(2, -(-T // s.shape[0]))
            reps = 1 << (reps - 1).bit_length()  # the reference doubles: 2, 4, 8, ... copies (the first 812 frames are the same)
            x[b] = s.repeat((reps, 1))[:T]
        for cw, k_in, n_out in self.layers:
            t_in = x.shape[1]
            if t_in % 2:  # an odd frame count: the missing partner of the last frame is the conv's zero padding
                x = torch.cat([x, torch.zeros(B, 1, k_in, dtype=torch.float32, device=dev)], dim=1)
            pairs = x.shape[1] // 2
            rag = Ragged([pairs] * B, dev, begins=[b * pairs for b in range(B)])
            y = ops.empty(B * pairs, n_out)
            ops.conv(cw, x.reshape(B * pairs, 2 * k_in), y, rag, act=ACT_RELU, compute=capi.COMPUTE_F32)
            x = y.view(B, pairs, n_out)
        steps, hid = x.shape[1], x.shape[2]
        h = x.reshape(B * steps, hid).contiguous()
        for w_ih, w_hh, b_ih, b_hh in self.gru:
            y = ops.empty(B * steps, hid)
            capi.check(ops.lib.tts_gru_layer(h.data_ptr(), hid, B, steps, hid, hid, w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(),
                                             b_hh.data_ptr(), y.data_ptr(), hid, ops.stream()), "tts_gru_layer")
            h = y
        ref = h.view(B, steps, hid)[:, -1].contiguous()
        rag_b = Ragged([B], dev)
        q = ops.conv(self.lin_q, ref, ops.empty(B, 64), rag_b, compute=capi.COMPUTE_F32)
        ctx = ops.empty(B, 64)
        capi.check(ops.lib.tts_style_tokens(q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(), B, self.n_tokens, 8, 8, ctx.data_ptr(), ops.stream()),
                   "tts_style_tokens")
        emb = ops.conv(self.lin_out, ctx, ops.empty(B, 64), rag_b, compute=capi.COMPUTE_F32)
        return (emb, ref) if return_ref else emb


# ---- reference audio on the host ------------------------------------------------------------------------------------
def read_audio(path):
    """(float32 samples [n] or [n, channels], sample rate).  soundfile when installed, else PCM WAV through the standard library."""
    try:
        import soundfile
        data, sr = soundfile.read(path)
        return np.asarray(data, dtype=np.float32), int(sr)
    except ImportError:
        pass
    with _wave.open(path, "rb") as f:
        sr, ch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: {8 * width}-bit PCM is not supported without the soundfile package")
    return (data.reshape(-1, ch) if ch > 1 else data), sr


def resample_sinc(x, sr_in, sr_out, width=6, rolloff=0.99):
    """torchaudio.transforms.Resample defaults (windowed-sinc interpolation, Hann window, lowpass_filter_width 6, rolloff 0.99)
    restated in numpy float64 - third party, PARITY UNPINNED."""
    if sr_in == sr_out:
        return np.asarray(x, dtype=np.float32)
    g = math.gcd(int(sr_in), int(sr_out))
    orig, new = int(sr_in) // g, int(sr_out) // g
    base = min(orig, new) * rolloff
    w = int(math.ceil(width * orig / base))
    idx = np.arange(-w, w + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -width, width)
    window = np.cos(t * np.pi / width / 2.0) ** 2
    t = t * np.pi
    kern = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * (base / orig)  # [new, 2w + orig]
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    xp = np.pad(x, (w, w + orig))
    n_blocks = (xp.size - kern.shape[1]) // orig + 1
    cols = np.arange(kern.shape[1])[None, :] + orig * np.arange(n_blocks)[:, None]
    out = (xp[cols] @ kern.T).reshape(-1)
    return out[: int(math.ceil(new * n / orig))].astype(np.float32)


def normalize_reference_audio(data, sr, target_sr=SR):
    """AudioPreprocessor.normalize_audio (:119-130) as far as it can be stated offline: mono (mean of the channels), loudness
    normalisation followed by peak normalisation (:80-94 - the loudness gain is a positive scalar, so after the division by the
    peak it cancels: audio / max|audio|; clips shorter than the meter's 0.4 s block stay as they are), resampling to 16 kHz.
    The voice-activity trim (:65-78, silero via torch.hub) is NOT applied."""
    x = np.asarray(data, dtype=np.float64)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if x.size >= int(0.4 * sr) and np.abs(x).max() > 0:
        x = x / np.abs(x).max()
    return resample_sinc(x, sr, target_sr)
