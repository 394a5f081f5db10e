"""Chunked (overlap-save) vocoding for long texts and streaming output.

Both generators are finite-impulse-response stacks (convs, transposed convs, 12-tap anti-alias filters): a sample further than the
stack's reach from a chunk edge does not see the edge.  A long mel is therefore cut into chunks of ``chunk_frames`` frames, each
extended by ``halo_frames`` of context on both sides; the extended chunks go through the vocoder as the utterances of ONE ragged
batch (or of several bounded ones) and only each chunk's own samples are kept.  With halo >= the reach the result is
BIT-IDENTICAL to vocoding the whole mel at once (same kernels, same per-row arithmetic) - tests/test_gpu_e2e.py checks it on
5 120 frames - while the workspace is bounded by the batch of chunks instead of the utterance, and the first audio is ready after
one chunk instead of after the whole text.

Reach of the BigVGAN / Avocodo generator in mel frames (InferenceBigVGAN.py:72-95, AMP.py:51-60): pre conv 3 + transposed convs
1 + 1/8 + 1/48 + 1/192 + per stage the k = 11 block's (5 + 15 + 25 dilated + 3 x 5 plain taps + 6 x 6 filter samples) = 96
samples at 8 / 48 / 192 / 384 samples per frame + the output conv -> 18.9 frames; the default halo is 24.
"""
import torch

from .ragged import Ragged

REACH_FRAMES = 19
DEFAULT_HALO = 24


def plan_chunks(n_frames, chunk_frames, halo_frames):
    """[(ext_begin, ext_end, keep_begin, keep_end)] in frames: chunk i keeps [keep_begin, keep_end) and is computed on [ext_begin, ext_end)."""
    assert chunk_frames > 0 and halo_frames >= 0
    out = []
    for lo in range(0, n_frames, chunk_frames):
        hi = min(n_frames, lo + chunk_frames)
        out.append((max(0, lo - halo_frames), min(n_frames, hi + halo_frames), lo, hi))
    return out


def stream_vocode(vocode, mel, chunk_frames=512, halo_frames=DEFAULT_HALO, max_batch=8):
    """Generator of waveform pieces (1-D tensors on the mel's device, in order) for ONE utterance's mel [T, 80].

    vocode(mel_packed [rows, 80], Ragged) -> (packed waveform, Ragged of samples): ``VocoderEngine.forward`` or
    ``NativePipeline.vocode``.  Chunks are vocoded ``max_batch`` at a time as one ragged batch."""
    assert mel.dim() == 2 and mel.shape[1] == 80
    dev = mel.device
    plan = plan_chunks(int(mel.shape[0]), chunk_frames, halo_frames)
    for g0 in range(0, len(plan), max_batch):
        group = plan[g0:g0 + max_batch]
        rag = Ragged([e1 - e0 for e0, e1, _, _ in group], dev, align=2)
        packed = torch.zeros(rag.total_rows, 80, dtype=torch.float32, device=dev)
        for (e0, e1, _, _), b in zip(group, rag.begins):
            packed[b:b + (e1 - e0)] = mel[e0:e1]
        wav, rw = vocode(packed, rag)
        for (e0, _, k0, k1), b in zip(group, rw.begins):
            yield wav[b + 384 * (k0 - e0): b + 384 * (k1 - e0)]


def chunked_vocode(vocode, mel, chunk_frames=512, halo_frames=DEFAULT_HALO, max_batch=8):
    """The whole waveform of one long mel, computed chunk-wise (bounded workspace)."""
    return torch.cat(list(stream_vocode(vocode, mel, chunk_frames, halo_frames, max_batch)))
