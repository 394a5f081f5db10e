"""The figure ``ToucanTTSInterface.forward(..., view=True / return_plot_as_filepath=True)`` draws (ToucanTTSInterface.py:171-226 of the
reference): waveform on top; below it the mel spectrogram on a mel-scaled frequency axis with one tick per phoneme at the centre of
its frames, dotted lines at the phoneme boundaries, solid lines at word boundaries with the words underneath, and the predicted
pitch of every voiced phoneme as a horizontal line at pitch x 1000 Hz.  matplotlib only (the reference goes through
``librosa.display.specshow``, which is not installed here: the mel axis is built from the same Slaney mel scale)."""
import numpy as np


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3.0))


def frame_boundaries(durations):
    """Utility/utils.py:291-299: cumulative frame counts [0, d0, d0+d1, ...] and the (integer) centre of every phoneme."""
    edges = np.concatenate([[0], np.cumsum(np.asarray(durations, dtype=np.int64))])
    centres = (edges[:-1] + edges[1:]) // 2
    return edges.tolist(), centres.tolist()


def draw(wave, mel, durations, pitch, phone_labels, text, sr_mel=16000, fmax=8000.0):
    """wave [S], mel [T, 80] (log-mel), durations [L] frames, pitch [L], phone_labels: one label per phoneme ('|' = word boundary).
    Returns the matplotlib figure."""
    import matplotlib
    import matplotlib.pyplot as plt

    wave, mel = np.asarray(wave, dtype=np.float32), np.asarray(mel, dtype=np.float32)
    T, n_mels = mel.shape
    fig, ax = plt.subplots(nrows=2, ncols=1, figsize=(9, 6))
    ax[0].plot(wave)
    ax[0].set_xlim(0, max(1, wave.shape[0]))
    # mel bins as cells between mel-spaced band edges, drawn on a mel-scaled axis (specshow(y_axis="mel", sr=16000))
    band_edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr_mel / 2.0), n_mels + 1))
    ax[1].pcolormesh(np.arange(T + 1), band_edges, mel.T, cmap="GnBu", shading="flat")
    ax[1].set_yscale("function", functions=(_hz_to_mel, _mel_to_hz))
    ax[1].set_ylim(0.0, sr_mel / 2.0)
    ax[0].yaxis.set_visible(False)
    ax[1].yaxis.set_visible(False)
    edges, centres = frame_boundaries(durations)
    n = min(len(centres), len(phone_labels))
    ax[1].xaxis.grid(True, which="minor")
    ax[1].set_xticks(centres[:n], minor=False)
    ax[1].set_xticklabels(list(phone_labels)[:n])
    word_bounds = [centres[i] for i in range(n) if phone_labels[i] == "|"]
    words = text.split()
    starts = [0] + word_bounds
    ends = word_bounds + [edges[-1]]
    mids = [(a + b) / 2 for a, b in zip(starts, ends)]
    if len(mids) == len(words) and words:
        below = ax[1].secondary_xaxis("bottom")
        below.tick_params(axis="x", direction="out", pad=24, colors="orange")
        below.set_xticks(mids, minor=False)
        below.set_xticklabels(words)
    else:  # (the reference falls back to a title when words and boundaries do not line up)
        ax[0].set_title(text)
    ax[1].vlines(x=edges, colors="green", linestyles="dotted", ymin=0.0, ymax=fmax, linewidth=1.0)
    ax[1].vlines(x=word_bounds, colors="orange", linestyles="solid", ymin=0.0, ymax=fmax, linewidth=1.2)
    p = np.asarray(pitch, dtype=np.float32).reshape(-1)
    for i in range(min(len(p), len(edges) - 1)):
        if p[i] != 0:
            ax[1].hlines(float(p[i]) * 1000.0, xmin=edges[i], xmax=edges[i + 1], color="magenta", linestyles="solid", linewidth=1.0)
    plt.subplots_adjust(left=0.05, bottom=0.12, right=0.95, top=0.9, wspace=0.0, hspace=0.0)
    return fig


def show_or_save(fig, path=None):
    import matplotlib.pyplot as plt

    if path is None:
        plt.show()
        return None
    fig.savefig(path)
    plt.close(fig)
    return path
