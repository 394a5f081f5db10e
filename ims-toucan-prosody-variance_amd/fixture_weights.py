"""Deterministic fixture weights in the reference's checkpoint schema.

No trained checkpoints exist offline (run_model_downloader.py:21-64 fetches them), so
parity and throughput runs use seeded synthetic weights.  Tensor names and shapes follow
the state_dicts of the reference's model classes:

* acoustic model  - TrainingInterfaces/Text_to_Spectrogram/ToucanTTS/ToucanTTS.py:43-208
  (== InferenceToucanTTS.py:86-178, loaded with strict load_state_dict at :180)
* Avocodo/HiFiGAN - InferenceInterfaces/InferenceArchitectures/InferenceAvocodo.py:29-66
* BigVGAN         - InferenceInterfaces/InferenceArchitectures/InferenceBigVGAN.py:37-70,
  TrainingInterfaces/Spectrogram_to_Wave/BigVGAN/AMP.py:22-49, Snake.py:44-46

The values are "tamed" random-init: the reference's own init
makes the duration predictor and the 18-block Glow numerically wild, so scales are chosen
such that activations stay O(1), durations land around 5 frames and every quirky code path
(CLN scale/bias MLPs, BatchNorm running stats, weight-norm g/v, snake alpha/beta) is
exercised with non-trivial numbers.

The generator is a counter-based integer hash (splitmix64) followed by exact float64
arithmetic only (no libm calls), so the same tensors come out bit-identically in the
survey container and on the GPU box.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _hash_name(name: str, seed: int) -> np.uint64:
    h = np.uint64(0xCBF29CE484222325)  # FNV-1a 64
    with np.errstate(over="ignore"):
        for ch in name.encode("utf-8"):
            h = np.uint64(h ^ np.uint64(ch))
            h = np.uint64(h * np.uint64(0x100000001B3))
        h = np.uint64(h ^ (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)))
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(name: str, n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n float64 values in [0,1) with 24-bit resolution (exactly representable in fp32)."""
    key = _hash_name(name, seed)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(16) + np.uint64(stream) + key
    bits = _splitmix64(ctr)
    return (bits >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def normal(name: str, shape, seed: int, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Approximately N(mean, std): Irwin-Hall sum of 12 uniforms (exact in float64)."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for s in range(12):
        acc += uniform01(name, n, seed, stream=s)
    out = (acc - 6.0) * std + mean
    return out.reshape(shape).astype(np.float32)


def uniform(name: str, shape, seed: int, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n, seed, stream=13)
    return (lo + (hi - lo) * u).reshape(shape).astype(np.float32)


# --------------------------------------------------------------------------------------
# acoustic model
# --------------------------------------------------------------------------------------
ATT = 192
HEADS = 4
FFN = 1536
N_MEL = 80
UTT = 64
N_LANG = 8000
GLOW_BLOCKS = 18
GLOW_LAYERS = 4
GLOW_SHARE = 4


def _xavier(name, shape, seed, gain=1.0):
    # fan computation as torch.nn.init.xavier_uniform_ (Utility/utils.py:436-460 uses it for dim>1)
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_out, fan_in = shape[0] * rf, shape[1] * rf
    a = gain * np.sqrt(6.0 / (fan_in + fan_out))
    return uniform(name, shape, seed, -a, a)


def _conformer(sd, prefix, seed, kernel, n_blocks=6):
    for b in range(n_blocks):
        p = f"{prefix}.encoders.{b}."
        sd[p + "self_attn.pos_bias_u"] = _xavier(p + "u", (HEADS, ATT // HEADS), seed)
        sd[p + "self_attn.pos_bias_v"] = _xavier(p + "v", (HEADS, ATT // HEADS), seed)
        for lin in ("linear_q", "linear_k", "linear_v", "linear_out"):
            sd[p + f"self_attn.{lin}.weight"] = _xavier(p + lin, (ATT, ATT), seed)
            sd[p + f"self_attn.{lin}.bias"] = normal(p + lin + "b", (ATT,), seed, 0.02)
        sd[p + "self_attn.linear_pos.weight"] = _xavier(p + "pos", (ATT, ATT), seed)
        for ff in ("feed_forward", "feed_forward_macaron"):
            sd[p + ff + ".w_1.weight"] = _xavier(p + ff + "1", (FFN, ATT, 1), seed)
            sd[p + ff + ".w_1.bias"] = normal(p + ff + "1b", (FFN,), seed, 0.02)
            sd[p + ff + ".w_2.weight"] = _xavier(p + ff + "2", (ATT, FFN, 1), seed)
            sd[p + ff + ".w_2.bias"] = normal(p + ff + "2b", (ATT,), seed, 0.02)
        c = p + "conv_module."
        sd[c + "pointwise_conv1.weight"] = _xavier(c + "pw1", (2 * ATT, ATT, 1), seed)
        sd[c + "pointwise_conv1.bias"] = normal(c + "pw1b", (2 * ATT,), seed, 0.02)
        sd[c + "depthwise_conv.weight"] = uniform(c + "dw", (ATT, 1, kernel), seed, -1.0, 1.0) * np.float32(
            1.0 / np.sqrt(kernel))
        sd[c + "depthwise_conv.bias"] = normal(c + "dwb", (ATT,), seed, 0.02)
        sd[c + "norm.weight"] = normal(c + "bnw", (ATT,), seed, 0.1, 1.0)
        sd[c + "norm.bias"] = normal(c + "bnb", (ATT,), seed, 0.1)
        sd[c + "norm.running_mean"] = normal(c + "bnm", (ATT,), seed, 0.1)
        sd[c + "norm.running_var"] = uniform(c + "bnv", (ATT,), seed, 0.5, 1.5)
        sd[c + "norm.num_batches_tracked"] = np.array(100, dtype=np.int64)
        sd[c + "pointwise_conv2.weight"] = _xavier(c + "pw2", (ATT, ATT, 1), seed)
        sd[c + "pointwise_conv2.bias"] = normal(c + "pw2b", (ATT,), seed, 0.02)
        for ln in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final"):
            sd[p + ln + ".weight"] = normal(p + ln + "w", (ATT,), seed, 0.1, 1.0)
            sd[p + ln + ".bias"] = normal(p + ln + "b", (ATT,), seed, 0.1)


def _predictor(sd, prefix, seed, n_layers, kernel, chans=256, multispeaker=True):
    for i in range(n_layers):
        cin = ATT if i == 0 else chans
        sd[f"{prefix}.conv.{i}.0.weight"] = _xavier(f"{prefix}.c{i}", (chans, cin, kernel), seed, gain=1.4)
        sd[f"{prefix}.conv.{i}.0.bias"] = normal(f"{prefix}.c{i}b", (chans,), seed, 0.05)
    for i in range(n_layers):
        if not multispeaker:  # single-speaker variant: plain LayerNorm(n_chans, dim=1) (VariancePredictor.py:47-48)
            sd[f"{prefix}.norms.{i}.weight"] = normal(f"{prefix}.n{i}w", (chans,), seed, 0.1, 1.0)
            sd[f"{prefix}.norms.{i}.bias"] = normal(f"{prefix}.n{i}b", (chans,), seed, 0.1)
            continue
        for which, b_last in (("W_scale", 1.0), ("W_bias", 0.0)):
            q = f"{prefix}.norms.{i}.{which}."
            # ConditionalLayerNorm.py:38-50 resets these to constants; small random weights keep the
            # conditioning path (s(e), b(e) MLPs of the utterance embedding) numerically alive.
            sd[q + "0.weight"] = normal(q + "0w", (UTT, UTT), seed, 0.1)
            sd[q + "0.bias"] = normal(q + "0b", (UTT,), seed, 0.1)
            sd[q + "2.weight"] = normal(q + "2w", (chans, UTT), seed, 0.1)
            sd[q + "2.bias"] = normal(q + "2b", (chans,), seed, 0.1)
            sd[q + "4.weight"] = normal(q + "4w", (chans, chans), seed, 0.02)
            # the reference divides by the variance (ConditionalLayerNorm.py:62); a scale of ~0.25
            # keeps the 7-layer pitch stack O(1)
            sd[q + "4.bias"] = normal(q + "4b", (chans,), seed, 0.02, 0.25 * b_last)
    sd[f"{prefix}.linear.weight"] = normal(f"{prefix}.lin", (1, chans), seed, 0.05)
    sd[f"{prefix}.linear.bias"] = normal(f"{prefix}.linb", (1,), seed, 0.05)


def _weight_norm_pair(sd, key, name, shape, seed, std=None, gain=1.0):
    """weight_g / weight_v as torch.nn.utils.weight_norm(dim=0) stores them."""
    v = _xavier(name + "v", shape, seed, gain) if std is None else normal(name + "v", shape, seed, std)
    norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
    g = norm * (1.0 + 0.1 * normal(name + "g", norm.shape, seed).astype(np.float64))
    sd[key + ".weight_g"] = g.astype(np.float32)
    sd[key + ".weight_v"] = v


def _glow(sd, seed):
    pre = "post_flow."
    sd[pre + "g_proj.weight"] = _xavier(pre + "gproj", (ATT, N_MEL + ATT, 5), seed)
    sd[pre + "g_proj.bias"] = normal(pre + "gprojb", (ATT,), seed, 0.02)
    C = N_MEL * 2
    perm = np.zeros((4, 4), dtype=np.float32)
    for i, j in enumerate((2, 0, 3, 1)):
        perm[i, j] = 1.0
    l_mask = np.tril(np.ones((4, 4), dtype=np.float32), -1)
    for b in range(GLOW_BLOCKS):
        a, n, c = 3 * b, 3 * b + 1, 3 * b + 2
        fa, fn, fc = (f"{pre}flows.{i}." for i in (a, n, c))
        sd[fa + "logs"] = normal(fa + "logs", (1, C, 1), seed, 0.05)
        sd[fa + "bias"] = normal(fa + "bias", (1, C, 1), seed, 0.1)
        sd[fn + "l"] = normal(fn + "l", (4, 4), seed, 0.2)
        sd[fn + "log_s"] = normal(fn + "log_s", (4,), seed, 0.1)
        sd[fn + "u"] = normal(fn + "u", (4, 4), seed, 0.2)
        sd[fn + "p"] = perm.copy()
        sd[fn + "sign_s"] = np.array([1.0, -1.0, 1.0, 1.0], dtype=np.float32) if b % 2 else np.ones(4, np.float32)
        sd[fn + "l_mask"] = l_mask.copy()
        sd[fn + "eye"] = np.eye(4, dtype=np.float32)
        sd[fc + "start.bias"] = normal(fc + "startb", (ATT,), seed, 0.02)
        _weight_norm_pair(sd, fc + "start", fc + "start", (ATT, N_MEL, 1), seed)
        # constructed as zeros in the reference (Glow.py:239-241); small values make the coupling non-trivial
        sd[fc + "end.weight"] = normal(fc + "end", (C, ATT, 1), seed, 0.02)
        sd[fc + "end.bias"] = normal(fc + "endb", (C,), seed, 0.02)
        # in_layers / res_skip_layers are shared inside each group of 4 blocks (Glow.py:325-327, :244-246);
        # the state_dict lists the same tensor under every block of the group.
        owner = f"{pre}flows.{3 * (b - b % GLOW_SHARE) + 2}."
        for i in range(GLOW_LAYERS):
            sd[fc + f"wn.in_layers.{i}.bias"] = normal(owner + f"in{i}b", (2 * ATT,), seed, 0.02)
            _weight_norm_pair(sd, fc + f"wn.in_layers.{i}", owner + f"in{i}", (2 * ATT, ATT, 5), seed)
            rs = 2 * ATT if i < GLOW_LAYERS - 1 else ATT
            sd[fc + f"wn.res_skip_layers.{i}.bias"] = normal(owner + f"rs{i}b", (rs,), seed, 0.02)
            _weight_norm_pair(sd, fc + f"wn.res_skip_layers.{i}", owner + f"rs{i}", (rs, ATT, 1), seed)
        sd[fc + "wn.cond_layer.bias"] = normal(fc + "condb", (2 * ATT * GLOW_LAYERS,), seed, 0.02)
        _weight_norm_pair(sd, fc + "wn.cond_layer", fc + "cond", (2 * ATT * GLOW_LAYERS, 2 * ATT, 1), seed)


def acoustic_state_dict(seed: int = 1234, n_lang: int = N_LANG, multispeaker: bool = True) -> dict:
    """name -> numpy array, schema of InferenceToucanTTS.ToucanTTS: multilingual + multispeaker by default; n_lang=None gives the
    single-language variant (lang_embs=None), multispeaker=False the single-speaker one (utt_embed_dim=None) - the three
    variants ToucanTTSInterface.py:55-63 tries in turn."""
    sd = {}
    e = "encoder."
    sd[e + "embed.0.weight"] = _xavier(e + "e0", (100, 62), seed)
    sd[e + "embed.0.bias"] = normal(e + "e0b", (100,), seed, 0.02)
    sd[e + "embed.2.weight"] = _xavier(e + "e2", (ATT, 100), seed)
    sd[e + "embed.2.bias"] = normal(e + "e2b", (ATT,), seed, 0.02)
    sd[e + "output_norm.weight"] = normal(e + "onw", (ATT,), seed, 0.1, 1.0)
    sd[e + "output_norm.bias"] = normal(e + "onb", (ATT,), seed, 0.1)
    if multispeaker:
        sd[e + "hs_emb_projection.weight"] = _xavier(e + "hs", (ATT, ATT + UTT), seed)
        sd[e + "hs_emb_projection.bias"] = normal(e + "hsb", (ATT,), seed, 0.02)
    if n_lang is not None:
        sd[e + "language_embedding.weight"] = normal(e + "lang", (n_lang, ATT), seed, 0.1)
    _conformer(sd, "encoder", seed, kernel=7)
    _predictor(sd, "duration_predictor", seed, 3, 3, multispeaker=multispeaker)
    # log-domain duration head (DurationPredictor.py:79): exp(x)-1 ~ 5 frames
    sd["duration_predictor.linear.weight"] = normal("durlin", (1, 256), seed, 0.03)
    sd["duration_predictor.linear.bias"] = np.array([np.log(6.0)], dtype=np.float32)
    _predictor(sd, "pitch_predictor", seed, 7, 5, multispeaker=multispeaker)
    _predictor(sd, "energy_predictor", seed, 2, 3, multispeaker=multispeaker)
    sd["pitch_embed.0.weight"] = normal("pemb", (ATT, 1, 1), seed, 0.3)
    sd["pitch_embed.0.bias"] = normal("pembb", (ATT,), seed, 0.02)
    sd["energy_embed.0.weight"] = normal("eemb", (ATT, 1, 1), seed, 0.3)
    sd["energy_embed.0.bias"] = normal("eembb", (ATT,), seed, 0.02)
    _conformer(sd, "decoder", seed, kernel=31)
    sd["feat_out.weight"] = _xavier("feat", (N_MEL, ATT), seed)
    sd["feat_out.bias"] = normal("featb", (N_MEL,), seed, 0.02)
    chans = [(256, 80), (256, 256), (256, 256), (256, 256), (80, 256)]
    for i, (co, ci) in enumerate(chans):
        sd[f"conv_postnet.postnet.{i}.0.weight"] = _xavier(f"pn{i}", (co, ci, 5), seed)
        sd[f"conv_postnet.postnet.{i}.1.weight"] = normal(f"pn{i}gw", (co,), seed, 0.1, 1.0)
        sd[f"conv_postnet.postnet.{i}.1.bias"] = normal(f"pn{i}gb", (co,), seed, 0.1)
    _glow(sd, seed)
    return sd


# --------------------------------------------------------------------------------------
# vocoders
# --------------------------------------------------------------------------------------
UP_RATES = (8, 6, 4, 2)
UP_KERNELS = (16, 12, 8, 4)
RES_KERNELS = (3, 7, 11)
RES_DILATIONS = (1, 3, 5)
VOC_CH = 512


def _voc_common(sd, seed, pre_name, ups_fmt, block_fmt, post_name, c1_fmt, c2_fmt, tag, post_gain=0.5):
    _weight_norm_pair(sd, pre_name, tag + "pre", (VOC_CH, N_MEL, 7), seed, std=1.0 / np.sqrt(N_MEL * 7))
    sd[pre_name + ".bias"] = normal(tag + "preb", (VOC_CH,), seed, 0.02)
    for i, (u, k) in enumerate(zip(UP_RATES, UP_KERNELS)):
        cin, cout = VOC_CH >> i, VOC_CH >> (i + 1)
        # ConvTranspose1d weight is (in, out, k); two taps reach each output sample
        _weight_norm_pair(sd, ups_fmt.format(i), tag + f"up{i}", (cin, cout, k), seed, std=1.0 / np.sqrt(2.0 * cin))
        sd[ups_fmt.format(i) + ".bias"] = normal(tag + f"up{i}b", (cout,), seed, 0.02)
        for j, kk in enumerate(RES_KERNELS):
            blk = block_fmt.format(3 * i + j)
            for d in range(3):
                for fmt, nm in ((c1_fmt, "c1"), (c2_fmt, "c2")):
                    key = blk + fmt.format(d)
                    _weight_norm_pair(sd, key, tag + f"{i}.{j}.{d}.{nm}", (cout, cout, kk), seed,
                                      std=0.7 / np.sqrt(cout * kk))
                    sd[key + ".bias"] = normal(tag + f"{i}.{j}.{d}.{nm}b", (cout,), seed, 0.02)
    cl = VOC_CH >> 4
    _weight_norm_pair(sd, post_name, tag + "post", (1, cl, 7), seed, std=post_gain / np.sqrt(cl * 7))
    sd[post_name + ".bias"] = normal(tag + "postb", (1,), seed, 0.02)


def hifigan_state_dict(seed: int = 4321) -> dict:
    """Avocodo/HiFiGAN generator, schema of InferenceAvocodo.HiFiGANGenerator (weight-normed)."""
    sd = {}
    _voc_common(sd, seed, "input_conv", "upsamples.{}.1", "blocks.{}.", "output_conv.1",
                "convs1.{}.1", "convs2.{}.1", "hfg.")
    # discriminator taps, unused at inference (InferenceAvocodo.py:61-62) but present in checkpoints
    for nm, c in (("out_proj_x1", 128), ("out_proj_x2", 64)):
        _weight_norm_pair(sd, nm, "hfg." + nm, (1, c, 7), seed, std=0.05)
        sd[nm + ".bias"] = normal("hfg." + nm + "b", (1,), seed, 0.02)
    return sd


def bigvgan_state_dict(seed: int = 5678) -> dict:
    """BigVGAN generator, schema of InferenceBigVGAN.BigVGAN.  The anti-alias filter buffers that
    alias_free_torch registers are NOT part of this dict (third-party names are unpinned); the
    filter is recomputed from its published formula on both sides."""
    sd = {}
    _voc_common(sd, seed, "conv_pre", "ups.{}.0", "resblocks.{}.", "conv_post",
                "convs1.{}", "convs2.{}", "bvg.", post_gain=0.12)
    for i in range(4):
        ch = VOC_CH >> (i + 1)
        for j in range(3):
            for a in range(6):
                key = f"resblocks.{3 * i + j}.activations.{a}.act."
                sd[key + "alpha"] = uniform("bvg." + key + "a", (ch,), seed, -0.5, 0.5)
                sd[key + "beta"] = uniform("bvg." + key + "b", (ch,), seed, -0.5, 0.5)
    sd["activation_post.act.alpha"] = uniform("bvg.post.a", (VOC_CH >> 4,), seed, -0.5, 0.5)
    sd["activation_post.act.beta"] = uniform("bvg.post.b", (VOC_CH >> 4,), seed, -0.5, 0.5)
    # out_proj_x1/x2 are plain (not weight-normed) convs here (InferenceBigVGAN.py:67-68)
    for nm, c in (("out_proj_x1", 128), ("out_proj_x2", 64)):
        sd[nm + ".weight"] = normal("bvg." + nm, (1, c, 7), seed, 0.05)
        sd[nm + ".bias"] = normal("bvg." + nm + "b", (1,), seed, 0.02)
    return sd


def default_utterance_embedding(seed: int = 2000) -> np.ndarray:
    return normal("default_emb", (UTT,), seed)


# --------------------------------------------------------------------------------------
# style embedding (GST): TrainingInterfaces/Spectrogram_to_Embedding/StyleEmbedding.py, GST.py
# --------------------------------------------------------------------------------------
GST_CHANS = (32, 32, 64, 64, 128, 128, 256, 256)
GST_UNITS, GST_TOKENS, GST_HEADS, GST_DIM = 256, 2000, 8, 64


def style_state_dict(seed: int = 8765) -> dict:
    """Reference-schema state dict of ``StyleEmbedding`` (keys ``gst.ref_enc.*``, ``gst.stl.*``; GST.py:31-57): kaiming-like
    conv weights, non-trivial BatchNorm running statistics, PyTorch-default GRU / Linear ranges, N(0,1) style tokens."""
    sd = {}
    cin = 1
    for i, c in enumerate(GST_CHANS):
        p = f"gst.ref_enc.convs.{3 * i}"
        sd[p + ".weight"] = normal("gst.conv%d" % i, (c, cin, 3, 3), seed, std=float(np.sqrt(2.0 / (cin * 9))))
        q = f"gst.ref_enc.convs.{3 * i + 1}"
        sd[q + ".weight"] = normal("gst.bnw%d" % i, (c,), seed, 0.1, 1.0)
        sd[q + ".bias"] = normal("gst.bnb%d" % i, (c,), seed, 0.1)
        sd[q + ".running_mean"] = normal("gst.bnm%d" % i, (c,), seed, 0.1)
        sd[q + ".running_var"] = uniform("gst.bnv%d" % i, (c,), seed, 0.5, 1.5)
        sd[q + ".num_batches_tracked"] = np.array(100, dtype=np.int64)
        cin = c
    k = float(1.0 / np.sqrt(GST_UNITS))
    for layer in range(2):
        for nm, shape in (("weight_ih", (3 * GST_UNITS, GST_UNITS)), ("weight_hh", (3 * GST_UNITS, GST_UNITS)), ("bias_ih", (3 * GST_UNITS,)),
                          ("bias_hh", (3 * GST_UNITS,))):
            sd[f"gst.ref_enc.gst.{nm}_l{layer}"] = uniform(f"gst.gru.{nm}{layer}", shape, seed, -k, k)
    sd["gst.stl.gst_embs"] = normal("gst.tokens", (GST_TOKENS, GST_DIM // GST_HEADS), seed)
    for nm, (o, i) in (("linear_q", (GST_DIM, GST_UNITS)), ("linear_k", (GST_DIM, GST_DIM // GST_HEADS)), ("linear_v", (GST_DIM, GST_DIM // GST_HEADS)),
                       ("linear_out", (GST_DIM, GST_DIM))):
        b = float(1.0 / np.sqrt(i))
        sd[f"gst.stl.mha.{nm}.weight"] = uniform("gst." + nm, (o, i), seed, -b, b)
        sd[f"gst.stl.mha.{nm}.bias"] = uniform("gst." + nm + "b", (o,), seed, -b, b)
    return sd


def reference_spectrogram(u: int, frames: int) -> np.ndarray:
    """Seeded stand-in for a log-mel spectrogram [frames, 80] (values in the range log10 mel energies take)."""
    return (normal(f"spec{u}", (frames, N_MEL), 7000 + u, 0.8) - np.float32(2.0)).astype(np.float32)
