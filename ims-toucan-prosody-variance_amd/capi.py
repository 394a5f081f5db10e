"""ctypes binding of include/toucan_tts.h (libtoucan_hip.so).

The library is mandatory: there is no CPU or PyTorch fallback anywhere in the product path.
``lib()`` raises if the shared object is missing or does not export the full ABI.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TOUCAN_HIP_LIB") or os.path.join(_PKG, "libtoucan_hip.so")  # the override is for A/B builds of the library only

MODE_LINEAR, MODE_GLU, MODE_GATED, MODE_COUPLING = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
PRE_NONE, PRE_LRELU, PRE_SNAKE = 0, 1, 2
COMPUTE_F32, COMPUTE_BF16, COMPUTE_F16 = 0, 1, 2
COMPUTE_F32X3 = 3  # fp32 tensors, every product as three fp16 MFMAs on split operands (include/toucan_tts.h TTS_COMPUTE_F32X3)

_p = C.c_void_p
_i = C.c_int32
_f = C.c_float


class TtsTile(C.Structure):
    _fields_ = [("row0", _i), ("seq_begin", _i), ("seq_end", _i), ("seq_id", _i)]


class TtsConvDesc(C.Structure):
    _fields_ = [
        ("x", _p), ("ldx", _i), ("cin", _i),
        ("w", _p), ("cin_pad", _i), ("wn", _i), ("half_pad", _i),
        ("bias", _p),
        ("y", _p), ("ldy", _i), ("cout", _i),
        ("taps", _i), ("dil", _i), ("pad_left", _i),
        ("pre_act", _i), ("pre_slope", _f),
        ("snake_alpha", _p), ("snake_beta", _p), ("snake_filt", _p),
        ("mode", _i), ("act", _i), ("alpha", _f),
        ("seqvec", _p), ("ld_seqvec", _i),
        ("preadd", _p), ("ld_preadd", _i),
        ("res", _p), ("ld_res", _i), ("res_scale", _f),
        ("aux", _p), ("ld_aux", _i),
        ("accumulate", _i),
        ("compute", _i),
        ("io_flags", _i),
        ("tiles", _p), ("n_tiles", _i), ("tile_rows", _i),
    ]


class TtsResblockDesc(C.Structure):
    _fields_ = [
        ("x", _p), ("ldx", _i),
        ("y", _p), ("ldy", _i),
        ("c", _i), ("taps", _i), ("dil", _i),
        ("w1", _p), ("b1", _p),
        ("w2", _p), ("b2", _p),
        ("act", _i), ("slope", _f),
        ("alpha1", _p), ("beta1", _p), ("alpha2", _p), ("beta2", _p), ("filt", _p),
        ("alpha", _f), ("res_scale", _f), ("accumulate", _i),
        ("io_bf16", _i),
        ("tiles", _p), ("n_tiles", _i), ("tile_rows", _i),
        ("compute", _i),
        ("fir_tab", _p),
    ]


class TtsWavenetDesc(C.Structure):
    _fields_ = [
        ("hs_in", _p), ("ld_in", _i),
        ("hs_out", _p), ("ld_out", _i),
        ("cond", _p), ("ld_cond", _i),
        ("w1", _p), ("b1", _p),
        ("w2", _p), ("b2", _p),
        ("cout2", _i),
        ("compute", _i),
        ("tiles", _p), ("n_tiles", _i), ("tile_rows", _i),
    ]


class TtsFfnDesc(C.Structure):
    _fields_ = [
        ("x", _p), ("ldx", _i),
        ("y", _p), ("ldy", _i),
        ("rows", _i), ("channels", _i),
        ("ln_g", _p), ("ln_b", _p),
        ("w", _p), ("b2", _p),
        ("post_g", _p), ("post_b", _p),
        ("hidden", _i), ("compute", _i),
        ("alpha", _f), ("eps", _f),
    ]


class TtsConfig(C.Structure):
    _fields_ = [("multilingual", _i), ("multispeaker", _i), ("vocoder", _i), ("precision", _i), ("small_tile_blocks", _i), ("post_bias", _f)]


IO_X_BF16, IO_Y_BF16, IO_RES_BF16, IO_F16 = 1, 2, 4, 8
IO_SPLIT_K = 16  # tts_conv1d, fp32: the caller accepts the split-K form on small grids (the frame stages of the fp32 acoustic model do)
IO_SPLIT_K_ALWAYS = 32  # ... at every grid size (its phoneme stages: one arithmetic upstream of the rounded durations whatever the batch)
ATT_KEY_SPLIT, ATT_KEY_SPLIT_ALWAYS = 1, 2  # tts_relpos_attention flags (include/toucan_tts.h)

# symbol -> (restype, argtypes); mirrors include/toucan_tts.h one to one
PROTOTYPES = {
    "tts_last_error": (C.c_char_p, []),
    "tts_abi_version": (C.c_int, []),
    "tts_diag_queue_nonzero": (C.c_int, []),
    "tts_diag_queue_slots_used": (C.c_int, []),
    "tts_conv1d_tile_rows": (C.c_int, [_i, _i]),
    "tts_conv1d_n_tile": (C.c_int, [_i, _i]),
    "tts_conv1d_small_tile_rows": (C.c_int, [_i, _i, _i]),
    "tts_conv1d": (C.c_int, [C.POINTER(TtsConvDesc), _p]),
    "tts_resblock_step": (C.c_int, [C.POINTER(TtsResblockDesc), _p]),
    "tts_resblock_tile_rows": (C.c_int, [_i]),
    "tts_snake_fir_table": (C.c_int, [_p, _p]),
    "tts_wavenet_layer": (C.c_int, [C.POINTER(TtsWavenetDesc), _p]),
    "tts_ffn_fused": (C.c_int, [C.POINTER(TtsFfnDesc), _p]),
    "tts_layernorm": (C.c_int, [_p, _i, _p, _i, _p, _p, _i, _i, _f, _p]),
    "tts_cond_layernorm": (C.c_int, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _p]),
    "tts_cln_mlp_weight_floats": (C.c_int64, [_i, _i]),
    "tts_cln_mlp": (C.c_int, [_p, _i, _i, _i, _p, _i, _p, _p]),
    "tts_l2_normalize": (C.c_int, [_p, _p, _i, _i, _p]),
    "tts_groupnorm": (C.c_int, [_p, _i, _p, _i, _p, _p, _i, _i, _f, _i, _p, _i, _p, _p, _i, _i, _p, _p]),
    "tts_groupnorm_workspace_floats": (C.c_int64, [_i, _i, _i]),
    "tts_relpos_attention": (C.c_int, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _p, _i, _i, _i, _p]),
    "tts_relpos_attention_f16": (C.c_int, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _p, _i, _i, _p]),
    "tts_dwconv_swish": (C.c_int, [_p, _i, _p, _i, _p, _p, _i, _i, _p, _i, _i, _p]),
    "tts_duration_from_log": (C.c_int, [_p, _p, _i, _p]),
    "tts_prosody_control": (C.c_int, [_p, _i, _p, _p, _p, _p, _p, _i, _f, _f, _f, _f, _p]),
    "tts_length_regulate": (C.c_int, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _i, _f, _p]),
    "tts_glow_invconv_actnorm": (C.c_int, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "tts_snake_aa": (C.c_int, [_p, _i, _p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _p]),
    "tts_conv_post": (C.c_int, [_p, _i, _i, _p, _f, _i, _f, _p, _p, _i, _i, _i, _p]),
    "tts_conv_post_snake_tile_rows": (C.c_int, []),
    "tts_conv_post_snake": (C.c_int, [_p, _i, _i, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "tts_gather_rows": (C.c_int, [_p, _i, _p, _p, _i, _i, _i, _p]),
    # per-speaker path (csrc/style.hip)
    "tts_gru_layer": (C.c_int, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    "tts_style_tokens": (C.c_int, [_p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "tts_complex_magnitude": (C.c_int, [_p, _i, _p, _i, _i, _i, _p]),
    "tts_log10_floor": (C.c_int, [_p, _i, _p, _i, _i, _i, _f, _p]),
    # stage API (csrc/pipeline.hip)
    "tts_create": (C.c_int, [C.POINTER(TtsConfig), C.POINTER(_p)]),
    "tts_destroy": (C.c_int, [_p]),
    "tts_load_weights": (C.c_int, [_p, C.c_char_p, _p, C.POINTER(C.c_int64), _i, _i]),
    "tts_workspace_bytes": (C.c_int64, [_p, _i, _i, _i]),
    "tts_workspace_claimed": (C.c_int64, [_p]),
    "tts_table_stats": (C.c_int, [_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tts_encoder": (C.c_int, [_p, _p, _p, _p, _p, _i, _p]),
    "tts_variance_predictors": (C.c_int, [_p, _p, _p, _p, _p]),
    "tts_control_and_regulate": (C.c_int, [_p, _f, _f, _f, _f, _p, _p]),
    "tts_decoder": (C.c_int, [_p, _p]),
    "tts_postnet": (C.c_int, [_p, _p]),
    "tts_postflow": (C.c_int, [_p, _p, _p]),
    "tts_mel": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_i), _p, _p]),
    "tts_prosody": (C.c_int, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "tts_copy_mel": (C.c_int, [_p, _p, _i, _p]),
    "tts_copy_prosody": (C.c_int, [_p, _p, _p, _p, _p]),
    "tts_profile": (C.c_int, [_p, _i, C.c_char_p]),
    "tts_profile_count": (_i, [_p]),
    "tts_profile_read": (C.c_int, [_p, _i, C.c_char_p, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tts_vocoder_bigvgan": (C.c_int, [_p, _p, _i, _p, _p, _i, _p, _p]),
    "tts_vocoder_hifigan": (C.c_int, [_p, _p, _i, _p, _p, _i, _p, _p]),
    "tts_synthesize_batch": (C.c_int, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _f, _f, _f, _f, _p, _p, _p, _p, C.c_int64, C.POINTER(C.c_int64), _p]),
}

_LIB = None
ABI_VERSION = 14  # include/toucan_tts.h TTS_ABI_VERSION: struct layouts and prototypes mirrored below


class ToucanHipError(RuntimeError):
    pass


def lib():
    """Load libtoucan_hip.so and bind every symbol of the ABI; raise loudly if anything is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ToucanHipError(
            f"{LIB_PATH} not found: the HIP extension is mandatory (no CPU fallback). "
            f"Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    # PyTorch-ROCm ships its own libamdhip64; load it FIRST so that libtoucan_hip.so binds to the same HIP
    # runtime (same soname) - two runtimes in one process cannot share streams or device pointers.
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    _assert_single_hip_runtime()
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise ToucanHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = handle.tts_abi_version()
    if got != ABI_VERSION:
        raise ToucanHipError(f"{LIB_PATH} reports ABI version {got}, this binding was written for {ABI_VERSION}: rebuild the "
                             f"library (python -c 'import __graft_entry__ as g; g.build()')")
    _LIB = handle
    return handle


def _assert_single_hip_runtime():
    try:
        with open("/proc/self/maps") as f:
            paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    if len(paths) > 1:
        raise ToucanHipError(f"two HIP runtimes are mapped ({sorted(paths)}); import torch before loading libtoucan_hip.so")


CALLS = 0  # ABI calls checked so far (bench.py reports the calls of one pass)


def check(rc, what=""):
    global CALLS
    CALLS += 1
    if rc != 0:
        msg = lib().tts_last_error().decode("utf-8", "replace")
        raise ToucanHipError(f"{what} failed with code {rc}: {msg}")
