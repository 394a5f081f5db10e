/*
 * toucan_tts.h - C ABI of the MI355X-native ToucanTTS inference kernels (libtoucan_hip.so).
 *
 * The reference (IMS-Toucan) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY.md 2.2); its "operator interface" for this path is the set of torch module forwards
 * listed in SURVEY.md section 8(a).  Each entry point below replaces one of those forwards (or a
 * fixed sub-sequence of one) and cites it.  The only caller is the build's own Python host
 * (ims-toucan-prosody-variance_amd/engine.py, via ctypes - see INTEGRATION.md for the binding).
 *
 * Conventions
 *  - All tensors are fp32 device pointers owned by the caller (torch allocations); nothing here
 *    allocates, frees or synchronises.  Every call enqueues on the hipStream_t passed last.
 *  - Activations are TIME-MAJOR and PACKED: a tensor is rows x channels with a row stride `ld`
 *    (in floats); the utterances of a batch are concatenated along the row axis.  Raggedness is
 *    described by a tile table (TtsTile[]) that the host builds once per batch and stage: each
 *    tile names the utterance it belongs to, so halo reads never cross an utterance boundary and
 *    reads outside [seq_begin, seq_end) return zero (== the reference's per-utterance zero padding).
 *  - Return value: 0 on success, negative TTS_E_* otherwise; tts_last_error() gives the text.
 *    No C++ exception crosses the boundary.
 */
#ifndef TOUCAN_TTS_H
#define TOUCAN_TTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tts_stream_t; /* hipStream_t */

#define TTS_OK 0
#define TTS_E_ARG (-1)     /* inconsistent shapes / unsupported configuration */
#define TTS_E_LAUNCH (-2)  /* hipLaunch error */

/* One tile of TILE_ROWS consecutive rows of one utterance. */
typedef struct {
  int32_t row0;      /* first packed row of the tile */
  int32_t seq_begin; /* first packed row of the utterance */
  int32_t seq_end;   /* one past the last packed row of the utterance */
  int32_t seq_id;    /* utterance index in the batch (rows of per-utterance vectors) */
} TtsTile;

/* epilogue modes of tts_conv1d */
#define TTS_MODE_LINEAR 0   /* v = act(acc + bias + seqvec + preadd)                              */
#define TTS_MODE_GLU 1      /* v = a * sigmoid(g)            (Layers/Convolution.py:47)            */
#define TTS_MODE_GATED 2    /* v = tanh(a) * sigmoid(g)      (wavenet.py:29-35)                    */
#define TTS_MODE_COUPLING 3 /* v = (aux - a) * exp(-g)       (Glow.py:260-264, reverse)            */
#define TTS_ACT_NONE 0
#define TTS_ACT_RELU 1
#define TTS_ACT_TANH 2
#define TTS_IO_X_BF16 1
#define TTS_IO_Y_BF16 2
#define TTS_IO_RES_BF16 4
#define TTS_IO_F16 8      /* the 16-bit tensors named by the three bits above are IEEE fp16 (compute 2) instead of bf16 */
#define TTS_IO_SPLIT_K 16 /* tts_conv1d, fp32 only: the caller accepts the split-K form when the grid is a handful of workgroups
                             (four interleaved partial sums: the result then depends on the grid at rounding-order level) */
#define TTS_IO_SPLIT_K_ALWAYS 32 /* tts_conv1d, fp32 only: the split-K form at EVERY grid size wherever the conv itself is eligible
                             (one arithmetic whatever the batch: the phoneme stages of the fp32 acoustic model, whose rounded durations
                             must not depend on the batch an utterance is in) */
/* tts_relpos_attention flags (fp32 kernel): the key-split form - four wavefronts share one block of 32 queries, split the keys
 * and merge (max, sum, output) in LDS: four interleaved partial sums, i.e. another rounding order than the plain form */
#define TTS_ATT_KEY_SPLIT 1        /* on grids of at most 64 workgroups (batch 1): the result then depends on the grid at rounding level */
#define TTS_ATT_KEY_SPLIT_ALWAYS 2 /* at every grid size: one arithmetic whatever the batch */
#define TTS_COMPUTE_F32 0  /* v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation */
#define TTS_COMPUTE_BF16 1 /* v_mfma_f32_32x32x16_bf16, fp32 accumulation (BASELINE.json configs[2]) */
#define TTS_COMPUTE_F16 2  /* v_mfma_f32_32x32x16_f16, fp32 accumulation (BASELINE.json configs[4]) */
#define TTS_COMPUTE_F32X3 3 /* fp32 tensors, every product as three fp16 matrix instructions on split operands (hi.hi + 2^-11 (hi.lo' + lo'.hi),
                               fp32 accumulation): ~22 significant bits per product at 16/3 of the fp32 matrix rate.  tts_conv1d only; its
                               16-bit weight argument then holds the two fp16 planes [hi | lo'] of [tap][cin_pad/8][wn][8] */
#define TTS_PRE_NONE 0
#define TTS_PRE_LRELU 1
#define TTS_PRE_SNAKE 2 /* anti-aliased SnakeBeta (see tts_snake_aa) applied while the input window is staged */

/*
 * Dense 1-D convolution over the packed time axis as an implicit GEMM on the matrix cores:
 *   acc[r, n] = sum_j sum_ci pre(x[r + j*dil - pad_left, ci]) * w[j][ci][n]
 *   y[r, n]   = alpha * epilogue(acc) + res_scale * res[r, n] (+ y[r, n] if accumulate)
 * Replaces every torch.nn.Conv1d / Linear / ConvTranspose1d on the path:
 *   Layers/MultiLayeredConv1d.py:50-51 (FFN), Layers/Attention.py:57-59,91,177 (q,k,v,out,pos),
 *   Layers/Convolution.py:25,28 (pointwise), Layers/VariancePredictor.py:70, DurationPredictor.py:67,
 *   Layers/PostNet.py:40-56 (convs), Glow.py:232-241,346-348, wavenet.py:64-82 (flow convs),
 *   InferenceBigVGAN.py:37-46 / InferenceAvocodo.py:29-43 (pre conv, transposed convs as 3-tap
 *   polyphase convs), AMP.py:22-43 / ResidualBlock.py:60-81 (dilated residual convs).
 * Weights are packed by the host as [taps][cin_pad][wn] (cin_pad multiple of 32, wn multiple of
 * the N tile; zero filled); in the dual modes wn holds the two halves [a | g], each `half_pad` wide.
 */
typedef struct {
  const float* x;      int32_t ldx;  int32_t cin;
  const void*  w;      int32_t cin_pad; int32_t wn; int32_t half_pad;
  const float* bias;   /* [cout] (dual modes: [2*cout], a then g) or NULL */
  float* y;            int32_t ldy;  int32_t cout;
  int32_t taps, dil, pad_left;
  int32_t pre_act;     float pre_slope;
  const float* snake_alpha; const float* snake_beta; const float* snake_filt; /* TTS_PRE_SNAKE: [cin], [cin], [12] */
  int32_t mode, act;   float alpha;
  const float* seqvec; int32_t ld_seqvec; /* per-utterance addend [n_seq, cout] or NULL */
  const float* preadd; int32_t ld_preadd; /* per-row addend (dual: g half at +cout) or NULL */
  const float* res;    int32_t ld_res; float res_scale;
  const float* aux;    int32_t ld_aux;    /* COUPLING: x1 */
  int32_t accumulate;
  int32_t compute;     /* TTS_COMPUTE_*: 0 fp32 MFMA (exact fp32 fma chain); 1 bf16 MFMA; 2 fp16 MFMA (both fp32 accumulate);
                          w must be packed in the matching element type */
  int32_t io_flags;    /* TTS_IO_* bits: which of x / y / res are 16-bit tensors in HBM (ld* then count 16-bit elements) and
                          whether those are bf16 or fp16 (TTS_IO_F16; a 16-bit compute mode only takes its own format) */
  const TtsTile* tiles; int32_t n_tiles; int32_t tile_rows; /* tts_conv1d_tile_rows() or the small form's 64 */
} TtsConvDesc;

/* BM (rows per tile) the conv kernel will use for this shape; build the tile table with it. */
int tts_conv1d_tile_rows(int32_t cout, int32_t mode);
/* N tile (columns) for this shape: pack weights with wn = roundup(cols, n_tile). */
int tts_conv1d_n_tile(int32_t cout, int32_t mode);
/* Small-batch form: returns 64 when this shape (with its packed width `packed_cols` = wn, or half_pad in the dual
 * modes) may also run on 64 x 64 tiles - build the tile table with 64 rows and pass tile_rows = 64 to put ~3x more
 * workgroups on the chip when rows/128 * cols/n_tile would leave CUs idle - else 0. */
int tts_conv1d_small_tile_rows(int32_t cout, int32_t mode, int32_t packed_cols);
int tts_conv1d(const TtsConvDesc* d, tts_stream_t stream);

/*
 * Fused vocoder residual step (bf16 or fp16 MFMA, fp32 accumulate; fp32 or 16-bit tensors in HBM):
 *   y = alpha * conv2(act(conv1(act(x)) + b1)) + b2) + res_scale * x   (+ y if accumulate)
 * conv1: `taps` taps, dilation `dil`; conv2: `taps` taps, dilation 1; both C -> C, 'same' zero padding per utterance.
 * act = TTS_PRE_LRELU (slope) or TTS_PRE_SNAKE (anti-aliased SnakeBeta with (alpha1,beta1) / (alpha2,beta2), filter [12]).
 * Replaces one dilation step of BigVGAN/AMP.py:53-58 (a1, c1, a2, c2, + x) or Layers/ResidualBlock.py:93-97; with
 * alpha = res_scale = 1/3 and accumulate it also forms the stage mean of InferenceBigVGAN.py:82-88.
 * Weights: bf16 / fp16 [taps][C/8][C][8] as produced for tts_conv1d(compute = 1 / 2).  C in {32, 64, 128, 256}.
 */
typedef struct {
  const float* x; int32_t ldx;
  float* y;       int32_t ldy;
  int32_t c, taps, dil;
  const void* w1; const float* b1;
  const void* w2; const float* b2;
  int32_t act; float slope;
  const float* alpha1; const float* beta1; const float* alpha2; const float* beta2; const float* filt;
  float alpha, res_scale; int32_t accumulate;
  int32_t io_bf16; /* 1: x and y are 16-bit tensors in HBM, in the format of `compute` (halves the traffic of this bandwidth-bound step) */
  const TtsTile* tiles; int32_t n_tiles; int32_t tile_rows;
  int32_t compute; /* TTS_COMPUTE_BF16 or TTS_COMPUTE_F16: element format of the weights, the LDS tiles and 16-bit x / y */
  const void* fir_tab; /* TTS_PRE_SNAKE, C <= 128: device copy of tts_snake_fir_table(filt) (the filters as matrix-core operands) */
} TtsResblockDesc;

/* The anti-alias filter as the per-lane A operands of the fused step's matrix-core FIR stages (csrc/snake_mfma.h): the banded
 * Toeplitz matrices of the 2x up-sampler and of the decimator, as fp16 fragments [4 operands][64 lanes][8 halfs] =
 * TTS_SNAKE_FIR_TABLE_BYTES bytes.  Pure host arithmetic on host pointers (no GPU, no stream); upload the result once per filter. */
#define TTS_SNAKE_FIR_TABLE_BYTES 4096
int tts_snake_fir_table(const float* filt /*[12], host*/, void* table /*host, TTS_SNAKE_FIR_TABLE_BYTES*/);

/* rows per tile the fused step uses for C channels (224 for C <= 128, 96 for C = 256): build the tile table with it */
int tts_resblock_tile_rows(int32_t c);
int tts_resblock_step(const TtsResblockDesc* d, tts_stream_t stream);

/*
 * One WaveNet layer of a PostFlow coupling block in one launch (16-bit MFMA configurations):
 *   acts = tanh(a) * sigmoid(g) with [a | g] = in_layer(h) + bias + cond   (5 taps, 192 -> 384; wavenet.py:104-110, :29-35)
 *   hs_out = hs_in + res_skip_layer(acts) + bias                           (1 tap, 192 -> cout2; wavenet.py:112-118)
 * hs = [hidden state h (192) | skip sum (192)] per packed row.  cout2 = 384: both halves are updated; cout2 = 192 (the last
 * layer): only the skip half is written.  hs_out must be another buffer than hs_in (the 5-tap conv reads two frames either side
 * of a tile).  Weights as packed for tts_conv1d(compute 1 / 2): w1 [5][24][384][8] (columns a | g), w2 [1][24][cout2][8].
 */
typedef struct {
  const float* hs_in;  int32_t ld_in;
  float* hs_out;       int32_t ld_out;
  const float* cond;   int32_t ld_cond;  /* this layer's 384 conditioning columns (a | g) of every row */
  const void* w1;      const float* b1;  /* bias [384]: a then g */
  const void* w2;      const float* b2;  /* bias [cout2] */
  int32_t cout2;       /* 384 or 192 */
  int32_t compute;     /* TTS_COMPUTE_BF16 or TTS_COMPUTE_F16 */
  const TtsTile* tiles; int32_t n_tiles; int32_t tile_rows; /* 64 */
} TtsWavenetDesc;
int tts_wavenet_layer(const TtsWavenetDesc* d, tts_stream_t stream);

/* Fused Conformer feed-forward module at kernel size 1 (Layers/EncoderLayer.py:84-90 and :128-136 with
 * Layers/MultiLayeredConv1d.py:50-51 and Layers/LayerNorm.py:24-36), one launch for the 16-bit MFMA configurations:
 *   y = [LayerNorm_post]( x + alpha * (W2 relu(W1 LayerNorm(x) + b1) + b2) )       rows are independent, y may be x
 * x, y: fp32 [rows, 192]; ln_g / ln_b: the module's norm; post_g / post_b: the block's norm_final (NULL: none); eps for both.
 * w: both weight matrices and b1 in the kernel's fragment order, 28 KB per 32 hidden channels:
 *   chunk c = [12 fragments of W1 | 6 x 2 fragments of W2 | 4 fragments of b1], one 16-bit fragment = [64 lanes][8]:
 *     W1 fragment ks   : lane (lk = lane / 32, r = lane % 32), element i  =  W1[32 c + r][16 ks + 8 lk + i]
 *     W2 fragment j, ab: lane (lk, r), element i  =  W2[32 j + r][32 c + 16 ab + (i < 4 ? 4 lk + i : 8 + 4 lk + i - 4)]
 *     b1 fragment q    : lane (lk, r), four fp32  =  b1[32 c + 8 q + 4 lk + 0 .. 3]   (the same for every r)
 *   (W1 [hidden, 192] = w_1.weight[:, :, 0], W2 [192, hidden] = w_2.weight[:, :, 0]; the k order of a W2 fragment is the order in
 *   which the first product's accumulator registers hold the hidden channels).
 * hidden: multiple of 32; compute: TTS_COMPUTE_BF16 or TTS_COMPUTE_F16 (the format of w). */
typedef struct {
  const float* x;      int32_t ldx;
  float* y;            int32_t ldy;
  int32_t rows;        int32_t channels;   /* 192 */
  const float* ln_g;   const float* ln_b;
  const void* w;       const float* b2;    /* [192] */
  const float* post_g; const float* post_b;
  int32_t hidden;      int32_t compute;
  float alpha;         float eps;
} TtsFfnDesc;
int tts_ffn_fused(const TtsFfnDesc* d, tts_stream_t stream);

/* y[r,:] = LayerNorm(x[r,:]) * g + b over `c` channels, eps as given. Layers/LayerNorm.py:24-36 (eps 1e-12). */
int tts_layernorm(const float* x, int32_t ldx, float* y, int32_t ldy, const float* gamma, const float* beta,
                  int32_t rows, int32_t c, float eps, tts_stream_t stream);

/* Conditional layer norm with the reference's quirk: y = s[u]*(x-mean)/var + b[u] (divide by the
 * VARIANCE, no eps, no sqrt).  Layers/ConditionalLayerNorm.py:52-67.  scale/shift are [n_seq, c]. */
int tts_cond_layernorm(const float* x, int32_t ldx, float* y, int32_t ldy, const float* scale, const float* shift,
                       int32_t c, const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, tts_stream_t stream);

/* The scale / shift MLPs of every ConditionalLayerNorm in one launch (Layers/ConditionalLayerNorm.py:26-35, 54-55:
 * Linear(d_in,d_in) Tanh Linear(d_in,d_out) Tanh Linear(d_out,d_out) applied to the normalised utterance embedding).
 * e: [n_seq, d_in]; out: [n_mlp, n_seq, d_out] (MLP m's rows are contiguous, ready to be tts_cond_layernorm's scale / shift);
 * weights: n_mlp blocks of tts_cln_mlp_weight_floats(d_in, d_out) floats, each
 *   [W0^T (d_in x d_in) | b0 (d_in) | W1^T (d_in x d_out) | b1 (d_out) | W2^T (d_out x d_out) | b2 (d_out)], W^T = [in][out].
 * d_in <= d_out <= 256. */
int64_t tts_cln_mlp_weight_floats(int32_t d_in, int32_t d_out);
int tts_cln_mlp(const float* e, int32_t n_seq, int32_t d_in, int32_t d_out, const float* weights, int32_t n_mlp, float* out,
                tts_stream_t stream);

/* Row-wise L2 normalisation (torch.nn.functional.normalize, eps 1e-12). InferenceToucanTTS.py:202, Conformer.py:132. */
int tts_l2_normalize(const float* x, float* y, int32_t rows, int32_t c, tts_stream_t stream);

/* GroupNorm over (c/groups channels x all frames of one utterance) + optional tanh + optional residual; two launches
 * (deterministic partial sums, then apply) for the whole batch. Layers/PostNet.py:44-56.  seq_begin/seq_end are device
 * arrays [n_seq]; max_len = longest utterance; workspace: tts_groupnorm_workspace_floats(n_seq, max_len, groups) floats. */
int tts_groupnorm(const float* x, int32_t ldx, float* y, int32_t ldy, const float* gamma, const float* beta,
                  int32_t c, int32_t groups, float eps, int32_t apply_tanh, const float* res, int32_t ld_res,
                  const int32_t* seq_begin, const int32_t* seq_end, int32_t n_seq, int32_t max_len, float* workspace,
                  tts_stream_t stream);
int64_t tts_groupnorm_workspace_floats(int32_t n_seq, int32_t max_len, int32_t groups);

/* Relative-position multi-head self-attention, flash style (scores never reach HBM):
 *   s[i,j] = ((q_i+u_h).k_j + (q_i+v_h).P[i-j]) / sqrt(dk); softmax over the utterance's keys; ctx = s.v
 * qkv: [rows, 3*h*dk] (q|k|v), ptab: [2*pmax-1, h*dk] with row (pmax-1+p) = linear_pos(pe(p)).
 * Layers/Attention.py:159-198 (rel_shift :138-157 folded into the index i-j), :66-92.
 * Matrix-core kernel (v_mfma_f32_32x32x2_f32, exact fp32 products); tile_rows must be 128.  flags: 0 = the plain form (one
 * accumulation order at every grid size), or TTS_ATT_KEY_SPLIT / TTS_ATT_KEY_SPLIT_ALWAYS. */
int tts_relpos_attention(const float* qkv, int32_t ld_qkv, const float* ptab, int32_t pmax, const float* bias_u,
                         const float* bias_v, float* ctx, int32_t ld_ctx, int32_t heads, int32_t dk,
                         const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, int32_t flags, tts_stream_t stream);
/* The same attention for the 16-bit configurations: the three contractions on v_mfma_f32_32x32x16_f16 (q + u, q + v, k, v, the
 * table and the probabilities rounded to fp16 - in the bf16 configuration too; fp32 scores, statistics and output).  Same
 * arguments; tile_rows must be 128. */
int tts_relpos_attention_f16(const float* qkv, int32_t ld_qkv, const float* ptab, int32_t pmax, const float* bias_u,
                             const float* bias_v, float* ctx, int32_t ld_ctx, int32_t heads, int32_t dk,
                             const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, tts_stream_t stream);

/* Depthwise conv over time (k taps, zero padded per utterance) + folded BatchNorm(eval) + Swish on the
 * GLU output. Layers/Convolution.py:50-51, Swish.py:18.  w: [k][c] (BN folded), b: [c]. */
int tts_dwconv_swish(const float* x, int32_t ldx, float* y, int32_t ldy, const float* w, const float* b, int32_t c,
                     int32_t k, const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, tts_stream_t stream);

/* Duration head: d = clamp(round(exp(x) - 1), 0) (round half to even). Layers/DurationPredictor.py:79. */
int tts_duration_from_log(const float* logd, int32_t* dur, int32_t n, tts_stream_t stream);

/* Linguistic overrides + prosody scaling for one batch, one workgroup per utterance.
 * InferenceToucanTTS.py:214-227 and _scale_variance :333-343. text is [rows, 62]. */
int tts_prosody_control(const float* text, int32_t ld_text, float* pitch, float* energy, int32_t* dur,
                        const int32_t* seq_begin, const int32_t* seq_end, int32_t n_seq, float duration_scale,
                        float pitch_scale, float energy_scale, float pause_scale, tts_stream_t stream);

/* LengthRegulator: frame f of utterance u copies phoneme row src(f) (exclusive scan of durations), fused with
 * the pitch/energy embedding add (InferenceToucanTTS.py:230-235, Layers/LengthRegulator.py:37-61).
 * Writes up[f,:] (ld_up) and, if dec_in != NULL, dec_in[f,:] = up * dec_scale (Conformer.py:116 x*sqrt(d)). */
int tts_length_regulate(const float* enc, int32_t ld_enc, const float* pitch, const float* energy,
                        const float* wp, const float* bp, const float* we, const float* be, const int32_t* dur,
                        const int32_t* phone_begin, const int32_t* phone_end, const int32_t* frame_begin,
                        int32_t n_seq, int32_t max_frames, int32_t max_phones, int32_t c, float* up, int32_t ld_up,
                        float* dec_in, int32_t ld_dec, float dec_scale, tts_stream_t stream);

/* Glow reverse step after the coupling: InvConvNear^-1 (4x4 on channel groups) then ActNorm^-1, in place.
 * Glow.py:93-128 (regrouping :102-103,:126-127) and :30-31. x: [rows, c] with c = 160. */
int tts_glow_invconv_actnorm(float* x, int32_t ldx, int32_t rows, int32_t c, const float* winv /*[4][4]*/,
                             const float* an_bias, const float* an_logs, tts_stream_t stream);

/* Anti-aliased SnakeBeta: 2x Kaiser-sinc up (12 taps), x + sin^2(x e^a)/(e^b + 1e-9), 2x low-pass down.
 * BigVGAN/Snake.py:56-69 + alias_free_torch Activation1d (third party, PARITY UNPINNED - see DESIGN.md). */
int tts_snake_aa(const float* x, int32_t ldx, float* y, int32_t ldy, const float* alpha, const float* beta,
                 const float* filt /*[12]*/, int32_t c, const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows,
                 int32_t io_flags /* TTS_IO_X_BF16 | TTS_IO_Y_BF16 | TTS_IO_F16 */, tts_stream_t stream);

/* Final vocoder conv: wav[r] = tanh(b + sum_j sum_ci pre(x[r+j-3, ci]) w[j][ci]); pre = LeakyReLU(slope) or none.
 * InferenceAvocodo.py:52-59, InferenceBigVGAN.py:92-95. */
int tts_conv_post(const float* x, int32_t ldx, int32_t cin, const float* w /*[7][cin]*/, float bias, int32_t pre_act,
                  float pre_slope, float* wav, const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows,
                  int32_t io_flags /* TTS_IO_X_BF16 | TTS_IO_F16 */, tts_stream_t stream);

/* BigVGAN's last two ops in one launch: anti-aliased SnakeBeta (activation_post) then the 7-tap output conv + tanh
 * (InferenceBigVGAN.py:90-95).  The activated tensor exists only in LDS.  cin must be 32; tile_rows must be
 * tts_conv_post_snake_tile_rows() (250: a 256-row window = 8 streamed groups of 32 frames per channel). */
int tts_conv_post_snake_tile_rows(void);
int tts_conv_post_snake(const float* x, int32_t ldx, int32_t cin, const float* w /*[7][cin]*/, float bias, const float* alpha,
                        const float* beta, const float* filt /*[12]*/, float* wav, const TtsTile* tiles, int32_t n_tiles,
                        int32_t tile_rows, int32_t io_flags /* TTS_IO_X_BF16 | TTS_IO_F16 */, tts_stream_t stream);

/* dst[i,:] = src[idx[i],:] (embedding lookup, Conformer.py:112-114 language_embedding). */
int tts_gather_rows(const float* src, int32_t ld_src, const int32_t* idx, float* dst, int32_t ld_dst, int32_t n,
                    int32_t c, tts_stream_t stream);

/* ---- per-speaker path: GST style embedding and log-mel front end (set_utterance_embedding(path), ToucanTTSInterface.py:103-114) ----
 * The Conv2d stack (as banded 2-tap convs), the projections, the windowed DFT and the mel projection run through tts_conv1d
 * (ims-toucan-prosody-variance_amd/style.py packs them); these four entries are the rest. */

/* One layer of torch.nn.GRU (batch_first, zero initial state) over `batch` sequences of `steps` rows: x [batch*steps, in_dim] -> y
 * [batch*steps, hidden] (all hidden states).  Weights transposed: w_ih_t [in_dim][3*hidden], w_hh_t [hidden][3*hidden], gate order
 * r, z, n.  hidden <= 256.  TrainingInterfaces/Spectrogram_to_Embedding/GST.py:141,158 (ReferenceEncoder.gst). */
int tts_gru_layer(const float* x, int32_t ldx, int32_t batch, int32_t steps, int32_t in_dim, int32_t hidden, const float* w_ih_t,
                  const float* w_hh_t, const float* b_ih, const float* b_hh, float* y, int32_t ldy, tts_stream_t stream);
/* Style-token attention, one query per utterance: ctx[b, h] = softmax_n(q[b,h] . k[n,h] / sqrt(dk)) v[n,h].  q [batch, heads*dk]
 * (projected), k / v [n_tokens, heads*dk] (projected tanh(tokens)), dk <= 8.  GST.py:205-219, Layers/Attention.py:66-92. */
int tts_style_tokens(const float* q, const float* k, const float* v, int32_t batch, int32_t n_tokens, int32_t heads, int32_t dk, float* ctx,
                     tts_stream_t stream);
/* y[r, c] = |x[r, c] + i x[r, bins + c]|: magnitude of a spectrum stored as [re | im].  AudioPreprocessor.py:109-110 (np.abs(stft)). */
int tts_complex_magnitude(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t rows, int32_t bins, tts_stream_t stream);
/* y = log10(max(eps, x)).  AudioPreprocessor.py:117. */
int tts_log10_floor(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t rows, int32_t c, float eps, tts_stream_t stream);

/* =====================================================================================================================
 * Stage API: a handle that owns the packed weights and the workspace, and one call per stage of the reference's forward pass
 * (csrc/pipeline.hip sequences the kernels above in C++).  One handle per (process, device); calls on a
 * handle are serialised by the caller's stream.  All pointers are device pointers unless marked host.  Allocation happens only
 * inside the handle (weights, workspace arenas that grow to the largest batch seen, cached tile tables).
 *
 * Mirrors: tts_encoder = Conformer.forward on the phoneme features (Layers/Conformer.py:92-134, InferenceToucanTTS.py:202-206);
 * tts_variance_predictors = VariancePredictor / DurationPredictor (InferenceToucanTTS.py:209-211); tts_control_and_regulate =
 * the control loop, _scale_variance and the LengthRegulator (InferenceToucanTTS.py:214-235, :333-343, LengthRegulator.py:37-61);
 * tts_decoder = decoder Conformer + feat_out (:238-239); tts_postnet (:241, PostNet.py:62-74); tts_postflow = Glow.forward(infer)
 * with the noise as an explicit input (:244-248, Glow.py:342-391); tts_vocoder_* = InferenceBigVGAN.py:72-95 /
 * InferenceAvocodo.py:69-80; tts_synthesize_batch = ToucanTTS.forward + the vocoder for a ragged batch
 * (ToucanTTSInterface.py:157-169 run once per utterance).
 * ===================================================================================================================== */
typedef struct TtsHandle TtsHandle;

typedef struct {
  int32_t multilingual;       /* checkpoint has encoder.language_embedding (ToucanTTSInterface.py:55-60) */
  int32_t multispeaker;       /* checkpoint has utterance-embedding conditioning (else plain LayerNorm predictors, :61-63) */
  int32_t vocoder;            /* 0 none, 1 Avocodo / HiFiGAN generator, 2 BigVGAN */
  int32_t precision;          /* TTS_COMPUTE_*: MFMA path of the GEMMs that carry a 16-bit weight copy; statistics stay fp32 */
  int32_t small_tile_blocks;  /* conv grids below this many workgroups use the 64-row small-batch tiles (0: default 1536) */
  float post_bias;            /* bias of the vocoder's output conv */
} TtsConfig;

int tts_create(const TtsConfig* cfg, TtsHandle** out);
int tts_destroy(TtsHandle* h);

/* Upload one packed tensor under `name` (a later call with the same name replaces it).  dtype: 0 f32, 1 bf16, 2 f16, 4 u8 are
 * copied to the device; 3 = int32 host metadata (the descriptor fields of a packed conv, kept on the host).
 * A packed conv "<n>" is the group "<n>.w" [taps][cin_pad][wn] f32, optional "<n>.w16" [taps][cin_pad/8][wn][8], optional
 * "<n>.bias", and "<n>.meta" = int32[16] {mode, taps, dil, pad_left, cin, cin_pad, cout, wn, half_pad, tile_rows,
 * small_tile_rows, small_only, n_tile, compute16, algo_taps, 0} (see tts_conv1d).  Names: INTEGRATION.md lists them; the host packer is
 * ims-toucan-prosody-variance_amd/native.py. */
int tts_load_weights(TtsHandle* h, const char* name, const void* host_ptr, const int64_t* shape, int32_t ndim, int32_t dtype);

/* Upper bound of the workspace a batch of B utterances of at most Lmax phonemes / Tmax frames will claim (bytes): the sum of what
 * the stage entries reserve for such a batch (the same expressions), growth slack of the arenas included.  Weights, the position
 * tables and the tile tables are not workspace. */
int64_t tts_workspace_bytes(const TtsHandle* h, int32_t B, int32_t Lmax, int32_t Tmax);
/* Bytes the handle's workspace arenas hold right now (after a batch: what that batch - and every larger one before it - claimed). */
int64_t tts_workspace_claimed(const TtsHandle* h);
/* Tile tables the handle has built so far: into a batch's table arena (stream-ordered copies, no allocation, no synchronisation:
 * every layout the first time it is seen) / with a permanent device allocation (a layout seen a second time: a benchmark's fixed
 * batch).  Real traffic - new utterance lengths in every batch - only ever counts in the first. */
int tts_table_stats(const TtsHandle* h, int64_t* arena_tables, int64_t* cached_tables);

/* text: packed phoneme features [sum L, 62]; utt_emb [B, 64] (NULL for the single-speaker variant); lang_ids [B] (NULL: no
 * language embedding); phone_lengths: host [B].  Starts a batch: later stages work on the handle's state. */
int tts_encoder(TtsHandle* h, const float* text, const float* utt_emb, const int32_t* lang_ids, const int32_t* phone_lengths /*host*/,
                int32_t B, tts_stream_t stream);
/* gold_*: packed per-phoneme values replacing a prediction (ToucanTTSInterface.py:139-141), or NULL to predict. */
int tts_variance_predictors(TtsHandle* h, const float* gold_pitch, const float* gold_energy, const int32_t* gold_durations,
                            tts_stream_t stream);
/* Applies the linguistic overrides and the four scaling factors, reads the integer durations back (the pass's one host round
 * trip), sizes the frame buffers and expands the phoneme rows.  frame_counts (host [B], may be NULL) receives the frames per utterance. */
int tts_control_and_regulate(TtsHandle* h, float duration_scale, float pitch_scale, float energy_scale, float pause_scale,
                             int32_t* frame_counts /*host*/, tts_stream_t stream);
int tts_decoder(TtsHandle* h, tts_stream_t stream);
int tts_postnet(TtsHandle* h, tts_stream_t stream);
/* z_noise: the 0.8 N(0,1) sample (Glow.py:363) in the squeezed frame layout, [total_frames / 2, 160]: row frame_begin[u]/2 + i of
 * utterance u holds frames 2i and 2i+1 (80 channels each). */
int tts_postflow(TtsHandle* h, const float* z_noise, tts_stream_t stream);
/* Where the batch's mel is: packed rows of 80 channels with row stride *ld; utterance u starts at frame_begins[u] (host [B], even)
 * and has frame_counts[u] frames (an odd count loses its last frame once the flow has run, glow_utils.py:31-32). */
int tts_mel(TtsHandle* h, const float** mel, int32_t* ld, int32_t* frame_begins /*host*/, int32_t* frame_counts /*host*/);
/* Copies the batch's mel (all packed rows, 80 channels) into dst with row stride ld_dst (floats). */
int tts_copy_mel(TtsHandle* h, float* dst, int32_t ld_dst, tts_stream_t stream);
/* Packed per-phoneme durations / pitch / energy of the batch after the control step (device pointers into the workspace). */
int tts_prosody(TtsHandle* h, const int32_t** durations, const float** pitch, const float** energy);
/* The same, copied into caller buffers of sum(phone_lengths) elements each (any of them may be NULL). */
int tts_copy_prosody(TtsHandle* h, int32_t* durations, float* pitch, float* energy, tts_stream_t stream);
/* Roofline instrumentation (bench.py): while enabled, every launch of the matrix-core kernel class `select` (NULL / "": every
 * class; names as in profiling.py: "resblock_step<64>", "conv1d_bf16<128x128>", ...) made by the stage entries is bracketed by
 * HIP events on the launch stream.  tts_profile(h, ...) also clears earlier records.  tts_profile_read waits for record `index`
 * and returns its duration and its algorithmic work (FLOPs, bytes of the tensors it reads and writes once + weights, output elements). */
int tts_profile(TtsHandle* h, int32_t enable, const char* select);
int32_t tts_profile_count(TtsHandle* h);
int tts_profile_read(TtsHandle* h, int32_t index, char* name, int32_t name_cap, double* ms, double* flops, double* bytes, double* elems);
/* mel: packed [rows, 80] with row stride ld_mel; wav: packed, utterance u at 384 * frame_begins[u], 384 * frame_counts[u] samples. */
int tts_vocoder_bigvgan(TtsHandle* h, const float* mel, int32_t ld_mel, const int32_t* frame_begins /*host*/,
                        const int32_t* frame_counts /*host*/, int32_t B, float* wav, tts_stream_t stream);
int tts_vocoder_hifigan(TtsHandle* h, const float* mel, int32_t ld_mel, const int32_t* frame_begins /*host*/,
                        const int32_t* frame_counts /*host*/, int32_t B, float* wav, tts_stream_t stream);
/* The whole pass.  z_noise NULL: the flow is skipped; wav NULL: no vocoder.  *wav_needed (may be NULL) receives the samples the
 * packed waveform takes; a too small wav_capacity is TTS_E_ARG (the mel stays available through tts_mel). */
int tts_synthesize_batch(TtsHandle* h, const float* text, const float* utt_emb, const int32_t* lang_ids, const int32_t* phone_lengths /*host*/,
                         int32_t B, const float* gold_pitch, const float* gold_energy, const int32_t* gold_durations, float duration_scale,
                         float pitch_scale, float energy_scale, float pause_scale, const float* z_noise, int32_t* frame_begins /*host*/,
                         int32_t* frame_counts /*host*/, float* wav, int64_t wav_capacity, int64_t* wav_needed /*host*/, tts_stream_t stream);

const char* tts_last_error(void);
/* Self-check: synchronises the device and returns how many words of the residual-step work queues are not zero (every
 * tts_resblock_step launch draws its tiles from one 16-word slot and must leave it zero again).  0 = clean; negative = HIP error.
 * The test suite calls it after every GPU test. */
int tts_diag_queue_nonzero(void);
/* Work-queue slots handed out on the current device so far: one per stream that has launched tts_resblock_step, plus one per
 * launch recorded into a HIP graph (a recorded launch keeps its slot for the life of the process; slots are never shared between
 * launches that can be in flight together).  Tests only. */
int tts_diag_queue_slots_used(void);
/* Bumped whenever a struct layout or a prototype in this header changes; a binding must refuse a library that reports
 * another value (the descriptors are passed by layout, a stale build would read garbage). */
#define TTS_ABI_VERSION 14
int tts_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TOUCAN_TTS_H */
