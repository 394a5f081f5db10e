// extern "C" boundary of libtoucan_hip.so (declared in include/toucan_tts.h).  No torch types, no
// exceptions, no allocation, no synchronisation: every call validates its arguments on the host and
// enqueues kernels on the caller's stream.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace tts {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int conv1d_dispatch(const TtsConvDesc& d, hipStream_t st);
int conv1d_tile_rows(int cout, int mode);
int conv1d_n_tile(int cout, int mode);
int conv1d_small_tile_rows(int cout, int mode, int packed_cols);
int layernorm(const float*, int, float*, int, const float*, const float*, int, int, float, hipStream_t);
int cond_layernorm(const float*, int, float*, int, const float*, const float*, int, const TtsTile*, int, int, hipStream_t);
long long cln_mlp_weight_floats(int, int);
int conv_post_snake_tile_rows();
int conv_post_snake(const float*, int, int, const float*, float, const float*, const float*, const float*, float*, const TtsTile*, int, int, int, hipStream_t);
int cln_mlp(const float*, int, int, int, const float*, int, float*, hipStream_t);
int l2_normalize(const float*, float*, int, int, hipStream_t);
int groupnorm(const float*, int, float*, int, const float*, const float*, int, int, float, int, const float*, int, const int*,
              const int*, int, int, float*, hipStream_t);
long groupnorm_workspace_floats(int, int, int);
int relpos_attention_mfma(const float*, int, const float*, int, const float*, const float*, float*, int, int, int, const TtsTile*, int,
                          int, int, hipStream_t);
int relpos_attention_f16(const float*, int, const float*, int, const float*, const float*, float*, int, int, int, const TtsTile*, int,
                          int, hipStream_t);
int dwconv_swish(const float*, int, float*, int, const float*, const float*, int, int, const TtsTile*, int, int, hipStream_t);
int duration_from_log(const float*, int*, int, hipStream_t);
int prosody_control(const float*, int, float*, float*, int*, const int*, const int*, int, float, float, float, float, hipStream_t);
int length_regulate(const float*, int, const float*, const float*, const float*, const float*, const float*, const float*,
                    const int*, const int*, const int*, const int*, int, int, int, int, float*, int, float*, int, float, hipStream_t);
int glow_invconv_actnorm(float*, int, int, int, const float*, const float*, const float*, hipStream_t);
int snake_aa(const float*, int, float*, int, const float*, const float*, const float*, int, const TtsTile*, int, int, int, hipStream_t);
int conv_post(const float*, int, int, const float*, float, int, float, float*, const TtsTile*, int, int, int, hipStream_t);
int gather_rows(const float*, int, const int*, float*, int, int, int, hipStream_t);
int resblock_step(const TtsResblockDesc& d, hipStream_t st);
int resblock_tile_rows(int c);
int snake_fir_table(const float* filt, void* table);

}  // namespace tts

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" {

const char* tts_last_error(void) { return tts::g_err; }
int tts_abi_version(void) { return TTS_ABI_VERSION; }

int tts_conv1d_tile_rows(int32_t cout, int32_t mode) { return tts::conv1d_tile_rows(cout, mode); }
int tts_conv1d_n_tile(int32_t cout, int32_t mode) { return tts::conv1d_n_tile(cout, mode); }
int tts_conv1d_small_tile_rows(int32_t cout, int32_t mode, int32_t packed_cols) { return tts::conv1d_small_tile_rows(cout, mode, packed_cols); }

int tts_conv1d(const TtsConvDesc* d, tts_stream_t stream) {
  if (!d) {
    tts::set_error("tts_conv1d: null descriptor");
    return TTS_E_ARG;
  }
  return tts::conv1d_dispatch(*d, ST(stream));
}

int tts_resblock_tile_rows(int32_t c) { return tts::resblock_tile_rows(c); }
int tts_snake_fir_table(const float* filt, void* table) { return tts::snake_fir_table(filt, table); }

int tts_resblock_step(const TtsResblockDesc* d, tts_stream_t stream) {
  if (!d) {
    tts::set_error("tts_resblock_step: null descriptor");
    return TTS_E_ARG;
  }
  return tts::resblock_step(*d, ST(stream));
}

int tts_layernorm(const float* x, int32_t ldx, float* y, int32_t ldy, const float* gamma, const float* beta, int32_t rows, int32_t c,
                  float eps, tts_stream_t stream) {
  return tts::layernorm(x, ldx, y, ldy, gamma, beta, rows, c, eps, ST(stream));
}

int64_t tts_cln_mlp_weight_floats(int32_t d_in, int32_t d_out) { return tts::cln_mlp_weight_floats(d_in, d_out); }
int tts_cln_mlp(const float* e, int32_t n_seq, int32_t d_in, int32_t d_out, const float* weights, int32_t n_mlp, float* out,
                tts_stream_t stream) {
  return tts::cln_mlp(e, n_seq, d_in, d_out, weights, n_mlp, out, ST(stream));
}
int tts_cond_layernorm(const float* x, int32_t ldx, float* y, int32_t ldy, const float* scale, const float* shift, int32_t c,
                       const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, tts_stream_t stream) {
  return tts::cond_layernorm(x, ldx, y, ldy, scale, shift, c, tiles, n_tiles, tile_rows, ST(stream));
}

int tts_l2_normalize(const float* x, float* y, int32_t rows, int32_t c, tts_stream_t stream) {
  return tts::l2_normalize(x, y, rows, c, ST(stream));
}

int tts_groupnorm(const float* x, int32_t ldx, float* y, int32_t ldy, const float* gamma, const float* beta, int32_t c, int32_t groups,
                  float eps, int32_t apply_tanh, const float* res, int32_t ld_res, const int32_t* seq_begin, const int32_t* seq_end,
                  int32_t n_seq, int32_t max_len, float* workspace, tts_stream_t stream) {
  return tts::groupnorm(x, ldx, y, ldy, gamma, beta, c, groups, eps, apply_tanh, res, ld_res, seq_begin, seq_end, n_seq, max_len, workspace,
                        ST(stream));
}

int64_t tts_groupnorm_workspace_floats(int32_t n_seq, int32_t max_len, int32_t groups) {
  return tts::groupnorm_workspace_floats(n_seq, max_len, groups);
}

int tts_relpos_attention(const float* qkv, int32_t ld_qkv, const float* ptab, int32_t pmax, const float* bias_u, const float* bias_v,
                         float* ctx, int32_t ld_ctx, int32_t heads, int32_t dk, const TtsTile* tiles, int32_t n_tiles,
                         int32_t tile_rows, int32_t flags, tts_stream_t stream) {
  return tts::relpos_attention_mfma(qkv, ld_qkv, ptab, pmax, bias_u, bias_v, ctx, ld_ctx, heads, dk, tiles, n_tiles, tile_rows, flags, ST(stream));
}

int tts_relpos_attention_f16(const float* qkv, int32_t ld_qkv, const float* ptab, int32_t pmax, const float* bias_u, const float* bias_v,
                             float* ctx, int32_t ld_ctx, int32_t heads, int32_t dk, const TtsTile* tiles, int32_t n_tiles,
                             int32_t tile_rows, tts_stream_t stream) {
  return tts::relpos_attention_f16(qkv, ld_qkv, ptab, pmax, bias_u, bias_v, ctx, ld_ctx, heads, dk, tiles, n_tiles, tile_rows, ST(stream));
}

int tts_dwconv_swish(const float* x, int32_t ldx, float* y, int32_t ldy, const float* w, const float* b, int32_t c, int32_t k,
                     const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, tts_stream_t stream) {
  return tts::dwconv_swish(x, ldx, y, ldy, w, b, c, k, tiles, n_tiles, tile_rows, ST(stream));
}

int tts_duration_from_log(const float* logd, int32_t* dur, int32_t n, tts_stream_t stream) {
  return tts::duration_from_log(logd, dur, n, ST(stream));
}

int tts_prosody_control(const float* text, int32_t ld_text, float* pitch, float* energy, int32_t* dur, const int32_t* seq_begin,
                        const int32_t* seq_end, int32_t n_seq, float duration_scale, float pitch_scale, float energy_scale,
                        float pause_scale, tts_stream_t stream) {
  return tts::prosody_control(text, ld_text, pitch, energy, dur, seq_begin, seq_end, n_seq, duration_scale, pitch_scale, energy_scale,
                              pause_scale, ST(stream));
}

int tts_length_regulate(const float* enc, int32_t ld_enc, const float* pitch, const float* energy, const float* wp, const float* bp,
                        const float* we, const float* be, const int32_t* dur, const int32_t* phone_begin, const int32_t* phone_end,
                        const int32_t* frame_begin, int32_t n_seq, int32_t max_frames, int32_t max_phones, int32_t c, float* up,
                        int32_t ld_up, float* dec_in, int32_t ld_dec, float dec_scale, tts_stream_t stream) {
  return tts::length_regulate(enc, ld_enc, pitch, energy, wp, bp, we, be, dur, phone_begin, phone_end, frame_begin, n_seq, max_frames,
                              max_phones, c, up, ld_up, dec_in, ld_dec, dec_scale, ST(stream));
}

int tts_glow_invconv_actnorm(float* x, int32_t ldx, int32_t rows, int32_t c, const float* winv, const float* an_bias,
                             const float* an_logs, tts_stream_t stream) {
  return tts::glow_invconv_actnorm(x, ldx, rows, c, winv, an_bias, an_logs, ST(stream));
}

int tts_snake_aa(const float* x, int32_t ldx, float* y, int32_t ldy, const float* alpha, const float* beta, const float* filt, int32_t c,
                 const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, int32_t io_flags, tts_stream_t stream) {
  return tts::snake_aa(x, ldx, y, ldy, alpha, beta, filt, c, tiles, n_tiles, tile_rows, io_flags, ST(stream));
}

int tts_conv_post_snake_tile_rows(void) { return tts::conv_post_snake_tile_rows(); }
int tts_conv_post_snake(const float* x, int32_t ldx, int32_t cin, const float* w, float bias, const float* alpha, const float* beta,
                        const float* filt, float* wav, const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, int32_t io_flags,
                        tts_stream_t stream) {
  return tts::conv_post_snake(x, ldx, cin, w, bias, alpha, beta, filt, wav, tiles, n_tiles, tile_rows, io_flags, ST(stream));
}
int tts_conv_post(const float* x, int32_t ldx, int32_t cin, const float* w, float bias, int32_t pre_act, float pre_slope, float* wav,
                  const TtsTile* tiles, int32_t n_tiles, int32_t tile_rows, int32_t io_flags, tts_stream_t stream) {
  return tts::conv_post(x, ldx, cin, w, bias, pre_act, pre_slope, wav, tiles, n_tiles, tile_rows, io_flags, ST(stream));
}

int tts_gather_rows(const float* src, int32_t ld_src, const int32_t* idx, float* dst, int32_t ld_dst, int32_t n, int32_t c,
                    tts_stream_t stream) {
  return tts::gather_rows(src, ld_src, idx, dst, ld_dst, n, c, ST(stream));
}


}  // extern "C"
