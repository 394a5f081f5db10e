// Bandwidth-bound per-utterance kernels: depthwise conv, duration head, prosody control, length regulator,
// Glow invconv/actnorm, anti-aliased snake, vocoder output conv.  Common rules: lane = channel (rows are
// contiguous in memory, so a wavefront touches whole 256-B segments), sliding windows live in registers
// (static unrolling), halos are resolved per utterance through the tile table.
#include "common.h"
#include "snake.h"

namespace tts {

// ------------------------------------------------------------------------------------------------
// Depthwise conv (k taps, zero padded) + BatchNorm(eval, folded on the host) + Swish.
// Layers/Convolution.py:50-51 (+ Swish.py:18).  Each thread: one channel x ROWS consecutive frames, the
// ROWS+K-1 input window is loaded once into registers (loads/output = (ROWS+K-1)/ROWS instead of K).
// ------------------------------------------------------------------------------------------------
template <int K, int ROWS>
__global__ __launch_bounds__(256) void dwconv_swish_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                           const float* __restrict__ w, const float* __restrict__ b, int c,
                                                           const TtsTile* __restrict__ tiles, int tile_rows) {
  const TtsTile t = tiles[blockIdx.x];
  const int ch = blockIdx.y * 64 + (threadIdx.x & 63);
  if (ch >= c) return;
  float wk[K];
#pragma unroll
  for (int j = 0; j < K; ++j) wk[j] = w[j * c + ch];
  const float bias = b[ch];
  constexpr int H = (K - 1) / 2;
  for (int g = threadIdx.x >> 6; g * ROWS < tile_rows; g += 4) {
    const int r0 = t.row0 + g * ROWS;
    if (r0 >= t.seq_end) break;
    float win[ROWS + K - 1];
#pragma unroll
    for (int i = 0; i < ROWS + K - 1; ++i) {
      const int r = r0 - H + i;
      win[i] = (r >= t.seq_begin && r < t.seq_end) ? x[(size_t)r * ldx + ch] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      if (r0 + i >= t.seq_end) break;
      float a = bias;
#pragma unroll
      for (int j = 0; j < K; ++j) a = fmaf(wk[j], win[i + j], a);
      y[(size_t)(r0 + i) * ldy + ch] = a * (1.0f / (1.0f + expf(-a)));
    }
  }
}

int dwconv_swish(const float* x, int ldx, float* y, int ldy, const float* w, const float* b, int c, int k, const TtsTile* tiles,
                 int n_tiles, int tile_rows, hipStream_t st) {
  TTS_CHECK_ARG(k == 7 || k == 31, "dwconv_swish: kernel size %d unsupported (7, 31)", k);
  TTS_CHECK_ARG(tile_rows % 8 == 0, "dwconv_swish: tile_rows must be a multiple of 8");
  if (n_tiles == 0) return TTS_OK;
  dim3 grid(n_tiles, (c + 63) / 64), block(256);
  if (k == 7)
    hipLaunchKernelGGL((dwconv_swish_kernel<7, 8>), grid, block, 0, st, x, ldx, y, ldy, w, b, c, tiles, tile_rows);
  else
    hipLaunchKernelGGL((dwconv_swish_kernel<31, 8>), grid, block, 0, st, x, ldx, y, ldy, w, b, c, tiles, tile_rows);
  return launch_status("dwconv_swish");
}

// ------------------------------------------------------------------------------------------------
// Layers/DurationPredictor.py:79: clamp(round(exp(x) - offset), min=0).long(), offset = 1, round half to even
// ------------------------------------------------------------------------------------------------
__global__ void duration_kernel(const float* __restrict__ logd, int* __restrict__ dur, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = rintf(expf(logd[i]) - 1.0f);
  v = fminf(fmaxf(v, 0.0f), 1.0e6f);
  dur[i] = (int)v;
}

int duration_from_log(const float* logd, int* dur, int n, hipStream_t st) {
  if (n == 0) return TTS_OK;
  hipLaunchKernelGGL(duration_kernel, dim3((n + 255) / 256), dim3(256), 0, st, logd, dur, n);
  return launch_status("duration_from_log");
}

// ------------------------------------------------------------------------------------------------
// InferenceToucanTTS.py:214-227 + _scale_variance :333-343.  One workgroup per utterance.
// Feature columns (Preprocessing/articulatory_features.py:817-901): phoneme 15, silence 16, word-boundary 21, voiced 61.
// ------------------------------------------------------------------------------------------------
__device__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ void scale_variance(float* seq, int r0, int r1, float scale, float* red) {
  // mean over the NON-ZERO entries; every entry (zeros included) is shifted, scaled, shifted back; negatives -> 0
  float s = 0.f, cnt = 0.f;
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    const float v = seq[r];
    if (v != 0.0f) { s += v; cnt += 1.f; }
  }
  s = block_sum(s, red);
  cnt = block_sum(cnt, red);
  const float avg = s / cnt;  // empty selection -> NaN, as torch's mean of an empty tensor
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    float v = seq[r] - avg;
    v = v * scale;
    v = v + avg;
    seq[r] = v < 0.0f ? 0.0f : v;
  }
}

__global__ __launch_bounds__(256) void prosody_control_kernel(const float* __restrict__ text, int ld_text, float* pitch, float* energy,
                                                              int* dur, const int* __restrict__ seq_begin, const int* __restrict__ seq_end,
                                                              float duration_scale, float pitch_scale, float energy_scale, float pause_scale) {
  __shared__ float red[4];
  const int u = blockIdx.x;
  const int r0 = seq_begin[u], r1 = seq_end[u];
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    const float* f = text + (size_t)r * ld_text;
    if (f[61] == 0.0f) pitch[r] = 0.0f;
    if (f[15] == 0.0f) energy[r] = 0.0f;
    int d = dur[r];
    if (f[21] == 1.0f) d = 0;
    if (f[16] == 1.0f && pause_scale != 1.0f) d = (int)rintf((float)d * pause_scale);
    if (duration_scale != 1.0f) d = (int)rintf((float)d * duration_scale);
    dur[r] = d;
  }
  __syncthreads();
  if (pitch_scale != 1.0f) scale_variance(pitch, r0, r1, pitch_scale, red);
  if (energy_scale != 1.0f) scale_variance(energy, r0, r1, energy_scale, red);
}

int prosody_control(const float* text, int ld_text, float* pitch, float* energy, int* dur, const int* sb, const int* se, int n_seq,
                    float ds, float ps, float es, float pause, hipStream_t st) {
  if (n_seq == 0) return TTS_OK;
  hipLaunchKernelGGL(prosody_control_kernel, dim3(n_seq), dim3(256), 0, st, text, ld_text, pitch, energy, dur, sb, se, ds, ps, es, pause);
  return launch_status("prosody_control");
}

// ------------------------------------------------------------------------------------------------
// LengthRegulator (Layers/LengthRegulator.py:37-61) fused with the pitch/energy embedding add
// (InferenceToucanTTS.py:230-232).  grid = (frame tiles of 64, utterances).  Every workgroup rebuilds the
// inclusive scan of its utterance's durations in LDS (L is a few hundred at most), finds the source phoneme
// of each of its 64 frames by binary search, then copies rows with lane = channel.
// ------------------------------------------------------------------------------------------------
constexpr int LR_MAX_PHONES = 4096;

__global__ __launch_bounds__(256) void length_regulate_kernel(const float* __restrict__ enc, int ld_enc, const float* __restrict__ pitch,
                                                              const float* __restrict__ energy, const float* __restrict__ wp,
                                                              const float* __restrict__ bp, const float* __restrict__ we,
                                                              const float* __restrict__ be, const int* __restrict__ dur,
                                                              const int* __restrict__ phone_begin, const int* __restrict__ phone_end,
                                                              const int* __restrict__ frame_begin, int c, float* __restrict__ up, int ld_up,
                                                              float* __restrict__ dec_in, int ld_dec, float dec_scale) {
  __shared__ int csum[LR_MAX_PHONES];
  __shared__ int wsum[4];
  __shared__ int src[64];
  const int u = blockIdx.y;
  const int p0 = phone_begin[u], L = phone_end[u] - p0;
  // inclusive scan, 256 phonemes per sweep with a running carry
  int carry = 0;
  for (int base = 0; base < L; base += 256) {
    const int i = base + threadIdx.x;
    int v = i < L ? dur[p0 + i] : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(v, o, 64);
      if (lane >= o) v += n;
    }
    if (lane == 63) wsum[wv] = v;
    __syncthreads();
    int add = carry;
    for (int k = 0; k < wv; ++k) add += wsum[k];
    if (i < L) csum[i] = v + add;
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  const bool all_zero = (carry == 0);  // LengthRegulator.py:52-53: an all-zero utterance becomes all ones
  const int T = all_zero ? L : carry;
  const int f0 = blockIdx.x * 64;
  if (f0 >= T) return;
  if (threadIdx.x < 64) {
    const int f = f0 + threadIdx.x;
    int s = 0;
    if (f < T) {
      if (all_zero) {
        s = f;
      } else {  // first phoneme whose inclusive sum exceeds f
        int lo = 0, hi = L - 1;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (csum[mid] > f) hi = mid; else lo = mid + 1;
        }
        s = lo;
      }
    }
    src[threadIdx.x] = s;
  }
  __syncthreads();
  const int fb = frame_begin[u];
  for (int e = threadIdx.x; e < 64 * c; e += 256) {
    const int fi = e / c, ch = e % c;
    const int f = f0 + fi;
    if (f >= T) break;
    const int p = p0 + src[fi];
    const float v = enc[(size_t)p * ld_enc + ch] + (pitch[p] * wp[ch] + bp[ch]) + (energy[p] * we[ch] + be[ch]);
    up[(size_t)(fb + f) * ld_up + ch] = v;
    if (dec_in) dec_in[(size_t)(fb + f) * ld_dec + ch] = v * dec_scale;
  }
}

int length_regulate(const float* enc, int ld_enc, const float* pitch, const float* energy, const float* wp, const float* bp,
                    const float* we, const float* be, const int* dur, const int* phone_begin, const int* phone_end,
                    const int* frame_begin, int n_seq, int max_frames, int max_phones, int c, float* up, int ld_up, float* dec_in,
                    int ld_dec, float dec_scale, hipStream_t st) {
  TTS_CHECK_ARG(max_phones <= LR_MAX_PHONES, "length_regulate: %d phonemes > %d", max_phones, LR_MAX_PHONES);
  if (n_seq == 0 || max_frames == 0) return TTS_OK;
  dim3 grid((max_frames + 63) / 64, n_seq);
  hipLaunchKernelGGL(length_regulate_kernel, grid, dim3(256), 0, st, enc, ld_enc, pitch, energy, wp, bp, we, be, dur, phone_begin,
                     phone_end, frame_begin, c, up, ld_up, dec_in, ld_dec, dec_scale);
  return launch_status("length_regulate");
}

// ------------------------------------------------------------------------------------------------
// Glow reverse: InvConvNear^-1 then ActNorm^-1, in place on [rows, c] (c = 160).
// Channel ch = a*(c/2) + 2*g + r belongs to mixing slot n = 2*a + r of group g (Glow.py:102-103);
// z[n_out, g] = sum_n Winv[n_out][n] x[n, g] (:124), regrouped back (:126-127), then (z - bias)*exp(-logs) (:30-31).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void glow_mix_kernel(float* __restrict__ x, int ldx, int rows, int c, const float* __restrict__ winv,
                                                       const float* __restrict__ an_bias, const float* __restrict__ an_logs) {
  const int G = c / 4;
  const size_t total = (size_t)rows * G;
  float w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = winv[i];
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int r = (int)(e / G), g = (int)(e % G);
    float* xr = x + (size_t)r * ldx;
    const int i0 = 2 * g, i1 = 2 * g + 1, i2 = c / 2 + 2 * g, i3 = c / 2 + 2 * g + 1;
    const float v0 = xr[i0], v1 = xr[i1], v2 = xr[i2], v3 = xr[i3];
    const int idx[4] = {i0, i1, i2, i3};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float z = w[o * 4 + 0] * v0;
      z = fmaf(w[o * 4 + 1], v1, z);
      z = fmaf(w[o * 4 + 2], v2, z);
      z = fmaf(w[o * 4 + 3], v3, z);
      xr[idx[o]] = (z - an_bias[idx[o]]) * expf(-an_logs[idx[o]]);
    }
  }
}

int glow_invconv_actnorm(float* x, int ldx, int rows, int c, const float* winv, const float* an_bias, const float* an_logs, hipStream_t st) {
  TTS_CHECK_ARG(c % 4 == 0, "glow_invconv_actnorm: c %% 4 != 0");
  if (rows == 0) return TTS_OK;
  size_t total = (size_t)rows * (c / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(glow_mix_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, rows, c, winv, an_bias, an_logs);
  return launch_status("glow_invconv_actnorm");
}

// ------------------------------------------------------------------------------------------------
// Anti-aliased SnakeBeta (BigVGAN/Snake.py:56-69 inside alias_free_torch's Activation1d - third party,
// PARITY UNPINNED, restated from its published algorithm):
//   u[2q]   = 2 * sum_{d=-3..2} x[q+d] f[5-2d]      u[2q+1] = 2 * sum_{d=-2..3} x[q+d] f[6-2d]   (x replicate padded)
//   s[n]    = u[n] + sin^2(u[n] * e^alpha) / (e^beta + 1e-9)
//   y[t]    = sum_{k=0..11} s[clamp(2t + k - 5, 0, 2T-1)] f[k]
// Each thread: one channel x ROWS consecutive frames; the 2x-rate signal exists only in registers
// (2*ROWS+10 values), so the up-sampled tensor never touches LDS or HBM.
// ------------------------------------------------------------------------------------------------
template <int NCH, bool XB, bool YB>
__global__ __launch_bounds__(256) void snake_aa_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const float* __restrict__ filt, int c, const TtsTile* __restrict__ tiles,
                                                       int tile_rows, int f16) {
  const TtsTile t = tiles[blockIdx.x];
  // work item = (8*NCH frames, channel), streamed (snake.h) so that only the first 8 frames pay the filter halo; consecutive
  // threads take consecutive channels, so a wavefront touches one contiguous run per row.  The 2x-rate signal lives in registers.
  constexpr int GR = 8 * NCH;
  const int item = blockIdx.y * 256 + threadIdx.x;
  const int ch = item % c;
  const int g = item / c;
  if (g * GR >= tile_rows) return;
  const int r0 = t.row0 + g * GR;
  if (r0 >= t.seq_end) return;
  const int r_end = min(t.seq_end, t.row0 + tile_rows);  // rows past the tile belong to the next tile's workgroup
  float f[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) f[k] = filt[k];
  const float ea = expf(alpha[ch]);
  const float inv_b = 1.0f / (expf(beta[ch]) + 1e-9f);
  const unsigned short* __restrict__ xh = reinterpret_cast<const unsigned short*>(x);
  unsigned short* __restrict__ yh = reinterpret_cast<unsigned short*>(y);
  snake_stream<NCH>([&](int q) { return XB ? load16(xh[(size_t)(t.seq_begin + q) * ldx + ch], f16) : x[(size_t)(t.seq_begin + q) * ldx + ch]; },
                    [&](int i, float v) {
                      if (r0 + i < r_end) {
                        if (YB) yh[(size_t)(r0 + i) * ldy + ch] = store16(v, f16);
                        else y[(size_t)(r0 + i) * ldy + ch] = v;
                      }
                    }, t.seq_end - t.seq_begin, r0 - t.seq_begin, f, ea, inv_b);
}

int snake_aa(const float* x, int ldx, float* y, int ldy, const float* alpha, const float* beta, const float* filt, int c,
             const TtsTile* tiles, int n_tiles, int tile_rows, int io_flags, hipStream_t st) {
  TTS_CHECK_ARG(tile_rows % 8 == 0, "snake_aa: tile_rows must be a multiple of 8");
  if (n_tiles == 0) return TTS_OK;
  constexpr int NCH = 4;
  const int items = ((tile_rows + 8 * NCH - 1) / (8 * NCH)) * c;
  dim3 grid(n_tiles, (items + 255) / 256), block(256);
  const bool xb = io_flags & TTS_IO_X_BF16, yb = io_flags & TTS_IO_Y_BF16;
#define TTS_SNAKE_LAUNCH(XB, YB) \
  hipLaunchKernelGGL((snake_aa_kernel<NCH, XB, YB>), grid, block, 0, st, x, ldx, y, ldy, alpha, beta, filt, c, tiles, tile_rows, (io_flags & TTS_IO_F16) ? 1 : 0)
  if (xb && yb) TTS_SNAKE_LAUNCH(true, true);
  else if (xb) TTS_SNAKE_LAUNCH(true, false);
  else if (yb) TTS_SNAKE_LAUNCH(false, true);
  else TTS_SNAKE_LAUNCH(false, false);
#undef TTS_SNAKE_LAUNCH
  return launch_status("snake_aa");
}

// ------------------------------------------------------------------------------------------------
// Vocoder output conv (cin -> 1, 7 taps) + tanh.  InferenceAvocodo.py:52-59 / InferenceBigVGAN.py:92-95.
// 256 output samples per workgroup; the (256+6) x cin window is staged once in LDS (odd pitch), each
// thread then reduces its own 7 x cin patch.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, int ldx, int cin, const float* __restrict__ w,
                                                        float bias, int pre_act, float pre_slope, float* __restrict__ wav,
                                                        const TtsTile* __restrict__ tiles, int io_flags) {
  extern __shared__ float lds[];
  const TtsTile t = tiles[blockIdx.x];
  const int pitch = cin + 1;
  float* xs = lds;                 // [262][cin+1]
  float* wsm = lds + 262 * pitch;  // [7][cin]
  for (int e = threadIdx.x; e < 7 * cin; e += 256) wsm[e] = w[e];
  for (int e = threadIdx.x; e < 262 * cin; e += 256) {
    const int wr = e / cin, ch = e % cin;
    const int r = t.row0 - 3 + wr;
    float v = 0.f;
    if (r >= t.seq_begin && r < t.seq_end)
      v = (io_flags & TTS_IO_X_BF16) ? load16(reinterpret_cast<const unsigned short*>(x)[(size_t)r * ldx + ch], io_flags & TTS_IO_F16) : x[(size_t)r * ldx + ch];
    if (pre_act == TTS_PRE_LRELU) v = v > 0.f ? v : v * pre_slope;
    xs[wr * pitch + ch] = v;
  }
  __syncthreads();
  const int row = t.row0 + threadIdx.x;
  if (row >= t.seq_end) return;
  float a = bias;
  for (int j = 0; j < 7; ++j) {
    const float* xr = xs + (threadIdx.x + j) * pitch;
    const float* wr = wsm + j * cin;
    for (int ch = 0; ch < cin; ++ch) a = fmaf(xr[ch], wr[ch], a);
  }
  wav[row] = tanhf(a);
}

int conv_post(const float* x, int ldx, int cin, const float* w, float bias, int pre_act, float pre_slope, float* wav,
              const TtsTile* tiles, int n_tiles, int tile_rows, int io_flags, hipStream_t st) {
  TTS_CHECK_ARG(tile_rows == 256, "conv_post: tile table must use 256 rows, got %d", tile_rows);
  TTS_CHECK_ARG(cin > 0 && cin <= 64, "conv_post: cin %d unsupported", cin);
  if (n_tiles == 0) return TTS_OK;
  size_t lds = (size_t)(262 * (cin + 1) + 7 * cin) * sizeof(float);
  hipLaunchKernelGGL(conv_post_kernel, dim3(n_tiles), dim3(256), lds, st, x, ldx, cin, w, bias, pre_act, pre_slope, wav, tiles, io_flags);
  return launch_status("conv_post");
}

// ------------------------------------------------------------------------------------------------
// activation_post (anti-aliased snake) + output conv (32 -> 1, 7 taps) + tanh in one launch (InferenceBigVGAN.py:90-95).
// 250 output samples per workgroup: the 256-row window (3 rows of conv halo each side) is 8 streamed groups of 32 frames x
// 32 channels = one snake stream per thread, written to LDS as fp32 (zero outside the utterance = the conv's padding);
// then thread r < 250 reduces its 7 x 32 patch.  Saves the 2 x 335 MB round trip of the activated tensor at batch 32.
// ------------------------------------------------------------------------------------------------
constexpr int CPS_TILE = 250, CPS_WIN = 256, CPS_C = 32;

template <bool XB>
__global__ __launch_bounds__(256) void conv_post_snake_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, float bias,
                                                             const float* __restrict__ alpha, const float* __restrict__ beta,
                                                             const float* __restrict__ filt, float* __restrict__ wav,
                                                             const TtsTile* __restrict__ tiles, int f16) {
  __shared__ float xs[CPS_WIN * (CPS_C + 1)];
  __shared__ float wsm[7 * CPS_C];
  const TtsTile t = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  if (tid < 7 * CPS_C) wsm[tid] = w[tid];
  const int T = t.seq_end - t.seq_begin;
  const int ch = tid & 31, g = tid >> 5;
  const int t0 = t.row0 - 3 + g * 32 - t.seq_begin;  // local frame of this thread's first window row
  float f[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) f[k] = filt[k];
  float* col = xs + (g * 32) * (CPS_C + 1) + ch;
  if (t0 + 31 >= 0 && t0 < T) {
    const float ea = expf(alpha[ch]), ib = 1.0f / (expf(beta[ch]) + 1e-9f);
    const unsigned short* __restrict__ xh = reinterpret_cast<const unsigned short*>(x);
    snake_stream<4>([&](int q) { return XB ? load16(xh[(size_t)(t.seq_begin + q) * ldx + ch], f16) : x[(size_t)(t.seq_begin + q) * ldx + ch]; },
                    [&](int i, float v) { col[i * (CPS_C + 1)] = (t0 + i >= 0 && t0 + i < T) ? v : 0.0f; }, T, t0, f, ea, ib);
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) col[i * (CPS_C + 1)] = 0.0f;
  }
  __syncthreads();
  const int row = t.row0 + tid;
  if (tid >= CPS_TILE || row >= t.seq_end) return;
  float a = bias;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const float* xr = xs + (tid + j) * (CPS_C + 1);
    const float* wr = wsm + j * CPS_C;
#pragma unroll
    for (int c = 0; c < CPS_C; ++c) a = fmaf(xr[c], wr[c], a);
  }
  wav[row] = tanhf(a);
}

int conv_post_snake_tile_rows() { return CPS_TILE; }

int conv_post_snake(const float* x, int ldx, int cin, const float* w, float bias, const float* alpha, const float* beta, const float* filt,
                    float* wav, const TtsTile* tiles, int n_tiles, int tile_rows, int io_flags, hipStream_t st) {
  TTS_CHECK_ARG(x && w && alpha && beta && filt && wav && tiles, "conv_post_snake: null pointer");
  TTS_CHECK_ARG(cin == CPS_C, "conv_post_snake: cin must be %d, got %d", CPS_C, cin);
  TTS_CHECK_ARG(tile_rows == CPS_TILE, "conv_post_snake: tile table must use %d rows, got %d", CPS_TILE, tile_rows);
  if (n_tiles == 0) return TTS_OK;
  if (io_flags & TTS_IO_X_BF16)
    hipLaunchKernelGGL(conv_post_snake_kernel<true>, dim3(n_tiles), dim3(256), 0, st, x, ldx, w, bias, alpha, beta, filt, wav, tiles, (io_flags & TTS_IO_F16) ? 1 : 0);
  else
    hipLaunchKernelGGL(conv_post_snake_kernel<false>, dim3(n_tiles), dim3(256), 0, st, x, ldx, w, bias, alpha, beta, filt, wav, tiles, 0);
  return launch_status("conv_post_snake");
}

}  // namespace tts
