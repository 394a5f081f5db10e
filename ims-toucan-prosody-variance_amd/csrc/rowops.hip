// Row-wise normalisations on packed time-major activations: one wavefront (64 lanes) per row, channels
// strided over the lanes so every global access is a contiguous 256-B segment; reductions are wave
// shuffles (no LDS).  All statistics in fp32, two-pass (mean, then centred second moment) like ATen.
#include "common.h"

namespace tts {

constexpr int MAX_PER_LANE = 8;  // channels <= 512

// Layers/LayerNorm.py:24-36 (eps 1e-12, LayerNorm.py:17)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int c, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float v[MAX_PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_PER_LANE; ++i) {
    const int ch = lane + 64 * i;
    v[i] = ch < c ? xr[ch] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAX_PER_LANE; ++i) {
    const int ch = lane + 64 * i;
    const float dlt = ch < c ? v[i] - mean : 0.f;
    q += dlt * dlt;
  }
  const float var = wave_sum(q) / (float)c;
  const float rstd = 1.0f / sqrtf(var + eps);
  float* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < MAX_PER_LANE; ++i) {
    const int ch = lane + 64 * i;
    if (ch < c) yr[ch] = (v[i] - mean) * rstd * gamma[ch] + beta[ch];
  }
}

// Layers/ConditionalLayerNorm.py:52-67: y = scale[u] * (x - mean) / var + shift[u]   (variance, not std; no eps)
__global__ __launch_bounds__(256) void cln_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                  const float* __restrict__ scale, const float* __restrict__ shift, int c,
                                                  const TtsTile* __restrict__ tiles, int tile_rows) {
  const TtsTile t = tiles[blockIdx.x];
  const int lane = threadIdx.x & 63;
  const float* sc = scale + (size_t)t.seq_id * c;
  const float* sh = shift + (size_t)t.seq_id * c;
  for (int rr = threadIdx.x >> 6; rr < tile_rows; rr += 4) {
    const int row = t.row0 + rr;
    if (row >= t.seq_end) break;
    const float* xr = x + (size_t)row * ldx;
    float v[MAX_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
      const int ch = lane + 64 * i;
      v[i] = ch < c ? xr[ch] : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
      const int ch = lane + 64 * i;
      const float dlt = ch < c ? v[i] - mean : 0.f;
      q += dlt * dlt;
    }
    const float var = wave_sum(q) / (float)c;
    float* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAX_PER_LANE; ++i) {
      const int ch = lane + 64 * i;
      if (ch < c) yr[ch] = sc[ch] * ((v[i] - mean) / var) + sh[ch];
    }
  }
}

// torch.nn.functional.normalize(x, dim=1): x / max(||x||_2, 1e-12)   (InferenceToucanTTS.py:202, Conformer.py:132)
__global__ __launch_bounds__(64) void l2norm_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int c) {
  const int row = blockIdx.x;
  const int lane = threadIdx.x;
  float q = 0.f;
  for (int ch = lane; ch < c; ch += 64) {
    const float v = x[(size_t)row * c + ch];
    q += v * v;
  }
  const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
  for (int ch = lane; ch < c; ch += 64) y[(size_t)row * c + ch] = x[(size_t)row * c + ch] / nrm;
}

// GroupNorm over (c/groups channels) x (all frames of ONE utterance) + affine + optional tanh + optional residual.
// Layers/PostNet.py:44-56 (GroupNorm(32,256) x4 with Tanh, GroupNorm(20,80) last), eps 1e-5.
// One workgroup per utterance; thread = channel (c <= 256).  Three sweeps over the utterance's frames
// (sum -> centred sum of squares -> apply); every sweep reads whole rows, i.e. contiguous c*4-byte segments.
// The statistics are a time-axis reduction, so they MUST be per utterance (padding may never leak in).
__global__ __launch_bounds__(256) void groupnorm_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, int c,
                                                        int groups, float eps, int apply_tanh, const float* __restrict__ res,
                                                        int ld_res, const int* __restrict__ seq_begin, const int* __restrict__ seq_end) {
  __shared__ float red[256];
  const int u = blockIdx.x;
  const int r0 = seq_begin[u], r1 = seq_end[u];
  const int ch = threadIdx.x;
  const int cpg = c / groups;
  const bool on = ch < c;
  const float cnt = (float)cpg * (float)(r1 - r0);
  // pass 1: mean
  float s = 0.f;
  if (on)
    for (int r = r0; r < r1; ++r) s += x[(size_t)r * ldx + ch];
  red[ch] = s;
  __syncthreads();
  float gsum = 0.f;
  if (on) {
    const int g0 = (ch / cpg) * cpg;
    for (int k = 0; k < cpg; ++k) gsum += red[g0 + k];
  }
  const float mean = gsum / cnt;
  __syncthreads();
  // pass 2: centred second moment
  float q = 0.f;
  if (on)
    for (int r = r0; r < r1; ++r) {
      const float dlt = x[(size_t)r * ldx + ch] - mean;
      q += dlt * dlt;
    }
  red[ch] = q;
  __syncthreads();
  float gq = 0.f;
  if (on) {
    const int g0 = (ch / cpg) * cpg;
    for (int k = 0; k < cpg; ++k) gq += red[g0 + k];
  }
  if (!on) return;
  const float rstd = 1.0f / sqrtf(gq / cnt + eps);
  const float ga = gamma[ch] * rstd, be = beta[ch] - mean * gamma[ch] * rstd;
  for (int r = r0; r < r1; ++r) {
    float v = x[(size_t)r * ldx + ch] * ga + be;
    if (apply_tanh) v = tanhf(v);
    if (res) v += res[(size_t)r * ld_res + ch];
    y[(size_t)r * ldy + ch] = v;
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, int ldx, float a, const float* __restrict__ z,
                                                    int ldz, float b, float* __restrict__ y, int ldy, int rows, int c) {
  const size_t n = (size_t)rows * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / c), ch = (int)(i % c);
    float v = a * x[(size_t)r * ldx + ch];
    if (z) v += b * z[(size_t)r * ldz + ch];
    y[(size_t)r * ldy + ch] = v;
  }
}

__global__ __launch_bounds__(64) void gather_rows_kernel(const float* __restrict__ src, int ld_src, const int* __restrict__ idx,
                                                         float* __restrict__ dst, int ld_dst, int c) {
  const int i = blockIdx.x;
  const float* s = src + (size_t)idx[i] * ld_src;
  for (int ch = threadIdx.x; ch < c; ch += 64) dst[(size_t)i * ld_dst + ch] = s[ch];
}

int gather_rows(const float* src, int ld_src, const int* idx, float* dst, int ld_dst, int n, int c, hipStream_t st) {
  if (n == 0) return TTS_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(64), 0, st, src, ld_src, idx, dst, ld_dst, c);
  return launch_status("gather_rows");
}

int layernorm(const float* x, int ldx, float* y, int ldy, const float* g, const float* b, int rows, int c, float eps, hipStream_t st) {
  TTS_CHECK_ARG(c > 0 && c <= 64 * MAX_PER_LANE, "layernorm: c=%d unsupported", c);
  if (rows == 0) return TTS_OK;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, ldx, y, ldy, g, b, rows, c, eps);
  return launch_status("layernorm");
}

int cond_layernorm(const float* x, int ldx, float* y, int ldy, const float* sc, const float* sh, int c, const TtsTile* tiles,
                   int n_tiles, int tile_rows, hipStream_t st) {
  TTS_CHECK_ARG(c > 0 && c <= 64 * MAX_PER_LANE, "cond_layernorm: c=%d unsupported", c);
  if (n_tiles == 0) return TTS_OK;
  hipLaunchKernelGGL(cln_kernel, dim3(n_tiles), dim3(256), 0, st, x, ldx, y, ldy, sc, sh, c, tiles, tile_rows);
  return launch_status("cond_layernorm");
}

int l2_normalize(const float* x, float* y, int rows, int c, hipStream_t st) {
  if (rows == 0) return TTS_OK;
  hipLaunchKernelGGL(l2norm_kernel, dim3(rows), dim3(64), 0, st, x, y, rows, c);
  return launch_status("l2_normalize");
}

int groupnorm(const float* x, int ldx, float* y, int ldy, const float* g, const float* b, int c, int groups, float eps,
              int apply_tanh, const float* res, int ld_res, const int* sb, const int* se, int n_seq, hipStream_t st) {
  TTS_CHECK_ARG(c > 0 && c <= 256 && groups > 0 && c % groups == 0, "groupnorm: c=%d groups=%d unsupported", c, groups);
  if (n_seq == 0) return TTS_OK;
  hipLaunchKernelGGL(groupnorm_kernel, dim3(n_seq), dim3(256), 0, st, x, ldx, y, ldy, g, b, c, groups, eps, apply_tanh, res, ld_res, sb, se);
  return launch_status("groupnorm");
}

int axpby(const float* x, int ldx, float a, const float* z, int ldz, float b, float* y, int ldy, int rows, int c, hipStream_t st) {
  if (rows == 0 || c == 0) return TTS_OK;
  size_t n = (size_t)rows * c;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, st, x, ldx, a, z, ldz, b, y, ldy, rows, c);
  return launch_status("axpby");
}

}  // namespace tts
