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
constexpr int CLN_ROWS = 8;  // rows per workgroup (two per wavefront)
__global__ __launch_bounds__(256) void cln_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                  const float* __restrict__ scale, const float* __restrict__ shift, int c,
                                                  const TtsTile* __restrict__ tiles, int tile_rows) {
  const TtsTile t = tiles[blockIdx.x];
  const int lane = threadIdx.x & 63;
  const float* sc = scale + (size_t)t.seq_id * c;
  const float* sh = shift + (size_t)t.seq_id * c;
  // blockIdx.y: a run of CLN_ROWS rows of the tile (one workgroup per 128-row tile left a wavefront 32 dependent row round trips:
  // 25 us for the 128 rows of a batch-1 pass, 48 us at batch 32)
  const int r_end = min(tile_rows, ((int)blockIdx.y + 1) * CLN_ROWS);
  for (int rr = blockIdx.y * CLN_ROWS + (threadIdx.x >> 6); rr < r_end; rr += 4) {
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
// The statistics are a time-axis reduction, so they MUST be per utterance (padding may never leak in).
// Two launches over a (16-frame chunk, utterance) grid, thread = channel (c <= 256, whole rows = contiguous segments; 64-frame
// chunks until round 3: 10 workgroups for the 640 frames of a batch-1 pass, 16 + 21 us per norm):
//   1. partial (sum, sum of squares) per (utterance, chunk, group) into a workspace - no atomics, so the result is
//      bit-reproducible from run to run and across ranks;
//   2. every workgroup adds the partials of its utterance in chunk order (fp64), then normalises its own 16 frames.
constexpr int GN_CHUNK = 16;

__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const float* __restrict__ x, int ldx, int c, int groups,
                                                                const int* __restrict__ seq_begin, const int* __restrict__ seq_end,
                                                                float* __restrict__ ws, int n_chunks) {
  __shared__ float rs[256], rq[256];
  const int u = blockIdx.y, chunk = blockIdx.x;
  const int r0 = seq_begin[u] + chunk * GN_CHUNK;
  const int r1 = min(seq_end[u], r0 + GN_CHUNK);
  const int ch = threadIdx.x, cpg = c / groups;
  float s = 0.f, q = 0.f;
  if (ch < c)
    for (int r = r0; r < r1; ++r) {
      const float v = x[(size_t)r * ldx + ch];
      s += v;
      q = fmaf(v, v, q);
    }
  rs[ch] = s;
  rq[ch] = q;
  __syncthreads();
  if (ch < groups) {
    float gs = 0.f, gq = 0.f;
    for (int k = 0; k < cpg; ++k) {
      gs += rs[ch * cpg + k];
      gq += rq[ch * cpg + k];
    }
    float* w = ws + (((size_t)u * n_chunks + chunk) * groups + ch) * 2;
    w[0] = gs;
    w[1] = gq;
  }
}

__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int c,
                                                              int groups, float eps, int apply_tanh, const float* __restrict__ res,
                                                              int ld_res, const int* __restrict__ seq_begin, const int* __restrict__ seq_end,
                                                              const float* __restrict__ ws, int n_chunks) {
  const int u = blockIdx.y, chunk = blockIdx.x;
  const int sb = seq_begin[u], se = seq_end[u];
  const int r0 = sb + chunk * GN_CHUNK;
  if (r0 >= se) return;
  const int r1 = min(se, r0 + GN_CHUNK);
  const int ch = threadIdx.x;
  if (ch >= c) return;
  const int cpg = c / groups, g = ch / cpg;
  const int used = (se - sb + GN_CHUNK - 1) / GN_CHUNK;  // chunks that hold frames of this utterance
  double s = 0.0, q = 0.0;
  for (int k = 0; k < used; ++k) {
    const float* w = ws + (((size_t)u * n_chunks + k) * groups + g) * 2;
    s += (double)w[0];
    q += (double)w[1];
  }
  const double cnt = (double)cpg * (double)(se - sb);
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float ga = gamma[ch] * rstd, be = beta[ch] - (float)mean * ga;
  for (int r = r0; r < r1; ++r) {
    float v = fmaf(x[(size_t)r * ldx + ch], ga, be);
    if (apply_tanh) v = tanhf(v);
    if (res) v += res[(size_t)r * ld_res + ch];
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

// ------------------------------------------------------------------------------------------------
// All scale / shift MLPs of the conditional layer norms in one launch: workgroup = (utterance, MLP), thread = output
// unit; the three small matrix-vector products run back to back through LDS.  (As 72 separate GEMM launches on a
// [B, 64] operand these cost ~0.8 ms of a 13 ms batch-1 pass.)  ConditionalLayerNorm.py:26-35, 54-55.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cln_mlp_kernel(const float* __restrict__ e, int d_in, int d_out, const float* __restrict__ weights,
                                                      long long per_mlp, float* __restrict__ out, int n_seq) {
  __shared__ float v0[256], v1[256];
  const int u = blockIdx.x, m = blockIdx.y, j = threadIdx.x;
  const float* w0 = weights + (size_t)m * per_mlp;
  const float* b0 = w0 + (size_t)d_in * d_in;
  const float* w1 = b0 + d_in;
  const float* b1 = w1 + (size_t)d_in * d_out;
  const float* w2 = b1 + d_out;
  const float* b2 = w2 + (size_t)d_out * d_out;
  if (j < d_in) v0[j] = e[(size_t)u * d_in + j];
  __syncthreads();
  float a = 0.f;
  if (j < d_in) {
    a = b0[j];
    for (int i = 0; i < d_in; ++i) a = fmaf(v0[i], w0[(size_t)i * d_in + j], a);
    a = tanhf(a);
  }
  __syncthreads();
  if (j < d_in) v1[j] = a;
  __syncthreads();
  if (j < d_out) {
    a = b1[j];
    for (int i = 0; i < d_in; ++i) a = fmaf(v1[i], w1[(size_t)i * d_out + j], a);
    a = tanhf(a);
  }
  __syncthreads();
  if (j < d_out) v0[j] = a;
  __syncthreads();
  if (j < d_out) {
    a = b2[j];
    for (int i = 0; i < d_out; ++i) a = fmaf(v0[i], w2[(size_t)i * d_out + j], a);
    out[((size_t)m * n_seq + u) * d_out + j] = a;
  }
}

long long cln_mlp_weight_floats(int d_in, int d_out) {
  return (long long)d_in * d_in + d_in + (long long)d_in * d_out + d_out + (long long)d_out * d_out + d_out;
}

int cln_mlp(const float* e, int n_seq, int d_in, int d_out, const float* weights, int n_mlp, float* out, hipStream_t st) {
  TTS_CHECK_ARG(e && weights && out, "cln_mlp: null pointer");
  TTS_CHECK_ARG(d_in > 0 && d_in <= d_out && d_out <= 256, "cln_mlp: needs 0 < d_in <= d_out <= 256, got %d, %d", d_in, d_out);
  if (n_seq == 0 || n_mlp == 0) return TTS_OK;
  hipLaunchKernelGGL(cln_mlp_kernel, dim3(n_seq, n_mlp), dim3(256), 0, st, e, d_in, d_out, weights, cln_mlp_weight_floats(d_in, d_out), out, n_seq);
  return launch_status("cln_mlp");
}

int cond_layernorm(const float* x, int ldx, float* y, int ldy, const float* sc, const float* sh, int c, const TtsTile* tiles,
                   int n_tiles, int tile_rows, hipStream_t st) {
  TTS_CHECK_ARG(c > 0 && c <= 64 * MAX_PER_LANE, "cond_layernorm: c=%d unsupported", c);
  if (n_tiles == 0) return TTS_OK;
  hipLaunchKernelGGL(cln_kernel, dim3(n_tiles, (tile_rows + CLN_ROWS - 1) / CLN_ROWS), dim3(256), 0, st, x, ldx, y, ldy, sc, sh, c, tiles, tile_rows);
  return launch_status("cond_layernorm");
}

int l2_normalize(const float* x, float* y, int rows, int c, hipStream_t st) {
  if (rows == 0) return TTS_OK;
  hipLaunchKernelGGL(l2norm_kernel, dim3(rows), dim3(64), 0, st, x, y, rows, c);
  return launch_status("l2_normalize");
}

int groupnorm(const float* x, int ldx, float* y, int ldy, const float* g, const float* b, int c, int groups, float eps,
              int apply_tanh, const float* res, int ld_res, const int* sb, const int* se, int n_seq, int max_len, float* ws,
              hipStream_t st) {
  TTS_CHECK_ARG(c > 0 && c <= 256 && groups > 0 && c % groups == 0, "groupnorm: c=%d groups=%d unsupported", c, groups);
  TTS_CHECK_ARG(ws != nullptr, "groupnorm: workspace of tts_groupnorm_workspace_floats() floats required");
  if (n_seq == 0 || max_len == 0) return TTS_OK;
  const int n_chunks = (max_len + GN_CHUNK - 1) / GN_CHUNK;
  dim3 grid(n_chunks, n_seq);
  hipLaunchKernelGGL(groupnorm_partial_kernel, grid, dim3(256), 0, st, x, ldx, c, groups, sb, se, ws, n_chunks);
  hipLaunchKernelGGL(groupnorm_apply_kernel, grid, dim3(256), 0, st, x, ldx, y, ldy, g, b, c, groups, eps, apply_tanh, res, ld_res, sb, se,
                     ws, n_chunks);
  return launch_status("groupnorm");
}

long groupnorm_workspace_floats(int n_seq, int max_len, int groups) {
  return (long)n_seq * ((max_len + GN_CHUNK - 1) / GN_CHUNK) * groups * 2;
}


}  // namespace tts
