// Kernels of the per-speaker path behind set_utterance_embedding(path) (ToucanTTSInterface.py:103-114): the GRU and the
// style-token attention of the GST reference encoder (TrainingInterfaces/Spectrogram_to_Embedding/GST.py:144-161, :205-219) and the
// element-wise ends of the log-mel front end (Preprocessing/AudioPreprocessor.py:96-117).  The Conv2d stack, the projections, the
// windowed DFT and the mel projection are dense products and run through tts_conv1d (style.py packs them).  This path runs once
// per reference voice on a few hundred frames: the kernels are written for clarity and exactness (fp32 throughout), not for peak.
#include "common.h"

namespace tts {

// One layer of torch.nn.GRU (batch_first, h0 = 0) for a batch of equally long sequences; one workgroup per sequence, thread j owns
// hidden unit j (gates r, z, n of that unit: rows j, H + j, 2H + j of the weight matrices, which are passed TRANSPOSED -
// [in][3H] - so that consecutive threads read consecutive addresses).
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr);  z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn));  h' = (1 - z) * n + z * h
__global__ __launch_bounds__(256) void gru_layer_kernel(const float* __restrict__ x, int ldx, int T, int I, int H, const float* __restrict__ w_ih_t,
                                                        const float* __restrict__ w_hh_t, const float* __restrict__ b_ih,
                                                        const float* __restrict__ b_hh, float* __restrict__ y, int ldy) {
  extern __shared__ float lds[];
  float* xs = lds;      // [I]
  float* hs = lds + I;  // [H]
  const int b = blockIdx.x, j = threadIdx.x;
  if (j < H) hs[j] = 0.0f;
  for (int t = 0; t < T; ++t) {
    const float* xr = x + ((size_t)b * T + t) * ldx;
    __syncthreads();  // hs of the previous step is complete; xs may be overwritten
    for (int i = j; i < I; i += blockDim.x) xs[i] = xr[i];
    __syncthreads();
    float hn = 0.0f;
    if (j < H) {
      float ir = b_ih[j], iz = b_ih[H + j], in = b_ih[2 * H + j];
      for (int i = 0; i < I; ++i) {
        const float v = xs[i];
        const float* w = w_ih_t + (size_t)i * 3 * H;
        ir = fmaf(v, w[j], ir);
        iz = fmaf(v, w[H + j], iz);
        in = fmaf(v, w[2 * H + j], in);
      }
      float hr = b_hh[j], hz = b_hh[H + j], hh = b_hh[2 * H + j];
      for (int i = 0; i < H; ++i) {
        const float v = hs[i];
        const float* w = w_hh_t + (size_t)i * 3 * H;
        hr = fmaf(v, w[j], hr);
        hz = fmaf(v, w[H + j], hz);
        hh = fmaf(v, w[2 * H + j], hh);
      }
      const float r = 1.0f / (1.0f + expf(-(ir + hr)));
      const float z = 1.0f / (1.0f + expf(-(iz + hz)));
      const float n = tanhf(in + r * hh);
      hn = (1.0f - z) * n + z * hs[j];
    }
    __syncthreads();  // every thread has read the old hs
    if (j < H) {
      hs[j] = hn;
      y[((size_t)b * T + t) * ldy + j] = hn;
    }
  }
}

int gru_layer(const float* x, int ldx, int batch, int T, int I, int H, const float* w_ih_t, const float* w_hh_t, const float* b_ih,
              const float* b_hh, float* y, int ldy, hipStream_t st) {
  TTS_CHECK_ARG(x && w_ih_t && w_hh_t && b_ih && b_hh && y, "gru: null pointer");
  TTS_CHECK_ARG(H > 0 && H <= 256 && I > 0 && I <= 4096 && T > 0, "gru: hidden %d (<= 256), input %d (<= 4096), steps %d", H, I, T);
  if (batch == 0) return TTS_OK;
  hipLaunchKernelGGL(gru_layer_kernel, dim3(batch), dim3(256), (size_t)(I + H) * sizeof(float), st, x, ldx, T, I, H, w_ih_t, w_hh_t, b_ih, b_hh,
                     y, ldy);
  return launch_status("gru");
}

// Style-token attention with one query per utterance (GST.py:205-219, Layers/Attention.py:66-92): workgroup = (utterance, head);
// scores over the N tokens, softmax, weighted sum of the value rows.  q [B, heads*dk] (already projected), k / v [N, heads*dk]
// (projected tanh(tokens): constants of the model, prepared by the host), ctx [B, heads*dk].
__global__ __launch_bounds__(256) void style_tokens_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           int n_tokens, int heads, int dk, float* __restrict__ ctx) {
  __shared__ float red[256];
  __shared__ float acc[256 * 8];
  const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, ld = heads * dk;
  const float scale = 1.0f / sqrtf((float)dk);
  float qv[8];
  for (int d = 0; d < dk; ++d) qv[d] = q[(size_t)b * ld + h * dk + d];
  float m = -INFINITY;
  for (int n = tid; n < n_tokens; n += 256) {
    float s = 0.0f;
    for (int d = 0; d < dk; ++d) s = fmaf(qv[d], k[(size_t)n * ld + h * dk + d], s);
    m = fmaxf(m, s * scale);
  }
  red[tid] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sum = 0.0f;
  for (int n = tid; n < n_tokens; n += 256) {
    float s = 0.0f;
    for (int d = 0; d < dk; ++d) s = fmaf(qv[d], k[(size_t)n * ld + h * dk + d], s);
    const float p = expf(s * scale - m);
    sum += p;
    for (int d = 0; d < dk; ++d) part[d] = fmaf(p, v[(size_t)n * ld + h * dk + d], part[d]);
  }
  red[tid] = sum;
  for (int d = 0; d < 8; ++d) acc[d * 256 + tid] = d < dk ? part[d] : 0.0f;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {  // fixed-order tree: deterministic
    if (tid < o) {
      red[tid] += red[tid + o];
      for (int d = 0; d < dk; ++d) acc[d * 256 + tid] += acc[d * 256 + tid + o];
    }
    __syncthreads();
  }
  if (tid < dk) ctx[(size_t)b * ld + h * dk + tid] = acc[tid * 256] / red[0];
}

int style_tokens(const float* q, const float* k, const float* v, int batch, int n_tokens, int heads, int dk, float* ctx, hipStream_t st) {
  TTS_CHECK_ARG(q && k && v && ctx, "style_tokens: null pointer");
  TTS_CHECK_ARG(dk > 0 && dk <= 8 && heads > 0 && n_tokens > 0, "style_tokens: head dim %d (<= 8), heads %d, tokens %d", dk, heads, n_tokens);
  if (batch == 0) return TTS_OK;
  hipLaunchKernelGGL(style_tokens_kernel, dim3(batch, heads), dim3(256), 0, st, q, k, v, n_tokens, heads, dk, ctx);
  return launch_status("style_tokens");
}

// |X| of a spectrum stored as [rows, re(0..nb) | im(0..nb)] (the windowed-DFT product of the log-mel front end)
__global__ void complex_magnitude_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int rows, int nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nb) return;
  const int r = i / nb, c = i % nb;
  const float re = x[(size_t)r * ldx + c], im = x[(size_t)r * ldx + nb + c];
  y[(size_t)r * ldy + c] = sqrtf(re * re + im * im);
}

int complex_magnitude(const float* x, int ldx, float* y, int ldy, int rows, int nb, hipStream_t st) {
  TTS_CHECK_ARG(x && y && rows >= 0 && nb > 0, "complex_magnitude: bad arguments");
  if (rows == 0) return TTS_OK;
  const long long n = (long long)rows * nb;
  hipLaunchKernelGGL(complex_magnitude_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ldx, y, ldy, rows, nb);
  return launch_status("complex_magnitude");
}

// y = log10(max(eps, x))   (AudioPreprocessor.py:117)
__global__ void log10_floor_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int rows, int c, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c) return;
  const int r = i / c, k = i % c;
  y[(size_t)r * ldy + k] = log10f(fmaxf(eps, x[(size_t)r * ldx + k]));
}

int log10_floor(const float* x, int ldx, float* y, int ldy, int rows, int c, float eps, hipStream_t st) {
  TTS_CHECK_ARG(x && y && rows >= 0 && c > 0, "log10_floor: bad arguments");
  if (rows == 0) return TTS_OK;
  const long long n = (long long)rows * c;
  hipLaunchKernelGGL(log10_floor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ldx, y, ldy, rows, c, eps);
  return launch_status("log10_floor");
}

}  // namespace tts

extern "C" {
int tts_gru_layer(const float* x, int32_t ldx, int32_t batch, int32_t steps, int32_t in_dim, int32_t hidden, const float* w_ih_t,
                  const float* w_hh_t, const float* b_ih, const float* b_hh, float* y, int32_t ldy, tts_stream_t stream) {
  return tts::gru_layer(x, ldx, batch, steps, in_dim, hidden, w_ih_t, w_hh_t, b_ih, b_hh, y, ldy, reinterpret_cast<hipStream_t>(stream));
}
int tts_style_tokens(const float* q, const float* k, const float* v, int32_t batch, int32_t n_tokens, int32_t heads, int32_t dk, float* ctx,
                     tts_stream_t stream) {
  return tts::style_tokens(q, k, v, batch, n_tokens, heads, dk, ctx, reinterpret_cast<hipStream_t>(stream));
}
int tts_complex_magnitude(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t rows, int32_t bins, tts_stream_t stream) {
  return tts::complex_magnitude(x, ldx, y, ldy, rows, bins, reinterpret_cast<hipStream_t>(stream));
}
int tts_log10_floor(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t rows, int32_t c, float eps, tts_stream_t stream) {
  return tts::log10_floor(x, ldx, y, ldy, rows, c, eps, reinterpret_cast<hipStream_t>(stream));
}
}
