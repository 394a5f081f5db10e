// Anti-aliased SnakeBeta on ROWS consecutive frames of one channel, entirely in registers.
// (BigVGAN/Snake.py:56-69 inside alias_free_torch's Activation1d - third party, PARITY UNPINNED, restated from its
// published algorithm):
//   u[2q]   = 2 * sum_{d=-3..2} x[q+d] f[5-2d]      u[2q+1] = 2 * sum_{d=-2..3} x[q+d] f[6-2d]   (x replicate padded)
//   s[n]    = u[n] + sin^2(u[n] * e^alpha) / (e^beta + 1e-9)
//   y[t]    = sum_{k=0..11} s[clamp(2t + k - 5, 0, 2T-1)] f[k]
// Shared by the stand-alone kernel (sequence_ops.hip) and the conv input staging (conv1d.hip, TTS_PRE_SNAKE).
#pragma once
#include "common.h"

namespace tts {

// The sine is one hardware v_sin_f32 (argument in revolutions, reduced by v_fract_f32): absolute error ~1e-6 for phases up to
// a few hundred radians (tests/test_gpu_kernels.py::test_snake_aa checks it against the fp64 emulator), versus ~1e-7 for
// ocml's sinf at 5x the cost and 3x the registers.

// out[i] = snake_aa(x)[t0 + i] for i < ROWS; t0 is the local frame index inside an utterance of T frames; load(q) returns
// frame q of the utterance (0 <= q < T, already clamped).  t0 may be negative / rows may lie beyond T: those outputs are
// meaningless (callers mask them), but every output with 0 <= t0+i < T is exact provided the group overlaps [0, T).
template <int ROWS, class LoadFn>
__device__ __forceinline__ void snake_rows_fn(LoadFn load, int T, int t0, const float (&f)[12], float ea, float inv_b, float (&out)[ROWS]) {
  constexpr int NX = ROWS + 12, NS = 2 * ROWS + 10;
  const float er = ea * 0.3183098861837907f, hb = 0.5f * inv_b;
  float xin[NX];  // x[t0-6 .. t0+ROWS+5], replicate padded inside the utterance
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    int q = t0 - 6 + i;
    q = q < 0 ? 0 : (q > T - 1 ? T - 1 : q);
    xin[i] = load(q);
  }
  float s[NS];  // s[m] <-> n = 2*t0 - 5 + m
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    // q = floor(n/2) = t0 - 3 + ((m+1)>>1);  xin index of x[q+d] = 3 + ((m+1)>>1) + d
    const int qi = 3 + ((m + 1) >> 1);
    float u = 0.f;  // h = u/2 (see snake_stream: the factor 2 is applied once per output frame)
    if (((m + 1) & 1) == 0) {  // n even (m odd): taps f[5-2d], d = -3..2
#pragma unroll
      for (int d = -3; d <= 2; ++d) u = fmaf(xin[qi + d], f[5 - 2 * d], u);
    } else {  // n odd: taps f[6-2d], d = -2..3
#pragma unroll
      for (int d = -2; d <= 3; ++d) u = fmaf(xin[qi + d], f[6 - 2 * d], u);
    }
    const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(u * er));
    s[m] = fmaf(hb, sn * sn, u);
  }
  // replicate padding of the 2x-rate signal: positions n < 0 take s[n=0], n > 2T-1 take s[n=2T-1]
  const int nbase = 2 * t0 - 5;
  if (nbase < 0 || nbase + NS - 1 > 2 * T - 1) {
    float s_lo = 0.f, s_hi = 0.f;
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      if (nbase + m == 0) s_lo = s[m];
      if (nbase + m == 2 * T - 1) s_hi = s[m];
    }
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      if (nbase + m < 0) s[m] = s_lo;
      if (nbase + m > 2 * T - 1) s[m] = s_hi;
    }
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 12; ++k) a = fmaf(s[2 * i + k], f[k], a);
    out[i] = 2.0f * a;
  }
}

// Streaming form: 8*NCH consecutive frames starting at local frame t0.  Chunk c reuses the last 12 input frames and the
// last 10 2x-rate samples of chunk c-1 (they are exactly the halo it would otherwise recompute), so only the first chunk
// pays the (26 samples / 8 frames) halo overhead of snake_rows_fn; with NCH = 5 the up-sampler + sine work per frame drops
// from 3.25 to 2.25 samples.  store(i, v) receives frame t0+i (callers mask frames outside [0, T)).  The 8 frames a chunk
// adds are requested one chunk ahead (reads run up to 21 frames ahead of the stores), so their latency hides under the math.
// PREFETCH = false keeps 8 registers free (the 80-register C = 32 residual step is faster without it).
template <int NCH, bool PREFETCH = true, class LoadFn, class StoreFn>
__device__ __forceinline__ void snake_stream(LoadFn load, StoreFn store, int T, int t0, const float (&f)[12], float ea, float inv_b) {
  // The factor 2 of the zero-stuffed interpolation is carried as a scale: h = u/2 and s/2 = h + (1/2b) sin^2(2h e^alpha) are
  // what the registers hold, and the 12-tap decimator output is doubled once per frame (one multiply per frame instead of
  // one per 2x-rate sample, and no second set of taps in scalar registers).
  const float er = ea * 0.3183098861837907f;  // phase in revolutions per unit of h
  const float hb = 0.5f * inv_b;
  float xin[20];  // x[tc-6 .. tc+13] of the current chunk
  float nxt[8];   // the 8 frames the next chunk adds, requested one chunk ahead so their latency hides under this chunk's math
  float s[26];    // s[m] <-> n = 2*tc - 5 + m
#pragma unroll
  for (int i = 0; i < 20; ++i) {
    int q = t0 - 6 + i;
    q = q < 0 ? 0 : (q > T - 1 ? T - 1 : q);
    xin[i] = load(q);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int tc = t0 + 8 * c;
    if (c > 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) xin[i] = xin[i + 8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (PREFETCH) {
          xin[12 + i] = nxt[i];
        } else {
          int q = tc + 6 + i;
          q = q < 0 ? 0 : (q > T - 1 ? T - 1 : q);
          xin[12 + i] = load(q);
        }
      }
    }
    if (PREFETCH && c + 1 < NCH) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int q = tc + 14 + i;
        q = q < 0 ? 0 : (q > T - 1 ? T - 1 : q);
        nxt[i] = load(q);
      }
    }
#pragma unroll
    for (int m = 0; m < 26; ++m) {
      if (c > 0 && m < 10) {
        s[m] = s[m + 16];  // positions 2tc-5 .. 2tc+4 were the tail of the previous chunk (already edge-corrected)
      } else {
        const int qi = 3 + ((m + 1) >> 1);
        float u;
        if (((m + 1) & 1) == 0) {
          u = xin[qi - 3] * f[11];
#pragma unroll
          for (int d = -2; d <= 2; ++d) u = fmaf(xin[qi + d], f[5 - 2 * d], u);
        } else {
          u = xin[qi - 2] * f[10];
#pragma unroll
          for (int d = -1; d <= 3; ++d) u = fmaf(xin[qi + d], f[6 - 2 * d], u);
        }
        const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(u * er));  // v_sin takes revolutions; sin(2 pi frac(r)) == sin(2 pi r)
        s[m] = fmaf(hb, sn * sn, u);
      }
    }
    const int nbase = 2 * tc - 5;
    if (nbase < 0 || nbase + 25 > 2 * T - 1) {
      float s_lo = 0.f, s_hi = 0.f;
#pragma unroll
      for (int m = 0; m < 26; ++m) {
        if (nbase + m == 0) s_lo = s[m];
        if (nbase + m == 2 * T - 1) s_hi = s[m];
      }
#pragma unroll
      for (int m = 0; m < 26; ++m) {
        if (nbase + m < 0) s[m] = s_lo;
        if (nbase + m > 2 * T - 1) s[m] = s_hi;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = s[2 * i] * f[0];
#pragma unroll
      for (int k = 1; k < 12; ++k) a = fmaf(s[2 * i + k], f[k], a);
      store(8 * c + i, 2.0f * a);
    }
  }
}

// The same on a packed fp32 tensor in global memory: channel ch of the utterance starting at packed row seq_begin.
template <int ROWS>
__device__ __forceinline__ void snake_rows(const float* __restrict__ x, int ldx, int ch, int seq_begin, int T, int t0,
                                           const float (&f)[12], float ea, float inv_b, float (&out)[ROWS]) {
  snake_rows_fn<ROWS>([&](int q) { return x[(size_t)(seq_begin + q) * ldx + ch]; }, T, t0, f, ea, inv_b, out);
}

}  // namespace tts
