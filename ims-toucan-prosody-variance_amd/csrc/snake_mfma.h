// Anti-aliased SnakeBeta with both FIR filters on the matrix cores (one wavefront, one 16-channel block, in place in LDS).
//
//   u[n] = 2 * sum_q x[q] U[n][q]        2x Kaiser-sinc up-sampler (12 taps, 6 per phase)        -> v_mfma_f32_16x16x32_f16
//   s[n] = u[n] + sin^2(u[n] e^alpha) / (e^beta + 1e-9)                                           -> VALU on the accumulators
//   y[t] = sum_n s[n] D[t][n]            12-tap low-pass + decimation by 2                        -> v_mfma_f32_16x16x32_f16
// (BigVGAN/Snake.py:56-69 inside alias_free_torch's Activation1d, restated in snake.h - third party, PARITY UNPINNED.)
//
// U and D are banded Toeplitz matrices: as MFMA A operands they are per-lane constants (4 registers per 16 x 32 block),
// built once per kernel.  The time axis is the contraction (K) axis of both products:
//   * x comes from a [time][channel] 16-bit LDS image; `ds_read_b64_tr_b16` hands every lane 4 consecutive time steps of its
//     channel (B operand: lane = channel, registers = time) - the transpose is free;
//   * the up-sampler's result has the channel on the lane and the 2x-rate time in the accumulator registers, which - converted
//     to fp16 pairs - IS the B operand of the decimator (cdna_hip_programming.md, "An accumulator tile as the next MFMA's
//     operand"): s never touches LDS.  The k order inside a step is permuted (two 4-row accumulator tiles side by side); D is
//     built in that order.
// Per 16 frames x 16 channels: 2 + 2 MFMAs (64 matrix-pipe cycles) and ~55 VALU instructions, against ~420 VALU instructions of
// the register-streamed form (snake_stream in snake.h), and the matrix pipe runs beside the VALU.
//
// Both filters run in fp16 whatever the conv precision: bf16 activations are exact in fp16 (8-bit mantissa into 11; the vocoder's
// activations are O(1), far inside fp16's range), the taps carry 2^-12 relative error and s is rounded to 11 bits - all well below
// the 8-bit rounding of the bf16 conv operand this feeds.
//
// Utterance edges: the reference replicate-pads x (5 | 5) before the up-sampler and s (5 | 6) before the decimator.  Replicate
// padding is linear, so it folds into the constants: a step whose reach leaves [0, T) rebuilds U / D with the out-of-range taps
// summed onto the edge sample (gen_up / gen_down clamp the tap positions; in the interior the clamp is the identity).  Image
// rows outside the utterance only need to be finite (the callers store zeros).  Output frames outside [0, T) are written as
// zero (the convs' zero padding).
//
// In place: a wavefront owns a run of 16-row output tiles of its channel block and sweeps it front to back.  Tile i reads
// image rows up to 16 i + 31 (relative to its first raw row) before it overwrites rows 6 + 16 i .. 6 + 16 i + 15, and the
// next read starts at row 16 i + 32 - 6 > the rows just written; the first fragment (6 rows of the previous run) and the one
// past the end of the run (rows of the next run) are fetched in `begin()`, BEFORE the workgroup barrier that precedes `sweep()`.
#pragma once
#include "common.h"

namespace tts {

typedef short tr16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct FirTaps {  // the 12 taps by value (register arguments of the out-of-line constant builders)
  float v[12];
  __device__ __forceinline__ float operator[](int i) const { return v[i]; }
};

struct SnakeFir {
  // image geometry (16-bit elements): row r of the image <-> local frame frame0 + r; outputs live 6 rows below their inputs' start
  unsigned short* img;
  int pitch;            // rows up to row_begin + 16 n_tiles + 31 are read: the image must hold finite values there (zero taps meet them)
  int frame0, T;
  int ch0;              // first channel of this wavefront's 16-channel block
  int row_begin;        // first raw row of the run: outputs start at row_begin + 6
  int n_tiles;          // 16-row output tiles of the run
  float er, inv_b;      // e^alpha / 2 pi (phase in revolutions per unit of u), 1 / (e^beta + 1e-9) of this lane's channel
  bf16x8 ua0, ua1, da0, da1;  // interior constants (fp16 bit patterns)
  bf16x8 x_head, x_tail;

  static __device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 p = {pack16<true>(v[0], v[1]), pack16<true>(v[2], v[3]), pack16<true>(v[4], v[5]), pack16<true>(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, p);
  }

  // A operand of the up-sampler for outputs n = 2 qo + p, qo = fw + 3 + 8 i + (row >> 1), from inputs q = fw + 8 g + j (frames).
  // u[2 qo] = 2 sum_{d=-3..2} x[qo + d] f[5 - 2 d];  u[2 qo + 1] = 2 sum_{d=-2..3} x[qo + d] f[6 - 2 d];  x replicate padded.
  static __device__ __attribute__((noinline)) bf16x8 gen_up(const FirTaps f, int i, int fw, int T, int lane) {
    const int row = lane & 15, g = lane >> 4;
    const int qo = fw + 3 + 8 * i + (row >> 1), p = row & 1;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = fw + 8 * g + j;
      float a = 0.0f;
#pragma unroll
      for (int e = 0; e < 6; ++e) {  // p = 0: d = e - 3, tap 11 - 2 e;  p = 1: d = e - 2, tap 10 - 2 e
        int qq = qo + e - 3 + p;
        qq = qq < 0 ? 0 : (qq > T - 1 ? T - 1 : qq);
        const float tap = p ? f[10 - 2 * e] : f[11 - 2 * e];
        a += (qq == q) ? tap : 0.0f;
      }
      v[j] = (qo >= 0 && qo < T) ? 2.0f * a : 0.0f;
    }
    return pack8(v);
  }

  // A operand of the decimator, K-step s of 2: outputs t = fo + row, y[t] = sum_k s2[clamp(2 t + k - 5)] f[k]; the k slot
  // (g, j) of the step holds 2x-rate sample n = 2 fo - 6 + 32 s + 4 g + j + (j >= 4 ? 12 : 0) (two accumulator tiles side by side)
  static __device__ __attribute__((noinline)) bf16x8 gen_down(const FirTaps f, int s, int fo, int T, int lane) {
    const int row = lane & 15, g = lane >> 4;
    const int t = fo + row;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = 2 * fo - 6 + 32 * s + 4 * g + j + (j >= 4 ? 12 : 0);
      float a = 0.0f;
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        int nn = 2 * t + k - 5;
        nn = nn < 0 ? 0 : (nn > 2 * T - 1 ? 2 * T - 1 : nn);
        a += (nn == n) ? f[k] : 0.0f;
      }
      v[j] = (t >= 0 && t < T) ? a : 0.0f;
    }
    return pack8(v);
  }

  // B operand of the up-sampler: image rows w .. w + 31, this block's 16 channels (lane: channel = lane & 15, k = 8 (lane >> 4) + j)
  __device__ __forceinline__ bf16x8 load_x(int w, int lane) const {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    typedef tr16x4 __attribute__((address_space(3))) * lds_ptr;
    const unsigned short* a = img + (w + 8 * g + q) * pitch + ch0 + 4 * p;
    const tr16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a);
    const tr16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * pitch));
    const bf16x8 x = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return x;
  }

  // interior test of the window starting at image row w (inputs w .. w + 21, outputs qo = w + 3 .. w + 18)
  __device__ __forceinline__ bool up_interior(int w) const { return frame0 + w >= 0 && frame0 + w + 21 <= T - 1; }
  __device__ __forceinline__ bool down_interior(int fo) const { return fo >= 3 && fo + 18 <= T - 1; }
  // ... and of a window / output tile that lies outside the utterance altogether: every row of its operand is zero (gen_up / gen_down
  // would return exactly that, after ~500 vector instructions each).  The tiles in front of an utterance's first frame and behind its
  // last one - most of the last 224-row tile of an utterance - are of this kind: rebuilding their constants step by step made that
  // one workgroup 3.5x slower than the others (107 k instead of 7.4 k cycles per sweep; at batch 1 the launch waited for it:
  // 147 us instead of ~45 us at C = 128).  Only the one or two steps that straddle an edge still build constants.
  __device__ __forceinline__ bool up_outside(int w) const { return frame0 + w + 18 < 0 || frame0 + w + 3 > T - 1; }
  __device__ __forceinline__ bool down_outside(int fo) const { return fo + 15 < 0 || fo > T - 1; }

  // x window -> 32 2x-rate samples of s as a K-step fragment of the decimator (INNER: no tap folds onto an utterance edge)
  template <bool INNER>
  __device__ __forceinline__ bf16x8 make_s(const FirTaps& f, const bf16x8& x, int w, int lane) const {
    bf16x8 a0 = ua0, a1 = ua1;
    if constexpr (!INNER) {
      if (!up_interior(w)) {  // wave-uniform
        if (up_outside(w)) {
          a0 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
          a1 = a0;
        } else {
          a0 = gen_up(f, 0, frame0 + w, T, lane);
          a1 = gen_up(f, 1, frame0 + w, T, lane);
        }
      }
    }
#ifdef SNAKE_SIN2  // (A/B switch: the five-operation form u + inv_b sin^2(u e^alpha))
    const f32x4v z = {0.f, 0.f, 0.f, 0.f};
    const f32x4v u0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, x), z, 0, 0, 0);
    const f32x4v u1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, x), z, 0, 0, 0);
    float s[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float u = r < 4 ? u0[r] : u1[r - 4];
      const float sn = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(u * er));  // v_sin takes revolutions
      s[r] = fmaf(inv_b, sn * sn, u);
    }
    return pack8(s);
#else
    // u + b sin^2(theta) = (u + b/2) - (b/2) cos(2 theta), theta = u e^alpha: the up-sampler's accumulators START from b/2 (a free add),
    // the phase comes out of one fma ((u + b/2) (e^alpha / pi) - (b/2) (e^alpha / pi), in revolutions), and the result out of one
    // more: four vector operations per sample instead of five - in a kernel whose first limit is vector issue.
    const float hb = 0.5f * inv_b, er2 = 2.0f * er, c0 = -hb * er2;
    const f32x4v z = {hb, hb, hb, hb};
    const f32x4v u0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, x), z, 0, 0, 0);
    const f32x4v u1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, x), z, 0, 0, 0);
    float s[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float up = r < 4 ? u0[r] : u1[r - 4];  // u + b/2
      const float cs = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(fmaf(up, er2, c0)));  // v_cos takes revolutions
      s[r] = fmaf(-hb, cs, up);
    }
    return pack8(s);
#endif
  }

  // everything a run needs from rows other wavefronts will overwrite: call before the workgroup barrier that precedes sweep()
  // interior constants: built once per filter on the host (snake_fir_table), one 16-byte load per operand and lane
  __device__ __forceinline__ void load_constants(const void* fir_tab, int lane) {
    const bf16x8* tab = reinterpret_cast<const bf16x8*>(fir_tab);
    ua0 = tab[0 * 64 + lane];
    ua1 = tab[1 * 64 + lane];
    da0 = tab[2 * 64 + lane];
    da1 = tab[3 * 64 + lane];
  }
  __device__ __forceinline__ void copy_constants(const SnakeFir& o) {
    ua0 = o.ua0; ua1 = o.ua1; da0 = o.da0; da1 = o.da1;
  }
  __device__ __forceinline__ void begin(int lane) {
    x_head = load_x(row_begin, lane);
    x_tail = load_x(row_begin + 16 * n_tiles, lane);
  }

  // The decimator runs TRANSPOSED: y^T = s^T D^T, i.e. the s fragment is the A operand (A[row = channel][k]) and the D fragment
  // the B operand (B[k][col = frame] has the same lane map as A[row = frame][k]: the same registers serve).  The result then has
  // the frame on the lane and four consecutive channels in the registers: one 8-byte LDS store per lane instead of four 2-byte ones.
  template <bool F16OUT, bool INNER>
  __device__ __forceinline__ void run(const FirTaps& f, int lane) {
    const int t = lane & 15, g = lane >> 4;
    bf16x8 s_prev = make_s<INNER>(f, x_head, row_begin, lane);
    unsigned short* dst = img + (row_begin + 6 + t) * pitch + ch0 + 4 * g;
    for (int i = 0; i < n_tiles; ++i, dst += 16 * pitch) {
      const int w = row_begin + 16 * (i + 1);
      const bf16x8 x = (i + 1 == n_tiles) ? x_tail : load_x(w, lane);
      const bf16x8 s_cur = make_s<INNER>(f, x, w, lane);
      bf16x8 d0 = da0, d1 = da1;
      if constexpr (!INNER) {
        const int fo = frame0 + row_begin + 6 + 16 * i;
        if (!down_interior(fo)) {  // (rows of D for frames outside [0, T) are zero: those outputs are the convs' zero padding)
          if (down_outside(fo)) {
            d0 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            d1 = d0;
          } else {
            d0 = gen_down(f, 0, fo, T, lane);
            d1 = gen_down(f, 1, fo, T, lane);
          }
        }
      }
      f32x4v y = {0.f, 0.f, 0.f, 0.f};
      y = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, s_prev), __builtin_bit_cast(f16x8, d0), y, 0, 0, 0);
      y = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, s_cur), __builtin_bit_cast(f16x8, d1), y, 0, 0, 0);
      *reinterpret_cast<uint2*>(dst) = make_uint2(pack16<F16OUT>(y[0], y[1]), pack16<F16OUT>(y[2], y[3]));
      s_prev = s_cur;
    }
  }

  template <bool F16OUT>
  __device__ __forceinline__ void sweep(const FirTaps& f, int lane) {
    // interior frames form one range: the run is interior iff its first and last windows / output tiles are
    const bool inner = up_interior(row_begin) && up_interior(row_begin + 16 * n_tiles) && down_interior(frame0 + row_begin + 6) &&
                       down_interior(frame0 + row_begin + 6 + 16 * (n_tiles - 1));
    if (inner) run<F16OUT, true>(f, lane);
    else run<F16OUT, false>(f, lane);
  }
};

}  // namespace tts
