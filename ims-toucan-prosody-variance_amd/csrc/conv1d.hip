// Implicit-GEMM 1-D convolution over packed, time-major activations on the gfx950 matrix cores.
//
//   acc[r, n] = sum_tap sum_ci pre(x[r + tap*dil - pad_left, ci]) * w[tap][ci][n]
//
// Work decomposition: one 256-thread workgroup (4 wavefronts of 64) owns a BM x BN output tile of ONE
// utterance (tile table), so the time halo never crosses an utterance and rows outside the utterance read
// as zero - exactly the reference's per-utterance zero padding.  For every 32-channel slab of the input the
// workgroup stages the (BM + (taps-1)*dil) x 32 activation window into LDS ONCE and reuses it for all taps
// (the window is what makes a k-tap conv cost one activation read instead of k); per tap a 32 x BN weight
// slab is staged and each wavefront issues 16 k-steps of v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, so
// the result matches an fp32 reference to rounding-order level).  The bf16 variant (compute == 1) stages
// the same window as bf16 and uses v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// LDS layout: activations [rows][32+1] floats (odd pitch -> the A-fragment column read, 32 lanes on 32
// consecutive rows, hits 32 distinct banks); weights [32][BN] floats (B-fragment read is a contiguous row).
//
// Reference ops replaced: see include/toucan_tts.h (tts_conv1d).
#include "common.h"
#include "snake.h"

namespace tts {

struct ConvArgs {
  TtsConvDesc d;
};

template <int TM, int TN, int WAVES_M, int WAVES_N, bool DUAL>
struct ConvCfg {
  static constexpr int BM = TM * WAVES_M * 32;
  static constexpr int BN = TN * WAVES_N * 32;
  static constexpr int BK = 32;
  static constexpr int XP = BK + 1;  // activation pitch in floats
};

__device__ __forceinline__ float pre_activation(float v, int pre_act, float slope) {
  if (pre_act == TTS_PRE_LRELU) return v > 0.0f ? v : v * slope;
  return v;
}

template <int TM, int TN, int WAVES_M, int WAVES_N, bool DUAL>
__global__ __launch_bounds__(256) void conv1d_f32_kernel(const TtsConvDesc d) {
  using C = ConvCfg<TM, TN, WAVES_M, WAVES_N, DUAL>;
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK, XP = C::XP;
  constexpr int NH = DUAL ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const TtsTile tile = d.tiles[blockIdx.x];
  const int n0 = blockIdx.y * BN;  // first output column (within a half in dual mode)
  const int halo = (d.taps - 1) * d.dil;
  const int win_rows = BM + halo;
  float* xs = lds;                  // [win_rows][XP]
  float* ws = lds + win_rows * XP;  // [NH][BK][BN]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int lrow = lane & 31;
  const int lk = lane >> 5;

  f32x16 acc[NH][TM][TN];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.0f;

  const int row_first = tile.row0 - d.pad_left;  // packed row of window row 0
  const bool vec_ok = ((d.ldx & 3) == 0) && ((d.cin & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.x) & 15) == 0);
  const float* __restrict__ W = reinterpret_cast<const float*>(d.w);
  const int n_chunks = d.cin_pad / BK;

  for (int ch = 0; ch < n_chunks; ++ch) {
    const int c0 = ch * BK;
    __syncthreads();  // previous chunk's readers are done with xs
    // ---- stage the activation window: win_rows x 32 channels ----
    if (d.pre_act == TTS_PRE_SNAKE) {
      // anti-aliased snake computed while staging: item = (8 window rows, channel); the activated tensor never reaches HBM
      float f[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) f[k] = d.snake_filt[k];
      const int T = tile.seq_end - tile.seq_begin;
      const int items = ((win_rows + 7) >> 3) * BK;
      for (int it = tid; it < items; it += 256) {
        const int chl = it & 31, wr0 = (it >> 5) * 8;
        const int cg = c0 + chl;
        const int t0 = row_first + wr0 - tile.seq_begin;  // local frame of the group's first row (may be < 0)
        float o[8];
        const bool live = cg < d.cin && t0 + 7 >= 0 && t0 < T;
        if (live) snake_rows<8>(d.x, d.ldx, cg, tile.seq_begin, T, t0, f, expf(d.snake_alpha[cg]), 1.0f / (expf(d.snake_beta[cg]) + 1e-9f), o);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (wr0 + i < win_rows) xs[(wr0 + i) * XP + chl] = (live && t0 + i >= 0 && t0 + i < T) ? o[i] : 0.0f;
      }
    } else if (vec_ok) {
      for (int e = tid; e < win_rows * (BK / 4); e += 256) {
        const int wr = e >> 3, c4 = (e & 7) * 4;
        const int gr = row_first + wr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr >= tile.seq_begin && gr < tile.seq_end && (c0 + c4) < d.cin)
          v = *reinterpret_cast<const float4*>(d.x + (size_t)gr * d.ldx + c0 + c4);
        float* dst = xs + wr * XP + c4;
        dst[0] = pre_activation(v.x, d.pre_act, d.pre_slope);
        dst[1] = pre_activation(v.y, d.pre_act, d.pre_slope);
        dst[2] = pre_activation(v.z, d.pre_act, d.pre_slope);
        dst[3] = pre_activation(v.w, d.pre_act, d.pre_slope);
      }
    } else {
      for (int e = tid; e < win_rows * BK; e += 256) {
        const int wr = e >> 5, c = e & 31;
        const int gr = row_first + wr;
        float v = 0.f;
        if (gr >= tile.seq_begin && gr < tile.seq_end && (c0 + c) < d.cin) v = d.x[(size_t)gr * d.ldx + c0 + c];
        xs[wr * XP + c] = pre_activation(v, d.pre_act, d.pre_slope);
      }
    }
    for (int tap = 0; tap < d.taps; ++tap) {
      __syncthreads();  // xs visible (first tap) / previous tap's readers done with ws
      // ---- stage the weight slab(s): [BK][BN] per half ----
      const float* wsrc = W + ((size_t)tap * d.cin_pad + c0) * d.wn + n0;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        for (int e = tid; e < BK * (BN / 4); e += 256) {
          const int k = e / (BN / 4), c4 = (e % (BN / 4)) * 4;
          const float4 v = *reinterpret_cast<const float4*>(wsrc + (size_t)k * d.wn + h * d.half_pad + c4);
          *reinterpret_cast<float4*>(ws + (h * BK + k) * BN + c4) = v;
        }
      }
      __syncthreads();
      const float* xa = xs + (wm * TM * 32 + lrow + tap * d.dil) * XP + lk;
      const float* wb = ws + lk * BN + wn * TN * 32 + lrow;
#pragma unroll 4
      for (int kk = 0; kk < BK / 2; ++kk) {
        float a[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = xa[i * 32 * XP + 2 * kk];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          float b[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[j] = wb[(h * BK + 2 * kk) * BN + j * 32];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[h][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[h][i][j], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + lrow;
      if (n >= d.cout) continue;
      const float ba = d.bias ? d.bias[n] : 0.0f;
      const float bg = (DUAL && d.bias) ? d.bias[d.cout + n] : 0.0f;
      const float sv = d.seqvec ? d.seqvec[(size_t)tile.seq_id * d.ld_seqvec + n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tile.row0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= tile.seq_end) continue;
        float v = acc[0][i][j][r] + ba + sv;
        if (d.preadd) v += d.preadd[(size_t)row * d.ld_preadd + n];
        if (DUAL) {
          float g = acc[NH - 1][i][j][r] + bg;
          if (d.preadd) g += d.preadd[(size_t)row * d.ld_preadd + d.cout + n];
          if (d.mode == TTS_MODE_GLU) {
            v = v * (1.0f / (1.0f + expf(-g)));
          } else if (d.mode == TTS_MODE_GATED) {
            v = tanhf(v) * (1.0f / (1.0f + expf(-g)));
          } else {  // COUPLING
            v = (d.aux[(size_t)row * d.ld_aux + n] - v) * expf(-g);
          }
        } else {
          if (d.act == TTS_ACT_RELU) v = fmaxf(v, 0.0f);
          else if (d.act == TTS_ACT_TANH) v = tanhf(v);
        }
        v *= d.alpha;
        if (d.res) v += d.res_scale * d.res[(size_t)row * d.ld_res + n];
        float* yp = d.y + (size_t)row * d.ldy + n;
        if (d.accumulate) v += *yp;
        *yp = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 variant: same decomposition, activations/weights converted to bf16 while staging,
// v_mfma_f32_32x32x16_bf16 with fp32 accumulators.  A fragment: lane l holds A[row l&31][k = 8*(l>>5)+j],
// B fragment: B[k = 8*(l>>5)+j][col l&31], j = 0..7 (cdna_hip_programming.md section 3).
// Weights arrive pre-packed by the host as bf16 [taps][cin_pad/8][wn][8] so a B fragment is one 16-byte read.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short f2bf(float f) {
  const __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (round to nearest even, NaN preserved)
  return __builtin_bit_cast(unsigned short, b);
}

template <int TM, int TN, int WAVES_M, int WAVES_N, bool DUAL>
__global__ __launch_bounds__(256) void conv1d_bf16_kernel(const TtsConvDesc d) {
  using C = ConvCfg<TM, TN, WAVES_M, WAVES_N, DUAL>;
  constexpr int BM = C::BM, BN = C::BN, BK = 32;
  constexpr int NH = DUAL ? 2 : 1;
  constexpr int XPB = BK + 8;  // bf16 elements per activation row (80 B pitch: 16-B aligned, conflict-light)
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const TtsTile tile = d.tiles[blockIdx.x];
  const int n0 = blockIdx.y * BN;
  const int halo = (d.taps - 1) * d.dil;
  const int win_rows = BM + halo;
  unsigned short* xs = reinterpret_cast<unsigned short*>(lds);  // [win_rows][XPB]
  unsigned short* ws = xs + ((win_rows * XPB + 7) & ~7);        // [NH][BK/8][BN][8]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int lrow = lane & 31;
  const int lk = lane >> 5;

  f32x16 acc[NH][TM][TN];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.0f;

  const int row_first = tile.row0 - d.pad_left;
  const bool vec_ok = ((d.ldx & 3) == 0) && ((d.cin & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.x) & 15) == 0);
  const unsigned short* __restrict__ W = reinterpret_cast<const unsigned short*>(d.w);
  const int n_chunks = d.cin_pad / BK;

  for (int ch = 0; ch < n_chunks; ++ch) {
    const int c0 = ch * BK;
    __syncthreads();
    if (d.pre_act == TTS_PRE_SNAKE) {
      float f[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) f[k] = d.snake_filt[k];
      const int T = tile.seq_end - tile.seq_begin;
      const int items = ((win_rows + 7) >> 3) * BK;
      for (int it = tid; it < items; it += 256) {
        const int chl = it & 31, wr0 = (it >> 5) * 8;
        const int cg = c0 + chl;
        const int t0 = row_first + wr0 - tile.seq_begin;
        float o[8];
        const bool live = cg < d.cin && t0 + 7 >= 0 && t0 < T;
        if (live) snake_rows<8>(d.x, d.ldx, cg, tile.seq_begin, T, t0, f, expf(d.snake_alpha[cg]), 1.0f / (expf(d.snake_beta[cg]) + 1e-9f), o);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (wr0 + i < win_rows) xs[(wr0 + i) * XPB + chl] = f2bf((live && t0 + i >= 0 && t0 + i < T) ? o[i] : 0.0f);
      }
    } else if (vec_ok) {
      for (int e = tid; e < win_rows * (BK / 4); e += 256) {
        const int wr = e >> 3, c4 = (e & 7) * 4;
        const int gr = row_first + wr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr >= tile.seq_begin && gr < tile.seq_end && (c0 + c4) < d.cin)
          v = *reinterpret_cast<const float4*>(d.x + (size_t)gr * d.ldx + c0 + c4);
        ushort4 o;
        o.x = f2bf(pre_activation(v.x, d.pre_act, d.pre_slope));
        o.y = f2bf(pre_activation(v.y, d.pre_act, d.pre_slope));
        o.z = f2bf(pre_activation(v.z, d.pre_act, d.pre_slope));
        o.w = f2bf(pre_activation(v.w, d.pre_act, d.pre_slope));
        *reinterpret_cast<ushort4*>(xs + wr * XPB + c4) = o;
      }
    } else {
      for (int e = tid; e < win_rows * BK; e += 256) {
        const int wr = e >> 5, c = e & 31;
        const int gr = row_first + wr;
        float v = 0.f;
        if (gr >= tile.seq_begin && gr < tile.seq_end && (c0 + c) < d.cin) v = d.x[(size_t)gr * d.ldx + c0 + c];
        xs[wr * XPB + c] = f2bf(pre_activation(v, d.pre_act, d.pre_slope));
      }
    }
    for (int tap = 0; tap < d.taps; ++tap) {
      __syncthreads();
      // weight slab: [BK/8][BN][8] bf16 per half; 16-byte units
      const unsigned short* wsrc = W + (((size_t)tap * (d.cin_pad / 8) + c0 / 8) * d.wn + n0) * 8;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        for (int e = tid; e < (BK / 8) * BN; e += 256) {
          const int kb = e / BN, n = e % BN;
          const uint4 v = *reinterpret_cast<const uint4*>(wsrc + ((size_t)kb * d.wn + h * d.half_pad + n) * 8);
          *reinterpret_cast<uint4*>(ws + ((h * (BK / 8) + kb) * BN + n) * 8) = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const bf16x8*>(xs + (wm * TM * 32 + i * 32 + lrow + tap * d.dil) * XPB + ks * 16 + lk * 8);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          bf16x8 b[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j)
            b[j] = *reinterpret_cast<const bf16x8*>(ws + ((h * (BK / 8) + ks * 2 + lk) * BN + wn * TN * 32 + j * 32 + lrow) * 8);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[h][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[h][i][j], 0, 0, 0);
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + lrow;
      if (n >= d.cout) continue;
      const float ba = d.bias ? d.bias[n] : 0.0f;
      const float bg = (DUAL && d.bias) ? d.bias[d.cout + n] : 0.0f;
      const float sv = d.seqvec ? d.seqvec[(size_t)tile.seq_id * d.ld_seqvec + n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tile.row0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= tile.seq_end) continue;
        float v = acc[0][i][j][r] + ba + sv;
        if (d.preadd) v += d.preadd[(size_t)row * d.ld_preadd + n];
        if (DUAL) {
          float g = acc[NH - 1][i][j][r] + bg;
          if (d.preadd) g += d.preadd[(size_t)row * d.ld_preadd + d.cout + n];
          if (d.mode == TTS_MODE_GLU) {
            v = v * (1.0f / (1.0f + expf(-g)));
          } else if (d.mode == TTS_MODE_GATED) {
            v = tanhf(v) * (1.0f / (1.0f + expf(-g)));
          } else {
            v = (d.aux[(size_t)row * d.ld_aux + n] - v) * expf(-g);
          }
        } else {
          if (d.act == TTS_ACT_RELU) v = fmaxf(v, 0.0f);
          else if (d.act == TTS_ACT_TANH) v = tanhf(v);
        }
        v *= d.alpha;
        if (d.res) v += d.res_scale * d.res[(size_t)row * d.ld_res + n];
        float* yp = d.y + (size_t)row * d.ldy + n;
        if (d.accumulate) v += *yp;
        *yp = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// shape -> tile configuration
// ------------------------------------------------------------------------------------------------
enum ConvShape { S_128x128, S_128x96, S_128x64, S_256x32, S_D128x96 };

static ConvShape pick_shape(int cout, int mode) {
  if (mode != TTS_MODE_LINEAR) return S_D128x96;
  if (cout <= 32) return S_256x32;
  if (cout <= 64) return S_128x64;
  const int pad96 = (cout + 95) / 96 * 96, pad128 = (cout + 127) / 128 * 128;
  return (pad96 < pad128) ? S_128x96 : S_128x128;
}

static void shape_dims(ConvShape s, int& bm, int& bn) {
  switch (s) {
    case S_128x128: bm = 128; bn = 128; break;
    case S_128x96: bm = 128; bn = 96; break;
    case S_128x64: bm = 128; bn = 64; break;
    case S_256x32: bm = 256; bn = 32; break;
    case S_D128x96: bm = 128; bn = 96; break;
  }
}

template <int TM, int TN, int WAVES_M, int WAVES_N, bool DUAL>
static int launch_cfg(const TtsConvDesc& d, hipStream_t st) {
  using C = ConvCfg<TM, TN, WAVES_M, WAVES_N, DUAL>;
  const int halo = (d.taps - 1) * d.dil;
  const int win_rows = C::BM + halo;
  const int nh = DUAL ? 2 : 1;
  const int n_tiles_n = ((DUAL ? d.half_pad : d.wn) + C::BN - 1) / C::BN;
  dim3 grid(d.n_tiles, n_tiles_n), block(256);
  if (d.compute == 0) {
    size_t lds = (size_t)(win_rows * C::XP + nh * C::BK * C::BN) * sizeof(float);
    TTS_CHECK_ARG(lds <= 160 * 1024, "conv1d: LDS %zu B exceeds 160 KiB (taps %d dil %d)", lds, d.taps, d.dil);
    auto k = conv1d_f32_kernel<TM, TN, WAVES_M, WAVES_N, DUAL>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, st, d);
  } else {
    size_t xs_elems = ((size_t)win_rows * (32 + 8) + 7) & ~(size_t)7;
    size_t lds = (xs_elems + (size_t)nh * 32 * C::BN) * sizeof(unsigned short);
    TTS_CHECK_ARG(lds <= 160 * 1024, "conv1d(bf16): LDS %zu B exceeds 160 KiB", lds);
    auto k = conv1d_bf16_kernel<TM, TN, WAVES_M, WAVES_N, DUAL>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, st, d);
  }
  return launch_status("conv1d");
}

int conv1d_dispatch(const TtsConvDesc& d, hipStream_t st) {
  TTS_CHECK_ARG(d.x && d.w && d.y && d.tiles, "conv1d: null pointer");
  TTS_CHECK_ARG(d.cin > 0 && d.cout > 0 && d.taps > 0 && d.dil > 0, "conv1d: bad dims");
  TTS_CHECK_ARG(d.cin_pad % 32 == 0 && d.cin_pad >= d.cin, "conv1d: cin_pad %d must be a multiple of 32 >= cin %d", d.cin_pad, d.cin);
  TTS_CHECK_ARG(d.mode >= 0 && d.mode <= 3, "conv1d: bad mode %d", d.mode);
  TTS_CHECK_ARG(d.mode != TTS_MODE_COUPLING || d.aux, "conv1d: coupling mode needs aux");
  TTS_CHECK_ARG(d.pre_act != TTS_PRE_SNAKE || (d.snake_alpha && d.snake_beta && d.snake_filt), "conv1d: PRE_SNAKE needs alpha/beta/filter");
  if (d.n_tiles == 0) return TTS_OK;
  const ConvShape s = pick_shape(d.cout, d.mode);
  int bm, bn;
  shape_dims(s, bm, bn);
  TTS_CHECK_ARG(d.tile_rows == bm, "conv1d: tile table built for %d rows, kernel needs %d", d.tile_rows, bm);
  const int cols = d.mode == TTS_MODE_LINEAR ? d.wn : d.half_pad;
  TTS_CHECK_ARG(cols % bn == 0 && cols >= d.cout, "conv1d: packed width %d not a multiple of the N tile %d (cout %d)", cols, bn, d.cout);
  TTS_CHECK_ARG(d.mode == TTS_MODE_LINEAR || d.wn == 2 * d.half_pad, "conv1d: dual mode needs wn == 2*half_pad");
  switch (s) {
    case S_128x128: return launch_cfg<2, 2, 2, 2, false>(d, st);
    case S_128x96: return launch_cfg<1, 3, 4, 1, false>(d, st);
    case S_128x64: return launch_cfg<2, 1, 2, 2, false>(d, st);
    case S_256x32: return launch_cfg<2, 1, 4, 1, false>(d, st);
    case S_D128x96: return launch_cfg<1, 3, 4, 1, true>(d, st);
  }
  return TTS_E_ARG;
}

int conv1d_tile_rows(int cout, int mode) {
  int bm, bn;
  shape_dims(pick_shape(cout, mode), bm, bn);
  return bm;
}

int conv1d_n_tile(int cout, int mode) {
  int bm, bn;
  shape_dims(pick_shape(cout, mode), bm, bn);
  return bn;
}

}  // namespace tts
