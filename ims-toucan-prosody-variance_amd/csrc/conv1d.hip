// Implicit-GEMM 1-D convolution over packed, time-major activations on the gfx950 matrix cores.
//
//   acc[r, n] = sum_tap sum_ci pre(x[r + tap*dil - pad_left, ci]) * w[tap][ci][n]
//
// Work decomposition: one 256-thread workgroup (4 wavefronts of 64) owns a BM x BN output tile of ONE
// utterance (tile table), so the time halo never crosses an utterance and rows outside the utterance read
// as zero - exactly the reference's per-utterance zero padding.  For every 32-channel slab of the input the
// workgroup stages the (BM + (taps-1)*dil) x 32 activation window into LDS ONCE and reuses it for all taps
// (the window is what makes a k-tap conv cost one activation read instead of k); per tap a weight slab is
// streamed through a double-buffered LDS ring (register prefetch under the MFMAs) and each wavefront issues
// 16 k-steps of v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, so the result matches an fp32 reference to
// rounding-order level).  The bf16 variant (compute == 1) stages 64-channel windows as bf16 and uses
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (weights pre-packed [tap][cin/8][cout][8]).
//
// LDS layout: activations [rows][32+1] floats (odd pitch -> the A-fragment column read, 32 lanes on 32
// consecutive rows, hits 32 distinct banks); weights [32][BN] floats (B-fragment read is a contiguous row).
//
// Tile forms (conv1d_dispatch): 128x128 / 128x96 / 128x64 / 256x32 by output width ("regular"); 64x64 and 64x128 tiles with a
// double-buffered, register-prefetched window when the host asks for 64-row tiles ("small-batch form": grids that would
// leave the chip idle); gemm_rows_kernel, an LDS-free operand stream for 1-tap convs in the small-batch form; and
// conv_splitk_f32_kernel, 32 x 32 tiles with the contraction split over the four wavefronts, for fp32 grids of a few workgroups
// (batch 1).  Every form multiplies TRANSPOSED (the weights are the MFMA A operand), so that a lane's accumulator registers are
// groups of four consecutive output channels of ONE row, and all share the epilogue arithmetic (epilogue_value: bias, per-utterance
// vector, pre-add, activation, GLU / gated / coupling, residual, accumulate, 16-bit I/O) behind conv_epilogue_t / conv_epilogue16_t.
//
// Reference ops replaced: see include/toucan_tts.h (tts_conv1d).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "snake.h"

#ifndef CONV_DIAG
#define CONV_DIAG 0  // diagnostic builds of conv1d_kernel (tools/build_variant.sh NAME -DCONV_DIAG=n): 1 no epilogue, 2 no MFMAs, 3 window staged for slab 0 only, 4 weight slab loaded once, 5 16-bit output wrapped into 1 024 rows, 6 per-workgroup clock stamps (tts_conv_diag_trace)
#endif

namespace tts {

template <int TM, int TN, int WAVES_M, int WAVES_N, bool DUAL>
struct ConvCfg {
  static constexpr int BM = TM * WAVES_M * 32;
  static constexpr int BN = TN * WAVES_N * 32;
};

__device__ __forceinline__ float pre_activation(float v, int pre_act, float slope) {
  if (pre_act == TTS_PRE_LRELU) return v > 0.0f ? v : v * slope;
  return v;
}

__device__ __forceinline__ unsigned short f2bf(float f) { return f32_to_bf16(f); }

// One 4-channel group into the LDS window: the bf16 window takes it as ONE 8-byte store (rows are 16-byte aligned and the group
// starts at a multiple of 4 channels) - four 2-byte stores per group made the staging pass as long as the MFMAs it feeds.
template <bool BF16, bool F16, class T>
__device__ __forceinline__ void store_group(T* dst, float a, float b, float c, float e) {
  if constexpr (BF16) {
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack16<F16>(a, b), pack16<F16>(c, e));
  } else {
    dst[0] = a;
    dst[1] = b;
    dst[2] = c;
    dst[3] = e;
  }
}

// fp32 value -> its two fp16 operands of the split product (compute 3): hi = fp16(v), lo' = fp16((v - hi) 2^11).  v - hi is exact
// in fp32 (hi keeps the leading 11 bits), so hi + 2^-11 lo' carries 22 bits of v; values beyond fp16's range become +-inf (the
// result is then NaN / inf, never a silently wrong number).
__device__ __forceinline__ void split3(float v, float& hi, float& lo) {
  hi = (float)(_Float16)v;
  lo = (v - hi) * 2048.0f;
}
// one 4-channel group into both planes of the split window (lo plane = hi plane + plane_elems)
__device__ __forceinline__ void store_group_x3(unsigned short* dst, size_t plane_elems, float a, float b, float c, float e) {
  float h[4], l[4];
  split3(a, h[0], l[0]);
  split3(b, h[1], l[1]);
  split3(c, h[2], l[2]);
  split3(e, h[3], l[3]);
  *reinterpret_cast<uint2*>(dst) = make_uint2(pack16<true>(h[0], h[1]), pack16<true>(h[2], h[3]));
  *reinterpret_cast<uint2*>(dst + plane_elems) = make_uint2(pack16<true>(l[0], l[1]), pack16<true>(l[2], l[3]));
}

template <bool BF16, bool F16>
struct Elem {
  using T = unsigned short;
  static __device__ __forceinline__ unsigned short cvt(float v) { return to16<F16>(v); }
};
template <bool F16>
struct Elem<false, F16> {
  using T = float;
  static __device__ __forceinline__ float cvt(float v) { return v; }
};

#if CONV_DIAG == 6
__device__ unsigned long long g_conv_trace[4096][4];
}  // namespace tts
extern "C" int tts_conv_diag_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tts::g_conv_trace), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : -1;
}
namespace tts {
#endif

// compile-time loop: f(integral_constant<int, K>) for K = K0 .. N-1
template <int K, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (K < N) {
    f(std::integral_constant<int, K>{});
    static_for<K + 1, N>(f);
  }
}

// One output element of the fused epilogue, up to the store: a / g are the raw accumulators (g: the gate half in the dual modes),
// ba / bg the biases, sv the per-utterance vector, pa / pg the pre-add, ax the coupling input, rv the residual.  Shared by every
// accumulator layout, so all of them round alike.
template <bool DUAL>
__device__ __forceinline__ float epilogue_value(const TtsConvDesc& d, float a, float g, float ba, float bg, float sv, float pa, float pg, float ax,
                                                float rv) {
  float v = a + ba + sv;
  if (d.preadd) v += pa;
  if (DUAL) {
    g += bg;
    if (d.preadd) g += pg;
    if (d.mode == TTS_MODE_GLU) {
      v = v * (1.0f / (1.0f + expf(-g)));
    } else if (d.mode == TTS_MODE_GATED) {
      v = tanhf(v) * (1.0f / (1.0f + expf(-g)));
    } else {  // COUPLING
      v = (ax - v) * expf(-g);
    }
  } else {
    if (d.act == TTS_ACT_RELU) v = fmaxf(v, 0.0f);
    else if (d.act == TTS_ACT_TANH) v = tanhf(v);
  }
  v *= d.alpha;
  if (d.res) v += d.res_scale * rv;
  return v;
}

// ... one element at a time (the layouts whose lanes hold single columns, and the ragged right edge of the others)
template <bool DUAL>
__device__ __forceinline__ void epilogue_element(const TtsConvDesc& d, int row, int n, float a, float g, float ba, float bg, float sv, bool io_f16) {
  const float pa = d.preadd ? d.preadd[(size_t)row * d.ld_preadd + n] : 0.0f;
  const float pg = (DUAL && d.preadd) ? d.preadd[(size_t)row * d.ld_preadd + d.cout + n] : 0.0f;
  const float ax = (DUAL && d.mode == TTS_MODE_COUPLING) ? d.aux[(size_t)row * d.ld_aux + n] : 0.0f;
  float rv = 0.0f;
  if (d.res)
    rv = (d.io_flags & TTS_IO_RES_BF16) ? load16(reinterpret_cast<const unsigned short*>(d.res)[(size_t)row * d.ld_res + n], io_f16)
                                        : d.res[(size_t)row * d.ld_res + n];
  float v = epilogue_value<DUAL>(d, a, g, ba, bg, sv, pa, pg, ax, rv);
  if (d.io_flags & TTS_IO_Y_BF16) {
    unsigned short* yp = reinterpret_cast<unsigned short*>(d.y) + (size_t)row * d.ldy + n;
    if (d.accumulate) v += load16(*yp, io_f16);
    *yp = store16(v, io_f16);
  } else {
    float* yp = d.y + (size_t)row * d.ldy + n;
    if (d.accumulate) v += *yp;
    *yp = v;
  }
}

// Can the epilogue move four consecutive columns of a row at a time?  (every tensor it touches: 16-byte (fp32) / 8-byte (16-bit)
// aligned base, leading dimension and half offset a multiple of four)
__device__ __forceinline__ bool epilogue_vec_ok(const TtsConvDesc& d) {
  auto al = [](const void* p, unsigned m) { return (reinterpret_cast<uintptr_t>(p) & (m - 1)) == 0; };
  bool ok = (d.cout & 3) == 0 && (d.ldy & 3) == 0 && al(d.y, (d.io_flags & TTS_IO_Y_BF16) ? 8 : 16);
  if (d.bias) ok = ok && al(d.bias, 16);
  if (d.seqvec) ok = ok && (d.ld_seqvec & 3) == 0 && al(d.seqvec, 16);
  if (d.preadd) ok = ok && (d.ld_preadd & 3) == 0 && al(d.preadd, 16);
  if (d.res) ok = ok && (d.ld_res & 3) == 0 && al(d.res, (d.io_flags & TTS_IO_RES_BF16) ? 8 : 16);
  if (d.mode == TTS_MODE_COUPLING) ok = ok && (d.ld_aux & 3) == 0 && al(d.aux, 16);
  return ok;
}

// Epilogue of a TRANSPOSED 32x32 accumulator (weights were the MFMA A operand): row of the tile = lane&31, column =
// (reg&3) + 8*(reg>>2) + 4*(lane>>5) - a lane owns four groups of four consecutive output channels of ONE row, so bias, pre-add,
// residual, accumulate and the store move 8 (16-bit tensors) or 16 (fp32) bytes at a time.  (With the output channel on the lane -
// the untransposed layout - a 128 x 128 tile left through 64 two-byte stores per lane: 58 us of a 147 us launch at 256 -> 256
// channels x 3 taps, 122 of 211 us with a residual read the same way, 340 of 609 us in the 128 -> 256 up-sampler; measured with
// -DCONV_DIAG=1.)  A row's pieces are written by one wavefront within a few hundred cycles: L2 merges them into whole lines.
template <int TM, int TN, int NH, bool DUAL>
__device__ __forceinline__ void conv_epilogue_t(const TtsConvDesc& d, const TtsTile& tile, int n0, int wm, int wn, int lrow, int lk,
                                                const f32x16 (&acc)[NH][TM][TN], const float* eb = nullptr, int eb_n = 0) {
  const bool io_f16 = d.io_flags & TTS_IO_F16;  // format of the 16-bit tensors of this call (else bf16)
  const bool vec = epilogue_vec_ok(d);
  const bool y16 = d.io_flags & TTS_IO_Y_BF16, r16 = d.io_flags & TTS_IO_RES_BF16;
  if constexpr (!DUAL) {
    if (vec && !d.preadd && eb) {
      // The common case (plain convs: bias, per-utterance vector, residual, accumulate).  What was measured on the 128 x 128 tile at
      // 256 -> 256 channels (clock stamps, -DCONV_DIAG=6): main loop 41.7 k cycles, epilogue 21.9 k - not the stores (they drain in
      // 0.4 k), but sixteen dependent global-load round trips per lane (bias, vector, residual of each 4-channel group, each ~1.4 k
      // cycles, and behind earlier stores: loads and stores share the in-order vmcnt counter, so a load issued after a store is usable
      // only once that store is acknowledged).  Hence: bias and per-utterance vector come from LDS (`eb`, staged at kernel start), and
      // all residual / accumulate reads of a 32-row block are issued together, in front of its stores: one round trip per block.
      auto col_of = [&](int j, int rq) { return n0 + (wn * TN + j) * 32 + 8 * rq + 4 * lk; };
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = tile.row0 + (wm * TM + i) * 32 + lrow;
        const int rowc = row < tile.seq_end ? row : tile.seq_end - 1;  // (clamped: unconditional loads; rows behind the utterance are never stored)
        constexpr int NG = TN * 4, CH = 8;  // 4-channel groups of the block, and how many of them travel together (registers: 8 per group)
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += CH) {
        uint4 rv[CH], yv[CH];
#pragma unroll
          for (int g = 0; g < CH; ++g) {
            if (g0 + g >= NG) continue;
            const int j = (g0 + g) >> 2, rq = (g0 + g) & 3;
            int n = col_of(j, rq);
            n = n < d.cout ? n : d.cout - 4;
            rv[g] = make_uint4(0, 0, 0, 0);
            yv[g] = make_uint4(0, 0, 0, 0);
            if (d.res) {
              if (r16) {
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(d.res) + (size_t)rowc * d.ld_res + n);
                rv[g].x = u.x; rv[g].y = u.y;
              } else {
                rv[g] = *reinterpret_cast<const uint4*>(d.res + (size_t)rowc * d.ld_res + n);
              }
            }
            if (d.accumulate) {
              if (y16) {
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(d.y) + (size_t)rowc * d.ldy + n);
                yv[g].x = u.x; yv[g].y = u.y;
              } else {
                yv[g] = *reinterpret_cast<const uint4*>(d.y + (size_t)rowc * d.ldy + n);
              }
            }
          }
        __builtin_amdgcn_sched_barrier(0);  // (all loads of the chunk in front of its first store)
        if (row < tile.seq_end) {
#pragma unroll
          for (int g = 0; g < CH; ++g) {
            if (g0 + g >= NG) continue;
            const int j = (g0 + g) >> 2, rq = (g0 + g) & 3;
            const int n = col_of(j, rq);
            if (n >= d.cout) continue;
            const float4 ba = *reinterpret_cast<const float4*>(eb + (n - n0));
            const float4 sv = *reinterpret_cast<const float4*>(eb + eb_n + (n - n0));
            const uint4 ru = rv[g], yu = yv[g];
            float4 r4, y4;
            if (r16) r4 = make_float4(load16(ru.x & 0xFFFF, io_f16), load16(ru.x >> 16, io_f16), load16(ru.y & 0xFFFF, io_f16), load16(ru.y >> 16, io_f16));
            else r4 = make_float4(__builtin_bit_cast(float, ru.x), __builtin_bit_cast(float, ru.y), __builtin_bit_cast(float, ru.z), __builtin_bit_cast(float, ru.w));
            if (y16) y4 = make_float4(load16(yu.x & 0xFFFF, io_f16), load16(yu.x >> 16, io_f16), load16(yu.y & 0xFFFF, io_f16), load16(yu.y >> 16, io_f16));
            else y4 = make_float4(__builtin_bit_cast(float, yu.x), __builtin_bit_cast(float, yu.y), __builtin_bit_cast(float, yu.z), __builtin_bit_cast(float, yu.w));
            float v0 = epilogue_value<false>(d, acc[0][i][j][4 * rq + 0], 0.f, ba.x, 0.f, sv.x, 0.f, 0.f, 0.f, r4.x);
            float v1 = epilogue_value<false>(d, acc[0][i][j][4 * rq + 1], 0.f, ba.y, 0.f, sv.y, 0.f, 0.f, 0.f, r4.y);
            float v2 = epilogue_value<false>(d, acc[0][i][j][4 * rq + 2], 0.f, ba.z, 0.f, sv.z, 0.f, 0.f, 0.f, r4.z);
            float v3 = epilogue_value<false>(d, acc[0][i][j][4 * rq + 3], 0.f, ba.w, 0.f, sv.w, 0.f, 0.f, 0.f, r4.w);
            if (d.accumulate) { v0 += y4.x; v1 += y4.y; v2 += y4.z; v3 += y4.w; }
            if (y16) {
#if CONV_DIAG == 5  // diagnostic: the output lands in 1 024 rows over and over (stays in L2: no HBM write stream)
              uint2* yp = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(d.y) + (size_t)(row & 1023) * d.ldy + n);
#else
              uint2* yp = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(d.y) + (size_t)row * d.ldy + n);
#endif
              *yp = make_uint2((unsigned int)store16(v0, io_f16) | ((unsigned int)store16(v1, io_f16) << 16),
                               (unsigned int)store16(v2, io_f16) | ((unsigned int)store16(v3, io_f16) << 16));
            } else {
              *reinterpret_cast<float4*>(d.y + (size_t)row * d.ldy + n) = make_float4(v0, v1, v2, v3);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = tile.row0 + (wm * TM + i) * 32 + lrow;
    if (row >= tile.seq_end) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n = n0 + (wn * TN + j) * 32 + 8 * rq + 4 * lk;
        if (n >= d.cout) continue;
        if (!vec) {
          for (int q = 0; q < 4 && n + q < d.cout; ++q) {
            const float ba = d.bias ? d.bias[n + q] : 0.0f;
            const float bg = (DUAL && d.bias) ? d.bias[d.cout + n + q] : 0.0f;
            const float sv = d.seqvec ? d.seqvec[(size_t)tile.seq_id * d.ld_seqvec + n + q] : 0.0f;
            epilogue_element<DUAL>(d, row, n + q, acc[0][i][j][4 * rq + q], acc[NH - 1][i][j][4 * rq + q], ba, bg, sv, io_f16);
          }
          continue;
        }
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
        const float4 ba = d.bias ? ld4(d.bias + n) : z4;
        const float4 bg = (DUAL && d.bias) ? ld4(d.bias + d.cout + n) : z4;
        const float4 sv = d.seqvec ? ld4(d.seqvec + (size_t)tile.seq_id * d.ld_seqvec + n) : z4;
        const float4 pa = d.preadd ? ld4(d.preadd + (size_t)row * d.ld_preadd + n) : z4;
        const float4 pg = (DUAL && d.preadd) ? ld4(d.preadd + (size_t)row * d.ld_preadd + d.cout + n) : z4;
        const float4 ax = (DUAL && d.mode == TTS_MODE_COUPLING) ? ld4(d.aux + (size_t)row * d.ld_aux + n) : z4;
        float4 rv = z4;
        if (d.res) {
          if (r16) {
            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(d.res) + (size_t)row * d.ld_res + n);
            rv = make_float4(load16(u.x & 0xFFFF, io_f16), load16(u.x >> 16, io_f16), load16(u.y & 0xFFFF, io_f16), load16(u.y >> 16, io_f16));
          } else {
            rv = ld4(d.res + (size_t)row * d.ld_res + n);
          }
        }
        float v[4];
        v[0] = epilogue_value<DUAL>(d, acc[0][i][j][4 * rq + 0], acc[NH - 1][i][j][4 * rq + 0], ba.x, bg.x, sv.x, pa.x, pg.x, ax.x, rv.x);
        v[1] = epilogue_value<DUAL>(d, acc[0][i][j][4 * rq + 1], acc[NH - 1][i][j][4 * rq + 1], ba.y, bg.y, sv.y, pa.y, pg.y, ax.y, rv.y);
        v[2] = epilogue_value<DUAL>(d, acc[0][i][j][4 * rq + 2], acc[NH - 1][i][j][4 * rq + 2], ba.z, bg.z, sv.z, pa.z, pg.z, ax.z, rv.z);
        v[3] = epilogue_value<DUAL>(d, acc[0][i][j][4 * rq + 3], acc[NH - 1][i][j][4 * rq + 3], ba.w, bg.w, sv.w, pa.w, pg.w, ax.w, rv.w);
        if (y16) {
          uint2* yp = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(d.y) + (size_t)row * d.ldy + n);
          if (d.accumulate) {
            const uint2 u = *yp;
            v[0] += load16(u.x & 0xFFFF, io_f16); v[1] += load16(u.x >> 16, io_f16);
            v[2] += load16(u.y & 0xFFFF, io_f16); v[3] += load16(u.y >> 16, io_f16);
          }
          *yp = make_uint2((unsigned int)store16(v[0], io_f16) | ((unsigned int)store16(v[1], io_f16) << 16),
                           (unsigned int)store16(v[2], io_f16) | ((unsigned int)store16(v[3], io_f16) << 16));
        } else {
          float4* yp = reinterpret_cast<float4*>(d.y + (size_t)row * d.ldy + n);
          if (d.accumulate) {
            const float4 u = *yp;
            v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w;
          }
          *yp = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
}

// Epilogue of a TRANSPOSED 16x16 accumulator (v_mfma_f32_16x16x4_f32 with the weights as the A operand): row = lane&15, columns
// reg + 4*(lane>>4) - four consecutive output channels of one row per lane: every operand is ONE vector load, all of them issued
// before the single vector store (with the output channel on the lane the four elements of a lane were four load - wait - store
// round trips, each load queued behind the previous element's store: ~6 us of a 7-11 us batch-1 launch).
template <int NH, bool DUAL>
__device__ __forceinline__ void conv_epilogue16_t(const TtsConvDesc& d, const TtsTile& tile, int n0, int row_base, int lane, const f32x4 (&acc)[NH]) {
  const bool io_f16 = d.io_flags & TTS_IO_F16;
  const int row = row_base + (lane & 15), n = n0 + 4 * (lane >> 4);
  if (row >= tile.seq_end || n >= d.cout) return;
  if (!epilogue_vec_ok(d)) {
    for (int q = 0; q < 4 && n + q < d.cout; ++q) {
      const float ba = d.bias ? d.bias[n + q] : 0.0f;
      const float bg = (DUAL && d.bias) ? d.bias[d.cout + n + q] : 0.0f;
      const float sv = d.seqvec ? d.seqvec[(size_t)tile.seq_id * d.ld_seqvec + n + q] : 0.0f;
      epilogue_element<DUAL>(d, row, n + q, acc[0][q], acc[NH - 1][q], ba, bg, sv, io_f16);
    }
    return;
  }
  const bool y16 = d.io_flags & TTS_IO_Y_BF16, r16 = d.io_flags & TTS_IO_RES_BF16;
  auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
  auto ld16 = [&](const float* base, size_t off) {  // four 16-bit elements -> fp32
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
    return make_float4(load16(u.x & 0xFFFF, io_f16), load16(u.x >> 16, io_f16), load16(u.y & 0xFFFF, io_f16), load16(u.y >> 16, io_f16));
  };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ba = z4, bg = z4, sv = z4, pa = z4, pg = z4, ax = z4, rv = z4, yv = z4;
  if (d.bias) ba = ld4(d.bias + n);
  if (DUAL && d.bias) bg = ld4(d.bias + d.cout + n);
  if (d.seqvec) sv = ld4(d.seqvec + (size_t)tile.seq_id * d.ld_seqvec + n);
  if (d.preadd) pa = ld4(d.preadd + (size_t)row * d.ld_preadd + n);
  if (DUAL && d.preadd) pg = ld4(d.preadd + (size_t)row * d.ld_preadd + d.cout + n);
  if (DUAL && d.mode == TTS_MODE_COUPLING) ax = ld4(d.aux + (size_t)row * d.ld_aux + n);
  if (d.res) {
    if (r16) rv = ld16(d.res, (size_t)row * d.ld_res + n);
    else rv = ld4(d.res + (size_t)row * d.ld_res + n);
  }
  if (d.accumulate) {
    if (y16) yv = ld16(d.y, (size_t)row * d.ldy + n);
    else yv = ld4(d.y + (size_t)row * d.ldy + n);
  }
  float v0 = epilogue_value<DUAL>(d, acc[0][0], acc[NH - 1][0], ba.x, bg.x, sv.x, pa.x, pg.x, ax.x, rv.x);
  float v1 = epilogue_value<DUAL>(d, acc[0][1], acc[NH - 1][1], ba.y, bg.y, sv.y, pa.y, pg.y, ax.y, rv.y);
  float v2 = epilogue_value<DUAL>(d, acc[0][2], acc[NH - 1][2], ba.z, bg.z, sv.z, pa.z, pg.z, ax.z, rv.z);
  float v3 = epilogue_value<DUAL>(d, acc[0][3], acc[NH - 1][3], ba.w, bg.w, sv.w, pa.w, pg.w, ax.w, rv.w);
  if (d.accumulate) { v0 += yv.x; v1 += yv.y; v2 += yv.z; v3 += yv.w; }
  if (y16) {
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(d.y) + (size_t)row * d.ldy + n) =
        make_uint2((unsigned int)store16(v0, io_f16) | ((unsigned int)store16(v1, io_f16) << 16),
                   (unsigned int)store16(v2, io_f16) | ((unsigned int)store16(v3, io_f16) << 16));
  } else {
    *reinterpret_cast<float4*>(d.y + (size_t)row * d.ldy + n) = make_float4(v0, v1, v2, v3);
  }
}

// Main loop (both precisions): steps s = (channel slab, tap).  The weight slab of step s+1 is fetched into
// registers right after the barrier that opens step s and written to the other LDS buffer after the MFMAs of
// step s, so its global/L2 latency hides under the matrix work; one barrier per step, one more per slab for the
// activation window.  bf16 uses 64-channel slabs (4 k-steps of v_mfma_f32_32x32x16_bf16 per tap), fp32 32-channel
// slabs (16 k-steps of v_mfma_f32_32x32x2_f32); either way a weight slab is BN*16/32 KiB.
// X3 (compute 3): fp32 activations and weights, every product as THREE fp16 matrix instructions on split operands -
//   a b ~= hi_a hi_b + 2^-11 (hi_a lo'_b + lo'_a hi_b),   hi = fp16(v), lo' = fp16((v - hi) 2^11),   fp32 accumulation
// (two accumulator sets: the hi.hi sums and the cross sums, combined once before the epilogue) - ~22 significant bits per product
// at 16/3 of the fp32 matrix rate.  The window and the weight slabs carry two fp16 planes each (weights pre-split by the host:
// w16 = [hi | lo'] planes of [tap][cin_pad/8][wn][8]); everything else - tiling, staging, epilogue - is the 16-bit path's.
// Wavefronts per SIMD the register allocator has to leave room for: three workgroups per CU for the 128 x 128 tile (its pipelined
// epilogue otherwise takes 107 + 64 registers, four over the line); no constraint elsewhere.
constexpr int conv_min_waves(int tm, int tn, bool dual, bool bf16, bool snake, bool x3) {
  return (((tm == 2 && tn == 2) || (tm == 1 && tn == 3)) && !dual && !snake && !x3) ? 3 : 1;  // (128 x 128 and 128 x 96 tiles, both element types)
}

template <int TM, int TN, int WAVES_M, int WAVES_N, bool DUAL, bool BF16, bool SNAKE, bool F16, bool X3 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(conv_min_waves(TM, TN, DUAL, BF16, SNAKE, X3))))
void conv1d_kernel(const TtsConvDesc d) {
  static_assert(BF16 || !F16, "F16 selects the element format of the 16-bit MFMA path");
  static_assert(!X3 || (BF16 && F16 && !SNAKE), "the split fp32 product runs on the fp16 path");
  using C = ConvCfg<TM, TN, WAVES_M, WAVES_N, DUAL>;
  using ET = typename Elem<BF16, F16>::T;
  constexpr int BM = C::BM, BN = C::BN;
  constexpr int NH = DUAL ? 2 : 1;
  constexpr int BK = BF16 ? 64 : 32;                // channels per slab
  // activation pitch (elements): bf16 rows are 16-B aligned and 9 (or 5) 16-B slots long -> conflict-free ds_read_b128;
  // fp32 rows have an odd dword pitch -> conflict-free column reads.  A conv with <= 32 input channels uses half rows.
  const int XP = BF16 ? (d.cin_pad <= 32 ? 40 : BK + 8) : BK + 1;
  constexpr int EPU = BF16 ? 8 : 4;                 // elements per 16-byte unit
  constexpr int UNITS = BK * BN / EPU;              // 16-byte units of one weight slab (per half)
  constexpr int UPT = UNITS / 256;                  // units per thread
  static_assert(UNITS % 256 == 0, "slab must split evenly over the workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
#if CONV_DIAG == 6
  const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime();
#endif

  const TtsTile tile = d.tiles[blockIdx.x];
  const int n0 = blockIdx.y * BN;  // first output column (within a half in dual mode)
  const int halo = (d.taps - 1) * d.dil;
  const int win_rows = BM + halo;
  // The 64-row small-batch form keeps TWO activation windows (a window is only 8-20 KB there): the window of channel slab
  // ch+1 is requested into registers while slab ch multiplies and written to the other buffer afterwards, so a slab boundary
  // no longer exposes a global-load round trip (at batch 1 that round trip, not the MFMAs, set the step time: ~2.5 us/slab).
  constexpr bool WIN2 = BM == 64 && !SNAKE && !X3;
  constexpr int NP = X3 ? 2 : 1;                     // operand planes (hi, lo')
  constexpr int GPR = BK / 4;                        // 4-channel groups per window row
  constexpr int PF = BF16 ? 5 : 3;                   // 16-byte register groups per thread: (64 + 16) rows x GPR / 256, rounded up
  const size_t xs_elems = ((size_t)win_rows * XP + 7) & ~(size_t)7;
  ET* xs0 = reinterpret_cast<ET*>(lds_raw);                                 // [1 or 2][win_rows][XP]  (X3: [2 planes][win_rows][XP])
  ET* ws = xs0 + ((WIN2 || X3) ? 2 : 1) * xs_elems;                         // [2][NH][NP][BK*BN]
  constexpr int WBUF = NH * NP * BK * BN;
  // plain convs: bias and per-utterance vector of the workgroup's BN columns, for the epilogue (visible behind the main loop's barriers)
  float* eb = DUAL ? nullptr : reinterpret_cast<float*>(ws + (size_t)2 * WBUF);  // [2][BN]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int lrow = lane & 31;
  const int lk = lane >> 5;

  f32x16 acc[NH][TM][TN];
  f32x16 accx[X3 ? NH : 1][X3 ? TM : 1][X3 ? TN : 1];  // X3: the cross sums hi.lo' + lo'.hi (scaled by 2^-11 at the end)
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[h][i][j][r] = 0.0f;
          if constexpr (X3) accx[h][i][j][r] = 0.0f;
        }

  if constexpr (!DUAL) {
    if (threadIdx.x < BN) {
      const int n = n0 + threadIdx.x;
      eb[threadIdx.x] = (d.bias && n < d.cout) ? d.bias[n] : 0.0f;
      eb[BN + threadIdx.x] = (d.seqvec && n < d.cout) ? d.seqvec[(size_t)tile.seq_id * d.ld_seqvec + n] : 0.0f;
    }
  }
  const int row_first = tile.row0 - d.pad_left;  // packed row of window row 0
  const bool x_bf16 = d.io_flags & TTS_IO_X_BF16;  // x is a 16-bit tensor ...
  const bool x_f16 = BF16 ? F16 : (d.io_flags & TTS_IO_F16) != 0;  // ... of this format (a 16-bit kernel only meets its own)
  const unsigned short* __restrict__ xh = reinterpret_cast<const unsigned short*>(d.x);  // the input viewed as 16-bit elements
  const bool vec_ok = ((d.ldx & 3) == 0) && ((d.cin & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.x) & (x_bf16 ? 7 : 15)) == 0);
  const bool win2 = WIN2 && vec_ok && win_rows * GPR <= PF * 256;  // else: one window, staged synchronously
  const ET* __restrict__ W = reinterpret_cast<const ET*>(d.w);
  const int n_chunks = (d.cin_pad + BK - 1) / BK;
  const int total_steps = n_chunks * d.taps;

  uint4 wreg[NH][NP][UPT];
  const size_t w_plane = (size_t)d.taps * d.cin_pad * d.wn;  // X3: elements between the hi and the lo' plane of the packed weights
  // 16-byte unit u of the slab of (chunk c0, tap): where it lives in global memory and in the LDS slab
  auto load_slab = [&](int c0, int tap, int kchunk) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
        const int u = tid + q * 256;
        size_t goff;
        bool ok;
        if constexpr (BF16) {
          int kb = u / BN;
          const int n = u % BN;
          ok = kb * 8 < kchunk;
          kb = ok ? kb : 0;  // always a valid address: the load is unconditional (no branch, no early wait), the value is masked
          goff = (((size_t)tap * (d.cin_pad / 8) + c0 / 8 + kb) * d.wn + n0 + h * d.half_pad + n) * 8;
        } else {
          int k = u / (BN / 4);
          const int c4 = (u % (BN / 4)) * 4;
          ok = k < kchunk;
          k = ok ? k : 0;
          goff = ((size_t)tap * d.cin_pad + c0 + k) * d.wn + n0 + h * d.half_pad + c4;
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          const uint4 v = *reinterpret_cast<const uint4*>(W + goff + pl * w_plane);
          wreg[h][pl][q] = ok ? v : make_uint4(0, 0, 0, 0);
        }
      }
  };
  auto store_slab = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
        const int u = tid + q * 256;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          *reinterpret_cast<uint4*>(ws + (size_t)buf * WBUF + (h * NP + pl) * BK * BN + u * EPU) = wreg[h][pl][q];
      }
  };

  uint4 pre[WIN2 ? PF : 1];
  auto win_decode = [&](int e, int c0, int& wr, int& c4, bool& ok, size_t& goff) __attribute__((always_inline)) {
    wr = e / GPR;
    c4 = (e % GPR) * 4;
    const int gr = row_first + wr;
    ok = gr >= tile.seq_begin && gr < tile.seq_end && (c0 + c4) < d.cin;
    const int grc = gr < tile.seq_begin ? tile.seq_begin : (gr >= tile.seq_end ? tile.seq_end - 1 : gr);
    const int cc = (c0 + c4) < d.cin ? (c0 + c4) : d.cin - 4;
    goff = (size_t)grc * d.ldx + cc;
  };
  // (the input-dtype test stays OUTSIDE the load loops: inside, the compiler if-converts it into "issue both loads, wait, select"
  //  per group and the requests serialise - measured ~1100 cycles per slab for five loads)
  auto win_request = [&](int c0) __attribute__((always_inline)) {
    if constexpr (WIN2) {
      const int total = win_rows * GPR;
      size_t goff[PF];
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        int e = tid + q * 256;
        e = e < total ? e : total - 1;  // clamped duplicate: unconditional loads, nothing is written for it
        int wr, c4;
        bool ok;
        win_decode(e, c0, wr, c4, ok, goff[q]);
      }
      if (x_bf16) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
          const uint2 raw = *reinterpret_cast<const uint2*>(xh + goff[q]);
          pre[q].x = raw.x;
          pre[q].y = raw.y;
        }
      } else {
#pragma unroll
        for (int q = 0; q < PF; ++q) pre[q] = *reinterpret_cast<const uint4*>(d.x + goff[q]);
      }
    }
  };
  auto win_commit = [&](int c0, ET* xs) __attribute__((always_inline)) {
    if constexpr (WIN2) {
      const int total = win_rows * GPR;
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int e = tid + q * 256;
        if (e < total) {
          int wr, c4;
          bool ok;
          size_t goff;
          win_decode(e, c0, wr, c4, ok, goff);
          float4 v;
          if (x_bf16)
            v = make_float4(load16(pre[q].x & 0xFFFF, x_f16), load16(pre[q].x >> 16, x_f16), load16(pre[q].y & 0xFFFF, x_f16), load16(pre[q].y >> 16, x_f16));
          else
            v = make_float4(__builtin_bit_cast(float, pre[q].x), __builtin_bit_cast(float, pre[q].y), __builtin_bit_cast(float, pre[q].z),
                            __builtin_bit_cast(float, pre[q].w));
          if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
          ET* dst = xs + wr * XP + c4;
          store_group<BF16, F16>(dst, pre_activation(v.x, d.pre_act, d.pre_slope), pre_activation(v.y, d.pre_act, d.pre_slope),
                            pre_activation(v.z, d.pre_act, d.pre_slope), pre_activation(v.w, d.pre_act, d.pre_slope));
        }
      }
    }
  };
  if (win2) {  // window of slab 0: requested and committed up front
    win_request(0);
    win_commit(0, xs0);
  }

  {
    const int k0 = d.cin_pad < BK ? d.cin_pad : BK;
    load_slab(0, 0, k0);
    store_slab(0);
  }

  int step = 0;
  for (int ch = 0; ch < n_chunks; ++ch) {
    const int c0 = ch * BK;
    const int kchunk = (d.cin_pad - c0) < BK ? (d.cin_pad - c0) : BK;
    ET* xs = xs0 + ((win2 && (ch & 1)) ? xs_elems : 0);
    if (!win2 && ch > 0) __syncthreads();  // the previous slab's MFMAs are done with xs
#if CONV_DIAG == 3
    if (ch == 0) {
#endif
    // ---- stage the activation window: win_rows x kchunk channels, through the input activation ----
    if (win2) {
      // already in LDS (committed at the end of the previous slab); request the next one now, it lands under this slab's MFMAs
      if (ch + 1 < n_chunks) win_request(c0 + BK);
    } else if (SNAKE) {
      // anti-aliased snake computed while staging: item = (8 window rows, channel); the activated tensor never reaches HBM
      float f[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) f[k] = d.snake_filt[k];
      const int T = tile.seq_end - tile.seq_begin;
      const int items = ((win_rows + 7) >> 3) * kchunk;  // kchunk is 32 or 64: no idle lanes on a half slab
      for (int it = tid; it < items; it += 256) {
        const int chl = it % kchunk, wr0 = (it / kchunk) * 8;
        const int cg = c0 + chl;
        const int t0 = row_first + wr0 - tile.seq_begin;  // local frame of the group's first row (may be < 0)
        float o[8];
        const bool live = cg < d.cin && t0 + 7 >= 0 && t0 < T;
        if constexpr (SNAKE)
          if (live) {
            const float ea = expf(d.snake_alpha[cg]), ib = 1.0f / (expf(d.snake_beta[cg]) + 1e-9f);
            if (x_bf16)
              snake_rows_fn<8>([&](int q) { return load16(xh[(size_t)(tile.seq_begin + q) * d.ldx + cg], x_f16); }, T, t0, f, ea, ib, o);
            else
              snake_rows<8>(d.x, d.ldx, cg, tile.seq_begin, T, t0, f, ea, ib, o);
          }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (wr0 + i < win_rows) xs[(wr0 + i) * XP + chl] = Elem<BF16, F16>::cvt((live && t0 + i >= 0 && t0 + i < T) ? o[i] : 0.0f);
      }
    } else if (BF16 && BM != 64 && vec_ok && x_bf16 && (d.cin & 7) == 0 && (d.ldx & 7) == 0 && (reinterpret_cast<uintptr_t>(d.x) & 15) == 0) {
      // 16-bit input of a 16-bit kernel (its own format): 8 channels = 16 bytes per load, and up to six loads per thread in flight -
      // the whole window of a 128-row tile (<= 178 rows x 8 units) in ONE memory round trip instead of three; without a
      // pre-activation the unit goes to LDS as it is
      // (large tiles only: compiled into the 64-row small-batch form too, it made several of its shapes 30 - 60 % slower although
      // they never take this branch)
      if constexpr (BF16 && BM != 64) {
        constexpr int PER8 = 6;
        const int q8 = kchunk >> 3;
        const int total = win_rows * q8;
        for (int base = tid; base < total; base += 256 * PER8) {
          uint4 raw[PER8];
#pragma unroll
          for (int p = 0; p < PER8; ++p) {
            int e = base + p * 256;
            e = e < total ? e : total - 1;
            const int wr = e / q8, c8 = (e % q8) * 8;
            const int gr = row_first + wr;
            const int grc = gr < tile.seq_begin ? tile.seq_begin : (gr >= tile.seq_end ? tile.seq_end - 1 : gr);
            const int cc = (c0 + c8) < d.cin ? (c0 + c8) : d.cin - 8;
            raw[p] = *reinterpret_cast<const uint4*>(xh + (size_t)grc * d.ldx + cc);
          }
#pragma unroll
          for (int p = 0; p < PER8; ++p) {
            const int e = base + p * 256;
            if (e >= total) continue;
            const int wr = e / q8, c8 = (e % q8) * 8;
            const int gr = row_first + wr;
            const bool ok = gr >= tile.seq_begin && gr < tile.seq_end && (c0 + c8) < d.cin;
            uint4 o = raw[p];
            if (d.pre_act != TTS_PRE_NONE) {
              const unsigned int w4[4] = {o.x, o.y, o.z, o.w};
              unsigned int r4[4];
#pragma unroll
              for (int q = 0; q < 4; ++q)
                r4[q] = pack16<F16>(pre_activation(from16<F16>(w4[q] & 0xFFFF), d.pre_act, d.pre_slope),
                                    pre_activation(from16<F16>(w4[q] >> 16), d.pre_act, d.pre_slope));
              o = make_uint4(r4[0], r4[1], r4[2], r4[3]);
            }
            if (!ok) o = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(xs + wr * XP + c8) = o;
          }
        }
      }
    } else if (vec_ok) {
      // PER independent 16-byte loads per thread are issued back to back (clamped addresses, no branches) before any of
      // them is consumed, so one trip pays the memory latency once instead of PER times
      constexpr int PER = 4;
      const int q4 = kchunk >> 2;  // float4 groups per row
      const int total = win_rows * q4;
      for (int base = tid; base < total; base += 256 * PER) {
        uint4 raw[PER];
        size_t goff[PER];
        bool okv[PER];
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          int e = base + p * 256;
          e = e < total ? e : total - 1;  // tail duplicates the last element (same value, benign)
          const int wr = e / q4, c4 = (e % q4) * 4;
          const int gr = row_first + wr;
          okv[p] = gr >= tile.seq_begin && gr < tile.seq_end && (c0 + c4) < d.cin;
          const int grc = gr < tile.seq_begin ? tile.seq_begin : (gr >= tile.seq_end ? tile.seq_end - 1 : gr);
          const int cc = (c0 + c4) < d.cin ? (c0 + c4) : d.cin - 4;
          goff[p] = (size_t)grc * d.ldx + cc;
        }
        if (x_bf16) {  // 4 bf16 = 8 bytes (dtype test outside the load loop, see win_request)
#pragma unroll
          for (int p = 0; p < PER; ++p) {
            const uint2 r2 = *reinterpret_cast<const uint2*>(xh + goff[p]);
            raw[p].x = r2.x;
            raw[p].y = r2.y;
          }
        } else {
#pragma unroll
          for (int p = 0; p < PER; ++p) raw[p] = *reinterpret_cast<const uint4*>(d.x + goff[p]);
        }
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          int e = base + p * 256;
          e = e < total ? e : total - 1;
          const int wr = e / q4, c4 = (e % q4) * 4;
          float4 v;
          if (x_bf16)
            v = make_float4(load16(raw[p].x & 0xFFFF, x_f16), load16(raw[p].x >> 16, x_f16), load16(raw[p].y & 0xFFFF, x_f16), load16(raw[p].y >> 16, x_f16));
          else
            v = make_float4(__builtin_bit_cast(float, raw[p].x), __builtin_bit_cast(float, raw[p].y), __builtin_bit_cast(float, raw[p].z),
                            __builtin_bit_cast(float, raw[p].w));
          if (!okv[p]) v = make_float4(0.f, 0.f, 0.f, 0.f);
          ET* dst = xs + wr * XP + c4;
          if constexpr (X3)
            store_group_x3(dst, xs_elems, pre_activation(v.x, d.pre_act, d.pre_slope), pre_activation(v.y, d.pre_act, d.pre_slope),
                           pre_activation(v.z, d.pre_act, d.pre_slope), pre_activation(v.w, d.pre_act, d.pre_slope));
          else
          store_group<BF16, F16>(dst, pre_activation(v.x, d.pre_act, d.pre_slope), pre_activation(v.y, d.pre_act, d.pre_slope),
                            pre_activation(v.z, d.pre_act, d.pre_slope), pre_activation(v.w, d.pre_act, d.pre_slope));
        }
      }
    } else {
      for (int e = tid; e < win_rows * kchunk; e += 256) {
        const int wr = e / kchunk, c = e % kchunk;
        const int gr = row_first + wr;
        float v = 0.f;
        if (gr >= tile.seq_begin && gr < tile.seq_end && (c0 + c) < d.cin)
          v = x_bf16 ? load16(xh[(size_t)gr * d.ldx + c0 + c], x_f16) : d.x[(size_t)gr * d.ldx + c0 + c];
        if constexpr (X3) {
          float hi, lo;
          split3(pre_activation(v, d.pre_act, d.pre_slope), hi, lo);
          xs[wr * XP + c] = f32_to_f16(hi);
          xs[xs_elems + wr * XP + c] = f32_to_f16(lo);
        } else {
        xs[wr * XP + c] = Elem<BF16, F16>::cvt(pre_activation(v, d.pre_act, d.pre_slope));
        }
      }
    }
#if CONV_DIAG == 3
    }
#endif
    for (int tap = 0; tap < d.taps; ++tap, ++step) {
      __syncthreads();  // xs and ws[step&1] visible; everybody is done reading ws[(step+1)&1]
      const bool more = step + 1 < total_steps;
#if CONV_DIAG == 4
      if (false) {
#else
      if (more) {
#endif
        const int ntap = tap + 1 < d.taps ? tap + 1 : 0;
        const int nc0 = tap + 1 < d.taps ? c0 : c0 + BK;
        const int nk = (d.cin_pad - nc0) < BK ? (d.cin_pad - nc0) : BK;
        load_slab(nc0, ntap, nk);
      }
      const ET* wb = ws + (size_t)(step & 1) * WBUF;
      if constexpr (BF16) {
        const int ksteps = kchunk >> 4;
        for (int ks = 0; ks < ksteps; ++ks) {
          bf16x8 a[TM], al[X3 ? TM : 1];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            a[i] = *reinterpret_cast<const bf16x8*>(xs + (wm * TM * 32 + i * 32 + lrow + tap * d.dil) * XP + ks * 16 + lk * 8);
            if constexpr (X3) al[i] = *reinterpret_cast<const bf16x8*>(xs + xs_elems + (wm * TM * 32 + i * 32 + lrow + tap * d.dil) * XP + ks * 16 + lk * 8);
          }
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            bf16x8 b[TN], bl[X3 ? TN : 1];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              b[j] = *reinterpret_cast<const bf16x8*>(wb + h * NP * BK * BN + ((ks * 2 + lk) * BN + wn * TN * 32 + j * 32 + lrow) * 8);
              if constexpr (X3) bl[j] = *reinterpret_cast<const bf16x8*>(wb + (h * NP + 1) * BK * BN + ((ks * 2 + lk) * BN + wn * TN * 32 + j * 32 + lrow) * 8);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
#if CONV_DIAG == 2
                asm volatile("" ::"v"(a[i]), "v"(b[j]));
#else
                acc[h][i][j] = mfma16<F16>(b[j], a[i], acc[h][i][j]);  // transposed: weights = A operand, the lane keeps a ROW (conv_epilogue_t)
#endif
                if constexpr (X3) {
                  accx[h][i][j] = mfma16<true>(bl[j], a[i], accx[h][i][j]);
                  accx[h][i][j] = mfma16<true>(b[j], al[i], accx[h][i][j]);
                }
              }
          }
        }
      } else {
        const float* xa = xs + (wm * TM * 32 + lrow + tap * d.dil) * XP + lk;
        const float* wbb = wb + lk * BN + wn * TN * 32 + lrow;
        const int kpairs = kchunk >> 1;
#pragma unroll 4
        for (int kk = 0; kk < kpairs; ++kk) {
          float a[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) a[i] = xa[i * 32 * XP + 2 * kk];
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            float b[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = wbb[h * BK * BN + 2 * kk * BN + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[h][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[h][i][j], 0, 0, 0);
          }
        }
      }
      if (more) store_slab((step + 1) & 1);
    }
    // the other window buffer was last read one slab ago and every wave has passed a barrier since: safe to overwrite;
    // the first barrier of the next slab publishes it
    if (win2 && ch + 1 < n_chunks) win_commit(c0 + BK, xs0 + ((ch & 1) ? 0 : xs_elems));
  }

  if constexpr (X3) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[h][i][j][r] = fmaf(accx[h][i][j][r], 1.0f / 2048.0f, acc[h][i][j][r]);
  }
#if CONV_DIAG == 6  // diagnostic: per-workgroup clocks (start, main loop done, epilogue issued, epilogue drained) of wavefront 0
  const unsigned long long diag_t1 = __builtin_amdgcn_s_memtime();
  conv_epilogue_t<TM, TN, NH, DUAL>(d, tile, n0, wm, wn, lrow, lk, acc, eb, BN);
  const unsigned long long diag_t2 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long diag_t3 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    const unsigned int slot = (blockIdx.y * gridDim.x + blockIdx.x) & 4095;
    g_conv_trace[slot][0] = diag_t0; g_conv_trace[slot][1] = diag_t1; g_conv_trace[slot][2] = diag_t2; g_conv_trace[slot][3] = diag_t3;
  }
#elif CONV_DIAG == 1
  {  // diagnostic build: no epilogue (one never-taken store keeps the accumulators alive)
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[0][i][j][r];
    if (t == 1.2345e-30f) d.y[0] = t;
  }
#else
  conv_epilogue_t<TM, TN, NH, DUAL>(d, tile, n0, wm, wn, lrow, lk, acc, eb, BN);
#endif
}

// ------------------------------------------------------------------------------------------------
// 1-tap convs (linears) in the small-batch form without LDS: every wavefront streams its own A rows and B columns straight
// from global memory / L2 into registers, DEPTH k-steps ahead, and never meets a barrier.  With a single tap there is no
// window to share between taps, and at 1-4 workgroups per CU the LDS-staged loop above is a chain of exposed round trips
// (barrier, slab load, window load: ~1.6 us per 64-channel step whatever the MFMA work).  Same 64 x 64 workgroup tile and
// wave layout as the small form, same packed weights, same epilogue.
//   bf16: A fragment = 8 consecutive channels of the lane's row (two 16-byte loads of fp32, or one of bf16, converted in
//         registers), B fragment = one 16-byte load from the [cin/8][wn][8] packing.
//   fp32: v_mfma_f32_32x32x2_f32 takes k = lane>>5 of a k-pair; the k order inside a group of 8 channels is permuted so that
//         one float4 of A per lane feeds four MFMAs (MFMA j of group g contracts channels 8g + j and 8g + 4 + j).
// ------------------------------------------------------------------------------------------------
template <bool DUAL, bool BF16, bool XB, bool F16>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const TtsConvDesc d) {
  constexpr int NH = DUAL ? 2 : 1;
  constexpr int DEPTH = 4;  // k-steps (bf16: 16 channels, fp32: 8 channels) in flight per wavefront
  const TtsTile tile = d.tiles[blockIdx.x];
  const int n0 = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, lrow = lane & 31, lk = lane >> 5;
  int row = tile.row0 + wm * 32 + lrow;
  row = row < tile.seq_end ? row : tile.seq_end - 1;  // rows past the utterance load a valid row, the epilogue drops them
  const int col = n0 + wn * 32 + lrow;

  f32x16 acc[NH][1][1];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[h][0][0][r] = 0.0f;

  // The operand stream is written with inline-asm loads and hand-counted waits: the loads return in order, so before k-step s
  // is consumed exactly (DEPTH-1) * L younger loads may stay in flight (L = loads per k-step).  Left to the compiler the same
  // loop either had its 16-byte loads split into dwords (the fragments are consumed lane-element by lane-element) or drained
  // every outstanding load at the loop head (it cannot count across the back edge); the asm is invisible to its counters, and
  // each wait is followed by empty asm statements that "redefine" the registers just waited for, so no use can move above it.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  // (by value: __builtin_bit_cast applied directly to an element of an ext-vector reads element 0 whatever the index)
  auto u2f = [](unsigned int bits) __attribute__((always_inline)) { return __builtin_bit_cast(float, bits); };
#define TTS_GLOAD128(dst_, ptr_) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst_) : "v"(ptr_) : "memory")
#define TTS_GLOAD32(dst_, ptr_) asm volatile("global_load_dword %0, %1, off" : "=v"(dst_) : "v"(ptr_) : "memory")
#define TTS_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define TTS_PIN(r_) asm volatile("" : "+v"(r_))
  if constexpr (BF16) {
    const int n_steps = d.cin_pad >> 4;  // multiple of DEPTH (dispatch guarantees cin % 64 == 0)
    constexpr int L = (XB ? 1 : 2) + NH;
    // A: 8 channels of this lane's row per k-step (fp32: two 16-byte loads, bf16: one); B: one 16-byte unit of [cin/8][wn][8]
    const char* xp = reinterpret_cast<const char*>(d.x) + ((size_t)row * d.ldx + lk * 8) * (XB ? 2 : 4);
    const size_t xstep = XB ? 32 : 64;
    const char* wp = reinterpret_cast<const char*>(d.w) + ((size_t)lk * d.wn + col) * 16;
    const size_t wstep = (size_t)d.wn * 32, whalf = (size_t)d.half_pad * 16;
    u32x4 a0[DEPTH], a1[DEPTH], b[NH][DEPTH];
    auto request = [&](int slot, int ks) __attribute__((always_inline)) {
      TTS_GLOAD128(a0[slot], xp + ks * xstep);
      if constexpr (!XB) TTS_GLOAD128(a1[slot], xp + ks * xstep + 16);
#pragma unroll
      for (int h = 0; h < NH; ++h) TTS_GLOAD128(b[h][slot], wp + ks * wstep + h * whalf);
    };
    auto arrive = [&](int slot) __attribute__((always_inline)) {
      TTS_WAIT_VM((DEPTH - 1) * L);
      TTS_PIN(a0[slot]);
      if constexpr (!XB) TTS_PIN(a1[slot]);
#pragma unroll
      for (int h = 0; h < NH; ++h) TTS_PIN(b[h][slot]);
    };
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) request(u, u);
    for (int base = 0; base < n_steps; base += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        arrive(u);
        u32x4 ap;  // 8 bf16 channels of this lane's row
        if constexpr (XB) {
          ap = a0[u];
          if (d.pre_act == TTS_PRE_LRELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              ap[q] = pack16<F16>(pre_activation(from16<F16>(a0[u][q] & 0xFFFF), d.pre_act, d.pre_slope),
                                  pre_activation(from16<F16>(a0[u][q] >> 16), d.pre_act, d.pre_slope));
          }
        } else {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            ap[q] = pack16<F16>(pre_activation(u2f(a0[u][2 * q]), d.pre_act, d.pre_slope), pre_activation(u2f(a0[u][2 * q + 1]), d.pre_act, d.pre_slope));
            ap[2 + q] = pack16<F16>(pre_activation(u2f(a1[u][2 * q]), d.pre_act, d.pre_slope), pre_activation(u2f(a1[u][2 * q + 1]), d.pre_act, d.pre_slope));
          }
        }
        u32x4 bf[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) bf[h] = b[h][u];  // copies: the slot is re-requested below while the MFMA may still read
#pragma unroll
        for (int h = 0; h < NH; ++h)
          acc[h][0][0] = mfma16<F16>(__builtin_bit_cast(bf16x8, bf[h]), __builtin_bit_cast(bf16x8, ap), acc[h][0][0]);  // (transposed, as in conv1d_kernel)
        int nxt = base + u + DEPTH;
        nxt = nxt < n_steps ? nxt : n_steps - 1;  // the tail re-requests the last step (unused): no branch in the stream
        request(u, nxt);
      }
    }
    TTS_WAIT_VM(0);  // drain the tail requests: their destination registers stay allocated (pinned) until here
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      TTS_PIN(a0[u]);
      if constexpr (!XB) TTS_PIN(a1[u]);
#pragma unroll
      for (int h = 0; h < NH; ++h) TTS_PIN(b[h][u]);
    }
  } else {
    const int n_groups = d.cin_pad >> 3;  // groups of 8 channels; multiple of DEPTH (cin % 32 == 0)
    constexpr int L = 1 + 4 * NH;
    // A: float4 at channels 8 g + 4 lk; B: rows 8 g + 4 lk + j of the [cin][wn] fp32 weights, column col
    const char* xp = reinterpret_cast<const char*>(d.x) + ((size_t)row * d.ldx + lk * 4) * 4;
    const char* wp = reinterpret_cast<const char*>(d.w) + ((size_t)(lk * 4) * d.wn + col) * 4;
    const size_t wrow = (size_t)d.wn * 4, whalf = (size_t)d.half_pad * 4;
    u32x4 a[DEPTH];
    unsigned int b[NH][DEPTH][4];
    auto request = [&](int slot, int g) __attribute__((always_inline)) {
      TTS_GLOAD128(a[slot], xp + (size_t)g * 32);
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) TTS_GLOAD32(b[h][slot][j], wp + ((size_t)g * 8 + j) * wrow + h * whalf);
    };
    auto arrive = [&](int slot) __attribute__((always_inline)) {
      TTS_WAIT_VM((DEPTH - 1) * L);
      TTS_PIN(a[slot]);
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) TTS_PIN(b[h][slot][j]);
    };
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) request(u, u);
    for (int base = 0; base < n_groups; base += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        arrive(u);
        float av[4], bv[NH][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          av[j] = pre_activation(u2f(a[u][j]), d.pre_act, d.pre_slope);
#pragma unroll
          for (int h = 0; h < NH; ++h) bv[h][j] = u2f(b[h][u][j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h = 0; h < NH; ++h) acc[h][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[h][j], av[j], acc[h][0][0], 0, 0, 0);
        int nxt = base + u + DEPTH;
        nxt = nxt < n_groups ? nxt : n_groups - 1;
        request(u, nxt);
      }
    }
    TTS_WAIT_VM(0);
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      TTS_PIN(a[u]);
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) TTS_PIN(b[h][u][j]);
    }
  }
#undef TTS_GLOAD128
#undef TTS_GLOAD32
#undef TTS_WAIT_VM
#undef TTS_PIN
  conv_epilogue_t<1, 1, NH, DUAL>(d, tile, n0, wm, wn, lrow, lk, acc);
}

// ------------------------------------------------------------------------------------------------
// Split-K form for fp32 convs whose small-batch grid is still a handful of workgroups (batch 1: the WaveNet in-layer of the
// flow is 15 workgroups of the 64 x 64 form, and each of its wavefronts then runs 960 dependent v_mfma_f32_32x32x2_f32 -
// 61 k cycles - alone on its SIMD while 240 CUs idle).  Here a workgroup owns a 32 x 32 output tile (both halves in the dual
// modes) and its four wavefronts split the contraction: k-step s = (tap, group of 16 channels) goes to wavefront s % 4, every
// wavefront streams its own operands LDS-free like gemm_rows_kernel (any number of taps: a tap is a row offset, rows outside
// the utterance contribute zero), the three partial tiles meet in LDS and wavefront 0 runs the shared epilogue.  16x the
// wavefronts of the 64 x 64 form on the same work; T16 (v_mfma_f32_16x16x4_f32, 16 x 16 tiles) 64x, for the grids that are
// still under one workgroup per two CUs in the 32 x 32 form (measured: the fp32 matrix instructions of one wavefront, not the
// operand stream, set the time - ~50 ns per 32x32x2 - so the only lever is more wavefronts).
// ONE accumulation order for both tile sizes: an output element is the sum of four partial sums (wavefront w: the k-steps
// w, w + 4, ...), each a chain of fused multiply-adds over its steps' channels in the order c, c + 4, c + 8, c + 12 for
// c = 0 .. 3 of a 16-channel group - which is what one 16x16x4 instruction per c does, and what two 32x32x2 instructions per c
// do (channels (c, c + 4), then (c + 8, c + 12): the matrix instructions are k-ordered fma chains).  So a launch's result
// does not depend on which of the two the grid heuristics pick (tests/test_gpu_kernels.py asserts bit equality).  It does
// differ from the tiled forms (one chain over all channels), so a caller opts in per launch: TTS_IO_SPLIT_K takes this form on
// small grids only (the frame stages of the fp32 acoustic model: results agree with the tiled forms to rounding-order level,
// which is what the fp32 configuration promises for the mel); TTS_IO_SPLIT_K_ALWAYS takes it at EVERY grid size (the
// phoneme stages of the fp32 acoustic model: the duration predictor rounds exp(log d) to integers, so everything upstream of it
// keeps one arithmetic whatever the batch - an utterance's frame count, pitch and energy are bit for bit the same at B = 1, in
// a batch of 32 and in an N-rank shard); the vocoder never does (chunked == whole, bit for bit).
// ------------------------------------------------------------------------------------------------
template <bool DUAL, bool T16>
__global__ __launch_bounds__(256) void conv_splitk_f32_kernel(const TtsConvDesc d) {
  constexpr int NH = DUAL ? 2 : 1;
  constexpr int T = T16 ? 16 : 32;   // rows and columns of the workgroup's output tile
  constexpr int GC = 16;             // channels per k-step (both tile sizes: one accumulation order, see above)
  constexpr int NA = T16 ? 1 : 2;    // float4 activation loads per lane and k-step (T16: 4 k-slots x 4 channels; 32 x 32: 2 k-slots x 4 channels, twice)
  constexpr int NB = 4 * NA;         // weight dwords per lane, half and k-step
  constexpr int AR = T16 ? 4 : 16;   // accumulator registers per lane
#ifndef TTS_SPLITK_DEPTH
#define TTS_SPLITK_DEPTH 4
#endif
  constexpr int DEPTH = T16 ? TTS_SPLITK_DEPTH : TTS_SPLITK_DEPTH / 2;  // k-steps in flight per wavefront (the same bytes either way; 4, 6, 8 measured alike)
  constexpr int L = NA + NB * NH;    // loads per k-step
  static_assert((DEPTH - 1) * L <= 63, "vmcnt is a 6-bit count");
  using Acc = typename std::conditional<T16, f32x4, f32x16>::type;
  __shared__ float part[3][NH][AR][64];
  const int subs = d.tile_rows / T;  // row blocks of this kernel per tile of the table (any of the table's forms: 64, 128, 256 rows)
  const TtsTile tile = d.tiles[blockIdx.x / subs];
  const int sub = blockIdx.x % subs;
  if (tile.row0 + sub * T >= tile.seq_end) return;  // (the whole workgroup: nothing to write)
  const int n0 = blockIdx.y * T;
  const int tid = threadIdx.x, lane = tid & 63, lrow = lane & (T - 1), kq = lane / T;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int orow = tile.row0 + sub * T + lrow;  // this lane's A row (before the tap offset)
  const int col = n0 + lrow;

  Acc acc[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int r = 0; r < AR; ++r) acc[h][r] = 0.0f;

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  auto u2f = [](unsigned int bits) __attribute__((always_inline)) { return __builtin_bit_cast(float, bits); };
#define TTS_GLOAD128(dst_, ptr_) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst_) : "v"(ptr_) : "memory")
#define TTS_GLOAD32(dst_, ptr_) asm volatile("global_load_dword %0, %1, off" : "=v"(dst_) : "v"(ptr_) : "memory")
#define TTS_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define TTS_PIN(r_) asm volatile("" : "+v"(r_))
  const int G = d.cin / GC;                       // k-steps per tap (dispatch: cin % 16 == 0, cin >= 32)
  const int S = d.taps * G;                       // k-steps of the whole contraction
  const int n_w = S > wave ? (S - wave + 3) >> 2 : 0;  // ... of this wavefront: s = wave, wave + 4, ...
  int tap_r = wave / G, g_r = wave % G;           // the next step to request (G may be smaller than 4)
  const size_t wrow = (size_t)d.wn * 4, whalf = (size_t)d.half_pad * 4;
  u32x4 a[DEPTH][NA];
  unsigned int b[NH][DEPTH][NB];
  bool inside[DEPTH];
  // requests the step (tap_r, g_r) and moves on; past the end it re-requests the last step of the contraction with `inside`
  // false, i.e. as a row of zeros: the stream stays branch-free, every wait count exact, and the matrix instructions
  // unconditional (a branch around them makes the compiler shuttle the accumulators between AGPRs and VGPRs at every step)
  auto request = [&](int slot) __attribute__((always_inline)) {
    const bool live = tap_r < d.taps;
    const int tp = live ? tap_r : d.taps - 1, gg = live ? g_r : G - 1;
    int r = orow + tp * d.dil - d.pad_left;
    inside[slot] = live && r >= tile.seq_begin && r < tile.seq_end;
    r = r < tile.seq_begin ? tile.seq_begin : (r >= tile.seq_end ? tile.seq_end - 1 : r);
    const char* xp = reinterpret_cast<const char*>(d.x) + ((size_t)r * d.ldx + gg * GC + kq * 4) * 4;
    const char* wp = reinterpret_cast<const char*>(d.w) + (((size_t)tp * d.cin_pad + gg * GC + kq * 4) * d.wn + col) * 4;
#pragma unroll
    for (int q = 0; q < NA; ++q) TTS_GLOAD128(a[slot][q], xp + q * 32);  // (32 x 32: the second float4 holds the channels 8 further on)
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) TTS_GLOAD32(b[h][slot][4 * q + j], wp + (8 * q + j) * wrow + h * whalf);
    if (live) {
      g_r += 4;
      while (g_r >= G) { g_r -= G; ++tap_r; }
    }
  };
  auto arrive = [&](int slot) __attribute__((always_inline)) {
    TTS_WAIT_VM((DEPTH - 1) * L);
#pragma unroll
    for (int q = 0; q < NA; ++q) TTS_PIN(a[slot][q]);
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < NB; ++j) TTS_PIN(b[h][slot][j]);
  };
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) request(u);
  for (int base = 0; base < n_w; base += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      arrive(u);
      float av[NA][4], bv[NH][NB];
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) av[q][j] = inside[u] ? pre_activation(u2f(a[u][q][j]), d.pre_act, d.pre_slope) : 0.0f;
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int j = 0; j < NB; ++j) bv[h][j] = u2f(b[h][u][j]);
      // channel c = j of the group: T16 contracts c, c + 4, c + 8, c + 12 in one instruction (k-slot q = lane / 16 supplies c + 4 q);
      // the 32 x 32 form in two (k-slots supply c, c + 4 - then c + 8, c + 12 from the second load): the same fma chain
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < NA; ++q)
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            // (transposed: the weights are the A operand, so a lane ends up with consecutive channels of ONE row - vector epilogue)
            if constexpr (T16) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[h][4 * q + j], av[q][j], acc[h], 0, 0, 0);
            else acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[h][4 * q + j], av[q][j], acc[h], 0, 0, 0);
          }
      request(u);
    }
  }
  TTS_WAIT_VM(0);  // drain the tail requests: their destination registers stay allocated (pinned) until here
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) {
#pragma unroll
    for (int q = 0; q < NA; ++q) TTS_PIN(a[u][q]);
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int j = 0; j < NB; ++j) TTS_PIN(b[h][u][j]);
  }
#undef TTS_GLOAD128
#undef TTS_GLOAD32
#undef TTS_WAIT_VM
#undef TTS_PIN
  if (wave != 0) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int r = 0; r < AR; ++r) part[wave - 1][h][r][lane] = acc[h][r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int r = 0; r < AR; ++r) acc[h][r] = ((acc[h][r] + part[0][h][r][lane]) + part[1][h][r][lane]) + part[2][h][r][lane];
  if constexpr (T16) {
    conv_epilogue16_t<NH, DUAL>(d, tile, n0, tile.row0 + sub * 16, lane, acc);
  } else {
    f32x16 acc32[NH][1][1];
#pragma unroll
    for (int h = 0; h < NH; ++h) acc32[h][0][0] = acc[h];
    conv_epilogue_t<1, 1, NH, DUAL>(d, tile, n0, sub, 0, lrow, kq, acc32);
  }
}

// ------------------------------------------------------------------------------------------------
// shape -> tile configuration
// ------------------------------------------------------------------------------------------------
enum ConvShape { S_128x128, S_128x96, S_128x64, S_256x32, S_D128x96, S_64x64, S_D64x64, S_64x128 };

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
    case S_64x64: bm = 64; bn = 64; break;
    case S_D64x64: bm = 64; bn = 64; break;
    case S_64x128: bm = 64; bn = 128; break;
  }
}

// Small-batch form: 64 x 64 tiles (four wavefronts, 2 x 2) put 3-4x more workgroups on the chip when the 128-row
// grid would leave most CUs idle; the host asks for it by building the tile table with 64 rows.  Needs the packed
// width (a multiple of 96 or 128 by construction) to also be a multiple of 64.
static bool small_form_ok(int cout, int mode, int cols) {
  if (cols % 64 != 0) return false;
  return mode != TTS_MODE_LINEAR || cout > 64;
}

template <int TM, int TN, int WAVES_M, int WAVES_N, bool DUAL, bool BF16, bool SNAKE, bool F16 = false, bool X3 = false>
static int launch_one(const TtsConvDesc& d, hipStream_t st) {
  using C = ConvCfg<TM, TN, WAVES_M, WAVES_N, DUAL>;
  constexpr int BK = BF16 ? 64 : 32, ESZ = BF16 ? 2 : 4, NH = DUAL ? 2 : 1;
  const int XP = BF16 ? (d.cin_pad <= 32 ? 40 : BK + 8) : BK + 1;
  const int halo = (d.taps - 1) * d.dil;
  const int win_rows = C::BM + halo;
  const int n_tiles_n = ((DUAL ? d.half_pad : d.wn) + C::BN - 1) / C::BN;
  dim3 grid(d.n_tiles, n_tiles_n), block(256);
  const size_t xs_elems = ((size_t)win_rows * XP + 7) & ~(size_t)7;
  const size_t lds = (((C::BM == 64 && !SNAKE) || X3 ? 2 : 1) * xs_elems + (size_t)2 * NH * (X3 ? 2 : 1) * BK * C::BN) * ESZ +
                     (DUAL ? 0 : 2 * C::BN * sizeof(float));  // (+ the epilogue's bias / per-utterance vector)
  TTS_CHECK_ARG(lds <= 160 * 1024, "conv1d: LDS %zu B exceeds 160 KiB (taps %d dil %d)", lds, d.taps, d.dil);
  auto k = conv1d_kernel<TM, TN, WAVES_M, WAVES_N, DUAL, BF16, SNAKE, F16, X3>;
  static unsigned long long lds_raised = 0;  // devices on which this instantiation's limit is already raised
  if (lds > 64 * 1024) {
    const hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(k), lds_raised);
    if (e != hipSuccess) {
      set_error("conv1d: raising the dynamic LDS limit failed: %s", hipGetErrorString(e));
      return TTS_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(k, grid, block, lds, st, d);
  return launch_status("conv1d");
}

template <int TM, int TN, int WAVES_M, int WAVES_N, bool DUAL>
static int launch_cfg(const TtsConvDesc& d, hipStream_t st) {
  if constexpr (!DUAL) {
    TTS_CHECK_ARG(!(d.pre_act == TTS_PRE_SNAKE && d.compute == 3), "conv1d: the snake prologue is not available with the split fp32 product");
    if (d.pre_act == TTS_PRE_SNAKE) {  // only the (non-dual) vocoder convs carry the snake prologue
      if (d.compute == 0) return launch_one<TM, TN, WAVES_M, WAVES_N, DUAL, false, true>(d, st);
      if (d.compute == 2) return launch_one<TM, TN, WAVES_M, WAVES_N, DUAL, true, true, true>(d, st);
      return launch_one<TM, TN, WAVES_M, WAVES_N, DUAL, true, true>(d, st);
    }
  }
  TTS_CHECK_ARG(d.pre_act != TTS_PRE_SNAKE, "conv1d: the snake prologue is not available in the dual modes");
  if (d.compute == 3) return launch_one<TM, TN, WAVES_M, WAVES_N, DUAL, true, false, true, true>(d, st);
  if (d.compute == 0) return launch_one<TM, TN, WAVES_M, WAVES_N, DUAL, false, false>(d, st);
  if (d.compute == 2) return launch_one<TM, TN, WAVES_M, WAVES_N, DUAL, true, false, true>(d, st);
  return launch_one<TM, TN, WAVES_M, WAVES_N, DUAL, true, false>(d, st);
}

// the LDS-free 1-tap path: single tap, no padding, whole 64-channel (bf16) / 32-channel (fp32) slabs, 16-byte aligned rows
static bool gemm_rows_ok(const TtsConvDesc& d) {
  if (d.compute == 3) return false;  // (the split fp32 product has the LDS-staged forms only)
  if (d.taps != 1 || d.pad_left != 0 || d.pre_act == TTS_PRE_SNAKE || d.cin != d.cin_pad) return false;
  const bool xb = d.io_flags & TTS_IO_X_BF16;
  if (xb && d.compute == 0) return false;
  if (d.cin % (d.compute ? 64 : 32) != 0) return false;
  // Measured on MI355X (tools/microbench_small.py): fp32 gains 25-40 % at every size (K = 192: 15.5 -> 11.5 us, K = 1536:
  // 78 -> 48 us; its MFMAs are 8x longer per byte loaded).  bf16 gains at latency-bound grid sizes (K = 1536, 30 workgroups:
  // 39 -> 29 us; K = 192: 11.5 -> 9.5 us) and is neutral once the grid fills the chip, where the LDS-staged form shares each
  // operand between two wavefronts (crossover measured at ~one workgroup per CU: K = 192 x N = 576 loses from 360
  // workgroups on, K = 1536 x N = 192 from 384) - so bf16 takes this path for grids up to 256 workgroups only.
  if (d.compute != 0 && std::getenv("TOUCAN_GEMM_ROWS_BF16") == nullptr) {
    const int cols = d.mode != TTS_MODE_LINEAR ? d.half_pad : d.wn;
    if ((long long)d.n_tiles * (cols / 64) > 256) return false;
  }
  if ((d.ldx & (xb ? 7 : 3)) != 0 || (reinterpret_cast<uintptr_t>(d.x) & 15) != 0) return false;
  return std::getenv("TOUCAN_NO_GEMM_ROWS") == nullptr;  // escape hatch for A/B measurements
}

// the split-K form: fp32, opted in by the caller.  TTS_IO_SPLIT_K: a grid of at most 128 workgroups if cut into 64 x 64 tiles and a
// contraction of at least 64 products per output (measured at batch 1 x 128 phonemes, the whole acoustic pass: 6.4 ms with a
// minimum depth of 256, 5.2 ms with 128, 5.0 ms with 64 - even the 192-deep 1-tap convs are faster on 4-16x the wavefronts).
// TTS_IO_SPLIT_K_ALWAYS: whatever the grid (the eligibility below then depends on the conv alone, never on the batch).
static bool splitk_ok(const TtsConvDesc& d, int cols) {
  if (d.compute != 0 || !(d.io_flags & (TTS_IO_SPLIT_K | TTS_IO_SPLIT_K_ALWAYS)) || (d.io_flags & TTS_IO_X_BF16) || d.pre_act == TTS_PRE_SNAKE) return false;
  if ((d.cin & 15) != 0 || d.cin < 32 || (d.ldx & 3) != 0 || (reinterpret_cast<uintptr_t>(d.x) & 15) != 0 || (cols & 31) != 0) return false;
  static const int min_depth = std::getenv("TOUCAN_SPLIT_K_MIN") ? std::atoi(std::getenv("TOUCAN_SPLIT_K_MIN")) : 64;  // (A/B runs)
  static const int max_grid = std::getenv("TOUCAN_SPLIT_K_GRID") ? std::atoi(std::getenv("TOUCAN_SPLIT_K_GRID")) : 128;  // (A/B runs)
  if ((long long)d.taps * d.cin < min_depth) return false;
  if (!(d.io_flags & TTS_IO_SPLIT_K_ALWAYS) && (long long)d.n_tiles * (d.tile_rows / 64) * ((cols + 63) / 64) > max_grid) return false;
  return std::getenv("TOUCAN_NO_SPLIT_K") == nullptr;  // escape hatch for A/B measurements
}

static int launch_splitk(const TtsConvDesc& d, int cols, hipStream_t st) {
  const bool dual = d.mode != TTS_MODE_LINEAR;
  // 16 x 16 tiles while the 32 x 32 grid is at most 128 workgroups (a speed choice only: both sizes sum in the same order)
  if (d.cin >= 64 && (long long)d.n_tiles * (d.tile_rows / 32) * (cols / 32) <= (std::getenv("TOUCAN_SPLIT_K16_GRID") ? std::atoi(std::getenv("TOUCAN_SPLIT_K16_GRID")) : 128) && std::getenv("TOUCAN_NO_SPLIT_K16") == nullptr) {
    dim3 grid(d.n_tiles * (d.tile_rows / 16), cols / 16), block(256);
    if (dual) hipLaunchKernelGGL((conv_splitk_f32_kernel<true, true>), grid, block, 0, st, d);
    else hipLaunchKernelGGL((conv_splitk_f32_kernel<false, true>), grid, block, 0, st, d);
    return launch_status("conv1d (split-K, 16 x 16)");
  }
  dim3 grid(d.n_tiles * (d.tile_rows / 32), cols / 32), block(256);
  if (dual) hipLaunchKernelGGL((conv_splitk_f32_kernel<true, false>), grid, block, 0, st, d);
  else hipLaunchKernelGGL((conv_splitk_f32_kernel<false, false>), grid, block, 0, st, d);
  return launch_status("conv1d (split-K)");
}

static int launch_gemm_rows(const TtsConvDesc& d, hipStream_t st) {
  const bool dual = d.mode != TTS_MODE_LINEAR, xb = d.io_flags & TTS_IO_X_BF16;
  const int cols = dual ? d.half_pad : d.wn;
  dim3 grid(d.n_tiles, cols / 64), block(256);
#define TTS_GEMM_ROWS(DUAL_, BF16_, XB_, F16_) hipLaunchKernelGGL((gemm_rows_kernel<DUAL_, BF16_, XB_, F16_>), grid, block, 0, st, d)
#define TTS_GEMM_ROWS2(DUAL_, XB_) do { if (d.compute == 2) TTS_GEMM_ROWS(DUAL_, true, XB_, true); else TTS_GEMM_ROWS(DUAL_, true, XB_, false); } while (0)
  if (d.compute == 0) {
    if (dual) TTS_GEMM_ROWS(true, false, false, false);
    else TTS_GEMM_ROWS(false, false, false, false);
  } else if (xb) {
    if (dual) TTS_GEMM_ROWS2(true, true);
    else TTS_GEMM_ROWS2(false, true);
  } else {
    if (dual) TTS_GEMM_ROWS2(true, false);
    else TTS_GEMM_ROWS2(false, false);
  }
#undef TTS_GEMM_ROWS2
#undef TTS_GEMM_ROWS
  return launch_status("conv1d (1-tap rows)");
}

int conv1d_dispatch(const TtsConvDesc& d, hipStream_t st) {
  TTS_CHECK_ARG(d.x && d.w && d.y && d.tiles, "conv1d: null pointer");
  TTS_CHECK_ARG(d.cin > 0 && d.cout > 0 && d.taps > 0 && d.dil > 0, "conv1d: bad dims");
  TTS_CHECK_ARG(d.cin_pad % 32 == 0 && d.cin_pad >= d.cin, "conv1d: cin_pad %d must be a multiple of 32 >= cin %d", d.cin_pad, d.cin);
  TTS_CHECK_ARG(d.mode >= 0 && d.mode <= 3, "conv1d: bad mode %d", d.mode);
  TTS_CHECK_ARG(d.compute >= 0 && d.compute <= 3, "conv1d: bad compute %d (0 fp32, 1 bf16, 2 fp16, 3 split fp32)", d.compute);
  TTS_CHECK_ARG(d.compute == 0 || d.compute == 3 || ((d.io_flags & TTS_IO_F16) != 0) == (d.compute == 2) || !(d.io_flags & 7),
                "conv1d: the 16-bit tensors of a bf16 / fp16 call must be in the call's own format");
  TTS_CHECK_ARG(d.compute != 3 || !(d.io_flags & 7), "conv1d: the split fp32 product takes and returns fp32 tensors");
  TTS_CHECK_ARG(d.mode != TTS_MODE_COUPLING || d.aux, "conv1d: coupling mode needs aux");
  TTS_CHECK_ARG(d.pre_act != TTS_PRE_SNAKE || (d.snake_alpha && d.snake_beta && d.snake_filt), "conv1d: PRE_SNAKE needs alpha/beta/filter");
  if (d.n_tiles == 0) return TTS_OK;
  const int cols = d.mode == TTS_MODE_LINEAR ? d.wn : d.half_pad;
  ConvShape s = pick_shape(d.cout, d.mode);
  if (d.tile_rows == 64 && small_form_ok(d.cout, d.mode, cols)) s = d.mode == TTS_MODE_LINEAR ? S_64x64 : S_D64x64;
  int bm, bn;
  shape_dims(s, bm, bn);
  TTS_CHECK_ARG(d.tile_rows == bm, "conv1d: tile table built for %d rows, kernel needs %d", d.tile_rows, bm);
  TTS_CHECK_ARG(cols % bn == 0 && cols >= d.cout, "conv1d: packed width %d not a multiple of the N tile %d (cout %d)", cols, bn, d.cout);
  TTS_CHECK_ARG(d.mode == TTS_MODE_LINEAR || d.wn == 2 * d.half_pad, "conv1d: dual mode needs wn == 2*half_pad");
  if (d.compute == 0 && std::getenv("TOUCAN_SPLIT_K_LOG"))  // (debugging aid: which form every fp32 launch takes)
    fprintf(stderr, "conv-f32 cin %d cout %d taps %d tile_rows %d tiles %d cols %d ldx %d flag %d -> splitk %d\n", d.cin, d.cout, d.taps, d.tile_rows,
            d.n_tiles, cols, d.ldx, (int)((d.io_flags & TTS_IO_SPLIT_K) != 0), (int)splitk_ok(d, cols));
  if (splitk_ok(d, cols)) return launch_splitk(d, cols, st);  // (whatever the table's tile rows: the kernel cuts its own row blocks out of them)
  if ((s == S_64x64 || s == S_D64x64) && gemm_rows_ok(d)) return launch_gemm_rows(d, st);
  // Small-batch form with 128-column tiles (four wavefronts side by side) once the grid fills the chip anyway: wide outputs
  // then re-read their activation rows half as often (batch 32: acoustic model +3-5 %); below that the 64-column tiles keep
  // twice the workgroups in flight (batch 1: 11.8 vs 12.2 ms with wide tiles everywhere).
  if (s == S_64x64 && cols % 128 == 0 && cols >= 256 && (long long)d.n_tiles * (cols / 64) >= 512) {
    s = S_64x128;
    bn = 128;
  }
  switch (s) {
    case S_128x128: return launch_cfg<2, 2, 2, 2, false>(d, st);
    case S_128x96: return launch_cfg<1, 3, 4, 1, false>(d, st);
    case S_128x64: return launch_cfg<2, 1, 2, 2, false>(d, st);
    case S_256x32: return launch_cfg<2, 1, 4, 1, false>(d, st);
    case S_D128x96: return launch_cfg<1, 3, 4, 1, true>(d, st);
    case S_64x64: return launch_cfg<1, 1, 2, 2, false>(d, st);
    case S_D64x64: return launch_cfg<1, 1, 2, 2, true>(d, st);
    case S_64x128: return launch_cfg<2, 1, 1, 4, false>(d, st);
  }
  return TTS_E_ARG;
}

int conv1d_tile_rows(int cout, int mode) {
  int bm, bn;
  shape_dims(pick_shape(cout, mode), bm, bn);
  return bm;
}

int conv1d_small_tile_rows(int cout, int mode, int packed_cols) { return small_form_ok(cout, mode, packed_cols) ? 64 : 0; }

int conv1d_n_tile(int cout, int mode) {
  int bm, bn;
  shape_dims(pick_shape(cout, mode), bm, bn);
  return bn;
}

}  // namespace tts
