// One WaveNet layer of the PostFlow's coupling blocks in one launch (16-bit MFMA configurations):
//   acts = tanh(a) * sigmoid(g),  [a | g] = in_layer(h) (5 taps, 192 -> 384) + bias + cond          wavenet.py:104-110, :29-35
//   [h | skip] += res_skip_layer(acts) (1 tap, 192 -> 384; last layer: 192 -> 192 into the skip sum)   wavenet.py:112-118
// Two launches (tts_conv1d in GATED mode, then an accumulating 1-tap conv) cost 37 + 29 us per layer at batch 32 - 72 layers per
// pass - almost all of it fixed per-launch / per-slab latency on tiny GEMMs (K = 192).  Here one 256-thread workgroup owns 64
// frames of one utterance and ALL output channels: the gate activations never leave LDS and the hidden state is read and
// written once.  Because a 5-tap conv reads two frames either side of the tile, the hidden state cannot be updated in place:
// the layer reads hs_in and writes hs_out (the host ping-pongs two buffers).
//
// Both products run transposed (weights are the MFMA A operand, accumulator row = output channel, lane = frame), so epilogues
// work on float4 / packed 8-byte pieces of a frame's row (see resblock.hip).  Weights stream as 32-channel slabs (24 KB) through
// a WN_RING-deep LDS ring filled by direct global->LDS loads (global_load_lds_dwordx4): no staging registers, WN_RING - 1 slabs in flight
// across the raw s_barrier of a step, counted s_waitcnt vmcnt (cdna_hip_programming.md, "Pipelining across barriers" - this
// kernel runs at one workgroup per CU and one wavefront per SIMD, the regime where that matters).
#include "common.h"

namespace tts {

namespace {
constexpr int WN_H = 192;            // hidden channels
constexpr int WN_BM = 64;            // frames per workgroup
constexpr int WN_TAPS = 5;
constexpr int WN_XP = WN_H + 8;      // LDS pitch of the window / of acts (16-bit elements)
constexpr int WN_KS = 32;            // channels per weight slab
constexpr int WN_SLAB_BYTES = (WN_KS / 8) * 384 * 16;  // 24 KB: [4][384][8] 16-bit
constexpr int WN_SLABS1 = WN_TAPS * (WN_H / WN_KS);    // 30 slab steps of the gated conv
constexpr int WN_SLABS2 = WN_H / WN_KS;                // 6 of the res/skip conv
// LDS ring slots: WN_RING - 1 slabs in flight (five slots = 120 KB of the CU's 160: a global -> LDS load lands ~1.1 us after it is
// issued, a slab step takes ~0.9 us)
constexpr int WN_RING = 5;
constexpr int WN_LOADS = 6;                            // global_load_lds instructions per wavefront and slab (narrow slabs re-request units)
}  // namespace

template <bool F16>
__global__ __launch_bounds__(256) void wavenet_layer_kernel(const TtsWavenetDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned short* xs = reinterpret_cast<unsigned short*>(lds_raw);                       // [68][XP] window of h, later [64][XP] acts
  unsigned char* ring = lds_raw + ((WN_BM + WN_TAPS - 1) * WN_XP * 2 + 255) / 256 * 256;  // [WN_RING][24 KB]
  const TtsTile tile = d.tiles[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, lrow = lane & 31, lk = lane >> 5;
  const int n2 = d.cout2;                       // 384 or 192
  const int units2 = (WN_KS / 8) * n2;          // 16-byte units of a res/skip slab
  const int total = WN_SLABS1 + WN_SLABS2;
  const char* w1 = reinterpret_cast<const char*>(d.w1);
  const char* w2 = reinterpret_cast<const char*>(d.w2);

  // direct global -> LDS copy of weight slab s into ring[s % WN_RING]: wave w moves units i*256 + w*64 + lane (1 KB per instruction).
  // Every slab is WN_LOADS instructions per wavefront (a narrow res/skip slab re-requests its last units), so the number of loads
  // in flight behind a slab is a compile-time constant.  One piece = one instruction per wavefront; a step spreads the pieces of
  // the slab it requests between its matrix instructions (issued in a block in front of them, each piece held the wavefront's
  // issue for 100+ cycles - MI355X_MICROARCH.md, "LDS-DMA piece issue cost": 38.2 -> 34.9 us per layer)
  auto issue_piece = [&](int s, int i) __attribute__((always_inline)) {
    if (s >= total) return;
    const bool second = s >= WN_SLABS1;
    const int units = second ? units2 : (WN_KS / 8) * 384;
    const char* src = second ? w2 + (size_t)(s - WN_SLABS1) * units2 * 16 : w1 + (size_t)s * WN_SLAB_BYTES;  // slabs are contiguous: [tap][k/8][n][8]
    unsigned char* dst = ring + (size_t)(s % WN_RING) * WN_SLAB_BYTES;
#ifdef WN_DIAG_NO_DMA  // (timing diagnostics only: no weight traffic, results are wrong)
    if (s >= WN_RING) return;
#endif
    const int u0 = i * 256 + wave * 64;
    int u = u0 + lane;
    u = u < units ? u : units - 1;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)u * 16),
                                     (void __attribute__((address_space(3)))*)(dst + (size_t)u0 * 16), 16, 0, 0);
  };
  auto issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WN_LOADS; ++i) issue_piece(s, i);
  };

#pragma unroll
  for (int i = 0; i < WN_RING - 1; ++i) issue(i);  // the first slabs land while the window is staged (total = 36 >= WN_RING - 1)
  // ---- window of the hidden state: rows row0 - 2 .. row0 + 65 (zero outside the utterance = the conv's zero padding) -> 16-bit
  {
    // PER independent 16-byte loads per thread are in flight before the first is consumed (clamped addresses, no branches): one
    // trip pays the HBM latency once, not once per load
    constexpr int Q4 = WN_H / 4, ROWS = WN_BM + WN_TAPS - 1, TOTAL = ROWS * Q4, PER = 7;
    for (int base = tid; base < TOTAL; base += 256 * PER) {
      float4 v[PER];
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        int e = base + p * 256;
        e = e < TOTAL ? e : TOTAL - 1;
        const int r = e / Q4, c4 = (e % Q4) * 4;
        const int gr = tile.row0 - 2 + r;
        const int grc = gr < tile.seq_begin ? tile.seq_begin : (gr >= tile.seq_end ? tile.seq_end - 1 : gr);
        v[p] = *reinterpret_cast<const float4*>(d.hs_in + (size_t)grc * d.ld_in + c4);
        if (gr != grc) v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        const int e = base + p * 256;
        if (e < TOTAL) {
          const int r = e / Q4, c4 = (e % Q4) * 4;
          *reinterpret_cast<uint2*>(xs + r * WN_XP + c4) = make_uint2(pack16<F16>(v[p].x, v[p].y), pack16<F16>(v[p].z, v[p].w));
        }
      }
    }
  }
  f32x16 acc[6];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

  // slab s has landed; the loads of the slabs behind it (min(WN_RING - 2, slabs left) x WN_LOADS) may stay in flight
  auto wait_for = [&](int s) __attribute__((always_inline)) {
    const int behind = total - 1 - s;  // slabs requested behind s so far: min(behind, WN_RING - 2)
    static_assert(WN_RING >= 3 && WN_RING <= 6, "one wait per number of slabs in flight below");
    if (WN_RING >= 6 && behind >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * WN_LOADS) : "memory");
    else if (WN_RING >= 5 && behind >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * WN_LOADS) : "memory");
    else if (WN_RING >= 4 && behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WN_LOADS) : "memory");
    else if (behind >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN_LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS stores (window / acts) are done before the barrier publishes them
    __builtin_amdgcn_s_barrier();
  };

  // ---- gated conv: acc[0..2] = a, acc[3..5] = g for channels wn*96 + j*32 ..; frames wm*32 ..
  for (int s = 0; s < WN_SLABS1; ++s) {
    wait_for(s);
    const int tap = s / (WN_H / WN_KS), k0 = (s % (WN_H / WN_KS)) * WN_KS;
    const unsigned short* wb = reinterpret_cast<const unsigned short*>(ring + (size_t)(s % WN_RING) * WN_SLAB_BYTES);
    // all fragments of the step first (14 LDS reads in flight), then the 12 MFMAs: with one wavefront per SIMD nothing else hides
    // an LDS round trip in front of every MFMA, which is what the compiler's own interleaving produced
    bf16x8 xf[WN_KS / 16], wf[WN_KS / 16][6];
#pragma unroll
    for (int kk = 0; kk < WN_KS / 16; ++kk) {
      xf[kk] = *reinterpret_cast<const bf16x8*>(xs + (wm * 32 + lrow + tap) * WN_XP + k0 + kk * 16 + lk * 8);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int col = (j < 3 ? 0 : WN_H) + wn * 96 + (j % 3) * 32 + lrow;
        wf[kk][j] = *reinterpret_cast<const bf16x8*>(wb + ((size_t)(kk * 2 + lk) * 384 + col) * 8);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifndef WN_DIAG_NO_MFMA
#pragma unroll
    for (int kk = 0; kk < WN_KS / 16; ++kk)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if ((kk * 6 + j) % 2 == 0) {  // one piece of the slab WN_RING - 1 steps ahead per two matrix instructions
          issue_piece(s + WN_RING - 1, (kk * 6 + j) / 2);
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[j] = mfma16<F16>(wf[kk][j], xf[kk], acc[j]);
      }
#else
    issue(s + WN_RING - 1);
    for (int j = 0; j < 6; ++j) acc[j][0] += bf16_to_f32(wf[0][j][0]) + bf16_to_f32(xf[1][1]) + bf16_to_f32(wf[1][j][2]);
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- acts = tanh(a + bias + cond) * sigmoid(g + bias + cond) -> LDS (over the window), 16-bit
  __builtin_amdgcn_s_barrier();  // every wave is done reading the window
  {
    const int t = wm * 32 + lrow, row = tile.row0 + t;
    const bool live = row < tile.seq_end;
    const float* cr = d.cond + (size_t)(live ? row : tile.seq_end - 1) * d.ld_cond;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int c = wn * 96 + j * 32 + 8 * rq + 4 * lk;
        const float4 ba = *reinterpret_cast<const float4*>(d.b1 + c), bg = *reinterpret_cast<const float4*>(d.b1 + WN_H + c);
        const float4 ca = *reinterpret_cast<const float4*>(cr + c), cg = *reinterpret_cast<const float4*>(cr + WN_H + c);
        const float av[4] = {acc[j][4 * rq] + ba.x + ca.x, acc[j][4 * rq + 1] + ba.y + ca.y, acc[j][4 * rq + 2] + ba.z + ca.z, acc[j][4 * rq + 3] + ba.w + ca.w};
        const float gv[4] = {acc[j + 3][4 * rq] + bg.x + cg.x, acc[j + 3][4 * rq + 1] + bg.y + cg.y, acc[j + 3][4 * rq + 2] + bg.z + cg.z,
                             acc[j + 3][4 * rq + 3] + bg.w + cg.w};
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // hardware exp / reciprocal (a few ulp, far below the 16-bit rounding of acts): with one wavefront per SIMD the
          // library tanhf / expf / IEEE division of 48 elements per lane were 6 us of a 37 us launch
          const float th = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * av[q])) - 1.0f;
          v[q] = th * __builtin_amdgcn_rcpf(1.0f + __expf(-gv[q]));
        }
        *reinterpret_cast<uint2*>(xs + t * WN_XP + c) = make_uint2(pack16<F16>(v[0], v[1]), pack16<F16>(v[2], v[3]));
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  }
  // ---- res / skip conv: output channels wn * (n2 / 2) + j * 32 .., j < n2 / 64
  const int half2 = n2 >> 1, nj = n2 >> 6;
  for (int s = WN_SLABS1; s < total; ++s) {
    wait_for(s);  // (also publishes acts on the first step)
    const int k0 = (s - WN_SLABS1) * WN_KS;
    const unsigned short* wb = reinterpret_cast<const unsigned short*>(ring + (size_t)(s % WN_RING) * WN_SLAB_BYTES);
    bf16x8 xf[WN_KS / 16], wf[WN_KS / 16][6];
#pragma unroll
    for (int kk = 0; kk < WN_KS / 16; ++kk) {
      xf[kk] = *reinterpret_cast<const bf16x8*>(xs + (wm * 32 + lrow) * WN_XP + k0 + kk * 16 + lk * 8);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int col = wn * half2 + (j < nj ? j : 0) * 32 + lrow;  // (j >= nj: a duplicate read, its MFMA is skipped)
        wf[kk][j] = *reinterpret_cast<const bf16x8*>(wb + ((size_t)(kk * 2 + lk) * n2 + col) * 8);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < WN_KS / 16; ++kk)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if ((kk * 6 + j) % 2 == 0) {
          issue_piece(s + WN_RING - 1, (kk * 6 + j) / 2);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j < nj) acc[j] = mfma16<F16>(wf[kk][j], xf[kk], acc[j]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- [h | skip] out = in + res_skip + bias (last layer: the skip half only)
  {
    const int row = tile.row0 + wm * 32 + lrow;
    if (row < tile.seq_end) {
      const int col0 = n2 == 384 ? 0 : WN_H;
      const float* ir = d.hs_in + (size_t)row * d.ld_in + col0;
      float* orow = d.hs_out + (size_t)row * d.ld_out + col0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (j < nj) {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int c = wn * half2 + j * 32 + 8 * rq + 4 * lk;
            const float4 b = *reinterpret_cast<const float4*>(d.b2 + c), x = *reinterpret_cast<const float4*>(ir + c);
            *reinterpret_cast<float4*>(orow + c) = make_float4((acc[j][4 * rq] + b.x) + x.x, (acc[j][4 * rq + 1] + b.y) + x.y,
                                                               (acc[j][4 * rq + 2] + b.z) + x.z, (acc[j][4 * rq + 3] + b.w) + x.w);
          }
        }
      }
    }
  }
}

int wavenet_layer(const TtsWavenetDesc& d, hipStream_t st) {
  TTS_CHECK_ARG(d.hs_in && d.hs_out && d.cond && d.w1 && d.b1 && d.w2 && d.b2 && d.tiles, "wavenet_layer: null pointer");
  TTS_CHECK_ARG(d.hs_in != d.hs_out, "wavenet_layer: the hidden state cannot be updated in place (the 5-tap conv reads neighbouring tiles)");
  TTS_CHECK_ARG(d.cout2 == 384 || d.cout2 == 192, "wavenet_layer: res/skip width %d (384, or 192 for the last layer)", d.cout2);
  TTS_CHECK_ARG(d.compute == TTS_COMPUTE_BF16 || d.compute == TTS_COMPUTE_F16, "wavenet_layer: 16-bit MFMA configurations only (compute %d)", d.compute);
  TTS_CHECK_ARG(d.tile_rows == WN_BM, "wavenet_layer: tile table must use %d rows, got %d", WN_BM, d.tile_rows);
  TTS_CHECK_ARG((d.ld_in & 3) == 0 && (d.ld_out & 3) == 0 && (d.ld_cond & 3) == 0 && ((uintptr_t)d.hs_in & 15) == 0 && ((uintptr_t)d.hs_out & 15) == 0 &&
                    ((uintptr_t)d.cond & 15) == 0 && ((uintptr_t)d.w1 & 15) == 0 && ((uintptr_t)d.w2 & 15) == 0 && ((uintptr_t)d.b1 & 15) == 0 &&
                    ((uintptr_t)d.b2 & 15) == 0,
                "wavenet_layer: rows and weights must be 16-byte aligned");
  if (d.n_tiles == 0) return TTS_OK;
  const size_t lds = ((size_t)(WN_BM + WN_TAPS - 1) * WN_XP * 2 + 255) / 256 * 256 + WN_RING * (size_t)WN_SLAB_BYTES;
  static unsigned long long raised[2] = {0, 0};
  const bool f16 = d.compute == TTS_COMPUTE_F16;
  const void* k = f16 ? reinterpret_cast<const void*>(wavenet_layer_kernel<true>) : reinterpret_cast<const void*>(wavenet_layer_kernel<false>);
  if (raise_lds_limit(k, raised[f16 ? 1 : 0]) != hipSuccess) {
    set_error("wavenet_layer: raising the dynamic LDS limit failed");
    return TTS_E_LAUNCH;
  }
  if (f16) hipLaunchKernelGGL(wavenet_layer_kernel<true>, dim3(d.n_tiles), dim3(256), lds, st, d);
  else hipLaunchKernelGGL(wavenet_layer_kernel<false>, dim3(d.n_tiles), dim3(256), lds, st, d);
  return launch_status("wavenet_layer");
}

}  // namespace tts

extern "C" int tts_wavenet_layer(const TtsWavenetDesc* d, tts_stream_t stream) {
  if (!d) {
    tts::set_error("tts_wavenet_layer: null descriptor");
    return TTS_E_ARG;
  }
  return tts::wavenet_layer(*d, reinterpret_cast<hipStream_t>(stream));
}
