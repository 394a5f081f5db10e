// Fused Conformer feed-forward module (kernel size 1):   y = [LN_post]( x + alpha * (W2 relu(W1 LN(x) + b1) + b2) )
//
// One launch replaces LayerNorm + two pointwise convs (+ the block's final LayerNorm) of Layers/EncoderLayer.py:84-90 / :128-136
// with Layers/MultiLayeredConv1d.py:50-51 (w_2(relu(w_1(x)))) at kernel size 1.  The hidden activation [rows, 1536] never exists
// outside registers: unfused it is written and read back once per module (2 x 126 MB at 20 480 rows), which - not the 24 GFLOP -
// is what made the pair of launches take 195 us.
//
// Decomposition (256 threads = 4 wavefronts, one per SIMD; 128 rows per workgroup, 32 per wavefront; rows are independent):
//   * Both products run transposed (weights are the MFMA A operand; accumulator row = channel, lane = frame), as in resblock.hip.
//   * A wavefront loads its 32 rows of x (fp32) straight into the B-operand layout (lane = frame, registers = 8 consecutive
//     channels per 16-channel k step), normalises them in registers (a row is spread over the two lanes l and l ^ 32: one
//     cross-lane add per statistic) and keeps the 16-bit operand xb[12] for the whole kernel.
//   * The hidden axis is walked in chunks of 32: h = W1[chunk] xb (12 MFMAs, K = 192), + b1, ReLU, 16-bit - and the accumulator
//     tile IS the B operand of the second product (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"):
//     y[j] += W2[j][chunk] h (12 MFMAs, six 32-channel output blocks, K = 32).  The k order inside the chunk is the one the
//     accumulator registers have (lane half lk holds hidden 8 rq + 4 lk + i); W2 is packed in that order on the host
//     (packing.pack_ffn).
//   * Weights: one 28 KB stage per chunk ([12 W1 fragments | 12 W2 fragments | b1 of the chunk as 4 fragments of per-lane
//     float4s], each fragment 64 lanes x 16 B = one ds_read_b128 per lane) streams through a five-deep LDS ring filled by global_load_lds (no staging registers), counted vmcnt, raw
//     s_barrier - the scheme of wavenet.hip.
//   * Epilogue in registers: + b2, * alpha, + x (re-read, L2), optional second LayerNorm (the block's norm_final), float4 stores.
// 16-bit MFMA configurations only (v_mfma_f32_32x32x16_{bf16,f16}); fp32 statistics, accumulators and residual stream.
#include "common.h"

namespace tts {

namespace {
constexpr int FF_C = 192;                  // model width
constexpr int FF_KS = FF_C / 16;           // k steps of the first product
constexpr int FF_J = FF_C / 32;            // 32-channel output blocks of the second product
constexpr int FF_ROWS = 128;               // rows per workgroup
constexpr int FF_STAGE = (FF_KS + 2 * FF_J + 4) * 1024;  // 28 KB: [12 W1 fragments][6 x 2 W2 fragments][4 fragments of b1 (fp32)]
// stages of the LDS ring: a chunk's 24 MFMAs take ~0.4 us, a global -> LDS load ~1.5 us to land - with three stages (two chunks of
// lead) the loop ran at the load latency (1.2 - 1.6 us per chunk measured); five stages give four chunks of lead
constexpr int FF_DEPTH = 5;
}  // namespace

// (LDS reads of the stages are ordinary loads.  Reading them through inline asm - to keep the compiler from ordering them
// against the global -> LDS loads - does not work: under register pressure it copies an asm output to an accumulator register
// right behind the asm statement, before the data has arrived.)
// Timing diagnostics (tools/build_variant.sh NAME -DFF_DIAG_CLOCK): s_memtime stamps inside the chunk loop, summed per workgroup
// into g_ff_clock (wave 0 of workgroup 0 only), read back by tts_ffn_diag_clock.  The shipped library carries no stamp.
#ifdef FF_DIAG_CLOCK
__device__ unsigned long long g_ff_clock[8];
#define FF_STAMP(k_)                                                                    \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[k_])::"memory");   \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
#else
#define FF_STAMP(k_)
#endif

template <bool F16>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(const TtsFfnDesc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* ring = lds_raw;                                         // [FF_DEPTH][28 KB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lk = lane >> 5;
  const int n_chunks = d.hidden >> 5;
  const char* wsrc = reinterpret_cast<const char*>(d.w);

  // direct global -> LDS copy of stage s into ring[s % 3]: wave w moves units i * 256 + w * 64 + lane (1 KB per instruction)
  auto issue_piece = [&](int s, int i) __attribute__((always_inline)) {  // piece i (of 7) of this wavefront's share of stage s
    const char* src = wsrc + (size_t)s * FF_STAGE;
    unsigned char* dst = ring + (size_t)(s % FF_DEPTH) * FF_STAGE;
    const int u0 = i * 256 + wave * 64;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)(u0 + lane) * 16),
                                     (void __attribute__((address_space(3)))*)(dst + (size_t)u0 * 16), 16, 0, 0);
  };
  auto issue = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < FF_STAGE / 4096; ++i) issue_piece(s, i);
  };
#pragma unroll
  for (int s0 = 0; s0 < FF_DEPTH - 1; ++s0)
    if (s0 < n_chunks) issue(s0);

  // ---- this lane's frame: 96 of its 192 channels (16 ks + 8 lk .. + 7), LayerNorm, 16-bit B operand
  const int row_raw = blockIdx.x * FF_ROWS + wave * 32 + lrow;
  const bool live = row_raw < d.rows;
  const int row = live ? row_raw : d.rows - 1;
  const float* xr = d.x + (size_t)row * d.ldx;
  bf16x8 xb[FF_KS];
  {
    float4 v[FF_KS][2];
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) {
      v[ks][0] = *reinterpret_cast<const float4*>(xr + 16 * ks + 8 * lk);
      v[ks][1] = *reinterpret_cast<const float4*>(xr + 16 * ks + 8 * lk + 4);
    }
    float s = 0.0f;
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) s += (v[ks][0].x + v[ks][0].y) + (v[ks][0].z + v[ks][0].w) + (v[ks][1].x + v[ks][1].y) + (v[ks][1].z + v[ks][1].w);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / FF_C);
    float q = 0.0f;
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) {
      const float e[8] = {v[ks][0].x - mean, v[ks][0].y - mean, v[ks][0].z - mean, v[ks][0].w - mean,
                          v[ks][1].x - mean, v[ks][1].y - mean, v[ks][1].z - mean, v[ks][1].w - mean};
#pragma unroll
      for (int i = 0; i < 8; ++i) q = fmaf(e[i], e[i], q);
    }
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / FF_C) + d.eps);
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) {
      const float4 g0 = *reinterpret_cast<const float4*>(d.ln_g + 16 * ks + 8 * lk), g1 = *reinterpret_cast<const float4*>(d.ln_g + 16 * ks + 8 * lk + 4);
      const float4 c0 = *reinterpret_cast<const float4*>(d.ln_b + 16 * ks + 8 * lk), c1 = *reinterpret_cast<const float4*>(d.ln_b + 16 * ks + 8 * lk + 4);
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 p = {pack16<F16>((v[ks][0].x - mean) * rstd * g0.x + c0.x, (v[ks][0].y - mean) * rstd * g0.y + c0.y),
                       pack16<F16>((v[ks][0].z - mean) * rstd * g0.z + c0.z, (v[ks][0].w - mean) * rstd * g0.w + c0.w),
                       pack16<F16>((v[ks][1].x - mean) * rstd * g1.x + c1.x, (v[ks][1].y - mean) * rstd * g1.y + c1.y),
                       pack16<F16>((v[ks][1].z - mean) * rstd * g1.z + c1.z, (v[ks][1].w - mean) * rstd * g1.w + c1.w)};
      xb[ks] = __builtin_bit_cast(bf16x8, p);
    }
  }

  f32x16 y[FF_J];
#pragma unroll
  for (int j = 0; j < FF_J; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) y[j][r] = 0.0f;

  // the loads of a stage have landed when at most `behind` younger stages (7 loads each, per wavefront) are still in flight;
  // then the workgroup barrier: every wavefront's part of the stage is there, and every wavefront is past its earlier LDS reads
  auto wait_for = [&](int behind) __attribute__((always_inline)) {
    if (behind >= 3) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
    else if (behind == 2) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  static_assert(FF_DEPTH == 5, "wait_for's cases cover up to three stages in flight");
  static_assert(FF_STAGE / 4096 == 7, "wait_for counts seven global_load_lds per wavefront and stage");

  // One chunk at a time: 28 fragment reads, first product, ReLU, second product.  (With one wavefront per SIMD these phases run one
  // after the other - 1.3 us per chunk measured, 3 x the MFMA time.  Reading the next chunk's fragments under this chunk's MFMAs
  // needs a second register set, which pushed the operands into accumulator registers and cost more in copies than it hid.)
  wait_for(n_chunks - 1 < FF_DEPTH - 2 ? n_chunks - 1 : FF_DEPTH - 2);  // stage 0 (stages 0 .. FF_DEPTH - 2 were issued at the top)
#ifdef FF_DIAG_CLOCK
  unsigned long long stamp[6], psum[5] = {0, 0, 0, 0, 0};
#endif
  for (int c = 0; c < n_chunks; ++c) {
    FF_STAMP(0);
    if (c > 0) {
      // stage c is complete (chunks up to c + FF_DEPTH - 3 have been issued: FF_DEPTH - 3 stages may stay in flight), and every
      // wavefront is past its reads of chunk c - 1 ...
      const int behind = n_chunks - 1 - c;
      wait_for(behind < FF_DEPTH - 3 ? behind : FF_DEPTH - 3);
    }
    FF_STAMP(1);
    // ... so that stage takes chunk c + FF_DEPTH - 2; its seven loads are issued between the MFMAs of the second product below (a
    // global -> LDS load costs 60 - 180 cycles of issue: 960 cycles per chunk measured when they sat in front of the LDS reads)
    const bool refill = c + FF_DEPTH - 2 < n_chunks && c > 0;
    const unsigned char* st = ring + (size_t)(c % FF_DEPTH) * FF_STAGE + lane * 16;
    bf16x8 a1[FF_KS], a2[FF_J][2], bbf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bbf[k] = *reinterpret_cast<const bf16x8*>(st + (FF_KS + 2 * FF_J + k) * 1024);  // (b1 rides in the stage)
    __builtin_amdgcn_sched_barrier(0);  // (b1, W1, W2 in the order of use: LDS reads return in order, every MFMA waits for its own operand only)
#pragma unroll
    for (int k = 0; k < FF_KS; ++k) {
      a1[k] = *reinterpret_cast<const bf16x8*>(st + k * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 2 * FF_J; ++k) a2[k >> 1][k & 1] = *reinterpret_cast<const bf16x8*>(st + (FF_KS + k) * 1024);
    __builtin_amdgcn_sched_barrier(0);
    FF_STAMP(2);
    // h = W1[chunk] xb + b1 (the accumulator starts from the bias)
    f32x16 h;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const float4 bb = __builtin_bit_cast(float4, bbf[rq]);
      h[4 * rq] = bb.x; h[4 * rq + 1] = bb.y; h[4 * rq + 2] = bb.z; h[4 * rq + 3] = bb.w;
    }
#pragma unroll
    for (int ks = 0; ks < FF_KS; ++ks) h = mfma16<F16>(a1[ks], xb[ks], h);
    FF_STAMP(3);
    // 16-bit, then ReLU on the packed pairs: registers 4 rq + i of this lane are hidden channels 32 c + 8 rq + 4 lk + i = k slots of
    // the second product.  (max(v, 0) of a bf16 / fp16 value = signed 16-bit integer max of its bit pattern with 0: negative
    // values have the sign bit set; -0 becomes +0.  One v_pk_max_i16 per pair instead of two v_max_f32 per pair.)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    auto relu2 = [](unsigned int pr) __attribute__((always_inline)) {
      const i16x2 z = {0, 0};
      return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pr), z));
    };
    const u32x4 p0 = {relu2(pack16<F16>(h[0], h[1])), relu2(pack16<F16>(h[2], h[3])), relu2(pack16<F16>(h[4], h[5])), relu2(pack16<F16>(h[6], h[7]))};
    const u32x4 p1 = {relu2(pack16<F16>(h[8], h[9])), relu2(pack16<F16>(h[10], h[11])), relu2(pack16<F16>(h[12], h[13])), relu2(pack16<F16>(h[14], h[15]))};
    const bf16x8 hb0 = __builtin_bit_cast(bf16x8, p0), hb1 = __builtin_bit_cast(bf16x8, p1);
    FF_STAMP(4);
#pragma unroll
    for (int j = 0; j < FF_J; ++j) {
      y[j] = mfma16<F16>(a2[j][0], hb0, y[j]);
      if (refill) issue_piece(c + FF_DEPTH - 2, j);
    }
#pragma unroll
    for (int j = 0; j < FF_J; ++j) {
      y[j] = mfma16<F16>(a2[j][1], hb1, y[j]);
      if (j == 0 && refill) issue_piece(c + FF_DEPTH - 2, FF_J);
    }
    __builtin_amdgcn_sched_barrier(0);
    FF_STAMP(5);
#ifdef FF_DIAG_CLOCK
#pragma unroll
    for (int k = 0; k < 5; ++k) psum[k] += stamp[k + 1] - stamp[k];
#endif
  }
#ifdef FF_DIAG_CLOCK
  if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) g_ff_clock[k] = psum[k];
    g_ff_clock[5] = n_chunks;
  }
#endif

  // ---- epilogue: lane = frame, registers 4 rq + i of block j = channels 32 j + 8 rq + 4 lk + i
  float o[FF_J][4][4];
#pragma unroll
  for (int j = 0; j < FF_J; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int c = 32 * j + 8 * rq + 4 * lk;
      const float4 b2 = *reinterpret_cast<const float4*>(d.b2 + c), xv = *reinterpret_cast<const float4*>(xr + c);
      o[j][rq][0] = xv.x + d.alpha * (y[j][4 * rq] + b2.x);
      o[j][rq][1] = xv.y + d.alpha * (y[j][4 * rq + 1] + b2.y);
      o[j][rq][2] = xv.z + d.alpha * (y[j][4 * rq + 2] + b2.z);
      o[j][rq][3] = xv.w + d.alpha * (y[j][4 * rq + 3] + b2.w);
    }
  if (d.post_g) {  // the block's final LayerNorm over the 192 channels of the frame (this lane's 96 + lane ^ 32's)
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < FF_J; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) s += (o[j][rq][0] + o[j][rq][1]) + (o[j][rq][2] + o[j][rq][3]);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / FF_C);
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < FF_J; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
#pragma unroll
        for (int i = 0; i < 4; ++i) q = fmaf(o[j][rq][i] - mean, o[j][rq][i] - mean, q);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / FF_C) + d.eps);
#pragma unroll
    for (int j = 0; j < FF_J; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int c = 32 * j + 8 * rq + 4 * lk;
        const float4 g = *reinterpret_cast<const float4*>(d.post_g + c), b = *reinterpret_cast<const float4*>(d.post_b + c);
        o[j][rq][0] = (o[j][rq][0] - mean) * rstd * g.x + b.x;
        o[j][rq][1] = (o[j][rq][1] - mean) * rstd * g.y + b.y;
        o[j][rq][2] = (o[j][rq][2] - mean) * rstd * g.z + b.z;
        o[j][rq][3] = (o[j][rq][3] - mean) * rstd * g.w + b.w;
      }
  }
  if (live) {
    float* yr = d.y + (size_t)row * d.ldy;
#pragma unroll
    for (int j = 0; j < FF_J; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<float4*>(yr + 32 * j + 8 * rq + 4 * lk) = make_float4(o[j][rq][0], o[j][rq][1], o[j][rq][2], o[j][rq][3]);
  }
}

int ffn_fused(const TtsFfnDesc& d, hipStream_t st) {
  TTS_CHECK_ARG(d.x && d.y && d.ln_g && d.ln_b && d.w && d.b2, "ffn_fused: null pointer");
  TTS_CHECK_ARG(d.channels == FF_C, "ffn_fused: model width %d unsupported (%d)", d.channels, FF_C);
  TTS_CHECK_ARG(d.hidden >= 32 && d.hidden % 32 == 0 && d.hidden <= 4096, "ffn_fused: hidden width %d must be a multiple of 32 (<= 4096)", d.hidden);
  TTS_CHECK_ARG(d.compute == TTS_COMPUTE_BF16 || d.compute == TTS_COMPUTE_F16, "ffn_fused: 16-bit MFMA configurations only (compute %d)", d.compute);
  TTS_CHECK_ARG((d.post_g == nullptr) == (d.post_b == nullptr), "ffn_fused: post_g and post_b go together");
  TTS_CHECK_ARG((d.ldx & 3) == 0 && (d.ldy & 3) == 0 && ((uintptr_t)d.x & 15) == 0 && ((uintptr_t)d.y & 15) == 0 && ((uintptr_t)d.w & 15) == 0 &&
                    ((uintptr_t)d.b2 & 15) == 0 && ((uintptr_t)d.ln_g & 15) == 0 && ((uintptr_t)d.ln_b & 15) == 0 &&
                    ((uintptr_t)d.post_g & 15) == 0 && ((uintptr_t)d.post_b & 15) == 0,
                "ffn_fused: rows, weights and vectors must be 16-byte aligned");
  if (d.rows <= 0) return TTS_OK;
  const size_t lds = FF_DEPTH * (size_t)FF_STAGE;
  static unsigned long long raised[2] = {0, 0};
  const bool f16 = d.compute == TTS_COMPUTE_F16;
  const void* k = f16 ? reinterpret_cast<const void*>(ffn_fused_kernel<true>) : reinterpret_cast<const void*>(ffn_fused_kernel<false>);
  if (raise_lds_limit(k, raised[f16 ? 1 : 0]) != hipSuccess) {
    set_error("ffn_fused: raising the dynamic LDS limit failed");
    return TTS_E_LAUNCH;
  }
  const int grid = (d.rows + FF_ROWS - 1) / FF_ROWS;
  if (f16) hipLaunchKernelGGL(ffn_fused_kernel<true>, dim3(grid), dim3(256), lds, st, d);
  else hipLaunchKernelGGL(ffn_fused_kernel<false>, dim3(grid), dim3(256), lds, st, d);
  return launch_status("ffn_fused");
}

}  // namespace tts

#ifdef FF_DIAG_CLOCK
extern "C" int tts_ffn_diag_clock(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(tts::g_ff_clock), sizeof(tts::g_ff_clock)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int tts_ffn_fused(const TtsFfnDesc* d, tts_stream_t stream) {
  if (!d) {
    tts::set_error("tts_ffn_fused: null descriptor");
    return TTS_E_ARG;
  }
  return tts::ffn_fused(*d, reinterpret_cast<hipStream_t>(stream));
}
