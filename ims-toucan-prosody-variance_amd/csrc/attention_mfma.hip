// Relative-position multi-head self-attention on the matrix cores (fp32-input MFMA, exact fp32 products).
//
//   s[i, j] = ((q_i + u_h) . k_j  +  (q_i + v_h) . P_h[i - j]) / sqrt(dk),  softmax over the utterance's keys,  ctx = s . v
//
// Same contract as relpos_attention_kernel (attention.hip, Layers/Attention.py:159-198); this version moves the three
// contractions to v_mfma_f32_32x32x2_f32 and keeps the online softmax in registers:
//
//   * one workgroup = 4 wavefronts = 128 consecutive queries of one (utterance, head); each wavefront owns 32 queries.
//     K, V (32 keys) and the 159-row window of the position table are staged in LDS once per key tile for all 4 waves.
//   * everything is computed TRANSPOSED, keys (or table rows, or feature dims) on the MFMA row axis and the QUERY on the
//     column axis = the lane: S^T = K (Q+u)^T, G^T = Pwin (Q+v)^T, O^T = V^T P^T.  A lane therefore owns one query: its
//     16 accumulator registers of S^T are 16 keys of that query, row statistics are in-register reductions plus one
//     exchange with the partner half-wave (lane ^ 32), and the running output O^T is rescaled by a per-lane scalar.
//   * (Q+u)^T / (Q+v)^T are the MFMA B operands: lane (i, hi) holds feature 2kk+hi of its own query for every k-pair kk,
//     24 registers each, loaded once.
//   * the Toeplitz term needs G^T[i - j + 31][i]: a per-lane row index, so G^T goes through a per-wave LDS scratch
//     (64 x 32 floats) and each lane reads back its 16 entries (the rel_shift of Attention.py:138-157 is this index).
//   * P^T (the probabilities) is used as the B operand of O^T += V^T P^T straight from the S^T accumulator registers:
//     register r of a lane is key (r&3)+8(r>>2)+4hi, so "k-pair r" of the product pairs keys (base_r, base_r+4) and the
//     A operand V^T is simply read from LDS with that key order.  No transpose, no conversion.
//   * keys beyond the utterance get -inf; queries beyond it are computed and discarded.
#include "common.h"

namespace tts {

constexpr int AM_DK = 48, AM_QT = 128, AM_KT = 32, AM_PW = AM_QT + AM_KT - 1;  // 159 table rows per key tile
constexpr int AM_PITCH = AM_DK + 1;                                           // odd pitch: column reads hit distinct banks
constexpr int AM_GP = 33;                                                     // scratch pitch

__global__ __launch_bounds__(256) void relpos_attention_mfma_kernel(const float* __restrict__ qkv, int ld_qkv,
                                                                    const float* __restrict__ ptab, int pmax,
                                                                    const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                                    float* __restrict__ ctx, int ld_ctx, int heads,
                                                                    const TtsTile* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) float am_lds[];
  float* Ks = am_lds;                              // [KT][PITCH]
  float* Vs = Ks + AM_KT * AM_PITCH;               // [KT][PITCH]
  float* Ps = Vs + AM_KT * AM_PITCH;               // [PW+1][PITCH]
  float* Gs = Ps + (AM_PW + 1) * AM_PITCH;         // [4][64*GP] per-wave scratch

  const TtsTile t = tiles[blockIdx.x];
  const int h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
  const int n = t.seq_end - t.seq_begin;
  const int qbase = t.row0 - t.seq_begin;     // local index of the workgroup's first query
  const int qw = qbase + wave * 32;           // local index of this wave's first query
  const int qi = qw + li;                     // this lane's query (local)
  const int qrow = t.seq_begin + (qi < n ? qi : n - 1);
  const int hd = heads * AM_DK;
  float* gs = Gs + wave * 64 * AM_GP;

  // B operands: feature 2kk+hi of (q+u) / (q+v) of the lane's query
  float qu[AM_DK / 2], qv[AM_DK / 2];
  {
    const float* qp = qkv + (size_t)qrow * ld_qkv + h * AM_DK;
#pragma unroll
    for (int kk = 0; kk < AM_DK / 2; ++kk) {
      const int d = 2 * kk + hi;
      const float q = qp[d];
      qu[kk] = q + bias_u[h * AM_DK + d];
      qv[kk] = q + bias_v[h * AM_DK + d];
    }
  }
  f32x16 o0, o1;  // O^T: rows d (0..31 | 32..63), column = query
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const float scale = 1.0f / sqrtf((float)AM_DK);

  for (int j0 = 0; j0 < n; j0 += AM_KT) {
    __syncthreads();
    // ---- stage K, V (keys j0..j0+31) and the table window (relative positions qbase-j0-31 .. qbase-j0+127) ----
    for (int e = tid; e < AM_KT * (AM_DK / 4); e += 256) {
      const int jj = e / (AM_DK / 4), c4 = (e % (AM_DK / 4)) * 4;
      const int jr = j0 + jj < n ? j0 + jj : n - 1;
      const float* base = qkv + (size_t)(t.seq_begin + jr) * ld_qkv + h * AM_DK + c4;
      const float4 kv = *reinterpret_cast<const float4*>(base + hd);
      const float4 vv = *reinterpret_cast<const float4*>(base + 2 * hd);
      float* kd = Ks + jj * AM_PITCH + c4;
      float* vd = Vs + jj * AM_PITCH + c4;
      kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
      vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
    }
    const int p0 = qbase - j0 - (AM_KT - 1);
    for (int e = tid; e < (AM_PW + 1) * (AM_DK / 4); e += 256) {
      const int w = e / (AM_DK / 4), c4 = (e % (AM_DK / 4)) * 4;
      int pr = pmax - 1 + p0 + w;
      pr = pr < 0 ? 0 : (pr > 2 * pmax - 2 ? 2 * pmax - 2 : pr);  // only reached by masked keys / discarded queries
      const float4 pv = *reinterpret_cast<const float4*>(ptab + (size_t)pr * hd + h * AM_DK + c4);
      float* pd = Ps + w * AM_PITCH + c4;
      pd[0] = pv.x; pd[1] = pv.y; pd[2] = pv.z; pd[3] = pv.w;
    }
    __syncthreads();

    // ---- S^T = K (Q+u)^T : rows = keys, column = query ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < AM_DK / 2; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[li * AM_PITCH + 2 * kk + hi], qu[kk], s, 0, 0, 0);

    // ---- G^T = Pwin (Q+v)^T for this wave's 63-row sub-window (rows w = wave*32 + 0..63 of the staged window) ----
    f32x16 g0, g1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
    const float* pw = Ps + (wave * 32) * AM_PITCH;
#pragma unroll
    for (int kk = 0; kk < AM_DK / 2; ++kk) {
      g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pw[li * AM_PITCH + 2 * kk + hi], qv[kk], g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pw[(32 + li) * AM_PITCH + 2 * kk + hi], qv[kk], g1, 0, 0, 0);
    }
    // through the per-wave scratch: G^T[w][i] at gs[w*GP + i]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int w = (r & 3) + 8 * (r >> 2) + 4 * hi;
      gs[w * AM_GP + li] = g0[r];
      gs[(32 + w) * AM_GP + li] = g1[r];
    }
    // same-wave LDS write -> read: the DS unit executes one wave's accesses in order; the wave barrier only pins the compiler
    __builtin_amdgcn_wave_barrier();
    float m_tile = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;  // key row of register r
      // relative position i - j = (qw + li) - (j0 + jj); sub-window row w = (i - j) - (qw - j0 - 31) = li - jj + 31
      const float bd = gs[(li - jj + 31) * AM_GP + li];
      float sc = (s[r] + bd) * scale;
      sc = (j0 + jj < n) ? sc : -INFINITY;
      s[r] = sc;
      m_tile = fmaxf(m_tile, sc);
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_new = fmaxf(m_run, m_tile);  // finite: key j0 is valid
    const float corr = __expf(m_run - m_new);
    float l_tile = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(s[r] - m_new);
      s[r] = p;
      l_tile += p;
    }
    l_tile += __shfl_xor(l_tile, 32, 64);
    l_run = l_run * corr + l_tile;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }

    // ---- O^T += V^T P^T : k-pair r = keys (base_r, base_r + 4), B operand = register r of P^T ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float a0 = Vs[key * AM_PITCH + li];
      const float a1 = Vs[key * AM_PITCH + (li < 16 ? 32 + li : 47)];  // d = 32..47 valid; rows 48..63 of O^T are discarded
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], o1, 0, 0, 0);
    }
  }

  // ---- ctx[i][h*dk + d] = O^T[d][i] / l : transpose through the scratch so that every row is written contiguously ----
  const float inv = 1.0f / l_run;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * hi;
    gs[d * AM_GP + li] = o0[r] * inv;
    gs[(32 + d) * AM_GP + li] = o1[r] * inv;
  }
  __builtin_amdgcn_wave_barrier();
  // each lane pair (i, hi) writes half of row i (d = hi*24 .. hi*24+23) as six 16-byte stores
  if (qi < n) {
    float* op = ctx + (size_t)(t.seq_begin + qi) * ld_ctx + h * AM_DK + hi * 24;
#pragma unroll
    for (int d = 0; d < 24; d += 4) {
      float4 v;
      v.x = gs[(hi * 24 + d + 0) * AM_GP + li];
      v.y = gs[(hi * 24 + d + 1) * AM_GP + li];
      v.z = gs[(hi * 24 + d + 2) * AM_GP + li];
      v.w = gs[(hi * 24 + d + 3) * AM_GP + li];
      *reinterpret_cast<float4*>(op + d) = v;
    }
  }
}

int relpos_attention_mfma(const float* qkv, int ld_qkv, const float* ptab, int pmax, const float* bias_u, const float* bias_v,
                          float* ctx, int ld_ctx, int heads, int dk, const TtsTile* tiles, int n_tiles, int tile_rows, hipStream_t st) {
  TTS_CHECK_ARG(dk == AM_DK, "relpos_attention: head dim %d unsupported (48 only)", dk);
  TTS_CHECK_ARG(tile_rows == AM_QT, "relpos_attention(mfma): tile table must use %d rows, got %d", AM_QT, tile_rows);
  TTS_CHECK_ARG((ld_qkv & 3) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ptab & 15) == 0, "relpos_attention: alignment");
  TTS_CHECK_ARG((ld_ctx & 3) == 0 && ((uintptr_t)ctx & 15) == 0, "relpos_attention: ctx alignment");
  if (n_tiles == 0) return TTS_OK;
  const size_t lds = (size_t)(2 * AM_KT * AM_PITCH + (AM_PW + 1) * AM_PITCH + 4 * 64 * AM_GP) * sizeof(float);
  static unsigned long long lds_raised = 0;  // per device (common.h)
  if (lds > 64 * 1024) (void)raise_lds_limit(reinterpret_cast<const void*>(relpos_attention_mfma_kernel), lds_raised);
  hipLaunchKernelGGL(relpos_attention_mfma_kernel, dim3(n_tiles, heads), dim3(256), lds, st, qkv, ld_qkv, ptab, pmax, bias_u, bias_v, ctx,
                     ld_ctx, heads, tiles);
  return launch_status("relpos_attention(mfma)");
}

}  // namespace tts
