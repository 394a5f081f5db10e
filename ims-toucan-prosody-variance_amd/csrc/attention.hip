// Relative-position multi-head self-attention, flash style: the [N, N] score matrix and the
// [N, 2N-1] position matrix of the reference (Layers/Attention.py:189-196) never exist in memory.
//
//   s[i, j] = ((q_i + u_h) . k_j  +  (q_i + v_h) . P_h[i - j]) / sqrt(dk)
//
// where P_h[p] = (linear_pos(pe(p)))_h depends only on the relative position p = i - j: the reference's
// rel_shift (Attention.py:138-157) maps column N-1-i+j of the position matrix onto key j, and column m of
// that matrix is relative position N-1-m (PositionalEncoding.py:114-129).  The host therefore keeps ONE
// table ptab[pmax-1+p, :] per layer and the kernel indexes it Toeplitz-style.
//
// Decomposition: one wavefront per (64-query tile, head).  Lane = query row: q+u, q+v and the output
// accumulator live in registers; keys/values are staged 32 at a time in LDS and broadcast to all lanes
// (ds_read_b128, same address), the 95-row P window is staged with an odd pitch so that the 64 lanes,
// which need 64 consecutive rows, hit distinct banks.  Online softmax per lane, one rescale per key tile.
// Keys outside the utterance are skipped, which equals the reference's mask -> float-min -> softmax -> 0.
#include "common.h"

namespace tts {

template <int DK>
__global__ __launch_bounds__(64) void relpos_attention_kernel(const float* __restrict__ qkv, int ld_qkv,
                                                              const float* __restrict__ ptab, int pmax,
                                                              const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                              float* __restrict__ ctx, int ld_ctx, int heads,
                                                              const TtsTile* __restrict__ tiles) {
  constexpr int QT = 64, KT = 32, PP = DK + 1, PW = QT + KT - 1;
  __shared__ __attribute__((aligned(16))) float Ks[KT * DK];
  __shared__ __attribute__((aligned(16))) float Vs[KT * DK];
  __shared__ float Ps[PW * PP];

  const TtsTile t = tiles[blockIdx.x];
  const int h = blockIdx.y;
  const int lane = threadIdx.x;
  const int n = t.seq_end - t.seq_begin;
  const int qbase = t.row0 - t.seq_begin;  // local index of the tile's first query
  const int row = t.row0 + lane;
  const bool valid = row < t.seq_end;
  const int rrow = valid ? row : t.seq_end - 1;
  const int hd = heads * DK;

  float qu[DK], qv[DK], o[DK];
  {
    const float* qp = qkv + (size_t)rrow * ld_qkv + h * DK;
#pragma unroll
    for (int d = 0; d < DK; ++d) {
      const float q = qp[d];
      qu[d] = q + bias_u[h * DK + d];
      qv[d] = q + bias_v[h * DK + d];
      o[d] = 0.f;
    }
  }
  float m_run = -INFINITY, l_run = 0.f;
  const float scale = 1.0f / sqrtf((float)DK);

  for (int j0 = 0; j0 < n; j0 += KT) {
    __syncthreads();
    // stage K and V rows j0 .. j0+KT-1 of this utterance (zero beyond the end)
    for (int e = lane; e < KT * (DK / 4); e += 64) {
      const int jj = e / (DK / 4), c4 = (e % (DK / 4)) * 4;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (j0 + jj < n) {
        const float* base = qkv + (size_t)(t.seq_begin + j0 + jj) * ld_qkv + h * DK + c4;
        kv = *reinterpret_cast<const float4*>(base + hd);
        vv = *reinterpret_cast<const float4*>(base + 2 * hd);
      }
      *reinterpret_cast<float4*>(Ks + jj * DK + c4) = kv;
      *reinterpret_cast<float4*>(Vs + jj * DK + c4) = vv;
    }
    // P window: window row w <-> relative position p = qbase - j0 - (KT-1) + w
    const int p0 = qbase - j0 - (KT - 1);
    for (int e = lane; e < PW * (DK / 4); e += 64) {
      const int w = e / (DK / 4), c4 = (e % (DK / 4)) * 4;
      int pr = pmax - 1 + p0 + w;
      pr = pr < 0 ? 0 : (pr > 2 * pmax - 2 ? 2 * pmax - 2 : pr);  // only reached by masked keys / idle lanes
      const float4 pv = *reinterpret_cast<const float4*>(ptab + (size_t)pr * hd + h * DK + c4);
      float* dst = Ps + w * PP + c4;
      dst[0] = pv.x; dst[1] = pv.y; dst[2] = pv.z; dst[3] = pv.w;
    }
    __syncthreads();

    float s[KT];
    float m_tile = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < KT; ++jj) {
      const float* kr = Ks + jj * DK;
      const float* pr = Ps + (lane - jj + (KT - 1)) * PP;
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int d = 0; d < DK; ++d) {
        a = fmaf(qu[d], kr[d], a);
        b = fmaf(qv[d], pr[d], b);
      }
      const float sc = (j0 + jj < n) ? (a + b) * scale : -INFINITY;
      s[jj] = sc;
      m_tile = fmaxf(m_tile, sc);
    }
    const float m_new = fmaxf(m_run, m_tile);  // finite: key j0 is always valid
    const float corr = __expf(m_run - m_new);
    l_run *= corr;
#pragma unroll
    for (int d = 0; d < DK; ++d) o[d] *= corr;
#pragma unroll
    for (int jj = 0; jj < KT; ++jj) {
      const float p = expf(s[jj] - m_new);
      l_run += p;
      const float* vr = Vs + jj * DK;
#pragma unroll
      for (int d = 0; d < DK; ++d) o[d] = fmaf(p, vr[d], o[d]);
    }
    m_run = m_new;
  }
  if (valid) {
    const float inv = 1.0f / l_run;
    float* op = ctx + (size_t)row * ld_ctx + h * DK;
#pragma unroll
    for (int d = 0; d < DK; ++d) op[d] = o[d] * inv;
  }
}

int relpos_attention(const float* qkv, int ld_qkv, const float* ptab, int pmax, const float* bias_u, const float* bias_v,
                     float* ctx, int ld_ctx, int heads, int dk, const TtsTile* tiles, int n_tiles, int tile_rows, hipStream_t st) {
  TTS_CHECK_ARG(dk == 48, "relpos_attention: head dim %d unsupported (48 only)", dk);
  TTS_CHECK_ARG(tile_rows == 64, "relpos_attention: tile table must use 64 rows, got %d", tile_rows);
  TTS_CHECK_ARG((ld_qkv & 3) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ptab & 15) == 0, "relpos_attention: alignment");
  if (n_tiles == 0) return TTS_OK;
  hipLaunchKernelGGL(relpos_attention_kernel<48>, dim3(n_tiles, heads), dim3(64), 0, st, qkv, ld_qkv, ptab, pmax, bias_u, bias_v,
                     ctx, ld_ctx, heads, tiles);
  return launch_status("relpos_attention");
}

}  // namespace tts
