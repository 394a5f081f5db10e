// Relative-position multi-head self-attention on the matrix cores (fp32-input MFMA, exact fp32 products).
//
//   s[i, j] = ((q_i + u_h) . k_j  +  (q_i + v_h) . P_h[i - j]) / sqrt(dk),  softmax over the utterance's keys,  ctx = s . v
//
// Layers/Attention.py:159-198 (rel_shift :138-157 folded into the index i - j); the three contractions run on
// v_mfma_f32_32x32x2_f32 and the online softmax stays in registers:
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
#include <cstdlib>

#include "common.h"

namespace tts {

constexpr int AM_DK = 48, AM_QT = 128, AM_KT = 32, AM_PW = AM_QT + AM_KT - 1;  // 159 table rows per key tile
constexpr int AM_PITCH = AM_DK + 1;                                           // odd pitch: column reads hit distinct banks
constexpr int AM_GP = 33;                                                     // scratch pitch

// SPLIT (grids of a few workgroups - batch 1): the four wavefronts of a workgroup share ONE block of 32 queries and split the keys
// (wavefront w takes keys j0 + 32 w .. of every 128-key step), then their running (max, sum, O^T) meet in LDS and wavefront 0
// writes the block: 4x the workgroups, a quarter of the dependent fp32 MFMA chain per wavefront.  Same staging, same products.
template <bool SPLIT>
__global__ __launch_bounds__(256) void relpos_attention_mfma_kernel(const float* __restrict__ qkv, int ld_qkv,
                                                                    const float* __restrict__ ptab, int pmax,
                                                                    const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                                    float* __restrict__ ctx, int ld_ctx, int heads,
                                                                    const TtsTile* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) float am_lds[];
  constexpr int KROWS = SPLIT ? 4 * AM_KT : AM_KT;  // keys staged per step
  float* Ks = am_lds;                              // [KROWS][PITCH]
  float* Vs = Ks + KROWS * AM_PITCH;               // [KROWS][PITCH]
  float* Ps = Vs + KROWS * AM_PITCH;               // [PW+1][PITCH]
  float* Gs = Ps + (AM_PW + 1) * AM_PITCH;         // [4][64*GP] per-wave scratch

  const TtsTile t = tiles[SPLIT ? blockIdx.x >> 2 : blockIdx.x];
  const int h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
  const int n = t.seq_end - t.seq_begin;
  const int qbase = t.row0 - t.seq_begin + (SPLIT ? 32 * (int)(blockIdx.x & 3) : 0);  // local index of the workgroup's first query
  if (SPLIT && qbase >= n) return;            // (the whole workgroup: a query block behind the utterance)
  const int qw = SPLIT ? qbase : qbase + wave * 32;  // local index of this wave's first query
  const int kw = SPLIT ? wave * 32 : 0;       // this wave's first key row of a staged step
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

  for (int j0 = 0; j0 < n; j0 += KROWS) {
    __syncthreads();
    // ---- stage K, V (keys j0..j0+KROWS-1) and the table window (159 relative positions: 128 queries x 32 keys, or 32 x 128) ----
    for (int e = tid; e < KROWS * (AM_DK / 4); e += 256) {
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
    const int p0 = qbase - j0 - (KROWS - 1);
    for (int e = tid; e < (AM_PW + 1) * (AM_DK / 4); e += 256) {
      const int w = e / (AM_DK / 4), c4 = (e % (AM_DK / 4)) * 4;
      int pr = pmax - 1 + p0 + w;
      pr = pr < 0 ? 0 : (pr > 2 * pmax - 2 ? 2 * pmax - 2 : pr);  // only reached by masked keys / discarded queries
      const float4 pv = *reinterpret_cast<const float4*>(ptab + (size_t)pr * hd + h * AM_DK + c4);
      float* pd = Ps + w * AM_PITCH + c4;
      pd[0] = pv.x; pd[1] = pv.y; pd[2] = pv.z; pd[3] = pv.w;
    }
    __syncthreads();
    if (SPLIT && j0 + kw >= n) continue;  // (this wavefront's 32 keys lie behind the utterance; the barriers above are met by all)

    // ---- S^T = K (Q+u)^T : rows = keys, column = query ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < AM_DK / 2; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(kw + li) * AM_PITCH + 2 * kk + hi], qu[kk], s, 0, 0, 0);

    // ---- G^T = Pwin (Q+v)^T for this wave's 63-row sub-window (rows w = wave*32 + 0..63 of the staged window) ----
    f32x16 g0, g1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
    // (SPLIT: query i = qbase + li, key j = j0 + 32 wave + jj -> window row (i - j) - p0 = (96 - 32 wave) + (li - jj + 31))
    const float* pw = Ps + (SPLIT ? 96 - 32 * wave : wave * 32) * AM_PITCH;
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
      sc = (j0 + kw + jj < n) ? sc : -INFINITY;
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
      const float a0 = Vs[(kw + key) * AM_PITCH + li];
      const float a1 = Vs[(kw + key) * AM_PITCH + (li < 16 ? 32 + li : 47)];  // d = 32..47 valid; rows 48..63 of O^T are discarded
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], o1, 0, 0, 0);
    }
  }

  if constexpr (SPLIT) {
    // ---- the four partial results of the query block meet in LDS (over K / V, which nobody reads any more) ----
    __syncthreads();
    float* mg = am_lds;  // [4 waves][34][64 lanes]: O^T registers 0..31, running max, running sum
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mg[(wave * 34 + r) * 64 + lane] = o0[r];
      mg[(wave * 34 + 16 + r) * 64 + lane] = o1[r];
    }
    mg[(wave * 34 + 32) * 64 + lane] = m_run;
    mg[(wave * 34 + 33) * 64 + lane] = l_run;
    __syncthreads();
    if (wave != 0) return;
    float mw[4], m_all = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      mw[w] = mg[(w * 34 + 32) * 64 + lane];
      m_all = fmaxf(m_all, mw[w]);  // finite: wavefront 0 saw key 0
    }
    l_run = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = __expf(mw[w] - m_all);  // 0 for a wavefront that saw no key (max -inf, sum 0)
      l_run += mg[(w * 34 + 33) * 64 + lane] * f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] += mg[(w * 34 + r) * 64 + lane] * f;
        o1[r] += mg[(w * 34 + 16 + r) * 64 + lane] * f;
      }
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

// ---------------------------------------------------------------------------------------------------------------------
// The same attention with the three contractions on the 16-bit matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulation) for the
// 16-bit configurations: 13 MFMAs per 32-key tile and wavefront instead of 104 (the fp32-input MFMA has 1/16 of the rate; at 640
// frames per utterance the fp32 kernel was the largest item of the acoustic model: 2.7 of 11.7 ms).  q + u, q + v, K, V, the
// position table and the probabilities are rounded to fp16 (also in the bf16 configuration: scores of magnitude 10 need the 11-bit
// mantissa); scores, soft-max statistics and the output accumulate in fp32.  Same decomposition and the same transposed products
// as above; what changes is the operand layout (8 consecutive k per lane and k step):
//   * K and the table window are staged as fp16 rows [row][feature] (A operands: one ds_read_b128 per k step);
//   * (q + u)^T, (q + v)^T: 3 x 8 fp16 features of the lane's query per operand (B operands, 12 registers each);
//   * P^T comes straight out of the S^T accumulator registers again: registers 0..7 of a lane are keys {4 hi + i, 8 + 4 hi + i}
//     of the tile's first 16, registers 8..15 the same of the second 16 - that IS a B operand of the 16-deep product if the A
//     operand (V^T) lists its keys in that order, so V is staged transposed, [feature][key slot], with the keys permuted.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int AH_PK = AM_DK + 8;   // fp16 row pitch of K / table rows (112 B: 16-byte aligned, odd number of 16-byte slots)
constexpr int AH_PV = AM_KT + 8;   // fp16 row pitch of V^T (80 B)
constexpr int AH_SUB = 2;          // 32-key tiles staged per step (one pair of barriers and one exposed round of loads per 64 keys; the
                                   // soft-max still advances 32 keys at a time, so the arithmetic is that of single-tile steps)
constexpr int AH_PROWS = AM_PW + 1 + 32 * (AH_SUB - 1);  // table rows per step: 128 queries x 64 keys -> 191 relative positions

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void relpos_attention_f16_kernel(const float* __restrict__ qkv, int ld_qkv,
                                                                   const float* __restrict__ ptab, int pmax,
                                                                   const float* __restrict__ bias_u, const float* __restrict__ bias_v,
                                                                   float* __restrict__ ctx, int ld_ctx, int heads,
                                                                   const TtsTile* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) float am_lds[];
  unsigned short* Ks = reinterpret_cast<unsigned short*>(am_lds);   // [SUB * KT][PK]
  unsigned short* Vt = Ks + AH_SUB * AM_KT * AH_PK;                 // [SUB][DK][PV]: V^T, keys in operand order
  unsigned short* Ps = Vt + AH_SUB * AM_DK * AH_PV;                 // [PROWS][PK]
  float* Gs = reinterpret_cast<float*>(Ps + AH_PROWS * AH_PK);      // [4][64 * GP] per-wave scratch

  const TtsTile t = tiles[blockIdx.x];
  const int h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
  const int n = t.seq_end - t.seq_begin;
  const int qbase = t.row0 - t.seq_begin;
  const int qw = qbase + wave * 32;
  const int qi = qw + li;
  const int qrow = t.seq_begin + (qi < n ? qi : n - 1);
  const int hd = heads * AM_DK;
  float* gs = Gs + wave * 64 * AM_GP;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

  // B operands: features 16 ks + 8 hi .. + 7 of (q + u) / (q + v) of the lane's query, fp16
  bf16x8 qu[AM_DK / 16], qv[AM_DK / 16];
  {
    const float* qp = qkv + (size_t)qrow * ld_qkv + h * AM_DK;
#pragma unroll
    for (int ks = 0; ks < AM_DK / 16; ++ks) {
      const int d0 = 16 * ks + 8 * hi;
      const float4 q0 = *reinterpret_cast<const float4*>(qp + d0), q1 = *reinterpret_cast<const float4*>(qp + d0 + 4);
      const float4 u0 = *reinterpret_cast<const float4*>(bias_u + h * AM_DK + d0), u1 = *reinterpret_cast<const float4*>(bias_u + h * AM_DK + d0 + 4);
      const float4 v0 = *reinterpret_cast<const float4*>(bias_v + h * AM_DK + d0), v1 = *reinterpret_cast<const float4*>(bias_v + h * AM_DK + d0 + 4);
      const u32x4 pu = {pack16<true>(q0.x + u0.x, q0.y + u0.y), pack16<true>(q0.z + u0.z, q0.w + u0.w), pack16<true>(q1.x + u1.x, q1.y + u1.y),
                        pack16<true>(q1.z + u1.z, q1.w + u1.w)};
      const u32x4 pv = {pack16<true>(q0.x + v0.x, q0.y + v0.y), pack16<true>(q0.z + v0.z, q0.w + v0.w), pack16<true>(q1.x + v1.x, q1.y + v1.y),
                        pack16<true>(q1.z + v1.z, q1.w + v1.w)};
      qu[ks] = __builtin_bit_cast(bf16x8, pu);
      qv[ks] = __builtin_bit_cast(bf16x8, pv);
    }
  }
  f32x16 o0, o1;  // O^T: rows d (0..31 | 32..63), column = query
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const float scale = 1.0f / sqrtf((float)AM_DK);

  // The table window of step j00 + 64 is REQUESTED (global loads into registers) right behind the barrier that publishes step j00's
  // operands, so it flies under the matrix work of step j00; it is converted and written to LDS at the top of the next iteration.
  // (Loaded, converted and stored in one go, every step began with an exposed round of 15 loads per thread; K / V are 6 of them.)
  constexpr int NKV = AH_SUB * AM_KT * (AM_DK / 4) / 256, NPT = AH_PROWS * (AM_DK / 4) / 256;  // items per thread: 3 and 9
  static_assert(AH_SUB * AM_KT * (AM_DK / 4) % 256 == 0 && AH_PROWS * (AM_DK / 4) % 256 == 0, "staging items must tile the workgroup");
  float4 rp[NPT];
  auto request = [&](int j00) __attribute__((always_inline)) {
    // window row w <-> relative position p0 + w; the LAST staged tile needs the lowest positions: p0 = qbase - (j00 + 32 (SUB - 1)) - 31
    const int p0 = qbase - j00 - (AH_SUB * AM_KT - 1);
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      const int e = tid + q * 256;
      const int w = e / (AM_DK / 4), c4 = (e % (AM_DK / 4)) * 4;
      int pr = pmax - 1 + p0 + w;
      pr = pr < 0 ? 0 : (pr > 2 * pmax - 2 ? 2 * pmax - 2 : pr);  // only reached by masked keys / discarded queries
      rp[q] = *reinterpret_cast<const float4*>(ptab + (size_t)pr * hd + h * AM_DK + c4);
    }
  };
  auto commit = [&](int j00) __attribute__((always_inline)) {
    // ---- K (rows), V (transposed, keys in operand order) of AH_SUB key tiles (loaded here: with them prefetched too the kernel needs
    // more than the 256 registers two workgroups per CU leave a wavefront) and the prefetched table window, fp16 ----
    float4 rk[NKV], rv[NKV];
#pragma unroll
    for (int q = 0; q < NKV; ++q) {
      const int e = tid + q * 256;
      const int jk = e / (AM_DK / 4), c4 = (e % (AM_DK / 4)) * 4;
      const int jr = j00 + jk < n ? j00 + jk : n - 1;
      const float* base = qkv + (size_t)(t.seq_begin + jr) * ld_qkv + h * AM_DK + c4;
      rk[q] = *reinterpret_cast<const float4*>(base + hd);
      rv[q] = *reinterpret_cast<const float4*>(base + 2 * hd);
    }
#pragma unroll
    for (int q = 0; q < NKV; ++q) {
      const int e = tid + q * 256;
      const int jk = e / (AM_DK / 4), c4 = (e % (AM_DK / 4)) * 4;
      *reinterpret_cast<uint2*>(Ks + jk * AH_PK + c4) = make_uint2(pack16<true>(rk[q].x, rk[q].y), pack16<true>(rk[q].z, rk[q].w));
      // key jj of its tile -> slot of the 16-deep products: k step jj >> 4; inside it lane half ((jj >> 2) & 1), element (jj & 3) + 4 ((jj >> 3) & 1)
      const int jj = jk & (AM_KT - 1);
      const int slot = (jj & 16) + 8 * ((jj >> 2) & 1) + (jj & 3) + 4 * ((jj >> 3) & 1);
      unsigned short* vt = Vt + (jk / AM_KT) * AM_DK * AH_PV;
      vt[(c4 + 0) * AH_PV + slot] = f32_to_f16(rv[q].x);
      vt[(c4 + 1) * AH_PV + slot] = f32_to_f16(rv[q].y);
      vt[(c4 + 2) * AH_PV + slot] = f32_to_f16(rv[q].z);
      vt[(c4 + 3) * AH_PV + slot] = f32_to_f16(rv[q].w);
    }
#pragma unroll
    for (int q = 0; q < NPT; ++q) {
      const int e = tid + q * 256;
      const int w = e / (AM_DK / 4), c4 = (e % (AM_DK / 4)) * 4;
      *reinterpret_cast<uint2*>(Ps + w * AH_PK + c4) = make_uint2(pack16<true>(rp[q].x, rp[q].y), pack16<true>(rp[q].z, rp[q].w));
    }
  };
  request(0);
  for (int j00 = 0; j00 < n; j00 += AH_SUB * AM_KT) {
    __syncthreads();  // (every wavefront is done with the previous step's LDS operands)
    commit(j00);
    __syncthreads();
    // (unconditional: past the last step the clamped addresses re-read the last rows - a value defined on one side of a branch only is
    //  what the register allocator spills first)
    request(j00 + AH_SUB * AM_KT);

    for (int sub = 0; sub < AH_SUB; ++sub) {
    const int j0 = j00 + sub * AM_KT;
    if (j0 >= n) break;  // (wave-uniform: a key tile wholly behind the utterance)
    const unsigned short* Ksub = Ks + sub * AM_KT * AH_PK;
    const unsigned short* Vsub = Vt + sub * AM_DK * AH_PV;
    // ---- S^T = K (Q+u)^T, G^T = Pwin (Q+v)^T (this wave's 64-row sub-window): 3 + 6 MFMAs ----
    f32x16 s, g0, g1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; g0[r] = 0.f; g1[r] = 0.f; }
    // (tile `sub` of the step: its positions start 32 (SUB - 1 - sub) rows into the staged window)
    const unsigned short* pw = Ps + (32 * (AH_SUB - 1 - sub) + wave * 32) * AH_PK;
#pragma unroll
    for (int ks = 0; ks < AM_DK / 16; ++ks) {
      const bf16x8 ak = *reinterpret_cast<const bf16x8*>(Ksub + li * AH_PK + 16 * ks + 8 * hi);
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(pw + li * AH_PK + 16 * ks + 8 * hi);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(pw + (32 + li) * AH_PK + 16 * ks + 8 * hi);
      s = mfma16<true>(ak, qu[ks], s);
      g0 = mfma16<true>(a0, qv[ks], g0);
      g1 = mfma16<true>(a1, qv[ks], g1);
    }
    // through the per-wave scratch: G^T[w][i] at gs[w*GP + i]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int w = (r & 3) + 8 * (r >> 2) + 4 * hi;
      gs[w * AM_GP + li] = g0[r];
      gs[(32 + w) * AM_GP + li] = g1[r];
    }
    __builtin_amdgcn_wave_barrier();
    float m_tile = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jj = (r & 3) + 8 * (r >> 2) + 4 * hi;  // key row of register r
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
      const float p = __expf(s[r] - m_new);
      s[r] = p;
      l_tile += p;
    }
    l_tile += __shfl_xor(l_tile, 32, 64);
    l_run = l_run * corr + l_tile;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= corr; o1[r] *= corr; }

    // ---- O^T += V^T P^T: two 16-deep k steps; the B operand of step ks is registers 8 ks .. 8 ks + 7 of P^T as fp16 ----
    const u32x4 pb0 = {pack16<true>(s[0], s[1]), pack16<true>(s[2], s[3]), pack16<true>(s[4], s[5]), pack16<true>(s[6], s[7])};
    const u32x4 pb1 = {pack16<true>(s[8], s[9]), pack16<true>(s[10], s[11]), pack16<true>(s[12], s[13]), pack16<true>(s[14], s[15])};
    const int d1 = li < 16 ? 32 + li : 47;  // d = 32..47 valid; rows 48..63 of O^T are discarded
    o0 = mfma16<true>(*reinterpret_cast<const bf16x8*>(Vsub + li * AH_PV + 8 * hi), __builtin_bit_cast(bf16x8, pb0), o0);
    o1 = mfma16<true>(*reinterpret_cast<const bf16x8*>(Vsub + d1 * AH_PV + 8 * hi), __builtin_bit_cast(bf16x8, pb0), o1);
    o0 = mfma16<true>(*reinterpret_cast<const bf16x8*>(Vsub + li * AH_PV + 16 + 8 * hi), __builtin_bit_cast(bf16x8, pb1), o0);
    o1 = mfma16<true>(*reinterpret_cast<const bf16x8*>(Vsub + d1 * AH_PV + 16 + 8 * hi), __builtin_bit_cast(bf16x8, pb1), o1);
    // (the per-wave scratch `gs` is written again by the next tile: this wave's reads of it are done - same-wave DS order)
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

int relpos_attention_f16(const float* qkv, int ld_qkv, const float* ptab, int pmax, const float* bias_u, const float* bias_v, float* ctx,
                         int ld_ctx, int heads, int dk, const TtsTile* tiles, int n_tiles, int tile_rows, hipStream_t st) {
  TTS_CHECK_ARG(dk == AM_DK, "relpos_attention_f16: head dim %d unsupported (48 only)", dk);
  TTS_CHECK_ARG(tile_rows == AM_QT, "relpos_attention_f16: tile table must use %d rows, got %d", AM_QT, tile_rows);
  TTS_CHECK_ARG((ld_qkv & 3) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ptab & 15) == 0 && ((uintptr_t)bias_u & 15) == 0 &&
                    ((uintptr_t)bias_v & 15) == 0,
                "relpos_attention_f16: alignment");
  TTS_CHECK_ARG((ld_ctx & 3) == 0 && ((uintptr_t)ctx & 15) == 0, "relpos_attention_f16: ctx alignment");
  if (n_tiles == 0) return TTS_OK;
  const size_t lds = (size_t)(AH_SUB * AM_KT * AH_PK + AH_SUB * AM_DK * AH_PV + AH_PROWS * AH_PK) * 2 + (size_t)4 * 64 * AM_GP * sizeof(float);
  static unsigned long long lds_raised = 0;
  if (lds > 64 * 1024) (void)raise_lds_limit(reinterpret_cast<const void*>(relpos_attention_f16_kernel), lds_raised);
  hipLaunchKernelGGL(relpos_attention_f16_kernel, dim3(n_tiles, heads), dim3(256), lds, st, qkv, ld_qkv, ptab, pmax, bias_u, bias_v, ctx, ld_ctx,
                     heads, tiles);
  return launch_status("relpos_attention_f16");
}

int relpos_attention_mfma(const float* qkv, int ld_qkv, const float* ptab, int pmax, const float* bias_u, const float* bias_v,
                          float* ctx, int ld_ctx, int heads, int dk, const TtsTile* tiles, int n_tiles, int tile_rows, int flags, hipStream_t st) {
  TTS_CHECK_ARG(dk == AM_DK, "relpos_attention: head dim %d unsupported (48 only)", dk);
  TTS_CHECK_ARG(tile_rows == AM_QT, "relpos_attention(mfma): tile table must use %d rows, got %d", AM_QT, tile_rows);
  TTS_CHECK_ARG((ld_qkv & 3) == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)ptab & 15) == 0, "relpos_attention: alignment");
  TTS_CHECK_ARG((ld_ctx & 3) == 0 && ((uintptr_t)ctx & 15) == 0, "relpos_attention: ctx alignment");
  if (n_tiles == 0) return TTS_OK;
  // batch 1: 4 (encoder) or 20 (decoder, 640 frames) workgroups with a dependent chain of 104 fp32 MFMAs per 32 keys each -
  // the key-split form puts 4x the workgroups on the chip.  Its result differs from the plain form's at rounding-order level, so
  // the CALLER chooses (include/toucan_tts.h): never (flags 0: the fp32 layers of the 16-bit configurations, whose results are
  // bit for bit independent of the batch), on small grids (TTS_ATT_KEY_SPLIT: the decoder of the fp32 configuration) or always
  // (TTS_ATT_KEY_SPLIT_ALWAYS: its encoder - upstream of the rounded durations, one arithmetic whatever the batch)
  const bool split = ((flags & TTS_ATT_KEY_SPLIT_ALWAYS) || ((flags & TTS_ATT_KEY_SPLIT) && (long long)n_tiles * heads <= 64)) &&
                     std::getenv("TOUCAN_NO_ATTENTION_SPLIT") == nullptr;
  const size_t lds = (size_t)(2 * (split ? 4 : 1) * AM_KT * AM_PITCH + (AM_PW + 1) * AM_PITCH + 4 * 64 * AM_GP) * sizeof(float);
  if (split) {
    static unsigned long long lds_raised = 0;  // per device (common.h)
    if (lds > 64 * 1024) (void)raise_lds_limit(reinterpret_cast<const void*>(relpos_attention_mfma_kernel<true>), lds_raised);
    hipLaunchKernelGGL(relpos_attention_mfma_kernel<true>, dim3(4 * n_tiles, heads), dim3(256), lds, st, qkv, ld_qkv, ptab, pmax, bias_u, bias_v,
                       ctx, ld_ctx, heads, tiles);
  } else {
    static unsigned long long lds_raised = 0;
    if (lds > 64 * 1024) (void)raise_lds_limit(reinterpret_cast<const void*>(relpos_attention_mfma_kernel<false>), lds_raised);
    hipLaunchKernelGGL(relpos_attention_mfma_kernel<false>, dim3(n_tiles, heads), dim3(256), lds, st, qkv, ld_qkv, ptab, pmax, bias_u, bias_v,
                       ctx, ld_ctx, heads, tiles);
  }
  return launch_status("relpos_attention(mfma)");
}

}  // namespace tts
