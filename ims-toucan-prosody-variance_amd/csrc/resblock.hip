// Fused vocoder residual step:   y = alpha * conv2(act2(conv1(act1(x)))) + res_scale * x   (+ y)
//
// One launch replaces the four ops of one dilation step of a BigVGAN AMP block (AMP.py:53-58: a1, c1, a2, c2, + x)
// or of a HiFiGAN residual block (ResidualBlock.py:93-97: LeakyReLU, conv, LeakyReLU, conv, + x).  The
// intermediate tensors (act1(x), conv1 output, act2 of it) exist only in LDS, so HBM sees one read of x (plus
// its halo) and one write of y instead of nine full-tensor passes.
//
// Decomposition (512 threads = 8 wavefronts; persistent workgroups that draw 224-row tiles of ONE utterance each from a
// work queue, see below):
//   * output tile: BM = 224 rows; conv1 is evaluated on M1 = 256 rows (BM + 16 each side: conv2's halo <= 5 and the
//     anti-alias filter's halo 6), wave w owns rows 32w..32w+31 and all C columns (C/32 accumulators of 32x32).
//   * the raw x window (M1 + 2*h1 + 12 rows, h1 = (k-1)/2*dil, all C channels) is staged once as fp16; the anti-aliased snake
//     runs in place on it with both FIR filters on the matrix cores (snake_mfma.h; LeakyReLU: element-wise while staging).
//   * both convs run TRANSPOSED (weights are the MFMA A operand): the accumulator has the output channel in the registers and
//     the frame on the lane.  conv1's result (+ bias) goes to LDS as 8-byte channel quadruples (t1, overlaying the window),
//     the second activation runs in place on t1, conv2 (dilation 1) reads it; waves 0..6 own the 224 output rows.
//   * weight slabs: C = 64 / 128 through a three-slot LDS ring filled by global_load_lds (two slabs in flight, counted vmcnt,
//     one raw barrier per tap); C = 32 keeps both convs' weights resident in LDS; C = 256 a register-staged two-slot ring.
//   * epilogue in registers: + bias2, alpha, + res_scale * x (the residual pieces are fetched during conv2), optional
//     accumulate, 8 / 16-byte row pieces straight from the accumulators - no LDS round trip, no barrier.
// Rows outside the utterance are zero after each activation (the reference zero-pads every conv per utterance).
// 16-bit MFMA (v_mfma_f32_32x32x16_bf16 / _f16) with fp32 accumulation; C in {32, 64, 128, 256}.
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <type_traits>
#include <vector>
#include <cstring>

#include "common.h"
#include "snake.h"
#include "snake_mfma.h"

namespace tts {


#ifndef RB_C32_WAVES
#define RB_C32_WAVES 4      // waves per SIMD the C = 32 instantiation is compiled for (tuning knob, see the launch bounds)
#endif
#ifndef RB_C32_PREFETCH
#define RB_C32_PREFETCH 1   // one-chunk-ahead input prefetch in the C = 32 snake (costs 8 registers)
#endif
#ifndef RB_C64_TPS
#define RB_C64_TPS 1        // taps per weight slab at C = 64 (tuning knob)
#endif
constexpr int RB_LEAD = 16;
// Wave priority inside the conv phases (matrix work) relative to every other phase (vector work): a co-resident workgroup's vector
// instructions then fill the issue slots a matrix instruction leaves free instead of delaying it (A/B knob; 0 = no priority change)
#ifndef RB_PRIO_CONV
#define RB_PRIO_CONV 0
#endif
#ifndef RB_PRIO_BASE
#define RB_PRIO_BASE 0
#endif
#define RB_SETPRIO(p_) do { if (RB_PRIO_CONV != RB_PRIO_BASE) __builtin_amdgcn_s_setprio(p_); } while (0)
#ifndef RB_STAGGER_DEFAULT
#define RB_STAGGER_DEFAULT 0   // (x 1 024 cycles; set from the measurement in DESIGN.md section 5)
#endif

// Workgroup barrier of this kernel: LDS traffic of the wavefront has landed (lgkmcnt), then s_barrier.  __syncthreads() also drains
// every global load in flight (vmcnt(0)) - here that would be the next tile's image, the residual and the next weight slab, all
// issued early on purpose.  Loads into registers are waited for where they are used; the kernel never reads back what it stored.
__device__ __forceinline__ void rb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Work queues of the persistent tile loop: one 16-word slot per launch in flight (zero when a launch starts; its last workgroup
// leaves it zero again).  Two launches may share a slot only if they can never be in flight together, so a slot belongs to a
// STREAM (launches of one stream run in order: launch_rb keeps a (device, stream) -> slot table), and every launch recorded into
// a HIP graph gets a slot of its own for the life of the process (a replay may run on any stream, beside eager launches or other
// graphs; the same graph never runs beside itself).  Slots are handed out once and never recycled: 4 MiB of device memory cover
// 65 536 of them, i.e. that many streams plus captured launches per process and device.
constexpr int RB_QUEUE_SLOTS = 1 << 16;
__device__ unsigned int g_rb_queue[RB_QUEUE_SLOTS][16];
// Arrival counters per CU (index: XCC_ID, then the shader-engine / shader-array / CU fields of HW_ID): the two workgroups a CU hosts
// at C <= 64 learn from the PARITY of their arrival which of the two they are, and the second one starts a quarter of a tile
// period late (`stagger`).  A tile is vector work (staging, snake), matrix work (conv), vector work (epilogue, snake), matrix work
// (conv): two co-resident workgroups that start together stay in lockstep - both on the vector pipe, then both on the matrix pipe,
// each time at half speed - while a quarter period apart one's convs run beside the other's snakes on every SIMD.  Speed only:
// never reset (two consecutive arrivals differ in parity whatever the count), and a wrong guess costs nothing but the overlap.
__device__ unsigned int g_rb_cu_arrivals[8 * 256];

// Timing diagnostics (tools/build_variant.sh NAME -DRB_DIAG_CLOCK[=wave]): every phase boundary of the matrix-core-snake path is
// stamped with s_memtime; one wavefront per workgroup stores its stamps into g_rb_trace, read back by
// tts_rb_diag_trace.  The shipped library carries no stamp.
#ifdef RB_DIAG_CLOCK
constexpr int RB_TRACE_MAX = 40000;
// per workgroup (plain stores, no atomics): [0..8] shader cycles per phase summed over its tiles, [9] tiles, [10] 100 MHz start,
// [11] 100 MHz end, [12] XCC_ID << 32 | HW_ID
__device__ unsigned long long g_rb_trace[RB_TRACE_MAX][16];
#define RB_STAMP(k_)                                                                             \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[k_])::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  } while (0)
#else
#define RB_STAMP(k_)
#endif
// C <= 128: conv1 on M1 = 256 rows (8 waves), 224 output rows, whole-C weight slabs.
// C == 256: conv1 on M1 = 128 rows (4 waves, 8 accumulators each), 96 output rows, 64-channel slabs (LDS: t1 alone is 66 KB).
template <int C>
struct RbCfg {
  // C = 64: 16 wavefronts on a 512-row window (480 output rows): ONE workgroup per CU does what two did - the same occupancy in
  // the vector phases - but shares every weight slab, reads half the halo, and a conv step carries two taps (32 matrix
  // instructions per SIMD and barrier instead of 16)
  static constexpr int M1 = C == 256 ? 128 : (C == 64 ? 512 : 256);
  static constexpr int BM = M1 - 2 * RB_LEAD;
  static constexpr int THREADS = 2 * M1;
  static constexpr int KC = C == 256 ? 64 : C;
  // snake over the act1(x) window: item = (8*NCH1 window rows, channel); the window is allocated in whole items so that
  // the streamed stores need no row bound
  static constexpr int NCH1 = (C == 32 || C == 256) ? 3 : 5;
  static constexpr int GR1 = 8 * NCH1;
  static constexpr int win_alloc(int h1) { return (M1 + 2 * h1 + GR1 - 1) / GR1 * GR1; }
  // matrix-core snake: 6 raw rows, the window rounded up to whole 16-row tiles, 6 raw rows
  // (+ 26: the last up-sampler window of a run reaches 31 rows past the run's end; those rows only meet zero taps)
  static constexpr int img_rows(int h1) { return (M1 + 2 * h1 + 15) / 16 * 16 + 32; }
  // The streamed snake stores come in two equivalent forms; which one the compiler schedules well differs per instantiation
  // (measured inside one run: C = 32 gains 10 % from the unguarded form; C = 64 and 256 lose 10 % to it - it hoists every
  // LDS address and spills): unguarded = no row bound (the window is padded to whole items) and one unsigned range compare.
  static constexpr bool UNGUARDED = C == 32;
  // taps per weight slab: a slab step costs one workgroup barrier, and at C = 32 / 64 a single tap is only 2 / 8 MFMAs per
  // wave: C = 32 moves a whole conv per step (<= 22 KB; lrelu step -8..-22 %, snake step -3..-11 % measured).  Two taps per
  // step at C = 64 measured neutral and doubled the scalar-register spills, so it keeps one
  static constexpr int TPS = C == 32 ? 11 : (C == 64 ? RB_C64_TPS : 1);
  // weight slabs through a three-slot LDS-DMA ring (see the kernel): slabs per tap
  static constexpr bool DMA = C == 64 || C == 128;
  static constexpr int SPLIT = C == 128 ? 2 : 1;
  static constexpr int DTAPS = C == 64 ? 2 : 1;       // taps per DMA slab (SPLIT == 1 only)
  static constexpr int ring_elems(int taps) { return DMA ? 3 * DTAPS * (KC / SPLIT) * C : 2 * (taps < TPS ? taps : TPS) * KC * C; }
};

// register-staged weight slab (up to 4 x 16 bytes per thread, named members so that it never becomes a stack array):
// unconditional (clamped) loads, so the prefetch is not fenced by a branch
template <int UNITS, int UPT, int THREADS>
struct RbSlab {
  static_assert(UPT >= 1 && UPT <= 8, "slab register staging holds at most 8 units per thread");
  uint4 r0, r1, r2, r3, r4, r5, r6, r7;
  int n_units;  // units the staged slab really holds (set by load, bounds the store)
  // n = 16-byte units this slab really holds (the last slab of a conv may carry fewer taps than the others)
  static __device__ __forceinline__ uint4 ld(const unsigned short* __restrict__ src, int tid, int q, int n) {
    int u = tid + q * THREADS;
    u = u < n ? u : n - 1;
    return *reinterpret_cast<const uint4*>(src + (size_t)u * 8);
  }
  static __device__ __forceinline__ void st(unsigned short* dst, int tid, int q, const uint4& v, int n) {
    const int u = tid + q * THREADS;
    if (u < n) *reinterpret_cast<uint4*>(dst + (size_t)u * 8) = v;
  }
  __device__ __forceinline__ void load(const unsigned short* __restrict__ src, int tid, int n) {
    n_units = n;
    r0 = ld(src, tid, 0, n);
    if constexpr (UPT > 1) r1 = ld(src, tid, 1, n);
    if constexpr (UPT > 2) r2 = ld(src, tid, 2, n);
    if constexpr (UPT > 3) r3 = ld(src, tid, 3, n);
    if constexpr (UPT > 4) r4 = ld(src, tid, 4, n);
    if constexpr (UPT > 5) r5 = ld(src, tid, 5, n);
    if constexpr (UPT > 6) r6 = ld(src, tid, 6, n);
    if constexpr (UPT > 7) r7 = ld(src, tid, 7, n);
  }
  __device__ __forceinline__ void store(unsigned short* dst, int tid) const {
    st(dst, tid, 0, r0, n_units);
    if constexpr (UPT > 1) st(dst, tid, 1, r1, n_units);
    if constexpr (UPT > 2) st(dst, tid, 2, r2, n_units);
    if constexpr (UPT > 3) st(dst, tid, 3, r3, n_units);
    if constexpr (UPT > 4) st(dst, tid, 4, r4, n_units);
    if constexpr (UPT > 5) st(dst, tid, 5, r5, n_units);
    if constexpr (UPT > 6) st(dst, tid, 6, r6, n_units);
    if constexpr (UPT > 7) st(dst, tid, 7, r7, n_units);
  }
};

// One tap of a transposed conv for one wavefront: acc[j] (32 output channels x 32 frames) += W_tap[32 j .., :] . act[frames + tap, :].
// ap: this lane's activation row at the tap's offset (+ 8 lk), bp: this lane's weight row of the slab (+ lk C 8).  All fragment
// reads of a group of k-slices are issued BEFORE the group's first MFMA (the compiler's own schedule waited for LDS four times per
// tap - read two fragments, wait, two MFMAs ... - so a tap cost ~1.5 k cycles for 256 cycles of matrix work per wavefront).
struct RbNoop {
  __device__ __forceinline__ void operator()() const {}
};
// after_reads(): called once, behind the first group's fragment reads and in front of its MFMAs (the DMA ring issues the next weight
// slab there: in front of the reads the compiler drains LDS before it lets them go)
template <int C, int KS, int TN, bool F16, class AfterReads = RbNoop>
__device__ __forceinline__ void rb_conv_tap(const unsigned short* ap, const unsigned short* bp, f32x16 (&acc)[TN], AfterReads after_reads = AfterReads()) {
  constexpr int GMAX = 48 / ((1 + TN) * 4);  // at most 48 fragment registers in flight
  constexpr int G = GMAX >= KS ? KS : (GMAX >= 4 ? 4 : (GMAX >= 2 ? 2 : 1));  // k-slices read together (a divisor of KS = 2, 4, 8)
#pragma unroll
  for (int g0 = 0; g0 < KS; g0 += G) {
    bf16x8 a[G], b[G][TN];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      a[g] = *reinterpret_cast<const bf16x8*>(ap + (g0 + g) * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[g][j] = *reinterpret_cast<const bf16x8*>(bp + ((size_t)(g0 + g) * 2 * C + j * 32) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (g0 == 0) {
      after_reads();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = mfma16<F16>(b[g][j], a[g], acc[j]);  // transposed product: accumulator row = output channel, column (lane) = frame
  }
}

// waves per SIMD the register allocation must leave room for: 2 / 2 / 1 / 1 workgroups per CU (C = 32 / 64 / 128 / 256)
// IOB: x / y are bf16 tensors in HBM (compile-time so that each instantiation carries one I/O path only)
// F16: the 16-bit element format everywhere in the kernel (LDS tiles, weights, 16-bit x / y) is IEEE fp16 instead of bf16
// MFIR: the two anti-aliased snakes run their FIR filters on the matrix cores (snake_mfma.h) instead of the register-streamed
//       VALU form; the activation window then carries 6 raw rows in front and behind (the filters' reach) and is transformed in place
template <int C, bool IOB, bool F16, bool MFIR>
// (C = 32: with the prefetching snake two spill-free workgroups per CU beat three at the 80-register cap by ~10 %)
__global__ __launch_bounds__(RbCfg<C>::THREADS, (C == 32 ? RB_C32_WAVES : (C == 64 ? 4 : (C == 128 ? 2 : 1)))) void resblock_step_kernel(const TtsResblockDesc d, const int queue_slot, const int stagger) {
  constexpr int RB_M1 = RbCfg<C>::M1, RB_BM = RbCfg<C>::BM, RB_THREADS = RbCfg<C>::THREADS;
  constexpr int KC = RbCfg<C>::KC;        // channels per weight slab (all of them for C <= 128: one step per tap, act1(x) staged once)
  constexpr int XP = KC + 8;              // act1(x) window pitch (bf16 elements; 16-B aligned rows, odd number of 16-B slots)
  constexpr int TP = C + 8;               // t1 pitch
  constexpr int TN = C / 32;              // 32-column accumulators per wave
  constexpr int NCH = C / KC;             // slabs per conv
  constexpr int TPS = RbCfg<C>::TPS;      // taps per weight slab
  constexpr int TAPW = KC * C;            // bf16 elements of one tap of a slab [KC/8][C][8]
  constexpr int SLAB = TPS * TAPW;        // one weight slab
  constexpr int UNITS = SLAB / 8;         // 16-byte units
  constexpr int UPT = (UNITS + RB_THREADS - 1) / RB_THREADS;
  // Weight slabs, C = 64 / 128 (DMA): a ring of THREE LDS slots filled by direct global -> LDS loads (global_load_lds_dwordx4: no
  // staging registers, no ds_write, nothing to wait for at the end of a step), two slabs in flight, counted vmcnt, one raw barrier
  // per step.  A slab is one tap at C = 64 (8 KB) and half a tap (64 of the 128 input channels, 16 KB) at C = 128: the ring is
  // 24 / 48 KB.  Steps run on across tiles (the ring position is carried), so the first two slabs of the next tile are in flight
  // behind the last conv2 steps.  C = 32 (both convs resident in LDS) and C = 256 keep the register-staged two-slot form.
  constexpr bool DMA = RbCfg<C>::DMA;
  constexpr int SPLIT = RbCfg<C>::SPLIT;            // slabs per tap
  constexpr int SLAB_K = KC / SPLIT;                // input channels per DMA slab
  constexpr int DTAPS = RbCfg<C>::DTAPS;            // taps per DMA slab
  static_assert(DTAPS == 1 || SPLIT == 1, "multi-tap slabs carry whole taps");
  constexpr int DMA_ELEMS = DTAPS * SLAB_K * C;     // 16-bit elements per ring slot
  constexpr int DMA_PW = DMA ? DMA_ELEMS / 8 / RB_THREADS : 1;  // global_load_lds instructions per wavefront and slab
  static_assert(!DMA || (DMA_ELEMS / 8) % RB_THREADS == 0, "a DMA slab is whole rounds of the workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

  // Persistent workgroups over a work queue: the grid is one residency's worth of workgroups, each takes tile after tile until
  // none is left (workgroups run at visibly different speeds - 765 .. 1013 us for the same 34 tiles was measured with a static
  // split - so the split is dynamic).  Neighbouring tiles share their halo rows, so every XCD owns a contiguous run of tiles
  // (q + (x < r) of them, q = n / 8, r = n % 8: a bijection for every n) and a workgroup draws from the run of the XCD it really
  // runs on (XCC_ID register): the tiles in flight on an XCD at any moment are neighbours, and its private L2 serves the halo
  // the neighbour already fetched.  A workgroup whose own run is used up takes from the other XCDs' runs: nothing guarantees that
  // every XCD hosts a workgroup (a grid smaller than 8, or a placement other than round-robin), so a run must never depend on
  // its own XCD for being processed.
  // queue[x] = next ticket of XCD x's run, queue[8] = workgroups that are done; the last one to finish zeroes the slot again.
  unsigned int* queue = g_rb_queue[queue_slot];
  const int q8 = d.n_tiles >> 3, r8 = d.n_tiles & 7;
  const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;  // HW_REG_XCC_ID[3:0]
  auto run_lo_of = [&](int x) __attribute__((always_inline)) { return x * q8 + (x < r8 ? x : r8); };
  auto run_n_of = [&](int x) __attribute__((always_inline)) { return q8 + (x < r8 ? 1 : 0); };
  // the tile a ticket of the own run stands for, -1 past the end of the run
  const int run_lo = run_lo_of(xcc), run_n = run_n_of(xcc);
  auto tile_of_ticket = [&](unsigned int ticket) __attribute__((always_inline)) { return ticket < (unsigned int)run_n ? run_lo + (int)ticket : -1; };
  // (one thread, own run exhausted) a tile of another XCD's run, -1 when every run is used up
  auto steal_tile = [&]() __attribute__((always_inline)) {
    int found = -1;
    for (int sft = 1; sft < 8 && found < 0; ++sft) {
      const int x = (xcc + sft) & 7, n = run_n_of(x);
      if (n > 0) {
        const unsigned int t2 = atomicAdd(&queue[x], 1u);
        if (t2 < (unsigned int)n) found = run_lo_of(x) + (int)t2;
      }
    }
    return found;
  };
  // (behind the queue's atomics the compiler no longer proves the tile table read-only and fetches entries with vector loads:
  // the values are wave-uniform, so they go back to scalar registers - every address below is built from them)
  auto load_tile = [&](int idx) __attribute__((always_inline)) {
    const TtsTile v = d.tiles[idx];
    TtsTile r;
    r.row0 = __builtin_amdgcn_readfirstlane(v.row0);
    r.seq_begin = __builtin_amdgcn_readfirstlane(v.seq_begin);
    r.seq_end = __builtin_amdgcn_readfirstlane(v.seq_end);
    r.seq_id = __builtin_amdgcn_readfirstlane(v.seq_id);
    return r;
  };
  auto retire = [&](int tid_) __attribute__((always_inline)) {
    if (tid_ == 0) {
      const unsigned int done = atomicAdd(&queue[8], 1u);
      if (done == gridDim.x - 1) {  // every other workgroup has drawn its last ticket before it counted itself here
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicExch(&queue[i], 0u);
      }
    }
  };

  RB_SETPRIO(RB_PRIO_BASE);
  const int h1 = (d.taps - 1) / 2 * d.dil, h2 = (d.taps - 1) / 2;
  const int win_rows = RB_M1 + 2 * h1;
  static_assert(!MFIR || (C <= 128 && KC == C), "the matrix-core snake needs whole-C slabs (one window pitch for xa and t1)");
  constexpr int PADR = MFIR ? 6 : 0;  // raw rows in front of / behind the window (reach of the up-sampler + decimator)
  // xa (act1(x) window) is dead once conv1 has finished, so t1 (conv1 output) overlays it
  const int img_rows = MFIR ? RbCfg<C>::img_rows(h1) : RbCfg<C>::win_alloc(h1);
  const size_t xa_elems = ((size_t)img_rows * XP + 7) & ~(size_t)7, t1_elems = (size_t)(RB_M1 + 2 * PADR) * TP;
  unsigned short* img = reinterpret_cast<unsigned short*>(lds_raw);         // [img_rows][XP]: PADR raw rows, the window, PADR raw rows
  unsigned short* xa = img + PADR * XP;                                     // [win_alloc][XP]
  unsigned short* t1 = xa;                                                  // [M1][TP]
  unsigned short* ws = img + (xa_elems > t1_elems ? xa_elems : t1_elems);   // [2][SLAB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;  // (prologue only: the tile loop re-derives them, see there)
  const bool snake = d.act == TTS_PRE_SNAKE;
  const unsigned short* __restrict__ xh = reinterpret_cast<const unsigned short*>(d.x);  // x viewed as bf16 (io_bf16)
  const int spc = DMA ? (d.taps + DTAPS - 1) / DTAPS * SPLIT : (d.taps + TPS - 1) / TPS;  // slab steps per channel chunk
  const int steps1 = NCH * spc, total_steps = 2 * steps1;
  const int slab_alloc = DMA ? DMA_ELEMS : (d.taps < TPS ? d.taps : TPS) * TAPW;  // LDS elements per slab buffer

  // constants every phase would otherwise fetch from global memory with the latency exposed (both biases, and the FIR operand
  // table of the matrix-core snake, which both sweeps load): copied into LDS once, visible after the staging barrier
  float* cst_b1 = reinterpret_cast<float*>(ws + (size_t)(DMA ? 3 : 2) * slab_alloc);    // [C]
  float* cst_b2 = cst_b1 + C;                                               // [C]
  float* cst_snk = cst_b2 + C;                                              // [4][C]: e^a1 / 2 pi, 1 / (e^b1 + 1e-9), same for the second snake (FIR_LDS)
  uint4* cst_fir = reinterpret_cast<uint4*>(cst_snk + (MFIR && C < 128 ? 4 * C : 0));  // [256] (FIR_LDS && snake)
  int* cst_ticket = reinterpret_cast<int*>(cst_fir + (MFIR && C < 128 ? 256 : 0));                   // [8] the next tile of this workgroup: index, -, -, -, its table entry
  unsigned int ticket_raw = 0;
  float f[12];  // (scalar loads: ahead of the first atomic)
#pragma unroll
  for (int k = 0; k < 12; ++k)
    f[k] = snake ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, d.filt[k]))) : 0.0f;  // (kept in scalar registers)
  if (tid == 0) ticket_raw = atomicAdd(&queue[xcc], 1u);  // (in flight while the constants below are fetched)
  unsigned int arrival = 0;
  if (stagger > 0 && tid == 0) arrival = atomicAdd(&g_rb_cu_arrivals[(xcc << 8) | ((__builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4)) & 255)], 1u);  // HW_ID[15:8]
  if (tid < C / 4) reinterpret_cast<float4*>(cst_b1)[tid] = reinterpret_cast<const float4*>(d.b1)[tid];
  else if (tid < C / 2) reinterpret_cast<float4*>(cst_b2)[tid - C / 4] = reinterpret_cast<const float4*>(d.b2)[tid - C / 4];
  // (C = 128 has no 4 KB of LDS left at 11 taps x dilation 5, but registers to spare: there the table's 16 registers stay live)
  constexpr bool FIR_LDS = MFIR && C < 128;
  if (FIR_LDS && snake && tid >= RB_THREADS - 256) cst_fir[tid - (RB_THREADS - 256)] = reinterpret_cast<const uint4*>(d.fir_tab)[tid - (RB_THREADS - 256)];
  SnakeFir fir_const;
  if (MFIR && !FIR_LDS && snake) fir_const.load_constants(d.fir_tab, lane);

  RbSlab<UNITS, UPT, RB_THREADS> wreg;
  // weight slab of step `step`: conv1 steps first, then conv2 (global layout [tap][C/8][C][8], one slab = one tap here)
  // (with TPS > 1 there is one channel chunk, so the taps of a slab are contiguous in global memory)
  static_assert(TPS == 1 || NCH == 1, "multi-tap slabs need whole-C slabs");
  auto slab_src = [&](int step) __attribute__((always_inline)) {
    const bool second = step >= steps1;
    const int sidx = second ? step - steps1 : step;
    const int chunk = sidx / spc, tap = (sidx % spc) * TPS;
    const unsigned short* W = reinterpret_cast<const unsigned short*>(second ? d.w2 : d.w1);
    return W + ((size_t)tap * (C / 8) + chunk * (KC / 8)) * C * 8;
  };
  auto slab_units = [&](int step) __attribute__((always_inline)) {
    const int tap = ((step >= steps1 ? step - steps1 : step) % spc) * TPS;
    const int nt = d.taps - tap < TPS ? d.taps - tap : TPS;
    return nt * (TAPW / 8);
  };
#ifdef RB_DIAG_NO_WLOAD  // (timing diagnostics only: weight slabs are fetched once, results are wrong)
#define load_slab(step_) do { if ((step_) == 0) wreg.load(slab_src(step_), tid, slab_units(step_)); } while (0)
#define store_slab(buf_) wreg.store(ws + (size_t)(buf_) * slab_alloc, tid)
#else
#define load_slab(step_) wreg.load(slab_src(step_), tid, slab_units(step_))
#define store_slab(buf_) wreg.store(ws + (size_t)(buf_) * slab_alloc, tid)
#endif
  // DMA ring: slab `step` of a tile (conv1's slabs first, then conv2's; tap-major, the input-channel halves of a tap in order) into
  // ring slot `slot`: thread t moves the 16-byte units t, t + THREADS, ... (a wavefront's 64 units are contiguous: 1 KB per instruction)
  auto dma_issue = [&](int step, int slot, int tid_) __attribute__((always_inline)) {
    if constexpr (DMA) {
      const bool second = step >= steps1;
      const int sidx = second ? step - steps1 : step;
      const int tap = sidx / SPLIT * DTAPS, half = sidx % SPLIT;
      const unsigned short* W = reinterpret_cast<const unsigned short*>(second ? d.w2 : d.w1);
      const unsigned short* src = W + ((size_t)tap * (C / 8) + half * (SLAB_K / 8)) * C * 8;
      unsigned short* dst = ws + (size_t)slot * DMA_ELEMS;
      const int wv = __builtin_amdgcn_readfirstlane(tid_ >> 6), ln = tid_ & 63;
      // (a conv's last slab may carry fewer taps: the units behind them re-read the slab's last valid unit - never past the weights)
      const int valid = (d.taps - tap < DTAPS ? d.taps - tap : DTAPS) * (SLAB_K / 8) * C;
#pragma unroll
      for (int q = 0; q < DMA_PW; ++q) {
        const int u0 = q * RB_THREADS + wv * 64;
        int u = u0 + ln;
        if constexpr (DTAPS > 1) u = u < valid ? u : valid - 1;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)u * 8),
                                         (void __attribute__((address_space(3)))*)(dst + (size_t)u0 * 8), 16, 0, 0);
      }
    }
  };
  // start of a step: this wavefront's part of the step's slab has landed (everything but the newest `keep` vector-memory operations
  // is done: the next slab's DMA_PW loads - issued one step ago - are the youngest, plus whatever the previous step issued behind
  // them), its LDS traffic is done, then the barrier: every wavefront's part has landed and nobody still reads the slot the next
  // DMA overwrites
  auto dma_step_barrier = [&](auto keep) __attribute__((always_inline)) {
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(keep)::value) : "memory");
    rb_barrier();
  };
  int ring_pos = 0;  // ring slot of the current step (carried across tiles)
#ifdef RB_DIAG_CLOCK
  unsigned long long stamp[10], phase_sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  int n_iter = 0;
#endif

  // ---- matrix-core snake: staging of the raw x image (16-byte units; row r of the image <-> local frame l0 - LEAD - h1 - 6 + r).
  // 16-bit x: every unit of the image fits one round of PERX loads per thread, and that round is issued ONE TILE AHEAD (before
  // conv2 of the previous tile; before the weight and constant loads for the first tile) - its memory latency never shows.
  constexpr int Q8 = C / 8;
  constexpr int PERX = !MFIR ? 1 : (IOB ? (C == 128 ? 11 : (C == 64 ? 6 : 3)) : 4);
  // one-tile-ahead prefetch into registers: C = 128 only (one workgroup per CU, 256 registers per lane).  C <= 64 and fp32 x
  // load the image at the top of its tile.  (Round 2 pulled the next image towards the L2 with one 4-byte touch per line: the
  // counters showed every touched line fetched twice - FETCH_SIZE 1.7-1.8 x algorithmic for C = 64 - and no time gained;
  // without it 1.04-1.05 x at the same speed, profiles/r03_pmc_resblock_traffic.json.)
  constexpr bool XPF = MFIR && IOB && C == 128;
  const int img_units = img_rows * Q8;
  uint4 xv[PERX], xv2[(MFIR && !IOB) ? PERX : 1];
  auto stage_issue = [&](const TtsTile& tl, int base) __attribute__((always_inline)) {
    const int Tn = tl.seq_end - tl.seq_begin, fr0 = tl.row0 - tl.seq_begin - RB_LEAD - h1 - PADR;
#pragma unroll
    for (int p = 0; p < PERX; ++p) {
      int e = base + p * RB_THREADS;
      e = e < img_units ? e : img_units - 1;
      const int r = e / Q8, c8 = (e % Q8) * 8;
      const int t = fr0 + r;
      const int tc = t < 0 ? 0 : (t > Tn - 1 ? Tn - 1 : t);  // (rows outside the utterance are zeroed when they are stored)
      if constexpr (IOB) {
        xv[p] = *reinterpret_cast<const uint4*>(xh + (size_t)(tl.seq_begin + tc) * d.ldx + c8);
      } else {
        xv[p] = *reinterpret_cast<const uint4*>(d.x + (size_t)(tl.seq_begin + tc) * d.ldx + c8);
        xv2[p] = *reinterpret_cast<const uint4*>(d.x + (size_t)(tl.seq_begin + tc) * d.ldx + c8 + 4);
      }
    }
  };
  auto stage_commit = [&](const TtsTile& tl, int base) __attribute__((always_inline)) {
    const int Tn = tl.seq_end - tl.seq_begin, fr0 = tl.row0 - tl.seq_begin - RB_LEAD - h1 - PADR;
#pragma unroll
    for (int p = 0; p < PERX; ++p) {
      const int e = base + p * RB_THREADS;
      if (e >= img_units) continue;
      const int r = e / Q8, c8 = (e % Q8) * 8;
      const int t = fr0 + r;
      uint4 o;
      if constexpr (IOB && F16) {
        o = xv[p];
      } else if constexpr (IOB) {  // bf16 -> fp16 (exact for |x| in fp16's range)
        const unsigned int w4[4] = {xv[p].x, xv[p].y, xv[p].z, xv[p].w};
        unsigned int o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = pack16<true>(bf16_to_f32(w4[q] & 0xFFFF), bf16_to_f32(w4[q] >> 16));
        o = make_uint4(o4[0], o4[1], o4[2], o4[3]);
      } else {
        auto fb = [](unsigned int b) { return __builtin_bit_cast(float, b); };
        o = make_uint4(pack16<true>(fb(xv[p].x), fb(xv[p].y)), pack16<true>(fb(xv[p].z), fb(xv[p].w)),
                       pack16<true>(fb(xv2[p].x), fb(xv2[p].y)), pack16<true>(fb(xv2[p].z), fb(xv2[p].w)));
      }
      // frames outside the utterance and the surplus rows behind the window: zeros
      if (t < 0 || t >= Tn || r >= win_rows + 2 * PADR) o = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(img + r * XP + c8) = o;
    }
  };
  if (tid == 0) {
    int first = tile_of_ticket(ticket_raw);
    if (first < 0) first = steal_tile();
    cst_ticket[0] = first;
    cst_ticket[1] = (int)(arrival & 1u);
  }
  rb_barrier();
  int cur_tile = __builtin_amdgcn_readfirstlane(cst_ticket[0]);
  if (stagger > 0 && __builtin_amdgcn_readfirstlane(cst_ticket[1]) != 0) {  // the CU's second workgroup: start `stagger` x 1 024 cycles late
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  if (cur_tile < 0) {  // (more workgroups than tiles)
    retire(tid);
    return;
  }
  TtsTile tile = load_tile(cur_tile);
  if (XPF && snake) stage_issue(tile, tid);

  if constexpr (DMA) {
    dma_issue(0, 0, tid);
    dma_issue(1, 1, tid);
  } else {
    load_slab(0);
    store_slab(0);
  }
  // Two slab steps in all (C = 32: a whole conv per slab) = both convs' weights fit the ring: after the first tile of this
  // workgroup they are simply there - no loads, no staging stores, no waits for them in any later tile.
  const bool weights_resident = total_steps == 2;
  bool first_tile = true;

  f32x16 acc[TN];

  // matrix-core snake: a lane's channel is the same in every tile and both sweeps: e^alpha / 2 pi and 1 / (e^beta + 1e-9) once
  // (C <= 64: through LDS, four registers less across the tile loop; C = 128: in registers, no LDS left)
  float er1 = 0.0f, ib1 = 0.0f, er2 = 0.0f, ib2 = 0.0f;
  if (MFIR && snake) {
    if constexpr (FIR_LDS) {
      if (tid < C) {
        cst_snk[tid] = expf(d.alpha1[tid]) * 0.15915494309189535f;
        cst_snk[C + tid] = 1.0f / (expf(d.beta1[tid]) + 1e-9f);
        cst_snk[2 * C + tid] = expf(d.alpha2[tid]) * 0.15915494309189535f;
        cst_snk[3 * C + tid] = 1.0f / (expf(d.beta2[tid]) + 1e-9f);
      }
    } else {
      const int chn = (wave % (C / 16)) * 16 + (lane & 15);
      er1 = expf(d.alpha1[chn]) * 0.15915494309189535f;
      ib1 = 1.0f / (expf(d.beta1[chn]) + 1e-9f);
      er2 = expf(d.alpha2[chn]) * 0.15915494309189535f;
      ib2 = 1.0f / (expf(d.beta2[chn]) + 1e-9f);
    }
  }

  const int tid_prologue = tid;
  for (;;) {
  // Everything a lane derives from its index (LDS addresses of every fragment, row / channel predicates ...) is the same in
  // every tile; hoisted out of this loop it would occupy a few hundred registers.  An opaque copy of the thread index per
  // iteration keeps those values where they are used.
  // The same holds between the phases of one tile: each section below derives its own copies, so that the fragment addresses
  // of conv1 are not kept alive through the sweeps and conv2 (and vice versa).
#define RB_LANE_IDS(n_)                      \
  int tid_opaque##n_ = tid_prologue;         \
  asm volatile("" : "+v"(tid_opaque##n_));   \
  const int tid = tid_opaque##n_, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lk = lane >> 5; \
  (void)lrow; (void)lk; (void)wave
  RB_LANE_IDS(1);
  bool ticket_drawn = false;
  int next_resolved = -1;
  int4 next_entry = make_int4(0, 0, 0, 0);
  if (!(MFIR && snake)) {  // (the matrix-core snake path draws behind its image loads, below)
    ticket_drawn = true;
    if (tid == 0) ticket_raw = atomicAdd(&queue[xcc], 1u);
  }
  const int T = tile.seq_end - tile.seq_begin;
  const int l0 = tile.row0 - tile.seq_begin;  // local frame of the tile's first output row
  RB_STAMP(0);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

  // ------------------------------------------------------------------ conv1 over act1(x)
  int step = 0;
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * KC;
    if (ch > 0) rb_barrier();
    // window row j <-> local frame l0 - LEAD - h1 + j
    const int wbase = l0 - RB_LEAD - h1;
    if (MFIR && snake) {
      if constexpr (MFIR) {
        // raw x (as fp16, zero outside the utterance) into the whole image, then both FIR filters of act1 on the matrix cores, in place
        for (int base = tid; base < img_units; base += RB_THREADS * PERX) {
          if (!XPF || base != tid) stage_issue(tile, base);
          if (!ticket_drawn) {  // the next tile's ticket: asked for behind the image loads (returns come in order), used after conv1
            ticket_drawn = true;
            if (tid == 0) ticket_raw = atomicAdd(&queue[xcc], 1u);
          }
          stage_commit(tile, base);
        }
        constexpr int CB = C / 16, SEG = (RB_THREADS / 64) / CB;
        const int cb = wave % CB, seg = wave / CB;
        const int tiles_total = (win_rows + 15) / 16;
        const int t_lo = seg * tiles_total / SEG, t_hi = (seg + 1) * tiles_total / SEG;
        SnakeFir fir;
        fir.img = img; fir.pitch = XP;
        fir.frame0 = wbase - PADR; fir.T = T; fir.ch0 = cb * 16;
        fir.row_begin = __builtin_amdgcn_readfirstlane(16 * t_lo); fir.n_tiles = __builtin_amdgcn_readfirstlane(t_hi - t_lo);
        RB_STAMP(1);
        rb_barrier();
        fir.er = FIR_LDS ? cst_snk[cb * 16 + (lane & 15)] : er1;
        fir.inv_b = FIR_LDS ? cst_snk[C + cb * 16 + (lane & 15)] : ib1;
        FirTaps ft;
#pragma unroll
        for (int k = 0; k < 12; ++k) ft.v[k] = f[k];
#ifndef RB_DIAG_NO_SWEEP1  // (timing diagnostics only: tools/build_variant.sh NAME -DRB_DIAG_NO_SWEEP1)
        if constexpr (FIR_LDS) fir.load_constants(cst_fir, lane);
        else fir.copy_constants(fir_const);
        fir.begin(lane);
        rb_barrier();
        RB_STAMP(2);
        fir.template sweep<F16>(ft, lane);
        RB_STAMP(3);
#endif
      }
    } else if (snake) {
      // anti-aliased snake while staging: item = (8*NCH1 window rows, channel), streamed so that only the first chunk pays the halo
      constexpr int NCH1 = RbCfg<C>::NCH1, GR = RbCfg<C>::GR1;
      const int items = ((win_rows + GR - 1) / GR) * KC;
      for (int it = tid; it < items; it += RB_THREADS) {
        const int chl = it % KC, wr0 = (it / KC) * GR;
        const int cg = c0 + chl;
        const int t0 = wbase + wr0;
        const bool live = t0 + GR - 1 >= 0 && t0 < T;
        if (live) {
          const float ea = expf(d.alpha1[cg]), ib = 1.0f / (expf(d.beta1[cg]) + 1e-9f);
          auto st = [&](int i, float v) {
            if constexpr (RbCfg<C>::UNGUARDED)
              xa[(wr0 + i) * XP + chl] = to16<F16>((unsigned)(t0 + i) < (unsigned)T ? v : 0.0f);
            else if (wr0 + i < win_rows)
              xa[(wr0 + i) * XP + chl] = to16<F16>((t0 + i >= 0 && t0 + i < T) ? v : 0.0f);
          };
          if constexpr (IOB)
            snake_stream<NCH1, C != 32 || RB_C32_PREFETCH>([&](int q) { return from16<F16>(xh[(size_t)(tile.seq_begin + q) * d.ldx + cg]); }, st, T, t0, f, ea, ib);
          else
            snake_stream<NCH1, C != 32 || RB_C32_PREFETCH>([&](int q) { return d.x[(size_t)(tile.seq_begin + q) * d.ldx + cg]; }, st, T, t0, f, ea, ib);
        } else {
          for (int i = 0; i < GR; ++i) xa[(wr0 + i) * XP + chl] = 0;
        }
      }
    } else if constexpr (IOB) {
      // bf16 input: 8 channels per 16-byte load; PER loads in flight per thread (clamped addresses, no branches)
      constexpr int Q8 = KC / 8, PER = 4;
      const int total = win_rows * Q8;
      for (int base = tid; base < total; base += RB_THREADS * PER) {
        uint4 v[PER];
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          int e = base + p * RB_THREADS;
          e = e < total ? e : total - 1;
          const int wr = e / Q8, c8 = (e % Q8) * 8;
          const int t = wbase + wr;
          const int tc = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
          v[p] = *reinterpret_cast<const uint4*>(xh + (size_t)(tile.seq_begin + tc) * d.ldx + c0 + c8);
          if (t < 0 || t >= T) v[p] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          int e = base + p * RB_THREADS;
          e = e < total ? e : total - 1;
          const int wr = e / Q8, c8 = (e % Q8) * 8;
          const unsigned int w4[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
          unsigned int o4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float lo = from16<F16>(w4[q] & 0xFFFF), hi2 = from16<F16>(w4[q] >> 16);
            lo = lo > 0.f ? lo : lo * d.slope;
            hi2 = hi2 > 0.f ? hi2 : hi2 * d.slope;
            o4[q] = pack16<F16>(lo, hi2);
          }
          *reinterpret_cast<uint4*>(xa + wr * XP + c8) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        }
      }
    } else {
      // PER independent 16-byte loads per thread in flight before the first one is consumed (clamped addresses, no branches)
      constexpr int Q4 = KC / 4, PER = 4;
      const int total = win_rows * Q4;
      for (int base = tid; base < total; base += RB_THREADS * PER) {
        float4 v[PER];
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          int e = base + p * RB_THREADS;
          e = e < total ? e : total - 1;
          const int wr = e / Q4, c4 = (e % Q4) * 4;
          const int t = wbase + wr;
          const int tc = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
          v[p] = *reinterpret_cast<const float4*>(d.x + (size_t)(tile.seq_begin + tc) * d.ldx + c0 + c4);
          if (t < 0 || t >= T) v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int p = 0; p < PER; ++p) {
          int e = base + p * RB_THREADS;
          e = e < total ? e : total - 1;
          const int wr = e / Q4, c4 = (e % Q4) * 4;
          *reinterpret_cast<uint2*>(xa + wr * XP + c4) =
              make_uint2(pack16<F16>(v[p].x > 0.f ? v[p].x : v[p].x * d.slope, v[p].y > 0.f ? v[p].y : v[p].y * d.slope),
                         pack16<F16>(v[p].z > 0.f ? v[p].z : v[p].z * d.slope, v[p].w > 0.f ? v[p].w : v[p].w * d.slope));
        }
      }
    }
    if (ch == 0) {
      // the ticket is back (it was drawn ahead of this tile's image loads, and returns come in order): the tile entry it
      // stands for is fetched now and published after conv1 - no wavefront ever waits for either.  (Every lane computes it - only
      // thread 0 drew a ticket, the others resolve ticket 0 - so that nothing here is defined on one side of a branch.)
      next_resolved = tile_of_ticket(ticket_raw);
      if (next_resolved < 0 && tid == 0) next_resolved = steal_tile();  // (rare: the end of this XCD's run)
      next_entry = *reinterpret_cast<const int4*>(d.tiles + (next_resolved >= 0 ? next_resolved : cur_tile));
    }
    RB_SETPRIO(RB_PRIO_CONV);
    if constexpr (DMA) {
      for (int tap0 = 0; tap0 < d.taps; tap0 += DTAPS) {
#pragma unroll
        for (int half = 0; half < SPLIT; ++half, ++step) {
          dma_step_barrier(std::integral_constant<int, DMA_PW>{});
          int nxt = step + 2;  // (wraps into the next tile's first slabs; issued even without a next tile - drained before the kernel ends)
          nxt = nxt >= total_steps ? nxt - total_steps : nxt;
          const unsigned short* wb = ws + (size_t)ring_pos * DMA_ELEMS;
          const unsigned short* ap = xa + (wave * 32 + lrow + tap0 * d.dil) * XP + half * SLAB_K + lk * 8;
          rb_conv_tap<C, SLAB_K / 16, TN, F16>(ap, wb + (lk * C + lrow) * 8, acc,
                                               [&]() __attribute__((always_inline)) { dma_issue(nxt, ring_pos >= 1 ? ring_pos - 1 : 2, tid); });
          if constexpr (DTAPS > 1) {
#pragma unroll
            for (int tt = 1; tt < DTAPS; ++tt)
              if (tap0 + tt < d.taps) rb_conv_tap<C, SLAB_K / 16, TN, F16>(ap + tt * d.dil * XP, wb + (size_t)tt * SLAB_K * C + (lk * C + lrow) * 8, acc);
          }
          ring_pos = ring_pos == 2 ? 0 : ring_pos + 1;
        }
      }
    } else {
    for (int tap0 = 0; tap0 < d.taps; tap0 += TPS, ++step) {
      rb_barrier();
      const bool restage = !weights_resident || first_tile;
      if (restage) load_slab(step + 1);  // step + 1 < total_steps always holds here (conv2 follows)
      const int nt = d.taps - tap0 < TPS ? d.taps - tap0 : TPS;
      for (int tt = 0; tt < nt; ++tt) {
        const unsigned short* wb = ws + (size_t)(step & 1) * slab_alloc + tt * TAPW;
        const int tap = tap0 + tt;
        rb_conv_tap<C, KC / 16, TN, F16>(xa + (wave * 32 + lrow + tap * d.dil) * XP + lk * 8, wb + (lk * C + lrow) * 8, acc);
      }
      if (restage) store_slab((step + 1) & 1);
    }
    }
    RB_SETPRIO(RB_PRIO_BASE);
  }

  if (tid == 0) {  // (read by everyone behind the next barriers)
    cst_ticket[0] = next_resolved;
    *reinterpret_cast<int4*>(cst_ticket + 4) = next_entry;
  }
  // ------------------------------------------------------------------ t1 = conv1 + bias (LeakyReLU applied here), bf16 in LDS
  int next_tile = -1;
  bool has_next = false;
  TtsTile tile_next = tile;
  {
  RB_LANE_IDS(2);
  // t1 row i <-> local frame l0 - LEAD + i
  RB_STAMP(4);
  rb_barrier();  // every wave is done reading xa (t1 overlays it)
  {
    // Both convs run transposed (weights as the A operand): a lane owns ONE frame and its accumulator registers hold four
    // consecutive output channels per group of four - a 16-bit quadruple is one 8-byte LDS store, and "outside the utterance"
    // is one predicate per lane instead of one per register.
    const int i = wave * 32 + lrow;      // t1 row of this lane
    const int t = l0 - RB_LEAD + i;
    const bool live = t >= 0 && t < T;   // act2 output outside the utterance is conv2's zero padding
    const bool as_f16 = F16 || (MFIR && snake);  // (the matrix-core snake reads fp16)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n0 = j * 32 + 8 * rq + 4 * lk;
        const float4 bb = *reinterpret_cast<const float4*>(cst_b1 + n0);
        float v[4] = {acc[j][4 * rq] + bb.x, acc[j][4 * rq + 1] + bb.y, acc[j][4 * rq + 2] + bb.z, acc[j][4 * rq + 3] + bb.w};
        if (!snake) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], v[q] * d.slope);  // LeakyReLU (0 < slope < 1)
        }
        uint2 o = as_f16 ? make_uint2(pack16<true>(v[0], v[1]), pack16<true>(v[2], v[3]))
                         : make_uint2(pack16<false>(v[0], v[1]), pack16<false>(v[2], v[3]));
        if (!live) o = make_uint2(0, 0);
        *reinterpret_cast<uint2*>(t1 + i * TP + n0) = o;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][4 * rq + q] = 0.0f;
      }
    }
  }
  rb_barrier();
  RB_STAMP(5);
  next_tile = __builtin_amdgcn_readfirstlane(cst_ticket[0]);
  has_next = next_tile >= 0;
  {
    const int4 e = *reinterpret_cast<const int4*>(cst_ticket + 4);
    tile_next.row0 = __builtin_amdgcn_readfirstlane(e.x);
    tile_next.seq_begin = __builtin_amdgcn_readfirstlane(e.y);
    tile_next.seq_end = __builtin_amdgcn_readfirstlane(e.z);
    tile_next.seq_id = __builtin_amdgcn_readfirstlane(e.w);
  }
  if (MFIR && snake) {
    if constexpr (MFIR) {
      // act2 in place on t1 (rows [0, M1) of the window; the 6 rows on either side only feed outputs conv2 never reads)
      constexpr int CB = C / 16, SEG = (RB_THREADS / 64) / CB;
      const int cb = wave % CB, seg = wave / CB;
      constexpr int tiles_total = RB_M1 / 16;
      SnakeFir fir;
      fir.img = img; fir.pitch = TP;  // (rows past M1 + 12 still hold act1(x): finite)
      fir.frame0 = l0 - RB_LEAD - PADR; fir.T = T; fir.ch0 = cb * 16;
      fir.row_begin = __builtin_amdgcn_readfirstlane(16 * (seg * tiles_total / SEG)); fir.n_tiles = tiles_total / SEG;
      fir.er = FIR_LDS ? cst_snk[2 * C + cb * 16 + (lane & 15)] : er2;
      fir.inv_b = FIR_LDS ? cst_snk[3 * C + cb * 16 + (lane & 15)] : ib2;
      FirTaps ft;
#pragma unroll
      for (int k = 0; k < 12; ++k) ft.v[k] = f[k];
#ifndef RB_DIAG_NO_SWEEP2
      if constexpr (FIR_LDS) fir.load_constants(cst_fir, lane);
      else fir.copy_constants(fir_const);
      fir.begin(lane);
      rb_barrier();
      RB_STAMP(6);
      fir.template sweep<F16>(ft, lane);
      RB_STAMP(7);
#endif
    }
  } else if (snake) {
    const int base = l0 - RB_LEAD;  // local frame of t1 row 0
    if constexpr (C == RB_THREADS) {
      // C = 256: one thread per channel streams down all M1 rows.  The stream reads every raw row (up to 13 rows ahead)
      // into registers before it overwrites it, and no other thread touches this channel: in place without a barrier.
      const int chn = tid;
      const int t0 = base;
      if (t0 + RB_M1 - 1 >= 0 && t0 < T) {
        snake_stream<RB_M1 / 8>([&](int q2) {
          int i = q2 - base;
          i = i < 0 ? 0 : (i > RB_M1 - 1 ? RB_M1 - 1 : i);
          return from16<F16>(t1[i * TP + chn]);
        }, [&](int i, float v) { t1[i * TP + chn] = to16<F16>((t0 + i >= 0 && t0 + i < T) ? v : 0.0f); }, T, t0, f, expf(d.alpha2[chn]),
                                1.0f / (expf(d.beta2[chn]) + 1e-9f));
      } else {
        for (int i = 0; i < RB_M1; ++i) t1[i * TP + chn] = 0;
      }
      rb_barrier();
    } else {
      // in place on t1: every thread first computes all its outputs into registers, then overwrites.
      // item = (8*NCH2 rows, channel): 256 rows x C channels over 512 threads = C/2 values per thread
      constexpr int NCH2 = C == 32 ? 2 : 4, GR = 8 * NCH2;
      constexpr int ITEMS = (RB_M1 / GR) * C / RB_THREADS;
      static_assert((RB_M1 / GR) * C % RB_THREADS == 0, "in-place snake items must tile the workgroup");
      float o[ITEMS][GR];
#pragma unroll
      for (int q = 0; q < ITEMS; ++q) {
        const int it = tid + q * RB_THREADS;
        const int chn = it % C, i0 = (it / C) * GR;
        const int t0 = base + i0;
        const bool live = t0 + GR - 1 >= 0 && t0 < T;
#pragma unroll
        for (int i = 0; i < GR; ++i) o[q][i] = 0.0f;
        if (live) {
          snake_stream<NCH2, C != 32 || RB_C32_PREFETCH>([&](int q2) {
            int i = q2 - base;
            i = i < 0 ? 0 : (i > RB_M1 - 1 ? RB_M1 - 1 : i);  // only reached by rows whose outputs are not consumed
            return from16<F16>(t1[i * TP + chn]);
          }, [&](int i, float v) { o[q][i] = (RbCfg<C>::UNGUARDED ? (unsigned)(t0 + i) < (unsigned)T : (t0 + i >= 0 && t0 + i < T)) ? v : 0.0f; }, T, t0, f, expf(d.alpha2[chn]),
                             1.0f / (expf(d.beta2[chn]) + 1e-9f));
        }
      }
      rb_barrier();
#pragma unroll
      for (int q = 0; q < ITEMS; ++q) {
        const int it = tid + q * RB_THREADS;
        const int chn = it % C, i0 = (it / C) * GR;
#pragma unroll
        for (int i = 0; i < GR; ++i) t1[(i0 + i) * TP + chn] = to16<F16>(o[q][i]);
      }
    }
  }

  }
  // ------------------------------------------------------------------ conv2 (dilation 1) over t1; output row o <-> t1 row LEAD + o
  {
  RB_LANE_IDS(3);
  // 16-bit x: the residual of this lane's output row is fetched in conv2's first step and waits in registers while conv2 runs
  // (its latency is paid beside the MFMAs, not in the epilogue).  It - and at C = 128 the next tile's image - is issued BEHIND the
  // first step's weight-slab load: loads return in order, so the slab's wait then lets them stay in flight, and the next slab's
  // wait a whole step later finds them done.
  constexpr bool PREFETCH_RES = IOB && C <= 128;
  uint2 xres[PREFETCH_RES ? TN : 1][4];
  // (the first step is peeled - not a flag inside the loop: a value defined on one side of a branch only is what the register
  // allocator spills first; for the same reason every wavefront fetches a residual, also the one without output rows)
  auto conv2_prefetch = [&]() __attribute__((always_inline)) {
    if constexpr (PREFETCH_RES) {
      int row = tile.row0 + (wave < RB_BM / 32 ? wave : 0) * 32 + lrow;
      row = row < tile.seq_end ? row : tile.seq_end - 1;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) xres[j][rq] = *reinterpret_cast<const uint2*>(xh + (size_t)row * d.ldx + j * 32 + 8 * rq + 4 * lk);
    }
    if (XPF && snake && has_next) stage_issue(tile_next, tid);  // the next tile's image: in flight while conv2 and the epilogue run
  };
  auto conv2_step = [&](auto first, int ch, int tap0) __attribute__((always_inline)) {
    rb_barrier();
    // (after the last step the ring's buffer 0 is free again: the next tile's first slab goes there)
    const bool more = (step + 1 < total_steps || has_next) && !weights_resident;  // (resident: conv2's slab came with the first tile's conv1)
    if (more) load_slab(step + 1 < total_steps ? step + 1 : 0);
    if constexpr (decltype(first)::value) conv2_prefetch();
    if (wave < RB_BM / 32) {
      const int nt = d.taps - tap0 < TPS ? d.taps - tap0 : TPS;
      for (int tt = 0; tt < nt; ++tt) {
        const unsigned short* wb = ws + (size_t)(step & 1) * slab_alloc + tt * TAPW;
        const int tap = tap0 + tt;
        rb_conv_tap<C, KC / 16, TN, F16>(t1 + (RB_LEAD - h2 + wave * 32 + lrow + tap) * TP + ch * KC + lk * 8, wb + (lk * C + lrow) * 8, acc);
      }
    }
    if (more) store_slab((step + 1) & 1);
    ++step;
  };
  // DMA form of a conv2 step: slab (tap, half).  keep: vector-memory operations that may stay in flight at the step's barrier
  auto conv2_step_dma = [&](auto first, auto keep, int tap, int half) __attribute__((always_inline)) {
    dma_step_barrier(keep);
    int nxt = step + 2;
    nxt = nxt >= total_steps ? nxt - total_steps : nxt;
    // (every wavefront - also the one without output rows - issues its part of the slab load; the residual and next-image loads go
    // behind it: the next barrier lets those stay in flight)
    auto issue = [&]() __attribute__((always_inline)) {
      dma_issue(nxt, ring_pos >= 1 ? ring_pos - 1 : 2, tid);
      if constexpr (decltype(first)::value) conv2_prefetch();
    };
    if (wave < RB_BM / 32) {
      const unsigned short* wb = ws + (size_t)ring_pos * DMA_ELEMS;
      const unsigned short* ap = t1 + (RB_LEAD - h2 + wave * 32 + lrow + tap) * TP + half * SLAB_K + lk * 8;
      rb_conv_tap<C, SLAB_K / 16, TN, F16>(ap, wb + (lk * C + lrow) * 8, acc, issue);
      if constexpr (DTAPS > 1) {
#pragma unroll
        for (int tt = 1; tt < DTAPS; ++tt)
          if (tap + tt < d.taps) rb_conv_tap<C, SLAB_K / 16, TN, F16>(ap + tt * TP, wb + (size_t)tt * SLAB_K * C + (lk * C + lrow) * 8, acc);
      }
    } else {
      issue();
    }
    ring_pos = ring_pos == 2 ? 0 : ring_pos + 1;
    ++step;
  };
  RB_SETPRIO(RB_PRIO_CONV);
  if constexpr (DMA) {
    constexpr int KEEP1 = DMA_PW + (PREFETCH_RES ? TN * 4 : 0);  // the step behind the peeled one: the residual loads may stay in flight
    conv2_step_dma(std::true_type{}, std::integral_constant<int, DMA_PW>{}, 0, 0);
    if constexpr (SPLIT == 2) conv2_step_dma(std::false_type{}, std::integral_constant<int, KEEP1>{}, 0, 1);
    for (int tap = DTAPS; tap < d.taps; tap += DTAPS) {
      if (SPLIT == 1 && tap == DTAPS) conv2_step_dma(std::false_type{}, std::integral_constant<int, KEEP1>{}, tap, 0);
      else conv2_step_dma(std::false_type{}, std::integral_constant<int, DMA_PW>{}, tap, 0);
      if constexpr (SPLIT == 2) conv2_step_dma(std::false_type{}, std::integral_constant<int, DMA_PW>{}, tap, 1);
    }
  } else {
  conv2_step(std::true_type{}, 0, 0);
  for (int ch = 0; ch < NCH; ++ch)
    for (int tap0 = ch == 0 ? TPS : 0; tap0 < d.taps; tap0 += TPS) conv2_step(std::false_type{}, ch, tap0);
  }
  RB_SETPRIO(RB_PRIO_BASE);

  if (has_next) rb_barrier();  // every wavefront is done reading t1 and the slab ring: the next tile's image may overwrite them
  RB_STAMP(8);
  // ------------------------------------------------------------------ epilogue
  // transposed accumulators again: lane = output row, registers = groups of four consecutive channels.  Residual read, scaling,
  // optional accumulate and the store happen in registers on 8-byte (16-bit tensors) or 16-byte (fp32) pieces of the row: no
  // LDS round trip, no barrier.  A row's pieces are written by one wavefront within a few hundred cycles, so L2 merges them
  // into whole lines before they reach HBM.
  if (wave < RB_BM / 32) {
    const int row = tile.row0 + wave * 32 + lrow;
    if (row < tile.seq_end) {
      unsigned short* yh = reinterpret_cast<unsigned short*>(d.y);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int n0 = j * 32 + 8 * rq + 4 * lk;
          const float4 bb = *reinterpret_cast<const float4*>(cst_b2 + n0);
          float v[4] = {d.alpha * (acc[j][4 * rq] + bb.x), d.alpha * (acc[j][4 * rq + 1] + bb.y), d.alpha * (acc[j][4 * rq + 2] + bb.z),
                        d.alpha * (acc[j][4 * rq + 3] + bb.w)};
          if constexpr (IOB) {
            uint2 xr;
            if constexpr (PREFETCH_RES) xr = xres[j][rq];
            else xr = *reinterpret_cast<const uint2*>(xh + (size_t)row * d.ldx + n0);
            v[0] += d.res_scale * from16<F16>(xr.x & 0xFFFF);
            v[1] += d.res_scale * from16<F16>(xr.x >> 16);
            v[2] += d.res_scale * from16<F16>(xr.y & 0xFFFF);
            v[3] += d.res_scale * from16<F16>(xr.y >> 16);
            uint2* yp = reinterpret_cast<uint2*>(yh + (size_t)row * d.ldy + n0);
            if (d.accumulate) {
              const uint2 yr = *yp;
              v[0] += from16<F16>(yr.x & 0xFFFF);
              v[1] += from16<F16>(yr.x >> 16);
              v[2] += from16<F16>(yr.y & 0xFFFF);
              v[3] += from16<F16>(yr.y >> 16);
            }
            *yp = make_uint2(pack16<F16>(v[0], v[1]), pack16<F16>(v[2], v[3]));
          } else {
            const float4 xr = *reinterpret_cast<const float4*>(d.x + (size_t)row * d.ldx + n0);
            v[0] += d.res_scale * xr.x;
            v[1] += d.res_scale * xr.y;
            v[2] += d.res_scale * xr.z;
            v[3] += d.res_scale * xr.w;
            float4* yp = reinterpret_cast<float4*>(d.y + (size_t)row * d.ldy + n0);
            if (d.accumulate) {
              const float4 yr = *yp;
              v[0] += yr.x;
              v[1] += yr.y;
              v[2] += yr.z;
              v[3] += yr.w;
            }
            *yp = make_float4(v[0], v[1], v[2], v[3]);
          }
          // one row piece at a time: interleaved, the eight pieces' temporaries (~80 registers beside the accumulators and the
          // prefetched residual) are what pushed the C = 64 instantiation over its 128 registers
          if constexpr (PREFETCH_RES) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
#ifdef RB_DIAG_CLOCK
  RB_STAMP(9);
#pragma unroll
  for (int k = 0; k < 9; ++k) phase_sum[k] += stamp[k + 1] - stamp[k];
  ++n_iter;
#endif
  }
  if (!has_next) break;
  first_tile = false;
  tile = tile_next;
  cur_tile = next_tile;
  }  // tiles of this workgroup
#undef RB_LANE_IDS
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two slabs requested past the last tile land before the LDS is released
  retire(tid_prologue);
#ifdef RB_DIAG_CLOCK
  if (MFIR && snake && tid == 64 * (RB_DIAG_CLOCK + 0)) {
    if (blockIdx.x < RB_TRACE_MAX) {
#pragma unroll
      for (int k = 0; k < 9; ++k) g_rb_trace[blockIdx.x][k] = phase_sum[k];
      g_rb_trace[blockIdx.x][9] = n_iter;
      g_rb_trace[blockIdx.x][10] = rt0;
      g_rb_trace[blockIdx.x][11] = __builtin_amdgcn_s_memrealtime();
      g_rb_trace[blockIdx.x][12] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
  }
#endif
}

}  // namespace tts
// Diagnostic (tests/conftest.py calls it after every GPU test): words of the work-queue slots that are not zero once the device
// is idle.  Every launch must leave its slot clean - a dirty slot would hand the launch that next draws it tickets that start
// in the middle of a run.  Returns the count (0 = clean), negative on a HIP error.
namespace tts {
static std::mutex g_rb_slot_lock;
static std::map<int, int> g_rb_slots_used;                             // device -> slots handed out so far
static std::map<std::pair<int, hipStream_t>, int> g_rb_stream_slot;    // (device, stream) -> its slot
}  // namespace tts
extern "C" int tts_diag_queue_nonzero(void) {
  static std::vector<unsigned int> host;
  int dev = 0, used = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
  {
    std::lock_guard<std::mutex> guard(tts::g_rb_slot_lock);
    used = tts::g_rb_slots_used[dev];
  }
  if (used == 0) return 0;
  host.resize((size_t)used * 16);
  if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(tts::g_rb_queue), host.size() * sizeof(unsigned int)) != hipSuccess) return -2;
  int n = 0;
  for (unsigned int v : host) n += v != 0;
  return n;
}
// slots handed out on the current device so far (tests: a captured launch takes a fresh one, a stream keeps its own)
extern "C" int tts_diag_queue_slots_used(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> guard(tts::g_rb_slot_lock);
  return tts::g_rb_slots_used[dev];
}
namespace tts {

#ifdef RB_DIAG_CLOCK
}  // namespace tts
extern "C" int tts_rb_diag_trace(unsigned long long* out, int n_workgroups) {
  if (n_workgroups > tts::RB_TRACE_MAX) n_workgroups = tts::RB_TRACE_MAX;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tts::g_rb_trace), (size_t)n_workgroups * 128) == hipSuccess ? n_workgroups : -1;
}
namespace tts {
#endif

#undef load_slab
#undef store_slab

// The work-queue slot of a launch on `st` (see g_rb_queue): the stream's own slot, or - while the stream is being captured into a
// HIP graph - a fresh one that stays with the recorded kernel node.  -1: the pool is used up.
static int rb_queue_slot(int dev, hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
  std::lock_guard<std::mutex> guard(g_rb_slot_lock);
  int& used = g_rb_slots_used[dev];
  if (!capturing) {
    auto it = g_rb_stream_slot.find({dev, st});
    if (it != g_rb_stream_slot.end()) return it->second;
  }
  if (used >= RB_QUEUE_SLOTS) return -1;
  const int slot = used++;
  if (!capturing) g_rb_stream_slot[{dev, st}] = slot;
  return slot;
}

template <int C, bool IOB, bool F16, bool MFIR>
static int launch_rb(const TtsResblockDesc& d, hipStream_t st) {
  constexpr int KC = RbCfg<C>::KC, RB_M1 = RbCfg<C>::M1, RB_BM = RbCfg<C>::BM, RB_THREADS = RbCfg<C>::THREADS;
  const int h1 = (d.taps - 1) / 2 * d.dil;
  const int img_rows = MFIR ? RbCfg<C>::img_rows(h1) : RbCfg<C>::win_alloc(h1);
  const size_t xa = (((size_t)img_rows * (KC + 8)) + 7) & ~(size_t)7, t1 = (size_t)(RB_M1 + (MFIR ? 12 : 0)) * (C + 8);
  size_t lds = ((xa > t1 ? xa : t1) + (size_t)RbCfg<C>::ring_elems(d.taps)) * 2 + (size_t)2 * C * 4 + (MFIR && C < 128 ? 4096 + 4 * C * 4 : 0) + 32;
  TTS_CHECK_ARG(lds <= 160 * 1024, "resblock_step: LDS %zu B exceeds 160 KiB", lds);
  auto k = resblock_step_kernel<C, IOB, F16, MFIR>;
  static unsigned long long lds_raised = 0;  // devices on which this instantiation's limit is already raised
  if (lds > 64 * 1024 && raise_lds_limit(reinterpret_cast<const void*>(k), lds_raised) != hipSuccess) {
    set_error("resblock_step: raising the dynamic LDS limit failed");
    return TTS_E_LAUNCH;
  }
  // persistent grid: as many workgroups as the device keeps resident at this LDS size (rounded up to whole rounds of the 8 XCDs),
  // never more than tiles
  // (queried once per instantiation, device and LDS size: the answers do not change, the runtime calls cost microseconds each)
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    set_error("resblock_step: hipGetDevice failed");
    return TTS_E_LAUNCH;
  }
  {
    static std::mutex cache_lock;
    static std::map<std::pair<int, size_t>, std::pair<int, int>> cache;  // (device, lds) -> (CUs, workgroups per CU)
    std::lock_guard<std::mutex> guard(cache_lock);
    auto it = cache.find({dev, lds});
    if (it == cache.end()) {
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), RB_THREADS, lds) != hipSuccess || cus < 1 || per_cu < 1) {
        set_error("resblock_step: occupancy query failed");
        return TTS_E_LAUNCH;
      }
      it = cache.emplace(std::make_pair(dev, lds), std::make_pair(cus, per_cu)).first;
    }
    cus = it->second.first;
    per_cu = it->second.second;
  }
  static const int fixed_per_cu = std::getenv("TOUCAN_RB_WG_PER_CU") ? std::atoi(std::getenv("TOUCAN_RB_WG_PER_CU")) : 0;  // (A/B runs)
  if (fixed_per_cu > 0) per_cu = fixed_per_cu;
  const long resident = (long)cus * per_cu;
  const int grid = (int)(fixed_per_cu < 0 || d.n_tiles < resident ? d.n_tiles : resident);  // (TOUCAN_RB_WG_PER_CU=-1: one workgroup per tile)
  const int queue_slot = rb_queue_slot(dev, st);
  if (queue_slot < 0) {
    set_error("resblock_step: all %d work-queue slots of device %d are taken (streams + launches captured into HIP graphs)", RB_QUEUE_SLOTS, dev);
    return TTS_E_LAUNCH;
  }
  // second workgroup of a CU (C <= 64: two per CU) starts late by this many units of 1 024 cycles - see g_rb_cu_arrivals
  static const int stagger_env = std::getenv("TOUCAN_RB_STAGGER") ? std::atoi(std::getenv("TOUCAN_RB_STAGGER")) : -1;
  const int stagger = (per_cu == 2 && grid == resident) ? (stagger_env >= 0 ? stagger_env : RB_STAGGER_DEFAULT) : 0;
  hipLaunchKernelGGL(k, dim3(grid), dim3(RB_THREADS), lds, st, d, queue_slot, stagger);
  return launch_status("resblock_step");
}

// Host mirror of SnakeFir::gen_up / gen_down in the interior of an utterance (no tap folds onto an edge sample): the per-lane
// A operands [ua0 | ua1 | da0 | da1][64 lanes][8 halfs] the kernel loads in SnakeFir::begin.
static unsigned short host_f32_to_f16(float f) {  // round to nearest even; the taps are normal numbers well inside fp16's range
  unsigned int u;
  memcpy(&u, &f, 4);
  const unsigned int sign = (u >> 16) & 0x8000u;
  const int exp = (int)((u >> 23) & 0xFF) - 127 + 15;
  unsigned int man = u & 0x7FFFFFu;
  if ((u & 0x7FFFFFFFu) == 0) return (unsigned short)sign;
  if (exp >= 31) return (unsigned short)(sign | 0x7C00u);
  if (exp <= 0) {  // subnormal half
    if (exp < -10) return (unsigned short)sign;
    man |= 0x800000u;
    const int shift = 14 - exp;
    unsigned int h = man >> shift;
    const unsigned int rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1))) ++h;
    return (unsigned short)(sign | h);
  }
  unsigned int h = ((unsigned int)exp << 10) | (man >> 13);
  const unsigned int rem = man & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;  // a carry into the exponent is the correct rounding
  return (unsigned short)(sign | h);
}

int snake_fir_table(const float* f, void* table) {
  TTS_CHECK_ARG(f && table, "snake_fir_table: null pointer");
  unsigned short* out = static_cast<unsigned short*>(table);
  for (int lane = 0; lane < 64; ++lane) {
    const int row = lane & 15, g = lane >> 4;
    for (int i = 0; i < 2; ++i) {  // up-sampler tile i: outputs n = 2 qo + p, qo = 3 + 8 i + (row >> 1), inputs q = 8 g + j
      const int qo = 3 + 8 * i + (row >> 1), p = row & 1;
      for (int j = 0; j < 8; ++j) {
        const int dq = 8 * g + j - qo;               // x[qo + dq]
        const int tap = 5 + p - 2 * dq;              // u[2 qo + p] = 2 sum_d x[qo + d] f[5 + p - 2 d]
        const bool in = p ? (dq >= -2 && dq <= 3) : (dq >= -3 && dq <= 2);
        out[(i * 64 + lane) * 8 + j] = host_f32_to_f16(in ? 2.0f * f[tap] : 0.0f);
      }
    }
    for (int s = 0; s < 2; ++s) {  // decimator K-step s: y[row] = sum_k s2[2 row + k - 5] f[k], slot (g, j) <-> n = -6 + 32 s + 4 g + j (+ 12)
      for (int j = 0; j < 8; ++j) {
        const int n = -6 + 32 * s + 4 * g + j + (j >= 4 ? 12 : 0);
        const int k = n - 2 * row + 5;
        out[((2 + s) * 64 + lane) * 8 + j] = host_f32_to_f16((k >= 0 && k <= 11) ? f[k] : 0.0f);
      }
    }
  }
  return TTS_OK;
}

int resblock_tile_rows(int c) { return c == 256 ? RbCfg<256>::BM : (c == 64 ? RbCfg<64>::BM : RbCfg<32>::BM); }

int resblock_step(const TtsResblockDesc& d, hipStream_t st) {
  TTS_CHECK_ARG(d.x && d.y && d.w1 && d.w2 && d.b1 && d.b2 && d.tiles, "resblock_step: null pointer");
  TTS_CHECK_ARG(d.c == 32 || d.c == 64 || d.c == 128 || d.c == 256, "resblock_step: C=%d unsupported (32, 64, 128, 256)", d.c);
  TTS_CHECK_ARG(d.taps >= 1 && d.taps <= 11 && (d.taps & 1) && d.dil >= 1, "resblock_step: taps %d / dil %d unsupported", d.taps, d.dil);
  TTS_CHECK_ARG((d.taps - 1) / 2 + 6 <= RB_LEAD, "resblock_step: conv2 halo too large");
  const int bm = resblock_tile_rows(d.c);
  TTS_CHECK_ARG(d.tile_rows == bm, "resblock_step: tile table must use %d rows for C=%d, got %d", bm, d.c, d.tile_rows);
  TTS_CHECK_ARG(d.act == TTS_PRE_LRELU || d.act == TTS_PRE_SNAKE, "resblock_step: act must be LRELU or SNAKE");
  TTS_CHECK_ARG(d.act != TTS_PRE_SNAKE || (d.alpha1 && d.beta1 && d.alpha2 && d.beta2 && d.filt), "resblock_step: snake parameters missing");
  TTS_CHECK_ARG(d.act != TTS_PRE_SNAKE || d.c > 128 || d.fir_tab || std::getenv("TOUCAN_SNAKE_VALU"), "resblock_step: fir_tab (tts_snake_fir_table) missing");
  TTS_CHECK_ARG((d.ldx & 3) == 0 && ((uintptr_t)d.x & 15) == 0, "resblock_step: x must be 16-byte aligned rows");
  TTS_CHECK_ARG(d.io_bf16 || ((d.ldy & 3) == 0 && ((uintptr_t)d.y & 15) == 0), "resblock_step: y must be 16-byte aligned rows");
  TTS_CHECK_ARG(((uintptr_t)d.b1 & 15) == 0 && ((uintptr_t)d.b2 & 15) == 0, "resblock_step: biases must be 16-byte aligned");
  TTS_CHECK_ARG(!d.io_bf16 || ((d.ldx & 7) == 0 && (d.ldy & 7) == 0 && ((uintptr_t)d.y & 15) == 0), "resblock_step: bf16 rows must be 16-byte aligned");
  TTS_CHECK_ARG(d.compute == 1 || d.compute == 2, "resblock_step: compute must be 1 (bf16) or 2 (fp16), got %d", d.compute);
  if (d.n_tiles == 0) return TTS_OK;
  // C <= 128: the snakes' FIR filters on the matrix cores (TOUCAN_SNAKE_VALU=1 keeps the register-streamed VALU form for A/B runs)
  static const bool valu_snake = std::getenv("TOUCAN_SNAKE_VALU") != nullptr;
#define TTS_RB(IOB_, F16_)                                                                                  \
  switch (d.c) {                                                                                            \
    case 32: return valu_snake ? launch_rb<32, IOB_, F16_, false>(d, st) : launch_rb<32, IOB_, F16_, true>(d, st);    \
    case 64: return valu_snake ? launch_rb<64, IOB_, F16_, false>(d, st) : launch_rb<64, IOB_, F16_, true>(d, st);    \
    case 128: return valu_snake ? launch_rb<128, IOB_, F16_, false>(d, st) : launch_rb<128, IOB_, F16_, true>(d, st); \
    default: return launch_rb<256, IOB_, F16_, false>(d, st);                                               \
  }
  if (d.compute == 2) {
    if (d.io_bf16) { TTS_RB(true, true) }
    TTS_RB(false, true)
  }
  if (d.io_bf16) { TTS_RB(true, false) }
  TTS_RB(false, false)
#undef TTS_RB
}

}  // namespace tts
