// Stage-level entry points of libtoucan_hip.so: a handle that owns the packed weights and the workspace, and C++ functions that
// sequence the kernels of one stage of the reference's forward pass (declared in include/toucan_tts.h, "stage API").
//
// This is the native counterpart of the launch sequencing in engine.py: same kernels, same order, same buffers' roles - a pass
// is a handful of calls through the C ABI instead of ~500.  Mirrors, stage by stage:
//   tts_encoder               Conformer.forward (Layers/Conformer.py:92-134) on the phoneme features
//   tts_variance_predictors   VariancePredictor.forward / DurationPredictor.inference (VariancePredictor.py:65-80, DurationPredictor.py:63-83)
//   tts_control_and_regulate  InferenceToucanTTS.py:214-235 (+ _scale_variance :333-343, LengthRegulator.py:37-61)
//   tts_decoder               decoder Conformer + feat_out (InferenceToucanTTS.py:238-239)
//   tts_postnet               PostNet.forward + residual (PostNet.py:62-74, InferenceToucanTTS.py:241)
//   tts_postflow              Glow.forward(infer=True) (Glow.py:342-391)
//   tts_vocoder_bigvgan/_hifigan  InferenceBigVGAN.py:72-95 / InferenceAvocodo.py:69-80
//   tts_synthesize_batch      all of the above for one ragged batch
// Host work here is layout arithmetic (tile tables, offsets) and launch ordering; every FLOP is in the kernels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "common.h"

namespace tts {

namespace {

constexpr int ATT = 192, HEADS = 4, DK = 48;

struct Dev {
  void* p = nullptr;
  int64_t shape[4] = {0, 0, 0, 0};
  int ndim = 0, dtype = 0;
  size_t bytes = 0;
};

struct ConvW {
  const float* w = nullptr;
  const void* w16 = nullptr;
  const float* bias = nullptr;
  int mode = 0, taps = 1, dil = 1, pad_left = 0, cin = 0, cin_pad = 0, cout = 0, wn = 0, half_pad = 0, tile_rows = 0, small_tile_rows = 0,
      small_only = 0, n_tile = 0, compute16 = 0, algo_taps = 1;
};

// packed ragged layout: utterance u occupies rows [begins[u], begins[u] + lengths[u])
struct Layout {
  std::vector<int> begins, lengths;
  int total = 0, max_len = 0;
  static Layout make(const int* len, int n, int align) {
    Layout l;
    int off = 0;
    for (int i = 0; i < n; ++i) {
      l.begins.push_back(off);
      l.lengths.push_back(len[i]);
      off += (len[i] + align - 1) / align * align;
      l.max_len = std::max(l.max_len, len[i]);
    }
    l.total = off;
    return l;
  }
  Layout scaled(int f) const {
    Layout l;
    for (size_t i = 0; i < begins.size(); ++i) {
      l.begins.push_back(begins[i] * f);
      l.lengths.push_back(lengths[i] * f);
    }
    l.total = total * f;
    l.max_len = max_len * f;
    return l;
  }
  Layout halved() const {  // Glow squeeze: an odd last frame is dropped (glow_utils.py:31-32); begins are even by construction
    Layout l;
    for (size_t i = 0; i < begins.size(); ++i) {
      l.begins.push_back(begins[i] / 2);
      l.lengths.push_back(lengths[i] / 2);
      l.max_len = std::max(l.max_len, lengths[i] / 2);
    }
    l.total = total / 2;
    return l;
  }
  int n() const { return (int)lengths.size(); }
};

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0;
  void reset() { off = 0; }
};

struct TileTab {
  TtsTile* dev = nullptr;
  int n = 0;
};

// Tile tables / utterance bounds of ONE batch's group of stages (acoustic stages, or the vocoder): pinned host staging and the device
// copy, filled by asynchronous copies on the group's stream - no device allocation, no synchronisation per table.  Two generations
// per group alternate from batch to batch.
struct TableArena {
  char* dev = nullptr;
  char* host = nullptr;
  size_t cap = 0, off = 0;
  hipEvent_t copied = nullptr;   // recorded behind the generation's latest copy: the staging bytes may be rewritten once it has passed
  hipEvent_t fence = nullptr;    // (reuse from another stream) everything enqueued on the previous stream when the arena is taken again
  bool armed = false;
  hipStream_t stream = nullptr;  // the stream the generation's copies and kernels were enqueued on
  std::unordered_map<std::string, TileTab> tiles;
  std::unordered_map<std::string, int*> bounds;
};

// one timed launch of the roofline leg (tts_profile): HIP events on the launch stream around a matrix-core kernel
struct ProfRec {
  std::string name;
  double flops = 0, bytes = 0, elems = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

}  // namespace

struct Handle {
  TtsConfig cfg;
  std::unordered_map<std::string, Dev> weights;
  bool resolved = false;
  Arena phone, frame, voc[2], tables;
  // tables of layouts that keep coming back (a benchmark's fixed batch): device allocations of their own, kept.  A layout is
  // built into the batch's TableArena the first time it is seen and moves here the second time (real traffic - new utterance
  // lengths every batch - never allocates or synchronises for a table).
  std::unordered_map<std::string, TileTab> tile_cache;
  std::unordered_map<std::string, int*> bounds_cache;
  std::unordered_set<std::string> seen_once;
  TableArena tabs[2][2];          // [0: acoustic stages, 1: vocoder][generation]
  int tab_gen[2] = {0, 0};
  int tab_which = 0;              // group of the stage entry that runs (set with split_mode)
  long long n_arena_tables = 0, n_cached_tables = 0;  // tables built into a batch arena / given a permanent allocation (tts_table_stats)
  int small_tile_blocks = 1536;
  int split_mode = 0;             // set by the stage entries; the fp32 configuration only.  2 (phoneme stages: encoder, predictors): split-K convs
                                  // and key-split attention at EVERY grid size - durations are a rounding of exp(log d), so everything upstream
                                  // of them keeps one arithmetic whatever the batch (an utterance's frame count cannot depend on the batch or
                                  // shard it is in); 1 (frame stages): the split forms on small grids only (TTS_IO_SPLIT_K / TTS_ATT_KEY_SPLIT:
                                  // the mel agrees to rounding order across batch sizes); 0 (vocoder: chunked == whole, bit for bit)
  bool no_fused_wavenet = false;  // TOUCAN_NO_FUSED_WAVENET: A/B switch, same meaning as in engine.py
  bool no_fused_ffn = false;      // TOUCAN_NO_FUSED_FFN: likewise
  bool no_f16_attention = false;  // TOUCAN_NO_F16_ATTENTION: likewise
  // relative position tables [block][2 pmax - 1][192] of the two Conformer stacks, built from the uploaded sinusoid table
  float* ptab[2] = {nullptr, nullptr};
  int pmax = 0;
  // state of the batch in flight
  Layout lp, lf;          // phoneme and frame layouts
  int B = 0;
  const float* text = nullptr;
  float *e_norm = nullptr, *enc = nullptr, *pitch = nullptr, *energy = nullptr, *cln = nullptr;
  int* dur = nullptr;
  float *cat = nullptr, *dec = nullptr, *mel0 = nullptr, *mel = nullptr;
  std::vector<int> frames;  // per utterance, after the control step
  bool have_flow = false;
  // profiling (bench.py's roofline leg): event pairs around the launches of the selected kernel class ("" = every class)
  bool prof_on = false, prof_detail = false;
  std::string prof_select;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> event_pool;
};

namespace {

#define TTS_TRY(expr)            \
  do {                           \
    const int rc_ = (expr);      \
    if (rc_ != TTS_OK) return rc_; \
  } while (0)

bool is16(const Handle* h);

int hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return TTS_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return TTS_E_LAUNCH;
}

// ---- arenas ---------------------------------------------------------------------------------------------------------
int arena_reserve(Arena& a, size_t bytes, hipStream_t st) {
  a.reset();
  if (a.cap >= bytes) return TTS_OK;
  if (a.base) {
    TTS_TRY(hip_ok(hipStreamSynchronize(st), "workspace: stream sync before regrowth"));
    (void)hipFree(a.base);
    a.base = nullptr;
    a.cap = 0;
  }
  const size_t want = bytes + bytes / 8 + (1 << 20);  // (= with_growth_slack(bytes): tts_workspace_bytes counts it)
  TTS_TRY(hip_ok(hipMalloc(reinterpret_cast<void**>(&a.base), want), "workspace: hipMalloc"));
  a.cap = want;
  return TTS_OK;
}

template <class T>
T* arena_alloc(Arena& a, size_t count) {
  const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
  if (a.off + bytes > a.cap) return nullptr;
  T* p = reinterpret_cast<T*>(a.base + a.off);
  a.off += bytes;
  return p;
}

#define TTS_ALLOC(var, arena, T, count)                                                        \
  T* var = arena_alloc<T>(arena, (size_t)(count));                                             \
  if (!var) {                                                                                  \
    set_error("%s: workspace exhausted (%zu of %zu bytes used)", __func__, (arena).off, (arena).cap); \
    return TTS_E_ARG;                                                                          \
  }

// ---- tile tables and utterance bounds (cached per layout signature) ---------------------------------------------------
std::string layout_key(const Layout& l, int tile_rows) {
  std::string k(reinterpret_cast<const char*>(l.begins.data()), l.begins.size() * sizeof(int));
  k.append(reinterpret_cast<const char*>(l.lengths.data()), l.lengths.size() * sizeof(int));
  k.append(reinterpret_cast<const char*>(&tile_rows), sizeof(int));
  return k;
}

void drop_tables(Handle* h) {
  for (auto& kv : h->tile_cache) (void)hipFree(kv.second.dev);
  for (auto& kv : h->bounds_cache) (void)hipFree(kv.second);
  h->tile_cache.clear();
  h->bounds_cache.clear();
}

// a stage group starts a batch: take the group's other table generation
int tables_begin(Handle* h, int which, hipStream_t st) {
  h->tab_which = which;
  h->tab_gen[which] ^= 1;
  TableArena& a = h->tabs[which][h->tab_gen[which]];
  if (a.armed) {
    // the staging bytes of the batch before last: its copies have long run unless the host is more than two batches ahead of the GPU
    TTS_TRY(hip_ok(hipEventSynchronize(a.copied), "tile tables: staging buffer still in flight"));
    if (a.stream != st) {  // its kernels ran on another stream: order this batch's copies behind everything enqueued there
      TTS_TRY(hip_ok(hipEventRecord(a.fence, a.stream), "tile tables: fence"));
      TTS_TRY(hip_ok(hipStreamWaitEvent(st, a.fence, 0), "tile tables: fence wait"));
    }
  }
  a.off = 0;
  a.stream = st;
  a.tiles.clear();
  a.bounds.clear();
  return TTS_OK;
}

// bytes -> the current generation of the running group (staging copy + one asynchronous upload); *dev_out: the device address
int table_put(Handle* h, const void* data, size_t bytes, hipStream_t st, void** dev_out) {
  TableArena& a = h->tabs[h->tab_which][h->tab_gen[h->tab_which]];
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (a.off + need > a.cap) {  // (rare: the first batches, or a batch far larger than any before)
    // tables already handed out from this generation are still referenced by launches in flight: nothing may be freed under them
    TTS_TRY(hip_ok(hipDeviceSynchronize(), "tile tables: sync before regrowth"));
    const size_t cap = std::max<size_t>(std::max<size_t>(a.cap * 2, a.off + need), (size_t)4 << 20);
    char *dev = nullptr, *host = nullptr;
    TTS_TRY(hip_ok(hipMalloc(reinterpret_cast<void**>(&dev), cap), "tile tables: hipMalloc"));
    TTS_TRY(hip_ok(hipHostMalloc(reinterpret_cast<void**>(&host), cap, hipHostMallocDefault), "tile tables: hipHostMalloc"));
    if (a.off) {
      memcpy(host, a.host, a.off);
      TTS_TRY(hip_ok(hipMemcpy(dev, a.dev, a.off, hipMemcpyDeviceToDevice), "tile tables: regrowth copy"));
      // (tables of this generation handed out before the regrowth keep their old addresses: rebuild the maps' pointers)
      for (auto& kv : a.tiles) kv.second.dev = reinterpret_cast<TtsTile*>(dev + (reinterpret_cast<char*>(kv.second.dev) - a.dev));
      for (auto& kv : a.bounds) kv.second = reinterpret_cast<int*>(dev + (reinterpret_cast<char*>(kv.second) - a.dev));
    }
    if (a.dev) (void)hipFree(a.dev);
    if (a.host) (void)hipHostFree(a.host);
    a.dev = dev; a.host = host; a.cap = cap;
  }
  if (!a.copied) {
    TTS_TRY(hip_ok(hipEventCreateWithFlags(&a.copied, hipEventDisableTiming), "tile tables: event"));
    TTS_TRY(hip_ok(hipEventCreateWithFlags(&a.fence, hipEventDisableTiming), "tile tables: event"));
  }
  memcpy(a.host + a.off, data, bytes);
  TTS_TRY(hip_ok(hipMemcpyAsync(a.dev + a.off, a.host + a.off, bytes, hipMemcpyHostToDevice, st), "tile tables: upload"));
  TTS_TRY(hip_ok(hipEventRecord(a.copied, st), "tile tables: event record"));
  a.armed = true;
  *dev_out = a.dev + a.off;
  a.off += need;
  return TTS_OK;
}

// first sighting of a layout: build into the batch arena; second sighting: it is a repeating layout - give it a permanent table
bool repeating(Handle* h, const std::string& key) {
  if (h->seen_once.count(key)) return true;
  if (h->seen_once.size() > 8192) h->seen_once.clear();
  h->seen_once.insert(key);
  return false;
}

int tiles_of(Handle* h, const Layout& l, int tile_rows, hipStream_t st, TileTab* out) {
  const std::string key = layout_key(l, tile_rows);
  auto it = h->tile_cache.find(key);
  if (it != h->tile_cache.end()) {
    *out = it->second;
    return TTS_OK;
  }
  TableArena& a = h->tabs[h->tab_which][h->tab_gen[h->tab_which]];
  auto ia = a.tiles.find(key);
  if (ia != a.tiles.end()) {
    *out = ia->second;
    return TTS_OK;
  }
  std::vector<TtsTile> host;
  for (int u = 0; u < l.n(); ++u)
    for (int r = 0; r < l.lengths[u]; r += tile_rows) host.push_back(TtsTile{l.begins[u] + r, l.begins[u], l.begins[u] + l.lengths[u], u});
  TileTab t;
  t.n = (int)host.size();
  if (!repeating(h, key)) {
    if (host.empty()) host.push_back(TtsTile{0, 0, 0, 0});
    void* dev;
    TTS_TRY(table_put(h, host.data(), host.size() * sizeof(TtsTile), st, &dev));
    t.dev = static_cast<TtsTile*>(dev);
    a.tiles[key] = t;
    ++h->n_arena_tables;
    *out = t;
    return TTS_OK;
  }
  if (h->tile_cache.size() > 512) {  // (no launch in flight may still read a table - on this stream or, when the caller runs the
                                     // vocoder of one batch beside the acoustic model of the next, on another: drain the device first)
    TTS_TRY(hip_ok(hipDeviceSynchronize(), "tile tables: sync before trimming the cache"));
    drop_tables(h);
  }
  TTS_TRY(hip_ok(hipMalloc(reinterpret_cast<void**>(&t.dev), std::max<size_t>(1, host.size()) * sizeof(TtsTile)), "tile table: hipMalloc"));
  if (!host.empty())
    TTS_TRY(hip_ok(hipMemcpyAsync(t.dev, host.data(), host.size() * sizeof(TtsTile), hipMemcpyHostToDevice, st), "tile table: upload"));
  TTS_TRY(hip_ok(hipStreamSynchronize(st), "tile table: upload sync"));  // (the host vector goes out of scope; once per repeating layout)
  h->tile_cache[key] = t;
  ++h->n_cached_tables;
  *out = t;
  return TTS_OK;
}

// (seq_begin[n] | seq_end[n]) device array
int bounds_of(Handle* h, const Layout& l, hipStream_t st, const int** sb, const int** se) {
  const std::string key = layout_key(l, -1);
  auto it = h->bounds_cache.find(key);
  int* dev = nullptr;
  TableArena& a = h->tabs[h->tab_which][h->tab_gen[h->tab_which]];
  if (it != h->bounds_cache.end()) {
    dev = it->second;
  } else if (a.bounds.count(key)) {
    dev = a.bounds[key];
  } else {
    std::vector<int> host(2 * std::max(1, l.n()));
    for (int u = 0; u < l.n(); ++u) {
      host[u] = l.begins[u];
      host[l.n() + u] = l.begins[u] + l.lengths[u];
    }
    if (!repeating(h, key)) {
      void* p;
      TTS_TRY(table_put(h, host.data(), host.size() * sizeof(int), st, &p));
      dev = static_cast<int*>(p);
      a.bounds[key] = dev;
    } else {
      TTS_TRY(hip_ok(hipMalloc(reinterpret_cast<void**>(&dev), host.size() * sizeof(int)), "bounds: hipMalloc"));
      TTS_TRY(hip_ok(hipMemcpyAsync(dev, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, st), "bounds: upload"));
      TTS_TRY(hip_ok(hipStreamSynchronize(st), "bounds: upload sync"));
      h->bounds_cache[key] = dev;
    }
  }
  *sb = dev;
  *se = dev + l.n();
  return TTS_OK;
}

// ---- weights ------------------------------------------------------------------------------------------------------------
const Dev* find(const Handle* h, const std::string& name) {
  auto it = h->weights.find(name);
  return it == h->weights.end() ? nullptr : &it->second;
}

int need(const Handle* h, const std::string& name, const Dev** out) {
  *out = find(h, name);
  if (!*out) {
    set_error("weight '%s' was not loaded (tts_load_weights)", name.c_str());
    return TTS_E_ARG;
  }
  return TTS_OK;
}

int fvec(const Handle* h, const std::string& name, const float** out) {
  const Dev* d;
  TTS_TRY(need(h, name, &d));
  *out = static_cast<const float*>(d->p);
  return TTS_OK;
}

// a packed conv: "<name>.w" (+ ".w16", ".bias") and the host-side descriptor fields in "<name>.meta" (int32[16], kept on the host)
int conv_of(const Handle* h, const std::string& name, ConvW* c) {
  const Dev *w, *m;
  TTS_TRY(need(h, name + ".w", &w));
  TTS_TRY(need(h, name + ".meta", &m));
  const int* mm = static_cast<const int*>(m->p);  // meta tensors live in host memory (dtype 3 is never uploaded)
  c->w = static_cast<const float*>(w->p);
  const Dev* w16 = find(h, name + ".w16");
  c->w16 = w16 ? w16->p : nullptr;
  const Dev* b = find(h, name + ".bias");
  c->bias = b ? static_cast<const float*>(b->p) : nullptr;
  c->mode = mm[0]; c->taps = mm[1]; c->dil = mm[2]; c->pad_left = mm[3]; c->cin = mm[4]; c->cin_pad = mm[5]; c->cout = mm[6];
  c->wn = mm[7]; c->half_pad = mm[8]; c->tile_rows = mm[9]; c->small_tile_rows = mm[10]; c->small_only = mm[11]; c->n_tile = mm[12];
  c->compute16 = mm[13];
  c->algo_taps = mm[14] > 0 ? mm[14] : c->taps;
  return TTS_OK;
}

// ---- profiling hook ---------------------------------------------------------------------------------------------------------
hipEvent_t prof_event(Handle* h) {
  if (!h->event_pool.empty()) {
    hipEvent_t e = h->event_pool.back();
    h->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

bool prof_wants(const Handle* h, const std::string& name) { return h->prof_on && (h->prof_select.empty() || h->prof_select == name); }

ProfRec* prof_open(Handle* h, const std::string& name, double flops, double bytes, double elems, hipStream_t st) {
  h->prof.emplace_back();
  ProfRec* r = &h->prof.back();
  r->name = name; r->flops = flops; r->bytes = bytes; r->elems = elems;
  r->e0 = prof_event(h);
  r->e1 = prof_event(h);
  (void)hipEventRecord(r->e0, st);
  return r;
}

// ---- one conv launch (engine.Ops.conv) ----------------------------------------------------------------------------------
struct T2 {  // a [rows, cols] view
  void* p;
  int ld;
  int bits;  // 32, or 16 (format = the handle's precision)
  T2() : p(nullptr), ld(0), bits(32) {}
  T2(void* p_, int ld_, int bits_ = 32) : p(p_), ld(ld_), bits(bits_) {}
};

struct ConvOpt {
  int pre = TTS_PRE_NONE;
  float pre_slope = 0.f;
  int act = TTS_ACT_NONE;
  float alpha = 1.f;
  const float* seqvec = nullptr; int ld_seqvec = 0;
  const float* preadd = nullptr; int ld_preadd = 0;
  T2 res; float res_scale = 1.f;
  const float* aux = nullptr; int ld_aux = 0;
  bool accumulate = false;
  bool fp32_only = false;  // run on the fp32 MFMA path even if a 16-bit copy exists
  bool no_split_k = false; // never the split-K form (the position tables: their rows must not depend on the table's size)
  const float *snake_alpha = nullptr, *snake_beta = nullptr, *snake_filt = nullptr;
};

int conv(Handle* h, const ConvW& cw, T2 x, T2 y, const Layout& l, hipStream_t st, const ConvOpt& o = ConvOpt()) {
  int tile_rows = cw.tile_rows;
  if (h->small_tile_blocks && cw.small_tile_rows && !cw.small_only) {
    const int cols = cw.mode == TTS_MODE_LINEAR ? cw.wn : cw.half_pad;
    const long long grid = (long long)((l.total + cw.tile_rows - 1) / cw.tile_rows) * (cols / cw.n_tile);
    if (grid < h->small_tile_blocks) tile_rows = cw.small_tile_rows;
  }
  TileTab tt;
  TTS_TRY(tiles_of(h, l, tile_rows, st, &tt));
  TtsConvDesc d;
  memset(&d, 0, sizeof(d));
  bool use16 = !o.fp32_only && h->cfg.precision != TTS_COMPUTE_F32 && cw.w16 != nullptr;
  if (use16 && cw.compute16 == TTS_COMPUTE_F32X3) {
    // the split fp32 product (same rule as engine.Ops.conv): not in the phoneme stages (exact fp32 upstream of the rounded
    // durations), and not where the exact fp32 split-K form is the fast one (frame stages on grids of a few workgroups)
    const int cols = cw.mode == TTS_MODE_LINEAR ? cw.wn : cw.half_pad;
    const bool small = (long long)tt.n * (tile_rows / 64) * ((cols + 63) / 64) <= 128;
    if (h->split_mode == 2 || (h->split_mode == 1 && small)) use16 = false;
  }
  d.x = static_cast<const float*>(x.p); d.ldx = x.ld; d.cin = cw.cin;
  d.w = use16 ? cw.w16 : cw.w; d.cin_pad = cw.cin_pad; d.wn = cw.wn; d.half_pad = cw.half_pad;
  d.bias = cw.bias;
  d.y = static_cast<float*>(y.p); d.ldy = y.ld; d.cout = cw.cout;
  d.taps = cw.taps; d.dil = cw.dil; d.pad_left = cw.pad_left;
  d.pre_act = o.pre; d.pre_slope = o.pre_slope;
  d.snake_alpha = o.snake_alpha; d.snake_beta = o.snake_beta; d.snake_filt = o.snake_filt;
  d.mode = cw.mode; d.act = o.act; d.alpha = o.alpha;
  d.seqvec = o.seqvec; d.ld_seqvec = o.ld_seqvec;
  d.preadd = o.preadd; d.ld_preadd = o.ld_preadd;
  d.res = static_cast<const float*>(o.res.p); d.ld_res = o.res.ld; d.res_scale = o.res_scale;
  d.aux = o.aux; d.ld_aux = o.ld_aux;
  d.accumulate = o.accumulate ? 1 : 0;
  d.compute = use16 ? cw.compute16 : TTS_COMPUTE_F32;
  const bool any16 = x.bits == 16 || y.bits == 16 || (o.res.p && o.res.bits == 16);
  d.io_flags = (x.bits == 16 ? TTS_IO_X_BF16 : 0) | (y.bits == 16 ? TTS_IO_Y_BF16 : 0) | ((o.res.p && o.res.bits == 16) ? TTS_IO_RES_BF16 : 0) |
               ((any16 && h->cfg.precision == TTS_COMPUTE_F16) ? TTS_IO_F16 : 0);
  // the fp32 configuration only: the fp32 layers of a 16-bit configuration keep one accumulation order at every batch size (an
  // utterance's result there does not depend on the batch it is in, bit for bit - tests/test_gpu_e2e.py asserts it)
  if (h->split_mode && !o.no_split_k && !is16(h) && d.compute == TTS_COMPUTE_F32)
    d.io_flags |= h->split_mode == 2 ? TTS_IO_SPLIT_K_ALWAYS : TTS_IO_SPLIT_K;
  d.tiles = tt.dev; d.n_tiles = tt.n; d.tile_rows = tile_rows;
  if (h->prof_on) {  // same class names and algorithmic work as profiling.py (kernel_class / ConvTimer.add)
    const bool dual = cw.mode != TTS_MODE_LINEAR;
    const int bm = tile_rows != cw.tile_rows ? tile_rows : cw.tile_rows, bn = tile_rows != cw.tile_rows ? 64 : cw.n_tile;
    char name[96];
    int nlen = snprintf(name, sizeof(name), "conv1d_%s<%dx%d%s>", d.compute == TTS_COMPUTE_F32 ? "f32" : (d.compute == TTS_COMPUTE_BF16 ? "bf16" : (d.compute == TTS_COMPUTE_F16 ? "f16" : "f32x3")),
                        bm, bn, dual ? ",dual" : "");
    if (h->prof_detail)  // per-shape classes (tools/conv_shapes.py): where the small-GEMM time goes
      snprintf(name + nlen, sizeof(name) - nlen, "[%dx%d k%d r%d]", cw.cin, cw.cout * (dual ? 2 : 1), cw.taps, l.total);
    if (prof_wants(h, name)) {
      double rows = 0;
      for (int n : l.lengths) rows += n;
      const double ctot = (double)cw.cout * (dual ? 2 : 1);
      ProfRec* r = prof_open(h, name, 2.0 * rows * cw.cin * ctot * cw.algo_taps,
                             rows * (cw.cin * (x.bits / 8.0) + cw.cout * (y.bits / 8.0)) + cw.taps * cw.cin * ctot * (d.compute == TTS_COMPUTE_F32 ? 4 : 2),
                             rows * cw.cout, st);
      const int rc = tts_conv1d(&d, st);
      (void)hipEventRecord(r->e1, st);
      return rc;
    }
  }
  return tts_conv1d(&d, st);
}

// ---- model description resolved from the weight table ---------------------------------------------------------------------
struct Block {
  const float *ln_g[5], *ln_b[5];  // norm_ff_macaron, norm_mha, norm_conv, norm_ff, norm_final
  ConvW ffm1, ffm2, qkv, out, pos, pw1, pw2, ff1, ff2;
  const float *u, *v, *dw_w, *dw_b;
  const Dev *ffm_fused = nullptr, *ff_fused = nullptr;  // packing.pack_ffn (16-bit configurations, kernel size 1), or null
};

int block_of(const Handle* h, const std::string& p, Block* b) {
  static const char* ln[5] = {"norm_ff_macaron", "norm_mha", "norm_conv", "norm_ff", "norm_final"};
  for (int i = 0; i < 5; ++i) {
    TTS_TRY(fvec(h, p + ln[i] + ".g", &b->ln_g[i]));
    TTS_TRY(fvec(h, p + ln[i] + ".b", &b->ln_b[i]));
  }
  TTS_TRY(conv_of(h, p + "ffm.w1", &b->ffm1));
  TTS_TRY(conv_of(h, p + "ffm.w2", &b->ffm2));
  TTS_TRY(conv_of(h, p + "ff.w1", &b->ff1));
  TTS_TRY(conv_of(h, p + "ff.w2", &b->ff2));
  b->ffm_fused = find(h, p + "ffm.fused");
  b->ff_fused = find(h, p + "ff.fused");
  TTS_TRY(conv_of(h, p + "qkv", &b->qkv));
  TTS_TRY(conv_of(h, p + "out", &b->out));
  TTS_TRY(conv_of(h, p + "pos", &b->pos));
  TTS_TRY(conv_of(h, p + "pw1", &b->pw1));
  TTS_TRY(conv_of(h, p + "pw2", &b->pw2));
  TTS_TRY(fvec(h, p + "u", &b->u));
  TTS_TRY(fvec(h, p + "v", &b->v));
  TTS_TRY(fvec(h, p + "dw_w", &b->dw_w));
  TTS_TRY(fvec(h, p + "dw_b", &b->dw_b));
  return TTS_OK;
}

// 16-bit configuration (bf16 / fp16: 16-bit tensors between the fused kernels)?  TTS_COMPUTE_F32X3 is a 32-bit configuration in
// every respect but the dense products of its frame stages and vocoder (three fp16 MFMAs on split operands)
bool is16(const Handle* h) { return h->cfg.precision == TTS_COMPUTE_BF16 || h->cfg.precision == TTS_COMPUTE_F16; }
int bits16(const Handle* h) { return is16(h) ? 16 : 32; }

// relative position tables of both stacks for positions -(pmax-1) .. pmax-1 (Attention.py:177, PositionalEncoding.py:90-130):
// ptab[s][l][pmax - 1 + p] = linear_pos_l(pe(p)); the sinusoid table "pe" is uploaded by the host (fp32, built like the reference)
int ensure_ptabs(Handle* h, hipStream_t st) {
  const Dev* pe;
  TTS_TRY(need(h, "pe", &pe));
  const int rows = (int)pe->shape[0], pmax = (rows + 1) / 2;
  if (h->pmax == pmax && h->ptab[0]) return TTS_OK;
  TTS_TRY(hip_ok(hipStreamSynchronize(st), "position tables: sync"));
  for (int s = 0; s < 2; ++s) {
    if (h->ptab[s]) (void)hipFree(h->ptab[s]);
    TTS_TRY(hip_ok(hipMalloc(reinterpret_cast<void**>(&h->ptab[s]), (size_t)6 * rows * ATT * sizeof(float)), "position tables: hipMalloc"));
  }
  int one = rows;
  const Layout l = Layout::make(&one, 1, 1);
  for (int s = 0; s < 2; ++s)
    for (int b = 0; b < 6; ++b) {
      ConvW pos;
      TTS_TRY(conv_of(h, std::string(s ? "dec." : "enc.") + std::to_string(b) + ".pos", &pos));
      ConvOpt o;
      o.fp32_only = true;
      o.no_split_k = true;
      TTS_TRY(conv(h, pos, T2(pe->p, ATT), T2(h->ptab[s] + (size_t)b * rows * ATT, ATT), l, st, o));
    }
  h->pmax = pmax;
  return TTS_OK;
}

// One feed-forward module in one launch (tts_ffn_fused): x <- [LN_post](x + 0.5 FFN(LN(x)))
int ffn_fused(Handle* h, const Dev* fused, const ConvW& w1, const ConvW& w2, const float* g, const float* b, const float* post_g, const float* post_b,
              float* x, int R, hipStream_t st) {
  TtsFfnDesc d;
  memset(&d, 0, sizeof(d));
  d.x = x; d.ldx = ATT; d.y = x; d.ldy = ATT; d.rows = R; d.channels = ATT;
  d.ln_g = g; d.ln_b = b;
  d.w = fused->p; d.b2 = w2.bias;
  d.post_g = post_g; d.post_b = post_b;
  d.hidden = (int)(fused->bytes / (28 * 1024)) * 32;
  d.compute = w1.compute16; d.alpha = 0.5f; d.eps = 1e-12f;
  if (h->prof_on) {
    const char* name = d.compute == TTS_COMPUTE_F16 ? "ffn_fused_f16" : "ffn_fused_bf16";
    if (prof_wants(h, name)) {
      ProfRec* r = prof_open(h, name, 4.0 * R * ATT * d.hidden, (double)R * ATT * 8 + (double)fused->bytes, (double)R * ATT, st);
      const int rc = tts_ffn_fused(&d, st);
      (void)hipEventRecord(r->e1, st);
      return rc;
    }
  }
  return tts_ffn_fused(&d, st);
}

// Layers/EncoderLayer.py:62-144 x 6 on the residual stream x [rows, 192] (already scaled by sqrt(192))
int conformer(Handle* h, int stack, float* x, const Layout& l, Arena& a, hipStream_t st) {
  const int R = l.total, kernel = stack ? 31 : 7;
  if (l.max_len > h->pmax) {
    set_error("conformer: %d positions exceed the uploaded position table (pmax %d): upload a larger 'pe'", l.max_len, h->pmax);
    return TTS_E_ARG;
  }
  const int b16 = bits16(h);
  TTS_ALLOC(ln, a, float, (size_t)R * ATT);
  TTS_ALLOC(hid, a, char, (size_t)R * 1536 * (b16 / 8));
  TTS_ALLOC(qkv, a, float, (size_t)R * 3 * ATT);
  TTS_ALLOC(ctx, a, float, (size_t)R * ATT);
  TTS_ALLOC(glu, a, float, (size_t)R * ATT);
  TTS_ALLOC(dwo, a, float, (size_t)R * ATT);
  TileTab t128, t64;
  TTS_TRY(tiles_of(h, l, 128, st, &t128));
  TTS_TRY(tiles_of(h, l, 64, st, &t64));
  const size_t prow = (size_t)(2 * h->pmax - 1) * ATT;
  for (int bi = 0; bi < 6; ++bi) {
    Block b;
    TTS_TRY(block_of(h, std::string(stack ? "dec." : "enc.") + std::to_string(bi) + ".", &b));
    ConvOpt relu, half, res1;
    relu.act = TTS_ACT_RELU;
    half.alpha = 0.5f; half.res = T2(x, ATT);
    res1.res = T2(x, ATT);
    // Macaron feed-forward (EncoderLayer.py:84-90)
    // (the same decision as engine.py: 16-bit configuration and packed weights loaded - not the number of rows: an utterance's
    // result must not depend on the batch it is in)
    const bool fuse_ffn = b16 == 16 && !h->no_fused_ffn;
    if (fuse_ffn && b.ffm_fused) {
      TTS_TRY(ffn_fused(h, b.ffm_fused, b.ffm1, b.ffm2, b.ln_g[0], b.ln_b[0], nullptr, nullptr, x, R, st));
    } else {
      TTS_TRY(tts_layernorm(x, ATT, ln, ATT, b.ln_g[0], b.ln_b[0], R, ATT, 1e-12f, st));
      TTS_TRY(conv(h, b.ffm1, T2(ln, ATT), T2(hid, 1536, b16), l, st, relu));
      TTS_TRY(conv(h, b.ffm2, T2(hid, 1536, b16), T2(x, ATT), l, st, half));
    }
    // relative-position self-attention (:93-116)
    TTS_TRY(tts_layernorm(x, ATT, ln, ATT, b.ln_g[1], b.ln_b[1], R, ATT, 1e-12f, st));
    TTS_TRY(conv(h, b.qkv, T2(ln, ATT), T2(qkv, 3 * ATT), l, st));
    if (b16 == 16 && !h->no_f16_attention)  // (16-bit configurations: the contractions on the fp16 matrix cores)
      TTS_TRY(tts_relpos_attention_f16(qkv, 3 * ATT, h->ptab[stack] + bi * prow, h->pmax, b.u, b.v, ctx, ATT, HEADS, DK, t128.dev, t128.n, 128, st));
    else
      TTS_TRY(tts_relpos_attention(qkv, 3 * ATT, h->ptab[stack] + bi * prow, h->pmax, b.u, b.v, ctx, ATT, HEADS, DK, t128.dev, t128.n, 128,
                                   b16 == 16 ? 0 : (h->split_mode == 2 ? TTS_ATT_KEY_SPLIT_ALWAYS : (h->split_mode == 1 ? TTS_ATT_KEY_SPLIT : 0)), st));
    TTS_TRY(conv(h, b.out, T2(ctx, ATT), T2(x, ATT), l, st, res1));
    // convolution module (:119-125, Convolution.py:31-55)
    TTS_TRY(tts_layernorm(x, ATT, ln, ATT, b.ln_g[2], b.ln_b[2], R, ATT, 1e-12f, st));
    TTS_TRY(conv(h, b.pw1, T2(ln, ATT), T2(glu, ATT), l, st));
    TTS_TRY(tts_dwconv_swish(glu, ATT, dwo, ATT, b.dw_w, b.dw_b, ATT, kernel, t64.dev, t64.n, 64, st));
    TTS_TRY(conv(h, b.pw2, T2(dwo, ATT), T2(x, ATT), l, st, res1));
    // feed-forward (:128-133) and the block's final norm (:135-136)
    if (fuse_ffn && b.ff_fused) {  // (with the block's final norm in its epilogue)
      TTS_TRY(ffn_fused(h, b.ff_fused, b.ff1, b.ff2, b.ln_g[3], b.ln_b[3], b.ln_g[4], b.ln_b[4], x, R, st));
    } else {
      TTS_TRY(tts_layernorm(x, ATT, ln, ATT, b.ln_g[3], b.ln_b[3], R, ATT, 1e-12f, st));
      TTS_TRY(conv(h, b.ff1, T2(ln, ATT), T2(hid, 1536, b16), l, st, relu));
      TTS_TRY(conv(h, b.ff2, T2(hid, 1536, b16), T2(x, ATT), l, st, half));
      TTS_TRY(tts_layernorm(x, ATT, x, ATT, b.ln_g[4], b.ln_b[4], R, ATT, 1e-12f, st));
    }
  }
  return TTS_OK;
}

size_t conformer_bytes(size_t R) { return R * (ATT * 4 * 4 + 1536 * 4 + 3 * ATT * 4) + 8 * 256; }

// Arena sizes: ONE set of expressions for the stage entries (what they reserve) and for tts_workspace_bytes (what it promises).
size_t phone_arena_bytes(size_t R, size_t B) {
  return conformer_bytes(R) + R * (100 + 3 * ATT + 6 * 256 + 16) * 4 + B * (64 + 2 * ATT + 24 * 256 + 8) * 4 + (1 << 16);
}
size_t frame_arena_bytes(size_t RF) {
  return conformer_bytes(RF) + RF * ((80 + ATT) + 2 * ATT + 3 * 80 + 2 * 256 + ATT + 160 + 4 * ATT + ATT + 8 * ATT + 64) * 4 + (64 << 20);
}
// vocoder: R packed mel rows.  Arena 1 holds the pre-conv output [R, 512] fp32 and later the stages 1 and 3, arena 0 the stages 0 and 2
// (a stage reads its input from the other arena).  Per stage: the up-sampled tensor, the stage output and two ping-pong buffers of
// [rows, channels] - plus, where the residual steps are not fused (fp32 configuration; 256 channels), the conv intermediate and
// for BigVGAN the two stand-alone snake outputs.  16-bit configurations keep 16-bit stage tensors.
size_t vocoder_stage_bytes(size_t R, int stage, bool fused_mode, bool big) {
  static const int UPS[4] = {8, 48, 192, 384};
  const size_t ch = 256 >> stage, rows = R * UPS[stage], se = fused_mode ? 2 : 4;
  const bool fused = fused_mode && ch <= 128;
  const size_t tensors = 4 + (fused ? 0 : (big ? 3 : 1));
  return tensors * (((rows * ch * se) + 255) & ~(size_t)255) + (1 << 20);
}
size_t vocoder_arena_bytes(size_t R, int arena, bool fused_mode, bool big) {
  size_t m = arena == 1 ? R * 512 * 4 + (1 << 20) : 0;
  for (int i = arena; i < 4; i += 2) m = std::max(m, vocoder_stage_bytes(R, i, fused_mode, big));
  return m;
}
size_t with_growth_slack(size_t bytes) { return bytes + bytes / 8 + (1 << 20); }  // what arena_reserve allocates for a request

int predictor(Handle* h, const char* name, int layers, int first_cln, float* out, hipStream_t st) {
  const Layout& l = h->lp;
  const int R = l.total;
  TTS_ALLOC(a, h->phone, float, (size_t)R * 256);
  TTS_ALLOC(bb, h->phone, float, (size_t)R * 256);
  TileTab t64;
  TTS_TRY(tiles_of(h, l, 64, st, &t64));
  const float* cur = h->enc;
  int ld = ATT;
  for (int i = 0; i < layers; ++i) {
    ConvW cw;
    TTS_TRY(conv_of(h, std::string(name) + ".conv." + std::to_string(i), &cw));
    ConvOpt o;
    o.act = TTS_ACT_RELU;
    o.fp32_only = true;
    TTS_TRY(conv(h, cw, T2(const_cast<float*>(cur), ld), T2(a, 256), l, st, o));
    if (h->cfg.multispeaker) {
      const float* sc = h->cln + (size_t)(2 * (first_cln + i)) * h->B * 256;
      TTS_TRY(tts_cond_layernorm(a, 256, bb, 256, sc, sc + (size_t)h->B * 256, 256, t64.dev, t64.n, 64, st));
    } else {
      const float *g, *b;
      TTS_TRY(fvec(h, std::string(name) + ".norm." + std::to_string(i) + ".g", &g));
      TTS_TRY(fvec(h, std::string(name) + ".norm." + std::to_string(i) + ".b", &b));
      TTS_TRY(tts_layernorm(a, 256, bb, 256, g, b, R, 256, 1e-12f, st));
    }
    cur = bb;
    ld = 256;
  }
  ConvW lin;
  TTS_TRY(conv_of(h, std::string(name) + ".lin", &lin));
  ConvOpt o;
  o.fp32_only = true;
  return conv(h, lin, T2(const_cast<float*>(cur), ld), T2(out, 1), l, st, o);
}

}  // namespace

// ======================================================================================================================
int pipeline_create(const TtsConfig* cfg, Handle** out) {
  TTS_CHECK_ARG(cfg && out, "tts_create: null pointer");
  TTS_CHECK_ARG(cfg->precision >= 0 && cfg->precision <= 3, "tts_create: precision %d (0 fp32, 1 bf16, 2 fp16, 3 split fp32)", cfg->precision);
  TTS_CHECK_ARG(cfg->vocoder >= 0 && cfg->vocoder <= 2, "tts_create: vocoder %d (0 none, 1 hifigan, 2 bigvgan)", cfg->vocoder);
  Handle* h = new Handle();
  h->cfg = *cfg;
  if (cfg->small_tile_blocks > 0) h->small_tile_blocks = cfg->small_tile_blocks;
  h->no_fused_wavenet = getenv("TOUCAN_NO_FUSED_WAVENET") != nullptr;
  h->no_fused_ffn = getenv("TOUCAN_NO_FUSED_FFN") != nullptr;
  h->no_f16_attention = getenv("TOUCAN_NO_F16_ATTENTION") != nullptr;
  *out = h;
  return TTS_OK;
}

int pipeline_destroy(Handle* h) {
  if (!h) return TTS_OK;
  (void)hipDeviceSynchronize();
  for (auto& kv : h->weights)
    if (kv.second.dtype != 3 && kv.second.p) (void)hipFree(kv.second.p);
    else free(kv.second.p);
  drop_tables(h);
  for (auto& group : h->tabs)
    for (TableArena& t : group) {
      if (t.dev) (void)hipFree(t.dev);
      if (t.host) (void)hipHostFree(t.host);
      if (t.copied) (void)hipEventDestroy(t.copied);
      if (t.fence) (void)hipEventDestroy(t.fence);
    }
  for (Arena* a : {&h->phone, &h->frame, &h->voc[0], &h->voc[1]})
    if (a->base) (void)hipFree(a->base);
  for (int s = 0; s < 2; ++s)
    if (h->ptab[s]) (void)hipFree(h->ptab[s]);
  delete h;
  return TTS_OK;
}

int pipeline_load(Handle* h, const char* name, const void* host, const int64_t* shape, int ndim, int dtype) {
  TTS_CHECK_ARG(h && name && host && shape, "tts_load_weights: null pointer");
  TTS_CHECK_ARG(ndim >= 1 && ndim <= 4, "tts_load_weights(%s): ndim %d", name, ndim);
  TTS_CHECK_ARG(dtype >= 0 && dtype <= 4, "tts_load_weights(%s): dtype %d (0 f32, 1 bf16, 2 f16, 3 i32 host metadata, 4 u8)", name, dtype);
  static const int esz[5] = {4, 2, 2, 4, 1};
  Dev d;
  d.ndim = ndim;
  d.dtype = dtype;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    d.shape[i] = shape[i];
    n *= (size_t)shape[i];
  }
  d.bytes = n * esz[dtype];
  auto it = h->weights.find(name);
  if (it != h->weights.end()) {  // replacing a tensor (e.g. a longer position table): nothing may still read the old one
    (void)hipDeviceSynchronize();
    if (it->second.dtype == 3) free(it->second.p);
    else (void)hipFree(it->second.p);
    h->weights.erase(it);
    if (!strcmp(name, "pe")) h->pmax = 0;
  }
  if (dtype == 3) {  // descriptor fields: read by the host sequencer only
    d.p = malloc(std::max<size_t>(d.bytes, 4));
    memcpy(d.p, host, d.bytes);
  } else {
    TTS_TRY(hip_ok(hipMalloc(&d.p, std::max<size_t>(d.bytes, 256)), "tts_load_weights: hipMalloc"));
    TTS_TRY(hip_ok(hipMemcpy(d.p, host, d.bytes, hipMemcpyHostToDevice), "tts_load_weights: upload"));
  }
  h->weights[name] = d;
  return TTS_OK;
}

// upper bound of the workspace a batch of B utterances with at most Lmax phonemes / Tmax frames each will claim: the sum of what the
// stage entries reserve for such a batch (same expressions), arena growth slack included (tts_workspace_claimed reports the real sum)
long long pipeline_workspace_bytes(const Handle* h, int B, int Lmax, int Tmax) {
  if (!h || B <= 0) return 0;
  const size_t RP = (size_t)B * Lmax, RF = (size_t)B * (Tmax + 1);  // (frame layouts start every utterance on an even row)
  size_t total = with_growth_slack(phone_arena_bytes(RP, B)) + with_growth_slack(frame_arena_bytes(RF));
  if (h->cfg.vocoder) {
    const bool fused_mode = is16(h), big = h->cfg.vocoder == 2;
    for (int a = 0; a < 2; ++a) total += with_growth_slack(vocoder_arena_bytes(RF, a, fused_mode, big));
  }
  return (long long)total;
}

long long pipeline_workspace_claimed(const Handle* h) {
  if (!h) return 0;
  return (long long)(h->phone.cap + h->frame.cap + h->voc[0].cap + h->voc[1].cap);
}

// ---- stage A.1: encoder ----------------------------------------------------------------------------------------------
int pipeline_encoder(Handle* h, const float* text, const float* utt_emb, const int* lang_ids, const int* phone_lengths, int B,
                     hipStream_t st) {
  TTS_CHECK_ARG(h && text && phone_lengths && B > 0, "tts_encoder: bad arguments");
  TTS_CHECK_ARG(!h->cfg.multispeaker || utt_emb, "tts_encoder: the multi-speaker checkpoint needs utterance embeddings");
  h->split_mode = 2;
  TTS_TRY(tables_begin(h, 0, st));  // a batch starts: its tile tables go to the acoustic stages' other generation
  h->lp = Layout::make(phone_lengths, B, 1);
  h->B = B;
  h->text = text;
  h->have_flow = false;
  const int R = h->lp.total;
  TTS_TRY(arena_reserve(h->phone, phone_arena_bytes(R, B), st));
  TTS_TRY(ensure_ptabs(h, st));
  Arena& a = h->phone;
  int bn = B;
  const Layout lb = Layout::make(&bn, 1, 1);
  TTS_ALLOC(e_norm, a, float, (size_t)B * 64);
  h->e_norm = e_norm;
  if (utt_emb) TTS_TRY(tts_l2_normalize(utt_emb, e_norm, B, 64, st));
  ConvW embed0, embed2;
  TTS_TRY(conv_of(h, "embed0", &embed0));
  TTS_TRY(conv_of(h, "embed2", &embed2));
  TTS_ALLOC(h100, a, float, (size_t)R * 100);
  ConvOpt o0;
  o0.act = TTS_ACT_TANH;
  o0.fp32_only = true;
  TTS_TRY(conv(h, embed0, T2(const_cast<float*>(text), 62), T2(h100, 100), h->lp, st, o0));
  ConvOpt o2;
  o2.alpha = sqrtf((float)ATT);
  o2.fp32_only = true;
  if (h->cfg.multilingual && lang_ids) {  // Conformer.py:112-114
    const float* table;
    TTS_TRY(fvec(h, "lang_table", &table));
    TTS_ALLOC(lang, a, float, (size_t)B * ATT);
    TTS_TRY(tts_gather_rows(table, ATT, lang_ids, lang, ATT, B, ATT, st));
    o2.seqvec = lang;
    o2.ld_seqvec = ATT;
  }
  TTS_ALLOC(x, a, float, (size_t)R * ATT);
  TTS_TRY(conv(h, embed2, T2(h100, 100), T2(x, ATT), h->lp, st, o2));
  TTS_TRY(conformer(h, 0, x, h->lp, a, st));
  const float *g, *b;
  TTS_TRY(fvec(h, "out_norm.g", &g));
  TTS_TRY(fvec(h, "out_norm.b", &b));
  TTS_TRY(tts_layernorm(x, ATT, x, ATT, g, b, R, ATT, 1e-12f, st));
  if (h->cfg.multispeaker) {  // Conformer.py:130-134: projection of [hidden | normalised utterance embedding]
    ConvW hs_h, hs_e;
    TTS_TRY(conv_of(h, "hs_h", &hs_h));
    TTS_TRY(conv_of(h, "hs_e", &hs_e));
    TTS_ALLOC(e_proj, a, float, (size_t)B * ATT);
    ConvOpt oe;
    oe.fp32_only = true;
    TTS_TRY(conv(h, hs_e, T2(e_norm, 64), T2(e_proj, ATT), lb, st, oe));
    TTS_ALLOC(enc, a, float, (size_t)R * ATT);
    ConvOpt oh;
    oh.seqvec = e_proj;
    oh.ld_seqvec = ATT;
    TTS_TRY(conv(h, hs_h, T2(x, ATT), T2(enc, ATT), h->lp, st, oh));
    h->enc = enc;
  } else {
    h->enc = x;
  }
  TTS_ALLOC(p, a, float, R);
  TTS_ALLOC(e, a, float, R);
  TTS_ALLOC(dd, a, int, R);
  h->pitch = p; h->energy = e; h->dur = dd;
  h->cln = nullptr;
  return TTS_OK;
}

// ---- stage A.2: pitch / energy / duration predictors (gold values replace a prediction) --------------------------------
int pipeline_predictors(Handle* h, const float* gold_pitch, const float* gold_energy, const int* gold_dur, hipStream_t st) {
  TTS_CHECK_ARG(h && h->enc, "tts_variance_predictors: run tts_encoder first");
  h->split_mode = 2;
  h->tab_which = 0;
  const int R = h->lp.total, B = h->B;
  Arena& a = h->phone;
  if (h->cfg.multispeaker && !(gold_pitch && gold_energy && gold_dur)) {
    const Dev* w;
    TTS_TRY(need(h, "cln_weights", &w));
    const int n_mlp = (int)(w->bytes / 4 / tts_cln_mlp_weight_floats(64, 256));
    TTS_ALLOC(cln, a, float, (size_t)n_mlp * B * 256);
    TTS_TRY(tts_cln_mlp(h->e_norm, B, 64, 256, static_cast<const float*>(w->p), n_mlp, cln, st));
    h->cln = cln;
  }
  // conditional layer norms are numbered pitch 0..6, energy 7..8, duration 9..11 (engine.py packs them in this order)
  if (gold_pitch) TTS_TRY(hip_ok(hipMemcpyAsync(h->pitch, gold_pitch, (size_t)R * 4, hipMemcpyDeviceToDevice, st), "gold pitch"));
  else TTS_TRY(predictor(h, "pitch", 7, 0, h->pitch, st));
  if (gold_energy) TTS_TRY(hip_ok(hipMemcpyAsync(h->energy, gold_energy, (size_t)R * 4, hipMemcpyDeviceToDevice, st), "gold energy"));
  else TTS_TRY(predictor(h, "energy", 2, 7, h->energy, st));
  if (gold_dur) {
    TTS_TRY(hip_ok(hipMemcpyAsync(h->dur, gold_dur, (size_t)R * 4, hipMemcpyDeviceToDevice, st), "gold durations"));
  } else {
    TTS_ALLOC(logd, a, float, R);
    TTS_TRY(predictor(h, "duration", 3, 9, logd, st));
    TTS_TRY(tts_duration_from_log(logd, h->dur, R, st));
  }
  return TTS_OK;
}

// ---- stage A.3 / B.0: control, the one host round trip, length regulator -----------------------------------------------
int pipeline_control_regulate(Handle* h, float duration_scale, float pitch_scale, float energy_scale, float pause_scale,
                              int* frames_out, hipStream_t st) {
  TTS_CHECK_ARG(h && h->enc && h->dur, "tts_control_and_regulate: run tts_encoder and tts_variance_predictors first");
  TTS_CHECK_ARG(duration_scale > 0.f, "tts_control_and_regulate: duration_scaling_factor must be positive");
  h->tab_which = 0;
  const int R = h->lp.total, B = h->B;
  const int *pb, *pe;
  TTS_TRY(bounds_of(h, h->lp, st, &pb, &pe));
  TTS_TRY(tts_prosody_control(h->text, 62, h->pitch, h->energy, h->dur, pb, pe, B, duration_scale, pitch_scale, energy_scale, pause_scale, st));
  std::vector<int> d_host(R);
  TTS_TRY(hip_ok(hipMemcpyAsync(d_host.data(), h->dur, (size_t)R * 4, hipMemcpyDeviceToHost, st), "durations to host"));
  TTS_TRY(hip_ok(hipStreamSynchronize(st), "durations to host: sync"));
  h->frames.assign(B, 0);
  for (int u = 0; u < B; ++u) {
    long long t = 0;
    for (int i = 0; i < h->lp.lengths[u]; ++i) t += d_host[h->lp.begins[u] + i];
    h->frames[u] = t > 0 ? (int)t : h->lp.lengths[u];  // LengthRegulator.py:52-53: an all-zero utterance becomes all ones
    if (frames_out) frames_out[u] = h->frames[u];
  }
  h->lf = Layout::make(h->frames.data(), B, 2);  // even begins: the Glow squeeze is a pure re-view
  const size_t RF = h->lf.total;
  TTS_TRY(arena_reserve(h->frame, frame_arena_bytes(RF), st));
  Arena& a = h->frame;
  TTS_ALLOC(cat, a, float, RF * (80 + ATT));  // [refined mel | up-sampled text] = g_proj input
  TTS_TRY(hip_ok(hipMemsetAsync(cat, 0, RF * (80 + ATT) * 4, st), "clear frame buffer"));
  TTS_ALLOC(dec, a, float, RF * ATT);
  h->cat = cat;
  h->dec = dec;
  const float *wp, *bp, *we, *be;
  TTS_TRY(fvec(h, "pitch_w", &wp));
  TTS_TRY(fvec(h, "pitch_b", &bp));
  TTS_TRY(fvec(h, "energy_w", &we));
  TTS_TRY(fvec(h, "energy_b", &be));
  const int *fb, *fe;
  TTS_TRY(bounds_of(h, h->lf, st, &fb, &fe));
  return tts_length_regulate(h->enc, ATT, h->pitch, h->energy, wp, bp, we, be, h->dur, pb, pe, fb, B, h->lf.max_len, h->lp.max_len, ATT,
                             cat + 80, 80 + ATT, dec, ATT, sqrtf((float)ATT), st);
}

// ---- stage B.1: decoder + feat_out --------------------------------------------------------------------------------------
int pipeline_decoder(Handle* h, hipStream_t st) {
  TTS_CHECK_ARG(h && h->dec, "tts_decoder: run tts_control_and_regulate first");
  h->split_mode = 1;
  h->tab_which = 0;
  TTS_TRY(conformer(h, 1, h->dec, h->lf, h->frame, st));
  ConvW fo;
  TTS_TRY(conv_of(h, "feat_out", &fo));
  TTS_ALLOC(mel0, h->frame, float, (size_t)h->lf.total * 80);
  h->mel0 = mel0;
  h->mel = nullptr;
  return conv(h, fo, T2(h->dec, ATT), T2(mel0, 80), h->lf, st);
}

// ---- stage B.2: PostNet + residual -------------------------------------------------------------------------------------
int pipeline_postnet(Handle* h, hipStream_t st) {
  TTS_CHECK_ARG(h && h->mel0, "tts_postnet: run tts_decoder first");
  h->split_mode = 1;
  h->tab_which = 0;
  const Layout& l = h->lf;
  const int RF = l.total;
  Arena& a = h->frame;
  TTS_ALLOC(x, a, float, (size_t)RF * 256);
  TTS_ALLOC(y, a, float, (size_t)RF * 256);
  TTS_ALLOC(y80, a, float, (size_t)RF * 80);
  const long long wsn = std::max<long long>(1, tts_groupnorm_workspace_floats(l.n(), l.max_len, 32));
  TTS_ALLOC(gws, a, float, wsn);
  const int *sb, *se;
  TTS_TRY(bounds_of(h, l, st, &sb, &se));
  const float* src = h->mel0;
  int ld = 80;
  for (int i = 0; i < 5; ++i) {
    ConvW cw;
    const float *g, *b;
    const std::string p = "postnet." + std::to_string(i);
    TTS_TRY(conv_of(h, p + ".conv", &cw));
    TTS_TRY(fvec(h, p + ".g", &g));
    TTS_TRY(fvec(h, p + ".b", &b));
    if (i < 4) {
      TTS_TRY(conv(h, cw, T2(const_cast<float*>(src), ld), T2(x, 256), l, st));
      TTS_TRY(tts_groupnorm(x, 256, y, 256, g, b, 256, 32, 1e-5f, 1, nullptr, 0, sb, se, l.n(), l.max_len, gws, st));
      src = y;
      ld = 256;
    } else {
      TTS_TRY(conv(h, cw, T2(const_cast<float*>(src), ld), T2(y80, 80), l, st));
      TTS_TRY(tts_groupnorm(y80, 80, h->cat, 80 + ATT, g, b, 80, 20, 1e-5f, 0, h->mel0, 80, sb, se, l.n(), l.max_len, gws, st));
    }
  }
  h->mel = h->cat;  // refined mel, row stride 80 + 192
  h->have_flow = false;
  return TTS_OK;
}

// ---- stage B.3: PostFlow (Glow, reverse pass) ---------------------------------------------------------------------------
// z_noise: [total_frames / 2, 160] = the 0.8 N(0,1) sample in the squeezed layout of the frame layout (rows of utterance u at
// frame_begin[u] / 2; Glow.py:363 draws it inside the model, here it is an explicit input)
int pipeline_postflow(Handle* h, const float* z_noise, hipStream_t st) {
  TTS_CHECK_ARG(h && h->mel == h->cat && h->cat, "tts_postflow: run tts_postnet first");
  TTS_CHECK_ARG(z_noise, "tts_postflow: z_noise is required");
  h->split_mode = 1;
  h->tab_which = 0;
  const Layout ls = h->lf.halved();
  const int RF = h->lf.total, RS = RF / 2;
  Arena& a = h->frame;
  const int b16 = bits16(h);
  ConvW gp;
  TTS_TRY(conv_of(h, "g_proj", &gp));
  TTS_ALLOC(g, a, float, (size_t)RF * ATT);
  TTS_TRY(conv(h, gp, T2(h->cat, 80 + ATT), T2(g, ATT), h->lf, st));
  TTS_ALLOC(x, a, float, (size_t)RS * 160);
  TTS_TRY(hip_ok(hipMemcpyAsync(x, z_noise, (size_t)RS * 160 * 4, hipMemcpyDeviceToDevice, st), "flow noise"));
  TTS_ALLOC(hs, a, float, (size_t)RS * 2 * ATT);  // [hidden state | skip sum]
  // 16-bit configurations: one launch per WaveNet layer (tts_wavenet_layer); the state ping-pongs between hs and hs2
  const bool fused = is16(h) && !h->no_fused_wavenet;
  float* hs2 = nullptr;
  TileTab t64;
  if (fused) {
    hs2 = arena_alloc<float>(a, (size_t)RS * 2 * ATT);
    if (!hs2) {
      set_error("tts_postflow: workspace exhausted");
      return TTS_E_ARG;
    }
    TTS_TRY(tiles_of(h, ls, 64, st, &t64));
  }
  TTS_ALLOC(acts, a, char, (size_t)RS * ATT * (b16 / 8));
  TTS_ALLOC(cond, a, float, (size_t)RS * 8 * ATT);
  float* skip = hs + ATT;
  for (int b = 17; b >= 0; --b) {
    const std::string p = "flow." + std::to_string(b) + ".", grp = "flowgrp." + std::to_string(b / 4) + ".";
    ConvW start, end, cnd;
    TTS_TRY(conv_of(h, p + "start", &start));
    TTS_TRY(conv_of(h, p + "end", &end));
    TTS_TRY(conv_of(h, p + "cond", &cnd));
    TTS_TRY(conv(h, start, T2(x, 160), T2(hs, 2 * ATT), ls, st));            // h = start(x0); the zero half clears the skip sum
    TTS_TRY(conv(h, cnd, T2(g, 2 * ATT), T2(cond, 8 * ATT), ls, st));       // squeeze of g = re-view [RS, 384]
    float* cur = hs;
    for (int i = 0; i < 4; ++i) {
      ConvW inl, rs;
      TTS_TRY(conv_of(h, grp + "inl." + std::to_string(i), &inl));
      TTS_TRY(conv_of(h, grp + "res_skip." + std::to_string(i), &rs));
      if (fused) {
        float* nxt = cur == hs ? hs2 : hs;
        TtsWavenetDesc w;
        memset(&w, 0, sizeof(w));
        w.hs_in = cur; w.ld_in = 2 * ATT; w.hs_out = nxt; w.ld_out = 2 * ATT;
        w.cond = cond + (size_t)i * 2 * ATT; w.ld_cond = 8 * ATT;
        w.w1 = inl.w16; w.b1 = inl.bias; w.w2 = rs.w16; w.b2 = rs.bias;
        w.cout2 = rs.cout; w.compute = inl.compute16;
        w.tiles = t64.dev; w.n_tiles = t64.n; w.tile_rows = 64;
        TTS_TRY(tts_wavenet_layer(&w, st));
        cur = nxt;
        continue;
      }
      ConvOpt oi;
      oi.preadd = cond + (size_t)i * 2 * ATT;
      oi.ld_preadd = 8 * ATT;
      TTS_TRY(conv(h, inl, T2(hs, 2 * ATT), T2(acts, ATT, b16), ls, st, oi));
      ConvOpt orr;
      orr.accumulate = true;
      TTS_TRY(conv(h, rs, T2(acts, ATT, b16), i < 3 ? T2(hs, 2 * ATT) : T2(skip, 2 * ATT), ls, st, orr));
    }
    ConvOpt oe;
    oe.aux = x + 80;
    oe.ld_aux = 160;
    oe.fp32_only = true;
    TTS_TRY(conv(h, end, T2(cur + ATT, 2 * ATT), T2(x + 80, 160), ls, st, oe));  // (four fused layers end in `hs` again)
    const float *winv, *ab, *al;
    TTS_TRY(fvec(h, p + "winv", &winv));
    TTS_TRY(fvec(h, p + "an_bias", &ab));
    TTS_TRY(fvec(h, p + "an_logs", &al));
    TTS_TRY(tts_glow_invconv_actnorm(x, 160, RS, 160, winv, ab, al, st));
  }
  h->mel = x;  // unsqueeze = re-view [2 RS, 80]
  h->have_flow = true;
  return TTS_OK;
}

// where the batch's mel lives: packed [rows, 80] with row stride *ld; utterance u at frame_begin[u] (2-aligned), frames[u]
// frames (one fewer than predicted for an odd count once the flow has run)
int pipeline_mel(Handle* h, const float** mel, int* ld, int* frame_begins, int* frame_counts) {
  TTS_CHECK_ARG(h && h->mel, "tts_mel: no mel yet");
  *mel = h->mel;
  *ld = h->have_flow ? 80 : 80 + ATT;
  for (int u = 0; u < h->B; ++u) {
    if (frame_begins) frame_begins[u] = h->lf.begins[u];
    if (frame_counts) frame_counts[u] = h->have_flow ? h->lf.lengths[u] / 2 * 2 : h->lf.lengths[u];
  }
  return TTS_OK;
}

// ---- vocoders ------------------------------------------------------------------------------------------------------------
int pipeline_vocoder(Handle* h, int kind, const float* mel, int ld_mel, const int* frame_begins, const int* frame_counts, int B, float* wav,
                     hipStream_t st) {
  TTS_CHECK_ARG(h && mel && frame_begins && frame_counts && wav && B > 0, "tts_vocoder: bad arguments");
  TTS_CHECK_ARG(kind == h->cfg.vocoder, "tts_vocoder: the handle was created for vocoder %d, not %d", h->cfg.vocoder, kind);
  h->split_mode = 0;
  TTS_TRY(tables_begin(h, 1, st));
  const bool big = kind == 2;
  Layout l;
  for (int u = 0; u < B; ++u) {
    l.begins.push_back(frame_begins[u]);
    l.lengths.push_back(frame_counts[u]);
    l.total = std::max(l.total, frame_begins[u] + frame_counts[u]);
    l.max_len = std::max(l.max_len, frame_counts[u]);
  }
  const int b16 = bits16(h);
  const bool fused_mode = b16 == 16;  // 16-bit configurations run the fused residual step for C <= 128 and keep 16-bit residual streams
  const size_t e = b16 / 8;
  size_t R = l.total;
  // stage buffers: x (input of the stage) lives in the other arena
  // both arenas at their size for the whole pass (the larger of the stages each one hosts: a stage would otherwise regrow the arena
  // its predecessor's input still lives in)
  const size_t R0 = R;
  TTS_TRY(arena_reserve(h->voc[1], vocoder_arena_bytes(R0, 1, fused_mode, big), st));
  TTS_TRY(arena_reserve(h->voc[0], vocoder_arena_bytes(R0, 0, fused_mode, big), st));
  TTS_ALLOC(x0, h->voc[1], float, R * 512);
  ConvW pre;
  TTS_TRY(conv_of(h, "voc.pre", &pre));
  TTS_TRY(conv(h, pre, T2(const_cast<float*>(mel), ld_mel), T2(x0, 512), l, st));
  T2 x(x0, 512);
  static const int UP[4] = {8, 6, 4, 2};
  const float* filt = nullptr;
  const void* fir_tab = nullptr;
  if (big) {
    TTS_TRY(fvec(h, "voc.filt", &filt));
    const Dev* ft;
    TTS_TRY(need(h, "voc.fir_tab", &ft));
    fir_tab = ft->p;
  }
  int ch = 512;
  for (int i = 0; i < 4; ++i) {
    ch /= 2;
    const int u = UP[i];
    const bool fused = fused_mode && ch <= 128;
    const int sb = fused_mode ? 16 : 32;  // element size of the stage's tensors
    const size_t se = sb / 8;
    Arena& a = h->voc[i & 1];
    const size_t RU = R * u;
    a.reset();  // (sized above; the stage two steps back is dead)
    ConvW up;
    TTS_TRY(conv_of(h, "voc.ups." + std::to_string(i), &up));
    TTS_ALLOC(y, a, char, RU * ch * se);
    ConvOpt ou;
    if (!big) { ou.pre = TTS_PRE_LRELU; ou.pre_slope = 0.1f; }
    TTS_TRY(conv(h, up, x, T2(y, u * ch, sb), l, st, ou));  // transposed conv as a 3-tap polyphase conv; [R, u ch] re-viewed as [R u, ch]
    R = RU;
    l = l.scaled(u);
    TTS_ALLOC(stage_out, a, char, R * ch * se);
    TTS_ALLOC(buf0, a, char, R * ch * se);
    TTS_ALLOC(buf1, a, char, R * ch * se);
    char *t1 = nullptr, *t2 = nullptr, *sa = nullptr;
    if (!fused) {
      t1 = arena_alloc<char>(a, R * ch * se);
      if (big) {
        t2 = arena_alloc<char>(a, R * ch * se);
        sa = arena_alloc<char>(a, R * ch * se);
      }
      if (!t1 || (big && (!t2 || !sa))) {
        set_error("tts_vocoder: workspace exhausted");
        return TTS_E_ARG;
      }
    }
    TileTab trb, t256;
    if (fused) TTS_TRY(tiles_of(h, l, tts_resblock_tile_rows(ch), st, &trb));
    if (big && !fused) TTS_TRY(tiles_of(h, l, 256, st, &t256));
    const int f16flag = (sb == 16 && h->cfg.precision == TTS_COMPUTE_F16) ? TTS_IO_F16 : 0;
    for (int j = 0; j < 3; ++j) {
      char* cur = y;
      char* bufs[2] = {buf0, buf1};
      for (int dd = 0; dd < 3; ++dd) {
        const std::string p = "voc.blk." + std::to_string(i) + "." + std::to_string(j) + "." + std::to_string(dd);
        ConvW c1, c2;
        TTS_TRY(conv_of(h, p + ".c1", &c1));
        TTS_TRY(conv_of(h, p + ".c2", &c2));
        const bool last = dd == 2;
        const float *a1 = nullptr, *b1 = nullptr, *a2 = nullptr, *b2 = nullptr;
        if (big) {
          TTS_TRY(fvec(h, p + ".a1", &a1));
          TTS_TRY(fvec(h, p + ".b1", &b1));
          TTS_TRY(fvec(h, p + ".a2", &a2));
          TTS_TRY(fvec(h, p + ".b2", &b2));
        }
        char* dst = last ? stage_out : bufs[dd % 2];
        if (fused) {  // one launch per dilation step: act, conv(dil), act, conv(1), + x (and the stage mean on the last step)
          TtsResblockDesc r;
          memset(&r, 0, sizeof(r));
          r.x = reinterpret_cast<const float*>(cur); r.ldx = ch; r.y = reinterpret_cast<float*>(dst); r.ldy = ch;
          r.c = ch; r.taps = c1.taps; r.dil = c1.dil;
          r.w1 = c1.w16; r.b1 = c1.bias; r.w2 = c2.w16; r.b2 = c2.bias;
          r.act = big ? TTS_PRE_SNAKE : TTS_PRE_LRELU; r.slope = 0.1f;
          r.alpha1 = a1; r.beta1 = b1; r.alpha2 = a2; r.beta2 = b2; r.filt = filt; r.fir_tab = fir_tab;
          r.alpha = last ? 1.0f / 3.0f : 1.0f; r.res_scale = r.alpha; r.accumulate = (last && j > 0) ? 1 : 0;
          r.io_bf16 = 1; r.tiles = trb.dev; r.n_tiles = trb.n; r.tile_rows = tts_resblock_tile_rows(ch); r.compute = c1.compute16;
          char pname[32];
          snprintf(pname, sizeof(pname), "resblock_step<%d>", ch);
          if (prof_wants(h, pname)) {
            double rows = 0;
            for (int n : l.lengths) rows += n;
            ProfRec* pr = prof_open(h, pname, 2.0 * rows * ch * ch * c1.taps * 2, 2.0 * rows * ch * se + 2.0 * c1.taps * ch * ch * 2, rows * ch, st);
            const int rc = tts_resblock_step(&r, st);
            (void)hipEventRecord(pr->e1, st);
            if (rc != TTS_OK) return rc;
          } else {
            TTS_TRY(tts_resblock_step(&r, st));
          }
          cur = dst;
          continue;
        }
        T2 src2;
        ConvOpt o2;
        if (big) {  // AMP.py:51-60: a1 -> c1 -> a2 -> c2 -> + x with stand-alone anti-aliased snakes
          const int fl = (sb == 16 ? (TTS_IO_X_BF16 | TTS_IO_Y_BF16) : 0) | f16flag;
          TTS_TRY(tts_snake_aa(reinterpret_cast<const float*>(cur), ch, reinterpret_cast<float*>(sa), ch, a1, b1, filt, ch, t256.dev, t256.n, 256, fl, st));
          TTS_TRY(conv(h, c1, T2(sa, ch, sb), T2(t1, ch, sb), l, st));
          TTS_TRY(tts_snake_aa(reinterpret_cast<const float*>(t1), ch, reinterpret_cast<float*>(t2), ch, a2, b2, filt, ch, t256.dev, t256.n, 256, fl, st));
          src2 = T2(t2, ch, sb);
        } else {  // ResidualBlock.py:83-98 with LeakyReLU(0.1)
          ConvOpt o1;
          o1.pre = TTS_PRE_LRELU; o1.pre_slope = 0.1f;
          TTS_TRY(conv(h, c1, T2(cur, ch, sb), T2(t1, ch, sb), l, st, o1));
          src2 = T2(t1, ch, sb);
          o2.pre = TTS_PRE_LRELU; o2.pre_slope = 0.1f;
        }
        o2.res = T2(cur, ch, sb);
        if (last) {  // stage output = mean of the three blocks (InferenceBigVGAN.py:82-88)
          o2.alpha = 1.0f / 3.0f; o2.res_scale = 1.0f / 3.0f; o2.accumulate = j > 0;
        }
        TTS_TRY(conv(h, c2, src2, T2(dst, ch, sb), l, st, o2));
        cur = dst;
      }
    }
    x = T2(stage_out, ch, sb);
  }
  const float* pw;
  TTS_TRY(fvec(h, "voc.post_w", &pw));
  const int xflag = (x.bits == 16 ? TTS_IO_X_BF16 : 0) | ((x.bits == 16 && h->cfg.precision == TTS_COMPUTE_F16) ? TTS_IO_F16 : 0);
  if (big) {
    const float *pa, *pb;
    TTS_TRY(fvec(h, "voc.post_a", &pa));
    TTS_TRY(fvec(h, "voc.post_b", &pb));
    TileTab tp;
    const int tr = tts_conv_post_snake_tile_rows();
    TTS_TRY(tiles_of(h, l, tr, st, &tp));
    return tts_conv_post_snake(static_cast<const float*>(x.p), ch, ch, pw, h->cfg.post_bias, pa, pb, filt, wav, tp.dev, tp.n, tr, xflag, st);
  }
  TileTab tp;
  TTS_TRY(tiles_of(h, l, 256, st, &tp));
  return tts_conv_post(static_cast<const float*>(x.p), ch, ch, pw, h->cfg.post_bias, TTS_PRE_LRELU, 0.01f, wav, tp.dev, tp.n, 256, xflag, st);  // InferenceAvocodo.py:53
}

}  // namespace tts

// ======================================================================================================================
extern "C" {

#define H(h) reinterpret_cast<tts::Handle*>(h)
#define ST(s) reinterpret_cast<hipStream_t>(s)

int tts_create(const TtsConfig* cfg, TtsHandle** out) { return tts::pipeline_create(cfg, reinterpret_cast<tts::Handle**>(out)); }
int tts_destroy(TtsHandle* h) { return tts::pipeline_destroy(H(h)); }
int tts_load_weights(TtsHandle* h, const char* name, const void* host_ptr, const int64_t* shape, int32_t ndim, int32_t dtype) {
  return tts::pipeline_load(H(h), name, host_ptr, shape, ndim, dtype);
}
int64_t tts_workspace_bytes(const TtsHandle* h, int32_t B, int32_t Lmax, int32_t Tmax) {
  return tts::pipeline_workspace_bytes(reinterpret_cast<const tts::Handle*>(h), B, Lmax, Tmax);
}
int64_t tts_workspace_claimed(const TtsHandle* h) { return tts::pipeline_workspace_claimed(reinterpret_cast<const tts::Handle*>(h)); }
int tts_table_stats(const TtsHandle* h, int64_t* arena_tables, int64_t* cached_tables) {
  if (!h || !arena_tables || !cached_tables) return TTS_E_ARG;
  *arena_tables = reinterpret_cast<const tts::Handle*>(h)->n_arena_tables;
  *cached_tables = reinterpret_cast<const tts::Handle*>(h)->n_cached_tables;
  return TTS_OK;
}
int tts_encoder(TtsHandle* h, const float* text, const float* utt_emb, const int32_t* lang_ids, const int32_t* phone_lengths, int32_t B,
                tts_stream_t stream) {
  return tts::pipeline_encoder(H(h), text, utt_emb, lang_ids, phone_lengths, B, ST(stream));
}
int tts_variance_predictors(TtsHandle* h, const float* gold_pitch, const float* gold_energy, const int32_t* gold_durations, tts_stream_t stream) {
  return tts::pipeline_predictors(H(h), gold_pitch, gold_energy, gold_durations, ST(stream));
}
int tts_control_and_regulate(TtsHandle* h, float duration_scale, float pitch_scale, float energy_scale, float pause_scale,
                             int32_t* frame_counts, tts_stream_t stream) {
  return tts::pipeline_control_regulate(H(h), duration_scale, pitch_scale, energy_scale, pause_scale, frame_counts, ST(stream));
}
int tts_decoder(TtsHandle* h, tts_stream_t stream) { return tts::pipeline_decoder(H(h), ST(stream)); }
int tts_postnet(TtsHandle* h, tts_stream_t stream) { return tts::pipeline_postnet(H(h), ST(stream)); }
int tts_postflow(TtsHandle* h, const float* z_noise, tts_stream_t stream) { return tts::pipeline_postflow(H(h), z_noise, ST(stream)); }
int tts_mel(TtsHandle* h, const float** mel, int32_t* ld, int32_t* frame_begins, int32_t* frame_counts) {
  return tts::pipeline_mel(H(h), mel, ld, frame_begins, frame_counts);
}
int tts_prosody(TtsHandle* h, const int32_t** durations, const float** pitch, const float** energy) {
  tts::Handle* hh = H(h);
  if (!hh || !hh->dur) {
    tts::set_error("tts_prosody: no batch in flight");
    return TTS_E_ARG;
  }
  if (durations) *durations = hh->dur;
  if (pitch) *pitch = hh->pitch;
  if (energy) *energy = hh->energy;
  return TTS_OK;
}
int tts_profile(TtsHandle* h, int32_t enable, const char* select) {
  tts::Handle* hh = H(h);
  if (!hh) {
    tts::set_error("tts_profile: null handle");
    return TTS_E_ARG;
  }
  for (auto& r : hh->prof) {
    hh->event_pool.push_back(r.e0);
    hh->event_pool.push_back(r.e1);
  }
  hh->prof.clear();
  hh->prof_on = enable != 0;
  hh->prof_detail = enable == 2;  // 2: conv classes carry their shape ("conv1d_bf16<64x64>[192x384 k5 r10240]")
  hh->prof_select = select ? select : "";
  return TTS_OK;
}
int32_t tts_profile_count(TtsHandle* h) { return h ? (int32_t)H(h)->prof.size() : 0; }
int tts_profile_read(TtsHandle* h, int32_t index, char* name, int32_t name_cap, double* ms, double* flops, double* bytes, double* elems) {
  tts::Handle* hh = H(h);
  if (!hh || index < 0 || index >= (int32_t)hh->prof.size()) {
    tts::set_error("tts_profile_read: index %d out of range", index);
    return TTS_E_ARG;
  }
  const tts::ProfRec& r = hh->prof[index];
  float t = 0.f;
  const hipError_t e = hipEventSynchronize(r.e1);
  if (e != hipSuccess || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) {
    tts::set_error("tts_profile_read: event timing failed");
    return TTS_E_LAUNCH;
  }
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", r.name.c_str());
  if (ms) *ms = t;
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  if (elems) *elems = r.elems;
  return TTS_OK;
}
int tts_copy_prosody(TtsHandle* h, int32_t* durations, float* pitch, float* energy, tts_stream_t stream) {
  tts::Handle* hh = H(h);
  if (!hh || !hh->dur) {
    tts::set_error("tts_copy_prosody: no batch in flight");
    return TTS_E_ARG;
  }
  const size_t n = (size_t)hh->lp.total * 4;
  hipError_t e = hipSuccess;
  if (durations) e = hipMemcpyAsync(durations, hh->dur, n, hipMemcpyDeviceToDevice, ST(stream));
  if (e == hipSuccess && pitch) e = hipMemcpyAsync(pitch, hh->pitch, n, hipMemcpyDeviceToDevice, ST(stream));
  if (e == hipSuccess && energy) e = hipMemcpyAsync(energy, hh->energy, n, hipMemcpyDeviceToDevice, ST(stream));
  if (e != hipSuccess) {
    tts::set_error("tts_copy_prosody: %s", hipGetErrorString(e));
    return TTS_E_LAUNCH;
  }
  return TTS_OK;
}
int tts_copy_mel(TtsHandle* h, float* dst, int32_t ld_dst, tts_stream_t stream) {
  tts::Handle* hh = H(h);
  if (!hh || !hh->mel || !dst) {
    tts::set_error("tts_copy_mel: no mel yet / null destination");
    return TTS_E_ARG;
  }
  const int ld = hh->have_flow ? 80 : 80 + tts::ATT;
  const hipError_t e = hipMemcpy2DAsync(dst, (size_t)ld_dst * 4, hh->mel, (size_t)ld * 4, 80 * 4, hh->lf.total, hipMemcpyDeviceToDevice, ST(stream));
  if (e != hipSuccess) {
    tts::set_error("tts_copy_mel: %s", hipGetErrorString(e));
    return TTS_E_LAUNCH;
  }
  return TTS_OK;
}
int tts_vocoder_bigvgan(TtsHandle* h, const float* mel, int32_t ld_mel, const int32_t* frame_begins, const int32_t* frame_counts, int32_t B,
                        float* wav, tts_stream_t stream) {
  return tts::pipeline_vocoder(H(h), 2, mel, ld_mel, frame_begins, frame_counts, B, wav, ST(stream));
}
int tts_vocoder_hifigan(TtsHandle* h, const float* mel, int32_t ld_mel, const int32_t* frame_begins, const int32_t* frame_counts, int32_t B,
                        float* wav, tts_stream_t stream) {
  return tts::pipeline_vocoder(H(h), 1, mel, ld_mel, frame_begins, frame_counts, B, wav, ST(stream));
}

// The whole pass for one ragged batch.  z_noise == NULL skips the flow (the refined mel is vocoded); wav == NULL skips the vocoder.
// wav_capacity: samples the caller's buffer holds; needs 384 * (last frame_begin + frame_count) - returned through *wav_needed
// when it does not fit (TTS_E_ARG), so that a caller can size the buffer and call tts_vocoder_* on tts_mel()'s result.
int tts_synthesize_batch(TtsHandle* h, const float* text, const float* utt_emb, const int32_t* lang_ids, const int32_t* phone_lengths,
                         int32_t B, const float* gold_pitch, const float* gold_energy, const int32_t* gold_durations, float duration_scale,
                         float pitch_scale, float energy_scale, float pause_scale, const float* z_noise, int32_t* frame_begins,
                         int32_t* frame_counts, float* wav, int64_t wav_capacity, int64_t* wav_needed, tts_stream_t stream) {
  tts::Handle* hh = H(h);
  int rc;
  if ((rc = tts_encoder(h, text, utt_emb, lang_ids, phone_lengths, B, stream)) != TTS_OK) return rc;
  if ((rc = tts_variance_predictors(h, gold_pitch, gold_energy, gold_durations, stream)) != TTS_OK) return rc;
  if ((rc = tts_control_and_regulate(h, duration_scale, pitch_scale, energy_scale, pause_scale, nullptr, stream)) != TTS_OK) return rc;
  if ((rc = tts_decoder(h, stream)) != TTS_OK) return rc;
  if ((rc = tts_postnet(h, stream)) != TTS_OK) return rc;
  if (z_noise && (rc = tts_postflow(h, z_noise, stream)) != TTS_OK) return rc;
  const float* mel;
  int32_t ld;
  std::vector<int32_t> fb(B), fc(B);
  if ((rc = tts_mel(h, &mel, &ld, fb.data(), fc.data())) != TTS_OK) return rc;
  int64_t need = 0;
  for (int u = 0; u < B; ++u) {
    if (frame_begins) frame_begins[u] = fb[u];
    if (frame_counts) frame_counts[u] = fc[u];
    need = std::max<int64_t>(need, 384ll * (fb[u] + fc[u]));
  }
  if (wav_needed) *wav_needed = need;
  if (!wav || hh->cfg.vocoder == 0) return TTS_OK;
  if (wav_capacity < need) {
    tts::set_error("tts_synthesize_batch: the waveform buffer holds %lld samples, %lld are needed", (long long)wav_capacity, (long long)need);
    return TTS_E_ARG;
  }
  return hh->cfg.vocoder == 2 ? tts_vocoder_bigvgan(h, mel, ld, fb.data(), fc.data(), B, wav, stream)
                              : tts_vocoder_hifigan(h, mel, ld, fb.data(), fc.data(), B, wav, stream);
}

}  // extern "C"
