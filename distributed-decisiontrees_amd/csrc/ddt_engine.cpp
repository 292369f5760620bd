// ddt_engine.cpp -- host side of libddt.so: the engine object, the scoring / classify launches and their workspaces, the pinned
// three-slot tuple feeder, and the C-ABI of include/ddt.h.  (Round 6 split the rest out: ddt_model.cpp = wire-format validation and
// parsing, ddt_image.cpp = device images / rank tables / pre-pass images + their test hooks, ddt_choice.cpp = which kernel.)
//
// Reference interfaces restated here (the reference has no host software; these are its hardware
// contracts): CSR map rtl/DTEngine/EngineCSR.sv:190-305; stream order and framing
// rtl/DTEngine/PCIeReceiver.sv:136-139,230-312; line packing rtl/DTEngine/core/PipelinedMUX.sv:65;
// model store rtl/DTEngine/core/DTPU.sv:282-354; result packing rtl/DTEngine/ResultsCombiner.sv:136-160.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "ddt_engine_priv.h"

// Persistent staging threads of one engine (round 2 spawned std::threads per chunk).  The caller's thread takes slice 0.
struct ddt_copy_pool {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  uint64_t generation = 0;
  size_t remaining = 0;
  bool stop = false;
  char* dst = nullptr;
  const char* src = nullptr;
  size_t bytes = 0, slice = 0, parts = 0;

  explicit ddt_copy_pool(int n_workers) {
    try {
      workers.reserve((size_t)n_workers);
      for (int w = 0; w < n_workers; ++w) workers.emplace_back([this, w] { run((size_t)w + 1u); });
    } catch (...) {  // a thread could not be started: the ones that run must be joined before the members go away
      shutdown();
      throw;
    }
  }
  ~ddt_copy_pool() { shutdown(); }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_go.notify_all();
    for (std::thread& t : workers)
      if (t.joinable()) t.join();
    workers.clear();
  }
  void piece(size_t i) const {
    const size_t b = i * slice;
    if (i < parts && b < bytes) memcpy(dst + b, src + b, b + slice <= bytes ? slice : bytes - b);
  }
  void run(size_t index) {
    uint64_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m);
      cv_go.wait(lk, [&] { return stop || generation != seen; });
      if (stop) return;
      seen = generation;
      lk.unlock();
      piece(index);
      lk.lock();
      if (--remaining == 0) cv_done.notify_one();
    }
  }
  void copy(void* d, const void* s, size_t n) {
    const size_t min_slice = 2u << 20;
    size_t want = n / min_slice;
    if (want > workers.size() + 1u) want = workers.size() + 1u;
    if (want <= 1) {
      memcpy(d, s, n);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m);
      dst = static_cast<char*>(d);
      src = static_cast<const char*>(s);
      bytes = n;
      parts = want;
      slice = ((n + want - 1) / want + 4095u) & ~(size_t)4095u;
      remaining = workers.size();  // every worker reports back, the ones without a piece at once
      ++generation;
    }
    cv_go.notify_all();
    piece(0);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return remaining == 0; });
  }
};


using namespace ddt;

extern "C" {
extern const int ddt_build_s2_checked, ddt_build_dma_checked;  // ddt_checks.cpp
}

namespace ddt {

void free_images(ddt_engine* e) {
  for (void** p : {&e->d_mc_img, &e->d_mc_img_slow, &e->d_fmap}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  e->mc_seg_chunks = 0;
  for (Ensemble& m : e->ens) {
    for (void** p : {&m.d_img, &m.d_img_slow, &m.d_tables, &m.d_tabK, &m.d_tabS, &m.d_prepass}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    for (Q16Part& part : m.parts) free_rank_device(part.rank);
    m.parts.clear();
    m.img_bytes = 0;
  }
}

void free_q16_workspace(ddt_engine* e) {
  for (int k = 0; k < kQSlots; ++k) {
    for (void** p : {&e->q_xT[k], &e->q_q[k], &e->q_flags[k], &e->q_state[k]}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    if (e->q_split[k]) (void)hipFree(e->q_split[k]);
    e->q_split[k] = nullptr;
    e->q_split_floats[k] = 0;
    e->q_rows[k] = 0;
    e->q_xT_valid[k] = false;
  }
}

int ensure_q16_workspace(ddt_engine* e, size_t n) {
  const uint64_t rows = (n + 1023) / 1024 * 1024;
  const int k = e->q_slot;
  // the transposed fp32 intermediate is only needed by the two-kernel pre-pass
  bool need_xT = e->sparse ? e->sp_rank.prepass.groups == 0 : (e->ens.empty() || e->ens[0].prepass.groups == 0);
  const bool r32 = e->sparse && variant(e->variant_id).r32();  // 32-bit rank words, a flag per tile of 128 tuples at least
  bool in_parts = false;  // the sum's state between the parts' launches; tables per part
  if (!e->sparse)
    for (const Ensemble& m : e->ens) in_parts = in_parts || !m.parts.empty();
  if (in_parts) {
    need_xT = false;
    for (const Ensemble& m : e->ens)
      for (const Q16Part& part : m.parts) need_xT = need_xT || part.rank.prepass.groups == 0;
  }
  if (rows <= e->q_rows[k] && (!need_xT || e->q_xT[k]) && (!in_parts || e->q_state[k])) return DDT_OK;
  HIP_TRY(e, hipDeviceSynchronize());
  for (void** p : {&e->q_xT[k], &e->q_q[k], &e->q_flags[k], &e->q_state[k]}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  const uint64_t cap = rows > e->q_rows[k] ? rows : e->q_rows[k];
  e->q_rows[k] = 0;
  e->q_xT_valid[k] = false;
  const uint32_t W = e->sparse ? tuple_words(e->p) : q16_words(e);
  if (need_xT) HIP_TRY(e, hipMalloc(&e->q_xT[k], cap * W * 4));
  HIP_TRY(e, hipMalloc(&e->q_q[k], cap * W * (r32 ? 4 : 2)));
  if (in_parts) HIP_TRY(e, hipMalloc(&e->q_state[k], cap * 2 * sizeof(float)));
  HIP_TRY(e, hipMalloc(&e->q_flags[k], (cap / (r32 ? 128 : 1024) + 2 + 2 * kQ16GroupedCounters + kQ16TileCounterWords) * 4));  // + the 8-byte work counters of the fused / grouped pre-pass + the _p kernels' tile counter
  e->q_rows[k] = cap;
  return DDT_OK;
}

void fill_args(const ddt_engine* e, const Ensemble& m, const void* d_tuples, size_t n, float* d_scores, ScoreArgs* a) {
  a->img = reinterpret_cast<const uint4*>(m.d_img);
  a->tuples = reinterpret_cast<const uint32_t*>(d_tuples);
  a->out = d_scores;
  a->n = n;
  a->tuple_words = variant(e->variant_id).kind == kKindQ16 ? q16_words(e) : tuple_words(e->p);  // (feature compaction: what the kernels see)
  a->n_trees = m.img_trees;
  a->n_chunks = m.img_chunks;
  a->levels = e->p.num_levels;
  a->clusters = e->p.clusters_per_tuple;
  a->miss_raw = e->p.missing_bits;
  a->miss_key = e->p.cmp_mode ? kMissSentinelIeee : e->p.missing_bits;
  a->ieee = e->p.cmp_mode;
  a->sum_mode = e->p.sum_mode;
  a->aux = nullptr;
  a->top_levels = 0;
  a->ev_mid = nullptr;
  a->num_cus = e->prop.multiProcessorCount > 0 ? (uint32_t)e->prop.multiProcessorCount : 256u;
  a->stream_blocks_per_cu = (uint32_t)e->stream_blocks_per_cu;
  a->stream_res_tiles = (uint32_t)e->stream_res_tiles;
  // (default: a window every 30 us, in ticks of the device's wall clock)
  a->stream_window_ticks = e->stream_window_ticks ? (uint32_t)e->stream_window_ticks : (uint32_t)(30ull * (uint64_t)e->wall_clock_khz / 1000ull);
  a->stream_res_off = 0;
}

void feeder_free(ddt_engine* e) {
  for (int b = 0; b < kFeederSlots; ++b) {
    if (e->pin_in[b]) (void)hipHostFree(e->pin_in[b]);
    if (e->pin_out[b]) (void)hipHostFree(e->pin_out[b]);
    if (e->dev_in[b]) (void)hipFree(e->dev_in[b]);
    if (e->dev_out[b]) (void)hipFree(e->dev_out[b]);
    e->pin_in[b] = e->pin_out[b] = e->dev_in[b] = e->dev_out[b] = nullptr;
  }
  e->feeder_cap_rows = e->feeder_cap_words = e->feeder_cap_outs = 0;
}

// `outs` = 4-byte output words per row (1 for scores; K + 1 for classify: K class scores + label)
int feeder_reserve(ddt_engine* e, size_t rows, size_t words, size_t outs) {
  if (e->feeder_cap_rows >= rows && e->feeder_cap_words >= words && e->feeder_cap_outs >= outs) return DDT_OK;
  feeder_free(e);
  if (!e->copy_stream) HIP_TRY(e, hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  for (int b = 0; b < kFeederSlots; ++b) {
    if (!e->fs[b]) HIP_TRY(e, hipStreamCreateWithFlags(&e->fs[b], hipStreamNonBlocking));
    if (!e->fe[b]) HIP_TRY(e, hipEventCreateWithFlags(&e->fe[b], hipEventDisableTiming));
    if (!e->fe_in[b]) HIP_TRY(e, hipEventCreateWithFlags(&e->fe_in[b], hipEventDisableTiming));
    HIP_TRY(e, hipHostMalloc(&e->pin_in[b], rows * words * 4, hipHostMallocDefault));
    HIP_TRY(e, hipHostMalloc(&e->pin_out[b], rows * outs * 4, hipHostMallocDefault));
    HIP_TRY(e, hipMalloc(&e->dev_in[b], rows * words * 4));
    HIP_TRY(e, hipMalloc(&e->dev_out[b], rows * outs * 4));
  }
  e->feeder_cap_rows = rows;
  e->feeder_cap_words = words;
  e->feeder_cap_outs = outs;
  return DDT_OK;
}

int ensure_q16_workspace(ddt_engine* e, size_t n);

// ---- kernel_timing: a ring of event triples {start, before the scoring kernel, end} ----------------------------------------------
// fold the oldest pending triples into the counters until at most keep_pending are left (waits for their end events)
void timing_resolve(ddt_engine* e, int keep_pending) {
  while (e->tev_pending > keep_pending) {
    hipEvent_t* t = e->tev[(e->tev_head - e->tev_pending + 2 * ddt_engine::kTimingRing) % ddt_engine::kTimingRing];
    float pre = 0.f, sc = 0.f;
    if (hipEventSynchronize(t[2]) == hipSuccess && hipEventElapsedTime(&pre, t[0], t[1]) == hipSuccess &&
        hipEventElapsedTime(&sc, t[1], t[2]) == hipSuccess) {
      e->st.last_prepass_ms = pre;
      e->st.last_score_ms = sc;
      e->st.sum_prepass_ms += pre;
      e->st.sum_score_ms += sc;
      e->st.timed_launches++;
    } else {
      (void)hipGetLastError();
    }
    e->tev_pending--;
  }
}

int timing_begin(ddt_engine* e, hipStream_t s) {
  timing_resolve(e, ddt_engine::kTimingRing - 1);  // a full ring: wait for the oldest launch
  hipEvent_t* t = e->tev[e->tev_head];
  for (int i = 0; i < 3; ++i)
    if (!t[i]) HIP_TRY(e, hipEventCreate(&t[i]));
  HIP_TRY(e, hipEventRecord(t[0], s));
  e->tev_cur = t;
  return DDT_OK;
}

int timing_end(ddt_engine* e, hipStream_t s) {
  hipEvent_t* t = e->tev_cur;
  e->tev_cur = nullptr;
  HIP_TRY(e, hipEventRecord(t[2], s));
  e->tev_head = (e->tev_head + 1) % ddt_engine::kTimingRing;
  e->tev_pending++;
  return DDT_OK;
}

// Small batches on the plain rank-quantised kernels (Variant::has_split): into how many slices to cut the launch (split == 0: one block per tile as always).
// One block walks the whole ensemble for its 1024 tuples -- 0.35 ms per call at 1000 trees, whatever the batch, while a batch of a few tiles
// leaves most CUs idle.  The walks are independent; the ADDS have the reference's order (per cluster acc <- x_g + acc over its PU groups,
// FPAggregator.v:79-131; then total <- acc_c + total over the clusters, Core.sv:486-541) and are left to launch_cm_combine:
//   batches whose (tile, cluster) blocks fill the chip: a slice = a cluster, whose accumulator the block carries itself (one partial per cluster);
//   smaller ones: a slice = `len` consecutive PU groups, one partial per group -- down to one block per (tile, group) for a single tile.
struct SplitPlan {
  uint32_t split = 0, len = 0;  // slices; chunks per slice (0: the slices are the clusters)
  uint32_t partials = 0;        // partial sums per tuple
};
// a partial sum per PU group costs 4 B x groups per tuple of workspace (and of traffic each way): no such plan beyond this many bytes per call
constexpr uint64_t kSplitWorkspaceCap = 256ull << 20;
bool split_fits(uint32_t partials, size_t n) { return (uint64_t)partials * ((n + 1023) / 1024 * 1024) * 4ull <= kSplitWorkspaceCap; }
// the partial sums of a cut launch: `floats` of them in this feeder slot's workspace
int ensure_split_workspace(ddt_engine* e, uint64_t floats, float** out) {
  const int k = e->q_slot;
  if (e->q_split_floats[k] < floats) {
    HIP_TRY(e, hipDeviceSynchronize());  // (earlier calls, on whatever stream, may still read the old buffer: as ensure_q16_workspace)
    if (e->q_split[k]) (void)hipFree(e->q_split[k]);
    e->q_split[k] = nullptr;
    e->q_split_floats[k] = 0;
    HIP_TRY(e, hipMalloc(&e->q_split[k], floats * sizeof(float)));
    e->q_split_floats[k] = floats;
  }
  *out = reinterpret_cast<float*>(e->q_split[k]);
  return DDT_OK;
}
static SplitPlan cluster_split_of(const ddt_engine* e, const Variant& v, const Ensemble& m, size_t n, bool reuse_prepass, bool all_classes) {
  SplitPlan sp;
  if (!v.has_split() || e->q16_cluster_split == 0 || all_classes || reuse_prepass || e->num_classes > 1 || n == 0) return sp;
  if (!m.parts.empty() && !v.deep()) return sp;  // (parts: the deep kernels' cut form only -- every group's sum goes out, no state between the parts)
  if (e->p.sum_mode == 1u) return sp;  // (the fp64 sum runs in stream order over the trees: nothing to cut)
  const uint32_t C = e->p.clusters_per_tuple ? e->p.clusters_per_tuple : 1u, real = (m.trees() + 7u) / 8u;
  if (v.deep()) {  // always runs of PU groups, a partial sum per group of the image (its EMPTY padding groups included: a slice ends where its part's image ends)
    const uint64_t tiles = (n + 1023) / 1024;
    // (automatic: up to 160 tiles -- the cut form runs one block of 16 waves per CU and writes a partial sum per group: 512 x d12 at 64 tiles 273 against 586 us,
    // at 256 tiles 624 against 610, profiles/r06_small_batches.md)
    if (real < 2u || (e->q16_cluster_split < 0 && tiles > std::min<uint64_t>(e->q16_split_max_tiles, 160u))) return sp;
    const uint32_t slots = 2u * (e->prop.multiProcessorCount > 0 ? (uint32_t)e->prop.multiProcessorCount : 256u);
    uint32_t want = e->q16_split_groups > 0 ? (uint32_t)e->q16_split_groups : (uint32_t)(slots / tiles);
    want = want > real ? real : want < 1u ? 1u : want;
    sp.len = (real + want - 1u) / want;
    sp.split = (real + sp.len - 1u) / sp.len;
    sp.partials = m.img_chunks * (uint32_t)v.chunk_trees / 8u;
    if ((sp.split < 2u && m.parts.empty()) || !split_fits(sp.partials, n)) sp = SplitPlan();
    return sp;
  }
  const uint32_t gpc = (uint32_t)v.chunk_trees / 8u, chunks = (real + gpc - 1u) / gpc;  // PU groups per chunk; chunks that hold a real tree
  if (chunks < 2u || (C & (C - 1u)) != 0u) return sp;
  const uint64_t tiles = (n + 1023) / 1024;
  if (e->q16_cluster_split < 0 && tiles > e->q16_split_max_tiles) return sp;
  const uint32_t by_cluster = v.cm() ? (C < real ? C : real) : 1u;  // (an image in stream order has no runs of a cluster's groups)
  const uint32_t slots = 2u * (e->prop.multiProcessorCount > 0 ? (uint32_t)e->prop.multiProcessorCount : 256u);  // resident blocks (two per CU)
  if (e->q16_split_groups != 0 && tiles * by_cluster < slots) {  // the clusters alone leave CUs idle: runs of groups, about one round of blocks
    uint32_t want = (uint32_t)(slots / tiles);
    if (e->q16_split_groups > 0) want = (uint32_t)e->q16_split_groups;  // (forced slice count: tests, A/B)
    if (want > chunks) want = chunks;
    if (want > by_cluster || e->q16_split_groups > 0) {
      sp.len = (chunks + want - 1u) / want;
      sp.split = (chunks + sp.len - 1u) / sp.len;
      sp.partials = chunks * gpc;
      if (sp.split >= 2u && split_fits(sp.partials, n)) return sp;
      sp = SplitPlan();
    }
  }
  if (by_cluster < 2u) return sp;
  sp.split = sp.partials = by_cluster;
  return sp;
}

// all_classes (only with e->mc_seg_chunks != 0, a "_p" kernel): ONE launch walks every class -- d_scores = [K][n] per-class sums (may
// be NULL), labels (may be NULL) = their argmax
int launch_score(ddt_engine* e, const Ensemble& m, const void* d_tuples, size_t n, float* d_scores, hipStream_t s,
                 bool reuse_prepass = false, bool all_classes = false, int32_t* labels = nullptr) {
  ScoreArgs a;
  fill_args(e, m, d_tuples, n, d_scores, &a);
  const Variant& v = variant(e->variant_id);
  Q16Aux qa;
  if (v.kind == kKindQ16) {
    int rc = ensure_q16_workspace(e, n);
    if (rc) return rc;
    if (!reuse_prepass) e->q_xT_valid[e->q_slot] = false;  // a new batch
    qa.xT = reinterpret_cast<uint32_t*>(e->q_xT[e->q_slot]);
    qa.q = reinterpret_cast<uint16_t*>(e->q_q[e->q_slot]);
    qa.tile_flags = reinterpret_cast<uint32_t*>(e->q_flags[e->q_slot]);
    const Ensemble& tm = e->ens[0];  // the rank tables are shared by all classes and owned by the first ensemble
    qa.tables = reinterpret_cast<const uint32_t*>(tm.d_tables);
    qa.tabP = reinterpret_cast<const uint32_t*>(tm.d_tabK);
    qa.tabS = reinterpret_cast<const uint16_t*>(tm.d_tabS);
    qa.Kpad = tm.Kpad;
    qa.skip_prepass = reuse_prepass ? 1u : 0u;
    qa.prepass_img = reinterpret_cast<const uint4*>(tm.d_prepass);
    qa.prepass = tm.prepass;
    qa.img_slow = reinterpret_cast<const uint4*>(m.d_img_slow);
    qa.n_pad = (n + 1023) / 1024 * 1024;
    qa.real_groups = (m.trees() + 7u) / 8u;
    // the image's EMPTY padding (trees up to whole chunks) is not walked: in tree order the real trees end after ceil(T / U) sub-groups,
    // in a cluster-major image the partly filled PU group may sit in the middle -- whole PU groups count there
    const uint32_t U = (uint32_t)v.ilp_trees;
    qa.walk_subgroups = e->q16_walk_padding ? 0u : (v.opt & 4) ? qa.real_groups * 8u / U : (m.trees() + U - 1u) / U;
    qa.prepass_nt = (uint32_t)e->q16_prepass_nt;
    qa.fmap = reinterpret_cast<const uint32_t*>(e->d_fmap);  // (feature compaction)
    qa.in_words = tuple_words(e->p);
    if (all_classes) {
      a.img = reinterpret_cast<const uint4*>(e->d_mc_img);
      qa.img_slow = reinterpret_cast<const uint4*>(e->d_mc_img_slow);
      a.n_trees = m.img_trees * e->num_classes;
      a.n_chunks = e->mc_seg_chunks * e->num_classes;
      qa.n_segs = e->num_classes;
      qa.seg_chunks = e->mc_seg_chunks;
      qa.labels = labels;
      // trees per class mod 8 in 1..4: the second sub-group of the partly filled PU group is all padding.  That group is the last of its
      // CLUSTER's run in the cluster-major image (cm_position), which is the image's end only for some (trees, clusters): the kernel is
      // told the chunk (one PU group per chunk at CT = 8) instead of assuming the last one
      qa.seg_tail_left = 0u;
      if (m.trees() % 8u >= 1u && m.trees() % 8u <= 4u && (uint32_t)v.chunk_trees == 8u) {
        const uint32_t Cc = e->p.clusters_per_tuple ? e->p.clusters_per_tuple : 1u;
        const uint32_t pos_chunk = cm_position((m.trees() - 1u) / 8u * 8u, m.trees(), Cc) / 8u;
        qa.seg_tail_left = e->mc_seg_chunks - pos_chunk;
      }
    }
    a.aux = &qa;
  }
  const bool timing = e->kernel_timing && e->q_slot == 0;
  e->tev_cur = nullptr;
  if (timing) {
    int rc = timing_begin(e, s);
    if (rc) return rc;
    if (v.kind == kKindQ16) a.ev_mid = e->tev_cur[1];
    else HIP_TRY(e, hipEventRecord(e->tev_cur[1], s));
  } else if (e->ev_fork && v.kind == kKindQ16 && !reuse_prepass) {
    a.ev_mid = e->ev_fork;  // multi-class calls: recorded between the shared pre-pass and class 0's scoring kernel
  }
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  hipError_t r = hipSuccess;
  // a small batch: one block per (tile, slice of the image) instead of one per tile, then the adds in the reference's order (launch_cm_combine)
  const SplitPlan sp = v.kind == kKindQ16 ? cluster_split_of(e, v, m, n, reuse_prepass, all_classes) : SplitPlan();
  // ... and the one-launch multi-class kernel ("_p": ONE block per tile walks every class): the plain kernel's cut form over the same image -- the
  // classes stand back to back in it, each cluster-major -- with a partial sum per PU group, the combine per class, then the argmax
  SplitPlan mcp;
  const int ix = all_classes ? find_variant("q16_d8_c8_u4_gl_s2_cm_x") : -1;
  if (ix >= 0 && e->q16_cluster_split != 0 && e->p.sum_mode != 1u && n > 0 && (v.opt & 64) == 0 && variant(ix).has_split()) {
    const uint64_t tiles = (n + 1023) / 1024;
    const uint32_t chunks = e->mc_seg_chunks * e->num_classes;
    // (automatic: up to 160 tiles -- a partial sum per PU group of every class: 10 x 100 trees at 64 tiles 150 against 392 us, at 256 tiles 435 against 417)
    if (e->q16_cluster_split > 0 || tiles <= std::min<uint64_t>(e->q16_split_max_tiles, 160u)) {
      const uint32_t slots = 2u * (e->prop.multiProcessorCount > 0 ? (uint32_t)e->prop.multiProcessorCount : 256u);
      uint32_t want = e->q16_split_groups > 0 ? (uint32_t)e->q16_split_groups : (uint32_t)(slots / tiles);
      want = want > chunks ? chunks : want < 1u ? 1u : want;
      mcp.len = (chunks + want - 1u) / want;
      mcp.split = (chunks + mcp.len - 1u) / mcp.len;
      mcp.partials = chunks + e->num_classes;  // (+ the class sums of a caller that asked for labels only)
      if (!split_fits(mcp.partials, n)) mcp = SplitPlan();
    }
  }
  float* partials = nullptr;
  if (sp.split || mcp.split) {
    int rc = ensure_split_workspace(e, (uint64_t)(sp.split ? sp.partials : mcp.partials) * qa.n_pad, &partials);
    if (rc) return rc;
  }
  const uint32_t clusters = e->p.clusters_per_tuple ? e->p.clusters_per_tuple : 1u;
  if (v.kind == kKindQ16 && !m.parts.empty()) {
    // the ensemble in parts: per part its rank pre-pass (the same workspace, stream order) and a scoring launch over its chunks of the
    // image; the reference-order sum is handed from launch to launch through the state workspace (Q16Aux)
    const size_t chunk_bytes = (size_t)v.tree_bytes_q16() * (size_t)v.chunk_trees;
    float* state = reinterpret_cast<float*>(e->q_state[e->q_slot]);
    uint32_t groups_before = 0;
    const uint32_t walk_all = qa.walk_subgroups;
    // the transposed tuples of this batch (transpose + rank pre-pass) serve every part that needs them -- and every CLASS of a one-vs-all model
    // in parts (the flag used to start at false in every call, so K classes transposed the same batch K times)
    bool xT_valid = reuse_prepass && e->q_xT_valid[e->q_slot];
    for (size_t k = 0; k < m.parts.size() && r == hipSuccess; ++k) {
      const Q16Part& part = m.parts[k];
      a.img = reinterpret_cast<const uint4*>(static_cast<const char*>(m.d_img) + part.chunk_begin * chunk_bytes);
      qa.img_slow = reinterpret_cast<const uint4*>(static_cast<const char*>(m.d_img_slow) + part.chunk_begin * chunk_bytes);
      a.n_chunks = part.chunks;
      a.n_trees = part.chunks * (uint32_t)v.chunk_trees;
      qa.tables = reinterpret_cast<const uint32_t*>(part.rank.d_tables);
      qa.tabP = reinterpret_cast<const uint32_t*>(part.rank.d_tabK);
      qa.tabS = reinterpret_cast<const uint16_t*>(part.rank.d_tabS);
      qa.Kpad = part.rank.Kpad;
      qa.prepass_img = reinterpret_cast<const uint4*>(part.rank.d_prepass);
      qa.prepass = part.rank.prepass;
      qa.skip_prepass = 0u;  // every part ranks the batch against its own tables (also the 2nd..Kth class of a multi-class model)
      qa.skip_transpose = (part.rank.prepass.groups == 0u && xT_valid) ? 1u : 0u;
      xT_valid = xT_valid || part.rank.prepass.groups == 0u;
      qa.group0 = groups_before;
      qa.state_in = k > 0 ? state : nullptr;
      qa.state_out = k + 1 < m.parts.size() ? state : nullptr;
      if (sp.split) {  // cut (deep kernels): every group's sum to its place among the ensemble's, no state between the parts
        const uint32_t here = part.chunks * (uint32_t)v.chunk_trees / 8u, real_here = qa.real_groups > groups_before ? std::min(here, qa.real_groups - groups_before) : 0u;
        qa.state_in = qa.state_out = nullptr;
        qa.split_len = sp.len;
        qa.split = (real_here + sp.len - 1u) / sp.len;
        a.out = partials;
        if (!qa.split) {  // (a part of EMPTY padding only)
          groups_before += here;
          continue;
        }
      }
      const uint32_t sgs_before = part.chunk_begin * (uint32_t)v.chunk_trees / (uint32_t)v.ilp_trees;
      qa.walk_subgroups = (walk_all > sgs_before) ? walk_all - sgs_before : 0u;  // (0 = everything: only the last part has padding)
      if (k + 1 < m.parts.size()) qa.walk_subgroups = 0u;
      if (k > 0) a.ev_mid = nullptr;  // (kernel_timing: the first part's pre-pass against everything behind it)
      r = v.launch(a, v, s);
      groups_before += a.n_trees / 8u;
      e->st.kernel_launches++;
    }
    e->q_xT_valid[e->q_slot] = xT_valid && r == hipSuccess;
    if (r == hipSuccess && sp.split) r = launch_cm_combine(partials, (size_t)qa.n_pad, n, qa.real_groups, clusters, true, true, d_scores, e->p.sum_mode == 2, s);
    else if (r == hipSuccess) e->st.kernel_launches--;  // (counted once more below)
  } else if (mcp.split) {
    const Variant& vx = variant(ix);
    const uint32_t chunks = e->mc_seg_chunks * e->num_classes;
    qa.real_groups = chunks;  // (the kernel's slices end with the image: every class's EMPTY padding groups are walked and never read)
    qa.n_segs = 1;
    qa.seg_chunks = 0;
    qa.labels = nullptr;
    qa.seg_tail_left = 0;
    qa.walk_subgroups = 0;
    qa.split = mcp.split;
    qa.split_len = mcp.len;
    a.out = partials;
    r = vx.launch(a, vx, s);
    float* cs = d_scores ? d_scores : partials + (size_t)chunks * qa.n_pad;
    const size_t cs_pitch = d_scores ? n : (size_t)qa.n_pad;
    if (r == hipSuccess) {
      r = launch_cm_combine(partials, (size_t)qa.n_pad, n, (m.trees() + 7u) / 8u, clusters, true, true, cs, e->p.sum_mode == 2, s, e->num_classes, e->mc_seg_chunks, cs_pitch);
      e->st.kernel_launches++;
    }
    if (r == hipSuccess && labels) {
      r = launch_argmax_strided(cs, e->num_classes, cs_pitch, n, labels, s);
      e->st.kernel_launches++;
    }
  } else if (sp.split) {
    qa.split = sp.split;
    qa.split_len = sp.len;
    a.out = partials;
    r = v.launch(a, v, s);
    if (r == hipSuccess) {
      r = launch_cm_combine(partials, (size_t)qa.n_pad, n, qa.real_groups, clusters, sp.len != 0u, v.cm(), d_scores, e->p.sum_mode == 2, s);
      e->st.kernel_launches++;
    }
  } else {
    r = v.launch(a, v, s);
  }
  if (r != hipSuccess) {
    e->tev_cur = nullptr;
    return fail(e, DDT_EHIP, "kernel launch (%s) -> %s", v.name, hipGetErrorString(r));
  }
  if (timing) {
    int rc = timing_end(e, s);
    if (rc) return rc;
  }
  e->st.kernel_launches++;
  return DDT_OK;
}

// class scores [K][n] into d_class_scores, then argmax into d_labels (if non-NULL)
int launch_classify(ddt_engine* e, const void* d_tuples, size_t n, float* d_class_scores, int32_t* d_labels, hipStream_t s) {
  // "_p" kernels: every class in one pass over the tuples, sums and labels written by the scoring kernel itself
  if (e->mc_seg_chunks) return launch_score(e, e->ens[0], d_tuples, n, d_class_scores, s, false, true, d_labels);
  // Two streams: class 0 (with the shared pre-pass) on the caller's stream, then odd classes on the engine's own stream and
  // even ones on the caller's.  Each class is one launch of n / tile blocks; its last wave of blocks leaves most CUs idle
  // for one block time (5 % of a 100-tree launch over 10 M tuples) -- with a second launch in flight those CUs have work.
  // (never with a "_p" kernel launched once per class: its persistent blocks take their tiles from ONE ticket counter in the batch's
  // workspace, which a second launch in flight would reset and share)
  const bool persistent = variant(e->variant_id).kind == kKindQ16 && (variant(e->variant_id).opt & 8) != 0;
  // ... nor with ensembles scored in parts: every part of every class writes the batch's rank workspace anew
  bool in_parts = false;
  for (const Ensemble& m : e->ens) in_parts = in_parts || !m.parts.empty();
  const bool two = e->class_streams && e->num_classes > 2 && !e->kernel_timing && !persistent && !in_parts;
  if (two && !e->class_stream) {
    HIP_TRY(e, hipStreamCreateWithFlags(&e->class_stream, hipStreamNonBlocking));
    for (hipEvent_t& ev : e->class_ev) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  for (uint32_t k = 0; k < e->num_classes; ++k) {
    // rank-quantised path: the q tiles of this batch are computed by the first class's launch and reused
    hipStream_t sk = (two && (k & 1u)) ? e->class_stream : s;
    // What the other stream needs is the tuples (written by whatever the caller's stream has queued) and, on the rank-quantised path,
    // the shared pre-pass -- not class 0's scoring kernel: the fork event sits between the two (kernels without a pre-pass: in front
    // of class 0's launch), so that class 1 overlaps class 0
    const bool ranked = variant(e->variant_id).kind == kKindQ16;
    if (two && k == 0) {
      if (ranked) e->ev_fork = e->class_ev[0];
      else HIP_TRY(e, hipEventRecord(e->class_ev[0], s));
    }
    int rc = launch_score(e, e->ens[k], d_tuples, n, d_class_scores + (size_t)k * n, sk, k > 0);
    e->ev_fork = nullptr;
    if (rc) return rc;
    if (two && k == 0) HIP_TRY(e, hipStreamWaitEvent(e->class_stream, e->class_ev[0], 0));
  }
  if (two) {
    HIP_TRY(e, hipEventRecord(e->class_ev[1], e->class_stream));
    HIP_TRY(e, hipStreamWaitEvent(s, e->class_ev[1], 0));
  }
  if (d_labels) {
    hipError_t r = launch_argmax(d_class_scores, e->num_classes, n, d_labels, s);
    if (r != hipSuccess) return fail(e, DDT_EHIP, "argmax -> %s", hipGetErrorString(r));
  }
  return DDT_OK;
}

int engine_score_device(ddt_engine* e, const void* d_tuples, size_t n, float* d_scores, hipStream_t s) {
  if (!e->sparse) return launch_score(e, e->ens[0], d_tuples, n, d_scores, s);
  const bool timing = e->kernel_timing && e->q_slot == 0;
  e->tev_cur = nullptr;
  if (timing) {
    int rc = timing_begin(e, s);
    if (rc) return rc;
    if (!(variant(e->variant_id).opt & (1 | 32))) HIP_TRY(e, hipEventRecord(e->tev_cur[1], s));  // fp32 tiles: no pre-pass (else: sparse_launch)
  }
  int rc = sparse_launch(e, 0, d_tuples, n, d_scores, s);
  if (rc) {
    e->tev_cur = nullptr;
    return rc;
  }
  if (timing && (rc = timing_end(e, s))) return rc;
  e->st.kernel_launches++;
  return DDT_OK;
}

int engine_classify_device(ddt_engine* e, const void* d_tuples, size_t n, float* d_class_scores, int32_t* d_labels, hipStream_t s) {
  if (e->sparse) {  // one sparse forest per class (two streams, see launch_classify), then the argmax over the per-class sums
    const bool two = e->class_streams && e->num_classes > 2 && !e->kernel_timing;
    if (two && !e->class_stream) {
      HIP_TRY(e, hipStreamCreateWithFlags(&e->class_stream, hipStreamNonBlocking));
      for (hipEvent_t& ev : e->class_ev) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    for (uint32_t k = 0; k < e->num_classes; ++k) {
      // rank-quantised kernels: the ranks of this batch are computed by the first class's launch and reused
      // the other stream starts behind whatever the caller's stream has queued up to here (the tuples may come from there) and -- the
      // rank-quantised kernels -- behind the rank pre-pass of the batch, which is part of class 0's launch; NOT behind class 0's
      // scoring kernel: the fork event is recorded between the two (sparse_launch), or in front of the launch when there is no pre-pass
      const bool ranked = (variant(e->variant_id).opt & (1 | 32)) != 0;
      if (two && k == 0) {
        if (ranked) e->ev_fork = e->class_ev[0];
        else HIP_TRY(e, hipEventRecord(e->class_ev[0], s));
      }
      int rc = sparse_launch(e, k, d_tuples, n, d_class_scores + (size_t)k * n, (two && (k & 1u)) ? e->class_stream : s, k > 0);
      e->ev_fork = nullptr;
      if (rc) return rc;
      e->st.kernel_launches++;
      if (two && k == 0) HIP_TRY(e, hipStreamWaitEvent(e->class_stream, e->class_ev[0], 0));
    }
    if (two) {
      HIP_TRY(e, hipEventRecord(e->class_ev[1], e->class_stream));
      HIP_TRY(e, hipStreamWaitEvent(s, e->class_ev[1], 0));
    }
    if (d_labels) {
      hipError_t r = launch_argmax(d_class_scores, e->num_classes, n, d_labels, s);
      if (r != hipSuccess) return fail(e, DDT_EHIP, "argmax -> %s", hipGetErrorString(r));
    }
    return DDT_OK;
  }
  return launch_classify(e, d_tuples, n, d_class_scores, d_labels, s);
}

void engine_enter_collective_job(ddt_engine* e) {
  if (!e || e->collective_job) return;
  e->collective_job = true;
  if (!e->loaded || e->sparse || e->forced_variant >= 0 || e->q16_persistent == 0) return;
  // the model that is already loaded: the persistent kernel reads the very image the plain cluster-major kernels read (same chunk
  // size, same order), so the choice can change without re-packing -- unless the ensemble is scored in parts (plain launch only)
  const Variant& cur = variant(e->variant_id);
  const int ip = find_variant("q16_d8_c8_u4_gl_s2_cm_p");
  if (ip < 0 || cur.kind != kKindQ16 || !(cur.opt & 4) || (cur.opt & 8) || e->num_classes != 1 || cur.levels != variant(ip).levels ||
      cur.chunk_trees != variant(ip).chunk_trees || e->ens.empty() || !e->ens[0].parts.empty())
    return;
  if (s2_disabled() || !variant_fits(variant(ip), e)) return;  // the same gates the automatic choice at load time goes through
  e->variant_id = ip;
}

void count_job(ddt_engine* e, size_t n) {
  e->st.score_calls++;
  e->st.tuples_in += n;
  e->st.tuples_out += n;
  e->st.tuple_lines_in += (uint64_t)n * (tuple_words(e->p) / 4);
  e->st.result_lines_out += (n + 3) / 4;
}

// contiguous shard g of `G` of a tree-id list: ceil(|list|/G) trees each (PCIeReceiver.sv:241-264)
std::vector<uint32_t> shard_of(const std::vector<uint32_t>& ids, uint32_t g, uint32_t G) {
  uint32_t b = 0, en = 0;
  (void)ddt_shard_range((uint32_t)ids.size(), g, G, &b, &en);  // g < G checked by the callers
  return std::vector<uint32_t>(ids.begin() + b, ids.begin() + en);
}

int load_common(ddt_engine* e, const ddt_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines,
                uint32_t num_classes, int interleaved, uint32_t shard_index, uint32_t shard_count) {
  if (!e) return DDT_EINVAL;
  if (!wl || !fl) return fail(e, DDT_EINVAL, "NULL model stream");
  int rc = validate(e, p, n_wlines, n_flines);
  if (rc) return rc;
  if (num_classes == 0 || num_classes > p->num_trees) return fail(e, DDT_EINVAL, "num_classes %u (trees %u)", num_classes, p->num_trees);
  if (!interleaved && p->num_trees % num_classes) return fail(e, DDT_EINVAL, "class-major layout needs num_trees %% num_classes == 0");
  const uint32_t per_class = (p->num_trees + num_classes - 1) / num_classes;
  if (shard_count == 0 || shard_index >= shard_count || shard_count > per_class)
    return fail(e, DDT_EINVAL, "shard %u of %u (trees per class %u)", shard_index, shard_count, per_class);
  const double t0 = now_ms();
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  std::vector<Ensemble> ens(num_classes);
  uint64_t lines = 0;
  for (uint32_t k = 0; k < num_classes; ++k) {
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < p->num_trees; ++i) {
      const uint32_t cls = interleaved ? i % num_classes : i / (p->num_trees / num_classes);
      if (cls == k) ids.push_back(i);
    }
    std::vector<uint32_t> mine = shard_of(ids, shard_index, shard_count);
    // an EMPTY shard is legal: ceil(T/G) trees per device leaves the trailing devices without trees when
    // (G-1)*ceil(T/G) >= T (e.g. T = 9, G = 4); such a device holds only EMPTY slots (DTPU.sv:544,760) and returns +0,
    // so every rank of a sharded job behaves the same and nobody is left waiting in a collective
    lines += (uint64_t)mine.size() * (p->weights_lines_per_tree + p->findex_lines_per_tree);
    rc = parse_trees(e, p, reinterpret_cast<const uint32_t*>(wl), reinterpret_cast<const uint16_t*>(fl), std::move(mine), &ens[k]);
    if (rc) return rc;
  }
  HIP_TRY(e, hipDeviceSynchronize());  // asynchronous scoring of the previous model may still be in flight
  free_images(e);
  free_q16_workspace(e);  // sized for the previous model's tuple width
  sparse_free(e);
  e->sps.clear();
  e->sparse = false;
  e->perfect_as_sparse = false;
  e->loaded = false;
  e->p = *p;
  e->nint = (1u << p->num_levels) - 1u;
  e->nleaf = 1u << p->num_levels;
  e->num_classes = num_classes;
  e->ens = std::move(ens);
  rc = select_and_build(e);
  if (rc) return rc;
  rc = maybe_score_as_sparse(e);
  if (rc) return rc;
  e->loaded = true;
  e->st.model_lines_in += lines;
  e->st.prog_ms += now_ms() - t0;
  return DDT_OK;
}

}  // namespace

// =====================================================================================================
// C-ABI
// =====================================================================================================
extern "C" {

int ddt_create(ddt_engine** out, int device_id) {
  if (!out) return DDT_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return DDT_ENODEVICE;  // never a CPU fallback
  if (device_id < 0 || device_id >= count) return DDT_EINVAL;
  std::unique_ptr<ddt_engine> e(new (std::nothrow) ddt_engine());
  if (!e) return DDT_ENOMEM;
  e->device = device_id;
  DeviceGuard dg(device_id);
  if (!dg.ok) return DDT_EHIP;
  if (hipGetDeviceProperties(&e->prop, device_id) != hipSuccess) return DDT_EHIP;
  if (strncmp(e->prop.gcnArchName, "gfx950", 6) != 0) return DDT_ENODEVICE;  // kernels are built for gfx950 only
  int khz = 0;  // the constant clock behind s_memrealtime (100 MHz on MI355X): the stream kernel's write windows are timed by it
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_id) == hipSuccess && khz > 0) e->wall_clock_khz = khz;
  else (void)hipGetLastError();
  *out = e.release();
  return DDT_OK;
}

void ddt_destroy(ddt_engine* e) {
  if (!e) return;
  DeviceGuard dg(e->device);
  (void)hipDeviceSynchronize();  // asynchronous ddt_*_device work may still read the images / workspaces
  feeder_free(e);
  for (int b = 0; b < kFeederSlots; ++b) {
    if (e->fs[b]) (void)hipStreamDestroy(e->fs[b]);
    if (e->fe[b]) (void)hipEventDestroy(e->fe[b]);
    if (e->fe_in[b]) (void)hipEventDestroy(e->fe_in[b]);
  }
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  delete e->pool;
  for (const auto& r : e->pinned) (void)hipHostUnregister(r.first);  // ranges the caller forgot to hand back
  if (e->ws) (void)hipFree(e->ws);
  for (auto& t : e->tev)
    for (hipEvent_t ev : t)
      if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->class_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (e->class_stream) (void)hipStreamDestroy(e->class_stream);
  free_images(e);
  free_q16_workspace(e);
  sparse_free(e);
  delete e;
}

int ddt_load_model_shard(ddt_engine* e, const ddt_params* p, const void* wl, size_t n_wlines, const void* fl,
                         size_t n_flines, uint32_t shard_index, uint32_t shard_count) {
  return load_common(e, p, wl, n_wlines, fl, n_flines, 1, 0, shard_index, shard_count);
}

int ddt_load_model(ddt_engine* e, const ddt_params* p, const void* wl, size_t n_wlines, const void* fl,
                   size_t n_flines) {
  return load_common(e, p, wl, n_wlines, fl, n_flines, 1, 0, 0, 1);
}

int ddt_load_model_multiclass(ddt_engine* e, const ddt_params* p, const void* wl, size_t n_wlines, const void* fl,
                              size_t n_flines, uint32_t num_classes, int interleaved, uint32_t shard_index,
                              uint32_t shard_count) {
  return load_common(e, p, wl, n_wlines, fl, n_flines, num_classes, interleaved, shard_index, shard_count);
}

int ddt_score_device(ddt_engine* e, const void* d_tuples, size_t n, float* d_scores, void* stream) {
  if (!e) return DDT_EINVAL;
  if (!e->loaded) return fail(e, DDT_ESTATE, "no model loaded");
  if (e->num_classes != 1) return fail(e, DDT_ESTATE, "multi-class model loaded: use ddt_classify*");
  if (n == 0) return DDT_OK;
  if (!d_tuples || !d_scores) return fail(e, DDT_EINVAL, "NULL device buffer");
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  int rc = engine_score_device(e, d_tuples, n, d_scores, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  count_job(e, n);
  return DDT_OK;
}

int ddt_classify_device(ddt_engine* e, const void* d_tuples, size_t n, float* d_class_scores, int32_t* d_labels,
                        void* stream) {
  if (!e) return DDT_EINVAL;
  if (!e->loaded) return fail(e, DDT_ESTATE, "no model loaded");
  if (n == 0) return DDT_OK;
  if (!d_tuples || !d_class_scores) return fail(e, DDT_EINVAL, "NULL device buffer");
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  int rc = engine_classify_device(e, d_tuples, n, d_class_scores, d_labels, reinterpret_cast<hipStream_t>(stream));
  if (rc) return rc;
  count_job(e, n);
  return DDT_OK;
}

int ddt_argmax_device(ddt_engine* e, const float* d_class_scores, uint32_t K, size_t n, int32_t* d_labels, void* stream) {
  if (!e) return DDT_EINVAL;
  if (K == 0 || (n && (!d_class_scores || !d_labels))) return fail(e, DDT_EINVAL, "bad argmax arguments");
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  hipError_t r = launch_argmax(d_class_scores, K, n, d_labels, reinterpret_cast<hipStream_t>(stream));
  if (r != hipSuccess) return fail(e, DDT_EHIP, "argmax -> %s", hipGetErrorString(r));
  return DDT_OK;
}

// Shared host-buffer path: pinned double buffer; while chunk i computes on stream i&1, chunk i+1 is copied in on
// the other.  classify: per chunk the device output is [K][cn] class scores followed by cn int32 labels.
// staging copy host -> pinned buffer on several threads: a single thread moves ~26 GB/s, less than the
// ~55 GB/s the PCIe Gen5 x16 link takes (the feeder would then be bound by memcpy, not by the link)
static void parallel_copy(ddt_engine* e, void* dst, const void* src, size_t bytes) {
  if (e->feeder_threads <= 1 || bytes < (4u << 20)) {
    memcpy(dst, src, bytes);
    return;
  }
  if (e->pool && (int)e->pool->workers.size() != e->feeder_threads - 1) {
    delete e->pool;
    e->pool = nullptr;
  }
  if (!e->pool) {
    try {
      e->pool = new ddt_copy_pool(e->feeder_threads - 1);
    } catch (...) {  // no threads / no memory for the pool (std::system_error, std::bad_alloc): copy on the calling thread
      e->pool = nullptr;
      memcpy(dst, src, bytes);
      return;
    }
  }
  e->pool->copy(dst, src, bytes);
}

// is [p, p + bytes) inside a range the caller pinned with ddt_host_register?  (then the DMA engine reads / writes it directly)
static bool host_registered(const ddt_engine* e, const void* p, size_t bytes) {
  const char* q = static_cast<const char*>(p);
  for (const auto& r : e->pinned)
    if (q >= static_cast<const char*>(r.first) && q + bytes <= static_cast<const char*>(r.first) + r.second) return true;
  return false;
}

static int score_host(ddt_engine* e, const void* tuple_lines, size_t n, float* scores_out, int32_t* labels_out,
                      float* class_scores_out) {
  const bool classify = labels_out != nullptr;
  const uint32_t K = classify ? e->num_classes : 1u;
  const double t0 = now_ms();
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  const size_t W = tuple_words(e->p);
  // rows per chunk: the option, but never more than kFeederMaxChunkBytes of tuples per slot (a 2048-feature model at the default 2^20
  // rows would pin 3 x 8 GiB of host memory and as much of the device's)
  size_t rows = e->feeder_rows < n ? e->feeder_rows : n;
  const size_t by_bytes = std::max<size_t>(kFeederMaxChunkBytes / (W * 4u) / 1024u * 1024u, 1024u);
  if (rows > by_bytes) rows = by_bytes;
  const size_t outs = classify ? K + 1 : 1;
  int rc = feeder_reserve(e, rows, W, outs);
  if (rc) return rc;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(tuple_lines);
  // buffers the caller pinned (ddt_host_register) go through the DMA engine directly: no staging copy in, no drain copy out
  const bool in_direct = host_registered(e, tuple_lines, n * W * 4);
  const bool out_direct = !classify && host_registered(e, scores_out, n * 4);
  size_t pending_off[kFeederSlots] = {}, pending_n[kFeederSlots] = {};
  auto drain = [&](int b) -> int {
    HIP_TRY(e, hipEventSynchronize(e->fe[b]));
    const size_t cn = pending_n[b], off = pending_off[b];
    const float* po = reinterpret_cast<const float*>(e->pin_out[b]);
    if (out_direct) {
      // the scores went straight into the caller's buffer
    } else if (!classify) {
      memcpy(scores_out + off, po, cn * 4);
    } else {
      if (class_scores_out)
        for (uint32_t k = 0; k < K; ++k) memcpy(class_scores_out + (size_t)k * n + off, po + (size_t)k * cn, cn * 4);
      memcpy(labels_out + off, po + (size_t)K * cn, cn * 4);
    }
    pending_n[b] = 0;
    return DDT_OK;
  };
  // Three slots, three streams: while chunk k computes, chunk k+1 crosses the link and chunk k+2 is being staged by the host
  // threads (with two slots the staging of k+2 could only begin once chunk k had left its slot).
  for (size_t off = 0, i = 0; off < n; off += rows, ++i) {
    const int b = (int)(i % kFeederSlots);
    const size_t cn = (n - off < rows) ? n - off : rows;
    if (pending_n[b] && (rc = drain(b))) return rc;
    // ALL host-to-device copies go down ONE stream, in order: copies issued on three streams at once share the copy engines and
    // the link badly (tools/h2d_probe.py on the GPU box: 1 GiB in 128 MiB pieces -- one stream 56 GB/s, three streams 40 GB/s);
    // the slot's own stream (kernels, scores back) waits for its chunk's event
    if (!in_direct) parallel_copy(e, e->pin_in[b], src + off * W, cn * W * 4);
    HIP_TRY(e, hipMemcpyAsync(e->dev_in[b], in_direct ? static_cast<const void*>(src + off * W) : e->pin_in[b], cn * W * 4, hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(e, hipEventRecord(e->fe_in[b], e->copy_stream));
    HIP_TRY(e, hipStreamWaitEvent(e->fs[b], e->fe_in[b], 0));
    float* dout = reinterpret_cast<float*>(e->dev_out[b]);
    e->q_slot = 1 + b;  // the feeder streams run concurrently: separate q16 workspaces
    if (!classify) rc = engine_score_device(e, e->dev_in[b], cn, dout, e->fs[b]);
    else rc = engine_classify_device(e, e->dev_in[b], cn, dout, reinterpret_cast<int32_t*>(dout + (size_t)K * cn), e->fs[b]);
    e->q_slot = 0;
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(out_direct ? static_cast<void*>(scores_out + off) : e->pin_out[b], e->dev_out[b], cn * outs * 4, hipMemcpyDeviceToHost, e->fs[b]));
    HIP_TRY(e, hipEventRecord(e->fe[b], e->fs[b]));
    pending_off[b] = off;
    pending_n[b] = cn;
  }
  for (int b = 0; b < kFeederSlots; ++b)
    if (pending_n[b] && (rc = drain(b))) return rc;
  count_job(e, n);
  e->st.exec_ms += now_ms() - t0;
  return DDT_OK;
}

int ddt_host_register(ddt_engine* e, void* ptr, size_t bytes) {
  if (!e) return DDT_EINVAL;
  if (!ptr || !bytes) return fail(e, DDT_EINVAL, "ddt_host_register: NULL / empty range");
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  for (const auto& r : e->pinned)
    if (r.first == ptr) return fail(e, DDT_EINVAL, "ddt_host_register: %p is registered already", ptr);
  const hipError_t r = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    return fail(e, DDT_EHIP, "hipHostRegister(%p, %zu) -> %s", ptr, bytes, hipGetErrorString(r));
  }
  try {
    e->pinned.emplace_back(ptr, bytes);
  } catch (const std::bad_alloc&) {
    (void)hipHostUnregister(ptr);
    return fail(e, DDT_ENOMEM, "ddt_host_register");
  }
  return DDT_OK;
}

int ddt_host_unregister(ddt_engine* e, void* ptr) {
  if (!e) return DDT_EINVAL;
  DeviceGuard dg(e->device);
  for (size_t i = 0; i < e->pinned.size(); ++i)
    if (e->pinned[i].first == ptr) {
      (void)hipDeviceSynchronize();
      (void)hipHostUnregister(ptr);
      e->pinned.erase(e->pinned.begin() + (long)i);
      return DDT_OK;
    }
  return fail(e, DDT_EINVAL, "ddt_host_unregister: %p was not registered through this engine", ptr);
}

int ddt_score(ddt_engine* e, const void* tuple_lines, size_t n, float* scores_out) {
  if (!e) return DDT_EINVAL;
  if (!e->loaded) return fail(e, DDT_ESTATE, "no model loaded");
  if (e->num_classes != 1) return fail(e, DDT_ESTATE, "multi-class model loaded: use ddt_classify*");
  if (n == 0) return DDT_OK;
  if (!tuple_lines || !scores_out) return fail(e, DDT_EINVAL, "NULL host buffer");
  return score_host(e, tuple_lines, n, scores_out, nullptr, nullptr);
}

int ddt_classify(ddt_engine* e, const void* tuple_lines, size_t n, int32_t* labels, float* class_scores) {
  if (!e) return DDT_EINVAL;
  if (!e->loaded) return fail(e, DDT_ESTATE, "no model loaded");
  if (n == 0) return DDT_OK;
  if (!tuple_lines || !labels) return fail(e, DDT_EINVAL, "NULL host buffer");
  return score_host(e, tuple_lines, n, nullptr, labels, class_scores);
}

int ddt_chain_sum_device(ddt_engine* e, const float* d_parts, uint32_t n_parts, size_t n, float* d_out, void* stream) {
  if (!e) return DDT_EINVAL;
  if (n_parts == 0 || (!d_parts && n) || (!d_out && n)) return fail(e, DDT_EINVAL, "bad chain-sum arguments");
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  hipError_t r = launch_chain_sum(d_parts, n_parts, n, d_out, e->loaded && e->p.sum_mode == 2, reinterpret_cast<hipStream_t>(stream));
  if (r != hipSuccess) return fail(e, DDT_EHIP, "chain_sum -> %s", hipGetErrorString(r));
  return DDT_OK;
}

int ddt_get_info(const ddt_engine* e, ddt_info* out) {
  if (!e || !out) return DDT_EINVAL;
  memset(out, 0, sizeof(*out));
  out->abi_version = DDT_ABI_VERSION;
  out->device_id = e->device;
  snprintf(out->device_name, sizeof(out->device_name), "%s (%s)", e->prop.name, e->prop.gcnArchName);
  out->num_cus = (uint32_t)e->prop.multiProcessorCount;
  out->clock_khz = (uint32_t)e->prop.clockRate;
  out->lds_bytes_per_cu = (uint32_t)e->prop.maxSharedMemoryPerMultiProcessor;
  if (!e->loaded) return DDT_OK;
  const Variant& v = variant(e->variant_id);
  if (!e->sparse && v.kind == kKindQ16 && !e->ens.empty()) out->prepass_groups = e->ens[0].prepass.groups;
  out->fallback_kernel = (v.kind == kKindGeneric || (v.kind == kKindSparse && (v.opt & 4))) ? 1u : 0u;
  out->build_checks = (ddt_build_s2_checked ? 1u : 0u) | (ddt_build_dma_checked ? 2u : 0u);
  if (e->sparse) {
    uint32_t trees = 0, depth = 0;
    uint64_t lines = 0, img = 0;
    for (const SparseForest& sp : e->sps) {
      trees += sp.trees();
      depth = sp.max_depth > depth ? sp.max_depth : depth;
      lines += sp.lines.size() / 4u;
      img += sp.top_bytes + sp.deep_bytes;
    }
    const SparseForest& s0 = e->sps.front();
    const SparseForest& s1 = e->sps.back();
    out->tree_begin = s0.ids.empty() ? 0u : s0.ids.front();
    out->tree_end = s1.ids.empty() ? out->tree_begin : s1.ids.back() + 1;
    out->num_levels = depth;
    out->num_features = e->p.num_features;
    out->tuple_words = tuple_words(e->p);
    out->variant = (uint32_t)e->variant_id;
    out->tile_tuples = v.tile();
    out->block_threads = (uint32_t)v.threads;
    out->lds_bytes = v.lds_bytes_sparse(out->tuple_words);
    out->model_bytes_unpadded = lines * 16ull;  // 16 bytes per internal node: the stream itself
    out->image_bytes = img;
    out->num_classes = e->num_classes;
    out->local_trees = trees;
    snprintf(out->variant_name, sizeof(out->variant_name), "%s", v.name);
    return DDT_OK;
  }
  const Ensemble& m0 = e->ens[0];
  uint32_t trees = 0;
  uint64_t img = 0;
  for (const Ensemble& m : e->ens) {
    trees += m.trees();
    img += m.img_bytes;
  }
  out->tree_begin = m0.ids.empty() ? 0u : m0.ids.front();
  out->tree_end = e->ens.back().ids.empty() ? out->tree_begin : e->ens.back().ids.back() + 1;
  out->num_levels = e->p.num_levels;
  out->num_features = e->p.num_features;
  out->tuple_words = tuple_words(e->p);
  out->variant = (uint32_t)e->variant_id;
  out->tile_tuples = v.tile();
  out->block_threads = (uint32_t)v.threads;
  out->lds_bytes = v.kind == kKindTile     ? v.lds_bytes(out->tuple_words)
                   : v.kind == kKindStream ? v.lds_bytes_stream(m0.img_trees, out->tuple_words)
                   : v.kind == kKindQ16    ? v.lds_bytes_q16(out->tuple_words)
                                           : generic_lds_bytes(e->p.num_levels, out->tuple_words, nullptr, nullptr, nullptr);
  out->model_bytes_unpadded = (uint64_t)trees * (4ull * ((2ull << e->p.num_levels) - 1) + 2ull * ((1ull << e->p.num_levels) - 1));
  out->image_bytes = img;
  out->num_classes = e->num_classes;
  out->local_trees = trees;
  snprintf(out->variant_name, sizeof(out->variant_name), "%s", v.name);
  return DDT_OK;
}

int ddt_get_stats(const ddt_engine* e_, ddt_stats* out) {
  if (!e_ || !out) return DDT_EINVAL;
  ddt_engine* e = const_cast<ddt_engine*>(e_);  // resolving pending event times is a logically-const refresh
  timing_resolve(e, 0);
  *out = e->st;
  return DDT_OK;
}

const char* ddt_strerror(int code) {
  switch (code) {
    case DDT_OK: return "ok";
    case DDT_EINVAL: return "invalid argument";
    case DDT_ENOMEM: return "out of memory";
    case DDT_EHIP: return "HIP runtime error";
    case DDT_ESTATE: return "bad call order (no model loaded?)";
    case DDT_EUNSUPPORTED: return "not supported";
    case DDT_ENODEVICE: return "no usable gfx950 device (this library has no CPU fallback)";
    default: return "unknown error";
  }
}

const char* ddt_last_error(const ddt_engine* e) { return e ? e->err : ""; }

int ddt_set_option(ddt_engine* e, const char* key, int64_t value) {
  if (!e || !key) return DDT_EINVAL;
  if (!strcmp(key, "variant")) {
    if (value >= num_variants()) return fail(e, DDT_EINVAL, "variant %lld out of range", (long long)value);
    const int before = e->forced_variant;
    e->forced_variant = value < 0 ? -1 : (int)value;
    if (e->loaded) {  // re-pack the loaded model for the forced kernel (or back to the automatic choice)
      DeviceGuard dg(e->device);
      if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
      HIP_TRY(e, hipDeviceSynchronize());
      e->loaded = false;
      if (e->perfect_as_sparse) {  // a perfect-tree model that was handed to the sparse path: the choice starts over from its perfect form
        sparse_free(e);
        e->sps.clear();
        e->sparse = false;
        e->perfect_as_sparse = false;
      }
      auto rebuild = [&]() -> int {
        if (e->sparse) return sparse_rebuild(e);
        const int r1 = select_and_build(e);
        return r1 ? r1 : maybe_score_as_sparse(e);
      };
      int rc = rebuild();
      if (rc) {  // e.g. the variant does not fit this model: the setting is not taken and the model stays loaded as it was
        char why[sizeof(e->err)];
        snprintf(why, sizeof(why), "%s", e->err);
        e->forced_variant = before;
        if (rebuild() == DDT_OK) e->loaded = true;
        snprintf(e->err, sizeof(e->err), "%s", why);
        return rc;
      }
      e->loaded = true;
    }
    return DDT_OK;
  }
  if (!strcmp(key, "sparse_dp")) {  // dense pair records (ddt_sparse_host.cpp sparse_rebuild): -1 automatic, 0 never, 1 always; effective at the next sparse load
    if (value < -1 || value > 1) return fail(e, DDT_EINVAL, "sparse_dp must be -1, 0 or 1");
    e->sparse_dp = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "sparse_dm")) {  // dense mid levels (ddt_sparse_host.cpp sparse_rebuild): -1 automatic, 0 never, 1..3 exactly; effective at the next sparse load
    if (value < -1 || value > 3) return fail(e, DDT_EINVAL, "sparse_dm must be -1..3");  // (-1: one mid level where the forest fills it)
    e->sparse_dm = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "sparse_top_levels") || !strcmp(key, "sparse_deep_order") || !strcmp(key, "sparse_q16") || !strcmp(key, "sparse_dk") || !strcmp(key, "sparse_r32")) {
    // sparse forests: K = levels staged in LDS (-1 = as many as fit), order of the deep records (0 level order,
    // 1 depth-first per sub-tree), rank-quantised kernels (1 = when they fit, 0 = never), dense level K (1 = where such a kernel
    // exists, 0 = never); a loaded sparse model is re-packed
    // "sparse_r32": 32-bit ranks + pair records on every deep level (-1 = automatic: deep forests of >= 64 trees, 0 = never, 1 = wherever such a kernel fits)
    const bool top = key[7] == 't', rq = key[7] == 'q', dk = !strcmp(key, "sparse_dk"), r32 = !strcmp(key, "sparse_r32");
    if (top && value >= 0 && (value < kSparseMinTop || value > kSparseMaxTop)) return fail(e, DDT_EINVAL, "sparse_top_levels must be -1 or %d..%d", kSparseMinTop, kSparseMaxTop);
    if (r32 && (value < -1 || value > 1)) return fail(e, DDT_EINVAL, "sparse_r32 must be -1, 0 or 1");
    if (!top && !r32 && (value < 0 || value > 1)) return fail(e, DDT_EINVAL, "%s must be 0 or 1", key);
    int& opt = top ? e->sparse_top_levels : rq ? e->sparse_q16 : dk ? e->sparse_dk : r32 ? e->sparse_r32 : e->sparse_deep_order;
    const int previous = opt;
    opt = (int)value;
    if (e->loaded && e->sparse) {
      DeviceGuard dg(e->device);
      if (!dg.ok) {
        opt = previous;
        return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
      }
      hipError_t hr = hipDeviceSynchronize();
      if (hr != hipSuccess) {
        opt = previous;
        return fail(e, DDT_EHIP, "hipDeviceSynchronize -> %s", hipGetErrorString(hr));
      }
      e->loaded = false;
      int rc = sparse_rebuild(e);
      if (rc) {
        // a refused change (no kernel fits that K for this tuple width, out of memory) keeps the previous setting AND the loaded
        // model, like a refused "variant": re-pack with the old value, report the original error
        char why[sizeof(e->err)];
        memcpy(why, e->err, sizeof(why));
        opt = previous;
        const int rc2 = sparse_rebuild(e);
        e->loaded = rc2 == 0;
        memcpy(e->err, why, sizeof(why));
        return rc;
      }
      e->loaded = true;
    }
    return DDT_OK;
  }
  if (!strcmp(key, "sparse_peel_last")) {  // A/B: 0 = every round of the sparse kernels' deep loop issues its gathers, the one after the deepest level too (as before round 5); effective at the next call
    e->sparse_peel_last = value != 0;
    return DDT_OK;
  }
  if (!strcmp(key, "sparse_idle_oob")) {  // A/B: 0 = finished walkers of the sparse kernels re-read record 0 (as before round 5); effective at the next call
    e->sparse_idle_oob = value != 0;
    return DDT_OK;
  }
  if (!strcmp(key, "generic_via_sparse")) {  // 1 (default): a perfect-tree model without a tuned kernel is scored by the sparse-forest kernels (maybe_score_as_sparse);
    e->generic_via_sparse = value != 0;       // 0: it stays on `generic` (A/B, tests).  Effective at the next model load
    return DDT_OK;
  }
  if (!strcmp(key, "feature_compaction")) {  // 1 (default): a model of more than 64 tuple words that tests at most 64 features runs on the rank-quantised kernels over
    e->feature_compaction = value != 0;       // the compacted columns; 0: never (A/B, tests).  Effective at the next model load
    return DDT_OK;
  }
  if (!strcmp(key, "leaf_domain_check")) {  // 1 (default): refuse -0 / sub-normal / Inf / NaN leaves in the reference-order sum
    e->leaf_domain_check = value != 0;
    return DDT_OK;
  }
  if (!strcmp(key, "reserve_rows")) {
    // pre-size the rank-quantised path's workspace (ranks, per-tile flags, transposed tuples) for calls of up to `value`
    // rows, so that the asynchronous ddt_*_device calls never have to synchronise and allocate on first use / growth
    if (value < 0) return fail(e, DDT_EINVAL, "reserve_rows must be >= 0");
    if (!e->loaded) return fail(e, DDT_ESTATE, "reserve_rows: load a model first (the workspace depends on its tuple width)");
    const Variant& cur = variant(e->variant_id);
    const bool ranked = e->sparse ? (cur.opt & (1 | 32)) != 0 : cur.kind == kKindQ16;
    if (!ranked || value == 0) return DDT_OK;  // nothing to reserve on the other paths
    DeviceGuard dg(e->device);
    if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
    return ensure_q16_workspace(e, (size_t)value);
  }
  if (!strcmp(key, "stream_blocks_per_cu")) {  // persistent stream kernel: blocks per CU, 0 (default) = the resident number
    if (value < 0 || value > 16) return fail(e, DDT_EINVAL, "stream_blocks_per_cu %lld not in 0..16", (long long)value);
    e->stream_blocks_per_cu = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "stream_res_tiles")) {  // stream kernel, phased result stores: 0 (default) = as many LDS slots as fit, 1 = direct stores, n = at most n
    if (value < 0 || value > 64) return fail(e, DDT_EINVAL, "stream_res_tiles %lld not in 0..64", (long long)value);
    e->stream_res_tiles = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "stream_window_ticks")) {  // ... the write window's period in 10 ns ticks of the constant 100 MHz clock; 0 = default (3000)
    if (value < 0 || (value != 0 && value < 100) || value > 10000000) return fail(e, DDT_EINVAL, "stream_window_ticks %lld not 0 or in 100..10000000", (long long)value);
    e->stream_window_ticks = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_walk_padding")) {  // A/B: 1 = the plain rank-quantised kernels walk the EMPTY padding trees of the last chunk too (as before round 4)
    e->q16_walk_padding = value != 0;
    return DDT_OK;
  }
  if (!strcmp(key, "class_streams")) {  // 1 (default): the classes of a multi-class model alternate between two streams; 0: one stream
    e->class_streams = value != 0;
    return DDT_OK;
  }
  if (!strcmp(key, "kernel_timing")) {
    e->kernel_timing = value != 0;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_grouped_prepass")) {  // 0: no pre-pass split over feature groups (G > 1); with q16_fused_prepass 0 too: transpose + rank kernels
    e->q16_grouped_prepass = value != 0;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_prepass_groups")) {  // 0 (default): the smallest number of feature groups that fits; 1, 2, 4, 8: exactly that (A/B)
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(e, DDT_EINVAL, "q16_prepass_groups must be 0, 1, 2, 4 or 8");
    e->q16_prepass_groups = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_max_table")) {  // distinct thresholds per feature in ONE rank table: an ensemble beyond it is scored in parts (next model load; A/B, tests)
    if (value < 255 || value > (int64_t)kQ16MaxTable) return fail(e, DDT_EINVAL, "q16_max_table must be in 255..%u", kQ16MaxTable);
    e->q16_max_table = (uint32_t)value;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_fused_prepass")) {  // 0: never the single-group form (all tables resident together); both 0: transpose + rank kernels.
                                            // These three take effect at the next model load (A/B and tests); defaults 1, 1, 0
    e->q16_fused_prepass = value != 0;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_prepass_nt")) {  // A/B: bit 0 = nontemporal stores of the rank tiles, bit 1 = nontemporal tuple loads (effective at the next call)
    if (value < 0 || value > 3) return fail(e, DDT_EINVAL, "q16_prepass_nt must be 0..3");
    e->q16_prepass_nt = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_cluster_split")) {  // 1 / 0: always / never cut a launch of the plain depth-8 cluster-major kernel at the clusters; -1: automatic (small batches)
    if (value < -1 || value > 1) return fail(e, DDT_EINVAL, "q16_cluster_split must be -1, 0 or 1");
    e->q16_cluster_split = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_split_groups")) {  // -1: automatic (runs of PU groups where the clusters alone leave CUs idle); 0: clusters only; > 0: that many slices (A/B, tests)
    if (value < -1 || value > 65535) return fail(e, DDT_EINVAL, "q16_split_groups out of range");
    e->q16_split_groups = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "sparse_split_max_tiles")) {  // automatic cut of the "sparse_r_*" launch: batches of up to this many tiles of the kernel's own size
    if (value < 0 || value > 0x7FFFFFFF) return fail(e, DDT_EINVAL, "sparse_split_max_tiles out of range");
    e->sparse_split_max_tiles = (uint32_t)value;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_split_max_tiles")) {  // automatic cluster split: batches of up to this many tiles of 1024 tuples
    if (value < 0 || value > 0x7FFFFFFF) return fail(e, DDT_EINVAL, "q16_split_max_tiles out of range");
    e->q16_split_max_tiles = (uint32_t)value;
    return DDT_OK;
  }
  if (!strcmp(key, "q16_persistent")) {  // 1 / 0: prefer / never pick the persistent "_p" rank-quantised kernel; -1: automatic.  Effective at the next model load
    if (value < -1 || value > 1) return fail(e, DDT_EINVAL, "q16_persistent must be -1, 0 or 1");
    e->q16_persistent = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "feeder_threads")) {
    if (value < 1 || value > 64) return fail(e, DDT_EINVAL, "feeder_threads must be 1..64");
    e->feeder_threads = (int)value;
    return DDT_OK;
  }
  if (!strcmp(key, "feeder_rows")) {
    if (value < 1) return fail(e, DDT_EINVAL, "feeder_rows must be >= 1");
    e->feeder_rows = (size_t)value;
    return DDT_OK;
  }
  return fail(e, DDT_EINVAL, "unknown option '%s'", key);
}

int ddt_shard_range(uint32_t num_trees, uint32_t shard_index, uint32_t shard_count, uint32_t* tree_begin, uint32_t* tree_end) {
  if (!tree_begin || !tree_end || shard_count == 0 || shard_index >= shard_count) return DDT_EINVAL;
  const uint64_t per = ((uint64_t)num_trees + shard_count - 1) / shard_count;
  const uint64_t b = (uint64_t)shard_index * per < num_trees ? (uint64_t)shard_index * per : num_trees;
  *tree_begin = (uint32_t)b;
  *tree_end = (uint32_t)(b + per < num_trees ? b + per : num_trees);
  return DDT_OK;
}

int ddt_num_variants(void) { return num_variants(); }

int ddt_variant_name(int v, char* buf, size_t buflen) {
  if (v < 0 || v >= num_variants() || !buf || !buflen) return DDT_EINVAL;
  snprintf(buf, buflen, "%s", variant(v).name);
  return DDT_OK;
}

// ---- synthetic inputs --------------------------------------------------------------
static inline float unit24(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }
static inline uint32_t fbits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }


int ddt_synth_model(uint32_t T, uint32_t D, uint32_t F, int dist, void* wlines, void* flines) {
  if (!wlines || !flines || T == 0 || D < 1 || D > 16 || F < 1 || F > 2048) return DDT_EINVAL;
  uint32_t* w = reinterpret_cast<uint32_t*>(wlines);
  uint16_t* f = reinterpret_cast<uint16_t*>(flines);
  const uint32_t nint = (1u << D) - 1u, ntot = (2u << D) - 1u;
  const size_t ws = (size_t)wlines_min(D) * 4u, fs = (size_t)flines_min(D) * 8u;
  memset(w, 0, (size_t)T * ws * 4);
  memset(f, 0, (size_t)T * fs * 2);
  for (uint32_t i = 0; i < T; ++i)
    for (uint32_t n = 0; n < ntot; ++n) {
      const uint64_t g = (uint64_t)i * (2ull << D) + n;
      const float u = unit24(splitmix64(kSeedM + 3ull * g + 1ull));
      if (n < nint) {
        const uint32_t j = (uint32_t)(splitmix64(kSeedM + 3ull * g) % F);
        const uint32_t mr = (uint32_t)(splitmix64(kSeedM + 3ull * g + 2ull) & 1ull);
        w[(size_t)i * ws + n] = fbits(dist == 1 ? u * 2.0f - 1.0f : u);
        f[(size_t)i * fs + n] = (uint16_t)(j | (mr << 13));
      } else {
        volatile float c = u - 0.5f;
        volatile float v = c * 0.2f;
        w[(size_t)i * ws + n] = fbits(v);
      }
    }
  return DDT_OK;
}

int64_t ddt_synth_sparse_model(uint32_t T, uint32_t max_depth, uint32_t F, uint32_t full_levels, uint32_t split_permille, int dist,
                               void* node_lines, size_t cap_lines, uint64_t* first) {
  if (T == 0 || max_depth < 1 || max_depth > 64 || F < 1 || F > 2048 || split_permille > 1000) return DDT_EINVAL;
  // breadth-first growth; the n-th internal node of tree i (BFS order) draws from hash base g = i << 24 | n:
  // feature h(8g) % F, threshold unit(h(8g+1)), missing direction h(8g+2) & 1, child `side` internal iff its depth is
  // < max_depth and (< full_levels or (h(8g+4+side) >> 20) % 1000 < split_permille), else a leaf (unit(h(8g+6+side)) - 0.5) * 0.2
  constexpr uint32_t kCap = 4u << 20;  // internal nodes per tree
  uint32_t* lines = reinterpret_cast<uint32_t*>(node_lines);
  std::vector<uint8_t> depth;
  size_t total = 0;
  for (uint32_t i = 0; i < T; ++i) {
    if (first) first[i] = total;
    depth.assign(1, 0);
    for (uint32_t n = 0; n < depth.size(); ++n) {
      const uint64_t g = ((uint64_t)i << 24) | n;
      const uint32_t d = depth[n];
      uint32_t en = (uint32_t)(splitmix64(kSeedS + 8ull * g) % F) | ((uint32_t)(splitmix64(kSeedS + 8ull * g + 2ull) & 1ull) << 13);
      const float u = unit24(splitmix64(kSeedS + 8ull * g + 1ull));
      uint32_t child[2];
      for (uint32_t side = 0; side < 2; ++side) {
        const uint64_t hs = splitmix64(kSeedS + 8ull * g + 4ull + side);
        const bool internal = d + 1u < max_depth && depth.size() < kCap &&
                              (d + 1u < full_levels || (uint32_t)((hs >> 20) % 1000ull) < split_permille);
        if (internal) {
          child[side] = (uint32_t)depth.size();
          depth.push_back((uint8_t)(d + 1u));
        } else {
          volatile float c = unit24(splitmix64(kSeedS + 8ull * g + 6ull + side)) - 0.5f;
          volatile float v = c * 0.2f;
          child[side] = fbits(v);
          en |= 1u << (14 + side);
        }
      }
      if (lines && total + n < cap_lines) {
        uint32_t* r = lines + (total + n) * 4u;
        r[0] = fbits(dist == 1 ? u * 2.0f - 1.0f : u);
        r[1] = en;
        r[2] = child[0];
        r[3] = child[1];
      }
    }
    total += depth.size();
  }
  if (first) first[T] = total;
  return (int64_t)total;
}

int ddt_synth_tuples_host(void* out_, uint64_t row0, size_t n, uint32_t F, int dist, uint32_t missing_bits) {
  if (!out_ || F < 1 || F > 2048) return DDT_EINVAL;
  uint32_t* out = reinterpret_cast<uint32_t*>(out_);
  const uint32_t W = (F + 3u) / 4u * 4u;
  for (size_t r = 0; r < n; ++r)
    for (uint32_t j = 0; j < W; ++j) {
      uint32_t bits = 0;
      if (j < F) {
        const uint64_t h = splitmix64(kSeedX + (row0 + r) * (uint64_t)F + j);
        float v = unit24(h);
        if (dist == 1) {
          v = v * 2.0f - 1.0f;
          bits = (((h >> 8) & 0xFFFFull) % 20ull == 0ull) ? missing_bits : fbits(v);
        } else {
          bits = fbits(v);
        }
      }
      out[r * W + j] = bits;
    }
  return DDT_OK;
}

int ddt_synth_tuples_device(ddt_engine* e, void* d_out, uint64_t row0, size_t n, uint32_t F, int dist,
                            uint32_t missing_bits, void* stream) {
  if (!e) return DDT_EINVAL;
  if (!d_out || F < 1 || F > 2048) return fail(e, DDT_EINVAL, "bad synth arguments");
  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  hipError_t r = launch_synth_tuples(reinterpret_cast<uint32_t*>(d_out), row0, n, F, dist, missing_bits,
                                     reinterpret_cast<hipStream_t>(stream));
  if (r != hipSuccess) return fail(e, DDT_EHIP, "synth_tuples -> %s", hipGetErrorString(r));
  return DDT_OK;
}

}  // extern "C"