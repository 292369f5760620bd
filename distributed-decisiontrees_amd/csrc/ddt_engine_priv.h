// ddt_engine_priv.h -- the engine object and the host-side helpers shared by the translation units of libddt.so
// (ddt_engine.cpp: launches, feeder, C-ABI; ddt_model.cpp / ddt_image.cpp / ddt_choice.cpp: perfect-tree models -- parsing, device images,
// kernel choice; ddt_sparse_host.cpp: sparse forests; ddt_comm.cpp: RCCL).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include "ddt_internal.h"

namespace ddt {

constexpr uint32_t kMaxLdsBytes = 160u * 1024u;     // MI355X: 160 KiB LDS per CU / workgroup
constexpr uint32_t kStreamLdsBudget = 40u * 1024u;  // stream kernels: keep >= 4 resident blocks per CU
constexpr size_t kFeederMaxChunkBytes = 512u << 20;  // tuples per feeder slot (three pinned + three device buffers of this size at most)
constexpr int kFeederSlots = 3, kQSlots = 1 + kFeederSlots;

inline double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// sorted distinct threshold keys per feature of one ensemble (q16 path)
struct RankTables {
  std::vector<std::vector<uint32_t>> keys;  // [W], ascending as signed int32
  uint32_t max_len = 0;
};

// host form of the rank tables of the q16 pre-pass (rank_kernel's flat tables + the LDS images of the LDS-resident pre-pass)
struct RankHostTables {
  std::vector<uint32_t> tab, tabK, pimg;
  std::vector<uint16_t> tabS;
  PrepassPlan pplan{};
  uint32_t Kpad = 0;
};
// ... and their device copies when no Ensemble owns them (rank-quantised sparse forests)
struct RankDevice {
  void* d_tables = nullptr;
  void* d_tabK = nullptr;
  void* d_tabS = nullptr;
  void* d_prepass = nullptr;
  PrepassPlan prepass;
  uint32_t Kpad = 0;
};

// One part of a rank-quantised ensemble that is scored in parts (more than 32767 distinct thresholds on a feature: Q16Aux): chunks
// [chunk_begin, chunk_begin + chunks) of the cluster-major image, ranked against tables of their own
struct Q16Part {
  uint32_t chunk_begin = 0, chunks = 0;
  RankDevice rank;
};

// One ensemble as parsed from the reference wire format: the trees this engine holds of one class
// (single-output models have exactly one ensemble), plus its device image for the active variant.
struct Ensemble {
  std::vector<uint32_t> ids;    // global tree ids in stream order
  std::vector<uint32_t> thr;    // [T][nint]   raw fp32 bit patterns (heap order, 0-based)
  std::vector<uint16_t> fidx;   // [T][nint]
  std::vector<uint8_t> mright;  // [T][nint]
  std::vector<uint32_t> leaf;   // [T][nleaf]
  void* d_img = nullptr;
  size_t img_bytes = 0;
  uint32_t img_trees = 0, img_chunks = 0;
  // rank-quantised path only: image with miss_right flags, per-feature threshold tables
  void* d_img_slow = nullptr;
  void* d_tables = nullptr;
  void* d_tabK = nullptr;   // q16: per-feature search parameters (Q16Aux::tabP)
  void* d_tabS = nullptr;   // q16: bucket starts (Q16Aux::tabS)
  void* d_prepass = nullptr;  // q16: LDS images of the LDS-resident rank pre-pass (Q16Aux::prepass_img), one per feature group
  PrepassPlan prepass;        // geometry of those images; groups == 0: transpose + rank kernels
  uint32_t Kpad = 0;
  std::vector<Q16Part> parts;  // >= 2: the ensemble is scored in parts (their tables live here, d_tables .. d_prepass above stay empty)
  uint32_t trees() const { return (uint32_t)ids.size(); }
};

// A sparse (explicit-children) forest as loaded: this engine's shard, node lines re-based per tree (include/ddt.h
// ddt_load_model_sparse); device images: ddt_internal.h "Sparse forests".
struct SparseForest {
  std::vector<uint32_t> ids;      // global tree ids held by this engine, stream order
  std::vector<uint64_t> first;    // [trees + 1] line index of each tree's root inside `lines`
  std::vector<uint32_t> lines;    // 4 words per internal node
  uint32_t max_depth = 0;         // deepest leaf (levels of compares on the longest path)
  void* d_top = nullptr;          // [groups][8][12 * 2^K] top images
  void* d_deep = nullptr;         // deep records
  size_t top_bytes = 0, deep_bytes = 0;
  uint32_t groups = 0;
  uint32_t r_rounds = 1;          // "sparse_r_*" images: record hops on the longest path below the top image (SparseAux::max_rounds)
  uint32_t trees() const { return (uint32_t)ids.size(); }
};

}  // namespace ddt

struct ddt_copy_pool;  // ddt_engine.cpp

struct ddt_engine {
  using Ensemble = ddt::Ensemble;
  using PrepassPlan = ddt::PrepassPlan;
  int device = -1;
  hipDeviceProp_t prop{};
  bool loaded = false;
  ddt_params p{};
  uint32_t nint = 0, nleaf = 0;
  uint32_t num_classes = 1;
  std::vector<Ensemble> ens;  // one per class
  int forced_variant = -1;
  int variant_id = 0;
  // feeder
  size_t feeder_rows = 1u << 20;
  int feeder_threads = 8;   // host threads that copy a chunk into the pinned staging buffer (one thread: ~26 GB/s < PCIe)
  hipStream_t fs[ddt::kFeederSlots] = {};   // three slots: staging of chunk k+2 | link transfer of k+1 | compute of k
  hipEvent_t fe[ddt::kFeederSlots] = {};
  hipStream_t copy_stream = nullptr;        // every host-to-device copy of the feeder, in order (one stream: full link rate)
  hipEvent_t fe_in[ddt::kFeederSlots] = {};  // chunk b has arrived on the device
  void* pin_in[ddt::kFeederSlots] = {};
  void* pin_out[ddt::kFeederSlots] = {};
  void* dev_in[ddt::kFeederSlots] = {};
  void* dev_out[ddt::kFeederSlots] = {};
  ddt_copy_pool* pool = nullptr;                         // persistent staging threads (feeder_threads - 1 workers + the caller)
  std::vector<std::pair<void*, size_t>> pinned;          // host ranges the caller pinned through ddt_host_register
  size_t feeder_cap_rows = 0, feeder_cap_words = 0, feeder_cap_outs = 0;
  // classify workspace (grow-only)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // rank-quantised path workspace (grow-only): transposed tuples, ranks, per-tile flags
  // slot 0: ddt_score_device / ddt_classify_device (stream ordered); slots 1..: the feeder's streams
  void* q_xT[ddt::kQSlots] = {};
  void* q_q[ddt::kQSlots] = {};
  void* q_flags[ddt::kQSlots] = {};
  void* q_state[ddt::kQSlots] = {};    // ensembles scored in parts: [2][rows] fp32 accumulator + running total between the parts' launches
  void* q_split[ddt::kQSlots] = {};    // small batches cut at the clusters (Q16Aux::split): [clusters][rows] fp32 partial sums
  uint64_t q_split_floats[ddt::kQSlots] = {};
  uint64_t q_rows[ddt::kQSlots] = {};  // capacity in rows (multiple of 1024)
  bool q_xT_valid[ddt::kQSlots] = {};  // q_xT holds the transposed tuples of the batch being scored (set by a part's transpose, cleared by the next batch)
  int q_slot = 0;
  int q16_grouped_prepass = 1;  // option "q16_grouped_prepass": 0 = never split the pre-pass over feature groups
  int q16_prepass_groups = 0;   // option "q16_prepass_groups": force the number of feature groups (A/B), 0 = automatic
  uint32_t q16_max_table = 38848;  // option "q16_max_table" (= kQ16MaxTable below): distinct thresholds per feature one part's rank table may hold (A/B, tests: 32767 = the limit until round 6)
  int q16_fused_prepass = 1;  // option "q16_fused_prepass": 0 forces the transpose + rank kernels (A/B, tests)
  int q16_prepass_nt = 0;     // option "q16_prepass_nt": bit 0 = nontemporal stores of the rank tiles, bit 1 = nontemporal tuple loads (A/B)
  int q16_persistent = -1;    // option "q16_persistent": 1 / 0 = prefer / never pick the persistent "_p" kernel, -1 = automatic
  int q16_cluster_split = -1; // option "q16_cluster_split": 1 / 0 = always / never cut a launch at the clusters, -1 = automatic (batches of up to q16_split_max_tiles tiles)
  uint32_t q16_split_max_tiles = 384;  // option "q16_split_max_tiles" (measured at 1000 trees: 256 tiles 344 vs 407 us, 512 tiles 632 vs 624; profiles/r06_small_batches.md)
  uint32_t sparse_split_max_tiles = 256;  // option "sparse_split_max_tiles": the 32-bit-rank sparse kernels' launch is cut into slices of C PU groups on batches of up to this many of their tiles
  int q16_split_groups = -1;  // option "q16_split_groups": slices finer than the clusters (a partial sum per PU group); -1 automatic, 0 never, > 0 that many slices
  bool collective_job = false;  // set by ddt_comm_create* / ddt_group_create* with more than one rank: collectives share the CUs with the scoring
  // "_p" kernels, multi-class models whose classes hold equally many trees: the classes' images back to back (fast / slow), so that
  // ONE launch walks every class (Q16Aux::n_segs); mc_seg_chunks = chunks per class, 0 = not built (one launch per class)
  void* d_mc_img = nullptr;
  void* d_mc_img_slow = nullptr;
  uint32_t mc_seg_chunks = 0;
  // optional per-call kernel timing (option "kernel_timing"): start / before scoring kernel / end
  // Every timed launch takes an event triple from a ring; ddt_get_stats resolves the pending ones (waiting for the newest), so a caller
  // may queue up to kTimingRing launches without a host synchronisation in between.
  bool kernel_timing = false;
  static constexpr int kTimingRing = 64;
  hipEvent_t tev[kTimingRing][3] = {};
  int tev_head = 0, tev_pending = 0;   // pending triples: tev_head - tev_pending .. tev_head - 1 (mod kTimingRing)
  hipEvent_t* tev_cur = nullptr;       // the triple of the launch being issued (its middle event is recorded by the kernel launcher)
  // multi-class models: the classes' scoring launches alternate between the caller's stream and this one, so that the tail of
  // one launch (the last, partly filled wave of blocks) overlaps the next class's launch (option "class_streams", default 1)
  hipStream_t class_stream = nullptr;
  hipEvent_t class_ev[2] = {nullptr, nullptr};
  hipEvent_t ev_fork = nullptr;  // set around class 0's launch of a rank-quantised multi-class call: recorded between its pre-pass and its scoring kernel
  int class_streams = 1;
  bool q16_walk_padding = false;  // option "q16_walk_padding" (A/B): walk the EMPTY padding trees of the last chunk as well
  int stream_blocks_per_cu = 0;  // option "stream_blocks_per_cu": persistent stream kernel, blocks per CU (0 = resident blocks)
  int wall_clock_khz = 100000;   // hipDeviceAttributeWallClockRate: the clock s_memrealtime counts (10 ns ticks on MI355X)
  int stream_res_tiles = 0;      // option "stream_res_tiles": stream kernel, score slots per wave for the phased result stores (0 = auto, 1 = direct stores)
  int stream_window_ticks = 0;   // option "stream_window_ticks": ... write window in 10 ns ticks (0 = default)
  // sparse forests (ddt_load_model_sparse)
  bool sparse = false;
  std::vector<ddt::SparseForest> sps;  // one per class (single-output models: exactly one)
  int sparse_top_levels = -1;   // option "sparse_top_levels": K, -1 = the most the LDS takes
  int sparse_deep_order = 0;    // option "sparse_deep_order": 0 = level order (default: measured faster), 1 = depth-first per sub-tree
  int sparse_dk = 1;            // option "sparse_dk": 1 = dense-level-K sparse kernels where they exist (default), 0 = 16-byte level K-1 records in LDS (128- / 64-tuple tiles only)
  int sparse_dm = -1;           // option "sparse_dm": dense mid levels of 8-byte records below the top image (-1 = where the forest fills them, 0 = never, 1..3 = exactly that many)
  int sparse_peel_last = 1;     // option "sparse_peel_last": 1 = the last possible round of a sparse kernel's deep loop issues no gather (default), 0 = as before round 5 (A/B)
  int sparse_dp = -1;           // option "sparse_dp": dense pair records for the two levels below the top image (-1 = where the forest fills level K at least half and level K+1 a quarter, 0 = never, 1 = always)
  int sparse_idle_oob = 1;      // option "sparse_idle_oob": 1 = a finished walker's gather is sent out of the buffer's range (no cache access; default), 0 = it re-reads record 0 (A/B)
  int sparse_q16 = 1;           // option "sparse_q16": 1 = rank-quantised sparse kernels when they fit (default), 0 = fp32 feature tiles
  ddt::RankDevice sp_rank;      // rank tables of the loaded sparse forests (rank-quantised kernels only; "sparse_r_*": the directories)
  int sparse_r32 = -1;          // option "sparse_r32": 32-bit ranks + pair records on every deep level (-1 = where it measured faster: ddt_sparse_host.cpp sparse_rebuild, 0 = never, 1 = wherever they fit)
  void* sp_r32_tab = nullptr;   // "sparse_r_*": the key blocks of the rank32 pre-pass (R32Aux::tab)
  size_t sp_r32_tab_bytes = 0;
  uint32_t sp_r32_blk_log2 = 2;
  // feature compaction (ddt_engine.cpp plan_feature_compaction): compact tuple word -> feature number, its inverse, the device copy of the map
  std::vector<uint16_t> fmap, finv;
  void* d_fmap = nullptr;
  int generic_via_sparse = 1;   // option "generic_via_sparse": a perfect-tree model that would land on `generic` goes to the sparse-forest kernels
  bool perfect_as_sparse = false;  // ... and did: e->sps was built from e->ens (ddt_engine.cpp maybe_score_as_sparse)
  int feature_compaction = 1;   // option "feature_compaction": 0 = never (models of more than 64 tuple words then stay off the rank-quantised kernels)
  int leaf_domain_check = 1;    // option "leaf_domain_check": reject leaves outside the exact domain of the reference adder
  ddt_stats st{};
  char err[256] = {0};
};

namespace ddt {

inline int fail(ddt_engine* e, int code, const char* fmt, ...) {
  if (e) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e->err, sizeof(e->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define HIP_TRY(e, call)                                                                          \
  do {                                                                                            \
    hipError_t _r = (call);                                                                       \
    if (_r != hipSuccess) return fail((e), DDT_EHIP, "%s -> %s", #call, hipGetErrorString(_r));   \
  } while (0)

// RAII: make the engine's device current for the duration of a C-ABI call and restore the caller's device (several
// engines on different GPUs may live in one process; kernel launches and allocations follow the CURRENT device)
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    else prev = -1;  // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// ---- ddt_model.cpp: the perfect-tree wire format ----
uint32_t wlines_min(uint32_t D);
uint32_t flines_min(uint32_t D);
int validate(ddt_engine* e, const ddt_params* p, size_t n_wlines, size_t n_flines);
int parse_trees(ddt_engine* eng, const ddt_params* p, const uint32_t* w, const uint16_t* f, std::vector<uint32_t> ids, ddt_engine::Ensemble* out);
uint32_t padded_trees(const Variant& v, uint32_t T);
uint32_t max_trees(const ddt_engine* e);
// ---- ddt_image.cpp: device images, rank tables, pre-pass images ----
uint32_t q16_words(const ddt_engine* e);                  // tuple words the rank-quantised kernels see (feature compaction)
inline uint32_t q16_feat(const ddt_engine* e, uint32_t j) { return e->fmap.empty() ? j : e->finv[j]; }
RankTables rank_tables(const ddt_engine* e);
bool build_prepass_image(const RankTables& rt, uint32_t W, uint32_t groups_wanted, bool allow_one, bool allow_many, std::vector<uint32_t>* pimg,
                         PrepassPlan* plan);
bool prepass_plan_exists(const ddt_engine* e);
uint32_t total_trees(const ddt_engine* e);
uint32_t cm_position(uint32_t i, uint32_t T, uint32_t Cc);
int build_image(ddt_engine* e, const Variant& v, ddt_engine::Ensemble& m);
int build_image_q16(ddt_engine* e, const Variant& v, ddt_engine::Ensemble& m, const RankTables& rt, bool upload_tables);
// ---- ddt_choice.cpp: which kernel ----
bool classes_equal(const ddt_engine* e);
bool s2_disabled();
bool deep_disabled();
bool variant_fits(const Variant& v, const ddt_engine* e);
int auto_variant(const ddt_engine* e);
int select_and_build(ddt_engine* e);
int maybe_score_as_sparse(ddt_engine* e);
uint32_t tuple_words(const ddt_params& p);               // (ddt_model.cpp)
// ---- ddt_engine.cpp ----
uint32_t thr_key(const ddt_params& p, uint32_t bits);
std::vector<uint32_t> shard_of(const std::vector<uint32_t>& ids, uint32_t g, uint32_t G);
void free_images(ddt_engine* e);
void free_q16_workspace(ddt_engine* e);
int find_variant(const char* name);
bool split_fits(uint32_t partials, size_t n);                                  // ddt_engine.cpp: a cut launch's partial sums within the workspace cap
int ensure_split_workspace(ddt_engine* e, uint64_t floats, float** out);
bool leaf_outside_exact_domain(uint32_t bits);
// one scoring pass of the loaded model (perfect or sparse) over device-resident tuples, asynchronous on `s`
int engine_score_device(ddt_engine* e, const void* d_tuples, size_t n, float* d_scores, hipStream_t s);
int engine_classify_device(ddt_engine* e, const void* d_tuples, size_t n, float* d_class_scores, int32_t* d_labels, hipStream_t s);
int ensure_q16_workspace(ddt_engine* e, size_t n);
// a multi-rank job was set up on this engine: from now on (and for the model that may already be loaded) the automatic kernel choice
// prefers the persistent rank-quantised kernel, whose blocks take tiles from a ticket counter and therefore do not wait for the
// CUs a collective's kernels occupy (profiles/r04_cu_mask_probe.md)
void engine_enter_collective_job(ddt_engine* e);
// kernel_timing: open / close the event triple of one launch on stream s (timing_begin records the start event)
int timing_begin(ddt_engine* e, hipStream_t s);
int timing_end(ddt_engine* e, hipStream_t s);
void timing_resolve(ddt_engine* e, int keep_pending);
// rank tables: host packing (no HIP call) and upload; tables longer than kQ16MaxTable keys do not fit the u16 ranks
// Ranks must stay below 0xFFFF and a table must fit the LDS of ONE block of rank_kernel: Kpad entries + the bucket starts (kQ16RankBuckets u16)
// within MI355X's 160 KiB per workgroup.  Up to 32767 keys the table is padded to the next power of two (128 KiB + 8 KiB); a longer one to a multiple
// of 32 entries with at least 64 pads behind its keys (q16_table_pad: a search probes up to P - 2 entries past its start, and P <= 64 unless the keys are
// degenerate -- the kernel clamps then) -- 38848 keys is what 152 KiB hold.
// (Until round 6 the limit was 32767: the reference's own example, 512 trees x depth 12 x 32 features = 65.4 k thresholds per feature, then
// needed a THIRD part of 16 trees -- a rank pre-pass and a scoring launch over every tuple for 3 % of the trees.)
constexpr uint32_t kQ16MaxTable = 38848;
constexpr uint32_t q16_table_pad(uint32_t max_len) {  // Q16Aux::Kpad: entries per table, > max_len (entry Kpad - 1 is always an INT_MAX pad)
  if (max_len >= 32768u) return (max_len + 64u + 31u) / 32u * 32u;
  uint32_t k = 2;
  while (k <= max_len) k <<= 1;
  return k;
}
constexpr uint32_t q16_rank_lds_bytes(uint32_t Kpad) { return Kpad * 4u + kQ16RankBuckets * 2u; }  // rank_kernel's LDS
static_assert(q16_rank_lds_bytes(q16_table_pad(kQ16MaxTable)) <= kMaxLdsBytes && q16_rank_lds_bytes(q16_table_pad(kQ16MaxTable + 32u)) > kMaxLdsBytes,
              "kQ16MaxTable = the longest table one block's LDS holds");
static_assert(kQ16MaxTable == 38848u, "ddt_engine::q16_max_table's initialiser");
static_assert(q16_table_pad(32767u) == 32768u && q16_table_pad(32768u) == 32832u && q16_table_pad(8160u) == 8192u, "table padding");
void finish_rank_tables(RankTables& rt);  // sort + unique every feature's keys, set max_len
int pack_rank_tables(ddt_engine* e, const RankTables& rt, uint32_t W, bool want_prepass, RankHostTables& h);
int upload_rank_tables(ddt_engine* e, const RankHostTables& h, RankDevice& d);
void free_rank_device(RankDevice& d);
// ddt_sparse_host.cpp
void sparse_free(ddt_engine* e);
int sparse_launch(ddt_engine* e, uint32_t cls, const void* d_tuples, size_t n, float* d_scores, hipStream_t s, bool reuse_prepass = false);
int sparse_rebuild(ddt_engine* e);

}  // namespace ddt
