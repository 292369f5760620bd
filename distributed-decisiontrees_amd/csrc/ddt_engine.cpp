// ddt_engine.cpp -- host side of libddt.so: model stream parsing/validation, device image packing,
// kernel variant selection, the pinned double-buffered tuple feeder, and the C-ABI of include/ddt.h.
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

uint32_t wlines_min(uint32_t D) { return (uint32_t)((((1ull << (D + 1)) - 1) + 3) / 4); }
uint32_t flines_min(uint32_t D) { return (uint32_t)((((1ull << D) - 1) + 7) / 8); }
uint32_t tuple_words(const ddt_params& p) { return (p.num_features + 3u) / 4u * 4u; }

int validate(ddt_engine* e, const ddt_params* p, size_t n_wlines, size_t n_flines) {
  if (!p) return fail(e, DDT_EINVAL, "params is NULL");
  if (p->num_trees == 0) return fail(e, DDT_EINVAL, "num_trees == 0");
  if (p->num_levels < 1 || p->num_levels > 16) return fail(e, DDT_EINVAL, "num_levels %u not in 1..16 (CSR205 is 4 bits)", p->num_levels);
  if (p->num_features < 1 || p->num_features > 2048) return fail(e, DDT_EINVAL, "num_features %u not in 1..2048 (DTPU.sv:72)", p->num_features);
  if (p->cmp_mode > 1) return fail(e, DDT_EINVAL, "cmp_mode %u", p->cmp_mode);
  if (p->sum_mode > 2) return fail(e, DDT_EINVAL, "sum_mode %u", p->sum_mode);
  const uint32_t c = p->clusters_per_tuple;
  if (c != 1 && c != 2 && c != 4 && c != 8) return fail(e, DDT_EINVAL, "clusters_per_tuple %u not in {1,2,4,8}", c);
  if (p->reserved[0] | p->reserved[1] | p->reserved[2]) return fail(e, DDT_EINVAL, "reserved fields must be 0");
  if (p->weights_lines_per_tree < wlines_min(p->num_levels))
    return fail(e, DDT_EINVAL, "weights_lines_per_tree %u < %u", p->weights_lines_per_tree, wlines_min(p->num_levels));
  if (p->findex_lines_per_tree < flines_min(p->num_levels))
    return fail(e, DDT_EINVAL, "findex_lines_per_tree %u < %u", p->findex_lines_per_tree, flines_min(p->num_levels));
  if (n_wlines < (size_t)p->num_trees * p->weights_lines_per_tree) return fail(e, DDT_EINVAL, "weights stream too short");
  if (n_flines < (size_t)p->num_trees * p->findex_lines_per_tree) return fail(e, DDT_EINVAL, "feature-index stream too short");
  return DDT_OK;
}

// Parse the trees `ids` out of the two streams (A2 packing: word k of a line = bits [32k+31:32k]).
int parse_trees(ddt_engine* eng, const ddt_params* p, const uint32_t* w, const uint16_t* f, std::vector<uint32_t> ids,
                Ensemble* out) {
  Ensemble m;
  const uint32_t D = p->num_levels, nint = (1u << D) - 1u, nleaf = 1u << D;
  const uint32_t T = (uint32_t)ids.size();
  try {
    m.thr.resize((size_t)T * nint);
    m.fidx.resize((size_t)T * nint);
    m.mright.resize((size_t)T * nint);
    m.leaf.resize((size_t)T * nleaf);
  } catch (const std::bad_alloc&) {
    return fail(eng, DDT_ENOMEM, "host model allocation failed");
  }
  for (uint32_t i = 0; i < T; ++i) {
    const uint32_t* wt = w + (size_t)ids[i] * p->weights_lines_per_tree * 4u;
    const uint16_t* ft = f + (size_t)ids[i] * p->findex_lines_per_tree * 8u;
    for (uint32_t n = 0; n < nint; ++n) {
      const uint16_t en = ft[n];
      const uint32_t j = en & 0x7FFu;  // DTPU.sv:628
      if (j >= p->num_features)
        return fail(eng, DDT_EINVAL, "tree %u node %u: feature index %u >= num_features %u", ids[i], n, j, p->num_features);
      if (en & (1u << 14))  // "next node is leaf" has no well-defined result in the published RTL
        return fail(eng, DDT_EUNSUPPORTED, "tree %u node %u: early-leaf flag (bit 14) is not supported; pad the tree to a perfect one", ids[i], n);
      m.thr[(size_t)i * nint + n] = wt[n];
      m.fidx[(size_t)i * nint + n] = (uint16_t)j;
      m.mright[(size_t)i * nint + n] = (uint8_t)((en >> 13) & 1u);  // DTPU.sv:659
    }
    for (uint32_t l = 0; l < nleaf; ++l) {
      const uint32_t lb = wt[nint + l];
      // The GPU adds are IEEE-754; the reference's FloPoCo adder treats sub-normal / Inf / NaN inputs as normals, keeps -0
      // and has no sub-normal results (FPAdder_2cycles_latency.v:313-320,376-385 behind the {0, |bits} wrapper of
      // FPAddersReduceTree.sv:94-95).  With every leaf +0 or normal in [2^-102, 2^96) no partial sum of fewer than 2^32 leaves
      // can be sub-normal (sums are multiples of the smallest leaf ulp, >= 2^-125), overflow or be -0: on that domain the two
      // adders differ in exactly one case, which sum_mode 2 reproduces (ddt_device.h radd_exact).
      if (p->sum_mode != 1 && eng && eng->leaf_domain_check && leaf_outside_exact_domain(lb))
        return fail(eng, DDT_EUNSUPPORTED,
                    "tree %u leaf %u = 0x%08X: leaves other than +0 and normal values with 2^-102 <= |v| < 2^96 (-0, sub-normal, tiny, huge, "
                    "Inf, NaN) are outside the domain where the IEEE adds are held to the reference adder (flush them to +0 when "
                    "exporting, use sum_mode 1, or set option leaf_domain_check = 0)",
                    ids[i], l, lb);
      m.leaf[(size_t)i * nleaf + l] = lb;
    }
  }
  m.ids = std::move(ids);
  *out = std::move(m);
  return DDT_OK;
}

bool leaf_outside_exact_domain(uint32_t bits) {
  const uint32_t ex = (bits >> 23) & 0xFFu;
  return bits != 0u && (ex < 25u || ex > 222u);  // -0, sub-normals, |v| < 2^-102, |v| >= 2^96, Inf, NaN
}

uint32_t thr_key(const ddt_params& p, uint32_t bits) {
  if (p.cmp_mode == 0) return bits;
  if ((bits & 0x7FFFFFFFu) > 0x7F800000u) return 0x80000000u;  // x < NaN is never true -> always right
  return ieee_key(bits);
}

uint32_t padded_trees(const Variant& v, uint32_t T) {
  const bool chunked = v.kind == kKindTile || v.kind == kKindQ16;
  uint32_t granule = (chunked && v.chunk_trees > 8) ? (uint32_t)v.chunk_trees : 8u;
  if (v.kind == kKindTile && (v.opt & 2) && granule < 2u * (uint32_t)v.chunk_trees) granule = 2u * (uint32_t)v.chunk_trees;  // even chunk count
  if (T == 0) T = 1;  // an empty shard (T < shard_count * ceil(T / shard_count)) is one group of EMPTY slots: scores +0
  return (T + granule - 1u) / granule * granule;  // whole PU groups of 8 (and whole chunks)
}

uint32_t max_trees(const ddt_engine* e) {
  uint32_t t = 0;
  for (const Ensemble& m : e->ens) t = m.trees() > t ? m.trees() : t;
  return t;
}

// ---- feature compaction (round 6) --------------------------------------------------------------------------------
// The rank-quantised kernels take tuples of at most 64 words (the u16 tile of 1024 tuples must fit LDS); the reference takes F <= 2048
// (DTPU.sv:22-25,628).  A model of more than 64 tuple words that TESTS at most 64 distinct features (ddt_engine::fmap: compact index ->
// feature number) still runs on them: the rank pre-pass gathers only those columns into its transposed intermediate, and tables, tiles,
// node records and kernels see a tuple of q16_words() words.  Everything else (the wire format, the feeder, ddt_info) keeps the caller's width.
uint32_t q16_words(const ddt_engine* e) { return e->fmap.empty() ? tuple_words(e->p) : (uint32_t)((e->fmap.size() + 3u) / 4u * 4u); }
inline uint32_t q16_feat(const ddt_engine* e, uint32_t j) { return e->fmap.empty() ? j : e->finv[j]; }

// q16: sorted distinct threshold keys (comparator domain) per feature, over the trees of EVERY ensemble of the
// engine: the classes of a multi-class model share one set of tables, so one transpose + rank pre-pass per batch
// serves all K scoring launches (launch_classify)
RankTables rank_tables(const ddt_engine* e) {
  RankTables rt;
  const uint32_t W = q16_words(e), nint = e->nint;
  rt.keys.resize(W);
  for (const Ensemble& m : e->ens)
    for (uint32_t i = 0; i < m.trees(); ++i)
      for (uint32_t n = 0; n < nint; ++n)
        rt.keys[q16_feat(e, m.fidx[(size_t)i * nint + n])].push_back(thr_key(e->p, m.thr[(size_t)i * nint + n]));
  finish_rank_tables(rt);
  return rt;
}

// LDS-resident rank pre-pass (fused_rank_kernel / grouped_rank_kernel, ddt_internal.h PrepassPlan).  The features are
// cut into G = 1, 2, 4 or 8 groups of 8 / 4 / 2 / 1 tuple lines whose tables fit one CU's LDS.  Exact LDS image of a group:
//   per feature  a skewed table of K + P keys (INT_MAX pads; entry i at word i + i/32)
//   then         the bucket starts of all its features (u16: number of keys in the buckets below)
//   then         per feature a segment table, kQ16Segments words {first bucket | log2(bucket width) << 16}
//   then         per feature 8 parameter words {K, lo, span, table byte offset, starts byte offset, segment table byte
//                offset, segment shift, 0}
// Bucket index of a key: d = min(key - lo, span); segment = d >> segment shift (<= 32 equal slices of the key range);
// bucket = first[segment] + ((d & segment mask) >> log2 width[segment]).  The bucket WIDTH is per segment: dense slices
// of the key range get narrow buckets, sparse ones wide buckets (thresholds uniform in VALUE are exponentially dense in
// IEEE key space -- with one global width half of them shared 1/13 of the buckets).  Widths are chosen greedily under
// the LDS budget: keep halving the width of the segment that holds the fullest bucket; P = power of two above the
// fullest bucket, so log2(P) probes from starts[bucket] finish the count.
struct SegFeature {
  uint32_t K = 0, lo = 0x7FFFFFFFu, span = 0, seg_shift = 0, nseg = 1;
  uint32_t sh[kQ16Segments] = {};  // log2(bucket width) per segment
};

uint32_t seg_buckets(const SegFeature& f, uint32_t s) { return 1u << (f.seg_shift - f.sh[s]); }

// fullest bucket of one segment (its keys, sorted) at bucket width 2^sh
uint32_t seg_fullest(const std::vector<uint32_t>& keys, const SegFeature& f, uint32_t sh) {
  uint32_t best = 0, run = 0, prev = 0xFFFFFFFFu;
  const uint32_t mask = (1u << f.seg_shift) - 1u;  // seg_shift <= 27
  for (uint32_t key : keys) {
    const uint32_t b = ((key - f.lo) & mask) >> sh;
    run = b == prev ? run + 1u : 1u;
    prev = b;
    best = run > best ? run : best;
  }
  return best;
}

// one group (features [f0, f1)): returns false when it cannot fit kMaxLdsBytes; img may be NULL to only ask
bool build_prepass_group(const RankTables& rt, uint32_t f0, uint32_t f1, std::vector<uint32_t>* img, uint32_t* par_off, uint32_t* P_out) {
  const uint32_t nf = f1 - f0;
  std::vector<SegFeature> F(nf);
  std::vector<std::vector<std::vector<uint32_t>>> seg_keys(nf);  // keys of each (feature, segment)
  for (uint32_t j = 0; j < nf; ++j) {
    const std::vector<uint32_t>& k = rt.keys[f0 + j];
    SegFeature& f = F[j];
    f.K = (uint32_t)k.size();
    if (!k.empty()) {
      f.lo = k.front();
      f.span = k.back() - k.front();  // int32 order: the difference fits 32 bits
      while ((f.span >> f.seg_shift) >= kQ16Segments) ++f.seg_shift;
      f.nseg = (f.span >> f.seg_shift) + 1u;
    }
    seg_keys[j].resize(f.nseg);
    for (uint32_t key : k) seg_keys[j][(key - f.lo) >> f.seg_shift].push_back(key);
  }
  struct Item {
    uint32_t full, j, s;
  };
  auto less_full = [](const Item& a, const Item& b) { return a.full < b.full; };
  for (uint32_t P = 2; P <= 65536u; P <<= 1) {
    // LDS left for the bucket starts once the tables carry P pads
    size_t words = 0;
    for (uint32_t j = 0; j < nf; ++j) {
      const uint32_t len = F[j].K + P;
      words += len + (len >> 5) + 1u;
    }
    words = (words + 3u) & ~(size_t)3u;
    const size_t fixed = words * 4u + (size_t)nf * (kQ16Segments + 8u) * 4u + 32u;
    if (fixed >= kMaxLdsBytes) return false;  // more pads only make it worse
    const size_t budget = (kMaxLdsBytes - fixed) / 2u;  // u16 entries for the whole group
    // one bucket per segment to start with, then keep halving the bucket width of the segment with the fullest bucket
    size_t used = 0;
    std::vector<Item> heap;
    std::vector<size_t> feat_buckets(nf, 0);
    for (uint32_t j = 0; j < nf; ++j)
      for (uint32_t s = 0; s < F[j].nseg; ++s) {
        F[j].sh[s] = F[j].seg_shift;
        ++used;
        ++feat_buckets[j];
        heap.push_back({(uint32_t)seg_keys[j][s].size(), j, s});
      }
    if (used > budget) continue;
    std::make_heap(heap.begin(), heap.end(), less_full);
    bool ok = false;
    for (;;) {
      std::pop_heap(heap.begin(), heap.end(), less_full);
      Item it = heap.back();
      if (it.full < P) {  // the fullest bucket of the whole group holds fewer than P keys
        ok = true;
        break;
      }
      SegFeature& f = F[it.j];
      const size_t cost = seg_buckets(f, it.s);  // halving the width adds as many buckets as the segment has
      if (f.sh[it.s] == 0u || used + cost > budget || feat_buckets[it.j] + cost > 32768u) break;  // cannot thin the fullest bucket
      --f.sh[it.s];
      used += cost;
      feat_buckets[it.j] += cost;
      it.full = seg_fullest(seg_keys[it.j][it.s], f, f.sh[it.s]);
      heap.back() = it;
      std::push_heap(heap.begin(), heap.end(), less_full);
    }
    if (!ok) continue;
    // layout
    std::vector<uint32_t> tab_off(nf), starts_off(nf), seg_off(nf);
    words = 0;
    for (uint32_t j = 0; j < nf; ++j) {
      const uint32_t len = F[j].K + P;
      tab_off[j] = (uint32_t)words * 4u;
      words += len + (len >> 5) + 1u;
    }
    words = (words + 3u) & ~(size_t)3u;
    size_t half = words * 2u;  // in u16 units
    for (uint32_t j = 0; j < nf; ++j) {
      starts_off[j] = (uint32_t)half * 2u;
      half += feat_buckets[j];
    }
    words = ((half + 1u) / 2u + 3u) & ~(size_t)3u;
    for (uint32_t j = 0; j < nf; ++j) {
      seg_off[j] = (uint32_t)words * 4u;
      words += kQ16Segments;
    }
    const uint32_t poff = (uint32_t)words * 4u;
    words += (size_t)nf * 8u;
    if (words * 4u > kMaxLdsBytes) continue;  // alignment padding pushed it over: next P has fewer buckets
    *par_off = poff;
    *P_out = P;
    if (!img) return true;
    img->assign(words, 0x7FFFFFFFu);
    for (uint32_t j = 0; j < nf; ++j) {
      const std::vector<uint32_t>& k = rt.keys[f0 + j];
      const SegFeature& f = F[j];
      for (uint32_t i = 0; i < k.size(); ++i) (*img)[tab_off[j] / 4u + i + (i >> 5)] = k[i];
      uint16_t* S = reinterpret_cast<uint16_t*>(img->data()) + starts_off[j] / 2u;
      uint32_t* seg = img->data() + seg_off[j] / 4u;
      uint32_t first = 0, run = 0;
      const uint32_t mask = (1u << f.seg_shift) - 1u;
      for (uint32_t s = 0; s < kQ16Segments; ++s) {
        if (s >= f.nseg) {
          seg[s] = 0u;
          continue;
        }
        seg[s] = first | (f.sh[s] << 16);
        const uint32_t nb = seg_buckets(f, s);
        std::vector<uint32_t> cnt(nb, 0u);
        for (uint32_t key : seg_keys[j][s]) ++cnt[((key - f.lo) & mask) >> f.sh[s]];
        for (uint32_t b = 0; b < nb; ++b) {
          S[first + b] = (uint16_t)run;  // run <= K <= 32767
          run += cnt[b];
        }
        first += nb;
      }
      uint32_t* Pp = img->data() + poff / 4u + (size_t)j * 8u;
      Pp[0] = f.K;
      Pp[1] = f.lo;
      Pp[2] = f.span;
      Pp[3] = tab_off[j];
      Pp[4] = starts_off[j];
      Pp[5] = seg_off[j];
      Pp[6] = f.seg_shift;
      Pp[7] = 0u;
    }
    return true;
  }
  return false;
}

// one candidate: G groups; pimg may be NULL to only plan
bool build_prepass_groups(const RankTables& rt, uint32_t W, uint32_t G, std::vector<uint32_t>* pimg, PrepassPlan* plan) {
  const uint32_t lines = 8u / G;  // tuple lines (4 features each) per group
  PrepassPlan pl{};
  std::vector<std::vector<uint32_t>> imgs(G);
  uint32_t used = 0;
  for (uint32_t g = 0; g < G; ++g) {
    const uint32_t f0 = g * lines * 4u < W ? g * lines * 4u : W, f1 = (g + 1u) * lines * 4u < W ? (g + 1u) * lines * 4u : W;
    if (f0 == f1) continue;  // narrow tuples: trailing groups are empty
    if (!build_prepass_group(rt, f0, f1, pimg ? &imgs[used] : nullptr, &pl.par_off[used], &pl.P[used])) return false;
    pl.line_lo[used] = g * lines;
    ++used;
  }
  if (used == 0) return false;
  if (pimg) {
    pimg->clear();
    for (uint32_t g = 0; g < used; ++g) {
      pl.img_off[g] = (uint32_t)(pimg->size() * 4u);
      pl.bytes[g] = (uint32_t)(imgs[g].size() * 4u);
      pimg->insert(pimg->end(), imgs[g].begin(), imgs[g].end());
    }
  }
  pl.groups = used;
  pl.lines = lines;
  *plan = pl;
  return true;
}

// groups_wanted: 0 = the cheapest G that fits, else exactly that G.  allow_one / allow_many: engine options.
// Cost of a candidate, ms per 100 M tuples of 32 features on one MI355X (fitted to profiles/archive/r02_prepass_ab_grid.log): a floor set by how
// the rows are read (G <= 2: whole 64-byte sectors per block; G = 4: half; G = 8: a quarter of every sector pulled
// through the L1) + the probes (log2 P dependent LDS reads with ~3.5-way bank conflicts), which hide less behind the
// loads the more of the time is load-bound.
bool build_prepass_image(const RankTables& rt, uint32_t W, uint32_t groups_wanted, bool allow_one, bool allow_many, std::vector<uint32_t>* pimg,
                         PrepassPlan* plan) {
  plan->groups = 0;
  if (W > 32u) return false;
  static const float base[4] = {3.28f, 3.14f, 3.72f, 4.62f}, per_probe[4] = {0.35f, 0.35f, 0.275f, 0.275f};
  uint32_t best_G = 0;
  float best = 0.f;
  for (uint32_t G = 1, i = 0; G <= kQ16MaxGroups; G <<= 1, ++i) {
    if (groups_wanted && G != groups_wanted) continue;
    if (G == 1u ? !allow_one : !allow_many) continue;
    PrepassPlan pl{};
    if (!build_prepass_groups(rt, W, G, nullptr, &pl)) continue;
    uint32_t P = 1, probes = 0;
    for (uint32_t g = 0; g < pl.groups; ++g) P = pl.P[g] > P ? pl.P[g] : P;
    while ((2u << probes) <= P) ++probes;  // log2 P
    const float cost = base[i] + per_probe[i] * (float)probes;
    if (!best_G || cost < best) best_G = G, best = cost;
  }
  if (!best_G) return false;
  return build_prepass_groups(rt, W, best_G, pimg, plan);
}

bool prepass_plan_exists(const ddt_engine* e) {
  PrepassPlan pl;
  if (!e->fmap.empty()) return false;  // compacted features: the pre-pass is the gathering transpose + rank_kernel
  return build_prepass_image(rank_tables(e), tuple_words(e->p), (uint32_t)e->q16_prepass_groups, e->q16_fused_prepass != 0, e->q16_grouped_prepass != 0,
                             nullptr, &pl);
}

uint32_t total_trees(const ddt_engine* e) {
  uint32_t t = 0;
  for (const Ensemble& m : e->ens) t += m.trees();
  return t;
}

constexpr uint32_t kQ16MinTreeLevels = 480;  // trees x levels from which the rank-quantised path wins with the LDS-resident pre-pass
                                             // (profiles/archive/r02_sweep_q16_small.json: 60 x d8 +4 %, 80 x d8 +5 %, 112 x d8 +7 %, 200 x d6 +9 %; round 3 with the _s2 walk,
                                             // profiles/archive/r03_sweep_fused_rank_experiment_ilp8_s2.json: 100 x d6 +17 %, 100 x d8 +11 %, while 30 x d6 still loses 15 %)

// a one-vs-all model whose classes hold equally many trees on this engine: their images can stand back to back (select_and_build)
bool classes_equal(const ddt_engine* e) {
  if (e->num_classes < 2) return false;
  for (const Ensemble& m : e->ens)
    if (m.trees() != e->ens[0].trees()) return false;
  return e->ens[0].trees() > 0;
}

// DDT_DISABLE_S2=1 in the environment: the AUTOMATIC choice skips the kernels that keep node records in SGPRs filled by inline-asm
// scalar loads ("_s2", opt bit 1 of the rank-quantised kernels).  A forced "variant" still takes them.  Without the variable the BUILD decides:
// ddt_build_s2_checked (ddt_checks.cpp) is 1 only when tools/check_s2_isa.py ran on this very binary and found nothing; a build whose check
// could not run (no disassembler on the build machine) switches them off by itself.  DDT_DISABLE_S2=0 is the explicit opt-in to unchecked kernels.
bool s2_disabled() {
  static const bool off = [] {
    const char* v = getenv("DDT_DISABLE_S2");
    if (v && v[0]) return v[0] != '0';
    return ddt_build_s2_checked == 0;
  }();
  return off;
}
// ... and the deep kernels ("q16d_*"), whose chunk barriers wait with a hand-counted `s_waitcnt vmcnt(N)` (ddt_deep.hip wait_for_dma): correct
// only for the instruction stream hipcc emitted, which tools/check_dma_waits.py verifies on the built binary (ddt_build_dma_checked).
// DDT_DISABLE_DEEP=1 / 0 overrides, like DDT_DISABLE_S2.
bool deep_disabled() {
  static const bool off = [] {
    const char* v = getenv("DDT_DISABLE_DEEP");
    if (v && v[0]) return v[0] != '0';
    return ddt_build_dma_checked == 0;
  }();
  return off;
}

uint32_t cm_position(uint32_t i, uint32_t T, uint32_t Cc);

bool variant_fits(const Variant& v, const ddt_engine* e) {
  if (v.kind == kKindSparse) return false;  // sparse forests pick their kernel in ddt_sparse_host.cpp
  if (v.kind == kKindGeneric) return true;
  if ((uint32_t)v.levels != e->p.num_levels) return false;
  uint32_t W = tuple_words(e->p);
  if (v.kind == kKindQ16) {
    W = q16_words(e);  // (feature compaction: the width the rank-quantised kernels see)
    // depth <= 8: two blocks per CU or it is not worth it; deeper trees have no other specialised kernel: one block
    if (W > v.max_tuple_words_q16() || v.lds_bytes_q16(W) > (((v.levels <= 8 || v.deep()) && !v.wide()) ? kMaxLdsBytes / 2u : kMaxLdsBytes)) return false;
    // deep kernels: their stage gathers address the image with 32-bit byte offsets through one buffer resource
    if (v.deep() && (uint64_t)padded_trees(v, max_trees(e)) * v.tree_bytes_q16() >= (1ull << 31)) return false;
    if ((v.opt & 4) && e->p.sum_mode == 1u) return false;  // cluster-major image order: not the stream order the fp64 sum is defined on
    if (rank_tables(e).max_len <= kQ16MaxTable) return true;
    // ... provided every PU group of 8 trees (the unit the parts are planned in: plan_q16_parts) stays within the u16 ranks by itself.  Up to
    // depth 12 it always does (8 x 4095 nodes); deeper trees on few features may not: counted per group and feature (nodes, an upper bound
    // of the distinct thresholds), in cluster-major order
    if (e->p.num_levels > 12u) {
      const uint32_t Cc = e->p.clusters_per_tuple ? e->p.clusters_per_tuple : 1u, nint = e->nint;
      for (const Ensemble& m : e->ens) {
        const uint32_t T = m.trees(), groups = (T + 7u) / 8u;
        std::vector<std::vector<uint32_t>> cnt(groups, std::vector<uint32_t>(W, 0u));
        for (uint32_t i = 0; i < T; ++i) {
          std::vector<uint32_t>& c = cnt[cm_position(i, T, Cc) / 8u];
          for (uint32_t n = 0; n < nint; ++n) ++c[q16_feat(e, m.fidx[(size_t)i * nint + n])];
        }
        for (const auto& c : cnt)
          for (uint32_t k : c)
            if (k > kQ16MaxTable) return false;
      }
    }
    // more distinct thresholds on a feature than u16 ranks hold: the plain cluster-major kernels score the ensemble in PARTS with
    // rank tables of their own (Q16Aux::state_in / state_out); one chunk of 8 trees never exceeds the limit
    // (the classes of a one-vs-all model are then scored one launch sequence per class, each class cut into parts of its own)
    return (v.opt & 4) && !(v.opt & 8);
  }
  if (v.kind == kKindStream)
    return W <= 4u * (uint32_t)v.opt && v.lds_bytes_stream(padded_trees(v, max_trees(e)), W) <= kStreamLdsBudget;
  if ((v.opt & 2) && W > 32u) return false;  // persistent form prefetches at most 8 lines per tuple
  return v.lds_bytes(W) <= kMaxLdsBytes;
}

int find_variant(const char* name) {
  for (int i = 0; i < num_variants(); ++i)
    if (!strcmp(variant(i).name, name)) return i;
  return -1;
}

int auto_variant(const ddt_engine* e) {
  // Preference order, first that fits wins; tuned from the sweeps under profiles/ (see DESIGN.md):
  // small ensembles that fit LDS whole -> streaming kernel (HBM-bound regime); otherwise the tile kernel with
  // the most waves per CU the feature tile allows; anything else -> generic.
  static const char* pref[] = {"stream_d4_u4_l4", "stream_d4_u4_l8", "stream_d6_u4_l4", "stream_d6_u4_l8", "stream_d8_u4_l8",
                               "stream_d7_u4_l8", "stream_d5_u4_l8", "stream_d3_u4_l8",
                               "d8_t1024_r1_c4_u4_dma_f", "d8_t512_r1_c8_u8_dma_f", "d8_t512_r1_c4_u4_dma_f", "d8_t256_r1_c4_u4_dma", "d8_t128_r1_c8_u8_dma", "d8_t64_r1_c8_u8_dma",
                               "d6_t1024_r1_c16_u4_dma", "d6_t512_r1_c16_u8_dma", "d6_t256_r1_c16_u4_dma", "d6_t128_r1_c16_u8_dma", "d6_t64_r1_c16_u8_dma",
                               "d4_t256_r1_c64_u8_dma", "d4_t128_r1_c64_u8_dma",
                               "d7_t1024_r1_c8_u4_dma", "d7_t256_r1_c8_u4_dma", "d7_t128_r1_c8_u8_dma",
                               "d5_t1024_r1_c32_u4_dma", "d5_t256_r1_c32_u4_dma", "d5_t128_r1_c32_u8_dma",
                               "d3_t256_r1_c128_u8_dma", "d3_t128_r1_c128_u8_dma"};
  // Rank-quantised path: its scoring kernel is ~1.3x faster per tree (32 waves/CU) but it pays a fixed transpose +
  // rank pre-pass per tuple.  Measured per 100 M tuples (profiles/archive/r01_*): q16 = 10.9 ms + 0.113 ms/tree, fp32 tile =
  // 3.2 ms + 0.147 ms/tree => break-even near 200 trees per engine; 250 trees (4-way shard of 1000) goes to q16.
  // With small tables (they all fit LDS together, e.g. a 125-tree shard) the pre-pass is one fused kernel and the
  // break-even drops accordingly (kQ16MinTreeLevels).
  // Perfect trees deeper than 8 levels (the reference's own example is 512 x depth 12, profiler/profiler.cpp:32-38; a depth-12 tree is exactly
  // one PU's memory, DTPU.sv:22-25): the deep rank-quantised kernels -- K = 8 / 9 levels out of LDS at two blocks per CU, the rest in
  // (D - K + 1) / 2 gathers of 16-byte records per tree.  Whatever the number of trees: the alternative is the generic kernel.
  if (e->p.num_levels > 8u && e->p.sum_mode != 1u && q16_words(e) <= 64u) {
    for (int i = 0; i < num_variants(); ++i)  // (table order: the two-blocks-per-CU forms first, then the wide ones for 33..64 words)
      if (variant(i).kind == kKindQ16 && variant(i).deep() && !deep_disabled() && variant_fits(variant(i), e)) return i;
  }
  // Tuples of 33..64 words, depth 8: the wide rank-quantised kernels (one block of 16 waves per CU, transpose + rank pre-pass) from the
  // same tree count on as the narrow ones -- 1000 x d8 x 64 / 48 / 33 features: 619 / 635 / 656 Mtuples/s against 432 / 533 / 535 on the fp32
  // tile kernels (profiles/r05_wide_and_deep_ab.md); below that tree count and beyond 64 words the fp32 tile kernels
  if (q16_words(e) > 32u && q16_words(e) <= 64u && total_trees(e) >= 224u) {
    static const char* wpref[] = {"q16w_d8_c8_u4_gl_s2_cm_x", "q16w_d8_c8_u4_gl"};  // (depth 8 only: at depth 6 the fp32 tile kernel is as fast)
    for (const char* name : wpref) {
      const int i = find_variant(name);
      if (i >= 0 && variant_fits(variant(i), e) && !((variant(i).opt & 2) && s2_disabled())) return i;
    }
  }
  uint32_t q16_min = 224u;
  if (q16_words(e) <= 32u && total_trees(e) * e->p.num_levels >= kQ16MinTreeLevels && total_trees(e) < 224u && prepass_plan_exists(e))
    q16_min = total_trees(e);
  if (total_trees(e) >= q16_min) {  // the pre-pass is shared by the classes of a multi-class model
    // (the cluster-major form only where there is a ring to save -- more than one cluster -- and the sum follows the reference's
    // order: the fp64 sum of sum_mode 1 runs in stream order, which a permuted image would change)
    // depth 8, reference-order sums (the fp64 sum of sum_mode 1 runs in stream order, which the cluster-major images would change):
    //   "_p"  persistent blocks -- a one-vs-all model whose classes hold equally many trees is walked in ONE launch, sums and labels
    //         written by the scoring kernel (10.57 vs 10.89 ms per 10 M tuples x 10 x 100 trees, and 11.37 before round 4).  For a plain
    //         ensemble on a GPU of its own the resident blocks buy nothing (12.92 vs 13.05 ms on a 125-tree shard, 94.8 vs 95.3 ms at
    //         1000 trees: profiles/r04_q16_pinned_persistent.md) -- but they take tiles from a ticket counter, so they do not wait for
    //         CUs that something else occupies: with 8 / 16 CUs of ONE XCD masked a shard's step takes 1.09x / 1.25x against 1.34x /
    //         1.97x for the plain launch, whose blocks the dispatcher deals round-robin over the XCDs (profiles/r04_cu_mask_probe.md).
    //         An engine inside a multi-rank job (RCCL's kernels on the same device) therefore takes it too;
    //   "_x"  the plain launch with the pinned LDS read order (four chains in flight per lane): +4.6 % over "_cm" at 1000 trees,
    //         +4 % on the shards; its single accumulator + running total also serves one cluster.
    if (e->p.sum_mode != 1u && !s2_disabled()) {
      const int ip = find_variant("q16_d8_c8_u4_gl_s2_cm_p");
      // (a one-vs-all model with UNEQUAL classes is one launch per class whatever the kernel: the plain launch then, also inside a job)
      if (ip >= 0 && variant_fits(variant(ip), e) && e->q16_persistent != 0 &&
          (e->q16_persistent == 1 || classes_equal(e) || (e->collective_job && e->num_classes == 1)))
        return ip;
      const int ix = find_variant("q16_d8_c8_u4_gl_s2_cm_x");
      if (ix >= 0 && variant_fits(variant(ix), e)) return ix;
    }
    if (e->p.clusters_per_tuple > 1u && e->p.sum_mode != 1u && !s2_disabled()) {
      const int i = find_variant("q16_d8_c8_u4_gl_s2_cm");
      if (i >= 0 && variant_fits(variant(i), e)) return i;
    }
    static const char* qpref[] = {"q16_d8_c8_u4_gl_s2", "q16_d8_c8_u4_gl", "q16_d8_c4_u4", "q16_d6_c16_u4_s2", "q16_d6_c16_u4", "q16_d4_c64_u8", "q16_d7_c8_u4_s2", "q16_d7_c8_u4", "q16_d5_c32_u4_s2", "q16_d5_c32_u4", "q16_d3_c128_u8"};
    for (const char* name : qpref) {
      const int i = find_variant(name);
      if (i >= 0 && variant_fits(variant(i), e) && !((variant(i).opt & 2) && s2_disabled())) return i;
    }
  }
  for (const char* name : pref) {
    const int i = find_variant(name);
    if (i >= 0 && variant_fits(variant(i), e)) return i;
  }
  return 0;
}

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
    e->q_rows[k] = 0;
    e->q_xT_valid[k] = false;
  }
}

// Device image of one ensemble for variant `v` (layouts: ddt_internal.h).  Host half -- no HIP call, also behind the test hook
// ddt_debug_model_image: the packed image for a tile / stream / generic variant; build_image uploads it.
int pack_image(ddt_engine* e, const Variant& v, const Ensemble& m, std::vector<uint32_t>& img, uint32_t* Tpad_out) {
  const uint32_t D = e->p.num_levels, T = m.trees(), nint = e->nint, nleaf = e->nleaf;
  const uint32_t tree_bytes = 12u << D;
  const uint32_t Tpad = padded_trees(v, T);  // EMPTY trees: every leaf +0 (DTPU.sv:544,760)
  const size_t bytes = (size_t)Tpad * tree_bytes;
  try {
    img.assign(bytes / 4, 0u);
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "image allocation (%zu bytes) failed", bytes);
  }
  // feature word of feature j: generic = j itself; tile/stream = absolute LDS byte address of row j
  const uint32_t row = v.row_bytes();
  const uint32_t feat_off = v.kind == kKindTile ? v.feat_off() : v.kind == kKindStream ? v.feat_off_stream(Tpad) : 0u;
  auto feature_word = [&](uint32_t j) {
    return v.kind == kKindGeneric ? j : v.kind == kKindStream ? v.feat_word_stream(Tpad, j) : feat_off + j * row;
  };
  const bool fused = v.kind == kKindTile && (v.opt & 1);
  const uint32_t first_last = 1u << (D - 1);  // 1-based index of the first last-level node
  for (uint32_t i = 0; i < Tpad; ++i) {
    uint32_t* t = img.data() + (size_t)i * (tree_bytes / 4);
    const bool empty = i >= T;  // EMPTY tree: zero thresholds and leaves; node words must still gather in range
    for (uint32_t n = 0; n < nint; ++n) {
      const uint32_t mm = n + 1;  // 1-based heap record
      const uint32_t j = empty ? 0u : m.fidx[(size_t)i * nint + n];
      const uint32_t word = feature_word(j) | ((!empty && m.mright[(size_t)i * nint + n]) ? kFlagMissRight : 0u);
      const uint32_t key = empty ? 0u : thr_key(e->p, m.thr[(size_t)i * nint + n]);
      if (fused && mm >= first_last) {  // layout 1: {thr, w2, leafL, leafR} at 4*2^D + 16*(m - 2^(D-1))
        const uint32_t r = mm - first_last;
        uint32_t* rec = t + (4u << D) / 4 + 4 * r;
        rec[0] = key;
        rec[1] = word;
        rec[2] = empty ? 0u : m.leaf[(size_t)i * nleaf + 2 * r];
        rec[3] = empty ? 0u : m.leaf[(size_t)i * nleaf + 2 * r + 1];
      } else {
        t[2 * mm + 0] = key;
        t[2 * mm + 1] = word;
      }
    }
    if (!fused && !empty) {
      uint32_t* lv = t + (8u << D) / 4;
      for (uint32_t l = 0; l < nleaf; ++l) lv[l] = m.leaf[(size_t)i * nleaf + l];
    }
  }
  *Tpad_out = Tpad;
  return DDT_OK;
}

int build_image(ddt_engine* e, const Variant& v, Ensemble& m) {
  std::vector<uint32_t> img;
  uint32_t Tpad = 0;
  const int rc = pack_image(e, v, m, img, &Tpad);
  if (rc) return rc;
  const size_t bytes = img.size() * 4u;
  if (m.d_img) (void)hipFree(m.d_img);
  m.d_img = nullptr;
  HIP_TRY(e, hipMalloc(&m.d_img, bytes));
  HIP_TRY(e, hipMemcpy(m.d_img, img.data(), bytes, hipMemcpyHostToDevice));
  m.img_bytes = bytes;
  m.img_trees = Tpad;
  m.img_chunks = v.kind == kKindTile ? Tpad / (uint32_t)v.chunk_trees : Tpad;
  return DDT_OK;
}

void finish_rank_tables(RankTables& rt) {
  rt.max_len = 0;
  for (auto& k : rt.keys) {
    std::sort(k.begin(), k.end(), [](uint32_t a, uint32_t b) { return (int32_t)a < (int32_t)b; });
    k.erase(std::unique(k.begin(), k.end()), k.end());
    if (k.size() > rt.max_len) rt.max_len = (uint32_t)k.size();
  }
}

// flat tables of rank_kernel ([W][Kpad] keys, per-feature search parameters, bucket starts) and -- want_prepass -- the LDS
// images of the LDS-resident pre-pass, from the sorted distinct threshold keys per feature
int pack_rank_tables(ddt_engine* e, const RankTables& rt, uint32_t W, bool want_prepass, RankHostTables& h) {
  uint32_t Kpad = 2;
  while (Kpad <= rt.max_len) Kpad <<= 1;  // power of two > max_len: the search reads indices < Kpad - 1
  std::vector<uint32_t>&tab = h.tab, &tabK = h.tabK, &pimg = h.pimg;
  std::vector<uint16_t>& tabS = h.tabS;
  PrepassPlan& pplan = h.pplan;
  try {
    tab.assign((size_t)W * Kpad, 0x7FFFFFFFu);
    tabK.assign((size_t)W * 8u, 0u);
    tabS.assign((size_t)W * kQ16RankBuckets, 0u);
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "rank table allocation failed");
  }
  for (uint32_t j = 0; j < W; ++j) {
    const std::vector<uint32_t>& k = rt.keys[j];
    const uint32_t K = (uint32_t)k.size();
    std::copy(k.begin(), k.end(), tab.begin() + (size_t)j * Kpad);
    // first level of the rank search (rank_kernel): slice the key range into kQ16RankBuckets equal pieces
    uint32_t* P = tabK.data() + (size_t)j * 8u;
    uint16_t* S = tabS.data() + (size_t)j * kQ16RankBuckets;
    P[0] = K;
    P[1] = P[2] = 0x7FFFFFFFu;  // unused feature: every x is "below lo" -> bucket 0 -> rank 0
    P[3] = 0u;
    P[4] = 1u;
    if (K) {
      const uint32_t lo = k.front(), hi = k.back(), span = hi - lo;  // int32 order: hi >= lo, the difference fits 32 bits
      uint32_t shift = 0;
      while ((span >> shift) >= kQ16RankBuckets) ++shift;
      std::vector<uint32_t> cnt(kQ16RankBuckets, 0u);
      for (uint32_t key : k) ++cnt[(key - lo) >> shift];
      uint32_t run = 0, max_len = 0;
      for (uint32_t b = 0; b < kQ16RankBuckets; ++b) {
        S[b] = (uint16_t)run;  // run <= K <= 32767
        run += cnt[b];
        max_len = cnt[b] > max_len ? cnt[b] : max_len;
      }
      uint32_t pow2 = 1;
      while (pow2 <= max_len) pow2 <<= 1;  // strictly more than the fullest slice
      P[1] = lo;
      P[2] = hi;
      P[3] = shift;
      P[4] = pow2;
    }
  }
  pplan = PrepassPlan{};
  if (want_prepass)
    (void)build_prepass_image(rt, W, (uint32_t)e->q16_prepass_groups, e->q16_fused_prepass != 0, e->q16_grouped_prepass != 0, &pimg, &pplan);
  if (want_prepass && getenv("DDT_DEBUG_PREPASS")) {
    fprintf(stderr, "[ddt] rank pre-pass: %u feature group(s) of %u line(s), longest table %u keys;", pplan.groups, pplan.lines, rt.max_len);
    for (uint32_t g = 0; g < pplan.groups; ++g) fprintf(stderr, " [P=%u, %u B]", pplan.P[g], pplan.bytes[g]);
    fprintf(stderr, "\n");
  }
  h.Kpad = Kpad;
  return DDT_OK;
}

void free_rank_device(RankDevice& d) {
  for (void** p : {&d.d_tables, &d.d_tabK, &d.d_tabS, &d.d_prepass}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  d.prepass = PrepassPlan{};
  d.Kpad = 0;
}

int upload_rank_tables(ddt_engine* e, const RankHostTables& h, RankDevice& d) {
  free_rank_device(d);
  HIP_TRY(e, hipMalloc(&d.d_tables, h.tab.size() * 4));
  HIP_TRY(e, hipMalloc(&d.d_tabK, h.tabK.size() * 4));
  HIP_TRY(e, hipMalloc(&d.d_tabS, h.tabS.size() * 2));
  HIP_TRY(e, hipMemcpy(d.d_tables, h.tab.data(), h.tab.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(d.d_tabK, h.tabK.data(), h.tabK.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(d.d_tabS, h.tabS.data(), h.tabS.size() * 2, hipMemcpyHostToDevice));
  if (h.pplan.groups && !h.pimg.empty()) {
    HIP_TRY(e, hipMalloc(&d.d_prepass, h.pimg.size() * 4));
    HIP_TRY(e, hipMemcpy(d.d_prepass, h.pimg.data(), h.pimg.size() * 4, hipMemcpyHostToDevice));
    d.prepass = h.pplan;
  }
  d.Kpad = h.Kpad;
  return DDT_OK;
}

// q16 images: per tree 2^D records {R (lo16) | row offset (hi16)} in a 1-based heap, then 2^D fp32 leaves.
// R = 1 + index of the node's threshold in its feature's table; the slow image carries miss_right in bit 16.
struct Q16HostImage {
  std::vector<uint32_t> fast, slow, tab, tabK, pimg;
  std::vector<uint16_t> tabS;
  PrepassPlan pplan{};
  uint32_t Tpad = 0, Kpad = 0;
  // ensembles scored in parts (more than kQ16MaxTable distinct thresholds on a feature): chunk ranges of the image and their tables
  std::vector<uint32_t> part_chunk_begin;  // [parts + 1]; empty = one part, tables above
  std::vector<RankTables> part_tables;
};

// position of tree i in a cluster-major ("_cm") image: the PU groups of cluster 0 (g % C == 0) first, in their order, then cluster 1's, ...
uint32_t cm_position(uint32_t i, uint32_t T, uint32_t Cc) {
  const uint32_t groups_real = (T + 7u) / 8u, g = i / 8u, c = g % Cc;
  uint32_t start = 0;  // groups of the clusters before c
  for (uint32_t k = 0; k < c; ++k) start += (groups_real + Cc - 1u - k) / Cc;
  return (start + g / Cc) * 8u + i % 8u;
}

// Cut a cluster-major image into parts whose distinct thresholds per feature fit the u16 ranks: chunks are taken in image order while
// every feature's key set stays within kQ16MaxTable (greedy; a chunk of 8 trees alone never exceeds it).
int plan_q16_parts(ddt_engine* e, const Variant& v, const Ensemble& m, Q16HostImage& h) {
  const uint32_t T = m.trees(), nint = e->nint, W = q16_words(e), CT = (uint32_t)v.chunk_trees;
  const uint32_t Cc = e->p.clusters_per_tuple ? e->p.clusters_per_tuple : 1u, Tpad = padded_trees(v, T), n_chunks = Tpad / CT;
  // (a part ends on a whole PU group -- the sum's state between two launches is {cluster accumulator, running total}, not a half group: the
  // deep kernels' chunks of 4 trees are taken in pairs)
  const uint32_t pc = CT < 8u ? 8u / CT : 1u;  // chunks per planning step
  std::vector<std::vector<uint32_t>> trees_of_chunk(n_chunks);
  for (uint32_t i = 0; i < T; ++i) trees_of_chunk[cm_position(i, T, Cc) / CT / pc * pc].push_back(i);
  try {
    h.part_chunk_begin.assign(1, 0u);
    h.part_tables.clear();
    RankTables cur;
    cur.keys.assign(W, {});
    auto merged_fits = [&](const std::vector<std::vector<uint32_t>>& add, RankTables* out) {
      RankTables t = cur;
      for (uint32_t j = 0; j < W; ++j) t.keys[j].insert(t.keys[j].end(), add[j].begin(), add[j].end());
      finish_rank_tables(t);
      if (t.max_len > kQ16MaxTable) return false;
      *out = std::move(t);
      return true;
    };
    for (uint32_t c = 0; c < n_chunks; c += pc) {
      std::vector<std::vector<uint32_t>> add(W);
      for (uint32_t i : trees_of_chunk[c])
        for (uint32_t n = 0; n < nint; ++n) add[q16_feat(e, m.fidx[(size_t)i * nint + n])].push_back(thr_key(e->p, m.thr[(size_t)i * nint + n]));
      RankTables next;
      if (merged_fits(add, &next)) {
        cur = std::move(next);
        continue;
      }
      h.part_tables.push_back(cur);  // close the part in front of chunk c
      h.part_chunk_begin.push_back(c);
      cur = RankTables{};
      cur.keys.assign(W, {});
      if (!merged_fits(add, &next)) return fail(e, DDT_EUNSUPPORTED, "one PU group of trees has more than %u distinct thresholds on a feature", kQ16MaxTable);
      cur = std::move(next);
    }
    h.part_tables.push_back(cur);
    h.part_chunk_begin.push_back(n_chunks);
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "rank table allocation failed");
  }
  return DDT_OK;
}

// host half of build_image_q16 (no HIP call; also behind the test hook ddt_debug_model_image)
int pack_image_q16(ddt_engine* e, const Variant& v, const Ensemble& m, const RankTables& rt, bool upload_tables, Q16HostImage& h) {
  const uint32_t D = e->p.num_levels, T = m.trees(), nint = e->nint, nleaf = e->nleaf, W = q16_words(e);
  const uint32_t tree_words = v.tree_bytes_q16() / 4u, Tpad = padded_trees(v, T);
  std::vector<uint32_t>&fast = h.fast, &slow = h.slow;
  try {
    fast.assign((size_t)Tpad * tree_words, 0u);
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "q16 image allocation failed");
  }
  const bool in_parts = rt.max_len > kQ16MaxTable;  // (variant_fits has checked that this kernel can score in parts)
  if (in_parts) {
    const int rc = plan_q16_parts(e, v, m, h);
    if (rc) return rc;
  } else {
    RankHostTables rk;
    const int rc = pack_rank_tables(e, rt, W, upload_tables && e->fmap.empty(), rk);
    if (rc) return rc;
    h.tab.swap(rk.tab);
    h.tabK.swap(rk.tabK);
    h.tabS.swap(rk.tabS);
    h.pimg.swap(rk.pimg);
    h.pplan = rk.pplan;
    h.Kpad = rk.Kpad;
  }
  const uint32_t row = v.wide() ? v.tile() : v.tile() * 2u;  // what a record's row-offset field counts in: bytes of a feature row of the u16 tile (wide: half of it)
  // word offsets of tree i's records and leaves: tree by tree (records, then leaves), or -- "_gl" variants -- per chunk the
  // records of its CT trees followed by the leaves of its CT trees (only the first half of a chunk is staged in LDS)
  const uint32_t CT = (uint32_t)v.chunk_trees, half = 1u << D;
  const bool gl = (v.opt & 1) != 0;
  auto rec_off = [&](uint32_t i) { return gl ? (size_t)(i / CT) * CT * tree_words + (size_t)(i % CT) * half : (size_t)i * tree_words; };
  auto leaf_off = [&](uint32_t i) { return gl ? (size_t)(i / CT) * CT * tree_words + (size_t)CT * half + (size_t)(i % CT) * half : (size_t)i * tree_words + half; };
  // "_cm" variants (opt bit 2): cluster-major image order -- the PU groups of cluster 0 (g % C == 0) first, in their order, then
  // cluster 1's, ...; padding groups stay behind the last real one.  Tree i sits at image position cm_pos(i).
  const bool cm = (v.opt & 4) != 0;
  const uint32_t Cc = e->p.clusters_per_tuple ? e->p.clusters_per_tuple : 1u;
  auto cm_pos = [&](uint32_t i) -> uint32_t { return cm ? cm_position(i, T, Cc) : i; };
  // the tables a tree's thresholds are ranked against: the ensemble's, or those of the part its chunk belongs to
  auto tables_of = [&](uint32_t pos) -> const RankTables& {
    if (!in_parts) return rt;
    const uint32_t c = pos / CT;
    size_t part = 0;
    while (h.part_chunk_begin[part + 1] <= c) ++part;
    return h.part_tables[part];
  };
  if (v.deep()) {
    // deep kernels (ddt_internal.h "deep rank-quantised kernels"): per chunk the tops of its CT trees, then their stage blocks
    const uint32_t K = (uint32_t)v.top, topw = (4u << K) / 4u, deepw = v.deep_bytes() / 4u, G = v.deep_stages();
    auto top_off = [&](uint32_t pos) { return (size_t)(pos / CT) * CT * tree_words + (size_t)(pos % CT) * topw; };
    auto deep_off = [&](uint32_t pos) { return (size_t)(pos / CT) * CT * tree_words + (size_t)CT * topw + (size_t)(pos % CT) * deepw; };
    slow = fast;
    for (uint32_t i = 0; i < T; ++i) {
      const uint32_t pos = cm_pos(i);
      const RankTables& trt = tables_of(pos);
      auto record = [&](uint32_t n, bool with_flag) -> uint32_t {  // node n of tree i (0-based heap)
        const uint32_t j = q16_feat(e, m.fidx[(size_t)i * nint + n]), key = thr_key(e->p, m.thr[(size_t)i * nint + n]);
        const auto& k = trt.keys[j];
        const uint32_t idx = (uint32_t)(std::lower_bound(k.begin(), k.end(), key, [](uint32_t a, uint32_t b) { return (int32_t)a < (int32_t)b; }) - k.begin());
        return (idx + 1u) | ((j * row) << 16) | ((with_flag && m.mright[(size_t)i * nint + n]) ? 1u << 16 : 0u);
      };
      for (int sl = 0; sl < 2; ++sl) {
        std::vector<uint32_t>& im = sl ? slow : fast;
        uint32_t* t = im.data() + top_off(pos);
        for (uint32_t n = 0; n + 1u < (1u << K); ++n) t[n + 1] = record(n, sl != 0);
        for (uint32_t g = 0; g < G; ++g) {
          const uint32_t L = v.deep_stage_level(g), first = (1u << L) - 1u;  // first node of level L, 0-based heap
          uint32_t* st = im.data() + deep_off(pos) + v.deep_stage_off(g) / 4u;
          for (uint32_t q = 0; q < (1u << L); ++q) {
            const uint32_t n = first + q;
            st[4u * q + 0u] = record(n, sl != 0);
            if (g + 1u < G) {  // pair: the node, its two children, the byte offset of its first grandchild's record in the next stage
              st[4u * q + 1u] = record(2u * n + 1u, sl != 0);
              st[4u * q + 2u] = record(2u * n + 2u, sl != 0);
              st[4u * q + 3u] = 64u * q;
            } else {  // terminal: level D-1 with its two leaves
              st[4u * q + 1u] = m.leaf[(size_t)i * nleaf + 2u * q];
              st[4u * q + 2u] = m.leaf[(size_t)i * nleaf + 2u * q + 1u];
              st[4u * q + 3u] = 0u;
            }
          }
        }
      }
    }
    h.Tpad = Tpad;
    return DDT_OK;
  }
  for (uint32_t i = 0; i < T; ++i) {
    uint32_t* t = fast.data() + rec_off(cm_pos(i));
    const RankTables& trt = tables_of(cm_pos(i));
    for (uint32_t n = 0; n < nint; ++n) {
      const uint32_t j = q16_feat(e, m.fidx[(size_t)i * nint + n]), key = thr_key(e->p, m.thr[(size_t)i * nint + n]);
      const auto& k = trt.keys[j];
      const uint32_t idx = (uint32_t)(std::lower_bound(k.begin(), k.end(), key, [](uint32_t a, uint32_t b) { return (int32_t)a < (int32_t)b; }) - k.begin());
      t[n + 1] = (idx + 1u) | ((j * row) << 16);
    }
    uint32_t* lf = fast.data() + leaf_off(cm_pos(i));
    for (uint32_t l = 0; l < nleaf; ++l) lf[l] = m.leaf[(size_t)i * nleaf + l];
  }
  slow = fast;
  for (uint32_t i = 0; i < T; ++i)
    for (uint32_t n = 0; n < nint; ++n)
      if (m.mright[(size_t)i * nint + n]) slow[rec_off(cm_pos(i)) + n + 1] |= 1u << 16;
  h.Tpad = Tpad;
  return DDT_OK;
}

int build_image_q16(ddt_engine* e, const Variant& v, Ensemble& m, const RankTables& rt, bool upload_tables) {
  Q16HostImage h;
  const int rc = pack_image_q16(e, v, m, rt, upload_tables, h);
  if (rc) return rc;
  const std::vector<uint32_t>&fast = h.fast, &slow = h.slow, &tab = h.tab, &tabK = h.tabK, &pimg = h.pimg;
  const std::vector<uint16_t>& tabS = h.tabS;
  const PrepassPlan& pplan = h.pplan;
  const uint32_t Tpad = h.Tpad, Kpad = h.Kpad;
  for (void** p : {&m.d_img, &m.d_img_slow, &m.d_tables, &m.d_tabK, &m.d_tabS, &m.d_prepass}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  for (Q16Part& part : m.parts) free_rank_device(part.rank);
  m.parts.clear();
  const size_t bytes = fast.size() * 4;
  HIP_TRY(e, hipMalloc(&m.d_img, bytes));
  HIP_TRY(e, hipMalloc(&m.d_img_slow, bytes));
  if (!h.part_tables.empty()) {  // scored in parts: every part brings its own tables (+ LDS images of its pre-pass)
    HIP_TRY(e, hipMemcpy(m.d_img, fast.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(m.d_img_slow, slow.data(), bytes, hipMemcpyHostToDevice));
    m.parts.resize(h.part_tables.size());
    for (size_t k = 0; k < m.parts.size(); ++k) {
      RankHostTables rk;
      int rc2 = pack_rank_tables(e, h.part_tables[k], q16_words(e), e->fmap.empty(), rk);
      if (!rc2) rc2 = upload_rank_tables(e, rk, m.parts[k].rank);
      if (rc2) return rc2;
      m.parts[k].chunk_begin = h.part_chunk_begin[k];
      m.parts[k].chunks = h.part_chunk_begin[k + 1] - h.part_chunk_begin[k];
    }
    m.prepass = PrepassPlan{};
    m.img_bytes = bytes;
    m.img_trees = Tpad;
    m.img_chunks = Tpad / (uint32_t)v.chunk_trees;
    m.Kpad = 0;
    if (getenv("DDT_DEBUG_PREPASS")) fprintf(stderr, "[ddt] the ensemble is scored in %zu parts (rank tables of their own)\n", m.parts.size());
    return DDT_OK;
  }
  if (upload_tables) {
    HIP_TRY(e, hipMalloc(&m.d_tables, tab.size() * 4));
    HIP_TRY(e, hipMalloc(&m.d_tabK, tabK.size() * 4));
    HIP_TRY(e, hipMalloc(&m.d_tabS, tabS.size() * 2));
  }
  HIP_TRY(e, hipMemcpy(m.d_img, fast.data(), bytes, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(m.d_img_slow, slow.data(), bytes, hipMemcpyHostToDevice));
  if (upload_tables) {
    HIP_TRY(e, hipMemcpy(m.d_tables, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(m.d_tabK, tabK.data(), tabK.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(m.d_tabS, tabS.data(), tabS.size() * 2, hipMemcpyHostToDevice));
    m.prepass = PrepassPlan{};
    if (pplan.groups && !pimg.empty()) {
      HIP_TRY(e, hipMalloc(&m.d_prepass, pimg.size() * 4));
      HIP_TRY(e, hipMemcpy(m.d_prepass, pimg.data(), pimg.size() * 4, hipMemcpyHostToDevice));
      m.prepass = pplan;
    }
  }
  m.img_bytes = bytes;
  m.img_trees = Tpad;
  m.img_chunks = Tpad / (uint32_t)v.chunk_trees;
  m.Kpad = Kpad;
  return DDT_OK;
}

// grow-only workspace of the q16 pre-pass (synchronous allocation on first use / growth)
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

// the features the loaded trees test, when compaction applies: more than 64 tuple words, at most 64 of them used (option "feature_compaction")
void plan_feature_compaction(ddt_engine* e) {
  e->fmap.clear();
  e->finv.clear();
  if (e->d_fmap) (void)hipFree(e->d_fmap);
  e->d_fmap = nullptr;
  const uint32_t W = tuple_words(e->p);
  if (!e->feature_compaction || W <= 64u) return;
  std::vector<uint8_t> used(W, 0);
  for (const Ensemble& m : e->ens)
    for (uint16_t j : m.fidx) used[j] = 1;
  std::vector<uint16_t> fmap;
  for (uint32_t j = 0; j < W; ++j)
    if (used[j]) fmap.push_back((uint16_t)j);
  if (fmap.empty() || fmap.size() > 64u) return;
  e->finv.assign(W, 0);
  for (size_t c = 0; c < fmap.size(); ++c) e->finv[fmap[c]] = (uint16_t)c;
  e->fmap.swap(fmap);
}

int select_and_build(ddt_engine* e) {
  plan_feature_compaction(e);
  int vid = e->forced_variant;
  if (vid >= 0) {
    if (vid >= num_variants()) return fail(e, DDT_EINVAL, "variant %d out of range", vid);
    if (!variant_fits(variant(vid), e))
      return fail(e, DDT_EUNSUPPORTED, "variant %s does not fit this model (D=%u, F=%u)", variant(vid).name,
                  e->p.num_levels, e->p.num_features);
  } else {
    vid = auto_variant(e);
  }
  if (variant(vid).kind != kKindQ16) {  // only the rank-quantised kernels read compacted tuples
    e->fmap.clear();
    e->finv.clear();
  } else if (!e->fmap.empty()) {  // the column map of the gathering transpose: one word per compacted tuple word, ~0 = padding
    std::vector<uint32_t> cols(q16_words(e), 0xFFFFFFFFu);
    for (size_t c = 0; c < e->fmap.size(); ++c) cols[c] = e->fmap[c];
    HIP_TRY(e, hipMalloc(&e->d_fmap, cols.size() * 4u));
    HIP_TRY(e, hipMemcpy(e->d_fmap, cols.data(), cols.size() * 4u, hipMemcpyHostToDevice));
  }
  RankTables rt;
  if (variant(vid).kind == kKindQ16) rt = rank_tables(e);
  for (Ensemble& m : e->ens) {
    int rc = variant(vid).kind == kKindQ16 ? build_image_q16(e, variant(vid), m, rt, &m == &e->ens[0])  // tables live in ens[0]
                                          : build_image(e, variant(vid), m);
    if (rc) return rc;
  }
  // "_p" kernels walk every class of a one-vs-all model in ONE launch when the classes' images can stand back to back: equally
  // many trees per class (=> equally many chunks and real PU groups).  Otherwise: one launch per class, as with every other kernel.
  for (void** p : {&e->d_mc_img, &e->d_mc_img_slow}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  e->mc_seg_chunks = 0;
  if (variant(vid).kind == kKindQ16 && (variant(vid).opt & 8) && e->num_classes > 1) {
    bool same = true;
    for (const Ensemble& m : e->ens) same = same && m.trees() == e->ens[0].trees() && m.img_bytes == e->ens[0].img_bytes;
    if (same && e->ens[0].img_bytes) {
      const size_t b = e->ens[0].img_bytes;
      HIP_TRY(e, hipMalloc(&e->d_mc_img, b * e->num_classes));
      HIP_TRY(e, hipMalloc(&e->d_mc_img_slow, b * e->num_classes));
      for (uint32_t k = 0; k < e->num_classes; ++k) {
        HIP_TRY(e, hipMemcpy(static_cast<char*>(e->d_mc_img) + k * b, e->ens[k].d_img, b, hipMemcpyDeviceToDevice));
        HIP_TRY(e, hipMemcpy(static_cast<char*>(e->d_mc_img_slow) + k * b, e->ens[k].d_img_slow, b, hipMemcpyDeviceToDevice));
      }
      e->mc_seg_chunks = e->ens[0].img_chunks;
    }
  }
  e->variant_id = vid;
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
      const uint32_t sgs_before = part.chunk_begin * (uint32_t)v.chunk_trees / (uint32_t)v.ilp_trees;
      qa.walk_subgroups = (walk_all > sgs_before) ? walk_all - sgs_before : 0u;  // (0 = everything: only the last part has padding)
      if (k + 1 < m.parts.size()) qa.walk_subgroups = 0u;
      if (k > 0) a.ev_mid = nullptr;  // (kernel_timing: the first part's pre-pass against everything behind it)
      r = v.launch(a, v, s);
      groups_before += a.n_trees / 8u;
      e->st.kernel_launches++;
    }
    e->q_xT_valid[e->q_slot] = xT_valid && r == hipSuccess;
    if (r == hipSuccess) e->st.kernel_launches--;  // (counted once more below)
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

// A perfect-tree model for which the automatic choice found no tuned kernel -- depth >= 9 with more than 64 tuple words TESTED (feature
// compaction above takes the others), depth 16 -- landed on `generic`, which gathers every feature of every visit from global memory
// (512 x depth 12 x 200 features: 17 Mtuples/s).  A perfect tree IS a sparse tree whose leaves all sit at depth D: such a model is handed to the
// sparse-forest path (top levels out of LDS, a feature tile of 64..256 tuples, 16-byte records below; ddt_sparse_host.cpp) -- same node
// semantics (DTPU.sv:579-720), same adder order, same EMPTY slots.  Option "generic_via_sparse" = 0 keeps `generic` (A/B, tests).
int maybe_score_as_sparse(ddt_engine* e) {
  e->perfect_as_sparse = false;
  if (!e->generic_via_sparse || e->forced_variant >= 0 || variant(e->variant_id).kind != kKindGeneric) return DDT_OK;
  const uint32_t D = e->p.num_levels, nint = e->nint, first_last = (1u << (D - 1u)) - 1u;
  // Measured on one MI355X, 4 M tuples, Mtuples/s on the sparse path against `generic` (profiles/r06_generic_cliffs.md): 512 x d12 x 200 features
  // 118.6 vs 16.0, x 100 features 285 vs 41, 256 x d9 x 400 144 vs 30, 64 x d15 x 200 482 vs 90; with the fp64 sum 512 x d12 x 32 528 vs 225, x 64
  // 446 vs 248, 256 x d10 x 32 1150 vs 579; 64 x d15 x 4 (PU groups beyond u16 ranks) 2074 vs 1138; 512 x d16 x 64 59 vs 55 -- and 512 x d16 x 32
  // 69 vs 97: at depth 16 with at most 32 tuple words `generic` (features in LDS, every walker alive to the last level either way) stays
  if (D >= 16u && tuple_words(e->p) <= 32u) return DDT_OK;
  if ((uint64_t)total_trees(e) * nint * 16ull > (3ull << 29)) return DDT_OK;  // (1.5 GiB of node lines: stay where we are)
  std::vector<SparseForest> sps(e->ens.size());
  try {
    for (size_t k = 0; k < e->ens.size(); ++k) {
      const Ensemble& m = e->ens[k];
      SparseForest& sp = sps[k];
      sp.ids = m.ids;
      sp.max_depth = D;
      sp.first.assign(1, 0u);
      sp.lines.resize((size_t)m.trees() * nint * 4u);
      for (uint32_t i = 0; i < m.trees(); ++i) {
        uint32_t* L = sp.lines.data() + (size_t)i * nint * 4u;
        for (uint32_t n = 0; n < nint; ++n) {  // 0-based heap: children 2n + 1, 2n + 2; the last level's children are the leaves
          const bool last = n >= first_last;
          L[4u * n + 0u] = m.thr[(size_t)i * nint + n];
          L[4u * n + 1u] = (uint32_t)m.fidx[(size_t)i * nint + n] | (m.mright[(size_t)i * nint + n] ? 1u << 13 : 0u) | (last ? 3u << 14 : 0u);
          L[4u * n + 2u] = last ? m.leaf[(size_t)i * e->nleaf + 2u * (n - first_last)] : 2u * n + 1u;
          L[4u * n + 3u] = last ? m.leaf[(size_t)i * e->nleaf + 2u * (n - first_last) + 1u] : 2u * n + 2u;
        }
        sp.first.push_back(sp.first.back() + nint);
      }
    }
  } catch (const std::bad_alloc&) {
    return DDT_OK;  // no memory for the second form: `generic` it is
  }
  const int generic_id = e->variant_id;
  e->sps = std::move(sps);
  e->sparse = true;
  const int rc = sparse_rebuild(e);
  if (rc != DDT_OK || (variant(e->variant_id).opt & 4)) {  // nothing fits, or only the sparse format's own correctness kernel: no gain
    sparse_free(e);
    e->sps.clear();
    e->sparse = false;
    e->variant_id = generic_id;
    e->err[0] = 0;
    return DDT_OK;
  }
  free_images(e);  // the generic image; the parsed trees (e->ens) stay for a later re-pack
  e->perfect_as_sparse = true;
  return DDT_OK;
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

int64_t ddt_debug_prepass_image(const uint32_t* keys, const uint32_t* counts, uint32_t n_words, uint32_t groups, uint32_t* image_out,
                                size_t image_cap_words, uint32_t plan_out[42]) {
  if (!keys || !counts || !plan_out || n_words == 0 || n_words > 32u || (n_words & 3u)) return DDT_EINVAL;
  if (groups != 0 && groups != 1 && groups != 2 && groups != 4 && groups != 8) return DDT_EINVAL;
  RankTables rt;
  rt.keys.resize(n_words);
  size_t off = 0;
  for (uint32_t w = 0; w < n_words; ++w) {
    if (counts[w] > kQ16MaxTable) return DDT_EUNSUPPORTED;
    rt.keys[w].assign(keys + off, keys + off + counts[w]);
    for (uint32_t i = 1; i < counts[w]; ++i)
      if (!((int32_t)rt.keys[w][i - 1] < (int32_t)rt.keys[w][i])) return DDT_EINVAL;  // sorted, distinct
    off += counts[w];
    rt.max_len = counts[w] > rt.max_len ? counts[w] : rt.max_len;
  }
  std::vector<uint32_t> img;
  PrepassPlan pl{};
  memset(plan_out, 0, 42 * sizeof(uint32_t));
  if (!build_prepass_image(rt, n_words, groups, true, true, &img, &pl)) return 0;
  plan_out[0] = pl.groups;
  plan_out[1] = pl.lines;
  for (uint32_t g = 0; g < pl.groups; ++g) {
    uint32_t* o = plan_out + 2 + 5 * g;
    o[0] = pl.img_off[g], o[1] = pl.bytes[g], o[2] = pl.par_off[g], o[3] = pl.P[g], o[4] = pl.line_lo[g];
  }
  if (image_out) {
    if (image_cap_words < img.size()) return DDT_EINVAL;
    memcpy(image_out, img.data(), img.size() * 4);
  }
  return (int64_t)img.size();
}

// Host-only test hook (include/ddt.h): parse + pack a perfect-tree model for kernel variant `variant_id` (-1: the engine's
// choice) exactly as ddt_load_model would, without touching a GPU.
int ddt_debug_model_image(const ddt_params* p, const void* wl, size_t n_wlines, const void* fl, size_t n_flines, int variant_id,
                          uint32_t* img_out, uint32_t* img_slow_out, size_t img_cap_words, uint32_t* tables_out, size_t tables_cap_words,
                          uint64_t info_out[12]) {
  if (!p || !wl || !fl || !info_out) return DDT_EINVAL;
  std::unique_ptr<ddt_engine> e(new (std::nothrow) ddt_engine());  // never created on a device: p, options, ens and err are used
  if (!e) return DDT_ENOMEM;
  int rc = validate(e.get(), p, n_wlines, n_flines);
  if (rc) return rc;
  e->p = *p;
  e->nint = (1u << p->num_levels) - 1u;
  e->nleaf = 1u << p->num_levels;
  std::vector<uint32_t> ids(p->num_trees);
  for (uint32_t i = 0; i < p->num_trees; ++i) ids[i] = i;
  e->ens.resize(1);
  rc = parse_trees(e.get(), p, reinterpret_cast<const uint32_t*>(wl), reinterpret_cast<const uint16_t*>(fl), std::move(ids), &e->ens[0]);
  if (rc) return rc;
  const int vid = variant_id < 0 ? auto_variant(e.get()) : variant_id;
  if (vid >= num_variants() || !variant_fits(variant(vid), e.get())) return DDT_EUNSUPPORTED;
  const Variant& v = variant(vid);
  const uint32_t W = tuple_words(e->p);
  std::vector<uint32_t> img;
  Q16HostImage h;
  uint32_t Tpad = 0;
  if (v.kind == kKindQ16) {
    rc = pack_image_q16(e.get(), v, e->ens[0], rank_tables(e.get()), true, h);
    Tpad = h.Tpad;
    if (!rc && !h.part_tables.empty()) return DDT_EUNSUPPORTED;  // an ensemble scored in parts has one table set per part: not exposed through this hook
  } else {
    rc = pack_image(e.get(), v, e->ens[0], img, &Tpad);
  }
  if (rc) return rc;
  const std::vector<uint32_t>& out = v.kind == kKindQ16 ? h.fast : img;
  memset(info_out, 0, 12 * sizeof(uint64_t));
  info_out[0] = out.size();
  info_out[1] = Tpad;
  info_out[2] = (uint64_t)v.kind;
  info_out[3] = (uint64_t)v.opt;
  info_out[4] = (uint64_t)v.chunk_trees;
  info_out[5] = v.tile();
  info_out[6] = v.kind == kKindTile ? v.feat_off() : v.kind == kKindStream ? v.feat_off_stream(Tpad) : v.kind == kKindQ16 ? v.feat_off_q16() : 0u;
  info_out[7] = v.kind == kKindQ16 ? v.tile() * 2u : v.kind == kKindGeneric ? 0u : v.row_bytes();
  info_out[8] = h.Kpad;
  info_out[9] = W;
  info_out[10] = (uint64_t)vid;
  info_out[11] = h.tab.size();
  if (img_out) {
    if (img_cap_words < out.size()) return DDT_EINVAL;
    memcpy(img_out, out.data(), out.size() * 4u);
  }
  if (img_slow_out && v.kind == kKindQ16) {
    if (img_cap_words < h.slow.size()) return DDT_EINVAL;
    memcpy(img_slow_out, h.slow.data(), h.slow.size() * 4u);
  }
  if (tables_out && v.kind == kKindQ16) {
    if (tables_cap_words < h.tab.size()) return DDT_EINVAL;
    memcpy(tables_out, h.tab.data(), h.tab.size() * 4u);
  }
  return DDT_OK;
}

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
