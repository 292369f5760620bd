// ddt_image.cpp -- what a loaded model looks like on the device (split out of ddt_engine.cpp in round 6): the packed images of the
// perfect-tree kernels (fp32 tile / stream / generic layouts, the rank-quantised layouts incl. the deep stage records, cluster-major order,
// ensembles in parts), the rank tables and the LDS images of the rank pre-pass, and the host-only test hooks that hand them out.
// Layouts: ddt_internal.h; model store of the reference: rtl/DTEngine/core/DTPU.sv:282-354.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "ddt_engine_priv.h"

using namespace ddt;

namespace ddt {

// ---- feature compaction (round 6) --------------------------------------------------------------------------------
// The rank-quantised kernels take tuples of at most 64 words (the u16 tile of 1024 tuples must fit LDS); the reference takes F <= 2048
// (DTPU.sv:22-25,628).  A model of more than 64 tuple words that TESTS at most 64 distinct features (ddt_engine::fmap: compact index ->
// feature number) still runs on them: the rank pre-pass gathers only those columns into its transposed intermediate, and tables, tiles,
// node records and kernels see a tuple of q16_words() words.  Everything else (the wire format, the feeder, ddt_info) keeps the caller's width.
uint32_t q16_words(const ddt_engine* e) { return e->fmap.empty() ? tuple_words(e->p) : (uint32_t)((e->fmap.size() + 3u) / 4u * 4u); }

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
//   per feature  a table of K + P keys (INT_MAX pads; linear since round 6: entry i at word i)
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
      words += len;  // (round 6: linear tables -- the probes start at a bucket's own first key, so no power-of-two stride lines them up on one bank)
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
      words += len;  // (round 6: linear tables -- the probes start at a bucket's own first key, so no power-of-two stride lines them up on one bank)
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
      for (uint32_t i = 0; i < k.size(); ++i) (*img)[tab_off[j] / 4u + i] = k[i];
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
          S[first + b] = (uint16_t)run;  // run <= K <= kQ16MaxTable
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
  const uint32_t Kpad = q16_table_pad(rt.max_len);  // > max_len (a power of two up to 32767 keys): the search reads indices < Kpad - 1
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
        S[b] = (uint16_t)run;  // run <= K <= kQ16MaxTable
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
      if (t.max_len > e->q16_max_table) return false;
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
      if (!merged_fits(add, &next)) return fail(e, DDT_EUNSUPPORTED, "one PU group of trees has more than %u distinct thresholds on a feature", e->q16_max_table);
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
  const bool in_parts = rt.max_len > e->q16_max_table;  // (variant_fits has checked that this kernel can score in parts)
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
}  // namespace ddt

extern "C" {

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

}  // extern "C"
