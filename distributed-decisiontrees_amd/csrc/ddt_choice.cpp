// ddt_choice.cpp -- which kernel scores a loaded perfect-tree model (split out of ddt_engine.cpp in round 6): what fits (variant_fits), the
// automatic preference (auto_variant; measurements under profiles/), feature compaction for wide models that test few features, and the
// hand-over of models without a tuned kernel to the sparse-forest path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "ddt_engine_priv.h"

extern "C" {
extern const int ddt_build_s2_checked, ddt_build_dma_checked;  // ddt_checks.cpp
}

namespace ddt {

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
    if (rank_tables(e).max_len <= e->q16_max_table) return true;
    // ... provided every PU group of 8 trees (the unit the parts are planned in: plan_q16_parts) stays within the u16 ranks by itself.  Up to
    // depth 12 it always does (8 x 4095 nodes); deeper trees on few features may not: counted per group and feature (nodes, an upper bound
    // of the distinct thresholds), in cluster-major order
    if (e->p.num_levels > 12u || e->q16_max_table < 8u * 4095u) {  // (or a limit lowered through the option "q16_max_table")
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
            if (k > e->q16_max_table) return false;
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
    static const char* qpref[] = {"q16_d8_c8_u4_gl_s2", "q16_d8_c8_u4_gl", "q16_d6_c16_u4_s2", "q16_d6_c16_u4", "q16_d4_c64_u8", "q16_d7_c8_u4_s2", "q16_d7_c8_u4", "q16_d5_c32_u4_s2", "q16_d5_c32_u4", "q16_d3_c128_u8"};
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

}  // namespace ddt
