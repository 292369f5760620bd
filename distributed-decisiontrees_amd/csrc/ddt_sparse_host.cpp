// ddt_sparse_host.cpp -- host side of the SPARSE (explicit-children) forest path: stream validation, tree sharding,
// packing of the device images (top-K heap per PU group + deep records) and the launch.  Format: include/ddt.h
// (ddt_load_model_sparse); device layout: ddt_internal.h "Sparse forests"; kernel: ddt_sparse.hip.
//
// Reference anchors: the perfect-tree engine this extends is rtl/DTEngine/core/DTPU.sv:579-760; its capacity limit and
// the (disabled) hook for trees that exceed it are DTPU.sv:20-28,736-745 and Core.sv:380 bit 8; entry bit layout
// DTPU.sv:628,637,659-661; contiguous per-device tree shards PCIeReceiver.sv:241-264; EMPTY slots DTPU.sv:544,760.
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <memory>
#include <new>

#include "ddt_engine_priv.h"

namespace ddt {

namespace {

struct Cursor {  // a position while expanding the top heap: an internal node of the tree, or a leaf value
  bool leaf;
  uint32_t v;    // node index / leaf bits
};

// Kernel choice for a sparse forest.  A forced id (option "variant") wins when it is a sparse kernel that fits; option
// "sparse_top_levels" fixes K and takes the widest tile that fits; otherwise the preference list below -- measured on
// BASELINE config 4 (512 trees x depth 16 x 64 features, profiles/archive/r02_sparse_sweep*.json): the deep phase is bound by the
// vector-memory pipe's lane-address rate, and hiding its latency needs >= 8 waves per CU, so a smaller K that lets two
// 256-tuple blocks share a CU beats a larger K with one.
int pick_variant(ddt_engine* e, uint32_t max_depth, bool ranks_fit) {
  const uint32_t W = tuple_words(e->p);
  auto fits = [&](int vid, uint32_t budget) {
    if (vid < 0 || variant(vid).kind != kKindSparse || variant(vid).lds_bytes_sparse(W) > budget) return false;
    if ((variant(vid).opt & 16) && e->p.num_features > 256u) return false;  // pair records carry feature numbers as bytes
    return !(variant(vid).opt & 1) || ranks_fit;  // rank-quantised kernels: every feature's table must fit 16-bit ranks
  };
  if (e->forced_variant >= 0) return fits(e->forced_variant, kMaxLdsBytes) ? e->forced_variant : -1;
  char name[40];
  // Rank-quantised kernels first (option "sparse_q16", default on): a u16 feature tile is half the LDS per tuple, so one block of
  // 1024 tuples = 16 waves holds the CU that the fp32 tile fills with 512 = 8; the deep phase is latency-bound and takes the
  // walkers (BASELINE config 4: profiles/archive/r03_sweep_sparse_q.json).  Largest K whose top images fit next to the tile.
  if (e->sparse_q16 && ranks_fit) {
    const int kq = e->sparse_top_levels >= 0 ? e->sparse_top_levels : (int)std::min<uint32_t>(std::max<uint32_t>(max_depth, kSparseMinTop), kSparseMaxTop);
    for (int K = kq; K >= (e->sparse_top_levels >= 0 ? kq : kSparseMinTop); --K)
      for (int d = e->sparse_dk ? 1 : 0; d >= 0; --d) {  // dense level K first: one more level fits next to the tile
        snprintf(name, sizeof(name), d ? "sparse_qd_k%d_u8_t1024" : "sparse_q_k%d_u8_t1024", K);
        const int vid = find_variant(name);
        if (fits(vid, kMaxLdsBytes)) return vid;
      }
  }
  // Dense level K (option "sparse_dk", default on): the same walk with a third less LDS per tree -- where it exists (256- and
  // 512-tuple tiles) it is tried first, so a geometry reaches a larger K or a second block per CU.
  if (e->sparse_top_levels >= 0) {
    for (int d = e->sparse_dk ? 1 : 0; d >= 0; --d)
      for (int T : {256, 128, 64}) {
        snprintf(name, sizeof(name), d ? "sparse_dk_k%d_u8_t%d" : "sparse_k%d_u8_t%d", e->sparse_top_levels, T);
        const int vid = find_variant(name);
        if (fits(vid, kMaxLdsBytes)) return vid;
      }
    return -1;
  }
  // deepest K worth staging: nothing is left for the deep phase beyond the deepest tree
  const int kcap = (int)std::min<uint32_t>(std::max<uint32_t>(max_depth, kSparseMinTop), kSparseMaxTop);
  // Geometries that keep >= 8 waves on a CU (then fewer, for very wide tuples), each with the largest K whose top
  // images fit next to the feature tile.
  static const struct { int T; uint32_t blocks; } geo[] = {{256, 2}, {128, 4}, {256, 1}, {128, 2}, {128, 1}, {64, 2}, {64, 1}};  // (512-tuple tiles never won a choice: removed in round 6)
  // The largest K wins, where a geometry with two or more blocks per CU counts one level more (one block's top phase overlaps the
  // others' deep phase: BASELINE config 4, profiles/archive/r03_sparse_dense_level_k.json -- K = 8 in two blocks 256.6 Mtuples/s, K = 9 in one
  // 243.4, K = 8 in one 232.5); ties go to the earlier geometry.
  int best = -1, best_score = -1;
  uint32_t best_waves = 0;
  for (const auto& g : geo) {
    const uint32_t waves = (uint32_t)g.T / 64u * g.blocks;
    if (best >= 0 && waves < best_waves) break;  // only fall to fewer waves when nothing fitted with more
    for (int K = kcap; K >= kSparseMinTop; --K) {
      snprintf(name, sizeof(name), "sparse_dk_k%d_u8_t%d", K, g.T);
      int vid = e->sparse_dk ? find_variant(name) : -1;
      if (!fits(vid, kMaxLdsBytes / g.blocks)) {
        snprintf(name, sizeof(name), "sparse_k%d_u8_t%d", K, g.T);
        vid = find_variant(name);
      }
      if (!fits(vid, kMaxLdsBytes / g.blocks)) continue;
      const int score = K + (g.blocks >= 2u && (uint32_t)K < max_depth ? 1 : 0);  // (nothing to overlap when the whole forest is in LDS)
      if (score > best_score) {
        best = vid;
        best_score = score;
        best_waves = waves;
      }
      break;
    }
  }
  // tuples too wide for any feature tile: the kernel that gathers its features from global memory (any width; the correctness path)
  if (best < 0) best = find_variant("sparse_gf_k6_u8_t256");
  return best;
}

}  // namespace

// sorted distinct threshold keys per feature over the forests of every class (they share one pre-pass per batch)
static RankTables sparse_rank_tables(const ddt_params& p, const std::vector<const SparseForest*>& sps) {
  RankTables rt;
  rt.keys.resize(tuple_words(p));
  for (const SparseForest* sp : sps)
    for (size_t n = 0; n < sp->lines.size() / 4u; ++n) rt.keys[sp->lines[4u * n + 1u] & 0x7FFu].push_back(thr_key(p, sp->lines[4u * n]));
  finish_rank_tables(rt);
  return rt;
}

void sparse_free(ddt_engine* e) {
  free_rank_device(e->sp_rank);
  if (e->sp_r32_tab) (void)hipFree(e->sp_r32_tab);
  e->sp_r32_tab = nullptr;
  e->sp_r32_tab_bytes = 0;
  for (SparseForest& sp : e->sps) {
    for (void** p : {&sp.d_top, &sp.d_deep}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    sp.top_bytes = sp.deep_bytes = 0;
    sp.groups = 0;
  }
}

static int sparse_pack(ddt_engine* e, const Variant& v, SparseForest& sp, const RankTables* rt);
static int sparse_pack_host_r(ddt_engine* e, const Variant& v, SparseForest& sp, const RankTables& rt, std::vector<uint32_t>& top, std::vector<uint32_t>& deep,
                              uint32_t* groups_out);
static int pack_rank32_tables(ddt_engine* e, const RankTables& rt, uint32_t W, RankHostTables& h, std::vector<uint32_t>& tab, uint32_t* blk_log2);

// "sparse_r_*" (32-bit ranks, pair records on every deep level; ddt_sparse_r.hip): the largest K whose blocks still come two to a CU, then
// one block per CU, 256-tuple tiles before 128
static int pick_r32_variant(ddt_engine* e, uint32_t max_depth) {
  const uint32_t W = tuple_words(e->p);
  if (e->p.num_features > 128u) return -1;  // a node word carries the feature number in 7 bits
  if (W > kSrMaxWords) return -1;            // the pre-pass's transpose stages 256 rows x W words in LDS
  char name[40];
  const int kcap = (int)std::min<uint32_t>(std::max<uint32_t>(max_depth, 9u), 10u);  // (a top image may be deeper than the forest: early leaves are padded)
  for (uint32_t budget : {kMaxLdsBytes / 2u, kMaxLdsBytes})
    for (int T : {256, 128})
      for (int K = e->sparse_top_levels >= 0 ? e->sparse_top_levels : kcap; K >= (e->sparse_top_levels >= 0 ? e->sparse_top_levels : 9); --K) {
        snprintf(name, sizeof(name), "sparse_r_k%d_u8_t%d", K, T);
        const int vid = find_variant(name);
        if (vid >= 0 && variant(vid).lds_bytes_sparse(W) <= budget) return vid;
      }
  return -1;
}

static int sparse_pack_host(ddt_engine* e, const Variant& v, const SparseForest& sp, const RankTables* rt, std::vector<uint32_t>& top,
                            std::vector<uint32_t>& deep, uint32_t* groups_out);

// Choose the kernel for the loaded forest(s) and pack the device images of every class for it.
int sparse_rebuild(ddt_engine* e) {
  uint32_t max_depth = 0;
  std::vector<const SparseForest*> all;
  for (const SparseForest& sp : e->sps) {
    max_depth = sp.max_depth > max_depth ? sp.max_depth : max_depth;
    all.push_back(&sp);
  }
  RankTables rt;
  try {
    rt = sparse_rank_tables(e->p, all);
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "rank table allocation failed");
  }
  int vid = pick_variant(e, max_depth, rt.max_len <= e->q16_max_table);
  // 32-bit ranks + pair records on every deep level (option "sparse_r32": -1 automatic, 0 never, 1 wherever such a kernel fits): for forests that
  // go well below the top image -- every two levels there cost ONE gather instead of two -- and hold enough trees to carry the rank pre-pass
  // (transpose + rank32_kernel per batch, which the fp32-tile kernels do not have)
  if (e->forced_variant >= 0 && variant(e->forced_variant).r32()) {
    const Variant& fv = variant(e->forced_variant);
    if (e->p.num_features > 128u || tuple_words(e->p) > kSrMaxWords || rt.max_len > kSrMaxTable || fv.lds_bytes_sparse(tuple_words(e->p)) > kMaxLdsBytes) vid = -1;
    else vid = e->forced_variant;
  } else if (e->forced_variant < 0 && e->sparse_r32 != 0 && rt.max_len <= kSrMaxTable) {
    // Automatic: where it measured faster on one MI355X (profiles/r06_sparse_r32.md section 4; 4 M tuples, trees x depth x features: 512 x 16 x 64 +12 %, the
    // same forest on 255 bins +37 %, 1000 x 13 x 28 +10 %, 128 x 14 x 20 +13 %, 512 x 16 x 32 +5 %, 512 x 16 x 128 +4 %) and not where it lost (64 x 16 x 64
    // -2.5 %, 256 x 12 x 64 -6 %, 512 x 10 x 64 -13 %, and 780 k thresholds per feature -- key blocks of 32, eight gathers per value in the pre-pass --
    // -49 %): depth >= 13, at least two trees per tuple word (the pre-pass costs per word, the pair records pay per tree), tuples of at most 128 words,
    // key blocks of four (at most 4 x 32767 distinct thresholds on a feature).
    uint32_t trees = 0;
    for (const SparseForest& sp : e->sps) trees += sp.trees();
    const uint32_t W = tuple_words(e->p);
    const bool pays = max_depth >= 13u && trees >= 2u * W && W <= kSrMaxWords && rt.max_len <= 4u * kSrMaxDir && !getenv("DDT_R32_BLK_LOG2");
    const int vr = (e->sparse_r32 > 0 || pays) ? pick_r32_variant(e, max_depth) : -1;
    if (vr >= 0) vid = vr;
  }
  // Dense mid levels (option "sparse_dm": -1 automatic, 0 never, M = exactly M): where the choice is a dense-level-K kernel that has
  // "sparse_dm<M>_*" siblings, the levels K .. K+M-1 become 8-byte heap records when the forest fills them at least half (the padding
  // under early leaves doubles per level).  Automatic = one mid level; more only when asked for (A/B)
  if (vid >= 0 && e->forced_variant < 0 && (e->sparse_dm != 0 || e->sparse_dp != 0) && (variant(vid).opt & 2) && !(variant(vid).opt & (4 | 8 | 16))) {  // (rank-quantised choices: the pair records only)
    const Variant& dkv = variant(vid);
    const uint32_t K = (uint32_t)dkv.levels;
    std::vector<double> nodes(K + 4u, 0.0);  // internal nodes per level K .. K+3 over all forests
    double trees = 0.0;
    std::vector<uint32_t> cur, nxt;
    for (const SparseForest& sp : e->sps) {
      trees += sp.trees();
      for (uint32_t i = 0; i < sp.trees(); ++i) {
        const uint32_t* L = sp.lines.data() + sp.first[i] * 4u;
        cur.assign(1, 0u);
        for (uint32_t lvl = 0; lvl < K + 3u && !cur.empty(); ++lvl) {
          if (lvl >= K) nodes[lvl] += (double)cur.size();
          nxt.clear();
          for (uint32_t n : cur)
            for (uint32_t side = 0; side < 2; ++side)
              if (!((L[4u * n + 1u] >> (14u + side)) & 1u)) nxt.push_back(L[4u * n + 2u + side]);
          cur.swap(nxt);
        }
      }
    }
    // Dense pair records (option "sparse_dp": -1 automatic, 0 never, 1 always where such a kernel exists): one gather for the levels K and K+1 and
    // a dense block at level K+2 -- when the forest fills level K at least half and level K+1 a quarter (a walker that ends above them would
    // fetch two padding records where the other forms fetch one; measured on seven fills of the config-4 generator, level K+1 30 .. 100 % and
    // level K+2 6 .. 73 % full: +2 .. +5 % over the better of the other two, profiles/r05_raw/s15_*, s16_*) and its feature numbers fit a byte
    bool took_pairs = false;
    if (e->sparse_dp != 0 && e->sparse_dm <= 0 && e->p.num_features <= 256u) {
      bool full = trees > 0.0;
      full = full && nodes[K] >= 0.5 * trees * (double)(1u << K) && nodes[K + 1u] >= 0.25 * trees * (double)(2u << K);
      char name[48];
      if (dkv.opt & 1) snprintf(name, sizeof(name), "sparse_qp_k%u_u8_t1024", K);
      else snprintf(name, sizeof(name), "sparse_dp_k%u_u8_t%d", K, dkv.threads);
      const int vp = find_variant(name);
      if ((full || e->sparse_dp > 0) && vp >= 0 && variant(vp).lds_bytes_sparse(tuple_words(e->p)) <= dkv.lds_bytes_sparse(tuple_words(e->p))) {
        vid = vp;
        took_pairs = true;
      }
    }
    for (int M = 3; M >= 1 && !took_pairs && e->sparse_dm != 0 && !(dkv.opt & 1); --M) {
      if (e->sparse_dm > 0 ? M != e->sparse_dm : M > 1) continue;  // automatic: ONE mid level (measured: +2-4 %; two +1 %, three -17 %)
      bool full = trees > 0.0;
      for (uint32_t lvl = K; lvl < K + (uint32_t)M; ++lvl) full = full && nodes[lvl] >= 0.5 * trees * (double)(1u << lvl);
      char name[48];
      snprintf(name, sizeof(name), "sparse_dm%d_k%u_u8_t%d", M, K, dkv.threads);
      const int vm = find_variant(name);
      if ((full || e->sparse_dm > 0) && vm >= 0 && variant(vm).lds_bytes_sparse(tuple_words(e->p)) <= dkv.lds_bytes_sparse(tuple_words(e->p))) {
        vid = vm;
        break;
      }
    }
  }
  if (vid < 0)
    return fail(e, DDT_EUNSUPPORTED, "no sparse kernel fits: %u tuple words need more than %u bytes of LDS (or the forced variant %d / "
                "sparse_top_levels %d is not a sparse kernel that fits)", tuple_words(e->p), kMaxLdsBytes, e->forced_variant, e->sparse_top_levels);
  sparse_free(e);
  free_q16_workspace(e);  // (its geometry follows the kernel family: u16 ranks in tiles of 1024, 32-bit rank words in tiles of the block size; callers have synchronised)
  if (variant(vid).r32()) {  // directory + key blocks of the 32-bit rank pre-pass (shared by all classes)
    RankHostTables h;
    std::vector<uint32_t> tab;
    uint32_t blk_log2 = 2;
    int rc = pack_rank32_tables(e, rt, tuple_words(e->p), h, tab, &blk_log2);
    if (rc) return rc;
    rc = upload_rank_tables(e, h, e->sp_rank);
    if (rc) return rc;
    HIP_TRY(e, hipMalloc(&e->sp_r32_tab, tab.size() * 4u));
    HIP_TRY(e, hipMemcpy(e->sp_r32_tab, tab.data(), tab.size() * 4u, hipMemcpyHostToDevice));
    e->sp_r32_tab_bytes = tab.size() * 4u;
    e->sp_r32_blk_log2 = blk_log2;
    for (SparseForest& sp : e->sps) {
      std::vector<uint32_t> top, deep;
      uint32_t groups = 0;
      rc = sparse_pack_host_r(e, variant(vid), sp, rt, top, deep, &groups);
      if (rc) return rc;
      HIP_TRY(e, hipMalloc(&sp.d_top, top.size() * 4u));
      HIP_TRY(e, hipMalloc(&sp.d_deep, deep.size() * 4u));
      HIP_TRY(e, hipMemcpy(sp.d_top, top.data(), top.size() * 4u, hipMemcpyHostToDevice));
      HIP_TRY(e, hipMemcpy(sp.d_deep, deep.data(), deep.size() * 4u, hipMemcpyHostToDevice));
      sp.top_bytes = top.size() * 4u;
      sp.deep_bytes = deep.size() * 4u;
      sp.groups = groups;
    }
    e->variant_id = vid;
    return DDT_OK;
  }
  const bool q = (variant(vid).opt & 1) != 0;
  if (q) {  // the tables of the rank pre-pass (shared by all classes)
    RankHostTables h;
    int rc = pack_rank_tables(e, rt, tuple_words(e->p), true, h);
    if (rc) return rc;
    rc = upload_rank_tables(e, h, e->sp_rank);
    if (rc) return rc;
  }
  for (SparseForest& sp : e->sps) {
    int rc = sparse_pack(e, variant(vid), sp, q ? &rt : nullptr);
    if (rc) return rc;
  }
  e->variant_id = vid;
  return DDT_OK;
}

// Pack the device images of one forest for kernel `v` and upload them.
// the host half of the packing (no HIP call: also behind the test hook ddt_debug_sparse_image)
// `rt` (rank-quantised kernels): a node's threshold word is its rank R = 1 + index of the threshold among the sorted distinct
// keys of its feature (x >= t  <=>  rank(x) >= R), else the threshold key itself
static int sparse_pack_host(ddt_engine* e, const Variant& v, const SparseForest& sp, const RankTables* rt, std::vector<uint32_t>& top,
                            std::vector<uint32_t>& deep, uint32_t* groups_out) {
  const uint32_t K = (uint32_t)v.levels, T = sp.trees();
  const uint32_t per_pass = std::max(1u, (uint32_t)v.chunk_trees / 8u);  // PU groups walked in lock-step (half groups: 1)
  uint32_t groups = T ? (T + 7u) / 8u : 1u;  // an empty shard is one group of EMPTY slots
  groups = (groups + per_pass - 1u) / per_pass * per_pass;  // whole passes: the padding groups are EMPTY slots too (+0)
  const bool dk = (v.opt & 2) != 0;  // dense level K: all K levels as 8-byte records in LDS, level K a dense block of deep records
  // dense MID levels ("sparse_dm<M>_*", opt bit 3): the levels K .. K+M-1 continue the heap as 8-byte records in the deep array (record of heap
  // node h at byte cbase + 8 h), the dense block of 16-byte records is level K+M (ddt_sparse.hip sparse_walk)
  // dense PAIR records ("sparse_dp_*", opt bit 4): the levels K and K+1 as ONE block of 2^K 16-byte records {key of the level-K node, keys of its
  // two children, their feature numbers as bytes + missing directions in the top byte} at byte cbase + 16 h (h = level-K heap index); the dense
  // block of ordinary records is level K+2
  const bool pairs = dk && (v.opt & 16) != 0;
  const uint32_t M = pairs ? 2u : (dk && (v.opt & 8)) ? (uint32_t)v.top : 0u, KD = K + M;  // KD = the level of the dense 16-byte block
  const uint32_t top_words = v.top_bytes_sparse() / 4u;  // per tree
  const uint32_t lvl8 = dk ? K : K - 1u;                 // levels stored as 8-byte heap records in the top image
  const uint32_t mid_words = pairs ? (4u << K) : 2u * ((1u << KD) - (1u << K));  // words of a tree's mid levels (8 bytes per record; pair records: 16 per level-K node)
  const uint32_t feat_off = v.feat_off_sparse(), row = v.row_bytes_sparse();
  auto feat_word = [&](uint32_t j) { return feat_off + j * row; };

  try {
    top.assign((size_t)groups * 8u * top_words, 0u);
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "sparse top image allocation failed");
  }
  // dummy record: both children are the leaf +0 (EMPTY slots and the padding under early leaves use it with their value)
  auto put8 = [&](uint32_t* t, uint32_t m, uint32_t key, uint32_t w) {
    t[2u * m + 0u] = key;
    t[2u * m + 1u] = w;
  };
  auto put16 = [&](uint32_t* rec, uint32_t key, uint32_t w, uint32_t l, uint32_t r) {
    rec[0] = key;
    rec[1] = w;
    rec[2] = l;
    rec[3] = r;
  };
  // record 0: a valid dummy (finished lanes keep re-reading it); dense level K: a whole block of 2^K of them, the level K of every
  // EMPTY slot
  try {
    deep.assign(dk ? (size_t)mid_words + ((size_t)4u << KD) : 4u, 0u);
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "sparse image allocation failed");
  }
  for (size_t q = 0; q < mid_words / 2u && !pairs; ++q) {  // (dense mid levels: the EMPTY slots' dummy heap; pair records: all zero = feature 0 against key 0)
    deep[2u * q] = 0u;
    deep[2u * q + 1u] = feat_word(0);
  }
  for (size_t q = 0; q < (deep.size() - mid_words) / 4u; ++q) put16(deep.data() + mid_words + 4u * q, 0u, feat_word(0) | kSpLeftLeaf | kSpRightLeaf, 0u, 0u);
  // word 0 of a tree's top image: the byte offset the kernel adds a heap index to -- 16 h for the dense level K, 8 h for dense mid levels
  const auto cbase_of = [&](size_t first_record) {
    return (M && !pairs) ? (uint32_t)(first_record << 4) - (8u << K) : (uint32_t)(first_record << 4) - (16u << K);
  };  // see ddt_internal.h

  std::vector<Cursor> cur, nxt;
  struct Patch {  // a child word to patch with the deep index of tree node `node` once the deep records of the tree are placed
    uint32_t node;
    bool in_deep;  // the word lives in `deep` (which grows: keep its index) / in `top`
    size_t word;
  };
  std::vector<Patch> pending;
  std::vector<uint32_t> order, stack;
  try {  // every container below grows with the model: an allocation failure is DDT_ENOMEM, never an exception across the C ABI
  for (uint32_t i = 0; i < groups * 8u; ++i) {
    uint32_t* t = top.data() + (size_t)i * top_words;
    uint32_t* last = t + (4u << K) / 4u;  // 16-byte records of level K-1 (not dense level K)
    if (i >= T) {  // EMPTY slot: contributes exactly +0 (DTPU.sv:544,760)
      for (uint32_t m = 1; m < (1u << lvl8); ++m) put8(t, m, 0u, feat_word(0));
      if (dk) t[0] = cbase_of(0);  // the shared dummy block
      else
        for (uint32_t r = 0; r < (1u << (K - 1)); ++r) put16(last + 4u * r, 0u, feat_word(0) | kSpLeftLeaf | kSpRightLeaf, 0u, 0u);
      continue;
    }
    const uint32_t* L = sp.lines.data() + sp.first[i] * 4u;
    auto child = [&](uint32_t n, uint32_t side) { return Cursor{((L[4u * n + 1u] >> (14u + side)) & 1u) != 0u, L[4u * n + 2u + side]}; };
    auto node_w = [&](uint32_t n) { return feat_word(L[4u * n + 1u] & 0x7FFu) | (((L[4u * n + 1u] >> 13) & 1u) ? kSpMissRight : 0u); };
    auto node_key = [&](uint32_t n) -> uint32_t {
      const uint32_t key = thr_key(e->p, L[4u * n]);
      if (!rt) return key;
      const auto& k = rt->keys[L[4u * n + 1u] & 0x7FFu];
      return 1u + (uint32_t)(std::lower_bound(k.begin(), k.end(), key, [](uint32_t a, uint32_t b) { return (int32_t)a < (int32_t)b; }) - k.begin());
    };
    // ---- top heap, level by level ----
    cur.assign(1, Cursor{false, 0u});
    for (uint32_t lvl = 0; lvl < lvl8; ++lvl) {
      nxt.clear();
      for (uint32_t k = 0; k < cur.size(); ++k) {
        const uint32_t m = (1u << lvl) + k;
        if (cur[k].leaf) {  // padding under an early leaf: any direction ends on the same value
          put8(t, m, 0u, feat_word(0));
          nxt.push_back(cur[k]);
          nxt.push_back(cur[k]);
        } else {
          const uint32_t n = cur[k].v;
          put8(t, m, node_key(n), node_w(n));
          nxt.push_back(child(n, 0));
          nxt.push_back(child(n, 1));
        }
      }
      cur.swap(nxt);
    }
    // ---- the first 16-byte records, whose children are leaves or deep records: level K-1 in the top image, or (dense level K) the
    //      tree's dense block of 2^K records at the end of the deep array -- behind its M dense mid levels of 8-byte records ----
    pending.clear();
    const size_t dense0 = deep.size() / 4u;
    if (dk) {
      if (dense0 + mid_words / 4u + ((size_t)1u << KD) >= (1ull << 28)) return fail(e, DDT_EUNSUPPORTED, "more than 2^28 deep records (4 GiB of 16-byte records)");
      deep.resize(deep.size() + mid_words + ((size_t)4u << KD));
      t[0] = cbase_of(dense0);
      for (uint32_t lvl = K; lvl < KD; ++lvl) {  // the mid levels continue the heap (cur = the 2^lvl cursors of level lvl)
        uint32_t* d8 = deep.data() + dense0 * 4u;
        nxt.clear();
        for (uint32_t k = 0; k < cur.size(); ++k) {
          const size_t w2 = 2u * (((size_t)1u << lvl) + k - ((size_t)1u << K));
          if (pairs) {  // record of the level-K node (k at level K, k >> 1 at level K+1): word 0 / 1 / 2 = the node's / left child's / right child's key
            uint32_t* pr = d8 + 4u * (lvl == K ? k : k >> 1);
            const uint32_t pos = lvl == K ? 0u : 1u + (k & 1u);
            if (cur[k].leaf) {  // padding under an early leaf: key 0, feature 0 -- any direction ends on the same value
              nxt.push_back(cur[k]);
              nxt.push_back(cur[k]);
            } else {
              const uint32_t n = cur[k].v;
              pr[pos] = node_key(n);
              pr[3] |= ((L[4u * n + 1u] & 0x7FFu) << (8u * pos)) | (((L[4u * n + 1u] >> 13) & 1u) << (24u + pos));
              nxt.push_back(child(n, 0));
              nxt.push_back(child(n, 1));
            }
            continue;
          }
          if (cur[k].leaf) {
            d8[w2] = 0u;
            d8[w2 + 1u] = feat_word(0);
            nxt.push_back(cur[k]);
            nxt.push_back(cur[k]);
          } else {
            const uint32_t n = cur[k].v;
            d8[w2] = node_key(n);
            d8[w2 + 1u] = node_w(n);
            nxt.push_back(child(n, 0));
            nxt.push_back(child(n, 1));
          }
        }
        cur.swap(nxt);
      }
    }
    for (uint32_t k = 0; k < cur.size(); ++k) {
      const size_t word = dk ? dense0 * 4u + mid_words + (size_t)k * 4u : (size_t)(last + 4u * k - top.data());
      uint32_t* rec = (dk ? deep.data() : top.data()) + word;
      if (cur[k].leaf) {
        put16(rec, 0u, feat_word(0) | kSpLeftLeaf | kSpRightLeaf, cur[k].v, cur[k].v);
        continue;
      }
      const uint32_t n = cur[k].v;
      uint32_t w = node_w(n);
      for (uint32_t side = 0; side < 2; ++side) {
        const Cursor c = child(n, side);
        if (c.leaf) {
          w |= side ? kSpRightLeaf : kSpLeftLeaf;
          rec[2u + side] = c.v;
        } else {
          pending.push_back(Patch{c.v, dk, word + 2u + side});
        }
      }
      rec[0] = node_key(n);
      rec[1] = w;
    }
    // ---- deep records of this tree: the sub-trees hanging below level K-1 ----
    // order 0: all of them level by level (breadth-first over the whole remainder of the tree);
    // order 1: one sub-tree after the other, each in depth-first pre-order (a parent's left child is the next
    //          record: half of the steps of a walk stay inside the cache line they are in)
    order.clear();
    if (e->sparse_deep_order == 0) {
      for (auto& pe : pending) order.push_back(pe.node);
      for (size_t q = 0; q < order.size(); ++q)
        for (uint32_t side = 0; side < 2; ++side) {
          const Cursor c = child(order[q], side);
          if (!c.leaf) order.push_back(c.v);
        }
    } else {
      for (auto& pe : pending) {
        stack.assign(1, pe.node);
        while (!stack.empty()) {
          const uint32_t n = stack.back();
          stack.pop_back();
          order.push_back(n);
          const Cursor r = child(n, 1), l = child(n, 0);
          if (!r.leaf) stack.push_back(r.v);
          if (!l.leaf) stack.push_back(l.v);  // popped first: pre-order, left before right
        }
      }
    }
    const size_t base = deep.size() / 4u;
    if (base + order.size() >= (1ull << 28)) return fail(e, DDT_EUNSUPPORTED, "more than 2^28 deep records (4 GiB of 16-byte records)");
    // node -> deep index: the tree's node indices are dense, use a scratch map sized by the tree
    const uint64_t cnt = sp.first[i + 1] - sp.first[i];
    std::vector<uint32_t> where(cnt, 0u);
    for (size_t q = 0; q < order.size(); ++q) where[order[q]] = (uint32_t)(base + q);
    deep.resize(deep.size() + order.size() * 4u);
    for (size_t q = 0; q < order.size(); ++q) {
      const uint32_t n = order[q];
      uint32_t* rec = deep.data() + (base + q) * 4u;
      uint32_t w = node_w(n);
      for (uint32_t side = 0; side < 2; ++side) {
        const Cursor c = child(n, side);
        if (c.leaf) w |= side ? kSpRightLeaf : kSpLeftLeaf;
        rec[2u + side] = c.leaf ? c.v : where[c.v];
      }
      rec[0] = node_key(n);
      rec[1] = w;
    }
    for (auto& pe : pending) (pe.in_deep ? deep.data() : top.data())[pe.word] = where[pe.node];
  }
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "sparse image allocation failed");
  }
  *groups_out = groups;
  return DDT_OK;
}

// Tables of the 32-bit rank pre-pass (ddt_internal.h R32Aux; ddt_sparse_r.hip rank32_kernel): per feature the keys in blocks of 2^blk_log2
// (padded with INT_MAX, one all-pad block behind them) and the DIRECTORY of the blocks' last keys, which goes through pack_rank_tables like a
// u16 table (flat [W][Kpad] + bucket starts + search parameters).  The block size is the smallest that keeps every directory within kSrMaxDir.
static int pack_rank32_tables(ddt_engine* e, const RankTables& rt, uint32_t W, RankHostTables& h, std::vector<uint32_t>& tab, uint32_t* blk_log2_out) {
  uint32_t bl = 2;
  if (const char* v = getenv("DDT_R32_BLK_LOG2")) bl = std::min(std::max(atoi(v), 2), 8);  // A/B: larger blocks = smaller directories (more resident blocks per CU), more gathers per value
  while (((uint64_t)rt.max_len + (1u << bl) - 1u) >> bl > kSrMaxDir) ++bl;
  const uint32_t B = 1u << bl;
  try {
    RankTables dirt;
    dirt.keys.resize(W);
    std::vector<uint32_t> koff(W, 0u);
    tab.clear();
    for (uint32_t j = 0; j < W; ++j) {
      const std::vector<uint32_t>& k = rt.keys[j];
      const uint32_t nb = ((uint32_t)k.size() + B - 1u) / B;
      koff[j] = (uint32_t)tab.size();
      tab.insert(tab.end(), k.begin(), k.end());
      tab.resize((size_t)koff[j] + (size_t)(nb + 1u) * B, 0x7FFFFFFFu);  // the last block's padding + one all-pad block
      // (the last block's entry is the last REAL key, not its INT_MAX padding: the directory's key range -- what its bucket index slices -- stays
      // the feature's own; values >= the largest key take the rank K without a search)
      for (uint32_t b = 0; b < nb; ++b) dirt.keys[j].push_back(b + 1u < nb ? tab[(size_t)koff[j] + (size_t)b * B + B - 1u] : k.back());
      if (dirt.keys[j].size() > dirt.max_len) dirt.max_len = (uint32_t)dirt.keys[j].size();
    }
    if ((uint64_t)tab.size() * 4u >= (1ull << 32)) return fail(e, DDT_EUNSUPPORTED, "rank tables of %zu keys exceed a 4 GiB buffer", tab.size());
    const int rc = pack_rank_tables(e, dirt, W, false, h);
    if (rc) return rc;
    for (uint32_t j = 0; j < W; ++j) {
      uint32_t* P = h.tabK.data() + (size_t)j * 8u;
      P[5] = koff[j];
      P[6] = (uint32_t)rt.keys[j].size();
      P[7] = rt.keys[j].empty() ? 0x7FFFFFFFu : rt.keys[j].back();
    }
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "rank table allocation failed");
  }
  *blk_log2_out = bl;
  return DDT_OK;
}

// Images of the "sparse_r_*" kernels (ddt_internal.h "32-bit ranks"): per tree a top image of one-word nodes and, in the deep array, its dense
// block of level-K pair records followed by the exit blocks of the pair records below, level by level (breadth-first: measured faster than
// depth-first for the one-record-per-level images, profiles/archive).  Also counts the record hops of the forest's longest path (sp.r_rounds).
static int sparse_pack_host_r(ddt_engine* e, const Variant& v, SparseForest& sp, const RankTables& rt, std::vector<uint32_t>& top, std::vector<uint32_t>& deep,
                              uint32_t* groups_out) {
  const uint32_t K = (uint32_t)v.levels, T = sp.trees();
  const uint32_t per_pass = std::max(1u, (uint32_t)v.chunk_trees / 8u);  // PU groups walked in lock-step
  uint32_t groups = T ? (T + 7u) / 8u : 1u;  // an empty shard is one group of EMPTY slots
  groups = (groups + per_pass - 1u) / per_pass * per_pass;  // whole passes: the padding groups are EMPTY slots too (+0)
  const uint32_t top_words = v.top_bytes_sparse() / 4u;  // 2^K per tree
  uint32_t rounds = 1;
  struct Todo {  // a pair record still to be written: rooted at tree node `node`, at record index `slot` of the deep array, `hop` records into the walk
    uint32_t node, hop;
    size_t slot;
  };
  try {
    top.assign((size_t)groups * 8u * top_words, 0u);
    // records 0 .. 3: {0, 0, 0, 0xFFFFFFC0} -- where a finished walker lands when it comes back from beyond the buffer's range with a record of zeros
    // (ddt_sparse_r.hip: no leaf flag, next block at byte 0), and which sends it out of range again (0xFFFFFFC0 + 32 r0 + 16 r1 <= 0xFFFFFFF0: no wrap);
    // records 4 .. 4 + 2^K - 1: LEAF(+0) -- the dense block every EMPTY slot shares (DTPU.sv:544,760: an EMPTY slot adds exactly +0)
    deep.assign(16u + ((size_t)4u << K), 0u);
    for (uint32_t q = 0; q < 4u; ++q) deep[4u * q + 3u] = 0xFFFFFFC0u;
    for (uint32_t q = 0; q < (1u << K); ++q) deep[16u + 4u * q] = kSrLeafRec;
    std::vector<Cursor> cur, nxt;
    std::vector<Todo> todo;
    for (uint32_t i = 0; i < groups * 8u; ++i) {
      uint32_t* t = top.data() + (size_t)i * top_words;
      if (i >= T) {
        t[0] = 64u - (16u << K);  // cbase of the shared block at byte 64
        continue;                // (node words 0: feature 0 against rank 0 -- any direction ends in the block of LEAF(+0))
      }
      const uint32_t* L = sp.lines.data() + sp.first[i] * 4u;
      auto child = [&](uint32_t n, uint32_t side) { return Cursor{((L[4u * n + 1u] >> (14u + side)) & 1u) != 0u, L[4u * n + 2u + side]}; };
      auto node_word = [&](uint32_t n) -> uint32_t {  // R << 15 | feature << 8 | flags
        const uint32_t j = L[4u * n + 1u] & 0x7FFu, key = thr_key(e->p, L[4u * n]);
        const auto& k = rt.keys[j];
        const uint32_t R = 1u + (uint32_t)(std::lower_bound(k.begin(), k.end(), key, [](uint32_t a, uint32_t b) { return (int32_t)a < (int32_t)b; }) - k.begin());
        return (R << kSrRankShift) | (((L[4u * n + 1u] >> 13) & 1u) ? kSrMissRight : 0u) | (((L[4u * n + 1u] >> 14) & 1u) ? kSrLeftLeaf : 0u) |
               (((L[4u * n + 1u] >> 15) & 1u) ? kSrRightLeaf : 0u) | (j << kSrFeatShift);
      };
      // ---- top heap, level by level (padding under an early leaf: node word 0, both children the leaf) ----
      cur.assign(1, Cursor{false, 0u});
      for (uint32_t lvl = 0; lvl < K; ++lvl) {
        nxt.clear();
        for (uint32_t k = 0; k < cur.size(); ++k) {
          const uint32_t m = (1u << lvl) + k;
          if (cur[k].leaf) {
            if (m) t[m] = 0u;
            nxt.push_back(cur[k]);
            nxt.push_back(cur[k]);
          } else {
            t[m] = node_word(cur[k].v);
            nxt.push_back(child(cur[k].v, 0));
            nxt.push_back(child(cur[k].v, 1));
          }
        }
        cur.swap(nxt);
      }
      // ---- the dense block of level K: a pair record per internal node, a LEAF record per (early) leaf ----
      const size_t dense0 = deep.size() / 4u;
      if ((dense0 + ((size_t)1u << K)) * 16u >= 0xFFFFFFC0ull) return fail(e, DDT_EUNSUPPORTED, "more than 4 GiB of deep records");
      deep.resize(deep.size() + ((size_t)4u << K), 0u);
      t[0] = (uint32_t)(dense0 * 16u) - (16u << K);
      todo.clear();
      for (uint32_t k = 0; k < cur.size(); ++k) {
        if (cur[k].leaf) {
          uint32_t* rec = deep.data() + (dense0 + k) * 4u;
          rec[0] = kSrLeafRec;
          rec[1] = rec[2] = cur[k].v;
        } else {
          todo.push_back(Todo{cur[k].v, 1u, dense0 + k});
        }
      }
      // ---- pair records, breadth-first: writing one appends its exit block (2 or 4 slots) to the deep array ----
      for (size_t q = 0; q < todo.size(); ++q) {
        const Todo td = todo[q];
        rounds = td.hop > rounds ? td.hop : rounds;
        const uint32_t n = td.node;
        const Cursor c0 = child(n, 0), c1 = child(n, 1);
        uint32_t ptr = 0u;
        const uint32_t inner = (c0.leaf ? 0u : 1u) + (c1.leaf ? 0u : 1u);
        if (inner) {
          const size_t block = deep.size() / 4u;
          if ((block + 2u * inner) * 16u >= 0xFFFFFFC0ull) return fail(e, DDT_EUNSUPPORTED, "more than 4 GiB of deep records");
          deep.resize(deep.size() + 8u * inner, 0u);
          ptr = (uint32_t)(block * 16u) - (c0.leaf ? 32u : 0u);  // slot of the grandchild on side r1 of the child on side r0: ptr + 32 r0 + 16 r1
          size_t slot = block;
          for (const Cursor& c : {c0, c1}) {
            if (c.leaf) continue;
            for (uint32_t side = 0; side < 2; ++side, ++slot) {
              const Cursor gc = child(c.v, side);
              if (gc.leaf) {  // a leaf two levels down: a LEAF record in the grandchild's slot
                uint32_t* rec = deep.data() + slot * 4u;
                rec[0] = kSrLeafRec;
                rec[1] = rec[2] = gc.v;
                rounds = td.hop + 1u > rounds ? td.hop + 1u : rounds;
              } else {
                todo.push_back(Todo{gc.v, td.hop + 1u, slot});
              }
            }
          }
        }
        uint32_t* rec = deep.data() + td.slot * 4u;
        rec[0] = node_word(n);
        rec[1] = c0.leaf ? c0.v : node_word(c0.v);
        rec[2] = c1.leaf ? c1.v : node_word(c1.v);
        rec[3] = ptr;
      }
    }
  } catch (const std::bad_alloc&) {
    return fail(e, DDT_ENOMEM, "sparse image allocation failed");
  }
  sp.r_rounds = rounds;
  *groups_out = groups;
  return DDT_OK;
}

static int sparse_pack(ddt_engine* e, const Variant& v, SparseForest& sp, const RankTables* rt) {
  std::vector<uint32_t> top, deep;
  uint32_t groups = 0;
  int rc = sparse_pack_host(e, v, sp, rt, top, deep, &groups);
  if (rc) return rc;
  HIP_TRY(e, hipMalloc(&sp.d_top, top.size() * 4u));
  HIP_TRY(e, hipMalloc(&sp.d_deep, deep.size() * 4u));
  HIP_TRY(e, hipMemcpy(sp.d_top, top.data(), top.size() * 4u, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(sp.d_deep, deep.data(), deep.size() * 4u, hipMemcpyHostToDevice));
  sp.top_bytes = top.size() * 4u;
  sp.deep_bytes = deep.size() * 4u;
  sp.groups = groups;
  return DDT_OK;
}

int sparse_launch(ddt_engine* e, uint32_t cls, const void* d_tuples, size_t n, float* d_scores, hipStream_t s, bool reuse_prepass) {
  const SparseForest& sp = e->sps[cls];
  const Variant& v = variant(e->variant_id);
  ScoreArgs a{};
  a.img = reinterpret_cast<const uint4*>(sp.d_top);
  a.tuples = reinterpret_cast<const uint32_t*>(d_tuples);
  a.out = d_scores;
  a.n = n;
  a.tuple_words = tuple_words(e->p);
  a.n_trees = sp.groups * 8u;
  a.n_chunks = sp.groups;
  a.levels = (uint32_t)v.levels;
  a.clusters = e->p.clusters_per_tuple;
  a.miss_raw = e->p.missing_bits;
  a.miss_key = e->p.cmp_mode ? kMissSentinelIeee : e->p.missing_bits;
  a.ieee = e->p.cmp_mode;
  a.sum_mode = e->p.sum_mode;
  a.num_cus = e->prop.multiProcessorCount > 0 ? (uint32_t)e->prop.multiProcessorCount : 256u;
  SparseAux x{};
  x.deep = reinterpret_cast<const uint4*>(sp.d_deep);
  x.n_groups = sp.groups;
  x.deep_bytes = (uint32_t)std::min<uint64_t>(sp.deep_bytes, 0xFFFFFFFFull);
  {  // visits of the kernel's deep loop on the forest's longest path: levels K-1 .. max_depth-1 (16-byte level K-1 records in LDS), K+M .. max_depth-1
     // (dense level K behind M dense mid levels); a shallower forest still consumes the one (dummy) record every walker fetches
    const uint32_t first_lvl = (v.opt & 2) ? (uint32_t)v.levels + ((v.opt & 16) ? 2u : (v.opt & 8) ? (uint32_t)v.top : 0u) : (uint32_t)v.levels - 1u;
    x.max_rounds = !(e->sparse_peel_last && e->sparse_idle_oob) ? 0xFFFFFFFFu :  // (the visit-only round tells a finished walker by the zeros of its out-of-range gather)
                    sp.max_depth > first_lvl ? sp.max_depth - first_lvl : 1u;
  }
  if (v.r32()) x.max_rounds = sp.r_rounds;  // record hops on the forest's longest path, counted by the packer
  x.idle_off = (e->sparse_idle_oob || v.r32()) ? 0xFFFFFFF0u : 0u;  // (the packers keep the deep array below 2^28 records, i.e. deep_bytes <= 0xFFFFFFF0: that offset is always out of range)
  if (v.r32()) {  // 32-bit ranks: the batch's rank words + per-tile missing flags come from the rank32 pre-pass (workspace slot e->q_slot)
    int rc = ensure_q16_workspace(e, n);
    if (rc) return rc;
    Q16Aux& qa = x.q16;
    qa.xT = reinterpret_cast<uint32_t*>(e->q_xT[e->q_slot]);
    qa.tile_flags = reinterpret_cast<uint32_t*>(e->q_flags[e->q_slot]);
    qa.tables = reinterpret_cast<const uint32_t*>(e->sp_rank.d_tables);  // the directories
    qa.tabP = reinterpret_cast<const uint32_t*>(e->sp_rank.d_tabK);
    qa.tabS = reinterpret_cast<const uint16_t*>(e->sp_rank.d_tabS);
    qa.Kpad = e->sp_rank.Kpad;
    qa.skip_prepass = reuse_prepass ? 1u : 0u;
    qa.n_pad = (n + 1023) / 1024 * 1024;
    x.r32.r = reinterpret_cast<uint32_t*>(e->q_q[e->q_slot]);
    x.r32.tab = reinterpret_cast<const uint32_t*>(e->sp_r32_tab);
    x.r32.tab_bytes = (uint32_t)e->sp_r32_tab_bytes;
    x.r32.blk_log2 = e->sp_r32_blk_log2;
    x.r32.tile = (uint32_t)v.threads;
    if (e->tev_cur) a.ev_mid = e->tev_cur[1];
    else if (e->ev_fork && !reuse_prepass) a.ev_mid = e->ev_fork;
  } else if (v.opt & 1) {  // rank-quantised: the batch's ranks + per-tile missing flags come from the q16 pre-pass (workspace slot e->q_slot)
    int rc = ensure_q16_workspace(e, n);
    if (rc) return rc;
    Q16Aux& qa = x.q16;
    qa.xT = reinterpret_cast<uint32_t*>(e->q_xT[e->q_slot]);
    qa.q = reinterpret_cast<uint16_t*>(e->q_q[e->q_slot]);
    qa.tile_flags = reinterpret_cast<uint32_t*>(e->q_flags[e->q_slot]);
    qa.tables = reinterpret_cast<const uint32_t*>(e->sp_rank.d_tables);
    qa.tabP = reinterpret_cast<const uint32_t*>(e->sp_rank.d_tabK);
    qa.tabS = reinterpret_cast<const uint16_t*>(e->sp_rank.d_tabS);
    qa.Kpad = e->sp_rank.Kpad;
    qa.skip_prepass = reuse_prepass ? 1u : 0u;
    qa.prepass_img = reinterpret_cast<const uint4*>(e->sp_rank.d_prepass);
    qa.prepass = e->sp_rank.prepass;
    qa.img_slow = nullptr;  // the sparse images carry the missing direction in every record
    qa.n_pad = (n + 1023) / 1024 * 1024;
    if (e->tev_cur) a.ev_mid = e->tev_cur[1];  // recorded between the pre-pass and the scoring kernel
    else if (e->ev_fork && !reuse_prepass) a.ev_mid = e->ev_fork;  // multi-class calls: where the other stream's classes may start
  }
  // A batch of a few tiles on the 32-bit-rank kernels: one block walks all the forest's PU groups for its 256 tuples (0.23 ms at 512 trees).  Cut into
  // slices of C consecutive groups (C = clusters: in stream order each cluster then takes ONE group's sum per slice, so the uncut walk's ring of
  // accumulators IS the groups' sums), the adds in the reference's order behind it (launch_cm_combine): FPAggregator.v:79-131, Core.sv:486-541.
  const uint32_t C = e->p.clusters_per_tuple ? e->p.clusters_per_tuple : 1u;
  bool cut = false;
  if (v.r32() && e->q16_cluster_split != 0 && e->p.sum_mode != 1u && !reuse_prepass && e->num_classes <= 1u && n > 0 && sp.groups > C) {
    const uint64_t tiles = (n + (uint32_t)v.threads - 1u) / (uint32_t)v.threads;  // (blocks of the uncut launch: two per CU)
    const uint32_t positions = (sp.groups + C - 1u) / C * C;
    cut = (e->q16_cluster_split > 0 || tiles <= e->sparse_split_max_tiles) && split_fits(positions, n);
    if (cut) {
      float* partials = nullptr;
      int rc = ensure_split_workspace(e, (uint64_t)positions * x.q16.n_pad, &partials);
      if (rc) return rc;
      x.q16.split = 1u;
      a.out = partials;
    }
  }
  a.aux = &x;
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be blamed on this launch
  hipError_t r = v.launch(a, v, s);
  if (r != hipSuccess) return fail(e, DDT_EHIP, "kernel launch (%s) -> %s", v.name, hipGetErrorString(r));
  if (cut) {
    r = launch_cm_combine(a.out, (size_t)x.q16.n_pad, n, sp.groups, C, true, false, d_scores, e->p.sum_mode == 2, s);
    if (r != hipSuccess) return fail(e, DDT_EHIP, "cm_combine -> %s", hipGetErrorString(r));
    e->st.kernel_launches++;
  }
  return DDT_OK;
}

}  // namespace ddt

using namespace ddt;

namespace {

// validate the trees `ids` of the stream and copy them (re-based) into `out`
int take_trees(ddt_engine* e, const ddt_params* p, const uint32_t* lines, size_t n_lines, const uint64_t* first, std::vector<uint32_t> ids,
               SparseForest* out) {
  SparseForest sp;
  sp.first.assign(1, 0u);
  std::vector<uint8_t> depth, seen;
  for (uint32_t id : ids) {
    if (first[id + 1] <= first[id]) return fail(e, DDT_EINVAL, "tree %u has no lines", id);
    if (first[id + 1] > n_lines)  // an inner entry of tree_first_line may point past the stream even when the last one does not
      return fail(e, DDT_EINVAL, "tree %u ends at line %llu of a stream of %zu lines", id, (unsigned long long)first[id + 1], n_lines);
    const uint64_t cnt = first[id + 1] - first[id];
    if (cnt > 0xFFFFFFFFull) return fail(e, DDT_EUNSUPPORTED, "tree %u has more than 2^32 nodes", id);
    const uint32_t* t = lines + first[id] * 4u;
    try {
      depth.assign(cnt, 0);
      seen.assign(cnt, 0);
    } catch (const std::bad_alloc&) {
      return fail(e, DDT_ENOMEM, "host model allocation failed");
    }
    seen[0] = 1;  // the root
    for (uint64_t n = 0; n < cnt; ++n) {
      // The lines of a tree must BE a tree: every node but the root is the child of exactly one earlier node.  (A shared child -- a DAG --
      // would walk fine, but the packers expand paths: 2^depth records from a few hundred bytes of stream.)
      if (!seen[n]) return fail(e, DDT_EINVAL, "tree %u node %llu is not the child of any earlier node", id, (unsigned long long)n);
      const uint32_t en = t[4u * n + 1u];
      if (en >> 16) return fail(e, DDT_EINVAL, "tree %u node %llu: word 1 bits [31:16] must be 0", id, (unsigned long long)n);
      if ((en & 0x7FFu) >= p->num_features)
        return fail(e, DDT_EINVAL, "tree %u node %llu: feature index %u >= num_features %u", id, (unsigned long long)n, en & 0x7FFu, p->num_features);
      if ((uint32_t)depth[n] + 1u > p->num_levels) return fail(e, DDT_EINVAL, "tree %u is deeper than num_levels %u", id, p->num_levels);
      if ((uint32_t)depth[n] + 1u > sp.max_depth) sp.max_depth = (uint32_t)depth[n] + 1u;
      for (uint32_t side = 0; side < 2; ++side) {
        const uint32_t cw = t[4u * n + 2u + side];
        if ((en >> (14u + side)) & 1u) {
          if (p->sum_mode != 1 && e->leaf_domain_check && leaf_outside_exact_domain(cw))
            return fail(e, DDT_EUNSUPPORTED,
                        "tree %u node %llu: leaf 0x%08X is not +0 or a normal value with 2^-102 <= |v| < 2^96: outside the domain where the IEEE adds are "
                        "held to the reference adder (flush to +0 when exporting, use sum_mode 1, or set option leaf_domain_check = 0)", id, (unsigned long long)n, cw);
          continue;
        }
        if (cw <= n || cw >= cnt)  // children after their parent: every walk terminates
          return fail(e, DDT_EINVAL, "tree %u node %llu: child index %u out of order / range (%llu nodes)", id, (unsigned long long)n, cw,
                      (unsigned long long)cnt);
        if (seen[cw]) return fail(e, DDT_EINVAL, "tree %u node %llu: child %u already has a parent (the lines of a tree must form a tree)", id, (unsigned long long)n, cw);
        seen[cw] = 1;
        depth[cw] = (uint8_t)(depth[n] + 1u);
      }
    }
    try {
      sp.lines.insert(sp.lines.end(), t, t + cnt * 4u);
    } catch (const std::bad_alloc&) {
      return fail(e, DDT_ENOMEM, "host model allocation failed");
    }
    sp.first.push_back(sp.first.back() + cnt);
  }
  sp.ids = std::move(ids);
  *out = std::move(sp);
  return DDT_OK;
}

}  // namespace

extern "C" int ddt_load_model_sparse_multiclass(ddt_engine* e, const ddt_params* p, const void* node_lines, size_t n_lines,
                                                const uint64_t* first, uint32_t num_classes, int interleaved, uint32_t shard_index,
                                                uint32_t shard_count) {
  if (!e) return DDT_EINVAL;
  if (!p || !node_lines || !first) return fail(e, DDT_EINVAL, "NULL argument");
  if (p->num_trees == 0) return fail(e, DDT_EINVAL, "num_trees == 0");
  if (p->num_levels < 1 || p->num_levels > 64) return fail(e, DDT_EINVAL, "num_levels %u not in 1..64 (depth bound of a sparse model)", p->num_levels);
  if (p->num_features < 1 || p->num_features > 2048) return fail(e, DDT_EINVAL, "num_features %u not in 1..2048 (DTPU.sv:72)", p->num_features);
  if (p->cmp_mode > 1 || p->sum_mode > 2) return fail(e, DDT_EINVAL, "cmp_mode %u / sum_mode %u", p->cmp_mode, p->sum_mode);
  const uint32_t c = p->clusters_per_tuple;
  if (c != 1 && c != 2 && c != 4 && c != 8) return fail(e, DDT_EINVAL, "clusters_per_tuple %u not in {1,2,4,8}", c);
  if (p->reserved[0] | p->reserved[1] | p->reserved[2]) return fail(e, DDT_EINVAL, "reserved fields must be 0");
  if (num_classes == 0 || num_classes > p->num_trees) return fail(e, DDT_EINVAL, "num_classes %u (trees %u)", num_classes, p->num_trees);
  if (!interleaved && p->num_trees % num_classes) return fail(e, DDT_EINVAL, "class-major layout needs num_trees %% num_classes == 0");
  const uint32_t per_class = (p->num_trees + num_classes - 1) / num_classes;
  if (shard_count == 0 || shard_index >= shard_count || shard_count > per_class)
    return fail(e, DDT_EINVAL, "shard %u of %u (trees per class %u)", shard_index, shard_count, per_class);
  if (first[0] != 0 || first[p->num_trees] > n_lines) return fail(e, DDT_EINVAL, "tree_first_line does not start at 0 / exceeds the stream");
  const double t0 = now_ms();
  const uint32_t* lines = reinterpret_cast<const uint32_t*>(node_lines);

  std::vector<SparseForest> sps(num_classes);
  uint64_t model_lines = 0;
  for (uint32_t k = 0; k < num_classes; ++k) {
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < p->num_trees; ++i) {
      const uint32_t cls = interleaved ? i % num_classes : i / (p->num_trees / num_classes);
      if (cls == k) ids.push_back(i);
    }
    // contiguous shards of ceil(T_class / G) trees (PCIeReceiver.sv:241-264); trailing shards may be empty: EMPTY slots, +0
    int rc = take_trees(e, p, lines, n_lines, first, shard_of(ids, shard_index, shard_count), &sps[k]);
    if (rc) return rc;
    model_lines += sps[k].lines.size() / 4u;
  }

  DeviceGuard dg(e->device);
  if (!dg.ok) return fail(e, DDT_EHIP, "hipSetDevice(%d) failed", e->device);
  HIP_TRY(e, hipDeviceSynchronize());  // asynchronous scoring of the previous model may still be in flight
  free_images(e);
  free_q16_workspace(e);
  sparse_free(e);
  e->ens.clear();
  e->loaded = false;
  e->p = *p;
  e->nint = e->nleaf = 0;
  e->num_classes = num_classes;
  e->sps = std::move(sps);
  e->sparse = true;
  int rc = sparse_rebuild(e);
  if (rc) return rc;
  e->loaded = true;
  e->st.model_lines_in += model_lines;
  e->st.prog_ms += now_ms() - t0;
  return DDT_OK;
}

extern "C" int ddt_load_model_sparse(ddt_engine* e, const ddt_params* p, const void* node_lines, size_t n_lines,
                                     const uint64_t* first, uint32_t shard_index, uint32_t shard_count) {
  return ddt_load_model_sparse_multiclass(e, p, node_lines, n_lines, first, 1, 0, shard_index, shard_count);
}

// Host-only test hook (include/ddt.h): validate + pack a sparse forest for kernel variant `variant_id` without touching a GPU.
extern "C" int ddt_debug_sparse_image(const ddt_params* p, const void* node_lines, size_t n_lines, const uint64_t* first, int variant_id,
                                      int deep_order, uint32_t* top_out, size_t top_cap_words, uint32_t* deep_out, size_t deep_cap_words,
                                      uint64_t info_out[6]) {
  if (!p || !node_lines || !first || !info_out) return DDT_EINVAL;
  if (variant_id < 0 || variant_id >= num_variants() || variant(variant_id).kind != kKindSparse) return DDT_EINVAL;
  if (p->num_trees == 0 || p->num_levels < 1 || p->num_levels > 64 || p->num_features < 1 || p->num_features > 2048) return DDT_EINVAL;
  if (first[0] != 0 || first[p->num_trees] > n_lines) return DDT_EINVAL;
  std::unique_ptr<ddt_engine> e(new (std::nothrow) ddt_engine());  // never created on a device: only p, the options and err are used
  if (!e) return DDT_ENOMEM;
  e->p = *p;
  e->sparse_deep_order = deep_order ? 1 : 0;
  std::vector<uint32_t> ids(p->num_trees);
  for (uint32_t i = 0; i < p->num_trees; ++i) ids[i] = i;
  SparseForest sp;
  int rc = take_trees(e.get(), p, reinterpret_cast<const uint32_t*>(node_lines), n_lines, first, std::move(ids), &sp);
  if (rc) return rc;
  const Variant& v = variant(variant_id);
  std::vector<uint32_t> top, deep;
  uint32_t groups = 0;
  RankTables rt;
  const bool q = (v.opt & 1) != 0;
  if (q || v.r32()) {
    try {
      rt = sparse_rank_tables(*p, {&sp});
    } catch (const std::bad_alloc&) {
      return DDT_ENOMEM;
    }
    if (rt.max_len > (v.r32() ? kSrMaxTable : e->q16_max_table)) return DDT_EUNSUPPORTED;
    if (v.r32() && (p->num_features > 256u || tuple_words(*p) > kSrMaxWords)) return DDT_EUNSUPPORTED;
  }
  rc = v.r32() ? sparse_pack_host_r(e.get(), v, sp, rt, top, deep, &groups) : sparse_pack_host(e.get(), v, sp, q ? &rt : nullptr, top, deep, &groups);
  if (rc) return rc;
  info_out[0] = top.size();
  info_out[1] = deep.size();
  info_out[2] = groups;
  info_out[3] = (uint64_t)v.levels | (v.r32() ? (uint64_t)sp.r_rounds << 32 : 0ull);
  info_out[4] = v.feat_off_sparse();
  info_out[5] = v.row_bytes_sparse();
  if (top_out) {
    if (top_cap_words < top.size()) return DDT_EINVAL;
    memcpy(top_out, top.data(), top.size() * 4u);
  }
  if (deep_out) {
    if (deep_cap_words < deep.size()) return DDT_EINVAL;
    memcpy(deep_out, deep.data(), deep.size() * 4u);
  }
  return DDT_OK;
}

// Host-only test hook (include/ddt.h): the tables of the 32-bit rank pre-pass from sorted distinct keys per tuple word.
extern "C" int ddt_debug_rank32_tables(const uint32_t* keys, const uint32_t* counts, uint32_t n_words, uint32_t* dir_out, size_t dir_cap_words,
                                       uint32_t* par_out, uint16_t* starts_out, uint32_t* tab_out, size_t tab_cap_words, uint64_t info_out[4]) {
  if (!keys || !counts || !info_out || n_words == 0 || n_words > 2048u || (n_words & 3u)) return DDT_EINVAL;
  RankTables rt;
  try {
    rt.keys.resize(n_words);
    size_t off = 0;
    for (uint32_t w = 0; w < n_words; ++w) {
      if (counts[w] > kSrMaxTable) return DDT_EUNSUPPORTED;
      rt.keys[w].assign(keys + off, keys + off + counts[w]);
      for (uint32_t i = 1; i < counts[w]; ++i)
        if (!((int32_t)rt.keys[w][i - 1] < (int32_t)rt.keys[w][i])) return DDT_EINVAL;  // sorted, distinct
      off += counts[w];
      rt.max_len = counts[w] > rt.max_len ? counts[w] : rt.max_len;
    }
  } catch (const std::bad_alloc&) {
    return DDT_ENOMEM;
  }
  std::unique_ptr<ddt_engine> e(new (std::nothrow) ddt_engine());
  if (!e) return DDT_ENOMEM;
  RankHostTables h;
  std::vector<uint32_t> tab;
  uint32_t bl = 2;
  const int rc = pack_rank32_tables(e.get(), rt, n_words, h, tab, &bl);
  if (rc) return rc;
  info_out[0] = h.Kpad;
  info_out[1] = bl;
  info_out[2] = tab.size();
  info_out[3] = 0;
  if (dir_out) {
    if (dir_cap_words < h.tab.size()) return DDT_EINVAL;
    memcpy(dir_out, h.tab.data(), h.tab.size() * 4u);
  }
  if (par_out) memcpy(par_out, h.tabK.data(), h.tabK.size() * 4u);
  if (starts_out) memcpy(starts_out, h.tabS.data(), h.tabS.size() * 2u);
  if (tab_out) {
    if (tab_cap_words < tab.size()) return DDT_EINVAL;
    memcpy(tab_out, tab.data(), tab.size() * 4u);
  }
  return DDT_OK;
}
